// ubench_exchange.hip — what ONE level's all-to-all hand-over between resident workgroups costs on MI355X, in the geometry of
// sssp_mbox_resident_kernel (nb workgroups x 1024 threads, one per CU; region (i -> j) starts with a 64-byte sector that
// holds a 16-byte header and the first six 8-byte messages; two parity buffers).  Nothing else happens in a level: the
// figure is the floor of the exchange itself (profiles/r06_floor_model.md).
//
//   mode 0  R1, as the kernel does it today: message part of the sector (3 x 16 B, sc1) -> every wave s_waitcnt vmcnt(0)
//           -> barrier -> header (16 B, sc1) ; receiver: 4 lanes per region poll the sector with one 16-byte sc1 load each
//   mode 1  self-validating sectors: the four lanes of a destination store the WHOLE sector (header + messages) with one
//           instruction, no drain, no barrier; every 16-byte piece carries the level tag in its first word, and the receiver
//           counts sectors whose pieces disagree with the header's tag (= a torn 64-byte write or read: must be zero for
//           the data-is-the-flag form to be usable at sector granularity)
//   mode 2  summary rows: sender i stores ONE row of nb words {tag << 16 | count} (16-byte sc1 stores: nb / 4 lanes),
//           receiver j polls word j of every row (4-byte sc1 loads, one lane per sender) — 245 x 16 full sectors written per
//           level instead of 60 k partial ones; no payload travels with the flag
//   mode 3  mode 1 + a second sector per region read in the same trip (what RS_FIRST = 13 would cost while polling)
//   mode 4 / 5 / 6  sender-major "express rows": sender i owns ONE contiguous row of nb entries of 16 / 32 / 64 bytes (entry d =
//           tag, count and the first 1 / 2 / 6 messages for destination d), rewritten as a whole every level with coalesced
//           16-byte sc1 stores (nb x E bytes = full lines); receiver j reads entry j of every row (E / 16 lanes per sender)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_exchange.hip -o tools/bin/ubench_exchange
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t THREADS = 1024;

__device__ __forceinline__ uint32_t quad_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, false); }

struct Out {
  unsigned long long ticks;      // wall_clock64 ticks (100 MHz) of workgroup 0 for all levels
  unsigned long long torn;       // sectors whose pieces carried different tags
  unsigned long long polls;      // poll iterations of wave 0 of workgroup 0
  unsigned int abort_;
};

template <int MODE>
__global__ void __launch_bounds__(THREADS) xchg(unsigned char* buf0, unsigned char* buf1, uint32_t bytes, uint32_t nb, uint32_t stride,
                                               uint32_t levels, uint32_t skew_every, Out* out) {
  __shared__ uint32_t s_abort;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, j = blockIdx.x;
  const uint32_t reg = tid / 4, q = tid % 4;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)buf0, 0, (int)bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)buf1, 0, (int)bytes, 0x00020000);
  if (tid == 0) s_abort = 0;
  __syncthreads();
  // region (i -> j): sector at ((j * nb + i) * stride) bytes; summary row of sender i at i * 1024 bytes behind the regions
  const uint32_t rows = nb * nb * stride;
  unsigned long long t_begin = 0, torn = 0, polls = 0;
  for (uint32_t lvl = 1; lvl <= levels; ++lvl) {
    if (lvl == 9) t_begin = wall_clock64();  // (the first levels pay cold misses)
    const __amdgpu_buffer_rsrc_t rs = (lvl & 1u) ? rs1 : rs0;
    const uint32_t tag = lvl;
    // an uneven arrival now and then: one workgroup is late by ~2 us (what a slow expander does to everybody)
    if (skew_every && (lvl % skew_every) == 0 && j == (lvl / skew_every) % nb) __builtin_amdgcn_s_sleep(127);
    // ---------------- send
    if (MODE == 0) {
      if (reg < nb && q != 0) {
        const u32x4 m = {tag, j, reg, q};
        __builtin_amdgcn_raw_buffer_store_b128(m, rs, (int)((reg * nb + j) * stride + q * 16u), 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid < nb) {
        const u32x4 h = {tag, 6u, j, tid};
        __builtin_amdgcn_raw_buffer_store_b128(h, rs, (int)((tid * nb + j) * stride), 0, 16);
      }
    } else if (MODE == 1 || MODE == 3) {
      if (reg < nb) {
        const u32x4 m = {tag, q == 0 ? 6u : j, reg, q};
        __builtin_amdgcn_raw_buffer_store_b128(m, rs, (int)((reg * nb + j) * stride + q * 16u), 0, 16);
        if (MODE == 3) __builtin_amdgcn_raw_buffer_store_b128(m, rs, (int)((reg * nb + j) * stride + 64u + q * 16u), 0, 16);
      }
    } else if (MODE >= 4) {
      constexpr uint32_t LPE = MODE == 4 ? 1u : MODE == 5 ? 2u : 4u;  // 16-byte lanes per entry
      for (uint32_t e = tid; e < nb * LPE; e += THREADS) {
        const u32x4 m = {tag, e % LPE == 0 ? 6u : j, e / LPE, e % LPE};
        __builtin_amdgcn_raw_buffer_store_b128(m, rs, (int)(rows + j * (nb * LPE * 16u) + e * 16u), 0, 16);
      }
    } else {
      if (tid * 4u < nb) {  // four words of the row per lane
        const u32x4 m = {tag << 16 | 1u, tag << 16 | 2u, tag << 16 | 3u, tag << 16 | 4u};
        __builtin_amdgcn_raw_buffer_store_b128(m, rs, (int)(rows + j * 1024u + tid * 16u), 0, 16);
      }
    }
    // ---------------- receive
    constexpr uint32_t LPE_R = MODE == 4 ? 1u : MODE == 5 ? 2u : 4u;
    bool pend = MODE == 2 ? tid < nb : MODE >= 4 ? tid < nb * LPE_R : reg < nb;
    uint32_t spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
      if (MODE >= 4) {
        // entry j of sender (tid / LPE_R)'s row; the lanes of an entry are neighbours (a row-aligned group of 1 / 2 / 4)
        u32x4 v = {0u, 0u, 0u, 0u};
        const uint32_t snd = tid / LPE_R, piece = tid % LPE_R;
        if (pend) v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(rows + snd * (nb * LPE_R * 16u) + (j * LPE_R + piece) * 16u), 0, 16);
        uint32_t t = v.x;
        if (LPE_R == 4) t = quad_first(v.x);
        if (LPE_R == 2) t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0xA0, 0xF, 0xF, false);  // quad_perm:[0,0,2,2]
        if (pend && t == tag) {
          if (v.x != tag) torn += 1;
          pend = false;
        }
      } else if (MODE == 2) {
        uint32_t v = 0;
        if (pend) v = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(rows + tid * 1024u + j * 4u), 0, 16);
        if (pend && (v >> 16) == tag) pend = false;
      } else {
        u32x4 v = {0u, 0u, 0u, 0u}, w = {0u, 0u, 0u, 0u};
        if (pend) v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((j * nb + reg) * stride + q * 16u), 0, 16);
        if (MODE == 3 && pend) w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((j * nb + reg) * stride + 64u + q * 16u), 0, 16);
        const uint32_t t = quad_first(v.x);
        if (pend && t == tag) {
          if (MODE != 0 && v.x != tag) torn += 1;   // the header is this level's, a piece of the same sector is not
          if (MODE == 3 && w.x != tag) torn += 1ull << 32;  // (second sector: a separate request, may legitimately lag)
          pend = false;
        }
      }
      if (tid < 64) polls += 1;
      if (!__any(pend)) break;
      if ((++spins & 31u) == 0u) {
        const unsigned long long now = wall_clock64();
        if (t0 == 0) t0 = now;
        const uint32_t ab = __hip_atomic_load(&out->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ab != 0u || now - t0 > 20000000ull) {  // 0.2 s
          if (lane == 0) {
            s_abort = 1u;
            __hip_atomic_store(&out->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          break;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (s_abort) return;
  }
  const unsigned long long t_end = wall_clock64();
  for (int d = 32; d >= 1; d >>= 1) torn += __shfl_xor(torn, d);
  if (lane == 0 && torn) atomicAdd(&out->torn, torn);
  if (j == 0 && tid == 0) {
    out->ticks = t_end - t_begin;
    out->polls = polls;
  }
}

int main(int argc, char** argv) {
  const uint32_t levels = argc > 1 ? (uint32_t)atoi(argv[1]) : 4008;
  int dev_cus = 0;
  CK(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0));
  Out* out;
  CK(hipMalloc(&out, sizeof(Out)));
  const char* names[] = {"R1: messages, drain, barrier, header (today)", "whole sector in one store, no drain", "summary rows (no payload with the flag)",
                         "whole sector + second sector per poll", "express rows, 16-byte entries", "express rows, 32-byte entries",
                         "express rows, 64-byte entries"};
  for (uint32_t nb : {245u, 123u, 62u}) {
    if ((int)nb > dev_cus) continue;
    for (uint32_t stride : {1408u}) {  // regions 1.4 KB apart (C3: ~166 arcs per pair of blocks) / packed
      const size_t bytes = (size_t)nb * nb * stride + (size_t)nb * nb * 64u + 4096u;
      unsigned char *b0, *b1;
      CK(hipMalloc(&b0, bytes));
      CK(hipMalloc(&b1, bytes));
      for (uint32_t skew : {0u, 16u}) {
        for (int mode = 0; mode < 7; ++mode) {
          if (mode == 3) continue;  // (measured: 5.1-5.6 us at 245 workgroups — polling two sectors costs more than the second trip it saves)
          CK(hipMemset(b0, 0, bytes));
          CK(hipMemset(b1, 0, bytes));
          CK(hipMemset(out, 0, sizeof(Out)));
          CK(hipDeviceSynchronize());
          switch (mode) {
            case 0: xchg<0><<<nb, THREADS>>>(b0, b1, (uint32_t)bytes, nb, stride, levels, skew, out); break;
            case 1: xchg<1><<<nb, THREADS>>>(b0, b1, (uint32_t)bytes, nb, stride, levels, skew, out); break;
            case 2: xchg<2><<<nb, THREADS>>>(b0, b1, (uint32_t)bytes, nb, stride, levels, skew, out); break;
            case 3: xchg<3><<<nb, THREADS>>>(b0, b1, (uint32_t)bytes, nb, stride, levels, skew, out); break;
            case 4: xchg<4><<<nb, THREADS>>>(b0, b1, (uint32_t)bytes, nb, stride, levels, skew, out); break;
            case 5: xchg<5><<<nb, THREADS>>>(b0, b1, (uint32_t)bytes, nb, stride, levels, skew, out); break;
            case 6: xchg<6><<<nb, THREADS>>>(b0, b1, (uint32_t)bytes, nb, stride, levels, skew, out); break;
          }
          CK(hipGetLastError());
          CK(hipDeviceSynchronize());
          Out h;
          CK(hipMemcpy(&h, out, sizeof(Out), hipMemcpyDeviceToHost));
          if (h.abort_) {
            printf("nb %3u stride %4u skew %2u  %-46s ABORTED (a wait ran into its limit)\n", nb, stride, skew, names[mode]);
            continue;
          }
          const double per = (double)h.ticks * 10.0 / (levels - 8);  // ns per level (100 MHz clock)
          printf("nb %3u stride %4u skew %2u  %-46s %7.1f ns per level, %5.2f polls per level, torn sectors %llu (second sector behind: %llu)\n",
                 nb, stride, skew, names[mode], per, (double)h.polls / levels, h.torn & 0xFFFFFFFFull, h.torn >> 32);
        }
      }
      CK(hipFree(b0));
      CK(hipFree(b1));
    }
  }
  return 0;
}
