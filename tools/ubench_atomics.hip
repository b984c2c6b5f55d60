// Micro-benchmark: random-access primitives the relaxation kernel is built from (run on MI355X).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t rnd(uint32_t i) { uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return (uint32_t)(z >> 32); }

template <int MODE>
__global__ void k(uint64_t* a64, uint32_t* a32, const uint32_t* idx, uint32_t n_ops, uint32_t n, uint64_t* sink) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t stride = gridDim.x * blockDim.x;
  uint64_t acc = 0;
  for (; i < n_ops; i += stride) {
    uint32_t t = idx[i];
    uint64_t v = ((uint64_t)rnd(i) << 32) | i;
    if (MODE == 0) { acc += atomicMin((unsigned long long*)&a64[t], (unsigned long long)v); }          // u64 min, returned
    if (MODE == 1) { atomicMin((unsigned long long*)&a64[t], (unsigned long long)v); }                 // u64 min, no return
    if (MODE == 2) { acc += atomicMin(&a32[t], (uint32_t)(v >> 32)); }                                 // u32 min, returned
    if (MODE == 3) { atomicMin(&a32[t], (uint32_t)(v >> 32)); }                                        // u32 min, no return
    if (MODE == 4) { acc += a64[t]; }                                                                  // gather u64
    if (MODE == 5) { a64[t] = v; }                                                                     // scatter u64
    if (MODE == 6) { acc += a32[t]; }                                                                  // gather u32
    if (MODE == 7) { if (v < a64[t]) acc += atomicMin((unsigned long long*)&a64[t], (unsigned long long)v); }  // precheck + min
    if (MODE == 8) { acc += __hip_atomic_fetch_min(&a64[t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } // wg-scope
    if (MODE == 9) { acc += atomicExch(&a32[t], (uint32_t)v); }
    // XCD-private partitions: block b runs on XCD b % 8 (observed mapping); partition = bits 4..6 of the index (128-B lines)
    if (MODE == 10) { uint32_t tp = (t & ~(7u << 4)) | ((blockIdx.x & 7u) << 4); acc += __hip_atomic_fetch_min(&a64[tp], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    if (MODE == 11) { uint32_t tp = (t & ~(7u << 4)) | ((blockIdx.x & 7u) << 4); acc += atomicMin((unsigned long long*)&a64[tp], (unsigned long long)v); }
    if (MODE == 12) { uint32_t tp = (t & ~(7u << 4)) | ((blockIdx.x & 7u) << 4); acc += a64[tp]; }
    if (MODE == 13) { uint32_t tp = (t & ~(7u << 4)) | ((blockIdx.x & 7u) << 4); if (v < a64[tp]) a64[tp] = v; }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

int main() {
  const uint32_t n = 1u << 20, n_ops = 10u << 20;
  uint64_t *a64, *sink; uint32_t *a32, *idx;
  CK(hipMalloc(&a64, n * 8)); CK(hipMalloc(&a32, n * 4)); CK(hipMalloc(&idx, n_ops * 4)); CK(hipMalloc(&sink, 8));
  std::vector<uint32_t> h(n_ops);
  uint64_t s = 12345;
  for (uint32_t i = 0; i < n_ops; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = (uint32_t)(s >> 33) % n; }
  CK(hipMemcpy(idx, h.data(), n_ops * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"atomicMin u64 ret", "atomicMin u64 noret", "atomicMin u32 ret", "atomicMin u32 noret", "gather u64", "scatter u64", "gather u32", "precheck+atomicMin u64 (2nd pass: mostly fails)", "fetch_min u64 workgroup scope", "atomicExch u32 ret", "XCD-private lines: fetch_min u64 wg scope", "XCD-private lines: atomicMin u64 device scope", "XCD-private lines: gather u64", "XCD-private lines: racy check+store u64"};
  for (int mode = 0; mode < 14; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(a64, 0xFF, n * 8)); CK(hipMemset(a32, 0xFF, n * 4));
      if (mode == 7) { k<0><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      switch (mode) {
        case 0: k<0><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 1: k<1><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 2: k<2><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 3: k<3><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 4: k<4><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 5: k<5><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 6: k<6><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 7: k<7><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 8: k<8><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 9: k<9><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 10: k<10><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 11: k<11><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 12: k<12><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
        case 13: k<13><<<2048, 256>>>(a64, a32, idx, n_ops, n, sink); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("%-50s %8.3f ms  %8.2f Gops/s\n", names[mode], ms, n_ops / ms * 1e-6);
    }
  }
  return 0;
}
