// compose_wide.h — the "wide" composition driver shared by compose_lookahead.hip (look-ahead filter stack) and
// compose_wide.hip (the ComposeFilterEnum filters): one wave per composed state of a BFS level, one launch set per level.
//
// The reference numbers composed states in first-touch order of a FIFO BFS (StateTable::find_id, lazy/state_table.rs:49-59;
// LazyFst::compute, lazy/lazy_fst.rs:226-269).  Per level [lo, hi):
//   la_emit   every state's arcs go to a segment reserved with one atomicAdd; every destination tuple (two 64-bit words)
//             is inserted into an open-addressing table together with atomicMin(position of the state in the level << 32 |
//             position of the arc in its segment) = the order of its first emission in this level;
//   la_first  a tuple is new iff it has no id yet; its first emission is the arc whose order equals the table's minimum;
//             firsts are counted per state;
//   (rocPRIM exclusive scan over the level, one 8-byte read-back: new states, overflow status)
//   la_assign firsts are numbered hi + rank: exactly the reference's ids;   la_patch  table slot -> id in the arcs.
// At the end the segments are gathered into CSR order.  No lane ever spins on another lane: a slot whose second key word is
// not written yet is retried on the next iteration of a wave-uniform loop.
//
// Everything here sits in an anonymous namespace: each translation unit instantiates its own copy with its policy.
#pragma once
#include <algorithm>
#include <cstdlib>

#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace wfst {
namespace {

constexpr uint64_t K_EMPTY = ~0ull;
constexpr uint64_t KHI_UNSET = ~0ull;  // second key word of a slot whose winner has not written it yet
constexpr uint32_t ID_UNSET = 0xFFFFFFFFu;
enum : uint32_t { LA_OK = 0, LA_OVERFLOW_STATES = 1, LA_OVERFLOW_ARCS = 2, LA_SWITCH_WIDE = 3 };
struct LaCaps {
  uint32_t S, A, H;  // composed states, composed arcs, hash slots (power of two)
};
struct Emitted {  // a composed arc and its destination tuple
  uint4 arc;
  uint64_t lo, hi;
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t* total) {
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d);
    if (lane >= (uint32_t)d) x += y;
  }
  *total = __shfl(x, 63);
  return x - v;
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, uint32_t src) {
  return ((uint64_t)(uint32_t)__shfl((uint32_t)(v >> 32), src) << 32) | (uint32_t)__shfl((uint32_t)v, src);
}
template <class T>
__device__ __forceinline__ T ld_l2(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ void st_l2(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float wtimes(float a, float b) { return a == INF ? a : (b == INF ? b : a + b); }
__device__ __forceinline__ uint32_t hash_128(uint64_t lo, uint64_t hi) {
  uint64_t h = lo ^ (hi * 0x9E3779B97F4A7C15ull);
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 33;
  return (uint32_t)h;
}
struct ArcReg {
  uint32_t il, ol;
  float w;
  uint32_t ns;
};
__device__ __forceinline__ ArcReg load_arc(const wfst_tr* p) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  return ArcReg{v.x, v.y, __uint_as_float(v.z), v.w};
}

// ---------------------------------------------------------------- wide path: one wave per composed state of a level
struct WideArena {
  uint64_t* t_lo;      // [S]
  uint64_t* t_hi;      // [S]
  uint64_t* klo;       // [H]
  uint64_t* khi;       // [H] KHI_UNSET until the slot's winner has written it
  uint64_t* hord;      // [H] min over this level's emissions of (state position in the level << 32 | arc position)
  uint32_t* hid;       // [H] state id, ID_UNSET while the tuple is new
  wfst_tr* arcs;       // [A] segments in reservation order; nextstate = table slot until la_patch
  uint64_t* a_lo;      // [A]
  uint64_t* a_hi;      // [A]
  uint32_t* seg_base;  // [S] first arc of the state's segment
  uint32_t* seg_cnt;   // [S+1]
  uint32_t* nfirst;    // [S+1] per state of the level: arcs that are the first emission of a new tuple
  uint32_t* fbase;     // [S+1] exclusive scan of nfirst
  float* fin;          // [S]
};
constexpr uint32_t MAX_PROBES = 512;  // open addressing at load <= 0.5: chains of tens at most
constexpr uint32_t CUR_SHARDS = 64;  // reservation cursors: one per 128-B line (same-address atomics serialise at ~12 ns:
constexpr uint32_t CUR_STRIDE = 32;  // 300 k states of one level on ONE cursor were 3.6 ms, the whole la_emit of that level)
struct WideCtl {
  uint32_t status;
  uint32_t pad[31];
  uint32_t cursor[CUR_SHARDS * CUR_STRIDE];  // shard j reserves inside [its start, limit[j]): a slice of the arc arena
  uint32_t limit[CUR_SHARDS];
};

__global__ void la_wide_init(WideArena ar, LaCaps caps, uint64_t lo0, uint64_t hi0, WideCtl* ctl) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t hmask = caps.H - 1;
  const uint32_t slot0 = hash_128(lo0, hi0) & hmask;
  for (uint32_t k = i; k < caps.H; k += gridDim.x * blockDim.x) {
    const bool s0 = k == slot0;
    ar.klo[k] = s0 ? lo0 : K_EMPTY;
    ar.khi[k] = s0 ? hi0 : KHI_UNSET;
    ar.hord[k] = ~0ull;
    ar.hid[k] = s0 ? 0u : ID_UNSET;
  }
  if (i == 0) {
    ar.t_lo[0] = lo0;
    ar.t_hi[0] = hi0;
    ctl->status = LA_OK;
  }
  if (i < CUR_SHARDS) {
    ctl->cursor[i * CUR_STRIDE] = i * (caps.A / CUR_SHARDS);
    ctl->limit[i] = (i + 1) * (caps.A / CUR_SHARDS);
  }
}

// after the arena has grown: an empty table gets the tuples numbered so far back (ids [0, n)), the reservation cursors
// move to the new part [a_old, a_new) of the arc arena
__global__ void la_wide_regrow(WideArena ar, LaCaps caps, uint32_t n, uint32_t a_old, WideCtl* ctl) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t hmask = caps.H - 1;
  for (uint32_t id = i; id < n; id += gridDim.x * blockDim.x) {
    const uint64_t lo = ar.t_lo[id], hi = ar.t_hi[id];
    for (uint32_t slot = hash_128(lo, hi) & hmask;; slot = (slot + 1) & hmask) {  // distinct keys: first empty slot wins
      if (atomicCAS((unsigned long long*)&ar.klo[slot], (unsigned long long)K_EMPTY, (unsigned long long)lo) == K_EMPTY) {
        ar.khi[slot] = hi;
        ar.hid[slot] = id;
        break;
      }
    }
  }
  if (i == 0) ctl->status = LA_OK;
  if (i < CUR_SHARDS) {
    const uint32_t per = (caps.A - a_old) / CUR_SHARDS;
    ctl->cursor[i * CUR_STRIDE] = a_old + i * per;
    ctl->limit[i] = a_old + (i + 1) * per;
  }
}
__global__ void la_wide_clear_table(WideArena ar, LaCaps caps) {
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < caps.H; k += gridDim.x * blockDim.x) {
    ar.klo[k] = K_EMPTY;
    ar.khi[k] = KHI_UNSET;
    ar.hord[k] = ~0ull;
    ar.hid[k] = ID_UNSET;
  }
}

// compute_trs of every state of the level [lo, hi): arcs into a reserved segment, destinations into the table.
// Policy P supplies the composition itself: P::Expand, make_expand(tuple words) and eval_item(expand, item, write, position,
// arrays, first emitted) -> number of arcs the item emits (see compose_lookahead.hip / compose_wide.hip).
template <class P>
__global__ void __launch_bounds__(256) la_emit(P pol, LaCaps caps, WideArena ar, uint32_t lo, uint32_t hi, WideCtl* ctl) {
  const uint32_t lane = lane_id();
  const uint32_t hmask = caps.H - 1;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t q = lo + wave; q < hi; q += n_waves) {
    if (ld_l2(&ctl->status) != LA_OK) return;  // the attempt is lost (arena or table overflow): the host retries bigger
    const typename P::Expand x = pol.make_expand(ar.t_lo[q], ar.t_hi[q]);
    const uint32_t n_items = x.n_it + 1;
    // size of the segment
    uint32_t cnt0 = 0, seg_total = 0;
    Emitted em0;
    for (uint32_t base = 0; base < n_items; base += 64) {
      const uint32_t j = base + lane;
      Emitted em;
      const uint32_t cnt = j < n_items ? pol.eval_item(x, j, false, 0, nullptr, nullptr, nullptr, &em) : 0u;
      if (base == 0) {
        cnt0 = cnt;
        em0 = em;
      }
      uint32_t s = cnt;
      for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
      seg_total += s;
    }
    uint32_t seg = 0;
    const uint32_t shard = wave % CUR_SHARDS;
    if (lane == 0) {
      seg = seg_total ? atomicAdd(&ctl->cursor[shard * CUR_STRIDE], seg_total) : 0u;
      ar.seg_base[q] = seg;
      ar.fin[q] = x.final_weight;
    }
    seg = __shfl(seg, 0);
    const bool fits = (uint64_t)seg + seg_total <= (uint64_t)ctl->limit[shard];
    if (lane == 0) {
      ar.seg_cnt[q] = fits ? seg_total : 0u;
      if (!fits) atomicMax(&ctl->status, (uint32_t)LA_OVERFLOW_ARCS);
    }
    if (!fits || seg_total == 0) continue;
    // the arcs, in item order
    uint32_t running = seg;
    for (uint32_t base = 0; base < n_items; base += 64) {
      const uint32_t j = base + lane;
      const bool have = j < n_items;
      // (the first chunk's counts are still in registers; states with more than 63 arcs on the iterated side recount)
      Emitted em = em0;
      const uint32_t cnt = base == 0 ? cnt0 : (have ? pol.eval_item(x, j, false, 0, nullptr, nullptr, nullptr, &em) : 0u);
      uint32_t total;
      const uint32_t pos = wave_excl_scan(cnt, lane, &total);
      if (cnt == 1) {
        *reinterpret_cast<uint4*>(ar.arcs + running + pos) = em.arc;
        ar.a_lo[running + pos] = em.lo;
        ar.a_hi[running + pos] = em.hi;
      } else if (cnt) {
        pol.eval_item(x, j, true, running + pos, ar.arcs, ar.a_lo, ar.a_hi, nullptr);
      }
      running += total;
    }
    // (the keys just written by other lanes of this wave are read back below: their stores only have to have left the CU;
    // an agent-scope fence would write back this XCD's whole L2 — thousands of waves doing that was 10x the kernel)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // destinations: slot of the tuple + the order of its first emission in this level
    for (uint32_t base = 0; base < seg_total; base += 64) {
      const uint32_t k = base + lane;
      const bool have = k < seg_total;
      uint64_t klo = K_EMPTY, khi = 0;
      if (have) {
        klo = ld_l2(&ar.a_lo[seg + k]);
        khi = ld_l2(&ar.a_hi[seg + k]);
      }
      uint32_t slot = hash_128(klo, khi) & hmask;
      bool done = !have;
      uint32_t probes = 0;
      while (__any(!done)) {  // (no lane ever spins on another: a slot whose second word is not there yet is retried)
        uint64_t prev = 0;
        if (!done) prev = atomicCAS((unsigned long long*)&ar.klo[slot], (unsigned long long)K_EMPTY, (unsigned long long)klo);
        const bool won = !done && prev == K_EMPTY;
        if (won) st_l2(&ar.khi[slot], khi);  // (a lane that reads it too early sees KHI_UNSET and retries the slot)
        const bool same_lo = !done && !won && prev == klo;
        uint64_t h = KHI_UNSET;
        if (same_lo) h = ld_l2(&ar.khi[slot]);
        if (won || (same_lo && h == khi)) done = true;
        else if (!done && !(same_lo && h == KHI_UNSET)) {
          slot = (slot + 1) & hmask;
          // a level can hold more new tuples than the table has room for (the state count is only checked between
          // levels): a probe sequence this long means the table is filling up -> give the attempt up, never spin
          if (++probes > MAX_PROBES) {
            atomicMax(&ctl->status, (uint32_t)LA_OVERFLOW_STATES);
            done = true;
          }
        }
      }
      if (have) {
        atomicMin((unsigned long long*)&ar.hord[slot], ((unsigned long long)(q - lo) << 32) | k);
        ar.arcs[seg + k].nextstate = slot;
      }
    }
  }
}

// per state of the level: how many of its arcs are the first emission of a tuple that has no id yet
__global__ void __launch_bounds__(256) la_first(WideArena ar, uint32_t lo, uint32_t hi, const WideCtl* ctl) {
  if (ld_l2(&ctl->status) != LA_OK) return;  // a lost attempt left segments unwritten: nothing here is valid
  const uint32_t lane = lane_id();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) ar.nfirst[hi - lo] = 0;  // the scan's extra element: fbase[hi - lo] = total
  for (uint32_t q = lo + wave; q < hi; q += n_waves) {
    const uint32_t seg = ar.seg_base[q], n = ar.seg_cnt[q];
    uint32_t c = 0;
    for (uint32_t base = 0; base < n; base += 64) {
      const uint32_t k = base + lane;
      bool first = false;
      if (k < n) {
        const uint32_t slot = ar.arcs[seg + k].nextstate;
        first = ld_l2(&ar.hid[slot]) == ID_UNSET && ld_l2(&ar.hord[slot]) == (((uint64_t)(q - lo) << 32) | k);
      }
      c += (uint32_t)__popcll(__ballot(first));
    }
    if (lane == 0) ar.nfirst[q - lo] = c;
  }
}

// numbers the new tuples in emission order: id = id_base + firsts before it (StateTable::find_id, state_table.rs:49-59)
__global__ void __launch_bounds__(256) la_assign(WideArena ar, uint32_t lo, uint32_t hi, uint32_t id_base) {
  const uint32_t lane = lane_id();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t q = lo + wave; q < hi; q += n_waves) {
    const uint32_t seg = ar.seg_base[q], n = ar.seg_cnt[q];
    uint32_t next = id_base + ar.fbase[q - lo];
    for (uint32_t base = 0; base < n; base += 64) {
      const uint32_t k = base + lane;
      bool first = false;
      uint32_t slot = 0;
      if (k < n) {
        slot = ar.arcs[seg + k].nextstate;
        // same predicate as la_first: only the one arc whose order the table kept can pass it for a new tuple, and only
        // its lane writes that tuple's id, so the ids written by other waves meanwhile do not disturb it
        first = ld_l2(&ar.hid[slot]) == ID_UNSET && ld_l2(&ar.hord[slot]) == (((uint64_t)(q - lo) << 32) | k);
      }
      const uint64_t m = __ballot(first);
      if (first) {
        const uint32_t id = next + lanes_below(m);
        st_l2(&ar.hid[slot], id);
        ar.t_lo[id] = ld_l2(&ar.klo[slot]);
        ar.t_hi[id] = ld_l2(&ar.khi[slot]);
      }
      next += (uint32_t)__popcll(m);
    }
  }
}

__global__ void __launch_bounds__(256) la_patch(WideArena ar, uint32_t lo, uint32_t hi) {
  const uint32_t lane = lane_id();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t q = lo + wave; q < hi; q += n_waves) {
    const uint32_t seg = ar.seg_base[q], n = ar.seg_cnt[q];
    for (uint32_t k = lane; k < n; k += 64) ar.arcs[seg + k].nextstate = ld_l2(&ar.hid[ar.arcs[seg + k].nextstate]);
  }
}

// segments -> CSR order
__global__ void __launch_bounds__(256) la_gather(WideArena ar, const uint32_t* __restrict__ off, wfst_tr* __restrict__ out,
                                                 uint32_t n_states) {
  const uint32_t lane = lane_id();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t q = wave; q < n_states; q += n_waves) {
    const uint32_t seg = ar.seg_base[q], n = ar.seg_cnt[q], o = off[q];
    for (uint32_t k = lane; k < n; k += 64) *reinterpret_cast<uint4*>(out + o + k) = *reinterpret_cast<const uint4*>(ar.arcs + seg + k);
  }
}


inline size_t wide_al16(size_t x) { return (x + 15) & ~(size_t)15; }
inline uint32_t wide_next_pow2(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return (uint32_t)p;
}

// the finished composition, still in the arena of the run (CSR order: off / arcs / fin)
struct WideOutput {
  DBuf<char> arena;
  uint32_t n_states = 0, n_arcs = 0, n_levels = 0;
  const uint32_t* off = nullptr;
  const wfst_tr* arcs = nullptr;
  const float* fin = nullptr;
};

// runs the level loop for policy `pol` from the start tuple (lo0, hi0).  When a level does not fit (states, arcs or a
// table filling up) the arena GROWS (x4, x2 beyond a million states) and the search goes on from that level: the numbered tuples, their
// segments and final weights are copied, the table is rebuilt from the ids, the level is emitted again — nothing that
// was finished is redone (restarting from scratch cost up to one full composition per overflow).
struct WideBuffers {
  DBuf<char> arena;
  WideArena ar{};
  uint32_t* d_off = nullptr;
  wfst_tr* d_out = nullptr;
  LaCaps caps{};
};
inline void wide_alloc(wfst_ctx* ctx, uint64_t est_s, uint64_t est_a, WideBuffers& w) {
  if (est_s > 0x7FFFFFF0ull || est_a > 0x7FFFFFF0ull) throw Error("compose: composition too large");
  const LaCaps caps{(uint32_t)est_s, (uint32_t)est_a, wide_next_pow2(2 * est_s + 128)};
  size_t bytes = 0;
  auto take = [&](size_t n) {
    const size_t o = bytes;
    bytes += wide_al16(n);
    return o;
  };
  const size_t o_tlo = take((size_t)caps.S * 8), o_thi = take((size_t)caps.S * 8), o_klo = take((size_t)caps.H * 8),
               o_khi = take((size_t)caps.H * 8), o_hord = take((size_t)caps.H * 8), o_hid = take((size_t)caps.H * 4),
               o_arcs = take((size_t)caps.A * 16), o_alo = take((size_t)caps.A * 8), o_ahi = take((size_t)caps.A * 8),
               o_sb = take((size_t)caps.S * 4), o_sc = take(((size_t)caps.S + 1) * 4), o_nf = take(((size_t)caps.S + 1) * 4),
               o_fb = take(((size_t)caps.S + 1) * 4), o_fin = take((size_t)caps.S * 4), o_off = take(((size_t)caps.S + 1) * 4),
               o_out = take((size_t)caps.A * 16);
  w.arena = DBuf<char>(*ctx->pool, bytes);
  char* b = w.arena.p;
  w.ar = WideArena{(uint64_t*)(b + o_tlo), (uint64_t*)(b + o_thi), (uint64_t*)(b + o_klo),  (uint64_t*)(b + o_khi),
                   (uint64_t*)(b + o_hord), (uint32_t*)(b + o_hid), (wfst_tr*)(b + o_arcs), (uint64_t*)(b + o_alo),
                   (uint64_t*)(b + o_ahi), (uint32_t*)(b + o_sb),  (uint32_t*)(b + o_sc),  (uint32_t*)(b + o_nf),
                   (uint32_t*)(b + o_fb),  (float*)(b + o_fin)};
  w.d_off = (uint32_t*)(b + o_off);
  w.d_out = (wfst_tr*)(b + o_out);
  w.caps = caps;
}

template <class P>
void run_wide(wfst_ctx* ctx, const P& pol, uint64_t lo0, uint64_t hi0, uint64_t est_s, uint64_t est_a, WideOutput& out) {
  hipStream_t st = ctx->stream;
  if (const char* e = std::getenv("WFST_WIDE_EST_STATES")) {  // tests: start from a tiny arena so that it has to grow
    est_s = std::max<uint64_t>(64, (uint64_t)std::atoll(e));
    est_a = 4 * est_s;
  } else {
    // whoever calls the wide driver has already seen the result outgrow one wave: room for a quarter of a million states
    // to begin with (clearing that table takes ~20 us)
    est_s = std::max<uint64_t>(est_s, 1ull << 18);
    est_a = std::max<uint64_t>(est_a, 4 * est_s);
  }
  WideBuffers w;
  wide_alloc(ctx, est_s, est_a, w);
  DBuf<WideCtl> d_ctl(*ctx->pool, 1);
  size_t temp_bytes = 0;
  HIP_CHECK(rocprim::exclusive_scan(nullptr, temp_bytes, w.ar.nfirst, w.ar.fbase, 0u, (size_t)0x7FFFFFF0u, rocprim::plus<uint32_t>(), st));
  DBuf<uint8_t> temp(*ctx->pool, temp_bytes);
  struct HostCtl {
    uint32_t status, n_new, n_arcs;
  };
  HostCtl* hc = (HostCtl*)ctx->pinned.get(sizeof(HostCtl));
  const uint32_t max_blocks = (uint32_t)ctx->n_cus * 8;
  la_wide_init<<<std::min<uint32_t>(max_blocks, (w.caps.H + 255) / 256), 256, 0, st>>>(w.ar, w.caps, lo0, hi0, d_ctl.p);
  uint32_t lo = 0, hi = 1, levels = 0;
  int grows = 0;
  while (lo < hi) {  // LazyFst::compute, lazy_fst.rs:235-259: level = ids [lo, hi)
    const uint32_t n_level = hi - lo;
    const uint32_t blocks = std::min<uint32_t>(max_blocks, (n_level + 3) / 4);
    la_emit<P><<<blocks, 256, 0, st>>>(pol, w.caps, w.ar, lo, hi, d_ctl.p);
    la_first<<<blocks, 256, 0, st>>>(w.ar, lo, hi, d_ctl.p);
    HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, w.ar.nfirst, w.ar.fbase, 0u, (size_t)n_level + 1, rocprim::plus<uint32_t>(), st));
    HIP_CHECK(hipMemcpyAsync(&hc->status, &d_ctl.p->status, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(&hc->n_new, w.ar.fbase + n_level, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (hc->status != LA_OK || (uint64_t)hi + hc->n_new > w.caps.S) {
      // the level does not fit: four times the room, everything numbered so far moves over, the level is emitted again
      ctx->stats.compose_retries++;
      if (++grows > 24) throw Error("compose: arena overflow after retries");
      WideBuffers nw;
      // (growing is cheap now, an over-sized table is not: its random accesses leave the caches — x2 once it is large)
      const uint64_t g = w.caps.S >= (1u << 20) ? 2 : 4;
      wide_alloc(ctx, g * w.caps.S, g * w.caps.A, nw);
      auto copy = [&](void* d, const void* s_, size_t n) { HIP_CHECK(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, st)); };
      copy(nw.ar.t_lo, w.ar.t_lo, (size_t)hi * 8);
      copy(nw.ar.t_hi, w.ar.t_hi, (size_t)hi * 8);
      copy(nw.ar.seg_base, w.ar.seg_base, (size_t)lo * 4);
      copy(nw.ar.seg_cnt, w.ar.seg_cnt, (size_t)lo * 4);
      copy(nw.ar.fin, w.ar.fin, (size_t)lo * 4);
      copy(nw.ar.arcs, w.ar.arcs, (size_t)w.caps.A * 16);  // (finished segments keep their positions; their tuple words
      la_wide_clear_table<<<std::min<uint32_t>(max_blocks, (nw.caps.H + 255) / 256), 256, 0, st>>>(nw.ar, nw.caps);  //  are dead)
      la_wide_regrow<<<std::min<uint32_t>(max_blocks, (hi + 255) / 256), 256, 0, st>>>(nw.ar, nw.caps, hi, w.caps.A, d_ctl.p);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(st));  // the old arena goes back to the pool
      w = std::move(nw);
      continue;  // the same level again
    }
    levels++;
    la_assign<<<blocks, 256, 0, st>>>(w.ar, lo, hi, hi);
    la_patch<<<blocks, 256, 0, st>>>(w.ar, lo, hi);
    lo = hi;
    hi += hc->n_new;
  }
  HIP_CHECK(hipGetLastError());
  const uint32_t n_states = hi;
  HIP_CHECK(hipMemsetAsync(w.ar.seg_cnt + n_states, 0, sizeof(uint32_t), st));
  HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, w.ar.seg_cnt, w.d_off, 0u, (size_t)n_states + 1, rocprim::plus<uint32_t>(), st));
  la_gather<<<std::min<uint32_t>(max_blocks, (n_states + 3) / 4), 256, 0, st>>>(w.ar, w.d_off, w.d_out, n_states);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(&hc->n_arcs, w.d_off + n_states, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  out.n_states = n_states;
  out.n_arcs = hc->n_arcs;
  out.n_levels = levels;
  out.off = w.d_off;
  out.arcs = w.d_out;
  out.fin = w.ar.fin;
  out.arena = std::move(w.arena);
}

}  // namespace
}  // namespace wfst
