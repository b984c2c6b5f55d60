// compose_wide.h — the "wide" composition driver shared by compose_lookahead.hip (look-ahead filter stack) and
// compose_wide.hip (the ComposeFilterEnum filters): G lanes per composed state of a BFS level (G = 8 / 16 / 64), three
// launches per level, level ranges and overflow status in a control block on the device.
//
// The reference numbers composed states in first-touch order of a FIFO BFS (StateTable::find_id, lazy/state_table.rs:49-59;
// LazyFst::compute, lazy/lazy_fst.rs:226-269).  Per level k (ids [lo, hi) = WideCtl::lvl[k]):
//   la_emit   every state's arcs go to a segment reserved with one atomicAdd; every destination tuple (two 64-bit words)
//             is inserted into an open-addressing table together with atomicMin(position of the state in the level << 32 |
//             position of the arc in its segment) = the order of its first emission in this level;
//   la_first  a tuple is new iff it has no id yet; its first emission is the arc whose order equals the table's minimum;
//             every block counts the firsts of a contiguous share of the level;
//   la_assign sums the block counts (overflow check, next level's range), numbers the firsts of its share hi + rank:
//             exactly the reference's ids.  Arcs keep table slots as destinations until la_gather (or la_patch_range
//             before a growth drops the table).
// The host queues several levels per look at the control block (run_wide_g) and grows the arena in place, ahead of the
// level that would not fit.  At the end the segments are gathered into CSR order.  No lane ever spins on another lane: a
// slot whose second key word is not written yet is retried on the next iteration of a wave-uniform loop.
//
// Everything here sits in an anonymous namespace: each translation unit instantiates its own copy with its policy.
#pragma once
#include <algorithm>
#include <cstdio>
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

// ---------------------------------------------------------------- wide path: G lanes per composed state of a level
struct WideArena {
  uint64_t* t_lo;      // [S]
  uint64_t* t_hi;      // [S]
  uint64_t* klo;       // [H]
  uint64_t* khi;       // [H] KHI_UNSET until the slot's winner has written it
  uint64_t* hord;      // [H] min over this level's emissions of (state position in the level << 32 | arc position)
  uint32_t* hid;       // [H] state id, ID_UNSET while the tuple is new
  wfst_tr* arcs;       // [A] segments in reservation order; nextstate = table slot until la_gather / la_patch_range
  uint64_t* a_lo;      // [A]
  uint64_t* a_hi;      // [A]
  uint32_t* seg_base;  // [S] first arc of the state's segment
  uint32_t* seg_cnt;   // [S+1]
  float* fin;          // [S]
};
constexpr uint32_t MAX_PROBES = 512;  // open addressing at load <= 0.5: chains of tens at most
constexpr uint32_t CUR_SHARDS = 64;  // reservation cursors: one per 128-B line (same-address atomics serialise at ~12 ns:
constexpr uint32_t CUR_STRIDE = 32;  // 300 k states of one level on ONE cursor were 3.6 ms, the whole la_emit of that level)
constexpr uint32_t LVL_RING = 64;    // level ranges live on the device: the host queues several levels per look
constexpr uint32_t WIDE_MAX_BLOCKS = 4096;
constexpr uint32_t WIDE_STAGE_MAX = 16;  // arcs of a state's searched side staged in LDS by la_emit
struct WideCtl {
  uint32_t status;
  uint32_t k_done;  // levels finished so far; lvl[k_done % LVL_RING] is the next one (lo == hi: the search is over)
  uint32_t slack_min, arcs_used;  // after the last finished level: least room left in a shard's slice, arcs reserved in all
  uint32_t per;                   // size of a shard's slice of the current arena part
  uint32_t pad[27];
  uint32_t lvl[LVL_RING][2];                  // ids [lo, hi) of level k at k % LVL_RING, written by la_assign of level k-1
  uint32_t cursor[CUR_SHARDS * CUR_STRIDE];  // shard j reserves inside [its start, limit[j]): a slice of the arc arena
  uint32_t limit[CUR_SHARDS];
  uint32_t bsum[WIDE_MAX_BLOCKS];  // la_first: new tuples first emitted by the states of block b's share of the level
};
constexpr size_t WIDE_CTL_HEAD = offsetof(WideCtl, cursor);  // what the host reads back per look

template <uint32_t G>
__device__ __forceinline__ uint32_t group_sum(uint32_t v) {
#pragma unroll
  for (int d = (int)G / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
template <uint32_t G>
__device__ __forceinline__ uint32_t group_excl_scan(uint32_t v, uint32_t sub, uint32_t* total) {
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < (int)G; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, G);
    if (sub >= (uint32_t)d) x += y;
  }
  *total = __shfl(x, G - 1, G);
  return x - v;
}
template <uint32_t G>
__device__ __forceinline__ uint64_t group_mask(uint64_t ballot, uint32_t grp) {
  if constexpr (G == 64) return ballot;
  else return (ballot >> (grp * G)) & ((1ull << G) - 1ull);
}

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
    ctl->k_done = 0;
    ctl->lvl[0][0] = 0;
    ctl->lvl[0][1] = 1;
    ctl->per = caps.A / CUR_SHARDS;
    ctl->slack_min = caps.A / CUR_SHARDS;
    ctl->arcs_used = 0;
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
  if (i == 0) {
    ctl->status = LA_OK;
    ctl->per = (caps.A - a_old) / CUR_SHARDS;
    ctl->slack_min = (caps.A - a_old) / CUR_SHARDS;
    ctl->arcs_used = 0;
  }
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
// table slot -> state id in the arcs of the finished states [q_lo, q_hi), before the table they point into is dropped
__global__ void __launch_bounds__(256) la_patch_range(WideArena ar, uint32_t q_lo, uint32_t q_hi) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, sub = t & 7u;
  for (uint32_t q = q_lo + (t >> 3); q < q_hi; q += (gridDim.x * blockDim.x) >> 3) {
    const uint32_t seg = ar.seg_base[q], n = ar.seg_cnt[q];
    for (uint32_t k = sub; k < n; k += 8) ar.arcs[seg + k].nextstate = ar.hid[ar.arcs[seg + k].nextstate];
  }
}

// compute_trs of every state of level `level` (ids [lo, hi) from the control block): arcs into a reserved segment,
// destinations into the table.  G lanes share a composed state (64 / G states per wave: a state of the recipes this serves
// has a handful of arcs on its iterated side, and a lone wave per state left most lanes — and most of the memory-level
// parallelism — idle); loops run to the longest trip count among the wave's groups so that shuffles stay wave-uniform.
// Policy P supplies the composition itself: P::Expand, make_expand(tuple words) and eval_item(expand, item, write, position,
// arrays, first emitted) -> number of arcs the item emits (see compose_lookahead.hip / compose_wide.hip).
#ifdef WFST_WIDE_WAVES_PER_EU  // experiments: cap the registers of la_emit for more waves per SIMD
#define WFST_WIDE_EMIT_ATTR __attribute__((amdgpu_waves_per_eu(WFST_WIDE_WAVES_PER_EU, WFST_WIDE_WAVES_PER_EU)))
#else
#define WFST_WIDE_EMIT_ATTR
#endif
template <class P, uint32_t G>
__global__ void __launch_bounds__(256) WFST_WIDE_EMIT_ATTR la_emit(P pol, LaCaps caps, WideArena ar, uint32_t level, WideCtl* ctl) {
  constexpr uint32_t SPW = 64 / G;
  __shared__ uint4 s_stage[(256 / G) * WIDE_STAGE_MAX];  // the searched side's arcs of the states the block works on
  const uint32_t lo = ctl->lvl[level % LVL_RING][0], hi = ctl->lvl[level % LVL_RING][1];
  if (lo >= hi) return;
  const uint32_t lane = lane_id(), sub = lane % G, grp = lane / G;
  const uint32_t hmask = caps.H - 1;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t q0 = lo + wave * SPW; q0 < hi; q0 += n_waves * SPW) {
    if (ld_l2(&ctl->status) != LA_OK) return;  // the attempt is lost (arena or table overflow): the host grows the arena
    const uint32_t q = q0 + grp;
    const bool valid = q < hi;
    const uint32_t qc = valid ? q : hi - 1;
    typename P::Expand x = pol.make_expand(ar.t_lo[qc], ar.t_hi[qc]);
    // The searched side's arcs go to LDS, one trip for the group: every item is matched against them by binary searches
    // (equal_range: ~8 dependent probes per item), and a probe of a 16-byte arc in memory is a round trip of its own — a
    // composed state of the look-ahead recipe (10 arcs searched, 2 items) walked ~20 of them one after the other, which was
    // most of this kernel (profiles/r05c_wide_lookahead.md).  From LDS the searches cost what their instructions cost.
    if (x.n_se <= WIDE_STAGE_MAX) {
      uint4* const stg = s_stage + (threadIdx.x / G) * WIDE_STAGE_MAX;
      for (uint32_t t = sub; t < x.n_se; t += G) stg[t] = *reinterpret_cast<const uint4*>(x.se_arcs + t);
      x.se_arcs = reinterpret_cast<const wfst_tr*>(stg);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (a group's slot is private to it: the wave's LDS accesses are ordered)
    const uint32_t n_items = valid ? x.n_it + 1 : 0u;
    // size of the segment
    uint32_t cnt0 = 0, seg_total = 0;
    Emitted em0;
    for (uint32_t base = 0; __any(base < n_items); base += G) {
      const uint32_t j = base + sub;
      Emitted em;
      const uint32_t cnt = j < n_items ? pol.eval_item(x, j, false, 0, nullptr, nullptr, nullptr, &em) : 0u;
      if (base == 0) {
        cnt0 = cnt;
        em0 = em;
      }
      seg_total += group_sum<G>(cnt);
    }
    // one reservation per WAVE (its 64 / G states): the cursors are 64 words, and same-address atomics serialise at ~12 ns —
    // one per composed state was 90 M of them in a 90 M-state run (profiles/r05c_wide_lookahead.md)
    uint32_t wave_total = 0, before = 0;
#pragma unroll
    for (uint32_t g2 = 0; g2 < SPW; ++g2) {
      const uint32_t v = (uint32_t)__shfl((int)(valid ? seg_total : 0u), (int)(g2 * G));
      before += g2 < grp ? v : 0u;
      wave_total += v;
    }
    const uint32_t shard = wave % CUR_SHARDS;
    uint32_t wbase = 0;
    if (lane == 0 && wave_total) wbase = atomicAdd(&ctl->cursor[shard * CUR_STRIDE], wave_total);
    wbase = (uint32_t)__shfl((int)wbase, 0);
    const uint32_t seg = wbase + before;
    const bool fits = (uint64_t)wbase + wave_total <= (uint64_t)ctl->limit[shard];
    if (sub == 0 && valid) {
      ar.seg_base[q] = seg;
      ar.fin[q] = x.final_weight;
      ar.seg_cnt[q] = fits ? seg_total : 0u;
      if (!fits) atomicMax(&ctl->status, (uint32_t)LA_OVERFLOW_ARCS);
    }
    const bool live = valid && fits && seg_total != 0;
    // the arcs, in item order
    uint32_t running = seg;
    for (uint32_t base = 0; __any(live && base < n_items); base += G) {
      const uint32_t j = base + sub;
      const bool have = live && j < n_items;
      // (the first chunk's counts are still in registers; states with more than G - 1 arcs on the iterated side recount)
      Emitted em = em0;
      uint32_t cnt = 0;
      if (have) cnt = base == 0 ? cnt0 : pol.eval_item(x, j, false, 0, nullptr, nullptr, nullptr, &em);
      uint32_t total;
      const uint32_t pos = group_excl_scan<G>(cnt, sub, &total);
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
    for (uint32_t base = 0; __any(live && base < seg_total); base += G) {
      const uint32_t k = base + sub;
      const bool have = live && k < seg_total;
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
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the staged arcs are read before the next states' overwrite them)
  }
}

// the share of a level that block b of a grid of nb blocks numbers: consecutive states, so that ids come out in level order
__device__ __forceinline__ void block_share(uint32_t lo, uint32_t hi, uint32_t* b_lo, uint32_t* b_hi) {
  const uint32_t n = hi - lo, chunk = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t a = (uint64_t)lo + (uint64_t)blockIdx.x * chunk;
  *b_lo = (uint32_t)min(a, (uint64_t)hi);
  *b_hi = (uint32_t)min(a + chunk, (uint64_t)hi);
}
// a tuple is new iff it has no id yet; its first emission is the arc whose order equals the table's minimum
__device__ __forceinline__ bool is_first_emission(const WideArena& ar, uint32_t slot, uint32_t pos_in_level, uint32_t k) {
  return ld_l2(&ar.hid[slot]) == ID_UNSET && ld_l2(&ar.hord[slot]) == (((uint64_t)pos_in_level << 32) | k);
}

// per block: how many arcs of its share of the level are the first emission of a tuple that has no id yet
template <uint32_t G>
__global__ void __launch_bounds__(256) la_first(WideArena ar, uint32_t level, WideCtl* ctl) {
  constexpr uint32_t SPW = 64 / G;
  __shared__ uint32_t s_part[4];
  const uint32_t lo = ctl->lvl[level % LVL_RING][0], hi = ctl->lvl[level % LVL_RING][1];
  if (lo >= hi || ctl->status != LA_OK) return;  // (a lost attempt left segments unwritten: nothing here is valid)
  uint32_t b_lo, b_hi;
  block_share(lo, hi, &b_lo, &b_hi);
  const uint32_t lane = lane_id(), sub = lane % G, grp = lane / G, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  uint32_t c = 0;
  for (uint32_t q = b_lo + wv * SPW + grp; q < b_hi; q += nwv * SPW) {
    const uint32_t seg = ar.seg_base[q], n = ar.seg_cnt[q];
    for (uint32_t k = sub; k < n; k += G) c += is_first_emission(ar, ar.arcs[seg + k].nextstate, q - lo, k) ? 1u : 0u;
  }
  c = group_sum<64>(c);
  if (lane == 0) s_part[wv] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (uint32_t i = 0; i < nwv; ++i) t += s_part[i];
    ctl->bsum[blockIdx.x] = t;
  }
}

// numbers the new tuples in emission order: id = hi + firsts before it (StateTable::find_id, state_table.rs:49-59), and
// publishes the next level's range.  Same grid as la_first.
template <uint32_t G>
__global__ void __launch_bounds__(256) la_assign(WideArena ar, LaCaps caps, uint32_t level, WideCtl* ctl) {
  constexpr uint32_t SPW = 64 / G;
  static_assert(4u * (64u / G) <= 128u, "la_assign numbers a tile of (waves per block) x (states per wave) states with two 64-lane scans: G >= 2");
  __shared__ uint32_t s_pre[4], s_tot[4], s_cnt[128], s_status;
  const uint32_t lo = ctl->lvl[level % LVL_RING][0], hi = ctl->lvl[level % LVL_RING][1];
  if (lo >= hi) {  // the search ended before this level: the levels queued behind it must see an empty range too
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ctl->lvl[(level + 1) % LVL_RING][0] = hi;
      ctl->lvl[(level + 1) % LVL_RING][1] = hi;
    }
    return;
  }
  // The status is read ONCE per workgroup: block 0 of this very launch may raise it (overflow) while the others are
  // between the two barriers below — waves of one workgroup that disagreed would sum uninitialised shares.
  if (threadIdx.x == 0) s_status = ctl->status;
  __syncthreads();
  if (s_status != LA_OK) return;
  const uint32_t lane = lane_id(), sub = lane % G, grp = lane / G, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  uint32_t pre = 0, tot = 0;
  for (uint32_t i = threadIdx.x; i < gridDim.x; i += blockDim.x) {
    const uint32_t v = ctl->bsum[i];
    tot += v;
    pre += i < blockIdx.x ? v : 0u;
  }
  pre = group_sum<64>(pre);
  tot = group_sum<64>(tot);
  if (lane == 0) {
    s_pre[wv] = pre;
    s_tot[wv] = tot;
  }
  __syncthreads();
  pre = tot = 0;
  for (uint32_t i = 0; i < nwv; ++i) {
    pre += s_pre[i];
    tot += s_tot[i];
  }
  if ((uint64_t)hi + tot > caps.S) {  // the level does not fit: nothing is numbered, the host grows the arena
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&ctl->status, (uint32_t)LA_OVERFLOW_STATES);
    return;
  }
  uint32_t b_lo, b_hi;
  block_share(lo, hi, &b_lo, &b_hi);
  const uint32_t T = nwv * SPW, me = wv * SPW + grp;  // states per tile, my state's place in it
  uint32_t running = hi + pre;
  for (uint32_t t0 = b_lo; t0 < b_hi; t0 += T) {
    const uint32_t q = t0 + me;
    const bool valid = q < b_hi;
    const uint32_t seg = valid ? ar.seg_base[q] : 0u, n = valid ? ar.seg_cnt[q] : 0u;
    uint32_t c = 0;
    for (uint32_t k = sub; k < n; k += G) c += is_first_emission(ar, ar.arcs[seg + k].nextstate, q - lo, k) ? 1u : 0u;
    c = group_sum<G>(c);
    if (sub == 0) s_cnt[me] = c;
    __syncthreads();
    // (a tile is at most 128 states: two 64-lane scans)
    const uint32_t v = lane < T ? s_cnt[lane] : 0u, v2 = 64u + lane < T ? s_cnt[64u + lane] : 0u;
    uint32_t tile_total, total2;
    const uint32_t excl = group_excl_scan<64>(v, lane, &tile_total);
    const uint32_t excl2 = tile_total + group_excl_scan<64>(v2, lane, &total2);
    tile_total += total2;
    const uint32_t e_lo = __shfl(excl, me & 63u), e_hi = __shfl(excl2, me & 63u);
    uint32_t next = running + (me < 64u ? e_lo : e_hi);
    __syncthreads();
    // same predicate as the count: only the one arc whose order the table kept can pass it for a new tuple, and only its
    // lane writes that tuple's id, so the ids written by other waves meanwhile do not disturb it
    for (uint32_t base = 0; __any(base < n); base += G) {
      const uint32_t k = base + sub;
      bool first = false;
      uint32_t slot = 0;
      if (k < n) {
        slot = ar.arcs[seg + k].nextstate;
        first = is_first_emission(ar, slot, q - lo, k);
      }
      const uint64_t m = group_mask<G>(__ballot(first), grp);
      if (first) {
        const uint32_t id = next + (uint32_t)__popcll(m & ((1ull << sub) - 1ull));
        st_l2(&ar.hid[slot], id);
        ar.t_lo[id] = ld_l2(&ar.klo[slot]);
        ar.t_hi[id] = ld_l2(&ar.khi[slot]);
      }
      next += (uint32_t)__popcll(m);
    }
    running += tile_total;
  }
  if (blockIdx.x == 0 && wv == 0) {
    // how full the arc slices are (the host grows the arena BEFORE a level that is unlikely to fit: an emission that
    // overflows is thrown away)
    static_assert(CUR_SHARDS == 64, "one lane per shard");
    const uint32_t cur = ctl->cursor[lane * CUR_STRIDE], lim = ctl->limit[lane], per = ctl->per;
    uint32_t slack = lim > cur ? lim - cur : 0u, used = min(cur, lim) - (lim - per);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      slack = min(slack, (uint32_t)__shfl_xor(slack, d));
      used += __shfl_xor(used, d);
    }
    if (lane == 0) {
      ctl->slack_min = slack;
      ctl->arcs_used = used;
      ctl->lvl[(level + 1) % LVL_RING][0] = hi;
      ctl->lvl[(level + 1) % LVL_RING][1] = hi + tot;
      ctl->k_done = level + 1;
    }
  }
}

// segments -> CSR order; states from `patch_from` on still carry table slots in their arcs
template <uint32_t G>
__global__ void __launch_bounds__(256) la_gather(WideArena ar, const uint32_t* __restrict__ off, wfst_tr* __restrict__ out,
                                                 uint32_t n_states, uint32_t patch_from) {
  constexpr uint32_t SPW = 64 / G;
  const uint32_t lane = lane_id(), sub = lane % G, grp = lane / G;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t q = wave * SPW + grp; q < n_states; q += n_waves * SPW) {
    const uint32_t seg = ar.seg_base[q], n = ar.seg_cnt[q], o = off[q];
    for (uint32_t k = sub; k < n; k += G) {
      uint4 a = *reinterpret_cast<const uint4*>(ar.arcs + seg + k);
      if (q >= patch_from) a.w = ar.hid[a.w];
      *reinterpret_cast<uint4*>(out + o + k) = a;
    }
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
               o_sb = take((size_t)caps.S * 4), o_sc = take(((size_t)caps.S + 1) * 4), o_fin = take((size_t)caps.S * 4),
               o_off = take(((size_t)caps.S + 1) * 4), o_out = take((size_t)caps.A * 16);
  w.arena = DBuf<char>(*ctx->pool, bytes);
  char* b = w.arena.p;
  w.ar = WideArena{(uint64_t*)(b + o_tlo), (uint64_t*)(b + o_thi), (uint64_t*)(b + o_klo),  (uint64_t*)(b + o_khi),
                   (uint64_t*)(b + o_hord), (uint32_t*)(b + o_hid), (wfst_tr*)(b + o_arcs), (uint64_t*)(b + o_alo),
                   (uint64_t*)(b + o_ahi), (uint32_t*)(b + o_sb),  (uint32_t*)(b + o_sc),  (float*)(b + o_fin)};
  w.d_off = (uint32_t*)(b + o_off);
  w.d_out = (wfst_tr*)(b + o_out);
  w.caps = caps;
}

// The level loop.  Level ranges, the overflow status and the count of finished levels live in the control block, so the
// host queues `batch` levels (three launches each) per look at it; levels queued behind the last one, or behind one that
// did not fit, return at once.  Grids are sized from the last width the host has seen (every kernel strides, so any grid
// is correct).
template <class P, uint32_t G>
void run_wide_g(wfst_ctx* ctx, const P& pol, uint64_t lo0, uint64_t hi0, uint64_t est_s, uint64_t est_a, WideOutput& out) {
  constexpr uint32_t SPB = 4 * (64 / G);  // states a block of 256 threads works on at a time
  hipStream_t st = ctx->stream;
  if (const char* e = std::getenv("WFST_WIDE_EST_STATES")) {  // tests: start from a tiny arena so that it has to grow
    est_s = std::max<uint64_t>(64, (uint64_t)std::atoll(e));
    est_a = 4 * est_s;
  } else {
    // whoever calls the wide driver has already seen the result outgrow one wave: room for a quarter of a million states
    // to begin with (clearing that table takes ~20 us)
    est_s = std::max<uint64_t>(est_s, 1ull << 18);
    // (four arcs per state while nothing is known; a caller that starts from a measured size gives both figures)
    est_a = std::max<uint64_t>(est_a, est_s <= (1ull << 22) ? 4 * est_s : est_s);
  }
  WideBuffers w;
  wide_alloc(ctx, est_s, est_a, w);
  DBuf<WideCtl> d_ctl(*ctx->pool, 1);
  size_t temp_bytes = 0;
  HIP_CHECK(rocprim::exclusive_scan(nullptr, temp_bytes, w.ar.seg_cnt, w.d_off, 0u, (size_t)0x7FFFFFF0u, rocprim::plus<uint32_t>(), st));
  DBuf<uint8_t> temp(*ctx->pool, temp_bytes);
  struct HostCtl {
    uint32_t head[WIDE_CTL_HEAD / 4];  // status, k_done, pad, lvl ring
    uint32_t n_arcs;
  };
  HostCtl* hc = (HostCtl*)ctx->pinned.get(sizeof(HostCtl));
  const uint32_t max_blocks = std::min<uint32_t>((uint32_t)ctx->n_cus * 8, WIDE_MAX_BLOCKS);
  la_wide_init<<<std::min<uint32_t>(max_blocks, (w.caps.H + 255) / 256), 256, 0, st>>>(w.ar, w.caps, lo0, hi0, d_ctl.p);
  uint32_t k = 0, lo = 0, hi = 1, patch_from = 0, w_prev = 0, batch_cap = 4;
  uint32_t lo_part = 0;  // first state whose arcs went into the current part of the arc arena
  double arcs_per_state = 4.0;
  int fixed_batch = 0;
  if (const char* e = std::getenv("WFST_WIDE_BATCH")) fixed_batch = std::max(1, std::min(32, std::atoi(e)));
  const bool proactive = !std::getenv("WFST_WIDE_NO_FORESIGHT");  // tests: let levels overflow
  const bool trace = std::getenv("WFST_WIDE_TRACE") != nullptr;
  int grows = 0;
  auto grow = [&]() {
    ctx->stats.compose_retries++;
    if (++grows > 24) throw Error("compose: arena overflow after retries");
    // the arcs of the finished levels still name table slots, and the table is about to be rebuilt
    if (lo > patch_from) la_patch_range<<<std::min<uint32_t>(max_blocks, (lo - patch_from + 31) / 32), 256, 0, st>>>(w.ar, patch_from, lo);
    patch_from = lo;
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
    lo_part = lo;
  };
  for (;;) {  // LazyFst::compute, lazy_fst.rs:235-259: level = ids [lo, hi)
    const uint32_t width = hi - lo;
    const double ratio = w_prev ? (double)width / w_prev : 3.0;
    // narrow levels are launch-bound: many per look; wide ones are not, and whatever is queued behind a level that does
    // not fit is wasted
    const uint32_t batch = fixed_batch ? (uint32_t)fixed_batch : std::min(batch_cap, width < 8192 ? 8u : (width < 65536 ? 4u : 2u));
    // room for the widest level of the batch if the growth of the last level goes on (every kernel strides, so a grid
    // that turns out too small is only slower; one that is too large costs ~3 us per launch)
    double head = 2.0;
    for (uint32_t j = 1; j < batch && ratio > 1.0 && head < 64.0; ++j) head *= ratio;
    const uint64_t want = (uint64_t)((double)width * std::min(head, 64.0) / SPB) + 1;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(max_blocks, std::max<uint64_t>(64, want));
    for (uint32_t j = 0; j < batch; ++j) {
      la_emit<P, G><<<blocks, 256, 0, st>>>(pol, w.caps, w.ar, k + j, d_ctl.p);
      la_first<G><<<blocks, 256, 0, st>>>(w.ar, k + j, d_ctl.p);
      la_assign<G><<<blocks, 256, 0, st>>>(w.ar, w.caps, k + j, d_ctl.p);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(hc->head, d_ctl.p, WIDE_CTL_HEAD, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const uint32_t status = hc->head[0];
    const uint32_t k_new = hc->head[1];
    if (k_new > k) w_prev = hc->head[32 + 2 * ((k_new - 1) % LVL_RING) + 1] - hc->head[32 + 2 * ((k_new - 1) % LVL_RING)];
    k = k_new;
    lo = hc->head[32 + 2 * (k % LVL_RING)];
    hi = hc->head[32 + 2 * (k % LVL_RING) + 1];
    if (trace)
      fprintf(stderr, "wide: k=%u [%u,%u) status=%u S=%u A=%u slack_min=%u arcs_used=%u batch=%u blocks=%u\n", k, lo, hi, status,
              w.caps.S, w.caps.A, hc->head[2], hc->head[3], batch, blocks);
    if (status != LA_OK) {
      // level k does not fit (arc slices, table or state count): more room, everything numbered so far moves over, the
      // level is emitted again
      grow();
      continue;
    }
    if (lo >= hi) break;  // the last level added nothing
    if (proactive) {
      // how many of the next levels are going to fit?  Level k adds about width x (growth of the last level) states and
      // width x (arcs per state so far) arcs, spread over the shards, the one after it that times the growth again ...
      // Only that many are queued; if not even level k fits the arena grows NOW: same cost as after the overflow, minus
      // the emission that would have been thrown away.
      if (lo > lo_part) arcs_per_state = std::max(1.0, (double)hc->head[3] / (lo - lo_part));
      const double r = w_prev ? std::min(8.0, std::max(1.0, (double)(hi - lo) / w_prev)) : 3.0;
      double slack = hc->head[2];
      auto levels_that_fit = [&](uint32_t limit) {
        double states = hi, wl = hi - lo, room = slack;
        uint32_t n = 0;
        for (; n < limit; ++n) {
          // (a level cannot discover more states than it emits arcs)
          const double arcs_shard = arcs_per_state * wl / CUR_SHARDS, new_states = std::min(1.25 * r, arcs_per_state) * wl;
          if (states + new_states > (double)w.caps.S || 1.5 * arcs_shard + 64.0 > room) break;
          states += new_states;
          room -= arcs_shard;
          wl *= r;
        }
        return n;
      };
      while ((batch_cap = levels_that_fit(8)) == 0) {
        if (trace) fprintf(stderr, "wide: growing ahead of level %u (width %u, growth %.2f, %.1f arcs per state)\n", k, hi - lo, r, arcs_per_state);
        const uint32_t a_old = w.caps.A;
        grow();
        slack = (double)((w.caps.A - a_old) / CUR_SHARDS);
      }
    }
  }
  const uint32_t n_states = hi;
  HIP_CHECK(hipMemsetAsync(w.ar.seg_cnt + n_states, 0, sizeof(uint32_t), st));
  HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, w.ar.seg_cnt, w.d_off, 0u, (size_t)n_states + 1, rocprim::plus<uint32_t>(), st));
  la_gather<G><<<std::min<uint32_t>(max_blocks, (n_states + SPB - 1) / SPB), 256, 0, st>>>(w.ar, w.d_off, w.d_out, n_states, patch_from);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(&hc->n_arcs, w.d_off + n_states, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  out.n_states = n_states;
  out.n_arcs = hc->n_arcs;
  out.n_levels = k;
  out.off = w.d_off;
  out.arcs = w.d_out;
  out.fin = w.ar.fin;
  out.arena = std::move(w.arena);
}

// `items_hint`: arcs a composed state has on its iterated side, on average (+ the loop item) — picks the lanes per state
template <class P>
void run_wide(wfst_ctx* ctx, const P& pol, uint64_t lo0, uint64_t hi0, uint64_t est_s, uint64_t est_a, double items_hint,
              WideOutput& out) {
  // (2 / 4 lanes per state where a state has a handful of items — a linear acceptor against a transducer: 2 — : the kernels move
  // a few sectors per composed state through a chain of dependent trips, and 32 / 16 states per wave instead of 8 is that many
  // more loads in flight for the same registers: the 90 M-state look-ahead composition 107.8 (8 lanes) -> 87.0 (4) -> 77.1 ms
  // (2), profiles/r06d_wide_lookahead.md)
  uint32_t g = items_hint <= 2.5 ? 2u : (items_hint <= 4.0 ? 4u : (items_hint <= 8.0 ? 8u : (items_hint <= 16.0 ? 16u : 64u)));
  if (const char* e = std::getenv("WFST_WIDE_GROUP")) g = (uint32_t)std::atoi(e);  // tests: 2, 4, 8, 16 or 64 lanes per state
  if (g == 2) run_wide_g<P, 2>(ctx, pol, lo0, hi0, est_s, est_a, out);
  else if (g == 4) run_wide_g<P, 4>(ctx, pol, lo0, hi0, est_s, est_a, out);
  else if (g == 8) run_wide_g<P, 8>(ctx, pol, lo0, hi0, est_s, est_a, out);
  else if (g == 16) run_wide_g<P, 16>(ctx, pol, lo0, hi0, est_s, est_a, out);
  else run_wide_g<P, 64>(ctx, pol, lo0, hi0, est_s, est_a, out);
}

}  // namespace
}  // namespace wfst
