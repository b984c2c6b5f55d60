// sssp.hip — n=1 shortest path on an HBM-resident CSR: frontier-synchronous tropical relaxation.
//
// Replaces rustfst::algorithms::shortest_path for nshortest == 1:
//   single_shortest_path            rustfst/src/algorithms/shortest_path.rs:173-239
//   single_shortest_path_backtrace  rustfst/src/algorithms/shortest_path.rs:241-282
// The reference relaxes states one at a time in AutoQueue order (queues/auto_queue.rs:23-99: top order
// on lattices, FIFO inside cyclic SCCs).  Here every frontier state of a sweep is relaxed at once:
//   key[t] = (order-preserving f32 bits of d[t]) << 32 | hops[t]      one u64 per state
//   relax arc (s,w,t):  cand = (enc(d[s] + w) << 32) | (hops[s] + 1);  atomicMin(&key[t], cand)
// The least fixed point of that recurrence is unique (f32 + is monotone), so the result does not depend
// on scheduling.  Ties are resolved canonically (DESIGN.md §Shortest path): fewest arcs, then the
// smallest (source state, arc position) predecessor, then the smallest final state id.
// No MFMA: this is sparse DP; the bound is HBM / L2-atomic traffic (20 B per arc relaxed by the
// SURVEY §8(d) accounting; this layout actually streams 8 B of arc + one 8-B atomic).
#include <cstdlib>

#include "common.h"
#include "fst_props.h"

namespace wfst {

namespace {

inline uint32_t __float_as_uint_host(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}

constexpr uint64_t KEY_INF = ~0ull;
constexpr uint32_t GROUP = 16;  // lanes cooperating on one frontier state (average fan-out ~10)

__device__ __forceinline__ uint32_t enc_f32(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e) {
  uint32_t b = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
  return __uint_as_float(b);
}

// Near-far schedule (a Delta-stepping relative that needs no buckets): sweep k only relaxes active states
// with d <= tau_k; the others stay in the frontier.  When a sweep activates nothing near, tau advances by
// delta (doubling on consecutive empty sweeps so that gaps are crossed in log time).  All of it is decided on
// the device from a few words per sweep kept in a ring, so sweeps still launch back-to-back without a host
// round trip.  When the near set is too small to fill the GPU (< near_low activations: such a sweep costs the
// launch floor whatever it does) and shrinking — the tail of a band — the band is widened by delta at once
// instead of draining through half a dozen near-empty sweeps.  Activations are counted with ONE atomicAdd per
// workgroup on a counter sharded 16 ways, one shard per 128-B line (atomics on one LINE serialise at ~12 ns each).
constexpr uint32_t RING = 256;
constexpr uint32_t NEAR_SHARDS = 16;
constexpr uint32_t NEAR_RING = 4;     // sweep k writes slot k % 4, reads k-1 and k-2, recycles k+1
constexpr uint32_t NEAR_STRIDE = 32;  // one shard per 128-B line
constexpr uint32_t IMP_RING = 512;    // per-sweep "something happened" flags, indexed by sweep % IMP_RING

// How a sweep hands its candidates (t, d[s] + w, hops[s] + 1) to the target states:
//  MODE_ATOMIC  atomicMin on key[t] after a plain pre-check.  Bound by the atomic rate of the memory system
//               (26.5 G/s on MI355X whatever the width, scope or locality: tools/ubench_atomics.hip), fine for
//               the many small sweeps.
//  MODE_BINS    "propagation blocking": the state space is cut into partitions of PB_PART states, workgroup p OWNS
//               partition p.  A sweep appends every candidate to the bin of its target's partition (12-B records,
//               space reserved per (workgroup, bin) from a two-pass LDS histogram); the NEXT sweep's workgroup p
//               first merges bin p into its partition with ds_min_u64 on an LDS copy of the keys, writes back the
//               improved keys and frontier flags, then scans its own partition.  No global atomics on keys, no
//               random gathers: arcs, candidates and keys all stream.  Used for the big sweeps.
enum : uint32_t { MODE_ATOMIC = 0, MODE_BINS = 1 };
constexpr uint32_t PB_SHIFT = 12;
constexpr uint32_t PB_PART = 1u << PB_SHIFT;              // states owned by one workgroup (32 KB of keys in LDS)
constexpr uint32_t PB_THREADS = 1024;
constexpr uint32_t PB_WAVES = PB_THREADS / 64;            // 16
constexpr uint32_t PB_CHUNKS = PB_PART / 64 / PB_WAVES;   // 64-state chunks scanned by each wave: 4
constexpr uint32_t PB_MAX_PARTS = 2048;                   // LDS histogram size: n <= 8.4M states
constexpr uint32_t PB_OFF = 0xFFFFFFFFu;                  // pb_low value that disables MODE_BINS

struct Cand {
  uint32_t enc, hops, t;  // candidate key (enc(d) : hops) for state t
};

struct PbArgs {
  uint32_t* tails;          // [2][parts]: candidates waiting in each bin of buffer (sweep & 1)
  const uint32_t* bin_off;  // [parts + 1]: bin q = [bin_off[q], bin_off[q+1]) — sized by the in-degree of partition q
  Cand* cand;               // [2][cand_stride]
  uint64_t cand_stride;     // = number of arcs (a sweep emits at most one candidate per arc)
  uint32_t parts;
  uint32_t pb_low;          // estimated activations from which a sweep goes through the bins; PB_OFF = never
};

struct Ctl {
  uint32_t base;          // first sweep index of the batch being replayed (graph nodes add their static offset)
  uint32_t pad0;
  uint32_t tau[RING];     // f32 bits of the threshold used by sweep k (written by sweep k, read by sweep k+1)
  uint32_t streak[RING];  // consecutive sweeps before k that activated nothing near
  uint32_t mode[RING];    // how sweep k emitted its candidates
  // activations with d <= tau_k made by sweep k / states left in the far set by sweep k: counters sharded 16
  // ways, shard j at [j * NEAR_STRIDE].  (A MODE_BINS sweep counts candidates / 4: it cannot know which improve.)
  uint32_t near[NEAR_RING][NEAR_SHARDS * NEAR_STRIDE];
  uint32_t far[NEAR_RING][NEAR_SHARDS * NEAR_STRIDE];
  unsigned long long prof[NEAR_SHARDS][16];  // [shard][0] arcs relaxed, [shard][1] frontier states relaxed (cumulative)
  unsigned long long best;    // enc(total) << 32 | final state
  // backtrace header
  uint32_t f_parent, hops;
  float final_weight, total;
  uint32_t has_path, pad;
};

struct SweepPlan {
  float tau;
  uint32_t streak, mode, prev_mode;
};

// Threshold and emission mode of sweep k from what sweeps k-1 and k-2 left in the rings.  Called by one full
// wave (all 64 lanes): lanes 0..15 fetch the shards of near[k-1], 16..31 those of near[k-2], 32..47 far[k-1], so the
// whole decision costs one load latency.  Every workgroup computes the same plan.
__device__ __forceinline__ SweepPlan sweep_plan(const Ctl* ctl, uint32_t sweep, float delta, uint32_t near_low,
                                                uint32_t pb_low) {
  SweepPlan pl{delta, 0u, MODE_ATOMIC, MODE_ATOMIC};
  if (sweep == 0) return pl;
  const uint32_t p = (sweep - 1) % RING;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t mine = 0;
  if (lane < NEAR_SHARDS) mine = ctl->near[(sweep - 1) % NEAR_RING][lane * NEAR_STRIDE];
  else if (lane < 2 * NEAR_SHARDS) mine = sweep >= 2 ? ctl->near[(sweep - 2) % NEAR_RING][(lane - NEAR_SHARDS) * NEAR_STRIDE] : 0u;
  else if (lane < 3 * NEAR_SHARDS) mine = ctl->far[(sweep - 1) % NEAR_RING][(lane - 2 * NEAR_SHARDS) * NEAR_STRIDE];
  const float prev = __uint_as_float(ctl->tau[p]);
  const uint32_t prev_streak = ctl->streak[p];
  pl.prev_mode = ctl->mode[p];
  for (int d = 8; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);  // sums inside each group of 16 lanes
  const uint32_t cnt = __shfl(mine, 0), before = __shfl(mine, 16), far = __shfl(mine, 32);
  bool advance = false;
  if (cnt >= near_low) {
    pl.tau = prev;
  } else if (cnt) {
    // a near set that cannot fill the GPU AND is shrinking (the tail of a band, not its growing head):
    // widen the band by delta and keep relaxing
    advance = cnt < before;
    pl.tau = advance ? prev + delta : prev;
  } else {
    const uint32_t st = min(prev_streak + 1u, 30u);
    pl.streak = st;
    pl.tau = prev + delta * (float)(1u << (st - 1u));
    advance = true;
  }
  const uint64_t est = (uint64_t)cnt + (advance ? far : 0u);  // activations this sweep will find in its frontier
  pl.mode = (pb_low != PB_OFF && est >= pb_low) ? MODE_BINS : MODE_ATOMIC;
  return pl;
}

__global__ void sssp_init_kernel(uint64_t* key, uint8_t* flags0, Ctl* ctl, uint32_t start) {
  // ctl was zeroed by a memset
  key[start] = (uint64_t)enc_f32(0.0f) << 32;  // d[source] = 1-bar, hops 0   (shortest_path.rs:204)
  flags0[start] = 1;
  ctl->best = KEY_INF;
}

struct SweepCounts {
  uint32_t near = 0, far = 0, states = 0, arcs = 0;
  bool any = false;
};

// Loads a 64-state chunk of the frontier: the lane that owns an active state fetches everything the relaxation of
// that state needs (ONE memory round trip for the whole chunk; the groups get it by shuffle later), clears its
// flag, and re-files the state in the next frontier when it is beyond tau.  Returns the mask of states to relax.
__device__ __forceinline__ uint64_t load_chunk(uint32_t base, uint32_t lane, uint32_t n, const uint32_t* __restrict__ offsets,
                                               const uint64_t* __restrict__ key, uint8_t* __restrict__ flags_cur,
                                               uint8_t* __restrict__ flags_next, float tau, uint64_t& my_ks, uint32_t& my_b,
                                               uint32_t& my_e, SweepCounts& cn, uint32_t flag) {
  const uint32_t sc = base + lane;
  bool act = flag != 0;
  my_ks = 0;
  my_b = 0;
  my_e = 0;
  if (__ballot(act) == 0) return 0;
  if (act) {
    flags_cur[sc] = 0;  // this buffer is the NEXT frontier two sweeps from now
    my_ks = key[sc];
    my_b = offsets[sc];
    my_e = offsets[sc + 1];
    if (dec_f32((uint32_t)(my_ks >> 32)) > tau) {  // far: stays in the frontier, is not relaxed in this sweep
      flags_next[sc] = 1;
      cn.any = true;
      cn.far += 1;
      act = false;
    } else {
      cn.states += 1;
      cn.arcs += my_e - my_b;
    }
  }
  return __ballot(act);
}

// MODE_ATOMIC emission for one chunk.  GROUP lanes share a state so that its arcs (contiguous 8-B {w,next} records)
// are read by consecutive lanes; each of the wave's 4 groups takes TWO states per round, so two independent
// load -> pre-check -> atomic chains per lane are in flight at once (these sweeps are latency bound).
__device__ __forceinline__ void relax_chunk_atomic(uint64_t mask, uint32_t lane, uint64_t my_ks, uint32_t my_b, uint32_t my_e,
                                                   const uint2* __restrict__ wn, uint64_t* __restrict__ key,
                                                   uint8_t* __restrict__ flags_next, float tau, SweepCounts& cn) {
  const uint32_t sub = lane % GROUP;  // lane inside its group
  const uint32_t grp = lane / GROUP;  // group inside the wave (0..3)
  while (mask) {
    const uint64_t m1 = mask & (mask - 1), m2 = m1 & (m1 - 1), m3 = m2 & (m2 - 1);
    const uint64_t m4 = m3 & (m3 - 1), m5 = m4 & (m4 - 1), m6 = m5 & (m5 - 1), m7 = m6 & (m6 - 1);
    const uint64_t mine_a = grp == 0 ? mask : grp == 1 ? m1 : grp == 2 ? m2 : m3;
    const uint64_t mine_b = grp == 0 ? m4 : grp == 1 ? m5 : grp == 2 ? m6 : m7;
    mask = m7 & (m7 - 1);
    const bool has_a = mine_a != 0, has_b = mine_b != 0;
    const int la = has_a ? __ffsll((unsigned long long)mine_a) - 1 : 0;
    const int lb = has_b ? __ffsll((unsigned long long)mine_b) - 1 : 0;
    // (shuffles stay outside any lane-dependent condition: a bpermute reads 0 from a lane that is masked off)
    const uint64_t ks_a = __shfl(my_ks, la), ks_b = __shfl(my_ks, lb);
    uint32_t ia = __shfl(my_b, la) + sub, ib = __shfl(my_b, lb) + sub;
    const uint32_t ea_all = __shfl(my_e, la), eb_all = __shfl(my_e, lb);
    const uint32_t ea = has_a ? ea_all : 0u, eb = has_b ? eb_all : 0u;
    const float da = dec_f32((uint32_t)(ks_a >> 32)), db = dec_f32((uint32_t)(ks_b >> 32));
    const uint32_t ha = (uint32_t)ks_a + 1u, hb = (uint32_t)ks_b + 1u;
    while (__any(ia < ea || ib < eb)) {
      const bool va = ia < ea, vb = ib < eb;
      uint2 aa = make_uint2(0x7F800000u, 0u), ab = make_uint2(0x7F800000u, 0u);  // {+inf, state 0}: never relaxes
      if (va) aa = wn[ia];
      if (vb) ab = wn[ib];
      const float ca = (da + __uint_as_float(aa.x)) + 0.0f;  // w1 (x) w2 = f32 add (tropical_weight.rs:60-70)
      const float cb = (db + __uint_as_float(ab.x)) + 0.0f;
      const bool fa = va && ca < INF, fb = vb && cb < INF;  // +inf never improves (shortest_path.rs:226)
      const uint64_t cka = ((uint64_t)enc_f32(ca) << 32) | ha, ckb = ((uint64_t)enc_f32(cb) << 32) | hb;
      // plain pre-check: keys only decrease, a stale read can only cost an extra atomic
      uint64_t ka = 0, kb = 0;
      if (fa) ka = key[aa.y];
      if (fb) kb = key[ab.y];
      const bool ta = fa && cka < ka, tb = fb && ckb < kb;
      uint64_t olda = 0, oldb = 0;
      if (ta) olda = atomicMin((unsigned long long*)&key[aa.y], (unsigned long long)cka);
      if (tb) oldb = atomicMin((unsigned long long*)&key[ab.y], (unsigned long long)ckb);
      if (ta && cka < olda) {
        flags_next[aa.y] = 1;
        cn.any = true;
        if (ca <= tau) cn.near += 1; else cn.far += 1;
      }
      if (tb && ckb < oldb) {
        flags_next[ab.y] = 1;
        cn.any = true;
        if (cb <= tau) cn.near += 1; else cn.far += 1;
      }
      ia += GROUP;
      ib += GROUP;
    }
  }
}

struct SweepShared {
  uint32_t any, near, far, states, arcs, mode, prev_mode;
  float tau;
};

// wave 0 plans the sweep; workgroup 0 publishes the plan and recycles the next sweep's counter slots
__device__ __forceinline__ void sweep_prologue(Ctl* ctl, uint32_t sweep, float delta, uint32_t near_low, uint32_t pb_low,
                                               SweepShared& sh) {
  if (threadIdx.x < 64) {
    const SweepPlan pl = sweep_plan(ctl, sweep, delta, near_low, pb_low);
    if (threadIdx.x == 0) {
      sh.tau = pl.tau;
      sh.mode = pl.mode;
      sh.prev_mode = pl.prev_mode;
      sh.any = sh.near = sh.far = sh.states = sh.arcs = 0;
    }
    if (blockIdx.x == 0) {
      const uint32_t slot = sweep % RING;
      if (threadIdx.x == 0) {
        ctl->tau[slot] = __float_as_uint(pl.tau);
        ctl->streak[slot] = pl.streak;
        ctl->mode[slot] = pl.mode;
      }
      if (threadIdx.x < NEAR_SHARDS) {
        ctl->near[(sweep + 1) % NEAR_RING][threadIdx.x * NEAR_STRIDE] = 0;
        ctl->far[(sweep + 1) % NEAR_RING][threadIdx.x * NEAR_STRIDE] = 0;
      }
    }
  }
  __syncthreads();
}

// one conditional plain store + a few sharded atomicAdds per workgroup
__device__ __forceinline__ void sweep_epilogue(Ctl* ctl, uint32_t* improved, uint32_t sweep, SweepCounts cn, SweepShared& sh) {
  const uint32_t lane = threadIdx.x & 63u;
  for (int d = 32; d >= 1; d >>= 1) {
    cn.near += __shfl_xor(cn.near, d);
    cn.far += __shfl_xor(cn.far, d);
    cn.states += __shfl_xor(cn.states, d);
    cn.arcs += __shfl_xor(cn.arcs, d);
  }
  const bool wave_any = __any(cn.any);
  if (lane == 0) {
    if (wave_any) sh.any = 1u;
    if (cn.near) atomicAdd(&sh.near, cn.near);
    if (cn.far) atomicAdd(&sh.far, cn.far);
    if (cn.states) {
      atomicAdd(&sh.states, cn.states);
      atomicAdd(&sh.arcs, cn.arcs);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t shard = blockIdx.x % NEAR_SHARDS;
    if (sh.any && *improved == 0u) *improved = 1u;
    if (sh.near) atomicAdd(&ctl->near[sweep % NEAR_RING][shard * NEAR_STRIDE], sh.near);
    if (sh.far) atomicAdd(&ctl->far[sweep % NEAR_RING][shard * NEAR_STRIDE], sh.far);
    if (sh.states) {
      atomicAdd(&ctl->prof[shard][0], (unsigned long long)sh.arcs);
      atomicAdd(&ctl->prof[shard][1], (unsigned long long)sh.states);
    }
  }
}

// One sweep, MODE_ATOMIC only (small FSTs, or bins switched off): relax every arc leaving the near frontier.
// The frontier is a byte flag per state (idempotent plain stores: no queue, no dedupe atomics); a wave scans 64
// flags with one coalesced load + ballot.
__global__ void __launch_bounds__(256) sssp_relax_kernel(const uint32_t* __restrict__ offsets,
                                                         const uint2* __restrict__ wn, uint64_t* __restrict__ key,
                                                         uint8_t* __restrict__ flags_cur,
                                                         uint8_t* __restrict__ flags_next, uint32_t n,
                                                         uint32_t* __restrict__ improved_ring, Ctl* __restrict__ ctl,
                                                         uint32_t sweep_offset, float delta, uint32_t near_low) {
  // the sweep index is (device-side batch base) + (static offset of this launch / graph node)
  const uint32_t sweep = ctl->base + sweep_offset;
  __shared__ SweepShared sh;
  sweep_prologue(ctl, sweep, delta, near_low, PB_OFF, sh);
  const float tau = sh.tau;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  SweepCounts cn;
  for (uint32_t base = wave * 64u; base < n; base += n_waves * 64u) {
    uint64_t my_ks;
    uint32_t my_b, my_e;
    const uint32_t flag = base + lane < n ? flags_cur[base + lane] : 0u;
    const uint64_t mask = load_chunk(base, lane, n, offsets, key, flags_cur, flags_next, tau, my_ks, my_b, my_e, cn, flag);
    relax_chunk_atomic(mask, lane, my_ks, my_b, my_e, wn, key, flags_next, tau, cn);
  }
  sweep_epilogue(ctl, improved_ring + (sweep % IMP_RING), sweep, cn, sh);
}

// visits the arcs of the states in `mask` (8 states per round: two per group of GROUP lanes, so that two arc loads
// per lane are in flight) and calls fn(candidate weight, hops, target)
template <class F>
__device__ __forceinline__ void for_each_candidate(uint64_t mask, uint32_t lane, uint64_t my_ks, uint32_t my_b, uint32_t my_e,
                                                   const uint2* __restrict__ wn, F&& fn) {
  const uint32_t sub = lane % GROUP, grp = lane / GROUP;
  while (mask) {
    const uint64_t m1 = mask & (mask - 1), m2 = m1 & (m1 - 1), m3 = m2 & (m2 - 1);
    const uint64_t m4 = m3 & (m3 - 1), m5 = m4 & (m4 - 1), m6 = m5 & (m5 - 1), m7 = m6 & (m6 - 1);
    const uint64_t mine_a = grp == 0 ? mask : grp == 1 ? m1 : grp == 2 ? m2 : m3;
    const uint64_t mine_b = grp == 0 ? m4 : grp == 1 ? m5 : grp == 2 ? m6 : m7;
    mask = m7 & (m7 - 1);
    const bool has_a = mine_a != 0, has_b = mine_b != 0;
    const int la = has_a ? __ffsll((unsigned long long)mine_a) - 1 : 0;
    const int lb = has_b ? __ffsll((unsigned long long)mine_b) - 1 : 0;
    const uint64_t ks_a = __shfl(my_ks, la), ks_b = __shfl(my_ks, lb);
    uint32_t ia = __shfl(my_b, la) + sub, ib = __shfl(my_b, lb) + sub;
    const uint32_t ea_all = __shfl(my_e, la), eb_all = __shfl(my_e, lb);
    const uint32_t ea = has_a ? ea_all : 0u, eb = has_b ? eb_all : 0u;
    const float da = dec_f32((uint32_t)(ks_a >> 32)), db = dec_f32((uint32_t)(ks_b >> 32));
    const uint32_t ha = (uint32_t)ks_a + 1u, hb = (uint32_t)ks_b + 1u;
    while (__any(ia < ea || ib < eb)) {
      const bool va = ia < ea, vb = ib < eb;
      uint2 aa = make_uint2(0x7F800000u, 0u), ab = make_uint2(0x7F800000u, 0u);
      if (va) aa = wn[ia];
      if (vb) ab = wn[ib];
      const float ca = (da + __uint_as_float(aa.x)) + 0.0f, cb = (db + __uint_as_float(ab.x)) + 0.0f;
      if (va && ca < INF) fn(ca, ha, aa.y);
      if (vb && cb < INF) fn(cb, hb, ab.y);
      ia += GROUP;
      ib += GROUP;
    }
  }
}

// One sweep with workgroup-owned partitions (see MODE_BINS above).  grid = number of partitions, 1024 threads.
//   1. if the previous sweep emitted through the bins: merge my bin into my partition (LDS), write back, set flags
//   2. scan: my own partition after a MODE_BINS sweep (only its owner knows when its flags are complete), else a
//      grid-strided share of all chunks
//   3. emit: MODE_ATOMIC as in sssp_relax_kernel, or MODE_BINS: count per target bin (LDS histogram), reserve space in
//      every bin with one atomicAdd per (workgroup, bin), write the 12-B candidate records
__global__ void __launch_bounds__(PB_THREADS) sssp_sweep_bins_kernel(const uint32_t* __restrict__ offsets,
                                                                     const uint2* __restrict__ wn, uint64_t* __restrict__ key,
                                                                     uint8_t* __restrict__ flags_cur,
                                                                     uint8_t* __restrict__ flags_next, uint32_t n,
                                                                     uint32_t* __restrict__ improved_ring, Ctl* __restrict__ ctl,
                                                                     uint32_t sweep_offset, float delta, uint32_t near_low,
                                                                     PbArgs pb) {
  __shared__ unsigned long long lds_key[PB_PART];
  __shared__ uint32_t lds_imp[PB_PART / 32];
  __shared__ uint32_t lds_hist[PB_MAX_PARTS];
  __shared__ SweepShared sh;
  const uint32_t sweep = ctl->base + sweep_offset;
  sweep_prologue(ctl, sweep, delta, near_low, pb.pb_low, sh);
  const float tau = sh.tau;
  const uint32_t mode = sh.mode, prev_mode = sh.prev_mode;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, p = blockIdx.x;
  SweepCounts cn;

  // ---- 1. merge the candidates the previous sweep left in my bin
  if (prev_mode == MODE_BINS) {
    uint32_t* my_tail = pb.tails + ((sweep - 1u) & 1u) * pb.parts + p;
    const uint32_t cnt = *my_tail;
    if (cnt) {  // uniform over the workgroup
      const uint32_t s0 = p * PB_PART;
      for (uint32_t i = tid; i < PB_PART; i += PB_THREADS) lds_key[i] = s0 + i < n ? key[s0 + i] : 0ull;
      if (tid < PB_PART / 32) lds_imp[tid] = 0;
      __syncthreads();
      const Cand* __restrict__ c = pb.cand + ((sweep - 1u) & 1u) * pb.cand_stride + pb.bin_off[p];
      for (uint32_t i = tid; i < cnt; i += PB_THREADS) {
        const Cand r = c[i];
        const uint32_t tl = r.t & (PB_PART - 1u);
        const unsigned long long ck = ((unsigned long long)r.enc << 32) | r.hops;
        if (ck < lds_key[tl]) {
          const unsigned long long old = atomicMin(&lds_key[tl], ck);
          if (ck < old) atomicOr(&lds_imp[tl >> 5], 1u << (tl & 31u));
        }
      }
      __syncthreads();
      for (uint32_t i = tid; i < PB_PART; i += PB_THREADS)
        if ((lds_imp[i >> 5] >> (i & 31u)) & 1u) {
          // when this sweep emits with global atomics, other workgroups may be lowering key[] right now
          if (mode == MODE_ATOMIC) atomicMin((unsigned long long*)&key[s0 + i], lds_key[i]);
          else key[s0 + i] = lds_key[i];
          flags_cur[s0 + i] = 1;
          cn.any = true;
        }
      if (tid == 0) *my_tail = 0;
      __syncthreads();  // the flags and keys just written are read back by this workgroup's scan
    }
  }

  // ---- 2. scan   ---- 3. emit
  uint64_t my_ks[PB_CHUNKS], masks[PB_CHUNKS];
  uint32_t my_b[PB_CHUNKS], my_e[PB_CHUNKS];
  if (mode == MODE_BINS)
    for (uint32_t q = tid; q < pb.parts; q += PB_THREADS) lds_hist[q] = 0;
  uint32_t chunk_base[PB_CHUNKS], flag[PB_CHUNKS];
#pragma unroll
  for (uint32_t j = 0; j < PB_CHUNKS; ++j) {  // all flag loads of this wave in flight together
    const uint32_t chunk = prev_mode == MODE_BINS ? p * (PB_PART / 64u) + wv + PB_WAVES * j
                                                  : (p * PB_WAVES + wv) + j * (gridDim.x * PB_WAVES);
    chunk_base[j] = chunk * 64u;
    flag[j] = chunk_base[j] + lane < n ? flags_cur[chunk_base[j] + lane] : 0u;
  }
#pragma unroll
  for (uint32_t j = 0; j < PB_CHUNKS; ++j) {
    masks[j] = load_chunk(chunk_base[j], lane, n, offsets, key, flags_cur, flags_next, tau, my_ks[j], my_b[j], my_e[j], cn,
                          flag[j]);
    if (mode == MODE_ATOMIC) relax_chunk_atomic(masks[j], lane, my_ks[j], my_b[j], my_e[j], wn, key, flags_next, tau, cn);
  }
  if (mode == MODE_BINS) {
    __syncthreads();  // histogram zeroed
    uint32_t c_near = 0, c_far = 0;
#pragma unroll
    for (uint32_t j = 0; j < PB_CHUNKS; ++j)
      for_each_candidate(masks[j], lane, my_ks[j], my_b[j], my_e[j], wn, [&](float c, uint32_t, uint32_t t) {
        atomicAdd(&lds_hist[t >> PB_SHIFT], 1u);
        if (c <= tau) c_near += 1; else c_far += 1;
      });
    __syncthreads();
    uint32_t* tails = pb.tails + (sweep & 1u) * pb.parts;
    for (uint32_t q = tid; q < pb.parts; q += PB_THREADS) {
      const uint32_t h = lds_hist[q];
      // lds_hist[q] becomes the running write position of this workgroup inside bin q
      if (h) lds_hist[q] = pb.bin_off[q] + atomicAdd(&tails[q], h);
    }
    __syncthreads();
    Cand* __restrict__ out = pb.cand + (sweep & 1u) * pb.cand_stride;
#pragma unroll
    for (uint32_t j = 0; j < PB_CHUNKS; ++j)
      for_each_candidate(masks[j], lane, my_ks[j], my_b[j], my_e[j], wn, [&](float c, uint32_t h1, uint32_t t) {
        const uint32_t pos = atomicAdd(&lds_hist[t >> PB_SHIFT], 1u);
        out[pos] = Cand{enc_f32(c), h1, t};
      });
    // which candidates improve is only known after the merge: count a quarter of them as activations
    cn.near += (c_near + 3u) / 4u;
    cn.far += (c_far + 3u) / 4u;
    cn.any |= (c_near | c_far) != 0;
  }
  sweep_epilogue(ctl, improved_ring + (sweep % IMP_RING), sweep, cn, sh);
}

// in-degree of every partition = capacity of its bin (a sweep emits at most one candidate per arc)
__global__ void __launch_bounds__(256) pb_indegree_kernel(const uint2* __restrict__ wn, uint64_t n_arcs, uint32_t parts,
                                                          uint32_t* __restrict__ counts) {
  __shared__ uint32_t h[PB_MAX_PARTS];
  for (uint32_t q = threadIdx.x; q < parts; q += blockDim.x) h[q] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_arcs; i += (uint64_t)gridDim.x * blockDim.x)
    atomicAdd(&h[wn[i].y >> PB_SHIFT], 1u);
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < parts; q += blockDim.x)
    if (h[q]) atomicAdd(&counts[q], h[q]);
}

// closes a batch: the next replay continues at base + count, and the flag slots half a ring ahead are recycled
__global__ void sssp_advance_kernel(Ctl* ctl, uint32_t* improved_ring, uint32_t count) {
  const uint32_t base = ctl->base;
  for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) improved_ring[(base + IMP_RING / 2 + i) % IMP_RING] = 0;
  __syncthreads();
  if (threadIdx.x == 0) ctl->base = base + count;
}

// f_parent = argmin over final states of (d[s] (x) rho(s), s)      (shortest_path.rs:214-220)
__global__ void sssp_final_kernel(const float* __restrict__ finals, const uint64_t* __restrict__ key, uint32_t n,
                                  Ctl* __restrict__ ctl) {
  unsigned long long best = KEY_INF;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const float f = finals[s];
    const uint64_t k = key[s];
    if (k == KEY_INF || !(f < INF)) continue;
    const float tot = (dec_f32((uint32_t)(k >> 32)) + f) + 0.0f;
    if (!(tot < INF)) continue;
    const unsigned long long c = ((unsigned long long)enc_f32(tot) << 32) | s;
    best = c < best ? c : best;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(best, d);
    best = o < best ? o : best;
  }
  // few atomics on the single result word: only waves that can still lower it try (same-address atomics cost ~12 ns each)
  if ((threadIdx.x & 63) == 0 && best != KEY_INF && best < __hip_atomic_load(&ctl->best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMin(&ctl->best, best);
}

// parent[t] = min (s,pos) over arcs with (d[s]+w, hops[s]+1) == (d[t], hops[t])
__global__ void __launch_bounds__(256) sssp_parent_kernel(const uint32_t* __restrict__ offsets,
                                                          const uint2* __restrict__ wn,
                                                          const uint64_t* __restrict__ key,
                                                          unsigned long long* __restrict__ parent, uint32_t n) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = tid % GROUP;
  const uint32_t n_groups = gridDim.x * blockDim.x / GROUP;
  for (uint32_t s = tid / GROUP; s < n; s += n_groups) {
    const uint64_t ks = key[s];
    if (ks == KEY_INF) continue;
    const float d = dec_f32((uint32_t)(ks >> 32));
    const uint32_t h1 = (uint32_t)ks + 1u;
    const uint32_t b = offsets[s], e = offsets[s + 1];
    for (uint32_t i = b + lane; i < e; i += GROUP) {
      const uint2 a = wn[i];
      const float c = (d + __uint_as_float(a.x)) + 0.0f;
      if (!(c < INF)) continue;
      const uint64_t ck = ((uint64_t)enc_f32(c) << 32) | h1;
      if (ck == key[a.y]) atomicMin(&parent[a.y], ((unsigned long long)s << 32) | (i - b));
    }
  }
}

__global__ void sssp_header_kernel(const float* __restrict__ finals, const uint64_t* __restrict__ key, Ctl* ctl) {
  const unsigned long long best = ctl->best;
  if (best == KEY_INF) {
    ctl->has_path = 0;
    ctl->hops = 0;
    return;
  }
  const uint32_t fp = (uint32_t)best;
  ctl->has_path = 1;
  ctl->f_parent = fp;
  ctl->hops = (uint32_t)key[fp];
  ctl->final_weight = finals[fp];
  ctl->total = dec_f32((uint32_t)(best >> 32));
}

// single_shortest_path_backtrace (shortest_path.rs:241-282): walk parent[] from f_parent; the arc of the
// k-th created state (k >= 1) is ifst.trs(parent state)[pos] re-targeted to state k-1.
__global__ void sssp_backtrace_kernel(const uint32_t* __restrict__ offsets, const wfst_tr* __restrict__ arcs,
                                      const unsigned long long* __restrict__ parent, const Ctl* __restrict__ ctl,
                                      wfst_tr* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  uint32_t cur = ctl->f_parent;
  const uint32_t hops = ctl->hops;
  for (uint32_t k = 0; k < hops; ++k) {
    const unsigned long long p = parent[cur];
    const uint32_t s = (uint32_t)(p >> 32), pos = (uint32_t)p;
    wfst_tr tr = arcs[offsets[s] + pos];
    tr.nextstate = k;
    out[k] = tr;
    cur = s;
  }
}

__global__ void sssp_export_kernel(const uint64_t* __restrict__ key, float* __restrict__ dist, uint32_t* __restrict__ hops,
                                   uint32_t n) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint64_t k = key[s];
  dist[s] = k == KEY_INF ? INF : dec_f32((uint32_t)(k >> 32));
  if (hops) hops[s] = k == KEY_INF ? 0xFFFFFFFFu : (uint32_t)k;
}

struct Solve {
  DBuf<uint64_t> key;
  DBuf<uint8_t> flags;  // two frontiers of n bytes
  DBuf<uint32_t> improved;
  DBuf<Ctl> ctl;
  DBuf<Cand> cand;      // MODE_BINS: two candidate buffers of n_arcs records
  DBuf<uint32_t> tails; // MODE_BINS: [2][parts]
  uint32_t sweeps = 0;
};

// bin layout of an FST (prefix sums of the partitions' in-degrees): derived data, cached on the handle
const uint32_t* pb_bin_offsets(wfst_ctx* ctx, const wfst_fst* f, uint32_t parts) {
  if (f->pb_bin_off && f->pb_parts == parts) return f->pb_bin_off->p;
  hipStream_t st = ctx->stream;
  DBuf<uint32_t> counts(*ctx->pool, parts);
  HIP_CHECK(hipMemsetAsync(counts.p, 0, parts * sizeof(uint32_t), st));
  const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((f->n_arcs + 255) / 256, (uint64_t)ctx->n_cus * 2));
  pb_indegree_kernel<<<blocks, 256, 0, st>>>(f->dev.wn, f->n_arcs, parts, counts.p);
  HIP_CHECK(hipGetLastError());
  std::vector<uint32_t> h(parts + 1, 0);
  HIP_CHECK(hipMemcpyAsync(h.data() + 1, counts.p, parts * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  for (uint32_t q = 0; q < parts; ++q) h[q + 1] += h[q];
  if (h[parts] != f->n_arcs) throw Error("internal: partition in-degrees do not add up to the arc count");
  auto buf = std::make_shared<DBuf<uint32_t>>(*ctx->pool, parts + 1);
  HIP_CHECK(hipMemcpyAsync(buf->p, h.data(), (parts + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipStreamSynchronize(st));
  f->pb_bin_off = buf;
  f->pb_parts = parts;
  return buf->p;
}

constexpr uint32_t MAX_BATCH = 64;

// Runs the relaxation to its fixed point. f must have a device copy and a start state.
// Sweeps are launched in batches without returning to the host, and the NEXT batch is enqueued before the
// host looks at the previous batch's flags (two batches in flight), so the device never idles on a host round
// trip.  A sweep that changes nothing leaves an empty frontier: everything enqueued behind it is a ~3 us no-op.
void run_relaxation(wfst_ctx* ctx, const wfst_fst* f, Solve& sv) {
  const uint32_t n = f->n_states;
  DevicePool& pool = *ctx->pool;
  sv.key = DBuf<uint64_t>(pool, n);
  const size_t n_pad = ((size_t)n + 15) & ~(size_t)15;
  sv.flags = DBuf<uint8_t>(pool, 2 * n_pad);
  sv.improved = DBuf<uint32_t>(pool, IMP_RING);
  sv.ctl = DBuf<Ctl>(pool, 1);
  hipStream_t st = ctx->stream;
  HIP_CHECK(hipMemsetAsync(sv.key.p, 0xFF, (size_t)n * sizeof(uint64_t), st));
  HIP_CHECK(hipMemsetAsync(sv.flags.p, 0, 2 * n_pad, st));
  HIP_CHECK(hipMemsetAsync(sv.improved.p, 0, IMP_RING * sizeof(uint32_t), st));
  HIP_CHECK(hipMemsetAsync(sv.ctl.p, 0, sizeof(Ctl), st));
  uint8_t* fl[2] = {sv.flags.p, sv.flags.p + n_pad};
  sssp_init_kernel<<<1, 1, 0, st>>>(sv.key.p, fl[0], sv.ctl.p, (uint32_t)f->start);
  uint32_t blocks = std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)ctx->n_cus * 8, (n + 255) / 256));
  // workgroup-owned partitions + candidate bins for the big sweeps (MODE_BINS).  EXPERIMENTAL, off unless
  // WFST_SSSP_BINS=1: measured on MI355X a binned sweep has ~20 us of fixed cost (six workgroup
  // barriers, the key-partition load, one reservation atomic per (workgroup, bin)), so it only beats the
  // atomic-bound sweep above ~1M arcs (36 vs 46 us) and its 1024-thread workgroups raise the floor of the many
  // small sweeps from 8 to 11 us: a net loss on the 1M-state benchmark (708 vs 597 us per solve).
  PbArgs pb{};
  pb.pb_low = PB_OFF;
  {
    uint64_t min_states = 65536, pb_low = 32768;
    bool on = false;
    if (const char* e = std::getenv("WFST_SSSP_BINS")) on = std::atoi(e) != 0;
    if (const char* e = std::getenv("WFST_SSSP_BINS_MIN_STATES")) min_states = (uint64_t)std::atoll(e);
    if (const char* e = std::getenv("WFST_SSSP_BINS_LOW")) pb_low = (uint64_t)std::atoll(e);
    const uint32_t parts = (uint32_t)(((uint64_t)n + PB_PART - 1) >> PB_SHIFT);
    if (on && n >= min_states && parts <= PB_MAX_PARTS && f->n_arcs > 0 && f->n_arcs < 0x7FFFFFFFull) {
      pb.parts = parts;
      pb.pb_low = (uint32_t)std::min<uint64_t>(pb_low, PB_OFF - 1);
      pb.bin_off = pb_bin_offsets(ctx, f, parts);
      pb.cand_stride = f->n_arcs;
      sv.cand = DBuf<Cand>(pool, 2 * f->n_arcs);
      sv.tails = DBuf<uint32_t>(pool, 2 * (size_t)parts);
      HIP_CHECK(hipMemsetAsync(sv.tails.p, 0, 2 * (size_t)parts * sizeof(uint32_t), st));
      pb.cand = sv.cand.p;
      pb.tails = sv.tails.p;
      blocks = parts;
    }
  }
  const bool bins = pb.pb_low != PB_OFF;
  // near-far only pays on branching graphs (label-correcting re-relaxes them many times); on lattices every
  // arc is relaxed once anyway.  delta = 1.5 x mean arc weight; +inf = plain frontier sweeps.
  float delta = INF;
  if (!f->has_negative && f->mean_weight > 0.0f && n >= 65536 && f->n_arcs >= 2ull * n) delta = 1.5f * f->mean_weight;
  if (const char* e = std::getenv("WFST_SSSP_DELTA")) delta = (float)std::atof(e);  // experiments / tests
  if (!(delta > 0.0f)) delta = INF;
  uint32_t near_low = 4096;  // activations below which a sweep is launch-latency bound anyway
  if (const char* e = std::getenv("WFST_SSSP_NEAR_LOW")) near_low = (uint32_t)std::atol(e);
  const uint64_t sweep_cap = 4ull * n + 64;
  ctx->stats.sweeps = 0;

  uint32_t sweeps_done = 0;
  if (ctx->profiling) {
    // one sweep at a time, bracketed by events; the kernels count the states and arcs they relax themselves
    auto prof_sum = [](const Ctl* c, int which) {
      uint64_t t = 0;
      for (uint32_t j = 0; j < NEAR_SHARDS; ++j) t += c->prof[j][which];
      return t;
    };
    uint32_t* h_imp = (uint32_t*)ctx->pinned.get(64 + sizeof(Ctl));
    Ctl* h_ctl = (Ctl*)((char*)h_imp + 64);
    ctx->sweep_trace.clear();
    uint64_t prev_arcs = 0, prev_states = 0;
    for (uint32_t k = 0;; ++k) {
      if (k > sweep_cap) throw Error("shortest_path: relaxation did not converge (negative-weight cycle?)");
      HIP_CHECK(hipEventRecord(ctx->ev0, st));
      if (bins)
        sssp_sweep_bins_kernel<<<blocks, PB_THREADS, 0, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, fl[k & 1u],
                                                             fl[(k & 1u) ^ 1u], n, sv.improved.p, sv.ctl.p, 0u, delta,
                                                             near_low, pb);
      else
        sssp_relax_kernel<<<blocks, 256, 0, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, fl[k & 1u], fl[(k & 1u) ^ 1u], n,
                                                  sv.improved.p, sv.ctl.p, 0u, delta, near_low);
      HIP_CHECK(hipEventRecord(ctx->ev1, st));
      sssp_advance_kernel<<<1, 64, 0, st>>>(sv.ctl.p, sv.improved.p, 1u);
      HIP_CHECK(hipMemcpyAsync(h_imp, sv.improved.p + (k % IMP_RING), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipMemcpyAsync(h_ctl, sv.ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
      // (bit 63 of the state count marks a sweep that emitted through the bins)
      ctx->sweep_trace.push_back({(double)ms, prof_sum(h_ctl, 0) - prev_arcs,
                                  (prof_sum(h_ctl, 1) - prev_states) | ((uint64_t)(h_ctl->mode[k % RING] == MODE_BINS) << 63)});
      prev_arcs = prof_sum(h_ctl, 0);
      prev_states = prof_sum(h_ctl, 1);
      ctx->stats.relax_ms += ms;
      ctx->stats.relax_launches += 1;
      sweeps_done = k + 1;
      if (!h_imp[0]) break;
    }
    HIP_CHECK(hipMemcpyAsync(h_ctl, sv.ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    ctx->stats.relax_arcs += prof_sum(h_ctl, 0);
    ctx->stats.relax_states += prof_sum(h_ctl, 1);
  } else {
    // One HIP graph = one batch: `count` sweep kernels (static offsets 0..count-1 from the device-side base), the
    // copy of the flag ring to pinned host memory and the advance kernel, chained.  A replay is ONE host call
    // instead of count+2, which matters twice: a sweep on a small frontier (~4 us) is shorter than a launch, and
    // the host thread of another context (bench.py overlaps the batch pipeline on a second stream) is not starved
    // of the runtime.  The graph is built with explicit nodes — stream capture would make every other thread's
    // hipStreamSynchronize fail while it is active.  Two replays are kept in flight.
    uint32_t* h_imp = (uint32_t*)ctx->pinned_flags.get(2 * IMP_RING * sizeof(uint32_t));
    auto get_graph = [&](int which, uint32_t count) -> hipGraphExec_t {
      wfst_ctx::SweepGraph& g = ctx->sweep_graph[which];
      const uint64_t key[8] = {(uint64_t)f->dev.offsets, (uint64_t)f->dev.wn, (uint64_t)sv.key.p, (uint64_t)sv.flags.p,
                               (uint64_t)sv.improved.p, (uint64_t)sv.ctl.p, ((uint64_t)n << 32) | __float_as_uint_host(delta),
                               (uint64_t)(h_imp + which * IMP_RING) ^ ((uint64_t)near_low << 48) ^ (uint64_t)pb.cand ^
                                   ((uint64_t)pb.tails << 1) ^ ((uint64_t)pb.bin_off << 2) ^ ((uint64_t)pb.pb_low << 20)};
      if (g.exec && std::memcmp(g.key, key, sizeof(key)) == 0) return g.exec;
      if (g.exec) HIP_CHECK(hipGraphExecDestroy(g.exec));
      if (g.graph) HIP_CHECK(hipGraphDestroy(g.graph));
      g.exec = nullptr;
      g.graph = nullptr;
      HIP_CHECK(hipGraphCreate(&g.graph, 0));
      hipGraphNode_t prev = nullptr;
      const uint32_t* a_offsets = f->dev.offsets;
      const uint2* a_wn = f->dev.wn;
      uint64_t* a_key = sv.key.p;
      uint32_t a_n = n;
      uint32_t* a_imp = sv.improved.p;
      Ctl* a_ctl = sv.ctl.p;
      float a_delta = delta;
      uint32_t a_low = near_low;
      for (uint32_t j = 0; j < count; ++j) {  // batches start at multiples of their size: flag parity is static
        uint8_t* a_fc = fl[j & 1u];
        uint8_t* a_fn = fl[(j & 1u) ^ 1u];
        uint32_t a_off = j;
        PbArgs a_pb = pb;
        void* args[] = {&a_offsets, &a_wn, &a_key, &a_fc, &a_fn, &a_n, &a_imp, &a_ctl, &a_off, &a_delta, &a_low, &a_pb};
        hipKernelNodeParams kp{};
        kp.func = bins ? (void*)sssp_sweep_bins_kernel : (void*)sssp_relax_kernel;
        kp.gridDim = dim3(blocks);
        kp.blockDim = dim3(bins ? PB_THREADS : 256u);
        kp.sharedMemBytes = 0;
        kp.kernelParams = args;
        kp.extra = nullptr;
        hipGraphNode_t node;
        HIP_CHECK(hipGraphAddKernelNode(&node, g.graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
        prev = node;
      }
      {
        hipGraphNode_t node;
        HIP_CHECK(hipGraphAddMemcpyNode1D(&node, g.graph, &prev, 1, h_imp + which * IMP_RING, sv.improved.p,
                                          IMP_RING * sizeof(uint32_t), hipMemcpyDeviceToHost));
        prev = node;
      }
      {
        uint32_t a_count = count;
        void* args[] = {&a_ctl, &a_imp, &a_count};
        hipKernelNodeParams kp{};
        kp.func = (void*)sssp_advance_kernel;
        kp.gridDim = dim3(1);
        kp.blockDim = dim3(64);
        kp.kernelParams = args;
        hipGraphNode_t node;
        HIP_CHECK(hipGraphAddKernelNode(&node, g.graph, &prev, 1, &kp));
      }
      HIP_CHECK(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
      std::memcpy(g.key, key, sizeof(key));
      return g.exec;
    };
    struct Batch {
      uint32_t first, count;
      int which;
    };
    uint32_t next_sweep = 0;
    auto enqueue_batch = [&](hipEvent_t ev) {
      // constant small batches while the solve is shallow, larger ones for deep lattices
      Batch b{next_sweep, next_sweep >= 64 ? MAX_BATCH : 8u, next_sweep >= 64 ? 1 : 0};
      HIP_CHECK(hipGraphLaunch(get_graph(b.which, b.count), st));
      HIP_CHECK(hipEventRecord(ev, st));
      next_sweep += b.count;
      return b;
    };
    hipEvent_t evs[2] = {ctx->ev0, ctx->ev1};
    Batch cur = enqueue_batch(evs[0]);
    int which = 0;
    for (;;) {
      if (next_sweep > sweep_cap) throw Error("shortest_path: relaxation did not converge (negative-weight cycle?)");
      // keep the device busy while the host inspects `cur`: the next batch is enqueued first.  Its copy of the
      // flag ring is a superset of cur's (slots are recycled half a ring later), so reading after it is safe.
      const Batch nxt = enqueue_batch(evs[which ^ 1]);
      HIP_CHECK(hipEventSynchronize(evs[which]));
      bool done = false;
      const uint32_t* hf = h_imp + cur.which * IMP_RING;
      for (uint32_t k = 0; k < cur.count; ++k) {
        sweeps_done = cur.first + k + 1;
        if (!hf[(cur.first + k) % IMP_RING]) {
          done = true;
          break;
        }
      }
      if (done) break;
      cur = nxt;
      which ^= 1;
    }
  }
  sv.sweeps = sweeps_done;
  ctx->stats.sweeps = sweeps_done;
}

// Builds the linear output FST exactly as single_shortest_path_backtrace does, including the property
// word (add_state / set_final / add_tr / set_start bookkeeping, then shortest_path_properties(.., true)).
wfst_fst* build_path_fst(wfst_ctx* ctx, bool has_path, uint32_t hops, float final_weight, const wfst_tr* path_arcs) {
  HostCsr h;
  uint64_t p = props::NULL_PROPS;
  uint32_t n_states = 0;
  int64_t start = -1;
  h.offsets.push_back(0);
  if (has_path) {
    n_states = hops + 1;
    h.finals.assign(n_states, INF);
    for (uint32_t k = 0; k <= hops; ++k) {
      p = props::add_state(p);
      if (k == 0) {
        h.finals[0] = final_weight;
        p = props::set_final(p, nullptr, &final_weight);
      } else {
        h.arcs.push_back(path_arcs[k - 1]);
        p = props::add_tr(p, k, path_arcs[k - 1], nullptr);
      }
      h.offsets.push_back((uint32_t)h.arcs.size());
    }
    start = hops;
    p = props::set_start(p);
  }
  p = props::shortest_path(p, true) & props::ALL;
  return make_host_fst(ctx, n_states, start, p, std::move(h));
}

}  // namespace

void shortest_distance(wfst_ctx* ctx, const wfst_fst* f, float* distance, uint32_t* hops) {
  const uint32_t n = f->n_states;
  if (f->start < 0 || n == 0) {
    for (uint32_t i = 0; i < n; ++i) {
      distance[i] = INF;
      if (hops) hops[i] = 0xFFFFFFFFu;
    }
    return;
  }
  ensure_device(const_cast<wfst_fst*>(f));
  Solve sv;
  run_relaxation(ctx, f, sv);
  DBuf<float> d(*ctx->pool, n);
  DBuf<uint32_t> hh(*ctx->pool, n);
  sssp_export_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(sv.key.p, d.p, hops ? hh.p : nullptr, n);
  HIP_CHECK(hipMemcpyAsync(distance, d.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (hops) HIP_CHECK(hipMemcpyAsync(hops, hh.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
}

wfst_fst* shortest_path_n1(wfst_ctx* ctx, const wfst_fst* f) {
  const uint32_t n = f->n_states;
  if (f->start < 0 || n == 0) return build_path_fst(ctx, false, 0, INF, nullptr);  // shortest_path.rs:185-187
  ensure_device(const_cast<wfst_fst*>(f));
  hipStream_t st = ctx->stream;
  Solve sv;
  run_relaxation(ctx, f, sv);
  sssp_final_kernel<<<std::min<uint32_t>((n + 255) / 256, (uint32_t)ctx->n_cus), 256, 0, st>>>(f->dev.finals, sv.key.p, n,
                                                                                                      sv.ctl.p);
  sssp_header_kernel<<<1, 1, 0, st>>>(f->dev.finals, sv.key.p, sv.ctl.p);
  Ctl* hc = (Ctl*)ctx->pinned.get(sizeof(Ctl));
  HIP_CHECK(hipMemcpyAsync(hc, sv.ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  if (!hc->has_path) return build_path_fst(ctx, false, 0, INF, nullptr);
  const uint32_t hops = hc->hops;
  const float final_weight = hc->final_weight;
  std::vector<wfst_tr> path(hops);
  if (hops) {
    DBuf<unsigned long long> parent(*ctx->pool, n);
    HIP_CHECK(hipMemsetAsync(parent.p, 0xFF, (size_t)n * sizeof(unsigned long long), st));
    const uint32_t blocks = std::min<uint32_t>((uint32_t)ctx->n_cus * 8, (uint32_t)(((uint64_t)n * GROUP + 255) / 256));
    sssp_parent_kernel<<<blocks, 256, 0, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, parent.p, n);
    DBuf<wfst_tr> out(*ctx->pool, hops);
    sssp_backtrace_kernel<<<1, 64, 0, st>>>(f->dev.offsets, f->dev.arcs, parent.p, sv.ctl.p, out.p);
    HIP_CHECK(hipMemcpyAsync(path.data(), out.p, (size_t)hops * sizeof(wfst_tr), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
  }
  return build_path_fst(ctx, true, hops, final_weight, path.data());
}

}  // namespace wfst
