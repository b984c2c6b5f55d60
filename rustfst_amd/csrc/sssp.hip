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
#include <cerrno>
#include <cstdio>
#include <sys/stat.h>
#include <sys/types.h>
#include <cstddef>
#include <chrono>
#include <cstring>

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <rocprim/device/device_scan.hpp>

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

// What the tail writes for the host is in host memory before a ticket stored after this is.  (Waiting for the stores'
// acknowledgements alone — `s_waitcnt vmcnt(0)`, without the L2 write-back of the system-scope fence — is NOT enough even
// for fine-grained pinned memory: measured, a 512-problem batch read results that had not landed.  The fence costs the
// tail kernel ~2 us.)
__device__ __forceinline__ void host_stores_done() { __threadfence_system(); }

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
// instead of draining through half a dozen near-empty sweeps.  Activations are counted with ONE atomicAdd per workgroup on a
// counter sharded 16 ways (same-address atomics serialise at ~12 ns each).
constexpr uint32_t RING = 256;
constexpr uint32_t NEAR_SHARDS = 16;
constexpr uint32_t NEAR_RING = 4;     // sweep k writes slot k % 4, reads k-1 and k-2, recycles k+1
constexpr uint32_t NEAR_STRIDE = 32;  // one shard per 128-B line: atomics on one LINE serialise like one address
constexpr uint32_t PROF_STRIDE = 16;  // u64 counters: one shard per 128-B line
constexpr uint32_t NF_STRIDE = 16;
constexpr uint32_t PROF_SHARDS = 64;
constexpr uint32_t IMP_RING = 512;  // per-sweep "something happened" flags, indexed by sweep % IMP_RING

struct Ctl {
  uint32_t base;          // first sweep index of the batch being replayed (graph nodes add their static offset)
  float tau0;             // threshold of sweep 0 (a multiple of delta)
  uint32_t tau[RING];     // f32 bits of the threshold used by sweep k (written by sweep k, read by sweep k+1)
  // number of activations with d <= tau_k made by sweep k: counter sharded 16 ways, shard j at [j * NEAR_STRIDE]
  uint32_t near[NEAR_RING][NEAR_SHARDS * NEAR_STRIDE];
  // atomic sweeps / binned levels: states left in the frontier BEYOND the threshold by sweep k (re-flagged far states and
  // far activations; an upper bound — a state may be counted twice), sharded like `near`.  With `near` it predicts the
  // next level's frontier, which decides the kernel that relaxes it (sssp_binned.h)
  uint32_t far[NEAR_RING][NEAR_SHARDS * NEAR_STRIDE];
  uint32_t streak[RING];  // consecutive sweeps before k that activated nothing near
  // mailbox sweeps: the mode launch k ran in (MODE_*), and the number of states waiting beyond the threshold (sharded like
  // `near`; a block adds the change of its own count, unsigned wrap-around)
  uint32_t mode[RING];
  // mailbox sweeps: what launch k counted, sharded like `near`, one u64 per shard: low word = near activations (the
  // states it expanded), high word = states left waiting beyond the threshold (every block reports its own, asleep or not)
  unsigned long long nf[NEAR_RING][NEAR_SHARDS * NF_STRIDE];
  // arcs / states relaxed so far (profiling only): sharded like `near`, shard j at [j * PROF_STRIDE]
  unsigned long long arcs[PROF_SHARDS * PROF_STRIDE];
  unsigned long long states[PROF_SHARDS * PROF_STRIDE];
  unsigned long long best;    // enc(total) << 32 | final state
  // backtrace header
  uint32_t f_parent, hops;
  float final_weight, total;
  uint32_t has_path, pad;
  // fused tail (sssp_tail_kernel): per-workgroup best final state, and the ticket that tells the last workgroup it is last
  unsigned long long tail_best[128];
  uint32_t tail_ticket, tail_pad;
};
constexpr uint32_t TAIL_BLOCKS = 128;
// what the host needs from the tail of a solve, written straight into pinned memory by sssp_tail_kernel
struct TailOut {
  uint32_t has_path, hops, pad, f_parent;
  float final_weight, total;
  uint32_t done;  // the launch's ticket, written last (after a system-scope fence): the host waits for this word instead of
                  // a HIP event (an event / stream wait retires the stream's finished launches first: ~15 us when ten of them
                  // are waiting, on the critical path of every query)
  uint32_t ties;  // states of the returned path with more than one optimal predecessor (the start state: with any), + 1 when
                  // several final states attain the optimum: 0 = the optimum is unique (wfst_stats.tied_choices)
};

// threshold of sweep k from what sweep k-1 left in the ring (every thread computes the same value)
// Called by one full wave (all 64 lanes): lanes 0..15 fetch the shards of sweep-1's counter, lanes 16..31 those of
// sweep-2's, so the whole decision costs one load latency.
__device__ __forceinline__ float sweep_tau(const Ctl* ctl, uint32_t sweep, float delta, uint32_t near_low,
                                           uint32_t* streak, uint32_t* prev_near, uint32_t* frontier_est = nullptr) {
  *streak = 0;
  *prev_near = 0;
  if (frontier_est) *frontier_est = 1u;
  if (sweep == 0) return ctl->tau0;
  const uint32_t p = (sweep - 1) % RING;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t mine = 0;
  if (lane < NEAR_SHARDS) mine = ctl->near[(sweep - 1) % NEAR_RING][lane * NEAR_STRIDE];
  else if (lane < 2 * NEAR_SHARDS && sweep >= 2) mine = ctl->near[(sweep - 2) % NEAR_RING][(lane - NEAR_SHARDS) * NEAR_STRIDE];
  else if (lane >= 2 * NEAR_SHARDS && lane < 3 * NEAR_SHARDS && frontier_est) mine = ctl->far[(sweep - 1) % NEAR_RING][(lane - 2 * NEAR_SHARDS) * NEAR_STRIDE];
  const float prev = __uint_as_float(ctl->tau[p]);
  const uint32_t prev_streak = ctl->streak[p];
  for (int d = 8; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);  // sums inside each group of 16 lanes
  const uint32_t cnt = __shfl(mine, 0), before = __shfl(mine, 16), far = __shfl(mine, 32);
  *prev_near = cnt;
  float t = prev;
  if (cnt >= near_low) {
    t = prev;
  } else if (cnt) {
    // a near set that cannot fill the GPU AND is shrinking (the tail of a band, not its growing head):
    // widen the band by delta and keep relaxing
    t = cnt < before ? prev + delta : prev;
  } else {
    const uint32_t st = min(prev_streak + 1u, 30u);
    *streak = st;
    t = prev + delta * (float)(1u << (st - 1u));
  }
  // the states this sweep will find near: what the last one activated below the threshold, plus — when the threshold moves —
  // at most everything that waits beyond it
  if (frontier_est) *frontier_est = cnt + (t != prev ? far : 0u);
  return t;
}

#include "sssp_mailbox.h"
#include "sssp_resident.h"
#include "sssp_binned.h"

// Initial state of a solve in ONE launch (five memsets + an init kernel cost five more launch gaps): every key and shadow
// +inf except the start state (d = 1-bar, 0 hops: shortest_path.rs:204), both flag buffers clear except the start state's
// flag in buffer 0, the activity ring and the control block zeroed.
__global__ void __launch_bounds__(256) sssp_setup_kernel(uint64_t* __restrict__ key, uint32_t* __restrict__ shadow,
                                                         uint32_t* __restrict__ flag_words, uint32_t n_flag_words,
                                                         uint32_t* __restrict__ improved, Ctl* __restrict__ ctl, uint32_t n,
                                                         uint32_t start, float tau0) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (uint32_t i = tid; i < n; i += nt) {
    const bool is_start = i == start;
    key[i] = is_start ? (uint64_t)enc_f32(0.0f) << 32 : KEY_INF;
    shadow[i] = is_start ? enc_f32(0.0f) : 0xFFFFFFFFu;
  }
  for (uint32_t i = tid; i < n_flag_words; i += nt) flag_words[i] = i == (start >> 2) ? 1u << (8u * (start & 3u)) : 0u;
  for (uint32_t i = tid; i < IMP_RING; i += nt) improved[i] = 0;
  // control block: rings, counters and header zero, except the first threshold and the best-final key (+inf)
  uint32_t* cw = (uint32_t*)ctl;
  constexpr uint32_t W_TAU0 = offsetof(Ctl, tau0) / 4, W_BEST = offsetof(Ctl, best) / 4;
  for (uint32_t i = tid; i < (uint32_t)(sizeof(Ctl) / 4); i += nt)
    cw[i] = i == W_TAU0 ? __float_as_uint(tau0) : (i == W_BEST || i == W_BEST + 1) ? 0xFFFFFFFFu : 0u;
}

// One sweep: relax every arc leaving the current frontier.
// The frontier is a byte flag per state (idempotent plain stores: no queue, no dedupe atomics); a wave
// scans 64 flags with one coalesced load + ballot, then GROUP lanes share each active state so its arcs
// (contiguous 8-B {w,next} records) are read by consecutive lanes.  The only atomic is the atomicMin of
// relaxations that pass the plain pre-check.
//
// Chasing: a relaxation that improves a target to a NEAR distance does not flag it for the next sweep but puts it on a
// small list private to the wave (LDS); after its flagged states the wave relaxes the listed ones too, up to
// `chase_rounds` states per launch (narrow frontiers are followed deep, growing ones are cut off), spilling to the flags
// whatever does not fit (`chase_cap` entries) or is left over — only in sweeps that follow one with fewer than
// `chase_low` near activations, and only by waves that had at most CHASE_OWN_MAX flagged states of their own.  A sweep
// costs a launch plus a chain of dependent memory trips (~7 us) however small its frontier is, and half of the sweeps
// of a solve are that small (the head and the tail of every band): chasing walks several levels of such a frontier
// inside one launch.  The fixed point does not depend on the order of relaxations, so results are unchanged; an entry
// whose key has been improved again since it was listed is dropped (the improver listed or flagged the state itself).
constexpr uint32_t CHASE_MAX = 128;  // list entries per wave (ring)
constexpr uint32_t CHASE_OWN_MAX = 4;

__global__ void __launch_bounds__(256) sssp_relax_kernel(const uint32_t* __restrict__ offsets,
                                                         const uint2* __restrict__ wn, uint64_t* __restrict__ key,
                                                         uint8_t* __restrict__ flags_cur,
                                                         uint8_t* __restrict__ flags_next, uint32_t n,
                                                         uint32_t* __restrict__ improved_ring, Ctl* __restrict__ ctl,
                                                         uint32_t sweep_offset, float delta, uint32_t near_low,
                                                         uint32_t* __restrict__ shadow, uint32_t chase_cap,
                                                         uint32_t chase_rounds, uint32_t chase_low, uint32_t profile,
                                                         uint32_t dense_low) {
  // `dense_low` (binned levels, sssp_binned.h; 0xFFFFFFFF = never): this launch is the first of its slot — it predicts the
  // level's frontier, publishes the mode, and leaves the level to the binned kernels behind it when the frontier is that large
  // the sweep index is (device-side batch base) + (static offset of this launch / graph node)
  const uint32_t sweep = ctl->base + sweep_offset;
  uint32_t* improved = improved_ring + (sweep % IMP_RING);
  __shared__ uint32_t s_any;   // some activity (improvement or deferral) in this workgroup
  __shared__ uint32_t s_near;  // near activations of this workgroup
  __shared__ float s_tau;
  __shared__ unsigned long long s_prof[2];  // arcs, states relaxed by this workgroup (profiling)
  __shared__ uint32_t s_small;             // the previous sweep left fewer than chase_low near activations
  __shared__ uint32_t s_far, s_dense;
  __shared__ uint2 s_chase[4][CHASE_MAX];  // {state, enc(d) it was listed with}
  const uint32_t slot = sweep % RING;
  const bool bin_on = dense_low != 0xFFFFFFFFu;
  if (threadIdx.x < 64) {
    uint32_t streak, prev_near, est = 0;
    const float t0 = sweep_tau(ctl, sweep, delta, near_low, &streak, &prev_near, bin_on ? &est : nullptr);
    const bool dense = bin_on && sweep > 0 && est >= dense_low;
    if (threadIdx.x == 0) {
      s_small = prev_near < chase_low ? 1u : 0u;
      s_prof[0] = 0;
      s_prof[1] = 0;
      s_tau = t0;
      s_any = 0;
      s_near = 0;
      s_far = 0;
      s_dense = dense ? 1u : 0u;
    }
    if (blockIdx.x == 0) {
      if (threadIdx.x == 0) {
        ctl->tau[slot] = __float_as_uint(t0);
        ctl->streak[slot] = streak;
        if (bin_on) ctl->mode[slot] = dense ? BN_MODE_DENSE : 0u;
      }
      if (threadIdx.x < NEAR_SHARDS) {  // recycle
        ctl->near[(sweep + 1) % NEAR_RING][threadIdx.x * NEAR_STRIDE] = 0;
        if (bin_on) ctl->far[(sweep + 1) % NEAR_RING][threadIdx.x * NEAR_STRIDE] = 0;
      }
    }
  }
  __syncthreads();
  if (s_dense) return;  // a dense level: sssp_bin_expand_kernel / sssp_bin_apply_kernel, queued behind this launch, relax it
  const float tau = s_tau;
  // only sweeps that follow a small one chase (a big sweep is not bound by its launch, and relaxing a state the moment
  // it is first improved, before the rest of the sweep's candidates for it have arrived, costs re-relaxations)
  if (!s_small) chase_cap = 0;
  uint32_t own_states = 0;  // flagged near states this wave relaxed itself
  uint32_t near_cnt = 0, far_cnt = 0;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t sub = lane % GROUP;         // lane inside its group
  const uint32_t grp = lane / GROUP;         // group inside the wave (0..3)
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  uint2* const chase = s_chase[threadIdx.x >> 6];
  uint32_t c_head = 0, c_tail = 0;  // wave-uniform ring positions (monotone; slot = position % CHASE_MAX)
  const uint64_t lanes_below = (1ull << lane) - 1ull;
  unsigned long long p_arcs = 0, p_states = 0;
  bool any = false;

  // the target of a successful relaxation: listed if it is near and there is room, flagged otherwise
  auto activate = [&](bool won, uint32_t t, uint32_t enc_d, float d) {
    const bool near = won && d <= tau;
    if (won) shadow[t] = enc_d;
    bool flag = won && !near;
    if (chase_cap) {
      const uint64_t bm = __ballot(near);
      if (bm) {
        const uint32_t room = chase_cap - (c_tail - c_head);
        const uint32_t rank = (uint32_t)__popcll(bm & lanes_below);
        if (near) {
          if (rank < room) chase[(c_tail + rank) % CHASE_MAX] = make_uint2(t, enc_d);
          else flag = true;
        }
        c_tail += min(room, (uint32_t)__popcll(bm));
      }
    } else {
      flag = won;
    }
    if (flag) {
      flags_next[t] = 1;
      any = true;
      near_cnt += near ? 1u : 0u;
      far_cnt += near ? 0u : 1u;
    }
  };

  // relaxes the arcs of up to 64 states: lane i owns state i of the chunk (act, its key, its arc range)
  auto relax_chunk = [&](bool act, uint64_t my_ks, uint32_t my_b, uint32_t my_e) {
    if (profile && act) {
      if (profile == 1u) p_states += 1;  // (profile == 2: the states column counts atomicMin attempts instead)
      p_arcs += my_e - my_b;
    }
    uint64_t mask = __ballot(act);
    while (mask) {
      // each of the wave's 4 groups takes TWO states per round (the 8 lowest set bits): two independent
      // load -> pre-check -> atomic chains per lane are in flight at once (the kernel is latency bound)
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
      // (shuffles stay outside any lane-dependent condition: a bpermute reads 0 from a lane that is masked off)
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
        // Plain pre-check against a 4-byte SHADOW of the distance half of the key (4 MB for 1M states: half the bytes
        // per gather and a far better L2 hit rate than the 8-byte keys).  shadow[t] is only ever written with values
        // that an atomicMin has installed in key[t], and keys only decrease, so shadow[t] >= enc(d[t]) at all times:
        // a candidate above it cannot improve; a stale (too high) shadow only costs an extra atomic; on equal
        // distance the hop half decides and the real key is read.
        const uint32_t eca = (uint32_t)(cka >> 32), ecb = (uint32_t)(ckb >> 32);
        uint32_t sa = 0, sb = 0;
        if (fa) sa = shadow[aa.y];
        if (fb) sb = shadow[ab.y];
        bool ta = fa && eca <= sa, tb = fb && ecb <= sb;
        if (ta && eca == sa) ta = cka < key[aa.y];
        if (tb && ecb == sb) tb = ckb < key[ab.y];
        uint64_t olda = 0, oldb = 0;
        if (ta) olda = atomicMin((unsigned long long*)&key[aa.y], (unsigned long long)cka);
        if (tb) oldb = atomicMin((unsigned long long*)&key[ab.y], (unsigned long long)ckb);
        if (profile == 2u) p_states += (ta ? 1u : 0u) + (tb ? 1u : 0u);
        activate(ta && cka < olda, aa.y, eca, ca);
        activate(tb && ckb < oldb, ab.y, ecb, cb);
        ia += GROUP;
        ib += GROUP;
      }
    }
  };

  for (uint32_t base = wave * 64u; base < n; base += n_waves * 64u) {
    const uint32_t sc = base + lane;
    bool act = sc < n && flags_cur[sc] != 0;
    if (__ballot(act) == 0) continue;
    // the lane that owns an active state fetches everything the relaxation of that state needs: ONE memory
    // round trip for the whole 64-state chunk; the groups get it by shuffle
    uint64_t my_ks = 0;
    uint32_t my_b = 0, my_e = 0;
    if (act) {
      flags_cur[sc] = 0;  // this buffer is the NEXT frontier two sweeps from now
      my_ks = key[sc];
      my_b = offsets[sc];
      my_e = offsets[sc + 1];
      if (dec_f32((uint32_t)(my_ks >> 32)) > tau) {  // far: stays in the frontier, is not relaxed in this sweep
        flags_next[sc] = 1;
        any = true;
        act = false;
        far_cnt += 1u;
      }
    }
    own_states += (uint32_t)__popcll(__ballot(act));
    relax_chunk(act, my_ks, my_b, my_e);
  }
  // a wave that had a real share of a big frontier (the band was widened) leaves its discoveries to the next sweep
  if (own_states > CHASE_OWN_MAX) chase_rounds = 0;
  // the wave's own discoveries (wave-synchronous: the list is private to the wave, LDS accesses of one wave are ordered)
  for (uint32_t chased = 0; c_tail != c_head;) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint32_t take = min(64u, c_tail - c_head);
    uint2 e = make_uint2(0u, 0u);
    if (lane < take) e = chase[(c_head + lane) % CHASE_MAX];
    c_head += take;
    if (chased >= chase_rounds) {  // budget spent: what is left goes to the next sweep
      if (lane < take) {
        flags_next[e.x] = 1;
        any = true;
        near_cnt += 1;
      }
      continue;
    }
    chased += take;
    bool act = lane < take;
    uint64_t my_ks = 0;
    uint32_t my_b = 0, my_e = 0;
    if (act) {
      my_ks = key[e.x];
      my_b = offsets[e.x];
      my_e = offsets[e.x + 1];
      act = (uint32_t)(my_ks >> 32) == e.y;  // improved again since: whoever did that listed or flagged it
    }
    relax_chunk(act, my_ks, my_b, my_e);
  }
  // one conditional plain store + one sharded atomicAdd per workgroup (thousands of same-address atomics per
  // sweep would serialise at ~12 ns each)
  for (int d = 32; d >= 1; d >>= 1) near_cnt += __shfl_xor(near_cnt, d);
  if (bin_on)
    for (int d = 32; d >= 1; d >>= 1) far_cnt += __shfl_xor(far_cnt, d);
  const bool wave_any = __any(any);
  if (lane == 0) {
    if (wave_any) s_any = 1u;
    if (near_cnt) atomicAdd(&s_near, near_cnt);
    if (bin_on && far_cnt) atomicAdd(&s_far, far_cnt);
  }
  if (profile) {
    for (int d = 32; d >= 1; d >>= 1) {
      p_arcs += __shfl_xor(p_arcs, d);
      p_states += __shfl_xor(p_states, d);
    }
    if (lane == 0 && (p_states | p_arcs)) {
      atomicAdd(&s_prof[0], p_arcs);
      atomicAdd(&s_prof[1], p_states);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_any && *improved == 0u) *improved = 1u;
    if (s_near) atomicAdd(&ctl->near[sweep % NEAR_RING][(blockIdx.x % NEAR_SHARDS) * NEAR_STRIDE], s_near);
    if (s_far) atomicAdd(&ctl->far[sweep % NEAR_RING][(blockIdx.x % NEAR_SHARDS) * NEAR_STRIDE], s_far);
    if (profile && (s_prof[0] | s_prof[1])) {
      atomicAdd(&ctl->arcs[(blockIdx.x % PROF_SHARDS) * PROF_STRIDE], s_prof[0]);
      atomicAdd(&ctl->states[(blockIdx.x % PROF_SHARDS) * PROF_STRIDE], s_prof[1]);
    }
  }
}

// profiling helper: keeps the GPU busy right before the event that opens a timed sweep, as the previous sweep does in an
// un-profiled solve (a launch into an idle GPU takes ~3 us longer)
__global__ void sssp_nop_kernel(const uint8_t* __restrict__ flags, uint32_t n, uint32_t* sink) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i] == 0xFF) *sink = i;  // flags are 0 / 1: never true
}

// closes a batch: the next replay continues at base + count, and the flag slots half a ring ahead are recycled
// and the batch's flags are mirrored into pinned host memory (plain stores over the bus: cheaper than a copy node)
__global__ void sssp_advance_kernel(Ctl* ctl, uint32_t* improved_ring, uint32_t count, uint32_t* host_ring) {
  const uint32_t base = ctl->base;
  for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
    if (host_ring) host_ring[(base + i) % IMP_RING] = improved_ring[(base + i) % IMP_RING];
    improved_ring[(base + IMP_RING / 2 + i) % IMP_RING] = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) ctl->base = base + count;
}

// f_parent = argmin over final states of (d[s] (x) rho(s), s)      (shortest_path.rs:214-220)
__global__ void __launch_bounds__(256) sssp_final_kernel(const float* __restrict__ finals, const uint64_t* __restrict__ key,
                                                        uint32_t n, Ctl* __restrict__ ctl) {
  __shared__ unsigned long long s_best[4];
  unsigned long long best = KEY_INF;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const float f = finals[s];
    if (!(f < INF)) continue;  // most states are not final: their key is never fetched
    const uint64_t k = key[s];
    if (k == KEY_INF) continue;
    const float tot = (dec_f32((uint32_t)(k >> 32)) + f) + 0.0f;
    if (!(tot < INF)) continue;
    const unsigned long long c = ((unsigned long long)enc_f32(tot) << 32) | s;
    best = c < best ? c : best;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(best, d);
    best = o < best ? o : best;
  }
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) best = s_best[w] < best ? s_best[w] : best;
    // one atomic per workgroup at most, and only if it can still lower the result (same-address atomics cost ~12 ns each)
    if (best != KEY_INF && best < __hip_atomic_load(&ctl->best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMin(&ctl->best, best);
  }
}

// Predecessor rule.  Class 0: the arc is tight in both halves of the key, (d[s] (x) w, hops[s] + 1) ==
// (d[t], hops[t]) — on weights whose f32 sums are exact every reached state but the start has such an arc.  Class 1, the
// fallback: the arc is tight in the distance and the source's key is lexicographically below the target's.  With
// inexact sums a state can keep a hop count it got from a label of its predecessor that was later improved in the
// distance (two different distances of the source can round to the same sum): then no arc is tight in the hop count,
// but the arc that produced the label is still tight in the distance and, for non-negative weights, its source's key is
// smaller, so a class-1 arc exists.  Keys strictly decrease along either class: the predecessor graph is acyclic.
constexpr unsigned long long PARENT_NONE = ~0ull;
__device__ __forceinline__ unsigned long long parent_class(uint64_t cand_key, uint64_t key_s, uint64_t key_t) {
  if (cand_key == key_t) return 0ull;
  if ((cand_key >> 32) == (key_t >> 32) && key_s < key_t) return 1ull << 63;
  return PARENT_NONE;
}

// parent[t] = min (class, s, pos) over the arcs parent_class admits
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
      const uint64_t ck = ((uint64_t)enc_f32(c) << 32) | h1, kt = key[a.y];
      const unsigned long long cls = parent_class(ck, ks, kt);
      if (cls != PARENT_NONE) atomicMin(&parent[a.y], cls | ((unsigned long long)s << 32) | (i - b));
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
                                      const uint64_t* __restrict__ key, const unsigned long long* __restrict__ parent,
                                      Ctl* __restrict__ ctl, wfst_tr* __restrict__ out, uint32_t out_cap) {
  if (threadIdx.x || blockIdx.x) return;
  uint32_t cur = ctl->f_parent, k = 0;
  // the walk ends at the start state, the only one whose key has no arcs in it (with class-1 predecessors its length
  // need not be the hop count of the final state's key)
  while ((uint32_t)key[cur] != 0u) {
    const unsigned long long p = parent[cur];
    if (p == PARENT_NONE) {  // no admissible predecessor (inexact sums with negative weights)
      ctl->pad |= 4u;
      return;
    }
    if (k >= out_cap) {  // longer than the buffer the host sized from the hop count: it retries with room for n arcs
      ctl->pad |= 16u;
      return;
    }
    const uint32_t s = (uint32_t)(p >> 32) & 0x7FFFFFFFu, pos = (uint32_t)p;
    wfst_tr tr = arcs[offsets[s] + pos];
    tr.nextstate = k;
    out[k] = tr;
    cur = s;
    ++k;
  }
  ctl->hops = k;
}

// ---- transpose of the CSR, cached on the FST handle once it is queried again: with it the
// canonical predecessor of the ~hops states ON the path is found by looking at their in-arcs only, instead of the
// parent pass over all arcs (105 us on 10M arcs).
__global__ void __launch_bounds__(256) rev_count_kernel(const uint2* __restrict__ wn, uint64_t n_arcs, uint32_t* __restrict__ indeg) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_arcs; i += (uint64_t)gridDim.x * blockDim.x)
    atomicAdd(&indeg[wn[i].y], 1u);
}
__global__ void __launch_bounds__(256) rev_fill_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn,
                                                      uint32_t n, uint32_t* __restrict__ cursor, uint4* __restrict__ rev_arc) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = tid % GROUP;
  const uint32_t n_groups = gridDim.x * blockDim.x / GROUP;
  for (uint32_t s = tid / GROUP; s < n; s += n_groups) {
    const uint32_t b = offsets[s], e = offsets[s + 1];
    for (uint32_t i = b + lane; i < e; i += GROUP) {
      const uint2 a = wn[i];
      rev_arc[atomicAdd(&cursor[a.y], 1u)] = make_uint4(s, i - b, a.x, 0u);
    }
  }
}

// ---- the same transpose WITHOUT a global atomic, for FSTs that have a mailbox plan (sssp_mailbox.h): region (i -> j) of the
// plan holds one slot per arc from block i to block j, destination-major — so the regions of destination block j, taken
// together, ARE the in-arcs of block j's states, already contiguous and already at their final place in rev_arc; what is
// left is the order inside the block.  rev_bucket_kernel: workgroup i writes a record {source, position, weight, target mod B}
// per arc of block i into its slot (one LDS atomicAdd on the region's cursor; the records of a region fill its lines front to
// back); rev_place_kernel: workgroup j counts its records per target state in LDS, scans, writes rev_off, and moves every
// record to its target's run.  10 M arcs: ~0.2 ms against ~1.2 ms for the two atomic passes above.
template <uint32_t LOG>
__global__ void __launch_bounds__(1024) rev_bucket_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn, uint32_t n,
                                                          uint32_t nb, const uint32_t* __restrict__ roff_t, uint4* __restrict__ rec) {
  constexpr uint32_t B = 1u << LOG;
  extern __shared__ uint32_t l_cur[];  // [nb] next slot of region (i -> j)
  const uint32_t i = blockIdx.x, s0 = i << LOG, s1 = min(n, s0 + B);
  for (uint32_t j = threadIdx.x; j < nb; j += blockDim.x) l_cur[j] = roff_t[(size_t)i * nb + j];
  __syncthreads();
  // 16 lanes per state, four states of a group in flight (their row bounds, then their first 16 arcs, are requested together:
  // one workgroup per compute unit, so the loads in flight per thread are the bandwidth)
  const uint32_t sub = threadIdx.x & 15u, grp = threadIdx.x >> 4;
  constexpr uint32_t U = 4, G = 1024 / 16;
  for (uint32_t sb = s0 + grp; sb < s1; sb += U * G) {
    uint32_t b[U], e[U];
    uint2 a[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      const uint32_t s = sb + u * G;
      b[u] = s < s1 ? offsets[s] : 0u;
      e[u] = s < s1 ? offsets[s + 1] : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) a[u] = b[u] + sub < e[u] ? wn[b[u] + sub] : make_uint2(0u, 0u);
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      const uint32_t s = sb + u * G;
      if (b[u] + sub < e[u]) {
        const uint32_t slot = atomicAdd(&l_cur[a[u].y >> LOG], 1u);
        rec[slot] = make_uint4(s, sub, a[u].x, a[u].y & (B - 1u));
      }
      for (uint32_t k = b[u] + sub + 16u; k < e[u]; k += 16u) {  // (rows beyond 16 arcs)
        const uint2 x = wn[k];
        const uint32_t slot = atomicAdd(&l_cur[x.y >> LOG], 1u);
        rec[slot] = make_uint4(s, k - b[u], x.x, x.y & (B - 1u));
      }
    }
  }
}
template <uint32_t LOG>
__global__ void __launch_bounds__(1024) rev_place_kernel(const uint32_t* __restrict__ roff, uint32_t nb, uint32_t n,
                                                         const uint4* __restrict__ rec, uint32_t* __restrict__ rev_off,
                                                         uint4* __restrict__ rev_arc) {
  constexpr uint32_t B = 1u << LOG, R = B / 1024;
  __shared__ uint32_t l_cnt[B];
  __shared__ uint32_t s_wsum[16];
  const uint32_t j = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t lo = roff[(size_t)j * nb], hi = roff[(size_t)(j + 1) * nb];  // all the in-arcs of block j
  for (uint32_t t = tid; t < B; t += 1024) l_cnt[t] = 0;
  __syncthreads();
  // (four records per thread in flight: one workgroup per compute unit is all the parallelism there is here)
  {
    uint32_t k = lo + tid;
    for (; k + 3u * 1024u < hi; k += 4u * 1024u) {
      const uint32_t t0 = rec[k].w, t1 = rec[k + 1024u].w, t2 = rec[k + 2048u].w, t3 = rec[k + 3072u].w;
      atomicAdd(&l_cnt[t0], 1u);
      atomicAdd(&l_cnt[t1], 1u);
      atomicAdd(&l_cnt[t2], 1u);
      atomicAdd(&l_cnt[t3], 1u);
    }
    for (; k < hi; k += 1024u) atomicAdd(&l_cnt[rec[k].w], 1u);
  }
  __syncthreads();
  // exclusive scan of the B counts: R consecutive counts per thread, a wave scan of the thread sums, the wave totals
  uint32_t c[R], mine = 0;
  for (uint32_t r = 0; r < R; ++r) {
    c[r] = l_cnt[tid * R + r];
    mine += c[r];
  }
  uint32_t x = mine;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d);
    if (lane >= (uint32_t)d) x += y;
  }
  if (lane == 63) s_wsum[wv] = x;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wv; ++w) base += s_wsum[w];
  uint32_t run = base + x - mine;
  const uint32_t s0 = j << LOG;
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t tl = tid * R + r;
    l_cnt[tl] = run;  // from here on: the next free position of the target's run
    if (s0 + tl < n) rev_off[s0 + tl] = lo + run;
    run += c[r];
  }
  if (j == gridDim.x - 1 && tid == 0) rev_off[n] = hi;
  __syncthreads();
  {
    uint32_t k = lo + tid;
    for (; k + 3u * 1024u < hi; k += 4u * 1024u) {
      const uint4 r0 = rec[k], r1 = rec[k + 1024u], r2 = rec[k + 2048u], r3 = rec[k + 3072u];
      const uint32_t p0 = atomicAdd(&l_cnt[r0.w], 1u), p1 = atomicAdd(&l_cnt[r1.w], 1u);
      const uint32_t p2 = atomicAdd(&l_cnt[r2.w], 1u), p3 = atomicAdd(&l_cnt[r3.w], 1u);
      rev_arc[lo + p0] = make_uint4(r0.x, r0.y, r0.z, 0u);
      rev_arc[lo + p1] = make_uint4(r1.x, r1.y, r1.z, 0u);
      rev_arc[lo + p2] = make_uint4(r2.x, r2.y, r2.z, 0u);
      rev_arc[lo + p3] = make_uint4(r3.x, r3.y, r3.z, 0u);
    }
    for (; k < hi; k += 1024u) {
      const uint4 r = rec[k];
      rev_arc[lo + atomicAdd(&l_cnt[r.w], 1u)] = make_uint4(r.x, r.y, r.z, 0u);
    }
  }
}

// single_shortest_path_backtrace (shortest_path.rs:241-282) over the transpose, by ONE wave: at every step the lanes test
// the in-arcs of the current state for tightness, (d[s] (x) w, hops[s] + 1) == (d[t], hops[t]), and the smallest
// (class, s, pos) wins — the same predecessor sssp_parent_kernel selects.  A step is TWO dependent trips: the in-arc
// entries (source, position, weight: nothing else of the arc is needed), then the sources' keys together with their own
// in-arc ranges (the winner's is the next step's).  The arcs of the walk themselves are fetched afterwards, all at once
// (their (state, position) wait in LDS), and go straight to `out` (pinned host memory when the path fits in it).
// Returns the number of steps taken; pad |= 4: no admissible predecessor, pad |= 8: longer than out_cap.
constexpr uint32_t WALK_LDS = 1024;
__device__ __forceinline__ uint32_t sssp_walk_back(const uint32_t* __restrict__ offsets, const wfst_tr* __restrict__ arcs,
                                                   const uint64_t* __restrict__ key, const uint32_t* __restrict__ rev_off,
                                                   const uint4* __restrict__ rev_arc, uint32_t fp, wfst_tr* __restrict__ out,
                                                   uint32_t out_cap, uint32_t& pad, uint2* s_walk /* LDS [WALK_LDS] */, uint32_t& ties) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t k = 0;
  uint64_t kt = key[fp];
  uint32_t rb = rev_off[fp], re = rev_off[fp + 1];
  // `ties`: states on the walk with more than one in-arc that is tight in the DISTANCE (d[s] (x) w == d[t]: any of them
  // continues an optimal path, whatever rule picks one) — the start state: with any (a zero-weight cycle through it)
  for (;;) {
    if ((uint32_t)kt == 0u) {  // the start state
      uint32_t tn = 0;
      for (uint32_t j = rb + lane; j < re; j += 64) {
        const uint4 ra = rev_arc[j];
        const uint64_t ks = key[ra.x];
        if (ks == KEY_INF) continue;
        const float c = (dec_f32((uint32_t)(ks >> 32)) + __uint_as_float(ra.z)) + 0.0f;
        if (c < INF && enc_f32(c) == (uint32_t)(kt >> 32)) tn += 1u;
      }
      if (__any(tn != 0u)) ties += 1u;
      break;
    }
    if (k >= out_cap) {
      pad |= 8u;
      break;
    }
    unsigned long long bp = PARENT_NONE;
    uint64_t my_ks = 0;
    uint32_t my_rb = 0, my_re = 0;
    uint32_t tn = 0;
    for (uint32_t j = rb + lane; j < re; j += 64) {
      const uint4 ra = rev_arc[j];
      const uint64_t ks = key[ra.x];
      const uint32_t sb = rev_off[ra.x], se = rev_off[ra.x + 1];
      if (ks == KEY_INF) continue;
      const float c = (dec_f32((uint32_t)(ks >> 32)) + __uint_as_float(ra.z)) + 0.0f;
      if (!(c < INF)) continue;
      if (enc_f32(c) == (uint32_t)(kt >> 32)) tn += 1u;
      const uint64_t ck = ((uint64_t)enc_f32(c) << 32) | ((uint32_t)ks + 1u);
      const unsigned long long cls = parent_class(ck, ks, kt);
      if (cls != PARENT_NONE) {
        const unsigned long long cand = cls | ((unsigned long long)ra.x << 32) | ra.y;
        if (cand < bp) {
          bp = cand;
          my_ks = ks;
          my_rb = sb;
          my_re = se;
        }
      }
    }
    unsigned long long best = bp;
    for (int d = 32; d >= 1; d >>= 1) {
      const unsigned long long o = __shfl_xor(best, d);
      best = o < best ? o : best;
    }
    if (best == PARENT_NONE) {  // no admissible predecessor: reported, not followed
      pad |= 4u;
      break;
    }
    {
      const unsigned long long tm = __ballot(tn != 0u);
      if (__popcll(tm) > 1 || __any(tn > 1u)) ties += 1u;
    }
    const int wl = __ffsll((unsigned long long)__ballot(bp == best)) - 1;  // (source, position) is unique: one lane holds it
    kt = __shfl(my_ks, wl);
    rb = __shfl(my_rb, wl);
    re = __shfl(my_re, wl);
    const uint32_t s = (uint32_t)(best >> 32) & 0x7FFFFFFFu, pos = (uint32_t)best;
    if (lane == 0) {
      if (k < WALK_LDS) {
        s_walk[k] = make_uint2(s, pos);
      } else {  // (a path longer than the list: its arcs are fetched on the way, one dependent trip more per step)
        wfst_tr tr = arcs[offsets[s] + pos];
        tr.nextstate = k;
        out[k] = tr;
      }
    }
    ++k;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // lane 0's list is in LDS before the wave reads it back
  for (uint32_t i = lane; i < min(k, WALK_LDS); i += 64) {
    const uint2 e = s_walk[i];
    wfst_tr tr = arcs[offsets[e.x] + e.y];
    tr.nextstate = i;
    out[i] = tr;
  }
  return k;
}

__global__ void __launch_bounds__(64) sssp_backtrace_rev_kernel(const uint32_t* __restrict__ offsets,
                                                               const wfst_tr* __restrict__ arcs, const uint64_t* __restrict__ key,
                                                               const uint32_t* __restrict__ rev_off,
                                                               const uint4* __restrict__ rev_arc, Ctl* __restrict__ ctl,
                                                               wfst_tr* __restrict__ out, uint32_t out_cap) {
  __shared__ uint2 s_walk[WALK_LDS];
  const uint32_t lane = threadIdx.x;
  if (!ctl->has_path) return;
  if (ctl->hops > out_cap) {  // the host falls back to the parent pass
    if (lane == 0) ctl->pad |= 8u;
    return;
  }
  uint32_t pad = 0, ties = 0;
  const uint32_t k = sssp_walk_back(offsets, arcs, key, rev_off, rev_arc, ctl->f_parent, out, out_cap, pad, s_walk, ties);
  if (lane == 0) {
    if (pad) ctl->pad |= pad;
    else ctl->hops = k;  // the real length of the walk
  }
}

// The tail of a repeated query in ONE launch: arg-min over the final states (sssp_final_kernel), the header
// (sssp_header_kernel) and the walk over the transpose (sssp_backtrace_rev_kernel), with the result header written into
// pinned host memory next to the path arcs.  Every workgroup leaves the best final state of its share; the workgroup
// that draws the last ticket reduces them and walks back.  (Three launches, a 1-thread kernel and a copy of the whole
// control block cost ~30 us at the end of every solve; 1024 workgroups each polling and lowering ONE word were most of
// the final-state search: same-address atomics serialise at ~12 ns.)
__global__ void __launch_bounds__(1024) sssp_tail_kernel(const float* __restrict__ finals, const uint64_t* __restrict__ key,
                                                         uint32_t n, Ctl* __restrict__ ctl,
                                                         const uint32_t* __restrict__ offsets, const wfst_tr* __restrict__ arcs,
                                                         const uint32_t* __restrict__ rev_off,
                                                         const uint4* __restrict__ rev_arc, wfst_tr* __restrict__ out,
                                                         uint32_t out_cap, TailOut* __restrict__ hout,
                                                         uint32_t* __restrict__ improved_ring, uint32_t adv_count,
                                                         uint32_t* __restrict__ host_ring, uint32_t done_ticket) {
  __shared__ unsigned long long s_best[16];
  __shared__ uint2 s_walk[WALK_LDS];
  __shared__ uint32_t s_last;
  const uint32_t lane = threadIdx.x & 63u;
  // (enc(total) << 32 | final state) and whether ANOTHER final state attains the same total: merged pairwise; between
  // workgroups the flag travels in bit 31 of the state word (state ids have 31 bits)
  constexpr unsigned long long TIE_BIT = 1ull << 31;
  auto merge = [](unsigned long long& best, bool& tie, unsigned long long o, bool o_tie) {
    if (o == KEY_INF) return;
    if ((o >> 32) == (best >> 32)) {
      tie = true;
      best = o < best ? o : best;
    } else if (o < best) {
      best = o;
      tie = o_tie;
    }
  };
  unsigned long long best = KEY_INF;
  bool tie = false;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const float f = finals[s];
    if (!(f < INF)) continue;  // most states are not final: their key is never fetched
    const uint64_t k = key[s];
    if (k == KEY_INF) continue;
    const float tot = (dec_f32((uint32_t)(k >> 32)) + f) + 0.0f;  // d[s] (x) rho(s), shortest_path.rs:214-220
    if (!(tot < INF)) continue;
    merge(best, tie, ((unsigned long long)enc_f32(tot) << 32) | s, false);
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(best, d);
    const bool ot = __shfl_xor((int)tie, d) != 0;
    merge(best, tie, o, ot);
  }
  if (lane == 0) s_best[threadIdx.x >> 6] = best == KEY_INF ? best : (best | (tie ? TIE_BIT : 0ull));
  __syncthreads();
  if (threadIdx.x == 0) {
    best = KEY_INF;
    tie = false;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {
      const unsigned long long o = s_best[w];
      merge(best, tie, o == KEY_INF ? o : (o & ~TIE_BIT), o != KEY_INF && (o & TIE_BIT) != 0ull);
    }
    __hip_atomic_store(&ctl->tail_best[blockIdx.x], best == KEY_INF ? best : (best | (tie ? TIE_BIT : 0ull)), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = atomicAdd(&ctl->tail_ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last || threadIdx.x >= 64) return;
  if (adv_count) {
    // closes the sweep batch queued in front of this launch (what sssp_advance_kernel does: the flags of its sweeps go to
    // pinned host memory, the slots half a ring ahead are recycled, the base moves on) — one launch and one flush of
    // host-memory writes less at the end of a predicted solve.  By the wave that writes the ticket at the end: ONE
    // system-scope fence orders all of it.
    const uint32_t base = ctl->base;
    for (uint32_t i = lane; i < adv_count; i += 64) {
      host_ring[(base + i) % IMP_RING] = improved_ring[(base + i) % IMP_RING];
      improved_ring[(base + IMP_RING / 2 + i) % IMP_RING] = 0;
    }
    if (lane == 0) ctl->base = base + adv_count;  // (every lane of the wave has read the old value above)
  }
  // the last workgroup's first wave: every other workgroup's result is in memory
  best = KEY_INF;
  tie = false;
  for (uint32_t b = lane; b < gridDim.x; b += 64) {
    const unsigned long long o = __hip_atomic_load(&ctl->tail_best[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    merge(best, tie, o == KEY_INF ? o : (o & ~TIE_BIT), o != KEY_INF && (o & TIE_BIT) != 0ull);
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(best, d);
    const bool ot = __shfl_xor((int)tie, d) != 0;
    merge(best, tie, o, ot);
  }
  uint32_t pad = ctl->pad;
  if (lane == 0) {
    ctl->tail_ticket = 0;  // the tail may run again on the same control block (a solve that outran its prediction)
    ctl->best = best;
  }
  if (best == KEY_INF) {
    if (lane == 0) {
      ctl->has_path = 0;
      ctl->hops = 0;
      hout->has_path = 0u;
      hout->hops = 0u;
      hout->pad = pad;
      hout->f_parent = 0u;
      hout->final_weight = INF;
      hout->total = INF;
      hout->ties = 0u;
      host_stores_done();
      __hip_atomic_store(&hout->done, done_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const uint32_t fp = (uint32_t)best;
  const float final_weight = finals[fp], total = dec_f32((uint32_t)(best >> 32));
  uint32_t k = 0, ties = tie ? 1u : 0u;
  if ((uint32_t)key[fp] > out_cap) pad |= 8u;  // the host falls back to the parent pass
  else k = sssp_walk_back(offsets, arcs, key, rev_off, rev_arc, fp, out, out_cap, pad, s_walk, ties);
  if (lane == 0) {
    ctl->has_path = 1;
    ctl->f_parent = fp;
    ctl->hops = (pad & 12u) ? (uint32_t)key[fp] : k;
    ctl->final_weight = final_weight;
    ctl->total = total;
    // (the walk's own flags go to the host only: a tail that ran ahead of the last sweeps is run again on this block)
    hout->has_path = 1u;
    hout->hops = (pad & 12u) ? (uint32_t)key[fp] : k;
    hout->pad = pad;
    hout->f_parent = fp;
    hout->final_weight = final_weight;
    hout->total = total;
    hout->ties = ties;
  }
  // (the walk's arcs were written by several lanes of this wave)
  host_stores_done();
  if (lane == 0) __hip_atomic_store(&hout->done, done_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the reference's tie order on acyclic inputs (wfst_ctx_set_tie_order; include/wfst.h).  rank[s] = position of s in
// the topological order of the reference's depth-first visit (host), r2s = its inverse.
// f_parent: the FIRST state in that order whose d[s] (x) rho(s) attains the minimum (shortest_path.rs:214-220: only a
// strict improvement replaces f_parent, and states are dequeued in that order)
__global__ void __launch_bounds__(256) sssp_final_ref_kernel(const float* __restrict__ finals, const uint64_t* __restrict__ key,
                                                            const uint32_t* __restrict__ rank, uint32_t n, Ctl* __restrict__ ctl) {
  unsigned long long best = KEY_INF;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const float f = finals[s];
    if (!(f < INF)) continue;
    const uint64_t k = key[s];
    if (k == KEY_INF) continue;
    const float tot = (dec_f32((uint32_t)(k >> 32)) + f) + 0.0f;
    if (!(tot < INF)) continue;
    const unsigned long long c = ((unsigned long long)enc_f32(tot) << 32) | rank[s];
    best = c < best ? c : best;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(best, d);
    best = o < best ? o : best;
  }
  if ((threadIdx.x & 63) == 0 && best != KEY_INF) atomicMin(&ctl->best, best);
}
// parent[t]: the FIRST arc, in (rank of the source, arc position), whose candidate d[s] (x) w equals d[t] — every later arc
// with the same candidate is no strict improvement and leaves the parent alone (shortest_path.rs:222-232)
__global__ void __launch_bounds__(256) sssp_parent_ref_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn,
                                                             const uint64_t* __restrict__ key, const uint32_t* __restrict__ rank,
                                                             unsigned long long* __restrict__ parent, uint32_t n) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = tid % GROUP;
  const uint32_t n_groups = gridDim.x * blockDim.x / GROUP;
  for (uint32_t s = tid / GROUP; s < n; s += n_groups) {
    const uint64_t ks = key[s];
    if (ks == KEY_INF) continue;
    const float d = dec_f32((uint32_t)(ks >> 32));
    const uint32_t b = offsets[s], e = offsets[s + 1];
    const unsigned long long rs = (unsigned long long)rank[s] << 32;
    for (uint32_t i = b + lane; i < e; i += GROUP) {
      const uint2 a = wn[i];
      const float c = (d + __uint_as_float(a.x)) + 0.0f;
      if (!(c < INF)) continue;
      if (enc_f32(c) == (uint32_t)(key[a.y] >> 32)) atomicMin(&parent[a.y], rs | (i - b));
    }
  }
}
__global__ void sssp_backtrace_ref_kernel(const uint32_t* __restrict__ offsets, const wfst_tr* __restrict__ arcs,
                                          const float* __restrict__ finals, const unsigned long long* __restrict__ parent,
                                          const uint32_t* __restrict__ r2s, uint32_t start, Ctl* __restrict__ ctl,
                                          wfst_tr* __restrict__ out, uint32_t out_cap) {
  if (threadIdx.x || blockIdx.x) return;
  const unsigned long long best = ctl->best;
  if (best == KEY_INF) {
    ctl->has_path = 0;
    ctl->hops = 0;
    return;
  }
  uint32_t cur = r2s[(uint32_t)best], k = 0;
  ctl->has_path = 1;
  ctl->f_parent = cur;
  ctl->final_weight = finals[cur];
  ctl->total = dec_f32((uint32_t)(best >> 32));
  while (cur != start) {
    const unsigned long long p = parent[cur];
    if (p == PARENT_NONE || k >= out_cap) {
      ctl->pad |= 4u;
      return;
    }
    const uint32_t s = r2s[(uint32_t)(p >> 32)], pos = (uint32_t)p;
    wfst_tr tr = arcs[offsets[s] + pos];
    tr.nextstate = k;
    out[k] = tr;
    cur = s;
    ++k;
  }
  ctl->hops = k;
}

__global__ void sssp_export_kernel(const uint64_t* __restrict__ key, float* __restrict__ dist, uint32_t* __restrict__ hops,
                                   uint32_t n) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint64_t k = key[s];
  dist[s] = k == KEY_INF ? INF : dec_f32((uint32_t)(k >> 32));
  if (hops) hops[s] = k == KEY_INF ? 0xFFFFFFFFu : (uint32_t)k;
}

// Resident solves of a device share a lease of TWO units: a grid that waits for its own workgroups must be resident as a
// whole, so a whole-device grid (one workgroup on ~245 of 256 compute units) takes both units, and a half-device grid
// (wfst_ctx_set_resident_share(ctx, 1): at most 128 workgroups) takes one — two of those fit the device together, and two
// contexts that answer queries on half the device each run their solves side by side (tools/two_queries.py).  Inside a
// process the lease is a counter per device; between processes it is an advisory lock on a file per device, held by a
// process while any of its solves holds a unit (flock, non-blocking: whoever does not get it takes one launch per level
// for that solve).  Neither covers a tenant that is not this library: the wait limit of the launch does.
struct ResidentLease {
  int held_device = -1, held_units = 0;
  ResidentLease() = default;
  ResidentLease(const ResidentLease&) = delete;
  ResidentLease& operator=(const ResidentLease&) = delete;
  // One descriptor per device and process (-1: no lock file, in-process lease only).  Opened read-only (flock works on a
  // read-only descriptor, so a file another user created is still lockable), never through a symbolic link, made world-readable
  // by whoever creates it whatever the umask; reopened after a fork (a child shares its parent's open file description, and
  // flock locks belong to the description: parent and child would both "hold" it).  Without a lock file two processes can
  // both start resident grids; the launch's wait limit is what then protects them, and that is said once on stderr.
  static int device_lock_fd(int device) {
    struct Slot { std::mutex mu; int fd = -1; pid_t pid = 0; bool tried = false; };
    static Slot slots[64];
    const unsigned d = (unsigned)device & 63u;
    Slot& sl = slots[d];
    std::lock_guard<std::mutex> lk(sl.mu);
    const pid_t me = ::getpid();
    if (sl.tried && sl.pid == me) return sl.fd;
    if (sl.tried && sl.fd >= 0) ::close(sl.fd);  // (the parent's description)
    sl.tried = true;
    sl.pid = me;
    sl.fd = -1;
    if (std::getenv("WFST_SSSP_NO_LOCKFILE")) return -1;
    const char* dir = std::getenv("WFST_LOCK_DIR");
    char path[512];
    std::snprintf(path, sizeof(path), "%s/.wfst_amd_resident_gpu%u.lock", dir ? dir : "/tmp", d);
    int fd = ::open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0 && errno == ENOENT) {
      fd = ::open(path, O_CREAT | O_EXCL | O_RDONLY | O_NOFOLLOW | O_CLOEXEC, 0644);
      if (fd >= 0) (void)::fchmod(fd, 0644);
      else if (errno == EEXIST) fd = ::open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);  // (somebody else created it meanwhile)
    }
    if (fd < 0) {
      static std::atomic<bool> said{false};
      if (!said.exchange(true))
        std::fprintf(stderr, "libwfst_amd: no per-device lock file (%s: %s); resident launches of several processes on one GPU are only "
                             "protected by their wait limit (WFST_LOCK_DIR selects another directory)\n", path, std::strerror(errno));
    }
    sl.fd = fd;
    return fd;
  }
  struct Units { std::mutex mu; int used = 0; int fd = -1; };
  static Units& units_of(int device) {
    static Units u[64];
    return u[(unsigned)device & 63u];
  }
  bool acquire(int device, int units = 2) {
    release();
    Units& u = units_of(device);
    std::lock_guard<std::mutex> lk(u.mu);
    if (u.used + units > 2) return false;
    if (u.used == 0) {
      const int fd = device_lock_fd(device);
      if (fd >= 0 && ::flock(fd, LOCK_EX | LOCK_NB) != 0) return false;  // another process holds the device's resident slot
      u.fd = fd;
    }
    u.used += units;
    held_device = device;
    held_units = units;
    return true;
  }
  void release() {
    if (held_units) {
      Units& u = units_of(held_device);
      std::lock_guard<std::mutex> lk(u.mu);
      u.used -= held_units;
      if (u.used == 0 && u.fd >= 0) {
        (void)::flock(u.fd, LOCK_UN);
        u.fd = -1;
      }
    }
    held_units = 0;
    held_device = -1;
  }
  ~ResidentLease() { release(); }
};

}  // namespace
}  // namespace wfst
bool wfst_ctx::resident_allowed() const {
  if (resident_hold) return false;
  if (resident_retry_at_ns == 0) return true;
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() >= resident_retry_at_ns;
}
void wfst_ctx::resident_aborted() {
  int64_t pause_ms = 50ll << std::min<uint32_t>(resident_abort_streak, 6u);
  if (const char* e = std::getenv("WFST_SSSP_RES_RETRY_MS")) pause_ms = std::max<long long>(0ll, std::atoll(e));  // tests
  resident_abort_streak += 1;
  resident_retry_at_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() +
                         pause_ms * 1000000ll;
  if (resident_retry_at_ns == 0) resident_retry_at_ns = 1;
}
namespace wfst {
namespace {

struct Solve {
  DBuf<uint64_t> key;
  DBuf<uint32_t> shadow;  // enc(d) half of the keys, for the pre-check gathers
  DBuf<uint8_t> flags;  // two frontiers of n bytes
  DBuf<uint32_t> improved;
  DBuf<Ctl> ctl;
  uint32_t sweeps = 0;
  // schedule parameters fixed by relax_setup
  uint8_t* fl[2] = {nullptr, nullptr};
  uint32_t blocks = 1, near_low = 4096;
  float delta = 0.0f;
  uint64_t sweep_cap = 0;
  // per-wave list of near discoveries relaxed inside the same launch, in sweeps that follow a small (< chase_low) one
  uint32_t chase_cap = 32, chase_rounds = 8, chase_low = 4096;  // chase_rounds = states a wave may chase per launch
  // mailbox sweeps (sssp_mailbox.h): owner-computes relaxation, chosen by relax_setup for branching graphs of <= 2^20 states
  bool mbox = false;
  std::shared_ptr<MboxPlan> plan;
  DBuf<uint2> mb_msgs;      // two message buffers of n_arcs entries
  DBuf<uint32_t> mb_words;  // counts (2 x nb^2), wrote (2 x nb), pend masks, blk_pend, blk_mind, blk_far, wl_cnt
  DBuf<uint4> mb_wl;        // work-list segments of the NARROW launches (nb x NW_SEG entries)
  DBuf<unsigned long long> mb_dbg;  // WFST_SSSP_MBOX_TRACE=<file>: per-block phase stamps of the first 64 sweeps
  MboxView mv{};
  uint32_t narrow_t = 0;     // mailbox: near + far-waiting states below which the sweeps hand over to NARROW launches (0 = never)
  uint64_t hint_mask = ~0ull;  // mailbox: bit k = launch k of this FST's last solve was not a busy WIDE sweep (gated launch)
  size_t mb_dyn = 0;         // dynamic LDS bytes of a mailbox launch
  bool force_big = false;    // tests: the many-blocks variant of the kernel on a small FST (WFST_SSSP_BIG=1)
  // resident launches (sssp_resident.h): the WIDE levels of the solve inside ONE launch, every workgroup on a CU of its own
  bool resident = false;
  uint32_t log = 12;         // log2 of the block size (13: every launch of the solve is a resident one)
  size_t res_dyn = 0;        // dynamic LDS bytes of a resident launch
  DBuf<uint2> rs_msgs;       // two parity buffers of plan->res_units entries
  DBuf<uint32_t> rs_abort;   // one word
  DBuf<unsigned long long> rs_trace;  // WFST_SSSP_RES_TRACE=<file>: per-level stamps
  ResView rv{};
  uint32_t res_max_levels = RS_LEVEL_CAP;
  uint32_t res_lps_umax = 8u | (2u << 8);  // lanes per listed state | states per lane group of the widest round (sssp_resident.h)
  ResidentLease lease;       // units of the device's resident lease this solve holds (2 = whole device, 1 = half: two of those run together)
  // binned levels (sssp_binned.h): the dense levels of the atomic sweeps as an owner-computes pass, chosen per level on the device
  bool binned = false;
  std::shared_ptr<BinPlan> bplan;
  DBuf<uint2> bn_msgs;       // one slot per arc
  DBuf<uint32_t> bn_cnt;     // [nbin * G]
  BinView bv{};
  uint32_t dense_low = 0xFFFFFFFFu;  // predicted frontier from which a level is a binned one
  size_t bn_dyn_expand = 0, bn_dyn_apply = 0;
};

constexpr uint32_t MAX_BATCH = 64;

// The relaxation runs to its fixed point in sweeps launched in batches without returning to the host, and the NEXT
// batch is enqueued before the host looks at the previous batch's flags (two batches in flight), so the device never
// idles on a host round trip.  A sweep that changes nothing leaves an empty frontier: everything enqueued behind it is
// a ~3 us no-op.
bool mbox_eligible(const wfst_fst* f) {
  return f->n_states <= (MB_NBMAX_BIG << MB_LOG) && f->n_arcs > 0 && f->n_arcs < 0x7FFFFFFFull && !f->has_negative;
}

// Region plan of the mailbox sweeps: arcs between every pair of blocks, scanned into region offsets.  Depends only on
// the (source, target) pairs of the arcs, so tr_sort leaves it valid; cached on the handle (owner's pool).
std::shared_ptr<MboxPlan> mbox_plan(wfst_ctx* ctx, const wfst_fst* f, uint32_t log) {
  std::lock_guard<std::mutex> lk(f->cache_mu);
  std::shared_ptr<MboxPlan>& slot = log == 13 ? f->mbox13 : f->mbox;
  if (slot) return slot;
  const uint32_t n = f->n_states, nb = (uint32_t)(((uint64_t)n + (1u << log) - 1) >> log);
  hipStream_t st = ctx->stream;
  DevicePool& owner_pool = f->owner_pool ? *f->owner_pool : *ctx->pool;
  auto p = std::make_shared<MboxPlan>();
  p->nb = nb;
  p->log = log;
  const size_t cells = (size_t)nb * nb;
  p->roff = DBuf<uint32_t>(owner_pool, cells + 1);
  p->roff_t = DBuf<uint32_t>(owner_pool, cells);
  DBuf<uint32_t> hist(*ctx->pool, cells + 1);
  if (log == 13) mbox_hist_kernel<13><<<nb, 1024, 0, st>>>(f->dev.offsets, f->dev.wn, n, nb, hist.p);
  else mbox_hist_kernel<12><<<nb, 1024, 0, st>>>(f->dev.offsets, f->dev.wn, n, nb, hist.p);
  HIP_CHECK(hipMemsetAsync(hist.p + cells, 0, sizeof(uint32_t), st));
  size_t temp_bytes = 0;
  HIP_CHECK(rocprim::exclusive_scan(nullptr, temp_bytes, hist.p, p->roff.p, 0u, cells + 1, rocprim::plus<uint32_t>(), st));
  DBuf<uint8_t> temp(*ctx->pool, temp_bytes);
  HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, hist.p, p->roff.p, 0u, cells + 1, rocprim::plus<uint32_t>(), st));
  mbox_transpose_kernel<<<(uint32_t)((cells + 255) / 256), 256, 0, st>>>(p->roff.p, nb, p->roff_t.p);
  // lanes per state of the resident kernel's expansion rounds (a lane takes two arcs): the fewest that cover all but 1/64 of
  // the rows in one pass — a lane group without arcs is a lane group that keeps no row in flight — and leave at most 1/16 of
  // the ARCS to the long-row pass (a few states with hundreds of arcs each, as in a decoding graph, would otherwise be walked 2 l
  // arcs at a time by one lane group while the rest of its wave waits)
  {
    DBuf<uint32_t> over(*ctx->pool, 16);
    DBuf<double> minw(*ctx->pool, 2);
    uint32_t h_over[16];
    double h_minw[2];
    HIP_CHECK(hipMemsetAsync(over.p, 0, sizeof(h_over), st));
    HIP_CHECK(hipMemsetAsync(minw.p, 0, sizeof(h_minw), st));
    mbox_degree_kernel<<<std::min<uint32_t>((n + 255) / 256, 1024u), 256, 0, st>>>(f->dev.offsets, f->dev.wn, n, over.p, minw.p);
    HIP_CHECK(hipMemcpyAsync(h_over, over.p, sizeof(h_over), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(h_minw, minw.p, sizeof(h_minw), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    p->mean_min_w = h_minw[1] > 0.0 ? (float)(h_minw[0] / h_minw[1]) : 0.0f;
    p->lps = 8;
    for (uint32_t l = 2; l < 8; ++l)
      if (h_over[l] <= n / 64u && (uint64_t)h_over[8 + l] <= f->n_arcs / 16u && h_over[8 + l] != 0xFFFFFFFFu) {
        p->lps = l;
        break;
      }
    if (const char* e = std::getenv("WFST_SSSP_LPS")) p->lps = std::max(2, std::min(8, std::atoi(e)));
  }
  uint32_t h_units = 0;
  if (nb <= MB_NBMAX) {  // the resident kernel's regions: header + one slot per arc, 64-byte aligned
    p->roffh = DBuf<uint32_t>(owner_pool, cells + 1);
    p->roffh_t = DBuf<uint32_t>(owner_pool, cells);
    DBuf<uint32_t> sizes(*ctx->pool, cells + 1);
    res_size_kernel<<<(uint32_t)((cells + 1 + 255) / 256), 256, 0, st>>>(hist.p, (uint32_t)cells, sizes.p);
    HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, sizes.p, p->roffh.p, 0u, cells + 1, rocprim::plus<uint32_t>(), st));
    mbox_transpose_kernel<<<(uint32_t)((cells + 255) / 256), 256, 0, st>>>(p->roffh.p, nb, p->roffh_t.p);
    HIP_CHECK(hipMemcpyAsync(&h_units, p->roffh.p + cells, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st));  // sizes is released here
    p->res_units = h_units;
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(st));  // hist / temp are released here
  slot = p;
  return p;
}

// Region plan of the binned levels (sssp_binned.h), cached on the handle like the mailbox plan.
bool bin_eligible(const wfst_fst* f) {
  return f->n_states <= (BN_MAXBINS << 14) && f->n_arcs > 0 && f->n_arcs < 0x7FFFFFFFull && !f->has_negative;
}
std::shared_ptr<BinPlan> bin_plan(wfst_ctx* ctx, const wfst_fst* f, uint32_t logd) {
  std::lock_guard<std::mutex> lk(f->cache_mu);
  if (f->binplan && f->binplan->logd == logd) return f->binplan;
  const uint32_t n = f->n_states;
  hipStream_t st = ctx->stream;
  DevicePool& owner_pool = f->owner_pool ? *f->owner_pool : *ctx->pool;
  auto p = std::make_shared<BinPlan>();
  p->logd = logd;
  p->nbin = (uint32_t)(((uint64_t)n + (1u << logd) - 1) >> logd);
  // two expand workgroups per compute unit; a workgroup's range is a whole number of chunks
  const uint32_t g_max = (uint32_t)std::min<int>(512, std::max<int>(1, 2 * ctx->n_cus));
  p->sg = (uint32_t)(((((uint64_t)n + g_max - 1) / g_max) + BN_CH - 1) / BN_CH * BN_CH);
  p->G = (uint32_t)(((uint64_t)n + p->sg - 1) / p->sg);
  const size_t cells = (size_t)p->nbin * p->G;
  p->roff = DBuf<uint32_t>(owner_pool, cells + 1);
  p->roff_t = DBuf<uint32_t>(owner_pool, cells);
  DBuf<uint32_t> hist(*ctx->pool, cells + 1);
  if (logd == 14) bin_hist_kernel<14><<<p->G, 1024, 0, st>>>(f->dev.offsets, f->dev.wn, n, p->sg, p->nbin, p->G, hist.p);
  else bin_hist_kernel<13><<<p->G, 1024, 0, st>>>(f->dev.offsets, f->dev.wn, n, p->sg, p->nbin, p->G, hist.p);
  HIP_CHECK(hipMemsetAsync(hist.p + cells, 0, sizeof(uint32_t), st));
  size_t temp_bytes = 0;
  HIP_CHECK(rocprim::exclusive_scan(nullptr, temp_bytes, hist.p, p->roff.p, 0u, cells + 1, rocprim::plus<uint32_t>(), st));
  DBuf<uint8_t> temp(*ctx->pool, temp_bytes);
  HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, hist.p, p->roff.p, 0u, cells + 1, rocprim::plus<uint32_t>(), st));
  bin_transpose_kernel<<<(uint32_t)((cells + 255) / 256), 256, 0, st>>>(p->roff.p, p->nbin, p->G, p->roff_t.p);
  uint32_t h_slots = 0;
  HIP_CHECK(hipMemcpyAsync(&h_slots, p->roff.p + cells, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(st));  // hist / temp are released here
  p->slots = h_slots;
  f->binplan = p;
  return p;
}

// what the next solve of this FST is sized from: the launches this one needed, and how long that count has been the same
void note_sweeps(const wfst_fst* f, uint32_t sweeps) {
  const uint32_t prev = f->last_sweeps.exchange(sweeps, std::memory_order_relaxed);
  if (prev == sweeps) f->stable_sweeps.fetch_add(1, std::memory_order_relaxed);
  else f->stable_sweeps.store(0, std::memory_order_relaxed);
}

// one relaxation sweep on the stream: `j` = position inside the batch (static flag / message parity), `off` = sweep
// index relative to the device-side base, `abs_sweep` = the absolute index (what the host has queued so far)
void launch_sweep(const wfst_fst* f, Solve& sv, uint32_t n, hipStream_t st, uint32_t j, uint32_t off, uint32_t abs_sweep,
                  uint32_t profile) {
  if (sv.mbox) {
    // (message parity from the ABSOLUTE sweep index: mailbox batches may hold an odd number of sweeps)
    // hint = the launch is probably NOT a busy WIDE sweep (what the last solve of this FST did in that slot, or unknown):
    // it then finds out its mode and whether it sleeps BEFORE it asks for its 48 KB of keys and offsets
    const uint32_t hint = profile || abs_sweep >= 64 ? 1u : (uint32_t)((sv.hint_mask >> abs_sweep) & 1ull);
    // resident launches take the odd slots: slot 0 is the head of the search (NARROW), a resident launch runs every WIDE
    // level that follows and the hand-over, the next slot is the NARROW launch that drains the search; whichever kernel
    // finds another mode in its slot does that mode's work (the two kernels leave the same state behind)
    if (sv.log == 13) {  // (blocks of 8192 states exist in the resident kernel only: it also runs the NARROW launches of such a solve)
      if (abs_sweep >= RS_MAX_SWEEP) throw Error("shortest_path: relaxation did not converge");
      sssp_mbox_resident_kernel<13><<<sv.mv.nb, MB_THREADS, sv.res_dyn, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, sv.mv, sv.rv, abs_sweep & 1u, n,
                                                                             sv.improved.p, sv.ctl.p, abs_sweep, sv.delta, sv.near_low,
                                                                             sv.narrow_t, sv.res_max_levels, sv.res_lps_umax);
    } else if (sv.resident && !profile && (abs_sweep & 1u) && abs_sweep < RS_MAX_SWEEP)
      sssp_mbox_resident_kernel<12><<<sv.mv.nb, MB_THREADS, sv.res_dyn, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, sv.mv, sv.rv, abs_sweep & 1u, n,
                                                                             sv.improved.p, sv.ctl.p, abs_sweep, sv.delta, sv.near_low,
                                                                             sv.narrow_t, sv.res_max_levels, sv.res_lps_umax);
    else if (sv.mv.nb > MB_NBMAX || sv.force_big)
      sssp_mbox_kernel<true><<<sv.mv.nb, MB_THREADS, sv.mb_dyn, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, sv.mv, abs_sweep & 1u, n,
                                                                      sv.improved.p, sv.ctl.p, abs_sweep, sv.delta, sv.near_low,
                                                                      profile, hint, sv.narrow_t);
    else
      sssp_mbox_kernel<false><<<sv.mv.nb, MB_THREADS, sv.mb_dyn, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, sv.mv, abs_sweep & 1u, n,
                                                                       sv.improved.p, sv.ctl.p, abs_sweep, sv.delta, sv.near_low,
                                                                       profile, hint, sv.narrow_t);
  }
  else {
    sssp_relax_kernel<<<sv.blocks, 256, 0, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, sv.fl[j & 1u], sv.fl[(j & 1u) ^ 1u], n,
                                                 sv.improved.p, sv.ctl.p, off, sv.delta, sv.near_low, sv.shadow.p, sv.chase_cap,
                                                 sv.chase_rounds, sv.chase_low, profile, sv.binned ? sv.dense_low : 0xFFFFFFFFu);
    if (sv.binned) {  // the same level as an owner-computes pass: both leave at once unless the launch above published a dense level
      if (sv.bplan->logd == 14) {
        sssp_bin_expand_kernel<14><<<sv.bv.G, BN_THREADS, sv.bn_dyn_expand, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, sv.fl[j & 1u],
                                                                                  sv.fl[(j & 1u) ^ 1u], n, sv.improved.p, sv.ctl.p, off,
                                                                                  sv.shadow.p, sv.bv, profile);
        sssp_bin_apply_kernel<14><<<sv.bv.nbin, BN_THREADS, sv.bn_dyn_apply, st>>>(sv.key.p, sv.shadow.p, sv.fl[(j & 1u) ^ 1u], n,
                                                                                   sv.improved.p, sv.ctl.p, off, sv.bv);
      } else {
        sssp_bin_expand_kernel<13><<<sv.bv.G, BN_THREADS, sv.bn_dyn_expand, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, sv.fl[j & 1u],
                                                                                  sv.fl[(j & 1u) ^ 1u], n, sv.improved.p, sv.ctl.p, off,
                                                                                  sv.shadow.p, sv.bv, profile);
        sssp_bin_apply_kernel<13><<<sv.bv.nbin, BN_THREADS, sv.bn_dyn_apply, st>>>(sv.key.p, sv.shadow.p, sv.fl[(j & 1u) ^ 1u], n,
                                                                                   sv.improved.p, sv.ctl.p, off, sv.bv);
      }
    }
  }
}

// relax_setup allocates and initialises the state of a solve (keys, frontier flags, control block) and fixes its schedule parameters
void relax_setup(wfst_ctx* ctx, const wfst_fst* f, Solve& sv) {
  const uint32_t n = f->n_states;
  DevicePool& pool = *ctx->pool;
  sv.key = DBuf<uint64_t>(pool, n);
  sv.shadow = DBuf<uint32_t>(pool, n);
  const size_t n_pad = ((size_t)n + 15) & ~(size_t)15;
  sv.flags = DBuf<uint8_t>(pool, 2 * n_pad);
  sv.improved = DBuf<uint32_t>(pool, IMP_RING);
  sv.ctl = DBuf<Ctl>(pool, 1);
  hipStream_t st = ctx->stream;
  sv.fl[0] = sv.flags.p;
  sv.fl[1] = sv.flags.p + n_pad;
  sv.blocks = std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)ctx->n_cus * 8, (n + 255) / 256));
  // near-far only pays on branching graphs (label-correcting re-relaxes them many times); on lattices every
  // arc is relaxed once anyway.  delta = 1.5 x mean arc weight; +inf = plain frontier sweeps.
  float delta = INF;
  if (!f->has_negative && f->mean_weight > 0.0f && n >= 65536 && f->n_arcs >= 2ull * n) {
    delta = 1.5f * f->mean_weight;
    // ... of a graph with ~10 arcs per state.  With more, a state's BEST arc is cheaper (the minimum of d weights) while the
    // mean is not, the distances shrink and a band of 1.5 mean weights swallows the search: everything is near, everything is
    // relaxed again and again; with fewer, the band holds a few dozen states per level for many levels.  The band follows
    // 10 / (arcs per state) — 1M states (tools/fan_sweep.py): fan-out 3 / 4 / 5 / 6 / 8: 247 / 254 / 223 / 236 / 229 -> 189 /
    // 193 / 191 / 200 / 223 us; 12 / 16 / 24: 366 / 638 / 1 262 -> 275 / 327 / 523 — except between 8.5 and 10 arcs per state,
    // where the schedule was swept on the benchmark's transducer and the unscaled band is the better one (fan-out 9: 233 against 256).
    const double deg = (double)f->n_arcs / (double)n;
    if (deg > 10.0 || deg < 8.5) delta = (float)(delta * 10.0 / deg);
  }
  if (const char* e = std::getenv("WFST_SSSP_DELTA")) delta = (float)std::atof(e);  // experiments / tests
  if (!(delta > 0.0f)) delta = INF;
  sv.delta = delta;
  sv.near_low = 4096;  // activations below which a sweep is launch-latency bound anyway
  if (const char* e = std::getenv("WFST_SSSP_NEAR_LOW")) sv.near_low = (uint32_t)std::atol(e);
  float tau0_mult = 1.0f;  // first band = tau0_mult x delta
  if (const char* e = std::getenv("WFST_SSSP_TAU0_MULT")) tau0_mult = (float)std::atof(e);
  static_assert(sizeof(Ctl) % 4 == 0, "Ctl is cleared word by word");
  // mailbox sweeps where they pay: branching graphs (the near-far case) whose state ids fit the message format
  // ... and few enough blocks that the per-pair regions still hold runs of messages: measured on MI355X (fan-out 10), the
  // mailbox launches win at 1M and 2M states (0.41 vs 0.50 ms, 0.84 vs 0.89), tie at 3M and lose at 5M (3.2 vs 2.4 ms: a
  // region then receives ~3 messages per sweep, and 1221 workgroups take five turns on 256 CUs).  Beyond MB_NB_DEFAULT
  // blocks the atomic sweeps stay the default; WFST_SSSP_MAILBOX=1 still selects the mailbox kernel up to 8M states.
  bool want_mbox = delta < INF && mbox_eligible(f) && ((n + MB_B - 1) >> MB_LOG) <= MB_NB_DEFAULT;
  int mbox_mode = want_mbox ? 1 : 0;
  if (const char* e = std::getenv("WFST_SSSP_MAILBOX")) mbox_mode = mbox_eligible(f) ? std::atoi(e) : 0;
  // Resident launches (sssp_resident.h) need every block on a CU of its own for the whole launch, and only one such solve
  // runs per device at a time: the lease is taken first, because it decides the block size — 4096 states, or 8192 for the
  // FSTs of 1M .. 2M states that then still fit one block per CU (every launch of such a solve is a resident one).
  bool want_res = false;
  sv.log = 12;
  sv.lease.release();
  if (mbox_mode >= 1) {
    int want = 1;
    if (const char* e = std::getenv("WFST_SSSP_RESIDENT")) want = std::atoi(e);
    const bool big_env = std::getenv("WFST_SSSP_BIG") && std::atoi(std::getenv("WFST_SSSP_BIG")) != 0;
    const uint32_t cus = (uint32_t)std::min<int>(ctx->n_cus, (int)MB_NBMAX);
    const uint32_t nb12 = (n + 4095u) >> 12, nb13 = (uint32_t)(((uint64_t)n + 8191u) >> 13);
    uint32_t log = 0;
    if (nb12 <= cus) log = 12;
    else if (nb13 <= cus && !std::getenv("WFST_SSSP_LOG12")) log = 13;
    if (log == 12 && nb13 <= cus && std::getenv("WFST_SSSP_LOG13") && std::atoi(std::getenv("WFST_SSSP_LOG13")) != 0) log = 13;  // experiments: half the workgroups
    // wfst_ctx_set_resident_share(ctx, 1): the resident grid may hold at most half of the compute units, so that a large batch
    // kernel of another context runs BESIDE it (a resident workgroup needs a compute unit of its own; a grid that wants 245 of
    // 256 waits for whatever holds more than 11 of them).  8192-state blocks where that is enough, one launch per level otherwise.
    if (ctx->resident_share == 1u) {
      const uint32_t half = cus / 2u;
      if (nb12 <= half) log = 12;
      else if (nb13 <= half && !std::getenv("WFST_SSSP_LOG12")) log = 13;
      else log = 0;
    }
    if (want && log && !ctx->profiling && ctx->resident_allowed() && !big_env &&
        sv.lease.acquire(ctx->device, ctx->resident_share == 1u ? 1 : 2)) {
      want_res = true;
      sv.log = log;
    }
  }
  if (mbox_mode >= 1) {
    // Two message buffers of one slot per arc, the nb^2 region tables and counts come from the pool: on a tight pool (or a
    // dense graph) the atomic sweeps, which need none of it, run instead.  The 8192-state plan exists in the resident kernel
    // only: when it cannot be had (the pool, a region buffer beyond the buffer-descriptor range, no room for staging), the
    // solve is planned again with 4096-state blocks and one launch per level — never refused.
    auto release_all = [&] {
      want_res = false;
      sv.rs_msgs.reset();
      sv.rs_abort.reset();
      sv.lease.release();
      sv.plan.reset();
      sv.mb_msgs.reset();
      sv.mb_words.reset();
      sv.mb_wl.reset();
      sv.mbox = false;
    };
    auto try_plan = [&](uint32_t log, bool res) -> bool {
      try {
        sv.log = log;
        sv.plan = mbox_plan(ctx, f, log);
        const uint32_t nb = sv.plan->nb;
        if (res) {  // what the resident launch needs beyond the plan
          const uint64_t bytes = sv.plan->res_units * sizeof(uint2);
          const size_t fixed = res_lds_bytes(log, nb, 0);
          const bool fits = sv.plan->res_units != 0 && bytes < 0x7FFFFFF0ull && fixed + 8u * 4u * nb <= 160u * 1024u;
          if (!fits) {
            if (log == 13) throw Error("resident launch not possible");
            res = false;
            sv.lease.release();
          }
        }
        sv.mb_msgs = DBuf<uint2>(pool, 2 * (size_t)f->n_arcs);
        const size_t w_cnt = (size_t)nb * nb, w_pend = (size_t)nb * ((1u << log) / 32);
        sv.mb_words = DBuf<uint32_t>(pool, 2 * w_cnt + 2 * nb + w_pend + 4 * nb);
        sv.mb_wl = DBuf<uint4>(pool, (size_t)nb << log);
        if (res) {
          sv.rs_msgs = DBuf<uint2>(pool, 2 * (size_t)sv.plan->res_units);
          sv.rs_abort = DBuf<uint32_t>(pool, 16);
        }
        want_res = res;
        sv.mbox = true;
        return true;
      } catch (const Error&) {
        (void)hipGetLastError();  // (a refused allocation must not surface at the next launch check)
        release_all();
        return false;
      }
    };
    if (std::getenv("WFST_SSSP_TEST_FAIL_LOG13") && sv.log == 13) {  // tests: the 8192-state plan refused
      release_all();
      sv.log = 12;
      try_plan(12, false);
    } else if (!try_plan(sv.log, want_res) && sv.log == 13) {
      try_plan(12, false);
    }
    if (!sv.mbox) sv.log = 12;
  }
  if (sv.mbox) {
    // The band once more, now that the plan knows the states' CHEAPEST arcs: the rule above reads the mean weight as if the
    // weights were uniform on [0, 2 mean) — a state's best of d arcs then costs 2 mean / (d + 1).  Where the cheapest arcs are
    // much cheaper than that (exponential, log-normal weights: a shortest path is made of them) the band shrinks by the same
    // ratio; where every arc costs about the same (unit weights: the search is a breadth-first one) it widens a little.
    // 1M states, fan-out 10 (tools/weight_shapes.py): exponential 410 -> 241 us, unit weights 205 -> ~125; uniform weights
    // (ratio 0.9 .. 1.1): untouched.
    if (sv.delta < INF && !std::getenv("WFST_SSSP_DELTA") && sv.plan->mean_min_w > 0.0f && f->mean_weight > 0.0f) {
      const double deg = (double)f->n_arcs / (double)n;
      const double ratio = (double)sv.plan->mean_min_w * (deg + 1.0) / (2.0 * (double)f->mean_weight);
      if (ratio < 0.7) delta = (float)(delta * std::max(ratio, 0.1));
      else if (ratio > 2.0) delta = delta * 1.4f;
      sv.delta = delta;
    }
    const uint32_t nb = sv.plan->nb;
    const size_t w_cnt = (size_t)nb * nb, w_pend = (size_t)nb * ((1u << sv.log) / 32);
    MboxView& mv = sv.mv;
    mv.roff = sv.plan->roff.p;
    mv.roff_t = sv.plan->roff_t.p;
    mv.msgs[0] = sv.mb_msgs.p;
    mv.msgs[1] = sv.mb_msgs.p + f->n_arcs;
    uint32_t* w = sv.mb_words.p;
    mv.cnt[0] = w;
    mv.cnt[1] = w + w_cnt;
    w += 2 * w_cnt;
    mv.wrote[0] = w;
    mv.wrote[1] = w + nb;
    w += 2 * nb;
    mv.pend = w;
    w += w_pend;
    mv.blk_pend = w;
    mv.blk_mind = w + nb;
    mv.blk_far = w + 2 * nb;
    mv.wl_cnt = w + 3 * nb;
    mv.wl = sv.mb_wl.p;
    mv.nb = nb;
    // staging depth: what the dynamic LDS budget leaves after the three per-destination tables
    mv.stg = std::max<uint32_t>(1u, std::min<uint32_t>(MB_STG_MAX, (MB_DYN_BUDGET - 12u * nb) / (8u * nb)));
    if (const char* e = std::getenv("WFST_SSSP_STG")) mv.stg = std::max<uint32_t>(1u, std::min<uint32_t>(mv.stg, (uint32_t)std::atoi(e)));
    if (const char* e = std::getenv("WFST_SSSP_BIG")) sv.force_big = std::atoi(e) != 0;
    sv.mb_dyn = (size_t)nb * mv.stg * sizeof(uint2) + 3u * (size_t)nb * sizeof(uint32_t);
    static std::once_flag lds_once[64];  // (a function attribute is per device)
    std::call_once(lds_once[(unsigned)ctx->device & 63u], [] {
      HIP_CHECK(hipFuncSetAttribute((const void*)sssp_mbox_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MB_DYN_BUDGET));
      HIP_CHECK(hipFuncSetAttribute((const void*)sssp_mbox_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MB_DYN_BUDGET));
    });
    // A mailbox sweep costs its ~10 us chain of dependent trips whatever it expands, up to ~256 states per block: the
    // threshold moves on as soon as the near set is below 1/16 of the states (measured, tools/param_sweep.py: 4096 ->
    // 65536 at 1M / 2M states is 0.407 -> 0.386 and 0.817 -> 0.741 ms; at 300 k states 16384 is the best)
    if (!std::getenv("WFST_SSSP_NEAR_LOW")) sv.near_low = std::min<uint32_t>(65536u, std::max<uint32_t>(4096u, n / 16u));
    // hand-over to NARROW launches when the near set plus everything waiting beyond the threshold is this small
    sv.narrow_t = 8192;
    if (const char* e = std::getenv("WFST_SSSP_NARROW")) sv.narrow_t = (uint32_t)std::atol(e);
    sv.hint_mask = f->last_hint_mask.load(std::memory_order_relaxed);
    if (const char* e = std::getenv("WFST_SSSP_HINT")) sv.hint_mask = std::atoi(e) ? ~0ull : 0ull;  // experiments: all / no launches gated
    mv.dbg = nullptr;
    if (std::getenv("WFST_SSSP_MBOX_TRACE")) {
      sv.mb_dbg = DBuf<unsigned long long>(pool, (size_t)MB_DBG_SWEEPS * nb * 16);
      HIP_CHECK(hipMemsetAsync(sv.mb_dbg.p, 0, (size_t)MB_DBG_SWEEPS * nb * 16 * 8, st));
      mv.dbg = sv.mb_dbg.p;
    }
    // Resident launches: every block needs a CU of its own for the whole launch (1024 threads x 128 registers and ~150 KB of
    // LDS fill one), so the grid must fit the device, and only one such solve runs per device at a time.  Not under the
    // per-sweep profiler (it times launches), not after a launch of this context gave up waiting.
    sv.resident = false;
    sv.rv = ResView{};
    sv.res_dyn = 0;
    {
      const uint64_t bytes = sv.plan->res_units * sizeof(uint2);
      constexpr size_t LDS_BYTES = 160u * 1024u;
      const size_t fixed = res_lds_bytes(sv.log, nb, 0);
      // (deeper than the one-level kernel's slots where the LDS has room: a round of the resident kernel holds up to 4 x 16 x
      // (64 / lps) states, 768 with 5 lanes per state)
      const uint32_t stg_lim = std::min<uint32_t>(RS_STG_MAX, (MB_DYN_BUDGET - 12u * nb) / (8u * nb));  // (mb_dyn below: the one-level kernel's budget)
      uint32_t stg_res = fixed + 8u * nb <= LDS_BYTES ? (uint32_t)std::min<size_t>(stg_lim, (LDS_BYTES - fixed) / (8u * nb)) : 0u;
      if (const char* e = std::getenv("WFST_SSSP_STG")) stg_res = std::max<uint32_t>(1u, std::min<uint32_t>(stg_res, (uint32_t)std::atoi(e)));
      if (want_res && !sv.force_big && sv.plan->res_units != 0 && bytes < 0x7FFFFFF0ull && stg_res >= 4 && sv.rs_msgs.p) {
        sv.resident = true;
        mv.stg = stg_res;
        {
          // states per lane group in the widest round: 4, unless such a round (4 x 16 x (64 / lps) states) would send more than
          // ~0.8 of the staging slots' worth to an average destination — what overflows leaves as 8-byte stores of its own
          // (2M states, 8192-state blocks, 21 slots: 454 us with 4, 432 with 2)
          const uint32_t lps = sv.plan->lps, g = 16u * (64u / lps);
          const double per_dest = 4.0 * g * ((double)f->n_arcs / (double)n) / (double)nb;
          uint32_t umax = per_dest <= 0.8 * mv.stg ? 4u : 2u;
          if (const char* e = std::getenv("WFST_SSSP_UMAX")) umax = std::atoi(e) >= 4 ? 4u : 2u;
          sv.res_lps_umax = lps | (umax << 8);
        }
        sv.mb_dyn = (size_t)nb * mv.stg * sizeof(uint2) + 3u * (size_t)nb * sizeof(uint32_t);
        sv.res_dyn = res_lds_bytes(sv.log, nb, mv.stg);
      } else {
        if (sv.log == 13) throw Error("shortest_path: internal error (the 8192-state plan without a resident launch)");
        sv.rs_msgs.reset();
        sv.rs_abort.reset();
        sv.lease.release();
      }
      if (sv.resident) {
        ResView& rv = sv.rv;
        rv.msgs[0] = sv.rs_msgs.p;
        rv.msgs[1] = sv.rs_msgs.p + sv.plan->res_units;
        rv.roffh = sv.plan->roffh.p;
        rv.roffh_t = sv.plan->roffh_t.p;
        rv.abort = sv.rs_abort.p;
        rv.bytes = (uint32_t)bytes;
        rv.tlim_ticks = 2000000u;  // 20 ms of wall_clock64
        if (const char* e = std::getenv("WFST_SSSP_RES_TLIM_US")) rv.tlim_ticks = (uint32_t)std::min<long long>(4000000000ll, std::atoll(e) * 100ll);
        sv.res_max_levels = RS_LEVEL_CAP;
        if (const char* e = std::getenv("WFST_SSSP_RES_LEVELS")) sv.res_max_levels = std::max<uint32_t>(2u, std::min<uint32_t>(RS_LEVEL_CAP, (uint32_t)std::atol(e)));
        rv.trace = nullptr;
        if (std::getenv("WFST_SSSP_RES_TRACE")) {
          sv.rs_trace = DBuf<unsigned long long>(pool, (size_t)RS_TRACE_LEVELS * nb * 4);
          HIP_CHECK(hipMemsetAsync(sv.rs_trace.p, 0, (size_t)RS_TRACE_LEVELS * nb * 4 * 8, st));
          rv.trace = sv.rs_trace.p;
        }
        // a resident level costs ~7 us when thin (a launch per level: ~10): the band's tail is cut a little later and the
        // hand-over to the NARROW launch comes a little earlier (measured on C3: 298.8 -> 289.5 us per solve)
        if (!std::getenv("WFST_SSSP_NEAR_LOW")) sv.near_low = std::min<uint32_t>(65536u, std::max<uint32_t>(4096u, n / 32u));
        // (what the NARROW launch can hold is ~48 - 96 entries per workgroup: handing it more per workgroup than that sends the
        // search back and forth between the modes — 150 k states, 37 blocks, threshold 16 384: 7.6 launches and 0.55 ms per
        // query from random sources; with 64 per block 3 launches and 0.40 ms: tools/varied_sources.py)
        if (!std::getenv("WFST_SSSP_NARROW")) sv.narrow_t = std::min<uint32_t>(16384u, std::max<uint32_t>(1024u, nb * (sv.log == 13 ? 128u : 64u)));
        static std::once_flag res_once[64];
        std::call_once(res_once[(unsigned)ctx->device & 63u], [] {
          HIP_CHECK(hipFuncSetAttribute((const void*)sssp_mbox_resident_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
          HIP_CHECK(hipFuncSetAttribute((const void*)sssp_mbox_resident_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        });
      }
    }
    if (sv.log == 13)
      sssp_mbox_setup_kernel<13><<<sv.blocks, 256, 0, st>>>(sv.key.p, mv, sv.improved.p, sv.ctl.p, f->dev.offsets, n, (uint32_t)f->start,
                                                            delta * (tau0_mult > 0.0f ? tau0_mult : 1.0f), sv.narrow_t, sv.rv.msgs[0],
                                                            sv.rv.msgs[1], sv.rv.roffh, sv.rv.abort);
    else
      sssp_mbox_setup_kernel<12><<<sv.blocks, 256, 0, st>>>(sv.key.p, mv, sv.improved.p, sv.ctl.p, f->dev.offsets, n, (uint32_t)f->start,
                                                            delta * (tau0_mult > 0.0f ? tau0_mult : 1.0f), sv.narrow_t, sv.rv.msgs[0],
                                                            sv.rv.msgs[1], sv.rv.roffh, sv.rv.abort);
  } else {
    // Binned levels (sssp_binned.h): the dense levels of the solve as owner-computes passes, chosen per level on the device.
    // Parity-green and measured (profiles/r05*): NOT faster than the atomic sweeps on MI355X at any size tried (5M states:
    // 2.9 vs 2.4 ms; a binned level costs ~50 us + 25 ps per arc against 14.5 ps per arc + 78 ps per atomic), so it is an
    // option, not the default.  WFST_SSSP_BINNED=1: on wherever the message format allows.
    sv.binned = false;
    sv.bplan.reset();
    int bin_mode = 0;
    if (const char* e = std::getenv("WFST_SSSP_BINNED")) bin_mode = bin_eligible(f) && delta < INF ? std::atoi(e) : 0;
    if (bin_mode >= 1 && !(ctx->n_cus > 0)) bin_mode = 0;
    if (bin_mode >= 1) {
      uint32_t logd = n <= (BN_MAXBINS << 13) ? 13u : 14u;
      if (const char* e = std::getenv("WFST_SSSP_BIN_LOG")) logd = std::atoi(e) == 14 ? 14u : logd;
      try {  // (a slot per arc and the region tables come from the pool: on a tight pool the atomic sweeps run every level)
        sv.bplan = bin_plan(ctx, f, logd);
        const BinPlan& bp = *sv.bplan;
        if (bp.slots == 0 || bp.slots >= 0xFFFF0000ull) throw Error("binned levels: region table out of range");
        sv.bn_msgs = DBuf<uint2>(pool, (size_t)bp.slots);
        sv.bn_cnt = DBuf<uint32_t>(pool, (size_t)bp.nbin * bp.G);
        HIP_CHECK(hipMemsetAsync(sv.bn_cnt.p, 0, (size_t)bp.nbin * bp.G * sizeof(uint32_t), st));
        BinView& bv = sv.bv;
        bv.roff = bp.roff.p;
        bv.roff_t = bp.roff_t.p;
        bv.msgs = sv.bn_msgs.p;
        bv.cnt = sv.bn_cnt.p;
        bv.nbin = bp.nbin;
        bv.G = bp.G;
        bv.sg = bp.sg;
        bv.hop_cap = 1u << (32u - logd);
        if (const char* e = std::getenv("WFST_SSSP_BIN_HOPCAP")) bv.hop_cap = std::max<uint32_t>(1u, std::min<uint32_t>(bv.hop_cap, (uint32_t)std::atol(e)));
        sv.bn_dyn_expand = bin_expand_lds(bp.nbin);
        sv.bn_dyn_apply = bin_apply_lds(logd);
        static std::once_flag bn_once[64];  // (a function attribute is per device)
        std::call_once(bn_once[(unsigned)ctx->device & 63u], [] {
          HIP_CHECK(hipFuncSetAttribute((const void*)sssp_bin_expand_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, BN_LDS_MAX));
          HIP_CHECK(hipFuncSetAttribute((const void*)sssp_bin_expand_kernel<14>, hipFuncAttributeMaxDynamicSharedMemorySize, BN_LDS_MAX));
          HIP_CHECK(hipFuncSetAttribute((const void*)sssp_bin_apply_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, BN_LDS_MAX));
          HIP_CHECK(hipFuncSetAttribute((const void*)sssp_bin_apply_kernel<14>, hipFuncAttributeMaxDynamicSharedMemorySize, BN_LDS_MAX));
        });
        // a binned level costs two launches and the keys of every bin that receives anything (12 B per state when all do):
        // it pays from a frontier of ~1/64 of the states (measured, profiles/r05*: the atomic sweep relaxes ~10 G arcs/s there)
        sv.dense_low = std::max<uint32_t>(16384u, n / 64u);
        if (const char* e = std::getenv("WFST_SSSP_DENSE_LOW")) sv.dense_low = (uint32_t)std::min<long long>(0xFFFFFFFEll, std::atoll(e));
        sv.binned = true;
      } catch (const Error&) {
        (void)hipGetLastError();  // (a refused allocation must not surface at the next launch check)
        sv.binned = false;
      }
      if (!sv.binned) {
        sv.bplan.reset();
        sv.bn_msgs.reset();
        sv.bn_cnt.reset();
      }
    }
    sssp_setup_kernel<<<sv.blocks, 256, 0, st>>>(sv.key.p, sv.shadow.p, (uint32_t*)sv.flags.p, (uint32_t)(2 * n_pad / 4),
                                                 sv.improved.p, sv.ctl.p, n, (uint32_t)f->start,
                                                 delta * (tau0_mult > 0.0f ? tau0_mult : 1.0f));
  }
  HIP_CHECK(hipGetLastError());
  ctx->stats.relax_kernel = sv.mbox ? (sv.resident ? 2u : 1u) : (sv.binned ? 3u : 0u);
  sv.sweep_cap = 4ull * n + 64;
  if (const char* e = std::getenv("WFST_SSSP_CHASE_CAP")) sv.chase_cap = std::min<uint32_t>((uint32_t)std::atol(e), CHASE_MAX);
  if (const char* e = std::getenv("WFST_SSSP_CHASE_ROUNDS")) sv.chase_rounds = (uint32_t)std::atol(e);
  if (const char* e = std::getenv("WFST_SSSP_CHASE_LOW")) sv.chase_low = (uint32_t)std::atol(e);
}

// Queues the sweeps of a solve in batches and finds the sweep that changed nothing.  start() queues the first batch and
// returns; finish() waits for it and continues until convergence (see the launch schedule in DESIGN.md §3.2).
struct SweepBatch {
  uint32_t first, count;
  int which;
};
struct SweepDriver {
  wfst_ctx* ctx = nullptr;
  const wfst_fst* f = nullptr;
  Solve* sv = nullptr;
  uint32_t n = 0;
  uint32_t* h_imp = nullptr;
  bool use_graphs = false;
  uint32_t next_sweep = 0, first_count = 8, sweeps_done = 0;
  bool predicted = false;  // the first batch is expected to cover the whole solve: nothing is queued behind it
  bool extended = false;   // more than the first batch was needed
  hipEvent_t evs[2] = {nullptr, nullptr};
  SweepBatch cur{};

  void init(wfst_ctx* c, const wfst_fst* fst, Solve* s) {
    ctx = c;
    f = fst;
    sv = s;
    n = fst->n_states;
    h_imp = (uint32_t*)ctx->pinned_flags.get(3 * IMP_RING * sizeof(uint32_t));
    if (const char* e = std::getenv("WFST_SSSP_GRAPH")) use_graphs = std::atoi(e) != 0;
    // A batch boundary costs ~14 us of idle GPU (profiles/r01d), so the FIRST batch of a solve is sized to what the
    // previous solve of this FST needed (+1 sweep to see the quiet one, rounded up to an even count; batch sizes stay
    // even because the flag parity of a sweep inside a batch is static).
    // The mailbox sweeps repeat themselves closely but not exactly (a state expanded while other waves of the same launch
    // improve it sends its old or its new key; NARROW launches follow discoveries in whatever order the atomics land), so a
    // repeated query gets the launches the last one needed, the quiet one included, plus ONE: a gated idle launch is ~3 us,
    // a second batch is a host round trip and a re-run of the fused tail.
    // (the margin is dropped once three solves in a row needed the same count: wfst_fst::stable_sweeps)
    const uint32_t last_sweeps = f->last_sweeps.load(std::memory_order_relaxed);
    // (wfst_fst_set_start resets the count of stable solves: a query from another source than the last one gets the spare launch back)
    const uint32_t margin = f->stable_sweeps.load(std::memory_order_relaxed) >= 3 ? 0u : 1u;
    if (last_sweeps) first_count = std::min<uint32_t>(MAX_BATCH, sv->mbox ? last_sweeps + margin : ((last_sweeps + 1 + 1) & ~1u));
    predicted = last_sweeps != 0 && (sv->mbox ? last_sweeps <= first_count : last_sweeps < first_count);
    evs[0] = ctx->ev0;
    evs[1] = ctx->ev1;
  }

  // One HIP graph = one batch (WFST_SSSP_GRAPH=1): `count` sweep kernels (static offsets from the device-side base) and the
  // advance kernel, chained, built with explicit nodes — stream capture would make every other thread's
  // hipStreamSynchronize fail while it is active.
  hipGraphExec_t get_graph(int which, uint32_t count) {
    wfst_ctx::SweepGraph& g = ctx->sweep_graph[which];
    const uint32_t blocks = sv->blocks, near_low = sv->near_low;
    const float delta = sv->delta;
    const uint64_t key[8] = {(uint64_t)f->dev.offsets, (uint64_t)f->dev.wn ^ ((uint64_t)count << 56), (uint64_t)sv->key.p,
                             (uint64_t)sv->flags.p,
                             (uint64_t)sv->improved.p ^ ((uint64_t)sv->shadow.p << 1), (uint64_t)sv->ctl.p,
                             ((uint64_t)n << 32) | __float_as_uint_host(delta),
                             (uint64_t)(h_imp + which * IMP_RING) ^ ((uint64_t)near_low << 48) ^ ((uint64_t)sv->chase_cap << 40) ^
                                 ((uint64_t)sv->chase_rounds << 20) ^ ((uint64_t)sv->chase_low << 4)};
    if (g.exec && std::memcmp(g.key, key, sizeof(key)) == 0) return g.exec;
    if (g.exec) HIP_CHECK(hipGraphExecDestroy(g.exec));
    if (g.graph) HIP_CHECK(hipGraphDestroy(g.graph));
    g.exec = nullptr;
    g.graph = nullptr;
    HIP_CHECK(hipGraphCreate(&g.graph, 0));
    hipGraphNode_t prev = nullptr;
    const uint32_t* a_offsets = f->dev.offsets;
    const uint2* a_wn = f->dev.wn;
    uint64_t* a_key = sv->key.p;
    uint32_t a_n = n;
    uint32_t* a_imp = sv->improved.p;
    Ctl* a_ctl = sv->ctl.p;
    float a_delta = delta;
    uint32_t a_low = near_low;
    uint32_t* a_shadow = sv->shadow.p;
    uint32_t a_cap = sv->chase_cap, a_rounds = sv->chase_rounds, a_clow = sv->chase_low, a_profile = 0, a_dense = 0xFFFFFFFFu;
    for (uint32_t j = 0; j < count; ++j) {
      uint8_t* a_fc = sv->fl[j & 1u];
      uint8_t* a_fn = sv->fl[(j & 1u) ^ 1u];
      uint32_t a_off = j;
      void* args[] = {&a_offsets, &a_wn,  &a_key,   &a_fc,  &a_fn,     &a_n,   &a_imp,    &a_ctl,
                      &a_off,     &a_delta, &a_low, &a_shadow, &a_cap, &a_rounds, &a_clow,   &a_profile, &a_dense};
      hipKernelNodeParams kp{};
      kp.func = (void*)sssp_relax_kernel;
      kp.gridDim = dim3(blocks);
      kp.blockDim = dim3(256);
      kp.sharedMemBytes = 0;
      kp.kernelParams = args;
      kp.extra = nullptr;
      hipGraphNode_t node;
      HIP_CHECK(hipGraphAddKernelNode(&node, g.graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
      prev = node;
    }
    {
      uint32_t a_count = count;
      uint32_t* a_host = h_imp + which * IMP_RING;
      void* args[] = {&a_ctl, &a_imp, &a_count, &a_host};
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
  }

  // `defer_advance`: the caller queues a kernel behind the batch that closes it (sssp_tail_kernel mirrors the flags and
  // advances the base itself) and records the event
  SweepBatch enqueue_batch(hipEvent_t ev, bool defer_advance = false) {
    hipStream_t st = ctx->stream;
    // after the first batch: constant small batches while the solve is shallow, larger ones for deep lattices
    SweepBatch b{next_sweep, 8u, 1};
    if (next_sweep == 0) b = SweepBatch{0u, first_count, 0};
    else if (next_sweep >= 64) b = SweepBatch{next_sweep, MAX_BATCH, 2};
    if (use_graphs && !sv->mbox && !sv->binned) {
      HIP_CHECK(hipGraphLaunch(get_graph(b.which, b.count), st));
    } else {
      // plain launches: the GPU starts on the first sweep while the host is still queueing the rest (a graph replay of
      // N nodes only starts after ~2.7 us x N of host-side work: 89 us for the 32-sweep replay, profiles/r01g)
      const bool time_chain = ctx->chain_timing && b.first == 0;
      if (time_chain) HIP_CHECK(hipEventRecord(ctx->ev_chain[0], st));
      for (uint32_t j = 0; j < b.count; ++j) launch_sweep(f, *sv, n, st, j, j, b.first + j, 0u);
      if (time_chain) HIP_CHECK(hipEventRecord(ctx->ev_chain[1], st));
      if (!defer_advance) sssp_advance_kernel<<<1, 64, 0, st>>>(sv->ctl.p, sv->improved.p, b.count, h_imp + b.which * IMP_RING);
      HIP_CHECK(hipGetLastError());
    }
    if (!defer_advance) HIP_CHECK(hipEventRecord(ev, st));
    next_sweep += b.count;
    return b;
  }
  uint32_t* host_flags(const SweepBatch& b) const { return h_imp + b.which * IMP_RING; }

  bool aborted = false;    // a resident launch raised FLAG_RES_ABORT
  uint64_t seen_busy = 0;  // bit k: launch k was a busy WIDE sweep (flag value 1 + MODE_WIDE)
  bool scan_flags(const SweepBatch& b) {  // true when a sweep of the batch changed nothing
    const uint32_t* hf = h_imp + b.which * IMP_RING;
    for (uint32_t k = 0; k < b.count; ++k) {
      sweeps_done = b.first + k + 1;
      const uint32_t v = hf[(b.first + k) % IMP_RING];
      if (!v) return true;
      if (v == FLAG_RES_ABORT) {  // a resident launch gave up waiting for its own workgroups: the caller solves again without
        aborted = true;
        return true;
      }
      if (v == FLAG_NARROW_CLEAN) return true;  // a NARROW launch that left nothing anywhere: the fixed point, certified
      // (WIDE and COLLECT launches have every block awake and loading its keys: no gate next time)
      if ((v == 1u + MODE_WIDE || v == 1u + MODE_COLLECT) && b.first + k < 64) seen_busy |= 1ull << (b.first + k);
    }
    return false;
  }
  // which launches of the next solve of this FST may skip the gate (they were busy WIDE sweeps this time)
  uint64_t hint_mask() const { return ~seen_busy; }

  void start(bool defer_advance = false) { cur = enqueue_batch(evs[0], defer_advance); }

  // The fused tail's ticket in pinned memory (wfst_sp_job::h_tail->done), when the first batch ends with one: the host waits
  // for that word — every result of the launch chain is in host memory before it — and falls back to the event.
  const volatile uint32_t* done_word = nullptr;
  uint32_t done_ticket = 0;
  bool done_seen = false;  // the ticket arrived: nothing of this job is still running on the stream
  void wait_first_batch() {
    if (done_word && !std::getenv("WFST_SSSP_EVENT_WAIT")) {
      const auto t0 = std::chrono::steady_clock::now();
      for (uint32_t spins = 0;; ++spins) {
        if (*done_word == done_ticket) {
          std::atomic_thread_fence(std::memory_order_acquire);
          done_seen = true;
          return;
        }
        cpu_relax();
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;  // (a long solve: sleep on the event)
      }
    }
    HIP_CHECK(hipEventSynchronize(evs[0]));
  }

  void finish() {
    int which = 0;
    bool done = false;
    if (predicted) {  // expected to finish inside the first batch: no idle sweeps were queued behind it
      wait_first_batch();
      done = scan_flags(cur);
      if (!done) {
        extended = true;
        cur = enqueue_batch(evs[0]);
      }
    }
    while (!done) {
      if (next_sweep > sv->sweep_cap) throw Error("shortest_path: relaxation did not converge (negative-weight cycle?)");
      // keep the device busy while the host inspects `cur`: the next batch is enqueued first
      const SweepBatch nxt = enqueue_batch(evs[which ^ 1]);
      extended = true;
      HIP_CHECK(hipEventSynchronize(evs[which]));
      done = scan_flags(cur);
      cur = nxt;
      which ^= 1;
    }
  }
};

// tuning aid: the phase stamps of a mailbox solve go to the file named by WFST_SSSP_MBOX_TRACE (u64 [64][nb][16])
void mbox_dump_trace(wfst_ctx* ctx, Solve& sv) {
  const char* path = std::getenv("WFST_SSSP_MBOX_TRACE");
  if (!sv.mbox || !sv.mb_dbg.p || !path) return;
  std::vector<unsigned long long> h((size_t)MB_DBG_SWEEPS * sv.mv.nb * 16);
  HIP_CHECK(hipMemcpyAsync(h.data(), sv.mb_dbg.p, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (FILE* fp = std::fopen(path, "wb")) {
    const uint32_t hdr[2] = {MB_DBG_SWEEPS, sv.mv.nb};
    std::fwrite(hdr, 4, 2, fp);
    std::fwrite(h.data(), 8, h.size(), fp);
    std::fclose(fp);
  }
}

// tuning aid: the level stamps of the resident launches go to the file named by WFST_SSSP_RES_TRACE (u64 [64][nb][4])
void res_dump_trace(wfst_ctx* ctx, Solve& sv) {
  const char* path = std::getenv("WFST_SSSP_RES_TRACE");
  if (!sv.resident || !sv.rs_trace.p || !path) return;
  std::vector<unsigned long long> h((size_t)RS_TRACE_LEVELS * sv.mv.nb * 4);
  HIP_CHECK(hipMemcpyAsync(h.data(), sv.rs_trace.p, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (FILE* fp = std::fopen(path, "wb")) {
    const uint32_t hdr[2] = {RS_TRACE_LEVELS, sv.mv.nb};
    std::fwrite(hdr, 4, 2, fp);
    std::fwrite(h.data(), 8, h.size(), fp);
    std::fclose(fp);
  }
}

// Runs the relaxation to its fixed point. f must have a device copy and a start state.
void run_relaxation(wfst_ctx* ctx, const wfst_fst* f, Solve& sv) {
  relax_setup(ctx, f, sv);
  const uint32_t n = f->n_states;
  hipStream_t st = ctx->stream;
  uint8_t* const* fl = sv.fl;
  const uint32_t blocks = sv.blocks;
  const uint64_t sweep_cap = sv.sweep_cap;
  ctx->stats.sweeps = 0;

  uint32_t sweeps_done = 0;
  if (ctx->profiling) {
    // one sweep at a time, bracketed by events; a counting kernel (outside the events) sizes the frontier
    uint32_t* h_imp = (uint32_t*)ctx->pinned.get(64 + sizeof(Ctl));
    Ctl* h_ctl = (Ctl*)((char*)h_imp + 64);
    ctx->sweep_trace.clear();
    uint64_t prev_arcs = 0, prev_states = 0;
    auto shard_sum = [](const unsigned long long* v) {
      uint64_t t = 0;
      for (uint32_t j = 0; j < PROF_SHARDS; ++j) t += v[j * PROF_STRIDE];
      return t;
    };
    for (uint32_t k = 0;; ++k) {
      if (k > sweep_cap) throw Error("shortest_path: relaxation did not converge (negative-weight cycle?)");
      sssp_nop_kernel<<<blocks, 256, 0, st>>>(fl[k & 1u], n, sv.improved.p);
      HIP_CHECK(hipEventRecord(ctx->ev0, st));
      launch_sweep(f, sv, n, st, k, 0u, k, std::getenv("WFST_SSSP_COUNT_ATOMICS") ? 2u : 1u);  // counts the states / arcs it relaxes
      HIP_CHECK(hipEventRecord(ctx->ev1, st));
      sssp_advance_kernel<<<1, 64, 0, st>>>(sv.ctl.p, sv.improved.p, 1u, nullptr);
      HIP_CHECK(hipMemcpyAsync(h_imp, sv.improved.p + (k % IMP_RING), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipMemcpyAsync(h_ctl, sv.ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
      const uint64_t arcs = shard_sum(h_ctl->arcs), states = shard_sum(h_ctl->states);
      ctx->sweep_trace.push_back({(double)ms, arcs - prev_arcs, states - prev_states, (sv.mbox || sv.binned) ? h_ctl->mode[k % RING] : 0u});
      prev_arcs = arcs;
      prev_states = states;
      ctx->stats.relax_ms += ms;
      ctx->stats.relax_launches += 1;
      sweeps_done = k + 1;
      if (!h_imp[0]) break;
    }
    ctx->stats.relax_arcs += prev_arcs;
    ctx->stats.relax_states += prev_states;
  } else {
    SweepDriver drv;
    drv.init(ctx, f, &sv);
    drv.start();
    drv.finish();
    if (drv.aborted) {  // a resident launch gave up waiting (its grid was not resident as a whole): one launch per level
      HIP_CHECK(hipStreamSynchronize(st));
      ctx->resident_aborted();
      ctx->stats.resident_aborts += 1;
      sv.lease.release();
      struct Hold {
        wfst_ctx* c;
        ~Hold() { c->resident_hold = false; }
      } hold{ctx};
      ctx->resident_hold = true;
      run_relaxation(ctx, f, sv);
      return;
    }
    if (sv.resident) ctx->resident_completed();
    sweeps_done = drv.sweeps_done;
    f->last_hint_mask.store(drv.hint_mask(), std::memory_order_relaxed);
  }
  sv.lease.release();
  sv.sweeps = sweeps_done;
  ctx->stats.sweeps = sweeps_done;
  note_sweeps(f, sweeps_done);
  mbox_dump_trace(ctx, sv);
  res_dump_trace(ctx, sv);
}

// Builds the linear output FST exactly as single_shortest_path_backtrace does, including the property
// word (add_state / set_final / add_tr / set_start bookkeeping, then shortest_path_properties(.., true)).
wfst_fst* build_path_fst(wfst_ctx* ctx, bool has_path, uint32_t hops, float final_weight, const wfst_tr* path_arcs) {
  return make_path_fst(ctx, has_path, hops, final_weight, path_arcs);
}

// The order in which the reference relaxes the states of an acyclic FST (queues/auto_queue.rs:23-99): state order when the
// TOP_SORTED bit is set, else the topological order of dfs_visit (dfs_visit.rs:97-187: root = start, then 0, 1, 2, ...; arcs
// in stored order) — TopOrderQueue from TopOrderVisitor (top_sort.rs:12-61) when the ACYCLIC bit is known, from the SCC
// numbering otherwise (scc_visitors.rs:173-179): both are the reverse finishing order.  false when the reference would use
// another discipline: LIFO (unweighted input), or an SCC queue (the input has a cycle).
bool reference_top_rank(const wfst_fst* f, std::vector<uint32_t>& rank) {
  const uint32_t n = f->n_states;
  if (f->start < 0 || n == 0) return false;
  ensure_host(f);
  const HostCsr& h = f->host;
  rank.resize(n);
  if (f->props & props::TOP_SORTED) {
    for (uint32_t s = 0; s < n; ++s) rank[s] = s;
    return true;
  }
  if (!(f->props & props::ACYCLIC)) {
    if (f->props & props::UNWEIGHTED) return false;  // LifoQueue
    bool unweighted = true;                          // scc_queue_type's scan
    for (const wfst_tr& a : h.arcs)
      if (!props::is_zero(a.weight) && !props::is_one(a.weight)) {
        unweighted = false;
        break;
      }
    if (unweighted) return false;
  }
  enum : uint8_t { White, Grey, Black };
  std::vector<uint8_t> color(n, White);
  std::vector<std::pair<uint32_t, uint32_t>> stack;
  uint32_t finished = 0;
  uint32_t root = (uint32_t)f->start;
  while (root < n) {
    color[root] = Grey;
    stack.push_back({root, h.offsets[root]});
    while (!stack.empty()) {
      const uint32_t s = stack.back().first;
      uint32_t& pos = stack.back().second;
      if (pos >= h.offsets[s + 1]) {
        color[s] = Black;
        rank[s] = n - 1 - finished++;
        stack.pop_back();
        continue;
      }
      const uint32_t t = h.arcs[pos++].nextstate;
      if (color[t] == White) {
        color[t] = Grey;
        stack.push_back({t, h.offsets[t]});
      } else if (color[t] == Grey) {
        return false;  // a cycle: the reference uses an SCC queue
      }
    }
    root = root == (uint32_t)f->start ? 0 : root + 1;
    while (root < n && color[root] != White) root++;
  }
  return true;
}

// shortest_path (nshortest = 1) with the reference's choice among tied optima, for an acyclic input whose relaxation order
// `rank` is known: distances on the GPU as always, then the final state and the predecessors by the rule above.
wfst_fst* shortest_path_reference_order(wfst_ctx* ctx, const wfst_fst* f, const std::vector<uint32_t>& rank) {
  const uint32_t n = f->n_states;
  hipStream_t st = ctx->stream;
  Solve sv;
  run_relaxation(ctx, f, sv);
  std::vector<uint32_t> r2s(n);
  for (uint32_t s = 0; s < n; ++s) r2s[rank[s]] = s;
  DBuf<uint32_t> d_rank(*ctx->pool, n), d_r2s(*ctx->pool, n);
  HIP_CHECK(hipMemcpyAsync(d_rank.p, rank.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(d_r2s.p, r2s.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  DBuf<unsigned long long> parent(*ctx->pool, n);
  DBuf<wfst_tr> out(*ctx->pool, n);
  HIP_CHECK(hipMemsetAsync(parent.p, 0xFF, (size_t)n * sizeof(unsigned long long), st));
  HIP_CHECK(hipMemsetAsync(&sv.ctl.p->best, 0xFF, sizeof(unsigned long long), st));
  HIP_CHECK(hipMemsetAsync(&sv.ctl.p->pad, 0, sizeof(uint32_t), st));
  sssp_final_ref_kernel<<<std::min<uint32_t>((n + 255) / 256, (uint32_t)ctx->n_cus * 4), 256, 0, st>>>(f->dev.finals, sv.key.p, d_rank.p, n,
                                                                                                          sv.ctl.p);
  const uint32_t blocks = std::min<uint32_t>((uint32_t)ctx->n_cus * 8, (uint32_t)(((uint64_t)n * GROUP + 255) / 256));
  sssp_parent_ref_kernel<<<blocks, 256, 0, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, d_rank.p, parent.p, n);
  sssp_backtrace_ref_kernel<<<1, 64, 0, st>>>(f->dev.offsets, f->dev.arcs, f->dev.finals, parent.p, d_r2s.p, (uint32_t)f->start, sv.ctl.p,
                                              out.p, n);
  Ctl* hc = (Ctl*)ctx->pinned.get(sizeof(Ctl));
  HIP_CHECK(hipMemcpyAsync(hc, sv.ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(st));
  if (hc->pad & 1u) throw Error("shortest_path: hop count overflow in the mailbox sweeps (internal error)");
  if (hc->pad & 4u) throw Error("shortest_path: no admissible predecessor on the path (internal error)");
  if (!hc->has_path) return build_path_fst(ctx, false, 0, INF, nullptr);
  const uint32_t len = hc->hops;
  const float final_weight = hc->final_weight;
  std::vector<wfst_tr> path(len);
  if (len) HIP_CHECK(hipMemcpy(path.data(), out.p, (size_t)len * sizeof(wfst_tr), hipMemcpyDeviceToHost));
  return build_path_fst(ctx, true, len, final_weight, path.data());
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

// Transpose of f (in-arcs as {source, position}); built the SECOND time shortest_path sees the same large FST — a one-shot
// query keeps the parent pass, a resident transducer that is queried again pays the build once, inside its second query:
// ~0.3 ms for 10 M arcs through the mailbox plan (rev_bucket_kernel / rev_place_kernel), ~1.2 ms by the two atomic passes for
// FSTs without a plan.  (Building it on a second stream beside the FIRST query was measured: that query's resident launch
// waits for the compute units the build holds — 0.65 -> 1.8 ms, and a one-shot query pays for a transpose it never uses.)
// `force`: the caller needs the in-arcs of the path's states now (tie order 1 on a cyclic input).
namespace {
void reverse_csr_build(wfst_ctx* ctx, const wfst_fst* f, RevCsr& r) {
  const uint32_t n = f->n_states;
  hipStream_t st = ctx->stream;
  DevicePool& owner_pool = f->owner_pool ? *f->owner_pool : *ctx->pool;  // cached with the handle: the owner's pool outlives it
  r.off = DBuf<uint32_t>(owner_pool, (size_t)n + 1);
  r.arc = DBuf<uint4>(owner_pool, f->n_arcs);
  const MboxPlan* plan = f->mbox ? f->mbox.get() : f->mbox13.get();  // (cache_mu is held by the caller)
  if (plan && !(std::getenv("WFST_SSSP_TRANSPOSE_PLAN") && std::atoi(std::getenv("WFST_SSSP_TRANSPOSE_PLAN")) == 0)) {
    DBuf<uint4> rec(*ctx->pool, f->n_arcs);  // one record per arc, bucketed by destination block
    if (plan->log == 13) {
      rev_bucket_kernel<13><<<plan->nb, 1024, plan->nb * sizeof(uint32_t), st>>>(f->dev.offsets, f->dev.wn, n, plan->nb, plan->roff_t.p, rec.p);
      rev_place_kernel<13><<<plan->nb, 1024, 0, st>>>(plan->roff.p, plan->nb, n, rec.p, r.off.p, r.arc.p);
    } else {
      rev_bucket_kernel<12><<<plan->nb, 1024, plan->nb * sizeof(uint32_t), st>>>(f->dev.offsets, f->dev.wn, n, plan->nb, plan->roff_t.p, rec.p);
      rev_place_kernel<12><<<plan->nb, 1024, 0, st>>>(plan->roff.p, plan->nb, n, rec.p, r.off.p, r.arc.p);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st));  // rec is released here
    return;
  }
  DBuf<uint32_t> indeg(*ctx->pool, (size_t)n + 1), cursor(*ctx->pool, n);
  HIP_CHECK(hipMemsetAsync(indeg.p, 0, ((size_t)n + 1) * sizeof(uint32_t), st));
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((f->n_arcs + 255) / 256, (uint64_t)ctx->n_cus * 8);
  rev_count_kernel<<<blocks, 256, 0, st>>>(f->dev.wn, f->n_arcs, indeg.p);
  // rev_off = exclusive scan of the in-degrees (n + 1 outputs: the extra zero input makes rev_off[n] the total)
  size_t temp_bytes = 0;
  HIP_CHECK(rocprim::exclusive_scan(nullptr, temp_bytes, indeg.p, r.off.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st));
  DBuf<uint8_t> temp(*ctx->pool, temp_bytes);
  HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, indeg.p, r.off.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st));
  HIP_CHECK(hipMemcpyAsync(cursor.p, r.off.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  const uint32_t fblocks = std::min<uint32_t>((uint32_t)ctx->n_cus * 8, (uint32_t)(((uint64_t)n * GROUP + 255) / 256));
  rev_fill_kernel<<<fblocks, 256, 0, st>>>(f->dev.offsets, f->dev.wn, n, cursor.p, r.arc.p);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(st));  // indeg / cursor / temp are released here
}
}  // namespace

const RevCsr* reverse_csr(wfst_ctx* ctx, const wfst_fst* f, bool force = false) {
  std::lock_guard<std::mutex> lk(f->cache_mu);
  if (f->rev_dev) return f->rev_dev.get();
  if (f->n_arcs >= 0xFFFFFFFFull || f->n_arcs == 0) return nullptr;
  if (!force) {
    if (f->sp_queries.fetch_add(1) + 1 < 2 || f->n_arcs < (1u << 18)) return nullptr;
    if (const char* e = std::getenv("WFST_SSSP_TRANSPOSE")) if (std::atoi(e) == 0) return nullptr;
  }
  auto r = std::make_shared<RevCsr>();
  reverse_csr_build(ctx, f, *r);
  f->rev_dev = r;
  return r.get();
}

}  // namespace wfst

// One single-shortest-path solve split in two halves: begin queues the set-up, the first batch of sweeps and (when the
// solve is predicted to fit in that batch and the transpose is cached) the final-state search, the backtrace and the
// read-back of the path behind it, then returns; end waits, continues the sweeps if the prediction was short, and
// builds the output FST.  The synchronous shortest_path(nshortest = 1) is begin followed by end.
struct wfst_sp_job {
  wfst_ctx* ctx = nullptr;
  const wfst_fst* f = nullptr;
  bool trivial = false;      // no start state: the result is the empty FST (shortest_path.rs:185-187)
  wfst_fst* ready = nullptr;  // the result, already computed (reference tie order: a synchronous path)
  bool tail_queued = false;  // final / header / backtrace / read-back already queued behind the first batch
  bool fused_tail = false;   // ... as the one-launch sssp_tail_kernel (result header in h_tail instead of hc)
  wfst::Solve sv;
  wfst::SweepDriver drv;
  const wfst::RevCsr* rev = nullptr;
  wfst::Ctl* hc = nullptr;
  wfst_tr* h_path = nullptr;
  wfst::TailOut* h_tail = nullptr;  // header of the result, written by sssp_tail_kernel (transpose cached)
  uint32_t done_ticket = 0;         // what the fused tail writes into h_tail->done when everything else is in host memory
  bool need_unique = false;         // tie order 1 on an input the reference does not relax in a topological order: the result
                                    // is returned only when the optimum is unique (then it IS the reference's), KO otherwise
};

namespace wfst {
namespace {
constexpr uint32_t PATH_PINNED = 4096;  // arcs of the path written straight into pinned memory by the backtrace

// `adv`: the sweep batch this tail closes (enqueued with defer_advance), or null
void queue_tail(wfst_sp_job* j, const SweepBatch* adv = nullptr) {
  wfst_ctx* ctx = j->ctx;
  const wfst_fst* f = j->f;
  const uint32_t n = f->n_states;
  hipStream_t st = ctx->stream;
  Solve& sv = j->sv;
  if (j->rev && !std::getenv("WFST_SSSP_SPLIT_TAIL")) {  // one launch, header straight into pinned memory
    sssp_tail_kernel<<<std::min<uint32_t>(TAIL_BLOCKS, (n + 1023) / 1024), 1024, 0, st>>>(
        f->dev.finals, sv.key.p, n, sv.ctl.p, f->dev.offsets, f->dev.arcs, j->rev->off.p, j->rev->arc.p, j->h_path,
        PATH_PINNED, j->h_tail, sv.improved.p, adv ? adv->count : 0u, adv ? j->drv.host_flags(*adv) : nullptr, j->done_ticket);
    j->fused_tail = true;
    return;
  }
  if (adv) throw Error("shortest_path: internal error (deferred advance without the one-launch tail)");
  j->fused_tail = false;
  sssp_final_kernel<<<std::min<uint32_t>((n + 255) / 256, (uint32_t)ctx->n_cus * 4), 256, 0, st>>>(f->dev.finals, sv.key.p,
                                                                                                          n, sv.ctl.p);
  sssp_header_kernel<<<1, 1, 0, st>>>(f->dev.finals, sv.key.p, sv.ctl.p);
  if (j->rev)
    sssp_backtrace_rev_kernel<<<1, 64, 0, st>>>(f->dev.offsets, f->dev.arcs, sv.key.p, j->rev->off.p,
                                                j->rev->arc.p, sv.ctl.p, j->h_path, PATH_PINNED);
  HIP_CHECK(hipMemcpyAsync(j->hc, sv.ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, st));
}
}  // namespace

wfst_sp_job* shortest_path_n1_begin(wfst_ctx* ctx, const wfst_fst* f) {
  std::unique_ptr<wfst_sp_job> j(new wfst_sp_job());
  j->ctx = ctx;
  j->f = f;
  const uint32_t n = f->n_states;
  if (f->start < 0 || n == 0) {
    j->trivial = true;
    return j.release();
  }
  ensure_device(const_cast<wfst_fst*>(f));
  if (ctx->tie_reference) {  // the reference's own choice among tied optima, where its relaxation order is a topological one
    std::vector<uint32_t> rank;
    ctx->stats.tied_choices = WFST_TIES_UNKNOWN;
    if (reference_top_rank(f, rank)) {
      j->ready = shortest_path_reference_order(ctx, f, rank);
      return j.release();
    }
    // A cycle (the reference relaxes inside an SCC queue) or unit weights (LIFO): its choice among tied optima is a function
    // of its whole relaxation history there.  A UNIQUE optimum needs no choice: the canonical path is then the reference's
    // path, and the tail kernel counts the tied choices along it (sssp_walk_back) — with any, the call is KO.
    j->need_unique = true;
  }
  ctx->stats.tied_choices = WFST_TIES_UNKNOWN;
  char* pin = (char*)ctx->pinned.get(sizeof(Ctl) + 128 + PATH_PINNED * sizeof(wfst_tr));
  j->hc = (Ctl*)pin;
  j->h_tail = (TailOut*)(pin + ((sizeof(Ctl) + 63) & ~(size_t)63));
  j->h_path = (wfst_tr*)(pin + ((sizeof(Ctl) + 63) & ~(size_t)63) + 64);
  static_assert(sizeof(TailOut) <= 64, "the path's arcs follow the header at + 64");
  static std::atomic<uint32_t> tickets{0};
  do j->done_ticket = tickets.fetch_add(1, std::memory_order_relaxed) + 1u; while (j->done_ticket == 0u);
  j->h_tail->done = 0u;
  j->rev = reverse_csr(ctx, f, j->need_unique);  // may build the transpose (second query of a large FST): before anything is queued
  if (ctx->profiling) {
    run_relaxation(ctx, f, j->sv);  // per-sweep events: synchronous
    return j.release();
  }
  relax_setup(ctx, f, j->sv);
  ctx->stats.sweeps = 0;
  j->drv.init(ctx, f, &j->sv);
  const bool fuse = j->drv.predicted && j->rev && !std::getenv("WFST_SSSP_SPLIT_TAIL") &&
                    !(j->drv.use_graphs && !j->sv.mbox && !j->sv.binned);  // (a sweep graph carries its own advance node)
  j->drv.start(/*defer_advance=*/fuse);
  if (fuse) {  // the tail closes the batch: flags to the host, base advanced, then the ticket (and the event) finish() waits for
    queue_tail(j.get(), &j->drv.cur);
    HIP_CHECK(hipEventRecord(j->drv.evs[0], ctx->stream));
    j->tail_queued = true;
    j->drv.done_word = &j->h_tail->done;
    j->drv.done_ticket = j->done_ticket;
  } else if (j->drv.predicted && j->rev) {
    queue_tail(j.get());
    j->tail_queued = true;
  }
  return j.release();
}

wfst_fst* shortest_path_n1_end(wfst_sp_job* job) {
  std::unique_ptr<wfst_sp_job> j(job);
  wfst_ctx* ctx = j->ctx;
  const wfst_fst* f = j->f;
  if (j->trivial) {
    ctx->stats.tied_choices = 0;
    return build_path_fst(ctx, false, 0, INF, nullptr);
  }
  if (j->ready) return j->ready;
  const uint32_t n = f->n_states;
  hipStream_t st = ctx->stream;
  Solve& sv = j->sv;
  if (!ctx->profiling) {
    j->drv.finish();
    if (j->drv.aborted) {  // a resident launch gave up waiting: the whole query again, one launch per level
      HIP_CHECK(hipStreamSynchronize(st));
      ctx->resident_aborted();
      ctx->stats.resident_aborts += 1;
      j.reset();  // (its buffers go back to the pool, the lease with them)
      struct Hold {
        wfst_ctx* c;
        ~Hold() { c->resident_hold = false; }
      } hold{ctx};
      ctx->resident_hold = true;
      return shortest_path_n1_end(shortest_path_n1_begin(ctx, f));
    }
    if (sv.resident) ctx->resident_completed();
    sv.lease.release();
    sv.sweeps = j->drv.sweeps_done;
    ctx->stats.sweeps = sv.sweeps;
    note_sweeps(f, sv.sweeps);
    f->last_hint_mask.store(j->drv.hint_mask(), std::memory_order_relaxed);
    mbox_dump_trace(ctx, sv);
    res_dump_trace(ctx, sv);
  }
  if (j->tail_queued && j->drv.extended) {  // the speculative tail ran on unfinished distances: once more
    HIP_CHECK(hipMemsetAsync(&sv.ctl.p->best, 0xFF, sizeof(unsigned long long), st));
    j->tail_queued = false;
  }
  if (!j->tail_queued) queue_tail(j.get());
  // (the ticket of the fused tail was the last thing this job had on the stream: nothing to wait for)
  if (!(j->tail_queued && j->fused_tail && j->drv.done_seen && !j->drv.extended) || ctx->chain_timing) HIP_CHECK(hipStreamSynchronize(st));
  else HIP_CHECK(hipPeekAtLastError());  // (no stream wait on this path.  Reports launch-configuration / sticky API errors the runtime has already
                                              // seen — NOT an asynchronous kernel fault: a chain that faults never writes its ticket, and the spin above then
                                              // ends in the stream wait, which reports it.  Peek, not Get: the error may belong to another thread's launch)
  if (ctx->chain_timing && !ctx->profiling) {  // the sweeps of this query as one chain (wfst_ctx_set_profiling(ctx, 2))
    ctx->stats.relax_ms = 0.0;
    ctx->stats.relax_launches = 0;
    if (j->drv.predicted && !j->drv.extended && !(j->drv.use_graphs && !sv.mbox && !sv.binned)) {
      float ms = 0.0f;
      HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_chain[0], ctx->ev_chain[1]));
      ctx->stats.relax_ms = ms;
      ctx->stats.relax_launches = j->drv.first_count;
    }
  }
  const Ctl* hc = j->hc;
  const uint32_t r_pad = j->fused_tail ? j->h_tail->pad : hc->pad, r_has_path = j->fused_tail ? j->h_tail->has_path : hc->has_path;
  if (r_pad & 1u) throw Error("shortest_path: hop count overflow in the mailbox sweeps (internal error)");
  if (r_pad & 4u) throw Error("shortest_path: no admissible predecessor on the path (inexact weight sums with negative weights)");
  if (!r_has_path) {  // no final state is reachable: the empty FST, and nothing to choose
    ctx->stats.tied_choices = 0;
    return build_path_fst(ctx, false, 0, INF, nullptr);
  }
  const uint32_t hops = j->fused_tail ? j->h_tail->hops : hc->hops;
  const float final_weight = j->fused_tail ? j->h_tail->final_weight : hc->final_weight;
  // tied choices along the returned path: counted by the one-launch tail's walk over the in-arcs (unknown on the other paths)
  const uint64_t ties = j->fused_tail && !(r_pad & 8u) ? (uint64_t)j->h_tail->ties : WFST_TIES_UNKNOWN;
  ctx->stats.tied_choices = ties;
  if (j->need_unique) {
    if (ties == WFST_TIES_UNKNOWN)
      throw Error("shortest_path: tie order 1 on a cyclic input: the optimum could not be certified unique (no transpose for this input, or a path "
                  "beyond the walk's buffer)");
    if (ties != 0)
      throw Error("shortest_path: ambiguous optimum: " + std::to_string(ties) + " tied choice(s) on the optimal path of a cyclic input — the "
                  "reference's choice there depends on its relaxation order (tie order 1 returns only what is provably its result)");
  }
  if (j->rev && !(r_pad & 8u)) return build_path_fst(ctx, true, hops, final_weight, j->h_path);
  // the parent pass (first query of an FST, or a path longer than the pinned buffer): a path has at most n - 1 arcs
  std::vector<wfst_tr> path;
  uint32_t len = 0;
  {
    DBuf<unsigned long long> parent(*ctx->pool, n);
    HIP_CHECK(hipMemsetAsync(parent.p, 0xFF, (size_t)n * sizeof(unsigned long long), st));
    HIP_CHECK(hipMemsetAsync(&sv.ctl.p->pad, 0, sizeof(uint32_t), st));
    const uint32_t blocks = std::min<uint32_t>((uint32_t)ctx->n_cus * 8, (uint32_t)(((uint64_t)n * GROUP + 255) / 256));
    sssp_parent_kernel<<<blocks, 256, 0, st>>>(f->dev.offsets, f->dev.wn, sv.key.p, parent.p, n);
    // the walk is as long as the final state's hop count except where class-1 predecessors lengthen it: room for four times
    // that (not 16 B per STATE of a multi-million-state input for a path of a few thousand arcs), all of n on the rare retry
    uint32_t cap = (uint32_t)std::min<uint64_t>(n, 4ull * hops + 1024);
    DBuf<wfst_tr> out(*ctx->pool, cap);
    for (;;) {
      sssp_backtrace_kernel<<<1, 64, 0, st>>>(f->dev.offsets, f->dev.arcs, sv.key.p, parent.p, sv.ctl.p, out.p, cap);
      HIP_CHECK(hipMemcpyAsync(j->hc, sv.ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      if (!(hc->pad & 16u) || cap >= n) break;
      cap = n;
      out = DBuf<wfst_tr>(*ctx->pool, cap);
      HIP_CHECK(hipMemsetAsync(&sv.ctl.p->pad, 0, sizeof(uint32_t), st));
    }
    if (hc->pad & 4u) throw Error("shortest_path: no admissible predecessor on the path (inexact weight sums with negative weights)");
    len = hc->hops;
    path.resize(len);
    if (len) HIP_CHECK(hipMemcpy(path.data(), out.p, (size_t)len * sizeof(wfst_tr), hipMemcpyDeviceToHost));
  }
  return build_path_fst(ctx, true, len, final_weight, path.data());
}

void shortest_path_n1_abandon(wfst_sp_job* job) {
  std::unique_ptr<wfst_sp_job> j(job);
  delete j->ready;
  if (!j->trivial) (void)hipStreamSynchronize(j->ctx->stream);  // the solve's buffers go back to the pool after this
}

wfst_ctx* sp_job_ctx(wfst_sp_job* job) { return job->ctx; }

wfst_fst* shortest_path_n1(wfst_ctx* ctx, const wfst_fst* f) {
  wfst_fst* tiny = nullptr;
  if (shortest_path_n1_tiny(ctx, f, &tiny)) return tiny;
  return shortest_path_n1_end(shortest_path_n1_begin(ctx, f));
}

}  // namespace wfst
