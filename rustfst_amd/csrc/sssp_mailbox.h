// sssp_mailbox.h — owner-computes relaxation sweeps ("mailbox sweeps") for sssp.hip.
//
// Included by sssp.hip inside namespace wfst { namespace { ... } } after Ctl / enc_f32.
//
// Same recurrence as sssp_relax_kernel (single_shortest_path, rustfst/src/algorithms/shortest_path.rs:173-239:
// relax arc (s,w,t) with nd = d[s] (x) w, keep the minimum), same (d, hops) key, same near-far schedule — the
// least fixed point is unique, so the keys it leaves are bit-identical.  What changes is WHERE the minimum is taken:
//
//   * the states are cut into blocks of MB_B = 4096; workgroup j OWNS block j and is the only one that writes key[] of
//     its states during a WIDE sweep.  It keeps the block's 4096 keys in LDS for the duration of the sweep (32 KB);
//   * a relaxation is a MESSAGE {enc(d[s] + w), hops[s] + 1, t mod 4096} (8 bytes) written by the workgroup that
//     owns s into the region reserved for the pair (block of s -> block of t).  Regions are static: region (i -> j)
//     has room for every arc from block i to block j (one message per arc per sweep at most), so a slot is one LDS
//     atomicAdd on a per-destination cursor — no global atomic, no reservation pass;
//   * next sweep the owner reads its regions (contiguous per destination: coalesced), applies the candidates with
//     LDS atomicMin, and expands the states that changed (or were waiting beyond the near-far threshold).
//
// A launch ("slot") runs in one of three MODES, decided on the device by every workgroup from the same few words of the
// control block (mbox_schedule), so the host still queues a solve's launches back to back without looking:
//   WIDE     the owner-computes sweep above: one level of the search per launch (~8.6 us even for one state);
//   NARROW   the frontier is a handful of states per block (the head of the search, the tail of the last band): every
//            workgroup follows ITS OWN discoveries level after level inside the launch, with global atomicMin on key[]
//            and a work list in LDS whose entries carry key and arc range (two dependent trips per level: arc rows, then
//            atomic + the target's offsets together); nothing is handed to another workgroup, so there is no barrier
//            between workgroups.  States beyond the threshold (or more than the list holds) are left in the waiting
//            masks for the next WIDE sweep;
//   COLLECT  the hand-over WIDE -> NARROW: the owners apply their inboxes as usual but, instead of expanding, write
//            the states they would have expanded into their segment of the global work list.
// Relaxation order does not change the fixed point, so the modes only change how fast it is reached.
//
// Limits: hop counts < 2^20 (20 hop bits + 12 state bits share a word of the message; checked at run time), no negative
// weights, n_arcs < 2^31, at most MB_NBMAX_BIG blocks (8M states).  Anything else takes sssp_relax_kernel.

constexpr uint32_t MB_LOG = 12;
constexpr uint32_t MB_B = 1u << MB_LOG;  // states per block
constexpr uint32_t MB_NBMAX = 256;       // blocks handled with one inbox region per 4 lanes in one pass (n <= 2^20)
constexpr uint32_t MB_NBMAX_BIG = 2048;  // blocks at most (n <= 2^23): inbox regions in passes of 256, shallower staging
constexpr uint32_t MB_NB_DEFAULT = 768;  // blocks up to which the mailbox launches are the default choice (3.1M states)
#ifndef WFST_MB_THREADS
#define WFST_MB_THREADS 1024
#endif
constexpr uint32_t MB_THREADS = WFST_MB_THREADS;
constexpr uint32_t MB_LPR = MB_THREADS / 256;  // lanes that share an inbox region / a destination's staged run
constexpr uint32_t MB_HOP_BITS = 32 - MB_LOG;
constexpr uint32_t MB_UNROLL = 8 * (1024 / MB_THREADS);  // active states a 16-lane group relaxes at once (independent load chains per lane)
#ifndef WFST_MB_STG_MAX
#define WFST_MB_STG_MAX 24
#endif
constexpr uint32_t MB_STG_MAX = WFST_MB_STG_MAX;  // messages per destination staged in LDS between two flushes (the rest is stored directly)
constexpr uint32_t MB_DYN_BUDGET = 100u * 1024u;  // dynamic LDS (staging + per-destination cursors) next to 57 KB static

// modes of a launch (also what the activity flag of a sweep holds, + 1)
constexpr uint32_t MODE_WIDE = 0, MODE_COLLECT = 1, MODE_NARROW = 2;
// A NARROW launch's flag: FLAG_NARROW_CLEAN = every workgroup drained what it was given and nothing waits anywhere (no
// message is in flight after such a launch either): the fixed point is reached and the host needs no quiet launch to see
// it.  FLAG_NARROW_LEFT = some workgroup left states waiting (spilled, beyond the threshold, messages of the head).
constexpr uint32_t FLAG_NARROW_CLEAN = 1u + MODE_NARROW, FLAG_NARROW_LEFT = 2u + MODE_NARROW;
// NARROW launches
constexpr uint32_t NW_SEG = MB_B;       // work-list segment of a block (COLLECT hands over every state it would have expanded)
constexpr uint32_t NW_CAP = 1024;       // entries per level a workgroup keeps in LDS (two lists of 16 KB)
constexpr uint32_t NW_GROW = 384;       // head of the search (one workgroup): a level wider than this goes back to the WIDE sweeps
constexpr uint32_t NW_GROW_MANY = 48;   // ... when every workgroup follows a segment (a level then costs atomics chip-wide)
constexpr uint32_t NW_SMALL = 1024;     // near + far-waiting states below which a sweep hands over even while it grows
constexpr uint32_t NW_UNROLL = 4 * (1024 / MB_THREADS);       // entries a 16-lane group relaxes at once
constexpr uint32_t NW_MAX_LEVELS = 4096;
constexpr uint32_t NW_DEG_SAT = 0xFFFu;  // arc count field of an entry; saturated = read offsets[s + 1]

struct MboxView {
  const uint32_t* roff;    // [nb*nb + 1] region offsets, destination-major: region (i -> j) starts at roff[j*nb + i]
  const uint32_t* roff_t;  // [nb*nb]     roff_t[i*nb + j] = roff[j*nb + i] (what the SENDER i reads, contiguous)
  uint2* msgs[2];          // [E] each: messages of even / odd sweeps
  uint32_t* cnt[2];        // [nb*nb] each: cnt[j*nb + i] = messages in region (i -> j)
  uint32_t* wrote[2];      // [nb] each: sender i left non-zero counts in this parity's column
  uint32_t* pend;          // [nb * MB_B/32] states improved but not yet expanded (waiting beyond the threshold)
  uint32_t* blk_pend;      // [nb] number of such states per block
  uint32_t* blk_mind;      // [nb] min enc(d) among them (a lower bound after a NARROW launch)
  uint32_t* blk_far;       // [nb] how many of them are beyond the threshold
  uint4* wl;               // [nb * NW_SEG] work list of the NARROW launches: {state, arc begin, enc(d), hops << 12 | arcs}
  uint32_t* wl_cnt;        // [nb] entries in segment j
  uint32_t nb;
  uint32_t stg;            // staging slots per destination (MB_STG_MAX, fewer for many blocks)
  unsigned long long* dbg;  // tuning only (WFST_SSSP_MBOX_TRACE): wall-clock stamps [sweep][block][16], or null
};
constexpr uint32_t MB_DBG_SWEEPS = 64;
#define MB_STAMP(p) do { if (mb.dbg && tid == 0 && sweep < MB_DBG_SWEEPS) mb.dbg[((size_t)sweep * nb + j) * 16 + (p)] = wall_clock64(); } while (0)

// ---- plan (cached on the FST handle): region offsets from the number of arcs between every pair of blocks
// (LOG = log2 of the block size: 12 everywhere except the resident launches of graphs of 1M .. 2M states, which own 8192 states)
template <uint32_t LOG>
__global__ void __launch_bounds__(1024) mbox_hist_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn,
                                                         uint32_t n, uint32_t nb, uint32_t* __restrict__ hist) {
  constexpr uint32_t MB_LOG = LOG, MB_B = 1u << LOG;
  __shared__ uint32_t l_h[MB_NBMAX_BIG];
  const uint32_t j = blockIdx.x;
  for (uint32_t d = threadIdx.x; d < nb; d += blockDim.x) l_h[d] = 0;
  __syncthreads();
  const uint32_t s0 = j << MB_LOG, s1 = min(n, s0 + MB_B);
  const uint32_t b = offsets[s0], e = offsets[s1];
  for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) atomicAdd(&l_h[wn[i].y >> MB_LOG], 1u);
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < nb; d += blockDim.x) hist[(size_t)d * nb + j] = l_h[d];  // destination-major
}
// how many states have more than 2 l arcs, l = 2 .. 7 (over[l]), and how many arcs those rows hold beyond their first 2 l
// (over[8 + l], saturating): what the resident kernel's long-row pass would have to do with l lanes per state (MboxPlan::lps)
// ... and the sum of the states' CHEAPEST finite arc weights (minw[0], double) over the states that have one (minw[1]): what a
// shortest path pays per hop is nearer to that than to the mean arc weight (the band of the near-far schedule, relax_setup)
__global__ void __launch_bounds__(256) mbox_degree_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn, uint32_t n,
                                                          uint32_t* __restrict__ over, double* __restrict__ minw) {
  __shared__ uint32_t s_over[16];
  if (threadIdx.x < 16) s_over[threadIdx.x] = 0;
  __syncthreads();
  double msum = 0.0, mcnt = 0.0;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const uint32_t d = offsets[s + 1] - offsets[s];
    float mn = INF;
    for (uint32_t k = offsets[s]; k < offsets[s + 1]; ++k) mn = fminf(mn, __uint_as_float(wn[k].x));
    if (mn < INF && mn > -INF) {
      msum += (double)mn;
      mcnt += 1.0;
    }
    for (uint32_t l = 2; l < 8; ++l)
      if (d > 2u * l) {
        atomicAdd(&s_over[l], 1u);
        atomicAdd(&s_over[8 + l], d - 2u * l);
      }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    msum += __shfl_xor(msum, d);
    mcnt += __shfl_xor(mcnt, d);
  }
  if ((threadIdx.x & 63u) == 0 && mcnt > 0.0) {
    atomicAdd(&minw[0], msum);
    atomicAdd(&minw[1], mcnt);
  }
  __syncthreads();
  if (threadIdx.x >= 2 && threadIdx.x < 8) {
    if (s_over[threadIdx.x]) atomicAdd(&over[threadIdx.x], s_over[threadIdx.x]);
    if (s_over[8 + threadIdx.x]) {  // (a count of arcs: saturates instead of wrapping)
      const uint32_t old = atomicAdd(&over[8 + threadIdx.x], s_over[8 + threadIdx.x]);
      if (old + s_over[8 + threadIdx.x] < old) atomicMax(&over[8 + threadIdx.x], 0xFFFFFFFFu);
    }
  }
}
__global__ void mbox_transpose_kernel(const uint32_t* __restrict__ roff, uint32_t nb, uint32_t* __restrict__ roff_t) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nb * nb) return;
  const uint32_t i = k / nb, j = k % nb;
  roff_t[k] = roff[(size_t)j * nb + i];
}

// Initial state of a mailbox solve in one launch.  With NARROW launches the start state is the one entry of its block's
// work-list segment (sweep 0 is NARROW); without, it waits in the pending mask (sweep 0 is WIDE).
template <uint32_t LOG>
__global__ void __launch_bounds__(256) sssp_mbox_setup_kernel(uint64_t* __restrict__ key, MboxView mb, uint32_t* __restrict__ improved,
                                                              Ctl* __restrict__ ctl, const uint32_t* __restrict__ offsets, uint32_t n,
                                                              uint32_t start, float tau0, uint32_t narrow_on, uint2* rs_msgs0,
                                                              uint2* rs_msgs1, const uint32_t* __restrict__ rs_roffh,
                                                              uint32_t* __restrict__ rs_abort) {
  constexpr uint32_t MB_LOG = LOG, MB_B = 1u << LOG, NW_SEG = MB_B;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  const uint32_t nb = mb.nb;
  if (rs_msgs0) {  // resident launches (sssp_resident.h): no region header carries a tag of this solve yet
    for (uint32_t i = tid; i < nb * nb; i += nt) {
      const uint32_t h = rs_roffh[i];
      rs_msgs0[h].x = 0;
      rs_msgs1[h].x = 0;
    }
    if (tid == 0) rs_abort[0] = 0;
  }
  for (uint32_t i = tid; i < n; i += nt) {
    const bool is_start = i == start;
    key[i] = is_start ? (uint64_t)enc_f32(0.0f) << 32 : KEY_INF;
  }
  for (uint32_t i = tid; i < nb * nb; i += nt) {
    mb.cnt[0][i] = 0;
    mb.cnt[1][i] = 0;
  }
  const bool waits = narrow_on == 0;
  for (uint32_t i = tid; i < nb * (MB_B / 32); i += nt)
    mb.pend[i] = waits && i == (start >> 5) ? 1u << (start & 31u) : 0u;
  for (uint32_t i = tid; i < nb; i += nt) {
    const bool mine = i == (start >> MB_LOG);
    mb.wrote[0][i] = 0;
    mb.wrote[1][i] = 0;
    mb.blk_pend[i] = waits && mine ? 1u : 0u;
    mb.blk_mind[i] = waits && mine ? enc_f32(0.0f) : 0xFFFFFFFFu;
    mb.blk_far[i] = 0;
    mb.wl_cnt[i] = !waits && mine ? 1u : 0u;
    if (!waits && mine) {
      const uint32_t b = offsets[start], c = offsets[start + 1] - b;
      mb.wl[(size_t)i * NW_SEG] = make_uint4(start, b, enc_f32(0.0f), min(c, NW_DEG_SAT));
    }
  }
  for (uint32_t i = tid; i < IMP_RING; i += nt) improved[i] = 0;
  uint32_t* cw = (uint32_t*)ctl;
  constexpr uint32_t W_TAU0 = offsetof(Ctl, tau0) / 4, W_BEST = offsetof(Ctl, best) / 4;
  for (uint32_t i = tid; i < (uint32_t)(sizeof(Ctl) / 4); i += nt)
    cw[i] = i == W_TAU0 ? __float_as_uint(tau0) : (i == W_BEST || i == W_BEST + 1) ? 0xFFFFFFFFu : 0u;
}

// minimum over the 64 lanes of a wave (result in every lane): DPP row shifts / broadcasts, no LDS crossbar trips
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x111, 0xF, 0xF, false));  // row_shr:1
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x112, 0xF, 0xF, false));  // row_shr:2
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x114, 0xF, 0xF, false));  // row_shr:4
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x118, 0xF, 0xF, false));  // row_shr:8
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- the schedule of a launch: threshold (same rules as sweep_tau) and mode, from what the previous launches left in the
// control block.  Two halves so that the loads can be issued together with everything else a launch asks for first.
struct SchedRaw {
  uint2 mine;
};
struct Sched {
  float tau;      // the threshold on record for this launch (what the next launch continues from)
  float tau_use;  // the threshold this launch lists / follows by: +inf in the hand-over launches (see below)
  uint32_t streak, prev_near, far_total, mode;
};
// ONE vector load for every word of the schedule (separate loads in separate branches, or scalar loads issued after the
// vector ones have been waited for, each cost a dependent trip of their own): lanes 0..15 the counters of sweep-1 (near
// activations | states waiting beyond the threshold), 16..31 those of sweep-2, lane 32 / 33 / 34 the threshold, streak and
// mode of sweep-1.
__device__ __forceinline__ SchedRaw mbox_sched_load(const Ctl* ctl, uint32_t sweep) {
  SchedRaw r{make_uint2(0u, 0u)};
  if (sweep == 0) return r;
  const uint32_t lane = threadIdx.x & 63u, p = (sweep - 1) % RING;
  size_t off = offsetof(Ctl, tau0);  // (lanes without a word of their own read a harmless one)
  if (lane < NEAR_SHARDS) off = offsetof(Ctl, nf) + ((size_t)((sweep - 1) % NEAR_RING) * NEAR_SHARDS * NF_STRIDE + lane * NF_STRIDE) * 8;
  else if (lane < 2 * NEAR_SHARDS) off = offsetof(Ctl, nf) + ((size_t)((sweep + NEAR_RING - 2) % NEAR_RING) * NEAR_SHARDS * NF_STRIDE + (lane - NEAR_SHARDS) * NF_STRIDE) * 8;
  else if (lane == 32) off = offsetof(Ctl, tau) + (size_t)p * 4;
  else if (lane == 33) off = offsetof(Ctl, streak) + (size_t)p * 4;
  else if (lane == 34) off = offsetof(Ctl, mode) + (size_t)p * 4;
  r.mine = *(const uint2*)((const unsigned char*)ctl + off);  // (dword-aligned 8-byte loads are fine in global memory)
  return r;
}
__device__ __forceinline__ Sched mbox_sched_eval(const Ctl* ctl, const SchedRaw& r, uint32_t sweep, float delta, uint32_t near_low,
                                                 uint32_t narrow_t) {
  Sched s{0.0f, 0.0f, 0u, 0u, 0u, MODE_WIDE};
  if (sweep == 0) {
    s.tau = s.tau_use = ctl->tau0;
    s.mode = narrow_t ? MODE_NARROW : MODE_WIDE;
    return s;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t prev_tau = (uint32_t)__builtin_amdgcn_readlane((int)r.mine.x, 32);
  const uint32_t prev_streak = (uint32_t)__builtin_amdgcn_readlane((int)r.mine.x, 33);
  const uint32_t prev_mode = (uint32_t)__builtin_amdgcn_readlane((int)r.mine.x, 34);
  const bool counts = lane < NEAR_SHARDS || (lane < 2 * NEAR_SHARDS && sweep >= 2);
  uint32_t near = counts ? r.mine.x : 0u, far = counts ? r.mine.y : 0u;
  // sums inside each row of 16 lanes (DPP row shifts: lane 15 of a row ends up with the row's total; no LDS trips)
  near += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)near, 0x111, 0xF, 0xF, false);  // row_shr:1
  far += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)far, 0x111, 0xF, 0xF, false);
  near += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)near, 0x112, 0xF, 0xF, false);  // row_shr:2
  far += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)far, 0x112, 0xF, 0xF, false);
  near += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)near, 0x114, 0xF, 0xF, false);  // row_shr:4
  far += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)far, 0x114, 0xF, 0xF, false);
  near += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)near, 0x118, 0xF, 0xF, false);  // row_shr:8
  far += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)far, 0x118, 0xF, 0xF, false);
  const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)near, 15), before = (uint32_t)__builtin_amdgcn_readlane((int)near, 31);
  s.far_total = (uint32_t)__builtin_amdgcn_readlane((int)far, 15);
  s.prev_near = cnt;
  const float prev = __uint_as_float(prev_tau);
  if (prev_mode == MODE_COLLECT) {
    // The hand-over launches (COLLECT, then this NARROW one) ignore the threshold: what still waits beyond it is at most
    // narrow_t states, listed and followed with everything else, so that ONE narrow launch drains the search instead of a
    // WIDE / COLLECT / NARROW round per remaining band.  The threshold on record stays where it was: if the frontier
    // grows again, the WIDE sweeps continue with the bands.
    s.tau = prev;
    s.tau_use = INF;
    s.mode = MODE_NARROW;
    return s;
  }
  // hand over when the near set is small and shrinking (the tail of the last band: whatever still waits beyond the
  // threshold is counted in, so the growing head of a band never qualifies), or tiny either way — but never in the two
  // levels behind the head of the search: a head that outgrew its workgroup posts near_low as its near count (to keep the
  // threshold where it is), the few hundred activations of the first WIDE level then look like a shrinking tail, and the
  // NARROW launch that follows grows straight back into the WIDE levels (27 of 96 random sources of the benchmark's transducer
  // took five launches instead of three: tools/varied_sources.py)
  if (narrow_t && sweep >= 3u && prev_mode == MODE_WIDE && cnt != 0u && cnt + s.far_total <= narrow_t &&
      (cnt < before || cnt + s.far_total <= min(narrow_t, NW_SMALL)))
    s.mode = MODE_COLLECT;
  if (cnt >= near_low) {
    s.tau = prev;
  } else if (cnt) {
    // a near set that cannot fill the GPU AND is shrinking (the tail of a band, not its growing head):
    // widen the band by delta and keep relaxing
    s.tau = cnt < before ? prev + delta : prev;
  } else {
    const uint32_t st = min(prev_streak + 1u, 30u);
    s.streak = st;
    s.tau = prev + delta * (float)(1u << (st - 1u));
  }
  s.tau_use = s.mode == MODE_COLLECT ? INF : s.tau;
  return s;
}

// A state improved by a NARROW launch that it does not follow itself: it waits in the masks for the next WIDE sweep.
// Nothing is read back (a returned atomic would be a third dependent trip per level): the block's far count may count a
// state twice — it only steers the schedule, and the owner recomputes it the next time it runs — and "somebody waits" is
// blk_mind != +inf (the owner keeps blk_pend exact for itself).
template <uint32_t LOG>
__device__ __forceinline__ void mbox_make_wait(const MboxView& mb, uint32_t t, uint32_t enc_d, bool is_far) {
  const uint32_t b = t >> LOG;
  atomicOr(&mb.pend[t >> 5], 1u << (t & 31u));
  atomicMin(&mb.blk_mind[b], enc_d);
  if (is_far) atomicAdd(&mb.blk_far[b], 1u);
}

// One NARROW launch of workgroup j: follows the entries of its segment, and what they improve, until nothing near is
// left, the frontier has grown beyond what one workgroup should carry, or (while states wait beyond the threshold) it
// has started to shrink — the rule by which sweep_tau widens the band.  `wl` = 2 x NW_CAP entries of LDS.
template <uint32_t LOG>
__device__ __forceinline__ void mbox_narrow(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn,
                                            uint64_t* __restrict__ key, const MboxView& mb, Ctl* __restrict__ ctl,
                                            uint32_t* __restrict__ improved, uint32_t sweep, float tau, uint32_t far_total,
                                            uint32_t near_low, uint32_t profile, uint32_t wl_n, bool waits, uint32_t bfar,
                                            uint4* wl, uint32_t* s_n /*[8]: list sizes [0..2] (level mod 3), found beyond the threshold [3], left waiting [4]*/,
                                            uint32_t par_out, const uint32_t* l_roff_out, uint32_t* l_cur, uint32_t* l_cap) {
  constexpr uint32_t MB_LOG = LOG, MB_B = 1u << LOG, MB_HOP_BITS = 32 - LOG, NW_SEG = MB_B;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, j = blockIdx.x, nb = mb.nb;
  const uint32_t sub = tid & 15u, grp = tid >> 4;
  // The head of the search (sweep 0, ONE workgroup at work): a candidate beyond the threshold is not this launch's
  // business — it leaves as a MESSAGE to the target's owner, like in a WIDE sweep (region (j -> d): nobody else writes
  // there in this launch), instead of a global atomicMin plus three more atomics to make it wait.  Nine candidates in ten
  // of the head are such.  A region only has room for the arcs from block j to block d; what does not fit takes the
  // atomic path.
  const bool head = sweep == 0;
  uint2* __restrict__ msgs_out = mb.msgs[par_out];
  if (head && wl_n)
    for (uint32_t d = tid; d < nb; d += MB_THREADS) l_cap[d] = mb.roff[(size_t)d * nb + j + 1] - mb.roff[(size_t)d * nb + j];
  unsigned long long* const nf = &ctl->nf[sweep % NEAR_RING][(j % NEAR_SHARDS) * NF_STRIDE];
  for (uint32_t e = tid; e < wl_n; e += MB_THREADS) {  // (a segment longer than the list: the rest keeps waiting)
    const uint4 en = mb.wl[(size_t)j * NW_SEG + e];
    if (e < NW_CAP) wl[e] = en;
    else mbox_make_wait<LOG>(mb, en.x, en.z, false);
  }
  if (tid == 0) {
    s_n[0] = min(wl_n, NW_CAP);
    s_n[1] = 0;
    s_n[2] = 0;
    s_n[3] = 0;
    s_n[4] = (waits || wl_n > NW_CAP) ? 1u : 0u;  // something is left waiting after this launch
    if (wl_n) mb.wl_cnt[j] = 0;
  }
  if (wl_n == 0) {  // (uniform) nothing to follow; whoever waits in this block's masks waits for a WIDE sweep
    if (tid == 0) {
      if (bfar) atomicAdd(nf, (unsigned long long)bfar << 32);
      if (waits) atomicMax(improved, FLAG_NARROW_LEFT);
    }
    return;
  }
  __syncthreads();
  // (the two lists alternate; their COUNTERS rotate through three words, so that the one a level will fill next can be zeroed a
  // level ahead and a level needs ONE barrier: a hop of the tail is ~2 us, two trips and this)
  uint32_t cur = 0, cnt_i = 0, prev_n = 0, prev2_n = 0, far_new = 0;
  bool left_any = false;
  unsigned long long p_arcs = 0, p_states = 0;
  bool grew = false;
  for (uint32_t level = 0;; ++level) {
    const uint32_t n_lv = min(s_n[cnt_i], NW_CAP);
    const uint32_t far_seen = s_n[3];
    const uint32_t cnt_o = cnt_i == 2u ? 0u : cnt_i + 1u, cnt_z = cnt_o == 2u ? 0u : cnt_o + 1u;
    if (tid == 0) s_n[cnt_z] = 0;  // (last level's size: everybody read it before that level's barrier; next level's output)
#ifdef WFST_NW_TRACE
    if (mb.dbg && tid == 0 && sweep < MB_DBG_SWEEPS && level < 13) mb.dbg[((size_t)sweep * nb + j) * 16 + 2 + level] = (wall_clock64() << 12) | min(n_lv, 4095u);
#endif
    if (n_lv == 0) break;
    // (8192-state blocks: a workgroup's share of the hand-over is twice as long — with the 4096-state limit the tail of a 1M-state
    // solve on half the device bounced back to the WIDE levels twice: 385 -> 350 us, 5 -> 3 launches; 2M states unchanged)
    if (n_lv > (sweep == 0 ? NW_GROW : (LOG == 13 ? 2u * NW_GROW_MANY : NW_GROW_MANY))) {
      grew = true;
      break;
    }
    // the head of the search is one workgroup: it applies sweep_tau's rule itself — a near set that shrinks while states
    // wait beyond the threshold is the tail of the band: widen instead.  (Later NARROW launches follow everything they
    // were given, whatever its distance: tau is +inf there.)
    if (sweep == 0 && level >= 2 && prev_n < prev2_n && (far_total | far_seen) != 0u) break;
    if (level >= NW_MAX_LEVELS) {
      grew = true;  // (keeps the threshold where it is: the rest is followed after the next WIDE sweep)
      break;
    }
    const uint4* __restrict__ in = wl + cur * NW_CAP;
    uint4* __restrict__ out = wl + (cur ^ 1u) * NW_CAP;
    uint32_t* n_out = &s_n[cnt_o];
    for (uint32_t r0 = 0; r0 < n_lv; r0 += (MB_THREADS / 16) * NW_UNROLL) {
      uint32_t i_[NW_UNROLL], end_[NW_UNROLL], h1_[NW_UNROLL];
      float d_[NW_UNROLL];
      bool more = false;
      for (uint32_t u = 0; u < NW_UNROLL; ++u) {
        const uint32_t e = r0 + grp + (MB_THREADS / 16) * u;
        i_[u] = end_[u] = h1_[u] = 0;
        d_[u] = 0.0f;
        if (e < n_lv) {
          const uint4 en = in[e];
          uint32_t c = en.w & NW_DEG_SAT;
          if (c == NW_DEG_SAT) c = offsets[en.x + 1] - en.y;
          i_[u] = en.y + sub;
          end_[u] = en.y + c;
          d_[u] = dec_f32(en.z);
          h1_[u] = (en.w >> MB_LOG) + 1u;
          if (profile && sub == 0) {
            p_arcs += c;
            p_states += 1;
          }
        }
        more |= i_[u] < end_[u];
      }
      more = __any(more);
      while (more) {
        uint2 a[NW_UNROLL];
        bool v[NW_UNROLL];
        for (uint32_t u = 0; u < NW_UNROLL; ++u) {
          v[u] = i_[u] < end_[u];
          a[u] = make_uint2(0x7F800000u, 0u);
          if (v[u]) a[u] = wn[i_[u]];
        }
        uint32_t enc[NW_UNROLL], tb[NW_UNROLL], te[NW_UNROLL];
        unsigned long long ck[NW_UNROLL], old[NW_UNROLL];
        for (uint32_t u = 0; u < NW_UNROLL; ++u) {
          const float c = (d_[u] + __uint_as_float(a[u].x)) + 0.0f;  // w1 (x) w2 = f32 add (tropical_weight.rs:60-70)
          v[u] = v[u] && c < INF;                                    // +inf never improves (shortest_path.rs:226)
          enc[u] = enc_f32(c);
          ck[u] = ((unsigned long long)enc[u] << 32) | h1_[u];
          old[u] = 0;
          tb[u] = te[u] = 0;
          if (head && v[u] && c > tau) {
            const uint32_t db = a[u].y >> MB_LOG, slot = atomicAdd(&l_cur[db], 1u);
            if (slot < l_cap[db]) {
              msgs_out[l_roff_out[db] + slot] = make_uint2((h1_[u] << MB_LOG) | (a[u].y & (MB_B - 1u)), enc[u]);
              if (h1_[u] >> MB_HOP_BITS) ctl->pad = 1u;
              far_new += 1u;
              left_any = true;
              v[u] = false;
            }
          }
          if (v[u]) {  // the atomic and the target's arc range travel together: the entry it may become needs no further trip
            old[u] = atomicMin((unsigned long long*)&key[a[u].y], ck[u]);
            tb[u] = offsets[a[u].y];
            te[u] = offsets[a[u].y + 1];
          }
        }
        more = false;
        for (uint32_t u = 0; u < NW_UNROLL; ++u) {
          const bool won = v[u] && ck[u] < old[u];
          if (won) {
            if (h1_[u] >> MB_HOP_BITS) ctl->pad = 1u;  // hop count beyond the message format: the host refuses the result
            bool listed = false;
            if (dec_f32(enc[u]) <= tau) {
              const uint32_t slot = atomicAdd(n_out, 1u);
              if (slot < NW_CAP) {
                out[slot] = make_uint4(a[u].y, tb[u], enc[u], (h1_[u] << MB_LOG) | min(te[u] - tb[u], NW_DEG_SAT));
                listed = true;
              }
              if (!listed) {
                mbox_make_wait<LOG>(mb, a[u].y, enc[u], false);
                left_any = true;
              }
            } else {
              mbox_make_wait<LOG>(mb, a[u].y, enc[u], true);
              far_new += 1u;
              left_any = true;
            }
          }
          i_[u] += 16;
          more |= i_[u] < end_[u];
        }
        more = __any(more);
      }
    }
    // what this level found beyond the threshold, for the shrink rule of the next levels
    if (__any(far_new != 0u)) {  // (never in the tail: tau is +inf there)
      uint32_t f = far_new;
      for (int d = 32; d >= 1; d >>= 1) f += __shfl_xor(f, d);
      if (lane == 0 && f) atomicAdd(&s_n[3], f);
      far_new = 0;
    }
    __syncthreads();  // the next list is complete; the current one is free
    prev2_n = prev_n;
    prev_n = n_lv;
    cur ^= 1u;
    cnt_i = cnt_o;
  }
  // whatever is still listed waits for the next WIDE sweep (near states: it expands them at once)
  {
    const uint32_t left = min(s_n[cnt_i], NW_CAP);
    const uint4* __restrict__ in = wl + cur * NW_CAP;
    for (uint32_t e = tid; e < left; e += MB_THREADS) mbox_make_wait<LOG>(mb, in[e].x, in[e].z, false);
    if (__any(left_any || left != 0u) && lane == 0) s_n[4] = 1u;
    __syncthreads();
    if (tid == 0) atomicMax(improved, s_n[4] ? FLAG_NARROW_LEFT : FLAG_NARROW_CLEAN);
    if (head) {  // the messages of the head: counts of the regions, like a WIDE sweep's publish
      bool any = false;
      for (uint32_t d = tid; d < nb; d += MB_THREADS) {
        const uint32_t c = min(l_cur[d], l_cap[d]);
        mb.cnt[par_out][(size_t)d * nb + j] = c;
        any |= c != 0;
      }
      if (__any(any) && lane == 0) mb.wrote[par_out][j] = 1u;
    }
    // a frontier that outgrew the workgroup keeps the threshold where it is (>= near_low activations: sweep_tau's rule);
    // otherwise the next sweep sees no near activations and widens the band.  High word: the states waiting beyond the
    // threshold in this block before the launch, and those this workgroup has made wait anywhere.
    if (tid == 0) {
      const unsigned long long add = ((unsigned long long)(bfar + s_n[3]) << 32) | (grew ? near_low : 0u);
      if (add) atomicAdd(nf, add);
    }
  }
  if (profile) {
    for (int d = 32; d >= 1; d >>= 1) {
      p_arcs += __shfl_xor(p_arcs, d);
      p_states += __shfl_xor(p_states, d);
    }
    if (lane == 0 && p_states) {
      atomicAdd(&ctl->arcs[(j % PROF_SHARDS) * PROF_STRIDE], p_arcs);
      atomicAdd(&ctl->states[(j % PROF_SHARDS) * PROF_STRIDE], p_states);
    }
  }
}

// One mailbox launch.  Workgroup j, WIDE mode:
//   trip 1  counts of its inbox regions, pending words, the schedule words, AND (speculatively, unless the host hints
//           that this slot is probably not WIDE or idle: `hint`) the block's keys and offsets
//           -> nothing arrives and nobody waiting is near: leave
//   trip 2  the messages; candidates applied with LDS atomicMin
//   scan    states whose key changed or that were waiting: written back; near ones (d <= tau) listed, far ones wait
//   trip 3  the arc rows of the listed states: candidate per arc -> LDS (same block) or the destination's staging slots
//   flush   staged messages leave as contiguous runs; counts of the regions written; the new waiting set
// The kernel is bound by its chain of dependent round trips (global AND LDS), so every phase asks for all it needs
// at once: no prefix sums, no searches, no shuffle reductions (ballots and one LDS atomic per wave instead).
// BIG: more than 256 blocks — inbox counts staged through LDS, regions visited in passes of 256.
template <bool BIG>
__global__ void __launch_bounds__(MB_THREADS) sssp_mbox_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn,
                                                               uint64_t* __restrict__ key, MboxView mb, uint32_t par_in, uint32_t n,
                                                               uint32_t* __restrict__ improved_ring, Ctl* __restrict__ ctl,
                                                               uint32_t sweep, float delta, uint32_t near_low, uint32_t profile,
                                                               uint32_t hint, uint32_t narrow_t) {
  extern __shared__ __align__(16) unsigned char mb_dyn[];
  __shared__ unsigned long long lkey[MB_B];
  __shared__ uint32_t l_off[MB_B + 1];
  __shared__ uint16_t a_state[MB_B];  // states expanded in this sweep
  __shared__ uint32_t s_wany[MB_THREADS / 64];
  __shared__ uint32_t s_an, s_sent, s_npend, s_mind, s_nfar;
  __shared__ uint32_t s_nw[8];
  __shared__ unsigned long long s_prof_arcs;
  constexpr uint32_t R = MB_B / MB_THREADS;  // states per thread
  constexpr uint32_t PW = MB_B / 32;         // pending words per block
  constexpr uint32_t WPR = MB_THREADS / 32;  // pending words between two states of one thread

  // The kernel-argument segment is four 64-byte lines, and the compiler fetches arguments where it first needs them: four
  // scalar-cache misses one after the other (~0.3 us each) in front of the prologue's loads.  One word of every line is
  // asked for here, together; the later fetches hit.
  asm volatile("" ::"s"(offsets), "s"(mb.cnt[1]), "s"(ctl), "s"(narrow_t));
  // `sweep` is the absolute sweep index: the host knows it (plain launches), which saves the trip to ctl->base
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t j = blockIdx.x, nb = mb.nb, stg = mb.stg;
  uint2* const l_stage = (uint2*)mb_dyn;                         // [nb * stg]
  uint32_t* const l_roff_out = (uint32_t*)(l_stage + nb * stg);  // [nb]
  uint32_t* const l_cur = l_roff_out + nb;                       // [nb]
  uint32_t* const l_base = l_cur + nb;                           // [nb]
  uint32_t* const l_cin = (uint32_t*)l_stage;                    // BIG: inbox counts / region offsets until the expansion
  uint32_t* const l_rin = l_cin + nb;
  const uint32_t par_out = par_in ^ 1u;
  const uint32_t s0 = j << MB_LOG;
  uint32_t* improved = improved_ring + (sweep % IMP_RING);
  MB_STAMP(0);

  // ---- trip 1: every load of the prologue is ISSUED before anything is consumed (an LDS store of a loaded value waits
  // for it, and loads come back in order: one such store in front of the schedule words makes the prologue two trips).
  // Inbox region i is read by threads 4i .. 4i+3 (BIG: in passes of 256 regions, counts staged through LDS).
  const uint32_t reg = tid / MB_LPR, q = tid % MB_LPR;
  constexpr uint32_t NBR = BIG ? MB_NBMAX_BIG / MB_THREADS : 1;  // per-destination table entries per thread
  uint32_t c_in = 0, rb_in = 0;
  uint32_t cin_r[NBR], rin_r[NBR], ro_r[NBR];
  if (!BIG) {
    if (reg < nb) {
      c_in = mb.cnt[par_in][j * nb + reg];
      rb_in = mb.roff[j * nb + reg];
    }
  }
  for (uint32_t r = 0; r < NBR; ++r) {
    const uint32_t d = tid + MB_THREADS * r;
    cin_r[r] = rin_r[r] = ro_r[r] = 0;
    if (d < nb) {
      if (BIG) {
        cin_r[r] = mb.cnt[par_in][(size_t)j * nb + d];
        rin_r[r] = mb.roff[(size_t)j * nb + d];
      }
      ro_r[r] = mb.roff_t[(size_t)j * nb + d];
    }
  }
  uint32_t pw[R];  // pending words of this thread's states (state tl = tid + 1024 r sits in word (tid >> 5) + 32 r)
  for (uint32_t r = 0; r < R; ++r) pw[r] = mb.pend[j * PW + (tid >> 5) + WPR * r];
  // (nothing loaded here is kept for the end of the kernel: a loaded value that has to be spilled is waited for on the
  // spot, in the middle of the prologue's loads — the waiting set and its statistics are stored unconditionally instead)
  const uint32_t bmind = mb.blk_mind[j], bfar = mb.blk_far[j];
  const bool wrote_out = mb.wrote[par_out][j] != 0u;
  const uint32_t wl_n = min(mb.wl_cnt[j], NW_SEG);
  unsigned long long kreg[R];
  uint32_t oreg[R], o_last = 0;
  auto bulk_load = [&]() {
    for (uint32_t r = 0; r < R; ++r) {
      const uint32_t s = s0 + tid + MB_THREADS * r;
      kreg[r] = KEY_INF;
      oreg[r] = 0;
      if (s < n) {
        kreg[r] = key[s];
        oreg[r] = offsets[s];
      } else if (s == n) {
        oreg[r] = offsets[n];
      }
    }
    // (the block's end offset: asked for by the LAST thread through its own index — a uniform address under `tid == 0`
    // becomes a scalar load that the first wave waits for on the spot, before it has issued the rest of its prologue)
    if (tid == MB_THREADS - 1 && s0 + MB_B <= n) {
      uint32_t t = tid;
      asm volatile("" : "+v"(t));  // (opaque: keeps the address in a vector register)
      o_last = offsets[s0 + t + MB_THREADS * (R - 1) + 1];
    }
  };
  auto bulk_store = [&]() {
    for (uint32_t r = 0; r < R; ++r) {
      l_off[tid + MB_THREADS * r] = oreg[r];
      lkey[tid + MB_THREADS * r] = kreg[r];
    }
    if (tid == MB_THREADS - 1) l_off[MB_B] = o_last;
  };
  if (!hint) bulk_load();
  // every wave works the schedule out for itself (the same few words: one trip, no LDS hand-over); its load goes last,
  // so the LDS stores below only wait for what was asked for before it
  const SchedRaw raw = mbox_sched_load(ctl, sweep);
  bool my_any = c_in != 0;
  for (uint32_t r = 0; r < NBR; ++r) {
    const uint32_t d = tid + MB_THREADS * r;
    if (d < nb) {
      if (BIG) {
        l_cin[d] = cin_r[r];
        l_rin[d] = rin_r[r];
        my_any |= cin_r[r] != 0;
      }
      l_roff_out[d] = ro_r[r];
      l_cur[d] = 0;
      l_base[d] = 0;
    }
  }
  if (!hint) bulk_store();
  const Sched sc = mbox_sched_eval(ctl, raw, sweep, delta, near_low, narrow_t);
  const float tau = sc.tau_use;
  const uint32_t mode = sc.mode;
  unsigned long long* const nf = &ctl->nf[sweep % NEAR_RING][(j % NEAR_SHARDS) * NF_STRIDE];
  if (tid < 64) {
    if (tid == 0) {
      s_an = 0;
      s_sent = 0;
      s_npend = 0;
      s_nfar = 0;
      s_mind = 0xFFFFFFFFu;
      s_prof_arcs = 0;
    }
    if (j == 0) {
      if (tid == 0) {
        ctl->tau[sweep % RING] = __float_as_uint(sc.tau);
        ctl->streak[sweep % RING] = sc.streak;
        ctl->mode[sweep % RING] = mode;
      }
      if (tid < NEAR_SHARDS) ctl->nf[(sweep + 1) % NEAR_RING][tid * NF_STRIDE] = 0;  // recycle
    }
  }
  {
    const bool wany = __ballot(my_any) != 0;
    if (lane == 0) s_wany[tid >> 6] = wany ? 1u : 0u;
  }
  __syncthreads();
  bool any_in = false;
  for (uint32_t w = 0; w < MB_THREADS / 64; w += 4) {
    const uint4 f = *(const uint4*)&s_wany[w];
    any_in |= (f.x | f.y | f.z | f.w) != 0;
  }
  MB_STAMP(1);
  if (mode == MODE_NARROW) {
    // (no messages are in flight: the launch before this one was a COLLECT, or this is sweep 0)
    if (wrote_out) {  // counts this workgroup published two sweeps ago are still in the column it writes now
      for (uint32_t d = tid; d < nb; d += MB_THREADS) mb.cnt[par_out][(size_t)d * nb + j] = 0;
      if (tid == 0) mb.wrote[par_out][j] = 0;
    }
    mbox_narrow<MB_LOG>(offsets, wn, key, mb, ctl, improved, sweep, tau, sc.far_total, near_low, profile, wl_n, bmind != 0xFFFFFFFFu, bfar,
                (uint4*)lkey, s_nw, par_out, l_roff_out, l_cur, l_base);
    MB_STAMP(15);
    return;
  }
  const bool waiting = bmind != 0xFFFFFFFFu;  // (blk_mind: the least distance among the block's waiting states)
  if (!any_in && (!waiting || dec_f32(bmind) > tau)) {
    // nothing arrives and nobody who waits is near: the block sleeps through this sweep
    if (wrote_out) {
      for (uint32_t d = tid; d < nb; d += MB_THREADS) mb.cnt[par_out][(size_t)d * nb + j] = 0;
      if (tid == 0) mb.wrote[par_out][j] = 0;
    }
    if (tid == 0) {
      if (waiting && *improved == 0u) *improved = 1u + mode;  // the solve is not over
      if (bfar) atomicAdd(nf, (unsigned long long)bfar << 32);  // its waiting states still count in the schedule
    }
    MB_STAMP(15);
    return;
  }
  if (hint) {  // gated launch: the block is awake, now it asks for its keys and offsets
    bulk_load();
    bulk_store();
    __syncthreads();
  }
  const bool collect = mode == MODE_COLLECT;

  // ---- trip 2: the messages
  const uint2* __restrict__ msgs_in = mb.msgs[par_in];
  constexpr uint32_t MU = 8;  // messages a thread requests at once (a region of up to 32 messages is one trip)
  if (!BIG) {
    for (uint32_t k0 = q; k0 < c_in; k0 += MB_LPR * MU) {
      uint2 m[MU];
      for (uint32_t u = 0; u < MU; ++u) {
        m[u] = make_uint2(0u, 0u);
        if (k0 + MB_LPR * u < c_in) m[u] = msgs_in[rb_in + k0 + MB_LPR * u];
      }
      for (uint32_t u = 0; u < MU; ++u)
        if (k0 + MB_LPR * u < c_in)
          atomicMin(&lkey[m[u].x & (MB_B - 1u)], ((unsigned long long)m[u].y << 32) | (m[u].x >> MB_LOG));
    }
  } else {
    for (uint32_t rg = reg; rg < nb; rg += MB_THREADS / MB_LPR) {
      const uint32_t c = l_cin[rg], rb = l_rin[rg];
      for (uint32_t k0 = q; k0 < c; k0 += MB_LPR * MU) {
        uint2 m[MU];
        for (uint32_t u = 0; u < MU; ++u) {
          m[u] = make_uint2(0u, 0u);
          if (k0 + MB_LPR * u < c) m[u] = msgs_in[rb + k0 + MB_LPR * u];
        }
        for (uint32_t u = 0; u < MU; ++u)
          if (k0 + MB_LPR * u < c)
            atomicMin(&lkey[m[u].x & (MB_B - 1u)], ((unsigned long long)m[u].y << 32) | (m[u].x >> MB_LOG));
      }
    }
  }
  __syncthreads();
  MB_STAMP(3);

  // ---- the states whose key changed or that were waiting: write back, list the near ones, keep the far ones waiting
  uint32_t far_w[R];  // the 32-state word of far (still waiting) states this lane belongs to
  uint32_t my_mind = 0xFFFFFFFFu;
  unsigned long long kn[R];  // keys after the inbox (what a later change from inside the block is measured against)
  {
    for (uint32_t r = 0; r < R; ++r) kn[r] = lkey[tid + MB_THREADS * r];
    bool near_[R];
    uint32_t n_near = 0;
    for (uint32_t r = 0; r < R; ++r) {
      const uint32_t tl = tid + MB_THREADS * r, s = s0 + tl;
      const bool chg = kn[r] != kreg[r];
      const bool act = chg || ((pw[r] >> (tl & 31u)) & 1u) != 0;
      if (chg) key[s] = kn[r];
      const uint32_t ed = (uint32_t)(kn[r] >> 32);
      near_[r] = act && dec_f32(ed) <= tau;
      const bool far = act && !near_[r];
      if (far) my_mind = min(my_mind, ed);
      const unsigned long long fm = __ballot(far), nm = __ballot(near_[r]);
      far_w[r] = (lane & 32u) ? (uint32_t)(fm >> 32) : (uint32_t)fm;
      n_near += (uint32_t)__popcll(nm);
    }
    uint32_t base = 0;
    if (n_near) {
      if (lane == 0) base = atomicAdd(&s_an, n_near);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    }
    for (uint32_t r = 0; r < R; ++r) {
      const unsigned long long nm = __ballot(near_[r]);
      if (near_[r]) a_state[base + (uint32_t)__popcll(nm & ((1ull << lane) - 1ull))] = (uint16_t)(tid + MB_THREADS * r);
      base += (uint32_t)__popcll(nm);
    }
  }
  __syncthreads();
  MB_STAMP(4);

  const uint32_t an = s_an;
  uint32_t sent = 0;  // wave-uniform
  unsigned long long p_arcs = 0;
  if (collect) {
    // ---- hand-over: the states this sweep would have expanded become the block's segment of the work list
    for (uint32_t e = tid; e < an; e += MB_THREADS) {
      const uint32_t tl = a_state[e];
      const unsigned long long k = lkey[tl];
      const uint32_t b = l_off[tl], c = l_off[tl + 1] - b;
      mb.wl[(size_t)j * NW_SEG + e] = make_uint4(s0 + tl, b, (uint32_t)(k >> 32), ((uint32_t)k << MB_LOG) | min(c, NW_DEG_SAT));
      if ((uint32_t)k >> MB_HOP_BITS) ctl->pad = 1u;
    }
    if (tid == 0) mb.wl_cnt[j] = an;
  } else {
    // ---- expansion: 16 lanes per state, MB_UNROLL states per group in flight; one flush of the staging slots per round
    const uint32_t sub = tid & 15u, grp = tid >> 4;  // 64 groups
    uint2* __restrict__ msgs_out = mb.msgs[par_out];
    constexpr uint32_t ROUND = (MB_THREADS / 16) * MB_UNROLL;  // states per round
    for (uint32_t r0 = 0; r0 < an; r0 += ROUND) {
      uint32_t tl_[MB_UNROLL];
      for (uint32_t u = 0; u < MB_UNROLL; ++u) {
        const uint32_t e = r0 + grp + (MB_THREADS / 16) * u;
        tl_[u] = e < an ? (uint32_t)a_state[e] : 0xFFFFFFFFu;
      }
      uint32_t i_[MB_UNROLL], end_[MB_UNROLL], h1_[MB_UNROLL];
      float d_[MB_UNROLL];
      bool more = false;
      for (uint32_t u = 0; u < MB_UNROLL; ++u) {
        i_[u] = end_[u] = h1_[u] = 0;
        d_[u] = 0.0f;
        if (tl_[u] != 0xFFFFFFFFu) {
          const unsigned long long k = lkey[tl_[u]];
          const uint32_t b = l_off[tl_[u]];
          end_[u] = l_off[tl_[u] + 1];
          d_[u] = dec_f32((uint32_t)(k >> 32));
          h1_[u] = (uint32_t)k + 1u;
          if (profile && sub == 0) p_arcs += end_[u] - b;
          i_[u] = b + sub;
        }
        more |= i_[u] < end_[u];
      }
      more = __any(more);
      while (more) {
        uint2 a[MB_UNROLL];
        bool v[MB_UNROLL];
        for (uint32_t u = 0; u < MB_UNROLL; ++u) {
          v[u] = i_[u] < end_[u];
          a[u] = make_uint2(0x7F800000u, 0u);
          if (v[u]) a[u] = wn[i_[u]];
        }
        // candidates: same-block targets never leave LDS; the others take a slot of their destination's region
        uint32_t enc[MB_UNROLL], slot[MB_UNROLL];
        for (uint32_t u = 0; u < MB_UNROLL; ++u) {
          const float c = (d_[u] + __uint_as_float(a[u].x)) + 0.0f;  // w1 (x) w2 = f32 add (tropical_weight.rs:60-70)
          v[u] = v[u] && c < INF;                                    // +inf never improves (shortest_path.rs:226)
          enc[u] = enc_f32(c);
          if (v[u] && (a[u].y >> MB_LOG) == j) {
            atomicMin(&lkey[a[u].y & (MB_B - 1u)], ((unsigned long long)enc[u] << 32) | h1_[u]);
            v[u] = false;
          }
        }
        for (uint32_t u = 0; u < MB_UNROLL; ++u) {
          slot[u] = 0;
          if (v[u]) slot[u] = atomicAdd(&l_cur[a[u].y >> MB_LOG], 1u);
        }
        more = false;
        for (uint32_t u = 0; u < MB_UNROLL; ++u) {
          if (v[u]) {
            const uint2 msg = make_uint2((h1_[u] << MB_LOG) | (a[u].y & (MB_B - 1u)), enc[u]);
            const uint32_t db = a[u].y >> MB_LOG, rel = slot[u] - l_base[db];
            if (rel < stg) l_stage[db * stg + rel] = msg;
            else msgs_out[l_roff_out[db] + slot[u]] = msg;
            if (h1_[u] >> MB_HOP_BITS) ctl->pad = 1u;  // hop count beyond the message format: the host refuses the result
          }
          sent += (uint32_t)__popcll(__ballot(v[u]));
          i_[u] += 16;
          more |= i_[u] < end_[u];
        }
        more = __any(more);  // the counter above is wave-uniform: the whole wave stays in the loop
      }
      MB_STAMP(12);
      __syncthreads();
      MB_STAMP(13);
      // flush: destination d's staged messages leave as one contiguous run (4 lanes per destination)
      for (uint32_t rg = reg; rg < nb; rg += MB_THREADS / MB_LPR) {
        const uint32_t b0 = l_base[rg], cnt = min(l_cur[rg] - b0, stg), ro = l_roff_out[rg] + b0;
        for (uint32_t k = q; k < cnt; k += MB_LPR) msgs_out[ro + k] = l_stage[rg * stg + k];
      }
      if (r0 + ROUND < an) {  // another round: its messages are staged from the current cursors on
        __syncthreads();
        for (uint32_t d = tid; d < nb; d += MB_THREADS) l_base[d] = l_cur[d];
        __syncthreads();
      }
    }
  }
  // ---- publish, part 1: the region counts.  The cursors are final (barrier of the last round), and an entry only needs a
  //      store when it is non-zero or when the column still holds this workgroup's counts of two sweeps ago: issued now,
  //      the stores are on their way while the waiting set is worked out below.
  for (uint32_t d = tid; d < nb; d += MB_THREADS) {
    const uint32_t c = l_cur[d];
    if (c != 0u || wrote_out) mb.cnt[par_out][(size_t)d * nb + j] = c;
  }
  if (profile)
    for (int d = 32; d >= 1; d >>= 1) p_arcs += __shfl_xor(p_arcs, d);
  if (lane == 0) {
    if (sent) atomicAdd(&s_sent, sent);
    if (profile && p_arcs) atomicAdd(&s_prof_arcs, p_arcs);
  }
  MB_STAMP(5);

  // ---- states improved from inside the block during the expansion: written back; they wait (expanded next sweep).
  //      The new waiting set.
  {
    uint32_t n_pend = 0, n_far = 0;
    for (uint32_t r = 0; r < R; ++r) {
      const uint32_t tl = tid + MB_THREADS * r;
      const unsigned long long k = lkey[tl];
      const bool selfc = k != kn[r];
      if (selfc) {
        key[s0 + tl] = k;
        my_mind = min(my_mind, (uint32_t)(k >> 32));
      }
      const unsigned long long sm = __ballot(selfc);
      const uint32_t nw = far_w[r] | ((lane & 32u) ? (uint32_t)(sm >> 32) : (uint32_t)sm);
      if ((lane & 31u) == 0) {
        n_pend += (uint32_t)__popc(nw);
        n_far += (uint32_t)__popc(far_w[r]);
        mb.pend[j * PW + (tid >> 5) + WPR * r] = nw;
      }
    }
    const unsigned long long has = __ballot(n_pend != 0);
    if (has) {
      my_mind = wave_min_u32(my_mind);
      n_pend += __shfl_xor(n_pend, 32);  // lanes 0 and 32 hold the two words of the wave
      n_far += __shfl_xor(n_far, 32);
      if (lane == 0) {
        atomicAdd(&s_npend, n_pend);
        if (n_far) atomicAdd(&s_nfar, n_far);
        atomicMin(&s_mind, my_mind);
      }
    }
  }
  __syncthreads();

  // ---- publish: region counts, activity
  const uint32_t total_sent = s_sent, npend = s_npend, nfar = s_nfar;
  const bool any_out = total_sent != 0;  // messages that left the block (same-block candidates are not counted)
  if (tid == 0) {
    if (any_out != wrote_out) mb.wrote[par_out][j] = any_out ? 1u : 0u;
    mb.blk_pend[j] = npend;
    mb.blk_mind[j] = s_mind;
    mb.blk_far[j] = nfar;
    if ((any_out || npend || (collect && an)) && *improved == 0u) *improved = 1u + mode;
    // what the schedule of the next launch reads: near activations of this sweep = the states it expanded (low word),
    // states of this block still waiting beyond the threshold (high word)
    if (an | nfar) atomicAdd(nf, ((unsigned long long)nfar << 32) | an);
    if (profile && !collect) {
      if (s_prof_arcs) atomicAdd(&ctl->arcs[(j % PROF_SHARDS) * PROF_STRIDE], s_prof_arcs);
      if (an) atomicAdd(&ctl->states[(j % PROF_SHARDS) * PROF_STRIDE], (unsigned long long)an);
    }
  }
  if (mb.dbg && tid == 0 && sweep < MB_DBG_SWEEPS) {
    mb.dbg[((size_t)sweep * nb + j) * 16 + 6] = wall_clock64();
    mb.dbg[((size_t)sweep * nb + j) * 16 + 7] = an;
    mb.dbg[((size_t)sweep * nb + j) * 16 + 8] = total_sent;
  }
}
