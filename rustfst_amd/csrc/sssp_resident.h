// sssp_resident.h — the WIDE phase of a mailbox solve as ONE launch (included by sssp.hip after sssp_mailbox.h).
//
// Same owner-computes scheme, same recurrence (single_shortest_path, rustfst/src/algorithms/shortest_path.rs:173-239),
// same near-far schedule and therefore the same sequence of levels as sssp_mbox_kernel launched once per level — but
// the workgroups do not leave between two levels:
//   * workgroup j keeps the keys and arc offsets of its block in LDS for ALL the levels it runs (loaded once, written
//     back once), and the set of states to look at as a bit mask in LDS: a candidate that lowers a key (the LDS atomicMin
//     returns the old one) sets the state's bit, the scan lists the near ones and leaves the far ones set — no per-thread
//     copies of keys between levels, nothing of a level's bookkeeping goes through memory;
//   * a level's hand-over between workgroups is the data itself.  Region (i -> j) starts with a 16-byte HEADER
//     {tag of the level, message count, sender's near | far counts, sender's waiting count | sent flag} in the same
//     64-byte sector as its first six messages; the sender writes messages with write-through (`sc1`) stores, every
//     storing wave drains (`s_waitcnt vmcnt(0)`), then the headers go out — cdna_hip_programming.md Guideline 16, form R1,
//     with the header as a data-tagged granule.  The receiver polls the first sector of each of its regions (4 lanes per
//     region, one `sc1` 16-byte load each: ONE trip brings the tag, the count and the first six candidates) and applies
//     a region as soon as its tag is the level's; thin levels never need a second trip for the inbox;
//   * the numbers the schedule needs (near activations and far-waiting states of the previous level, whether anybody
//     sent or still waits) travel in the headers: every workgroup sums the same nb headers and takes the same decision
//     (threshold, WIDE / COLLECT, fixed point) without a counter in memory.
// The launch ends with the COLLECT level (hand-over to a NARROW launch of sssp_mbox_kernel), at the fixed point, or
// after `max_levels`; what it leaves in memory (keys, waiting masks, work-list segments, the schedule ring) is exactly
// what a sssp_mbox_kernel launch in the same mode would have left, so the two kernels can follow each other in any
// order.  Level 0 of a launch reads the inbox the previous launch left in the one-level kernel's format.
//
// Needs every workgroup resident (nb <= CUs, one 1024-thread workgroup per CU): the host checks it; every wait is
// bounded by a wall-clock limit — a workgroup that gives up raises `abort`, everybody leaves, the host runs the solve
// again with one launch per level.

constexpr uint32_t RS_HDR = 2;      // 8-byte units of a region header
constexpr uint32_t RS_FIRST = 6;    // messages sharing the header's 64-byte sector
constexpr uint32_t RS_ALIGN = 8;    // regions start on 64-byte boundaries (units)
#ifndef WFST_RS_MU
#define WFST_RS_MU 4
#endif
constexpr uint32_t RS_MU = WFST_RS_MU;  // 16-byte message loads a lane keeps in flight while it reads the rest of a region
constexpr uint32_t FLAG_RES_ABORT = 0xAB0u;  // activity flag of a launch that gave up waiting
constexpr uint32_t RS_MAX_SWEEP = 60000;     // tag = (sweep + 1) << 16 | level
constexpr uint32_t RS_LEVEL_CAP = 65000;
constexpr uint32_t RS_STG_MAX = 48;  // staging slots per destination of a resident launch (LDS permitting)
constexpr uint32_t RS_DUMMY = 32;    // cursors (and 64 staging slots) that lanes without a candidate use instead of branching
constexpr uint32_t RS_DYN_BUDGET = 100u * 1024u;  // dynamic LDS (staging + per-destination tables) next to ~57 KB static

typedef unsigned int rs_u32x4 __attribute__((ext_vector_type(4)));

struct ResView {
  uint2* msgs[2];           // regions of odd / even levels (level l reads msgs[l & 1], written by level l - 1)
  const uint32_t* roffh;    // [nb*nb] destination-major: header of region (i -> j) at unit roffh[j*nb + i]
  const uint32_t* roffh_t;  // [nb*nb] sender-major copy
  uint32_t* abort;          // one word, zeroed per solve
  unsigned long long* trace;  // tuning (WFST_SSSP_RES_TRACE): [level][block][4] wall-clock stamps, or null
  uint32_t bytes;           // of one parity buffer
  uint32_t tlim_ticks;      // wall_clock64 ticks (100 MHz) a workgroup waits for a header before it gives up
};
constexpr uint32_t RS_TRACE_LEVELS = 64;

__global__ void res_size_kernel(const uint32_t* __restrict__ hist, uint32_t cells, uint32_t* __restrict__ out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < cells) out[k] = (RS_HDR + hist[k] + RS_ALIGN - 1u) & ~(RS_ALIGN - 1u);
  else if (k == cells) out[k] = 0;
}

// sum over the 64 lanes of a wave, result in lane 63 (DPP row shifts / broadcasts: no LDS trips)
__device__ __forceinline__ uint32_t wave_sum_to_last(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31
  return v;
}
__device__ __forceinline__ uint32_t quad_first(uint32_t v) {  // lane 4k's value in lanes 4k .. 4k+3
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, false);  // quad_perm:[0,0,0,0]
}

// LDS of a resident workgroup, all of it dynamic (a block of 8192 states needs more than the 64 KB a kernel may declare
// statically): scalars, keys, arc offsets, the expansion list, the waiting mask, then the staging slots and the three
// per-destination tables.
struct ResScalars {
  unsigned long long tot[2][2];  // per level parity: near | far << 32, waiting | senders << 32 (sums over the headers)
  uint32_t lv[2][6];             // per level parity: an, sent, npend, nfar, spare, spare
  uint32_t nw[8];
  uint32_t abort_;
  uint32_t pad_[3];
};
static_assert(sizeof(ResScalars) % 16 == 0, "the arrays behind the scalars stay 16-byte aligned");
constexpr size_t res_lds_bytes(uint32_t log, uint32_t nb, uint32_t stg) {
  return sizeof(ResScalars) + ((size_t)8 << log) + ((size_t)4 << log) + 16 + ((size_t)2 << log) + ((size_t)4 << log) / 32 +
         ((size_t)nb * stg + 64) * 8 + (3 * (size_t)nb + RS_DUMMY) * 4;
}

// One round of a level's expansion: `lps` lanes per listed state (2 .. 8, chosen per FST from its out-degrees: MboxPlan::lps),
// each lane fetching TWO arcs in one 16-byte load, U states per lane group in flight (U = 1, 2, 4 by the length of the
// level's list: a thin level runs the short instances).  A wave holds 64 / lps groups (the lanes left over idle); rows of more
// than 2 lps arcs are finished afterwards, one state at a time, from the bounds in LDS.  No arrays of flags, no branch around a
// load (a lane without an arc reads arc 0 and drops it; a lane whose second arc lies beyond its row reads one arc too many and
// drops it: the arrays behind `wn` in the FST's arena keep the last row's overshoot inside the allocation).  What a lane
// keeps per state while its U loads are in flight is the two arcs, the distance and the hop word; whether it has arcs is two
// bits of a mask.  History (profiles/r06c_expand_round.md, r06g_lanes_per_state.md): 16 lanes x one 8-byte arc, U = 6, was
// 248 us at 1M states where 5 lanes x two arcs, U = 4 (fan-out 10: 94 % of the lanes busy instead of 62 %, 48 states in
// flight per wave instead of 24 for fewer registers) is 236: the round is a chain of latencies, and what it is worth is the
// number of rows in flight.  Candidate per arc -> LDS (a state of the same block: its key is lowered at once and it waits
// for the next level) or the destination's staging slots.  l_cur[d] counts the round's messages for destination d; one
// that finds the slots full is stored directly at its place in the region.  Returns nonzero if this lane sent anything.
template <uint32_t LOG, uint32_t U>
__device__ __forceinline__ uint32_t rs_expand_round(uint32_t LPS, uint32_t G, uint32_t r0, uint32_t an, uint32_t grp, uint32_t sub, const uint2* __restrict__ wn,
                                                     const uint16_t* a_state, unsigned long long* lkey, uint32_t* l_pend, const uint32_t* l_off,
                                                     uint32_t j, uint32_t* l_cur, const uint32_t* l_base, uint2* l_stage, uint32_t stg,
                                                     const uint32_t* l_roff_out, unsigned long long* __restrict__ msgs_out,
                                                     uint32_t* __restrict__ pad) {
  constexpr uint32_t HOP_BITS = 32 - LOG, B = 1u << LOG;  // (G = groups per workgroup; the idle lanes of a wave carry a grp beyond every list)
  if (!__any(r0 + grp < an)) return 0u;
  struct __attribute__((packed, aligned(8))) Arc2 {
    rs_u32x4 v;
  };
  rs_u32x4 a[U];
  float d[U];
  uint32_t hs[U], vmask = 0, lmask = 0, ovf = 0;
  for (uint32_t u = 0; u < U; ++u) {
    const uint32_t e = r0 + grp + G * u;
    const bool has = e < an;  // (e wraps for an idle lane: grp = 2^31)
    const uint32_t tl = a_state[has ? e : r0];
    const unsigned long long k = lkey[tl];
    const uint32_t b = l_off[tl], en = l_off[tl + 1];
    const uint32_t h1 = (uint32_t)k + 1u;
    d[u] = dec_f32((uint32_t)(k >> 32));
    hs[u] = h1 << LOG;
    const uint32_t i0 = b + 2u * sub;
    const bool v0 = has && i0 < en, v1 = has && i0 + 1u < en;
    vmask |= ((v0 ? 1u : 0u) | (v1 ? 2u : 0u)) << (2u * u);
    lmask |= (has && en - b > 2u * LPS ? 1u : 0u) << u;
    ovf |= v0 ? h1 >> HOP_BITS : 0u;
    a[u] = ((const Arc2*)(wn + (v0 ? i0 : 0u)))->v;
  }
  if (ovf) *pad = 1u;  // hop count beyond the message format: the host refuses the result
  uint32_t sent = 0;
  // The two candidates of a state together (their LDS atomics are issued before either is waited for), and WITHOUT a branch:
  // the round is bound by instruction issue (~56 instructions per candidate in the branchy form, a third of them exec-mask
  // bookkeeping), so a lane without a candidate for another block counts on a dummy cursor and stages into a dummy slot of its
  // own instead of jumping around the code; the one branch left is the candidate for a state of this very block (LDS
  // atomicMin at once: sent to itself through region (j -> j) it was 85 us in ONE region of the dense level — arc 0 of every
  // state of the benchmark's transducer stays in the block, and a region is read by four lanes).  Distances on this path are
  // sums of non-negative weights: enc_f32 is "set the sign bit".
  const uint32_t nb_ = (uint32_t)(l_base - l_cur);  // (the three per-destination tables are nb words each, + RS_DUMMY)
  const uint32_t lane31 = threadIdx.x & (RS_DUMMY - 1u);
  for (uint32_t u = 0; u < U; ++u) {
    const uint32_t w_[2] = {a[u].x, a[u].z}, nx[2] = {a[u].y, a[u].w};
    uint32_t enc[2], slot[2], vl = 0;
    for (uint32_t h = 0; h < 2; ++h) {
      const float c = (d[u] + __uint_as_float(w_[h])) + 0.0f;  // w1 (x) w2 = f32 add (tropical_weight.rs:60-70)
      enc[h] = __float_as_uint(c) | 0x80000000u;
      const bool valid = ((vmask >> (2u * u + h)) & 1u) != 0u && c < INF;  // +inf never improves (shortest_path.rs:226)
      const bool local = valid && (nx[h] >> LOG) == j, remote = valid && !local;
      vl |= (remote ? 1u : 0u) << h;
      slot[h] = atomicAdd(&l_cur[remote ? nx[h] >> LOG : 2u * nb_ + lane31], 1u);
      if (local) {  // same block: never leaves LDS; the state waits for the next level
        const uint32_t tl_ = nx[h] & (B - 1u);
        const unsigned long long c_ = ((unsigned long long)enc[h] << 32) | (hs[u] >> LOG);
        if (c_ < atomicMin(&lkey[tl_], c_)) atomicOr(&l_pend[tl_ >> 5], 1u << (tl_ & 31u));
      }
    }
    bool over = false;
    for (uint32_t h = 0; h < 2; ++h) {
      const bool valid = ((vl >> h) & 1u) != 0u;
      const uint32_t db = nx[h] >> LOG;
      const bool fits = valid && slot[h] < stg;
      over |= valid && !fits;
      l_stage[fits ? __umul24(db, stg) + slot[h] : __umul24(nb_, stg) + (threadIdx.x & 63u)] = make_uint2(hs[u] | (nx[h] & (B - 1u)), enc[h]);
    }
    sent |= vl;
    if (__any(over)) {  // (staging slots full: stored directly at its place in the region)
      for (uint32_t h = 0; h < 2; ++h) {
        const uint32_t db = nx[h] >> LOG;
        if (((vl >> h) & 1u) != 0u && slot[h] >= stg)
          __hip_atomic_store(&msgs_out[l_roff_out[db] + l_base[db] + slot[h]], ((unsigned long long)enc[h] << 32) | hs[u] | (nx[h] & (B - 1u)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // rows of more than 2 LPS arcs: the rest of each, state by state (uniform per lane group; the wave stays together)
  if (__any(lmask != 0u)) {
    for (uint32_t u = 0; u < U; ++u) {
      if (!__any(((lmask >> u) & 1u) != 0u)) continue;
      const bool lng = ((lmask >> u) & 1u) != 0u;
      const uint32_t e = r0 + grp + G * u;
      const uint32_t tl = a_state[lng ? e : r0];
      const uint32_t en = lng ? l_off[tl + 1] : 0u;
      uint32_t i = lng ? l_off[tl] + 2u * LPS + sub : 0u;
      while (__any(i < en)) {
        const uint2 ar = wn[i < en ? i : 0u];
        const float c = (d[u] + __uint_as_float(ar.x)) + 0.0f;
        if (i < en && c < INF) {
          const uint32_t enc1 = enc_f32(c), db = ar.y >> LOG, tl_ = ar.y & (B - 1u);
          if (db == j) {
            const unsigned long long c_ = ((unsigned long long)enc1 << 32) | (hs[u] >> LOG);
            if (c_ < atomicMin(&lkey[tl_], c_)) atomicOr(&l_pend[tl_ >> 5], 1u << (tl_ & 31u));
          } else {
            const uint32_t sl = atomicAdd(&l_cur[db], 1u);
            const uint2 msg = make_uint2(hs[u] | tl_, enc1);
            if (sl < stg) l_stage[db * stg + sl] = msg;
            else __hip_atomic_store(&msgs_out[l_roff_out[db] + l_base[db] + sl], ((unsigned long long)msg.y << 32) | msg.x, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
            sent |= 1u;
          }
        }
        i += LPS;
      }
    }
  }
  return sent;
}

template <uint32_t LOG>
__global__ void __launch_bounds__(MB_THREADS) sssp_mbox_resident_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn,
                                                                        uint64_t* __restrict__ key, MboxView mb, ResView rv, uint32_t par_in,
                                                                        uint32_t n, uint32_t* __restrict__ improved_ring, Ctl* __restrict__ ctl,
                                                                        uint32_t sweep, float delta, uint32_t near_low, uint32_t narrow_t,
                                                                        uint32_t max_levels, uint32_t lps_umax) {
  constexpr uint32_t MB_LOG = LOG, MB_B = 1u << LOG, MB_HOP_BITS = 32 - LOG, NW_SEG = MB_B;
  extern __shared__ __align__(16) unsigned char mb_dyn[];
  ResScalars& sc_ = *(ResScalars*)mb_dyn;
  unsigned long long (&s_tot)[2][2] = sc_.tot;
  uint32_t (&s_lv)[2][6] = sc_.lv;
  uint32_t* const s_nw = sc_.nw;
  uint32_t& s_abort = sc_.abort_;
  unsigned long long* const lkey = (unsigned long long*)(mb_dyn + sizeof(ResScalars));  // [MB_B]
  uint32_t* const l_off = (uint32_t*)(lkey + MB_B);                                     // [MB_B + 1] (+ 3 words of padding)
  uint16_t* const a_state = (uint16_t*)(l_off + MB_B + 4);                              // [MB_B] the states this level expands
  uint32_t* const l_pend = (uint32_t*)(a_state + MB_B);  // [MB_B / 32] states whose key was lowered since they were last expanded, or that wait
  constexpr uint32_t R = MB_B / MB_THREADS;
  constexpr uint32_t PW = MB_B / 32;
  constexpr uint32_t WPR = MB_THREADS / 32;
  asm volatile("" ::"s"(offsets), "s"(mb.cnt[1]), "s"(ctl), "s"(rv.roffh), "s"(max_levels));
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t j = blockIdx.x, nb = mb.nb, stg = mb.stg;
  uint2* const l_stage = (uint2*)(l_pend + PW);                  // [nb * stg]
  uint32_t* const l_roff_out = (uint32_t*)(l_stage + nb * stg + 64);  // [nb] first message slot of region (j -> d)
  uint32_t* const l_cur = l_roff_out + nb;                            // [nb]
  uint32_t* const l_base = l_cur + nb;                                // [nb] (+ RS_DUMMY cursors behind it)
  const uint32_t s0 = j << MB_LOG;
  uint32_t* improved = improved_ring + (sweep % IMP_RING);
  const uint32_t reg = tid / MB_LPR, q = tid % MB_LPR;
  static_assert(MB_LPR == 4, "the header sector is read by four lanes");

  // A launch queued behind the solve's last one (the host sizes a batch from the previous solve, plus a margin): the launch
  // before it certified the fixed point, or changed nothing, or gave up — nothing to load, nobody to wait for.
  if (sweep > 0) {
    const uint32_t pf = improved_ring[(sweep - 1u) % IMP_RING];
    if (pf == 0u || pf == FLAG_NARROW_CLEAN || pf == FLAG_RES_ABORT) return;
  }
  // ---- prologue: as sssp_mbox_kernel's first trip (everything issued before anything is consumed)
  uint32_t c_in = 0, rb_in = 0, hdr_in = 0;
  if (reg < nb) {
    c_in = mb.cnt[par_in][j * nb + reg];
    rb_in = mb.roff[j * nb + reg];
    hdr_in = rv.roffh[j * nb + reg];
  }
  uint32_t ro_out = 0;
  if (tid < nb) ro_out = rv.roffh_t[(size_t)j * nb + tid];
  uint32_t pw[R];
  for (uint32_t r = 0; r < R; ++r) pw[r] = mb.pend[j * PW + (tid >> 5) + WPR * r];
  const uint32_t bmind = mb.blk_mind[j], bfar = mb.blk_far[j];
  const bool wrote_out = mb.wrote[par_in ^ 1u][j] != 0u;
  const uint32_t wl_n = min(mb.wl_cnt[j], NW_SEG);
  // (launch 0 of a solve with NARROW launches is the head of the search: it never looks at the block's keys or offsets — this
  // kernel runs it for FSTs of 8192-state blocks, and 96 KB per workgroup is a trip and 23 MB nobody needs)
  const bool narrow0 = sweep == 0u && narrow_t != 0u;
  unsigned long long kreg[R];
  uint32_t oreg[R], o_last = 0;
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t s = s0 + tid + MB_THREADS * r;
    kreg[r] = KEY_INF;
    oreg[r] = 0;
    if (narrow0) continue;
    if (s < n) {
      kreg[r] = key[s];
      oreg[r] = offsets[s];
    } else if (s == n) {
      oreg[r] = offsets[n];
    }
  }
  if (tid == MB_THREADS - 1 && s0 + MB_B <= n && !narrow0) {
    uint32_t t = tid;
    asm volatile("" : "+v"(t));
    o_last = offsets[s0 + t + MB_THREADS * (R - 1) + 1];
  }
  const SchedRaw raw = mbox_sched_load(ctl, sweep);
  if (tid < nb) {
    l_roff_out[tid] = ro_out + RS_HDR;
    l_cur[tid] = 0;
    l_base[tid] = 0;
  }
  for (uint32_t r = 0; r < R; ++r) {
    l_off[tid + MB_THREADS * r] = oreg[r];
    lkey[tid + MB_THREADS * r] = kreg[r];
    if ((tid & 31u) == 0) l_pend[(tid >> 5) + WPR * r] = pw[r];
  }
  if (tid == MB_THREADS - 1) l_off[MB_B] = o_last;
  const Sched sc = mbox_sched_eval(ctl, raw, sweep, delta, near_low, narrow_t);
  if (tid < 24) {
    if (tid < 4) ((unsigned long long*)s_tot)[tid] = 0;
    if (tid >= 8 && tid < 20) ((uint32_t*)s_lv)[tid - 8] = 0u;
    if (tid == 20) s_abort = 0;
  }
  if (j == 0 && tid < NEAR_SHARDS) ctl->nf[(sweep + 1) % NEAR_RING][tid * NF_STRIDE] = 0;  // recycle
  __syncthreads();

  unsigned long long* const nf = &ctl->nf[sweep % NEAR_RING][(j % NEAR_SHARDS) * NF_STRIDE];
  if (wrote_out) {  // counts this workgroup published two sweeps ago in the one-level format: nothing is published there now
    for (uint32_t d = tid; d < nb; d += MB_THREADS) mb.cnt[par_in ^ 1u][(size_t)d * nb + j] = 0;
    if (tid == 0) mb.wrote[par_in ^ 1u][j] = 0;
  }
  if (sc.mode == MODE_NARROW) {  // launched where a NARROW launch was due: do what sssp_mbox_kernel would have done
    if (j == 0 && tid == 0) {
      ctl->tau[sweep % RING] = __float_as_uint(sc.tau);
      ctl->streak[sweep % RING] = sc.streak;
      ctl->mode[sweep % RING] = sc.mode;
    }
    if (sweep == 0) {  // the head of the search sends its far candidates in the one-level format (region offsets of that plan)
      if (tid < nb) l_roff_out[tid] = mb.roff_t[(size_t)j * nb + tid];
      __syncthreads();
    }
    mbox_narrow<LOG>(offsets, wn, key, mb, ctl, improved, sweep, sc.tau_use, sc.far_total, near_low, 0u, wl_n, bmind != 0xFFFFFFFFu, bfar,
                     (uint4*)lkey, s_nw, par_in ^ 1u, l_roff_out, l_cur, l_base);
    return;
  }

  // ---- state that lives across the levels
  float tau_rec = sc.tau, tau = sc.tau_use;
  uint32_t mode = sc.mode, streak = sc.streak;
  uint32_t cnt_prev = sc.prev_near;  // near activations of the level before the current one
  const uint32_t tag_base = (sweep + 1u) << 16;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)rv.msgs[0], 0, (int)rv.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)rv.msgs[1], 0, (int)rv.bytes, 0x00020000);
  uint32_t last_an = 0, last_nfar = 0;
  bool clean = false;
#define RS_STAMP(p) do { if (rv.trace && tid == 0 && lvl < RS_TRACE_LEVELS) rv.trace[((size_t)lvl * nb + j) * 4 + (p)] = wall_clock64(); } while (0)
  // a candidate for local state `tl`: the minimum is taken in LDS; a key that went down marks its state
#define RS_APPLY(tl_, cand_) do { const uint32_t tl__ = (tl_); const unsigned long long c__ = (cand_);                 \
    if (c__ < atomicMin(&lkey[tl__], c__)) atomicOr(&l_pend[tl__ >> 5], 1u << (tl__ & 31u)); } while (0)

  for (uint32_t lvl = 0;; ++lvl) {
    const uint32_t ps = lvl & 1u;
    // ---------------- inbox
    if (lvl == 0) {
      const uint2* __restrict__ msgs_in = mb.msgs[par_in];
      constexpr uint32_t MU = 8;
      for (uint32_t k0 = q; k0 < c_in; k0 += MB_LPR * MU) {
        uint2 m[MU];
        for (uint32_t u = 0; u < MU; ++u) {
          m[u] = make_uint2(0u, 0u);
          if (k0 + MB_LPR * u < c_in) m[u] = msgs_in[rb_in + k0 + MB_LPR * u];
        }
        for (uint32_t u = 0; u < MU; ++u)
          if (k0 + MB_LPR * u < c_in) RS_APPLY(m[u].x & (MB_B - 1u), ((unsigned long long)m[u].y << 32) | (m[u].x >> MB_LOG));
      }
    } else {
      const uint32_t want = tag_base | lvl;
      const __amdgpu_buffer_rsrc_t rs = ps ? rs1 : rs0;
      const uint32_t hdr_b = hdr_in * 8u;  // the header sector of region (reg -> j); this lane reads bytes 16 q .. 16 q + 15 of it
      bool pend = reg < nb;
      uint32_t t_nf = 0, t_ps = 0;
      uint32_t spins = 0;
      unsigned long long t0 = 0;
      // Nobody's header can be here sooner than a write-through store takes to land and a load to come back (~2 us: the
      // exchange micro-benchmark's floor), and a poll that finds nothing is nb sector reads queued in front of the other
      // workgroups' message stores: sleep through the part of the wait that cannot end (56 x 64 clocks; 0 / 8 / 16 / 32 / 48 /
      // 64 / 96: 236.7 / 236.6 / 236.6 / 231.4 / 228.6 / 226.8 / 233.8 us per 1M-state solve, same box; the sleep between
      // two polls, 1 .. 16 or growing, makes no difference)
#ifndef WFST_RS_PRESLEEP
#define WFST_RS_PRESLEEP 56
#endif
      if (WFST_RS_PRESLEEP) __builtin_amdgcn_s_sleep(WFST_RS_PRESLEEP);
      for (;;) {
        rs_u32x4 v = {0u, 0u, 0u, 0u};
        if (pend) v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(hdr_b + q * 16u), 0, 16);
        const uint32_t tag = quad_first(v.x), cnt = quad_first(v.y);
        if (pend && tag == want) {
          if (q == 0) {
            t_nf = v.z;
            t_ps = v.w;
          } else {
            const uint32_t i0 = 2u * (q - 1u);
            if (i0 < cnt) RS_APPLY(v.x & (MB_B - 1u), ((unsigned long long)v.y << 32) | (v.x >> MB_LOG));
            if (i0 + 1u < cnt) RS_APPLY(v.z & (MB_B - 1u), ((unsigned long long)v.w << 32) | (v.z >> MB_LOG));
          }
          // the rest of the region: pairs of messages, pair p = messages RS_FIRST + 2p, + 1 at byte 64 + 16 p
          for (uint32_t p0 = q; RS_FIRST + 2u * p0 < cnt; p0 += MB_LPR * RS_MU) {
            rs_u32x4 m[RS_MU];
            for (uint32_t u = 0; u < RS_MU; ++u) {
              const uint32_t p = p0 + MB_LPR * u;
              m[u] = rs_u32x4{0u, 0u, 0u, 0u};
              if (RS_FIRST + 2u * p < cnt) m[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(hdr_b + 64u + p * 16u), 0, 16);
            }
            for (uint32_t u = 0; u < RS_MU; ++u) {
              const uint32_t i0 = RS_FIRST + 2u * (p0 + MB_LPR * u);
              if (i0 < cnt) RS_APPLY(m[u].x & (MB_B - 1u), ((unsigned long long)m[u].y << 32) | (m[u].x >> MB_LOG));
              if (i0 + 1u < cnt) RS_APPLY(m[u].z & (MB_B - 1u), ((unsigned long long)m[u].w << 32) | (m[u].z >> MB_LOG));
            }
          }
          pend = false;
        }
        if (!__any(pend)) break;
        // somebody's header is not there yet: poll again (bounded by the wall clock; everybody leaves once anybody gave up)
        if ((++spins & 31u) == 0u || rv.tlim_ticks == 0u) {
          const unsigned long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          const uint32_t ab = __hip_atomic_load(rv.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ab != 0u || now - t0 > (unsigned long long)rv.tlim_ticks || rv.tlim_ticks == 0u) {  // (limit 0: tests of the fallback)
            if (lane == 0) {
              s_abort = 1u;
              __hip_atomic_store(rv.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
          }
        }
#ifndef WFST_RS_POLLSLEEP
#define WFST_RS_POLLSLEEP 1
#endif
        __builtin_amdgcn_s_sleep(WFST_RS_POLLSLEEP);
      }
      // the senders' figures: one header per 4 lanes (the other lanes hold zeros).  (245 LDS atomics on one address instead of
      // these shifts cost ~1 us per level: measured)
      const uint32_t a_near = wave_sum_to_last(t_nf & 0xFFFFu), a_far = wave_sum_to_last(t_nf >> 16);
      const uint32_t a_pend = wave_sum_to_last(t_ps & 0xFFFFu), a_sent = wave_sum_to_last(t_ps >> 16);
      if (lane == 63) {
        atomicAdd(&s_tot[ps][0], ((unsigned long long)a_far << 32) | a_near);
        atomicAdd(&s_tot[ps][1], ((unsigned long long)a_sent << 32) | a_pend);
      }
    }
    __syncthreads();  // B1: the inbox is applied
    RS_STAMP(0);
    if (lvl > 0) {
      if (s_abort) {
        // (sticky: FLAG_RES_ABORT is the largest flag value and every writer of this launch's flag takes the maximum, so a
        // workgroup that finishes its levels normally next to one that gave up cannot put another value over it)
        if (tid == 0) atomicMax(improved, FLAG_RES_ABORT);
        return;
      }
      const unsigned long long ta = s_tot[ps][0], tb = s_tot[ps][1];
      const uint32_t cnt = (uint32_t)ta, far_total = (uint32_t)(ta >> 32), pend_total = (uint32_t)tb, senders = (uint32_t)(tb >> 32);
      if (senders == 0u && pend_total == 0u) {  // the level before this one changed nothing anywhere: the fixed point
        clean = true;
        break;
      }
      // the schedule of this level: mbox_sched_eval's rules on the same figures
      const float prev = tau_rec;
      uint32_t m = MODE_WIDE;
      if (narrow_t && sweep + lvl >= 3u && cnt != 0u && cnt + far_total <= narrow_t && (cnt < cnt_prev || cnt + far_total <= min(narrow_t, NW_SMALL)))
        m = MODE_COLLECT;  // (never in the two levels behind the head: see mbox_sched_eval)
      if (lvl + 1u >= max_levels) m = MODE_COLLECT;  // (the launch is over: hand whatever is active to the next one)
      uint32_t st_new = 0;
      if (cnt >= near_low) {
        tau_rec = prev;
      } else if (cnt) {
        tau_rec = cnt < cnt_prev ? prev + delta : prev;
      } else {
        st_new = min(streak + 1u, 30u);
        tau_rec = prev + delta * (float)(1u << (st_new - 1u));
      }
      streak = st_new;
      mode = m;
      tau = m == MODE_COLLECT ? INF : tau_rec;
      cnt_prev = cnt;
    }
    if (tid < 8) {  // the other parity's level scalars: everybody has read them (B1 is behind their last readers)
      if (tid < 2) s_tot[ps ^ 1u][tid] = 0;
      else s_lv[ps ^ 1u][tid - 2] = 0u;
    }
    const bool collect = mode == MODE_COLLECT;
#ifdef WFST_RS_STAMP_ALT
    RS_STAMP(1);
#endif

    // ---------------- the marked states: near ones listed (and unmarked), far ones keep waiting
    {
      const uint32_t enc_tau = enc_f32(tau);  // (the encoding is monotone: d <= tau <=> enc(d) <= enc(tau); +inf lists every marked state)
      unsigned long long nm[R];
      uint32_t n_near = 0, n_far = 0;
      for (uint32_t r = 0; r < R; ++r) {
        const uint32_t tl = tid + MB_THREADS * r;
        const uint32_t w = l_pend[tl >> 5];
        nm[r] = 0;
        if (__ballot(w != 0u) == 0ull) continue;  // (nobody marked among the wave's 64 states)
        const uint32_t ed = (uint32_t)(lkey[tl] >> 32);
        const bool act = ((w >> (tl & 31u)) & 1u) != 0;
        const bool near = act && ed <= enc_tau;
        const unsigned long long fm = __ballot(act && !near);
        nm[r] = __ballot(near);
        if (w != 0u && (lane & 31u) == 0) l_pend[tl >> 5] = (lane & 32u) ? (uint32_t)(fm >> 32) : (uint32_t)fm;  // (its 32 lanes have read the word)
        n_near += (uint32_t)__popcll(nm[r]);
        n_far += (uint32_t)__popcll(fm);
      }
      if (n_near) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_lv[ps][0], n_near);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        for (uint32_t r = 0; r < R; ++r) {
          if ((nm[r] >> lane) & 1ull) a_state[base + (uint32_t)__popcll(nm[r] & ((1ull << lane) - 1ull))] = (uint16_t)(tid + MB_THREADS * r);
          base += (uint32_t)__popcll(nm[r]);
        }
      }
      if (n_far && lane == 0) atomicAdd(&s_lv[ps][3], n_far);
    }
    __syncthreads();  // B2
#ifndef WFST_RS_STAMP_ALT
    RS_STAMP(1);
#endif
    const uint32_t an = s_lv[ps][0];
    last_an = an;
    if (collect) {
      // ---- hand-over: the states this level would have expanded become the block's segment of the work list
      for (uint32_t e = tid; e < an; e += MB_THREADS) {
        const uint32_t tl = a_state[e];
        const unsigned long long k = lkey[tl];
        const uint32_t b = l_off[tl], c = l_off[tl + 1] - b;
        mb.wl[(size_t)j * NW_SEG + e] = make_uint4(s0 + tl, b, (uint32_t)(k >> 32), ((uint32_t)k << MB_LOG) | min(c, NW_DEG_SAT));
        if ((uint32_t)k >> MB_HOP_BITS) ctl->pad = 1u;
      }
      // (tau = +inf: nobody is far, nothing was expanded: nothing waits)
      last_nfar = 0;
      break;
    }

    // ---------------- expansion (rs_expand_round), messages into the regions of level lvl + 1 (write-through stores)
    {
      // lanes per listed state and states per lane group of the widest round: the host's choice per FST and launch
      // (MboxPlan::lps; umax = 4 unless a round of 4 G states would overrun the staging slots, see sssp.hip)
      const uint32_t LPS = lps_umax & 0xFFu, UMAX = lps_umax >> 8, GPW = 64u / LPS;  // GPW groups per wave (a group never straddles two waves)
      const uint32_t sub = lane % LPS, grp = lane / LPS < GPW ? (tid >> 6) * GPW + lane / LPS : 0x80000000u;
      unsigned long long* __restrict__ msgs_out = (unsigned long long*)rv.msgs[ps ^ 1u];
      const __amdgpu_buffer_rsrc_t rs_out = ps ? rs0 : rs1;
      const uint32_t G = (MB_THREADS / 64u) * GPW, ROUND = G * UMAX;
      static_assert(MB_THREADS == 1024, "a resident workgroup is sixteen waves");
      const uint32_t an_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)an);
      uint32_t sent = 0;
#define RS_EXPAND(U_, r0_) rs_expand_round<LOG, U_>(LPS, G, r0_, an_u, grp, sub, wn, a_state, lkey, l_pend, l_off, j, l_cur, l_base, l_stage, stg, l_roff_out, msgs_out, &ctl->pad)
      if (an_u <= ROUND) {
        // one round (every level but a band's widest): the staged runs start at the regions' first slots
        if (an_u > 2u * G) sent = RS_EXPAND(4, 0u);  // (only with UMAX = 4)
        else if (an_u > G) sent = RS_EXPAND(2, 0u);
        else if (an_u) sent = RS_EXPAND(1, 0u);
        __syncthreads();
        // flush: destination d's staged messages leave as one contiguous run (4 lanes per destination), two messages per
        // 16-byte write-through store (an 8-byte `sc1` store is a fabric write of its own: half the transactions to drain);
        // message k sits at unit ro + k, and ro is even
        if (reg < nb) {
          const uint32_t last = min(l_cur[reg], stg), ro = l_roff_out[reg];
          const uint2* __restrict__ stv = l_stage + reg * stg;
          for (uint32_t k = 2u * q; k + 1u < last; k += 2u * MB_LPR) {
            const uint2 m0 = stv[k], m1 = stv[k + 1u];
            const rs_u32x4 pr = {m0.x, m0.y, m1.x, m1.y};
            __builtin_amdgcn_raw_buffer_store_b128(pr, rs_out, (int)((ro + k) * 8u), 0, 16);
          }
          if (q == 1 && (last & 1u)) {
            const uint2 msg = stv[last - 1u];
            __hip_atomic_store(&msgs_out[ro + last - 1u], ((unsigned long long)msg.y << 32) | msg.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      } else {
        // several rounds: after each, the staged runs go out behind what the earlier rounds wrote (l_base) and the counts start again
        for (uint32_t r0 = 0; r0 < an_u; r0 += ROUND) {
          if (UMAX >= 4u) sent |= RS_EXPAND(4, r0);
          else sent |= RS_EXPAND(2, r0);
          __syncthreads();
          if (reg < nb) {
            const uint32_t first = l_base[reg], cnt = l_cur[reg], last = first + min(cnt, stg), ro = l_roff_out[reg];
            const uint2* __restrict__ stv = l_stage + reg * stg - first;  // staged message k at stv[k]
            const uint32_t p_lo = (first + 1u) & ~1u;
            if (q == 0 && (first & 1u) && last > first) {
              const uint2 msg = stv[first];
              __hip_atomic_store(&msgs_out[ro + first], ((unsigned long long)msg.y << 32) | msg.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (uint32_t k = p_lo + 2u * q; k + 1u < last; k += 2u * MB_LPR) {
              const uint2 m0 = stv[k], m1 = stv[k + 1u];
              const rs_u32x4 pr = {m0.x, m0.y, m1.x, m1.y};
              __builtin_amdgcn_raw_buffer_store_b128(pr, rs_out, (int)((ro + k) * 8u), 0, 16);
            }
            if (q == 1 && last > p_lo && ((last - p_lo) & 1u)) {
              const uint2 msg = stv[last - 1u];
              __hip_atomic_store(&msgs_out[ro + last - 1u], ((unsigned long long)msg.y << 32) | msg.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (q == 0) {  // (the four lanes of a destination are one wave: they have read both)
              l_base[reg] = first + cnt;
              l_cur[reg] = 0;
            }
          }
          __syncthreads();
        }
      }
#undef RS_EXPAND
      if (__any(sent != 0u) && lane == 0) s_lv[ps][1] = 1u;
    }
    RS_STAMP(2);
    // ---------------- what waits now: the far states of the scan and the states lowered from inside the block (every
    //                  expansion's LDS atomics are behind the last round's barrier; no round at all: the scan's are behind B2)
    if (tid < MB_B / 32) {
      uint32_t c = (uint32_t)__popc(l_pend[tid]);
      c = wave_sum_to_last(c);
      if (lane == 63 && c) atomicAdd(&s_lv[ps][2], c);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its messages have left (Guideline 16 R1)
    __syncthreads();  // B3
    // ---------------- publish: the headers of level lvl + 1
    {
      const uint32_t total_sent = s_lv[ps][1], npend = s_lv[ps][2], nfar = s_lv[ps][3];
      last_nfar = nfar;
      if (tid < nb) {
        const uint32_t c = l_base[tid] + l_cur[tid];
        const rs_u32x4 h = {tag_base | (lvl + 1u), c, (an & 0xFFFFu) | (nfar << 16), (npend & 0xFFFFu) | ((total_sent != 0u ? 1u : 0u) << 16)};
        __builtin_amdgcn_raw_buffer_store_b128(h, ps ? rs0 : rs1, (int)((l_roff_out[tid] - RS_HDR) * 8u), 0, 16);
        l_cur[tid] = 0;
        l_base[tid] = 0;
      }
    }
    RS_STAMP(3);
  }
#undef RS_APPLY

  // ---------------- the launch is over: keys, waiting masks, schedule ring as a one-level launch would leave them
  __syncthreads();
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t tl = tid + MB_THREADS * r, s = s0 + tl;
    if (s < n) key[s] = lkey[tl];
    if ((tid & 31u) == 0) mb.pend[j * PW + (tid >> 5) + WPR * r] = 0;
  }
  const bool collect_exit = !clean;
  if (tid == 0) {
    mb.blk_pend[j] = 0;
    mb.blk_mind[j] = 0xFFFFFFFFu;
    mb.blk_far[j] = 0;
    mb.wl_cnt[j] = collect_exit ? last_an : 0u;
    if (collect_exit) {
      if (last_an) atomicMax(improved, 1u + MODE_COLLECT);
      if (last_an | last_nfar) atomicAdd(nf, ((unsigned long long)last_nfar << 32) | last_an);
    } else if (j == 0) {
      atomicMax(improved, FLAG_NARROW_CLEAN);
    }
    if (j == 0) {
      ctl->tau[sweep % RING] = __float_as_uint(tau_rec);
      ctl->streak[sweep % RING] = streak;
      ctl->mode[sweep % RING] = collect_exit ? MODE_COLLECT : MODE_WIDE;
    }
  }
#undef RS_STAMP
}
