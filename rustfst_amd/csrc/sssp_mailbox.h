// sssp_mailbox.h — owner-computes relaxation sweeps ("mailbox sweeps") for sssp.hip.
//
// Included by sssp.hip inside namespace wfst { namespace { ... } } after Ctl / sweep_tau / enc_f32.
//
// Same recurrence as sssp_relax_kernel (single_shortest_path, rustfst/src/algorithms/shortest_path.rs:173-239:
// relax arc (s,w,t) with nd = d[s] (x) w, keep the minimum), same (d, hops) key, same near-far schedule — the
// least fixed point is unique, so the keys it leaves are bit-identical.  What changes is WHERE the minimum is taken:
//
//   * the states are cut into blocks of MB_B = 4096; workgroup j OWNS block j and is the only one that ever writes
//     key[] of its states.  It keeps the block's 4096 keys in LDS for the duration of a sweep (32 KB);
//   * a relaxation is a MESSAGE {enc(d[s] + w), hops[s] + 1, t mod 4096} (8 bytes) written by the workgroup that
//     owns s into the region reserved for the pair (block of s -> block of t).  Regions are static: region (i -> j)
//     has room for every arc from block i to block j (one message per arc per sweep at most), so a slot is one LDS
//     atomicAdd on a per-destination cursor — no global atomic, no reservation pass;
//   * next sweep the owner reads its regions (contiguous per destination: coalesced), applies the candidates with
//     LDS atomicMin, and expands the states that changed (or were waiting beyond the near-far threshold).
//
// Why: on MI355X the atomic sweep is bound by the 26.5 G/s global-atomic rate while a band is discovered and by 64-byte
// sectors moved for every 4..8-byte gather / atomic / flag store (1.5 GB per solve of the 1M-state graph against 0.2 GB
// algorithmic, profiles/r01h); a compute unit also retires only about one DIVERGENT lane access every three cycles.
// Here every access to memory is a coalesced stream except the read of an active state's arc row: messages are staged
// per destination in LDS and flushed as contiguous runs, the owner reads its regions as contiguous runs, arcs that stay
// inside the block never leave LDS, and nothing is looked up per arc on the sending side (a per-target filter was
// tried: its 2-byte gathers cost more than the messages it saved once those were coalesced).
//
// Limits: n <= 2^20 states (20 hop bits + 12 state bits share a word of the message), no negative weights (the hop
// count of a tentative label is then < n), n_arcs < 2^32.  Anything else takes sssp_relax_kernel.

constexpr uint32_t MB_LOG = 12;
constexpr uint32_t MB_B = 1u << MB_LOG;  // states per block
constexpr uint32_t MB_NBMAX = 256;       // blocks (n <= 2^20)
constexpr uint32_t MB_THREADS = 1024;
constexpr uint32_t MB_HOP_BITS = 32 - MB_LOG;
constexpr uint32_t MB_UNROLL = 8;  // active states a 16-lane group relaxes at once (independent load chains per lane)
#ifndef MB_WT
#define MB_WT 0  // staged messages leave with write-through (sc1) stores: nothing is left dirty in the L2 for the end of the kernel
#endif
constexpr uint32_t MB_STG = 24;    // messages per destination staged in LDS between two flushes (the rest is stored directly)

struct MboxView {
  const uint32_t* roff;    // [nb*nb + 1] region offsets, destination-major: region (i -> j) starts at roff[j*nb + i]
  const uint32_t* roff_t;  // [nb*nb]     roff_t[i*nb + j] = roff[j*nb + i] (what the SENDER i reads, contiguous)
  uint2* msgs[2];          // [E] each: messages of even / odd sweeps
  uint32_t* cnt[2];        // [nb*nb] each: cnt[j*nb + i] = messages in region (i -> j)
  uint32_t* wrote[2];      // [nb] each: sender i left non-zero counts in this parity's column
  uint32_t* pend;          // [nb * MB_B/32] states improved but not yet expanded (waiting beyond the threshold)
  uint32_t* blk_pend;      // [nb] number of such states per block
  uint32_t* blk_mind;      // [nb] min enc(d) among them
  uint32_t nb;
  unsigned long long* dbg;  // tuning only (WFST_SSSP_MBOX_TRACE): wall-clock stamps [sweep][block][16], or null
};
constexpr uint32_t MB_DBG_SWEEPS = 64;
#define MB_STAMP(p) do { if (mb.dbg && tid == 0 && sweep < MB_DBG_SWEEPS) mb.dbg[((size_t)sweep * nb + j) * 16 + (p)] = wall_clock64(); } while (0)

// ---- plan (cached on the FST handle): region offsets from the number of arcs between every pair of blocks
__global__ void __launch_bounds__(1024) mbox_hist_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn,
                                                         uint32_t n, uint32_t nb, uint32_t* __restrict__ hist) {
  __shared__ uint32_t l_h[MB_NBMAX];
  const uint32_t j = blockIdx.x;
  for (uint32_t d = threadIdx.x; d < nb; d += blockDim.x) l_h[d] = 0;
  __syncthreads();
  const uint32_t s0 = j << MB_LOG, s1 = min(n, s0 + MB_B);
  const uint32_t b = offsets[s0], e = offsets[s1];
  for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) atomicAdd(&l_h[wn[i].y >> MB_LOG], 1u);
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < nb; d += blockDim.x) hist[d * nb + j] = l_h[d];  // destination-major
}
__global__ void mbox_transpose_kernel(const uint32_t* __restrict__ roff, uint32_t nb, uint32_t* __restrict__ roff_t) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nb * nb) return;
  const uint32_t i = k / nb, j = k % nb;
  roff_t[k] = roff[j * nb + i];
}

// Initial state of a mailbox solve in one launch.
__global__ void __launch_bounds__(256) sssp_mbox_setup_kernel(uint64_t* __restrict__ key, MboxView mb, uint32_t* __restrict__ improved,
                                                              Ctl* __restrict__ ctl, uint32_t n, uint32_t start, float tau0) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  const uint32_t nb = mb.nb;
  for (uint32_t i = tid; i < n; i += nt) {
    const bool is_start = i == start;
    key[i] = is_start ? (uint64_t)enc_f32(0.0f) << 32 : KEY_INF;
  }
  for (uint32_t i = tid; i < nb * nb; i += nt) {
    mb.cnt[0][i] = 0;
    mb.cnt[1][i] = 0;
  }
  for (uint32_t i = tid; i < nb * (MB_B / 32); i += nt)
    mb.pend[i] = i == (start >> 5) ? 1u << (start & 31u) : 0u;
  for (uint32_t i = tid; i < nb; i += nt) {
    mb.wrote[0][i] = 0;
    mb.wrote[1][i] = 0;
    mb.blk_pend[i] = i == (start >> MB_LOG) ? 1u : 0u;
    mb.blk_mind[i] = i == (start >> MB_LOG) ? enc_f32(0.0f) : 0xFFFFFFFFu;
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

// One mailbox sweep.  Workgroup j:
//   trip 1  counts of its inbox regions, pending words, threshold, AND (speculatively) the block's keys and offsets
//           -> nothing arrives and nobody waiting is near: leave
//   trip 2  the messages; candidates applied with LDS atomicMin
//   scan    states whose key changed or that were waiting: written back; near ones (d <= tau) listed, far ones wait
//   trip 3  the arc rows of the listed states: candidate per arc -> LDS (same block) or the destination's staging slots
//   flush   staged messages leave as contiguous runs; counts of the regions written; the new waiting set
// The kernel is bound by its chain of dependent round trips (global AND LDS), so every phase asks for all it needs
// at once: no prefix sums, no searches, no shuffle reductions (ballots and one LDS atomic per wave instead).
__global__ void __launch_bounds__(MB_THREADS) sssp_mbox_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn,
                                                               uint64_t* __restrict__ key, MboxView mb, uint32_t par_in, uint32_t n,
                                                               uint32_t* __restrict__ improved_ring, Ctl* __restrict__ ctl,
                                                               uint32_t sweep, float delta, uint32_t near_low, uint32_t profile) {
  __shared__ unsigned long long lkey[MB_B];
  __shared__ uint32_t l_off[MB_B + 1];
  __shared__ uint16_t a_state[MB_B];  // states expanded in this sweep
  __shared__ uint2 l_stage[MB_NBMAX * MB_STG];
  __shared__ uint32_t l_roff_out[MB_NBMAX], l_cur[MB_NBMAX], l_base[MB_NBMAX];
  __shared__ uint32_t s_wany[MB_THREADS / 64];
  __shared__ uint32_t s_an, s_sent, s_npend, s_mind;
  __shared__ unsigned long long s_prof_arcs;
  constexpr uint32_t R = MB_B / MB_THREADS;  // states per thread
  constexpr uint32_t PW = MB_B / 32;         // pending words per block
  constexpr uint32_t WPR = MB_THREADS / 32;  // pending words between two states of one thread

  // `sweep` is the absolute sweep index: the host knows it (plain launches), which saves the trip to ctl->base
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t j = blockIdx.x, nb = mb.nb;
  const uint32_t par_out = par_in ^ 1u;
  const uint32_t s0 = j << MB_LOG;
  uint32_t* improved = improved_ring + (sweep % IMP_RING);
  MB_STAMP(0);

  // ---- trip 1.  Inbox region i is read by threads 4i .. 4i+3.
  const uint32_t reg = tid >> 2, q = tid & 3u;
  uint32_t c_in = 0, rb_in = 0;
  if (reg < nb) {
    c_in = mb.cnt[par_in][j * nb + reg];
    rb_in = mb.roff[j * nb + reg];
  }
  uint32_t pw[R];  // pending words of this thread's states (state tl = tid + 1024 r sits in word (tid >> 5) + 32 r)
  for (uint32_t r = 0; r < R; ++r) pw[r] = mb.pend[j * PW + (tid >> 5) + WPR * r];
  const uint32_t bp = mb.blk_pend[j], bmind = mb.blk_mind[j], wrote_out = mb.wrote[par_out][j];
  unsigned long long kreg[R];
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t s = s0 + tid + MB_THREADS * r;
    kreg[r] = KEY_INF;
    uint32_t o = 0;
    if (s < n) {
      kreg[r] = key[s];
      o = offsets[s];
    } else if (s == n) {
      o = offsets[n];
    }
    l_off[tid + MB_THREADS * r] = o;
    lkey[tid + MB_THREADS * r] = kreg[r];
  }
  if (tid == 0) l_off[MB_B] = s0 + MB_B <= n ? offsets[s0 + MB_B] : 0u;
  if (tid < nb) {
    l_roff_out[tid] = mb.roff_t[j * nb + tid];
    l_cur[tid] = 0;
    l_base[tid] = 0;
  }
  // every wave works the threshold out for itself (the same few words: one trip, no LDS hand-over)
  uint32_t streak, prev_near;
  const float tau = sweep_tau(ctl, sweep, delta, near_low, &streak, &prev_near);
  if (tid < 64) {
    if (tid == 0) {
      s_an = 0;
      s_sent = 0;
      s_npend = 0;
      s_mind = 0xFFFFFFFFu;
      s_prof_arcs = 0;
    }
    if (j == 0) {
      if (tid == 0) {
        ctl->tau[sweep % RING] = __float_as_uint(tau);
        ctl->streak[sweep % RING] = streak;
      }
      if (tid < NEAR_SHARDS) ctl->near[(sweep + 1) % NEAR_RING][tid * NEAR_STRIDE] = 0;  // recycle
    }
  }
  {
    const bool wany = __ballot(c_in != 0) != 0;
    if (lane == 0) s_wany[tid >> 6] = wany ? 1u : 0u;
  }
  __syncthreads();
  bool any_in = false;
  for (uint32_t w = 0; w < MB_THREADS / 64; w += 4) {
    const uint4 f = *(const uint4*)&s_wany[w];
    any_in |= (f.x | f.y | f.z | f.w) != 0;
  }
  MB_STAMP(1);
  const bool waiting = bp != 0;
  if (!any_in && (!waiting || dec_f32(bmind) > tau)) {
    // nothing arrives and nobody who waits is near: the block sleeps through this sweep
    if (wrote_out) {  // counts this workgroup published two sweeps ago are still in the column it writes now
      for (uint32_t d = tid; d < nb; d += MB_THREADS) mb.cnt[par_out][d * nb + j] = 0;
      if (tid == 0) mb.wrote[par_out][j] = 0;
    }
    if (waiting && tid == 0 && *improved == 0u) *improved = 1u;  // the solve is not over
    MB_STAMP(15);
    return;
  }

  // ---- trip 2: the messages
  const uint2* __restrict__ msgs_in = mb.msgs[par_in];
  {
    constexpr uint32_t MU = 8;  // messages a thread requests at once (a region of up to 32 messages is one trip)
    for (uint32_t k0 = q; k0 < c_in; k0 += 4u * MU) {
      uint2 m[MU];
      for (uint32_t u = 0; u < MU; ++u) {
        m[u] = make_uint2(0u, 0u);
        if (k0 + 4u * u < c_in) m[u] = msgs_in[rb_in + k0 + 4u * u];
      }
      for (uint32_t u = 0; u < MU; ++u)
        if (k0 + 4u * u < c_in)
          atomicMin(&lkey[m[u].x & (MB_B - 1u)], ((unsigned long long)m[u].y << 32) | (m[u].x >> MB_LOG));
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

  // ---- expansion: 16 lanes per state, MB_UNROLL states per group in flight; one flush of the staging slots per round
  const uint32_t an = s_an;
  const uint32_t sub = tid & 15u, grp = tid >> 4;  // 64 groups
  uint2* __restrict__ msgs_out = mb.msgs[par_out];
  uint32_t sent = 0;  // wave-uniform
  unsigned long long p_arcs = 0;
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
          if (rel < MB_STG) l_stage[db * MB_STG + rel] = msg;
          else msgs_out[l_roff_out[db] + slot[u]] = msg;
          if (h1_[u] >> MB_HOP_BITS) ctl->pad = 1u;  // cannot happen (header); the host refuses the result if it does
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
    if (reg < nb) {
      const uint32_t b0 = l_base[reg], cnt = min(l_cur[reg] - b0, MB_STG), ro = l_roff_out[reg] + b0;
      for (uint32_t k = q; k < cnt; k += 4) {
        const uint2 sm = l_stage[reg * MB_STG + k];
        if (MB_WT) __hip_atomic_store((unsigned long long*)&msgs_out[ro + k], ((unsigned long long)sm.y << 32) | sm.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else msgs_out[ro + k] = sm;
      }
    }
    if (r0 + ROUND < an) {  // another round: its messages are staged from the current cursors on
      __syncthreads();
      if (tid < nb) l_base[tid] = l_cur[tid];
      __syncthreads();
    }
  }
  if (profile)
    for (int d = 32; d >= 1; d >>= 1) p_arcs += __shfl_xor(p_arcs, d);
  if (lane == 0) {
    if (sent) atomicAdd(&s_sent, sent);
    if (profile && p_arcs) atomicAdd(&s_prof_arcs, p_arcs);
  }
  __syncthreads();
  MB_STAMP(5);

  // ---- states improved from inside the block during the expansion: written back; they wait (expanded next sweep).
  //      The new waiting set.
  {
    uint32_t n_pend = 0;
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
        if (nw != pw[r]) mb.pend[j * PW + (tid >> 5) + WPR * r] = nw;
      }
    }
    const unsigned long long has = __ballot(n_pend != 0);
    if (has) {
      my_mind = wave_min_u32(my_mind);
      n_pend += __shfl_xor(n_pend, 32);  // lanes 0 and 32 hold the two words of the wave
      if (lane == 0) {
        atomicAdd(&s_npend, n_pend);
        atomicMin(&s_mind, my_mind);
      }
    }
  }
  __syncthreads();

  // ---- publish: region counts, activity
  const uint32_t total_sent = s_sent, npend = s_npend;
  const bool any_out = total_sent != 0;  // messages that left the block (same-block candidates are not counted)
  if (any_out || wrote_out)
    for (uint32_t d = tid; d < nb; d += MB_THREADS) mb.cnt[par_out][d * nb + j] = l_cur[d];
  if (tid == 0) {
    if (any_out != (wrote_out != 0)) mb.wrote[par_out][j] = any_out ? 1u : 0u;
    if (npend != bp) mb.blk_pend[j] = npend;
    if (s_mind != bmind) mb.blk_mind[j] = s_mind;
    if ((any_out || npend) && *improved == 0u) *improved = 1u;
    // near activations of this sweep = the states it expanded (what the threshold schedule reads next sweep)
    if (an) atomicAdd(&ctl->near[sweep % NEAR_RING][(j % NEAR_SHARDS) * NEAR_STRIDE], an);
    if (profile) {
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
