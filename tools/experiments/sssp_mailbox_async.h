// sssp_mailbox_async.h — mailbox relaxation with SEVERAL ROUNDS PER LAUNCH (included by sssp.hip after sssp_mailbox.h).
//
// Same owner-computes scheme as sssp_mailbox.h (workgroup j owns block j of 4096 states, candidates travel as 8-byte
// messages through the region reserved for their (source block, destination block) pair, minima are LDS atomics), same
// recurrence (shortest_path.rs:173-239), same unique fixed point.  What changes: a launch does not stop after one level.
// A hop between two compute units through memory costs ~5 us on MI355X whether it crosses a kernel boundary or not, but
// a kernel boundary also reloads every block's keys, drains the L2 and runs every workgroup in lock step: 33 levels cost
// 33 x 8.5 us before any arc is relaxed (profiles/r02*).  Here every workgroup keeps its keys in LDS and LOOPS:
//     poll the heads of its inbox regions -> apply the new messages -> expand the near states whose key changed ->
//     write messages -> publish the heads of the regions written -> poll again ...
// and leaves the launch when the whole grid is quiet.  Nothing ever waits for another workgroup (a round is a poll, not
// a spin on an event; every loop is bounded by a round count and a wall-clock limit), so a grid that is not fully
// resident, or a hand-off that is seen late, costs time only: whatever is left over (unread messages, unexpanded
// states) is picked up by the next launch, and the host stops at the first launch in which nobody did anything — the
// same convergence test as for the one-level kernels.
//
// Hand-off (cdna_hip_programming.md Guideline 16, form R1; per-XCD L2s are not coherent, a CU's L1 is never refreshed):
//   sender    message words with relaxed agent-scope stores (write-through `sc1`) -> EVERY storing wave drains
//             (`s_waitcnt vmcnt(0)`) -> __syncthreads() -> the region's head counter with a relaxed agent-scope store
//   receiver  head counters and message words with relaxed agent-scope loads (`sc1`: bypass the L1)
// Regions are rings of cap >= 2 x (arcs between the two blocks) + 2 slots (a power of two): a sender only starts a round
// of expansion when every destination ring has room for one message per arc (it reads the receivers' tail counters,
// published the same way); a stale tail only delays it.
//
// Quiescence inside a launch (Mattern's four-counter test on two monotone global sums): S = messages published,
// R = messages applied AND their consequences sent.  A workgroup adds to S BEFORE it publishes heads; what it has consumed
// it OWES to R and pays back only in a round whose scan finds nothing left to expand (an expansion may improve states of
// the own block through LDS, which is work nobody else can see); a workgroup that finds work without owing anything
// (states left dirty by the previous launch) first adds a token of its own to S.  So S == R, read twice by an idle
// workgroup with the same value, means that no message is in flight and nobody is working.  A wrong "quiet" would only
// end the launch early; a workgroup that sees S != R stuck for MA_STALL_POLLS polls (receivers that left on the time
// limit) leaves too.
//
// Threshold schedule: tau is fixed during a launch; the next launch keeps it if messages or near states were left over
// and advances it by delta x 2^streak otherwise (the far states waiting beyond it are then released) — one band of the
// near-far schedule per launch instead of one level per launch.

constexpr uint32_t MA_MAX_ROUNDS_DEFAULT = 1u << 14;
constexpr uint32_t MA_UNROLL = 4;  // states a 16-lane group expands at once (the persistent state of a round loop needs registers too)
constexpr unsigned long long MA_TLIM_TICKS = 100000ull;  // 1 ms of wall_clock64 (100 MHz): a launch never polls longer
constexpr uint32_t MA_STALL_POLLS = 48;  // idle polls with S != R and neither moving before a workgroup gives up on the launch

struct MboxGlobal {  // zeroed by the setup kernel
  uint32_t sent[NEAR_SHARDS * NEAR_STRIDE];  // S, sharded, one shard per 128-B line
  uint32_t recv[NEAR_SHARDS * NEAR_STRIDE];  // R
};

struct MboxAView {
  const uint2* rinfo;   // [nb*nb] receiver-major {region offset, cap - 1} of region (i -> j) at [j*nb + i]
  const uint4* sinfo;   // [nb*nb] sender-major   {region offset, cap - 1, arcs i -> j, 0} at [i*nb + j]
  uint2* msgs;          // [sum of caps]
  uint32_t* head_r;     // [nb*nb] receiver-major: messages sender i has published to j, at [j*nb + i]
  uint32_t* tail_s;     // [nb*nb] sender-major:   messages receiver j has consumed from i, at [i*nb + j]
  uint32_t* cur_s;      // [nb*nb] sender-major, private to the sender between launches (== head)
  uint32_t* tail_r;     // [nb*nb] receiver-major, private to the receiver between launches (== tail)
  uint32_t* pend;       // [nb * MB_B/32] states whose key changed since they were last expanded
  uint32_t* blk_pend;   // [nb]
  uint32_t* blk_mind;   // [nb] min enc(d) among them
  MboxGlobal* g;
  uint32_t nb;
};

// ---- plan: ring capacities and offsets from the arc counts between blocks (hist is destination-major, as built by
// mbox_hist_kernel)
__global__ void mboxa_caps_kernel(const uint32_t* __restrict__ hist, uint32_t cells, uint32_t* __restrict__ caps) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > cells) return;
  uint32_t c = 0;
  if (k < cells && hist[k]) {
    const uint32_t need = 2u * hist[k] + 2u;
    c = 4;
    while (c < need) c <<= 1;
  }
  caps[k] = c;
}
__global__ void mboxa_info_kernel(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ caps,
                                  const uint32_t* __restrict__ roff, uint32_t nb, uint2* __restrict__ rinfo,
                                  uint4* __restrict__ sinfo) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nb * nb) return;
  const uint32_t jj = k / nb, ii = k % nb;  // k = j*nb + i  (destination-major)
  const uint32_t mask = caps[k] ? caps[k] - 1u : 0u;
  rinfo[k] = make_uint2(roff[k], mask);
  sinfo[ii * nb + jj] = make_uint4(roff[k], mask, hist[k], 0u);
}

__global__ void __launch_bounds__(256) sssp_mboxa_setup_kernel(uint64_t* __restrict__ key, MboxAView mb, uint32_t* __restrict__ improved,
                                                               Ctl* __restrict__ ctl, uint32_t n, uint32_t start, float tau0) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  const uint32_t nb = mb.nb;
  for (uint32_t i = tid; i < n; i += nt) key[i] = i == start ? (uint64_t)enc_f32(0.0f) << 32 : KEY_INF;
  for (uint32_t i = tid; i < nb * nb; i += nt) {
    mb.head_r[i] = 0;
    mb.tail_s[i] = 0;
    mb.cur_s[i] = 0;
    mb.tail_r[i] = 0;
  }
  for (uint32_t i = tid; i < nb * (MB_B / 32); i += nt) mb.pend[i] = i == (start >> 5) ? 1u << (start & 31u) : 0u;
  for (uint32_t i = tid; i < nb; i += nt) {
    mb.blk_pend[i] = i == (start >> MB_LOG) ? 1u : 0u;
    mb.blk_mind[i] = i == (start >> MB_LOG) ? enc_f32(0.0f) : 0xFFFFFFFFu;
  }
  for (uint32_t i = tid; i < (uint32_t)(sizeof(MboxGlobal) / 4); i += nt) ((uint32_t*)mb.g)[i] = 0;
  for (uint32_t i = tid; i < IMP_RING; i += nt) improved[i] = 0;
  uint32_t* cw = (uint32_t*)ctl;
  constexpr uint32_t W_TAU0 = offsetof(Ctl, tau0) / 4, W_BEST = offsetof(Ctl, best) / 4;
  for (uint32_t i = tid; i < (uint32_t)(sizeof(Ctl) / 4); i += nt)
    cw[i] = i == W_TAU0 ? __float_as_uint(tau0) : (i == W_BEST || i == W_BEST + 1) ? 0xFFFFFFFFu : 0u;
}

#define MA_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// threshold of launch `sweep`: kept while anything near is left over, advanced otherwise (called by one full wave)
__device__ __forceinline__ float mboxa_tau(const Ctl* ctl, const MboxGlobal* g, uint32_t sweep, float delta, uint32_t* streak) {
  *streak = 0;
  if (sweep == 0) return ctl->tau0;
  const uint32_t p = (sweep - 1) % RING;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t mine = 0;
  if (lane < NEAR_SHARDS) mine = ctl->near[(sweep - 1) % NEAR_RING][lane * NEAR_STRIDE];
  else if (lane < 2 * NEAR_SHARDS) mine = g->sent[(lane - NEAR_SHARDS) * NEAR_STRIDE];
  else if (lane < 3 * NEAR_SHARDS) mine = g->recv[(lane - 2 * NEAR_SHARDS) * NEAR_STRIDE];
  const float prev = __uint_as_float(ctl->tau[p]);
  const uint32_t prev_streak = ctl->streak[p];
  for (int d = 8; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);  // sums inside each group of 16 lanes
  const uint32_t left = __shfl(mine, 0), s_sum = __shfl(mine, 16), r_sum = __shfl(mine, 32);
  if (left != 0 || s_sum != r_sum) return prev;
  const uint32_t st = min(prev_streak + 1u, 30u);
  *streak = st;
  return prev + delta * (float)(1u << (st - 1u));
}

__global__ void __launch_bounds__(MB_THREADS) sssp_mboxa_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn,
                                                                uint64_t* __restrict__ key, MboxAView mb, uint32_t n,
                                                                uint32_t* __restrict__ improved_ring, Ctl* __restrict__ ctl,
                                                                uint32_t sweep, float delta, uint32_t max_rounds, uint32_t profile) {
  __shared__ unsigned long long lkey[MB_B];
  __shared__ uint32_t l_off[MB_B + 1];
  __shared__ uint16_t a_state[MB_B];
  __shared__ uint2 l_stage[MB_NBMAX * MB_STG];
  __shared__ uint32_t l_roff_out[MB_NBMAX], l_mask_out[MB_NBMAX], l_arcs_out[MB_NBMAX];
  __shared__ uint32_t l_cur[MB_NBMAX], l_base[MB_NBMAX], l_pub[MB_NBMAX], l_tail[MB_NBMAX];
  __shared__ uint32_t s_wany[MB_THREADS / 64], s_wrecv[MB_THREADS / 64];
  __shared__ uint32_t s_an[2], s_noroom[2], s_S, s_R, s_late, s_npend, s_mind, s_nearleft;
  __shared__ unsigned long long s_prof_arcs, s_prof_states;
  constexpr uint32_t R = MB_B / MB_THREADS;
  constexpr uint32_t PW = MB_B / 32;
  constexpr uint32_t WPR = MB_THREADS / 32;

  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t j = blockIdx.x, nb = mb.nb;
  const uint32_t s0 = j << MB_LOG;
  uint32_t* improved = improved_ring + (sweep % IMP_RING);
  const unsigned long long t_start = tid == 0 ? wall_clock64() : 0ull;

  // ---- launch prologue: everything a round needs that does not change during the launch
  const uint32_t reg = tid >> 2, q = tid & 3u;  // inbox region `reg` is read by threads 4 reg .. 4 reg + 3
  uint32_t rb_in = 0, mask_in = 0, c_cons = 0;
  if (reg < nb) {
    const uint2 ri = mb.rinfo[j * nb + reg];
    rb_in = ri.x;
    mask_in = ri.y;
    c_cons = mb.tail_r[j * nb + reg];
  }
  uint32_t dirty = 0;  // bit r: state tid + 1024 r changed since it was last expanded
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t w = mb.pend[j * PW + (tid >> 5) + WPR * r];
    dirty |= ((w >> (tid & 31u)) & 1u) << r;
  }
  const uint32_t bp0 = mb.blk_pend[j], bmind0 = mb.blk_mind[j];
  unsigned long long kprev[R];
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t s = s0 + tid + MB_THREADS * r;
    kprev[r] = KEY_INF;
    uint32_t o = 0;
    if (s < n) {
      kprev[r] = key[s];
      o = offsets[s];
    } else if (s == n) {
      o = offsets[n];
    }
    l_off[tid + MB_THREADS * r] = o;
    lkey[tid + MB_THREADS * r] = kprev[r];
  }
  if (tid == 0) l_off[MB_B] = s0 + MB_B <= n ? offsets[s0 + MB_B] : 0u;
  if (tid < nb) {
    const uint4 si = mb.sinfo[j * nb + tid];
    l_roff_out[tid] = si.x;
    l_mask_out[tid] = si.y;
    l_arcs_out[tid] = si.z;
    const uint32_t c = mb.cur_s[j * nb + tid];
    l_cur[tid] = c;
    l_base[tid] = c;
    l_pub[tid] = c;
  }
  uint32_t streak;
  const float tau = mboxa_tau(ctl, mb.g, sweep, delta, &streak);
  if (tid == 0) {
    s_an[0] = s_an[1] = 0;
    s_noroom[0] = s_noroom[1] = 0;
    s_npend = 0;
    s_mind = 0xFFFFFFFFu;
    s_nearleft = 0;
    s_prof_arcs = 0;
    s_prof_states = 0;
    if (j == 0) {
      ctl->tau[sweep % RING] = __float_as_uint(tau);
      ctl->streak[sweep % RING] = streak;
    }
  }
  if (j == 0 && tid < NEAR_SHARDS) ctl->near[(sweep + 1) % NEAR_RING][tid * NEAR_STRIDE] = 0;  // recycle

  uint32_t owed = 0;  // consumed messages (and own tokens) not yet returned to R
  if (bp0 != 0) {  // dirty states from the previous launch: announced before the first poll of anybody can see "quiet"
    if (tid == 0) atomicAdd(&mb.g->sent[(j % NEAR_SHARDS) * NEAR_STRIDE], 1u);
    owed = 1;
  }
  uint32_t stall_n = 0, stall_S = 0, stall_R = 0;
  bool did_any = false;      // this workgroup did something in this launch
  uint32_t quiet_S = 0;      // S seen by the previous idle poll (valid when quiet_n > 0)
  uint32_t quiet_n = 0;
  uint2* __restrict__ msgs = mb.msgs;

  for (uint32_t round = 0; round < max_rounds; ++round) {
    // ---- poll: heads of the inbox regions, tails of the outbox regions, the two global sums (one trip)
    uint32_t head = c_cons;
    if (reg < nb) head = __hip_atomic_load(&mb.head_r[j * nb + reg], MA_RLX_AGENT);
    if (tid < nb) l_tail[tid] = __hip_atomic_load(&mb.tail_s[j * nb + tid], MA_RLX_AGENT);
    if (wave == MB_THREADS / 64 - 1) {
      uint32_t v = 0;
      if (lane < NEAR_SHARDS) v = __hip_atomic_load(&mb.g->sent[lane * NEAR_STRIDE], MA_RLX_AGENT);
      else if (lane < 2 * NEAR_SHARDS) v = __hip_atomic_load(&mb.g->recv[(lane - NEAR_SHARDS) * NEAR_STRIDE], MA_RLX_AGENT);
      for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d);
      if (lane == 0) s_S = v;
      if (lane == NEAR_SHARDS) s_R = v;
    }
    const uint32_t fresh = head - c_cons;  // new messages in this thread's region
    {
      const unsigned long long any = __ballot(fresh != 0);
      // messages of the wave's regions: every 4th lane carries one region's count
      uint32_t cnt = (q == 0) ? fresh : 0u;
      cnt += __shfl_xor(cnt, 4);
      cnt += __shfl_xor(cnt, 8);
      cnt += __shfl_xor(cnt, 16);
      cnt += __shfl_xor(cnt, 32);
      if (lane == 0) {
        s_wany[wave] = any != 0 ? 1u : 0u;
        s_wrecv[wave] = cnt;
      }
    }
    if (tid == 0) s_late = wall_clock64() - t_start > MA_TLIM_TICKS ? 1u : 0u;
    const uint32_t par = round & 1u;
    __syncthreads();
    bool any_in = false;
    uint32_t recv_round = 0;
    for (uint32_t w = 0; w < MB_THREADS / 64; w += 4) {
      const uint4 f = *(const uint4*)&s_wany[w];
      const uint4 c = *(const uint4*)&s_wrecv[w];
      any_in |= (f.x | f.y | f.z | f.w) != 0;
      recv_round += c.x + c.y + c.z + c.w;
    }
    const uint32_t S_seen = s_S, R_seen = s_R, late = s_late;
    owed += recv_round;

    // ---- the new messages: [c_cons, head) of every region, 4 threads per region, MU per thread and trip
    if (any_in) {
      constexpr uint32_t MU = 8;
      for (uint32_t k0 = c_cons + q; (int32_t)(head - k0) > 0; k0 += 4u * MU) {
        unsigned long long m[MU];
        for (uint32_t u = 0; u < MU; ++u) {
          m[u] = 0;
          if ((int32_t)(head - (k0 + 4u * u)) > 0)
            m[u] = __hip_atomic_load((const unsigned long long*)&msgs[rb_in + ((k0 + 4u * u) & mask_in)], MA_RLX_AGENT);
        }
        for (uint32_t u = 0; u < MU; ++u)
          if ((int32_t)(head - (k0 + 4u * u)) > 0) {
            const uint32_t lo = (uint32_t)m[u], hi = (uint32_t)(m[u] >> 32);  // {hops << 12 | state, enc(d)}
            atomicMin(&lkey[lo & (MB_B - 1u)], ((unsigned long long)hi << 32) | (lo >> MB_LOG));
          }
      }
      if (fresh != 0 && q == 0) __hip_atomic_store(&mb.tail_s[reg * nb + j], head, MA_RLX_AGENT);  // slots are free again
      c_cons = head;
      __syncthreads();
    }

    // ---- states whose key changed: written back; near dirty states are listed for expansion
    unsigned long long kn[R];
    for (uint32_t r = 0; r < R; ++r) kn[r] = lkey[tid + MB_THREADS * r];
    uint32_t listed = 0, n_near = 0;
    for (uint32_t r = 0; r < R; ++r) {
      if (kn[r] != kprev[r]) {
        key[s0 + tid + MB_THREADS * r] = kn[r];
        kprev[r] = kn[r];
        dirty |= 1u << r;
      }
      const bool near = ((dirty >> r) & 1u) != 0 && dec_f32((uint32_t)(kn[r] >> 32)) <= tau;
      listed |= (near ? 1u : 0u) << r;
      n_near += (uint32_t)__popcll(__ballot(near));
    }
    uint32_t base = 0;
    if (n_near) {
      if (lane == 0) base = atomicAdd(&s_an[par], n_near);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    }
    for (uint32_t r = 0; r < R; ++r) {
      const bool near = ((listed >> r) & 1u) != 0;
      const unsigned long long nm = __ballot(near);
      if (near) a_state[base + (uint32_t)__popcll(nm & ((1ull << lane) - 1ull))] = (uint16_t)(tid + MB_THREADS * r);
      base += (uint32_t)__popcll(nm);
    }
    // room for one message per arc in every destination ring (a stale tail only makes this stricter)
    if (tid < nb && l_arcs_out[tid] != 0 && (l_cur[tid] - l_tail[tid]) + l_arcs_out[tid] > l_mask_out[tid] + 1u) s_noroom[par] = 1u;
    __syncthreads();
    const uint32_t an = s_an[par];
    const bool expand = an != 0 && s_noroom[par] == 0;
    if (an != 0 && owed == 0) {  // work that no message announced (improvements from inside the block): a token of its own
      if (tid == 0) atomicAdd(&mb.g->sent[(j % NEAR_SHARDS) * NEAR_STRIDE], 1u);
      owed = 1;
    }
    if (tid == 0) {  // the other parity's words were last read a barrier ago: ready for the next round
      s_an[par ^ 1u] = 0;
      s_noroom[par ^ 1u] = 0;
    }
    if (expand) dirty &= ~listed;

    // ---- expansion: 16 lanes per state, MA_UNROLL states per group in flight; one flush of the staging slots per pass
    if (expand) {
      const uint32_t sub = tid & 15u, grp = tid >> 4;
      unsigned long long p_arcs = 0;
      constexpr uint32_t PASS = (MB_THREADS / 16) * MA_UNROLL;
      for (uint32_t r0 = 0; r0 < an; r0 += PASS) {
        uint32_t i_[MA_UNROLL], end_[MA_UNROLL], h1_[MA_UNROLL];
        float d_[MA_UNROLL];
        bool more = false;
        for (uint32_t u = 0; u < MA_UNROLL; ++u) {
          const uint32_t e = r0 + grp + (MB_THREADS / 16) * u;
          i_[u] = end_[u] = h1_[u] = 0;
          d_[u] = 0.0f;
          if (e < an) {
            const uint32_t tl = a_state[e];
            const unsigned long long k = lkey[tl];
            const uint32_t b = l_off[tl];
            end_[u] = l_off[tl + 1];
            d_[u] = dec_f32((uint32_t)(k >> 32));
            h1_[u] = (uint32_t)k + 1u;
            if (profile && sub == 0) p_arcs += end_[u] - b;
            i_[u] = b + sub;
          }
          more |= i_[u] < end_[u];
        }
        more = __any(more);
        while (more) {
          uint2 a[MA_UNROLL];
          bool v[MA_UNROLL];
          for (uint32_t u = 0; u < MA_UNROLL; ++u) {
            v[u] = i_[u] < end_[u];
            a[u] = make_uint2(0x7F800000u, 0u);
            if (v[u]) a[u] = wn[i_[u]];
          }
          uint32_t enc[MA_UNROLL], slot[MA_UNROLL];
          for (uint32_t u = 0; u < MA_UNROLL; ++u) {
            const float c = (d_[u] + __uint_as_float(a[u].x)) + 0.0f;  // w1 (x) w2 = f32 add (tropical_weight.rs:60-70)
            v[u] = v[u] && c < INF;                                    // +inf never improves (shortest_path.rs:226)
            enc[u] = enc_f32(c);
            if (v[u] && (a[u].y >> MB_LOG) == j) {  // the target lives in this block: the candidate never leaves LDS
              atomicMin(&lkey[a[u].y & (MB_B - 1u)], ((unsigned long long)enc[u] << 32) | h1_[u]);
              v[u] = false;
            }
          }
          for (uint32_t u = 0; u < MA_UNROLL; ++u) {
            slot[u] = 0;
            if (v[u]) slot[u] = atomicAdd(&l_cur[a[u].y >> MB_LOG], 1u);
          }
          more = false;
          for (uint32_t u = 0; u < MA_UNROLL; ++u) {
            if (v[u]) {
              const unsigned long long msg =
                  ((unsigned long long)enc[u] << 32) | (h1_[u] << MB_LOG) | (a[u].y & (MB_B - 1u));
              const uint32_t db = a[u].y >> MB_LOG, rel = slot[u] - l_base[db];
              if (rel < MB_STG) l_stage[db * MB_STG + rel] = make_uint2((uint32_t)msg, (uint32_t)(msg >> 32));
              else __hip_atomic_store((unsigned long long*)&msgs[l_roff_out[db] + (slot[u] & l_mask_out[db])], msg, MA_RLX_AGENT);
              if (h1_[u] >> MB_HOP_BITS) ctl->pad = 1u;  // cannot happen (sssp_mailbox.h); the host refuses the result
            }
            i_[u] += 16;
            more |= i_[u] < end_[u];
          }
          more = __any(more);
        }
        __syncthreads();
        // flush: destination d's staged messages leave as one contiguous run (4 lanes per destination)
        if (reg < nb) {
          const uint32_t b0 = l_base[reg], cnt = min(l_cur[reg] - b0, MB_STG), ro = l_roff_out[reg], mk = l_mask_out[reg];
          for (uint32_t k = q; k < cnt; k += 4) {
            const uint2 sm = l_stage[reg * MB_STG + k];
            __hip_atomic_store((unsigned long long*)&msgs[ro + ((b0 + k) & mk)], ((unsigned long long)sm.y << 32) | sm.x,
                               MA_RLX_AGENT);
          }
        }
        __syncthreads();
        if (tid < nb) l_base[tid] = l_cur[tid];
        __syncthreads();
      }
      if (profile) {
        for (int d = 32; d >= 1; d >>= 1) p_arcs += __shfl_xor(p_arcs, d);
        if (lane == 0 && p_arcs) atomicAdd(&s_prof_arcs, p_arcs);
        if (tid == 0) s_prof_states += an;
      }
      // ---- publish: every storing wave drains, S is raised BEFORE the heads move
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      uint32_t pub = 0;
      if (tid < nb) pub = l_cur[tid] - l_pub[tid];
      for (int d = 32; d >= 1; d >>= 1) pub += __shfl_xor(pub, d);  // waves 0..3 hold the destinations
      if (wave < 4 && lane == 0 && pub) {
        atomicAdd(&mb.g->sent[(j % NEAR_SHARDS) * NEAR_STRIDE], pub);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid < nb && l_cur[tid] != l_pub[tid]) {
        __hip_atomic_store(&mb.head_r[tid * nb + j], l_cur[tid], MA_RLX_AGENT);
        l_pub[tid] = l_cur[tid];
      }
    }
    // what was consumed is returned to R in a round that found nothing (left) to expand (S first, then R)
    if (an == 0 && owed != 0) {
      if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicAdd(&mb.g->recv[(j % NEAR_SHARDS) * NEAR_STRIDE], owed);
      }
      owed = 0;
    }

    // ---- leave when the grid is quiet: idle here, S == R twice in a row with the same S
    const bool worked = any_in || expand;
    did_any |= worked;
    if (worked || an != 0) {
      quiet_n = 0;
      stall_n = 0;
    } else if (S_seen == R_seen) {
      if (quiet_n != 0 && quiet_S != S_seen) quiet_n = 0;
      quiet_S = S_seen;
      if (++quiet_n >= 2) break;
    } else {
      quiet_n = 0;
      if (stall_n != 0 && stall_S == S_seen && stall_R == R_seen) {
        if (++stall_n >= MA_STALL_POLLS) break;
      } else {
        stall_n = 1;
        stall_S = S_seen;
        stall_R = R_seen;
      }
    }
    if (late) break;
    if (!worked) __builtin_amdgcn_s_sleep(8);
  }
  // tokens still held go back (the launch is over for this workgroup; what is dirty is announced again next launch)
  if (owed != 0 && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    atomicAdd(&mb.g->recv[(j % NEAR_SHARDS) * NEAR_STRIDE], owed);
  }

  // ---- epilogue: keys changed from inside the block since the last scan are written back; what is still dirty waits
  //      for the next launch; private counters go back to memory
  __syncthreads();
  {
    uint32_t my_mind = 0xFFFFFFFFu, n_pend = 0, n_nearleft = 0;
    for (uint32_t r = 0; r < R; ++r) {
      const unsigned long long k = lkey[tid + MB_THREADS * r];
      if (k != kprev[r]) {
        key[s0 + tid + MB_THREADS * r] = k;
        dirty |= 1u << r;
      }
      const bool dz = ((dirty >> r) & 1u) != 0;
      if (dz) {
        const uint32_t ed = (uint32_t)(k >> 32);
        my_mind = min(my_mind, ed);
        n_nearleft += dec_f32(ed) <= tau ? 1u : 0u;
      }
      const unsigned long long dm = __ballot(dz);
      const uint32_t nw = (lane & 32u) ? (uint32_t)(dm >> 32) : (uint32_t)dm;
      if ((lane & 31u) == 0) {
        n_pend += (uint32_t)__popc(nw);
        mb.pend[j * PW + (tid >> 5) + WPR * r] = nw;
      }
    }
    for (int d = 32; d >= 1; d >>= 1) n_nearleft += __shfl_xor(n_nearleft, d);
    const unsigned long long has = __ballot(n_pend != 0);
    if (has) {
      my_mind = wave_min_u32(my_mind);
      n_pend += __shfl_xor(n_pend, 32);
      if (lane == 0) {
        atomicAdd(&s_npend, n_pend);
        atomicMin(&s_mind, my_mind);
      }
    }
    if (lane == 0 && n_nearleft) atomicAdd(&s_nearleft, n_nearleft);
  }
  if (reg < nb && q == 0) mb.tail_r[j * nb + reg] = c_cons;
  if (tid < nb) mb.cur_s[j * nb + tid] = l_cur[tid];
  __syncthreads();
  if (tid == 0) {
    const uint32_t npend = s_npend;
    if (npend != bp0) mb.blk_pend[j] = npend;
    if (s_mind != bmind0) mb.blk_mind[j] = s_mind;
    if ((did_any || npend) && *improved == 0u) *improved = 1u;
    if (s_nearleft) atomicAdd(&ctl->near[sweep % NEAR_RING][(j % NEAR_SHARDS) * NEAR_STRIDE], s_nearleft);
    if (profile) {
      if (s_prof_arcs) atomicAdd(&ctl->arcs[(j % PROF_SHARDS) * PROF_STRIDE], s_prof_arcs);
      if (s_prof_states) atomicAdd(&ctl->states[(j % PROF_SHARDS) * PROF_STRIDE], s_prof_states);
    }
  }
}
