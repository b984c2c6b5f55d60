// sssp_binned.h — the DENSE levels of the atomic frontier sweeps as an owner-computes pass ("binned levels").
// Included by sssp.hip inside namespace wfst { namespace { ... } } after Ctl / enc_f32 / sweep_tau.
//
// Same recurrence (single_shortest_path, rustfst/src/algorithms/shortest_path.rs:173-239: relax arc (s, w, t) with
// nd = d[s] (x) w, keep the minimum), same (d, hops) key, same byte-flag frontier and near-far schedule as
// sssp_relax_kernel — the least fixed point is unique, so the keys are bit-identical whichever kernel runs a level.
//
// Why: beyond ~3M states the atomic sweeps are bound by their random accesses, not by streaming.  Measured at 5M states /
// 50M arcs (profiles/r05a_levels_5m.md): 70 M relaxations = 70 M 4-byte gathers of the distance shadow (a 64-byte
// sector each: 4.5 GB through the fabric) + 13 M returning 64-bit atomics, 2.4 ms; the ten levels of a band's growth
// run at 9 - 36 G arcs/s.  A level with a frontier this large is better served the way the mailbox launches serve
// small graphs: no gather, no atomic —
//   * EXPAND (sssp_bin_expand_kernel): workgroup g owns a contiguous range of SOURCE states; it scans their frontier
//     flags, lists the near ones, reads their arc rows and turns every relaxation into an 8-byte MESSAGE
//     {hops + 1, t mod 2^LOGD, enc(d[s] + w)} for the BIN (block of 2^LOGD destination states) of t.  Region (g -> b) is
//     static (room for every arc from range g to bin b: one message per arc per level at most), so a slot is one LDS
//     atomicAdd on the region's cursor, and the region fills front to back;
//   * APPLY (sssp_bin_apply_kernel): workgroup b owns bin b; it loads the bin's keys into LDS, streams its regions
//     (contiguous, destination-major), takes the minimum with LDS atomicMin, writes back the keys that went down and
//     flags those states for the next level.
// Per relaxation that is 8 B of arc + 8 B written + 8 B read, all streamed, instead of a 64-byte sector per gather;
// the price is 12 B per state of key traffic per level, which only a DENSE level repays.  So the kernel is chosen PER
// LEVEL, on the device: sssp_relax_kernel (first launch of every slot) predicts the level's frontier from the counts
// the previous level left (near activations, + the states waiting beyond the threshold when the threshold moves) and
// publishes the mode; a thin level is relaxed by it and the two kernels here leave at once, a dense one the other way
// round.  Both use the same key[], shadow[] and frontier flags: the hand-over between them is the frontier itself.
//
// Limits: hop counts < 2^(32 - LOGD) in a message (a state beyond that relaxes its arcs with the atomic path inside
// the expand kernel), no negative weights, n_arcs < 2^31, at most BN_MAXBINS bins.

constexpr uint32_t BN_THREADS = 1024;
constexpr uint32_t BN_CH = 2048;       // source states per chunk of the expand kernel (one listing + one flush each)
constexpr uint32_t BN_MAXBINS = 1024;  // bins at most (2^23 states with 8192-state bins, 2^24 with 16384)
constexpr uint32_t BN_ALIGN = 16;      // slots (8 B each) a region's size is rounded up to
constexpr uint32_t BN_MODE_DENSE = 7;  // ctl->mode[slot] of a binned level (0 = the atomic sweep ran it)
constexpr int BN_LDS_MAX = 159 * 1024;  // dynamic LDS a kernel here may ask for (160 KB per compute unit, less its few static words)

struct BinView {
  const uint32_t* roff;    // [nbin*G + 1] region offsets, destination-major: region (g -> b) starts at roff[b*G + g]
  const uint32_t* roff_t;  // [G*nbin]     the same, sender-major (what workgroup g of the expand kernel reads)
  uint2* msgs;             // [slots] one slot per arc, regions rounded up to BN_ALIGN slots
  uint32_t* cnt;           // [nbin*G] messages in region (g -> b); written by the sender, zeroed by the reader
  uint32_t nbin, G;
  uint32_t sg;             // source states per expand workgroup (a multiple of BN_CH)
  uint32_t hop_cap;        // hops + 1 from which a state's relaxations take the atomic path (1 << (32 - LOGD); tests: less)
};
constexpr size_t bin_expand_lds(uint32_t nbin) { return (size_t)(BN_THREADS / 64) * 128 * 16 + 2 * (size_t)nbin * 4; }
constexpr size_t bin_apply_lds(uint32_t logd) { return ((size_t)8 << logd) + ((size_t)1 << logd) / 8; }

// ---- plan (cached on the FST handle): arcs from every source range to every bin, scanned into region offsets
template <uint32_t LOGD>
__global__ void __launch_bounds__(1024) bin_hist_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn, uint32_t n,
                                                        uint32_t sg, uint32_t nbin, uint32_t G, uint32_t* __restrict__ hist) {
  __shared__ uint32_t l_h[BN_MAXBINS];
  const uint32_t g = blockIdx.x;
  for (uint32_t b = threadIdx.x; b < nbin; b += blockDim.x) l_h[b] = 0;
  __syncthreads();
  const uint32_t s0 = min(n, g * sg), s1 = min(n, s0 + sg);
  const uint32_t b0 = offsets[s0], e0 = offsets[s1];
  for (uint32_t i = b0 + threadIdx.x; i < e0; i += blockDim.x) atomicAdd(&l_h[wn[i].y >> LOGD], 1u);
  __syncthreads();
  // (destination-major; every region starts on a 128-byte line: its slots are filled front to back by one workgroup and
  // read back sixteen bytes per lane)
  for (uint32_t b = threadIdx.x; b < nbin; b += blockDim.x) hist[(size_t)b * G + g] = (l_h[b] + BN_ALIGN - 1u) & ~(BN_ALIGN - 1u);
}
__global__ void bin_transpose_kernel(const uint32_t* __restrict__ roff, uint32_t nbin, uint32_t G, uint32_t* __restrict__ roff_t) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nbin * G) return;
  const uint32_t g = k / nbin, b = k % nbin;
  roff_t[k] = roff[(size_t)b * G + g];
}

// A relaxation that does not fit the message format (hop count): the atomic path of sssp_relax_kernel, without chasing.
__device__ __forceinline__ void bin_relax_direct(uint64_t* __restrict__ key, uint32_t* __restrict__ shadow, uint8_t* __restrict__ flags_next,
                                                 uint32_t t, uint32_t enc_c, uint32_t h1, float tau, uint32_t& near_cnt, uint32_t& far_cnt) {
  const unsigned long long ck = ((unsigned long long)enc_c << 32) | h1;
  const unsigned long long old = atomicMin((unsigned long long*)&key[t], ck);
  if (ck < old) {
    atomicMin(&shadow[t], enc_c);  // (several lanes may win in turn: the shadow must end at the smallest)
    flags_next[t] = 1;
    if (dec_f32(enc_c) <= tau) near_cnt += 1u;
    else far_cnt += 1u;
  }
}

// ---- EXPAND: frontier flags of a range of source states -> messages in the bins' regions.
// Wave-autonomous: no barrier between the set-up and the publish.  A wave takes 128 consecutive states of the range at a
// time (two per lane: one 2-byte load for their flags, issued one iteration ahead), asks for keys and arc ranges of the
// flagged ones together, lists the near ones in a wave-private piece of LDS (in state order: consecutive active states
// share the 128-byte lines of their arc rows), then relaxes the list sixteen states at a time (16 lanes per state, four
// states per 16-lane group in flight).  A message takes its slot with one LDS atomicAdd on the region's cursor (shared by
// the workgroup's waves) and goes straight to its place in the region: the region is written front to back by one
// workgroup, so its lines fill up in that compute unit's L2 slice and leave it whole.  (Staging the messages per bin in
// LDS and flushing runs behind workgroup barriers — the first version — cost ~60 us per level in barriers and exposed
// trips: profiles/r05b_binned_first_version.md.)
template <uint32_t LOGD>
__global__ void __launch_bounds__(BN_THREADS) sssp_bin_expand_kernel(const uint32_t* __restrict__ offsets, const uint2* __restrict__ wn,
                                                                     uint64_t* __restrict__ key, uint8_t* __restrict__ flags_cur,
                                                                     uint8_t* __restrict__ flags_next, uint32_t n,
                                                                     uint32_t* __restrict__ improved_ring, Ctl* __restrict__ ctl,
                                                                     uint32_t sweep_offset, uint32_t* __restrict__ shadow, BinView bv,
                                                                     uint32_t profile) {
  constexpr uint32_t B = 1u << LOGD, U = 4, WAVES = BN_THREADS / 64, CHW = 128;
  extern __shared__ __align__(16) unsigned char bn_dyn[];
  __shared__ uint32_t s_any, s_near, s_far;
  __shared__ unsigned long long s_prof[2];
  const uint32_t sweep = ctl->base + sweep_offset;
  const uint32_t slot_k = sweep % RING;
  // the mode and the threshold of this level, as the slot's first launch (sssp_relax_kernel) published them
  if (ctl->mode[slot_k] != BN_MODE_DENSE) return;
  const float tau = __uint_as_float(ctl->tau[slot_k]);
  uint32_t* improved = improved_ring + (sweep % IMP_RING);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, g = blockIdx.x;
  const uint32_t nbin = bv.nbin, hop_cap = bv.hop_cap;
  uint4* const w_ent = (uint4*)bn_dyn + wv * CHW;               // [WAVES][CHW] {enc(d), hops + 1, arc begin, arc end}: this wave's list
  uint32_t* const l_cur = (uint32_t*)((uint4*)bn_dyn + WAVES * CHW);  // [nbin] messages written to region (g -> b) so far
  uint32_t* const l_base = l_cur + nbin;                        // [nbin] first slot of region (g -> b)
  const uint32_t* __restrict__ my_roff = bv.roff_t + (size_t)g * nbin;
  for (uint32_t b = tid; b < nbin; b += BN_THREADS) {
    l_cur[b] = 0;
    l_base[b] = my_roff[b];
  }
  if (tid == 0) {
    s_any = 0;
    s_near = 0;
    s_far = 0;
    s_prof[0] = 0;
    s_prof[1] = 0;
  }
  const uint32_t s_lo = min(n, g * bv.sg), s_hi = min(n, s_lo + bv.sg);
  const uint32_t sub = lane & 15u, grp = lane >> 4;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  uint32_t near_cnt = 0, far_cnt = 0;
  unsigned long long p_arcs = 0, p_states = 0;
  bool any = false;
  uint2* __restrict__ msgs = bv.msgs;
  // (s_lo is a multiple of BN_CH and the flag buffers are 16-byte aligned: the 2-byte flag loads are aligned; a pair that
  // straddles the end of the range reads one flag of the next range, or one byte of the buffer's padding, and drops it)
  auto load_flags = [&](uint32_t base) -> uint32_t {
    const uint32_t s = base + 2u * lane;
    return s < s_hi ? (uint32_t)*(const uint16_t*)(flags_cur + s) : 0u;
  };
  __syncthreads();

  uint32_t base = s_lo + wv * CHW;
  uint32_t fl = base < s_hi ? load_flags(base) : 0u;
  for (; base < s_hi; base += WAVES * CHW) {
    const uint32_t nbase = base + WAVES * CHW;
    const uint32_t fl_next = nbase < s_hi ? load_flags(nbase) : 0u;  // one iteration ahead
    const uint32_t s0 = base + 2u * lane;
    bool f[2] = {(fl & 0xFFu) != 0u && s0 < s_hi, (fl & 0xFF00u) != 0u && s0 + 1u < s_hi};
    fl = fl_next;
    if (__ballot(f[0] || f[1]) == 0ull) continue;
    // ---- keys and arc ranges of the flagged states, all asked for together
    uint64_t k[2] = {0, 0};
    uint32_t ob[2] = {0, 0}, oe[2] = {0, 0};
    for (uint32_t r = 0; r < 2; ++r)
      if (f[r]) {
        k[r] = key[s0 + r];
        ob[r] = offsets[s0 + r];
        oe[r] = offsets[s0 + r + 1];
      }
    if (f[0] && f[1]) *(uint16_t*)(flags_cur + s0) = 0;  // this buffer is the NEXT frontier two levels from now
    else if (f[0]) flags_cur[s0] = 0;
    else if (f[1]) flags_cur[s0 + 1] = 0;
    bool nr[2];
    for (uint32_t r = 0; r < 2; ++r) {
      nr[r] = false;
      if (f[r]) {
        if (dec_f32((uint32_t)(k[r] >> 32)) > tau) {  // far: stays in the frontier, is not relaxed in this level
          flags_next[s0 + r] = 1;
          far_cnt += 1u;
          any = true;
        } else {
          nr[r] = true;
          if (profile) {
            p_states += 1;
            p_arcs += oe[r] - ob[r];
          }
        }
      }
    }
    // ---- the wave's list, in state order: lane l's states are 2l and 2l + 1
    const unsigned long long m0 = __ballot(nr[0]), m1 = __ballot(nr[1]);
    const uint32_t an = (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
    if (an == 0u) continue;
    {
      const uint32_t pos = (uint32_t)__popcll(m0 & lanes_below) + (uint32_t)__popcll(m1 & lanes_below);
      if (nr[0]) w_ent[pos] = make_uint4((uint32_t)(k[0] >> 32), (uint32_t)k[0] + 1u, ob[0], oe[0]);
      if (nr[1]) w_ent[pos + (nr[0] ? 1u : 0u)] = make_uint4((uint32_t)(k[1] >> 32), (uint32_t)k[1] + 1u, ob[1], oe[1]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the list is private to the wave: its LDS accesses are ordered)
    // ---- expansion: 16 lanes per listed state, U states per 16-lane group in flight
    for (uint32_t e0 = 0; e0 < an; e0 += 4 * U) {
      uint32_t i[U], end[U], hs[U];
      float d[U];
      for (uint32_t u = 0; u < U; ++u) {
        const uint32_t e = e0 + grp + 4 * u;
        const bool has = e < an;
        const uint4 en = w_ent[has ? e : 0u];
        d[u] = dec_f32(en.x);
        hs[u] = en.y;
        i[u] = has ? en.z + sub : 0u;
        end[u] = has ? en.w : 0u;
      }
      uint2 a[U];
      for (uint32_t u = 0; u < U; ++u) a[u] = wn[i[u] < end[u] ? i[u] : 0u];  // (a lane without an arc reads arc 0 and drops it)
      for (;;) {
        uint32_t more = 0;
        for (uint32_t u = 0; u < U; ++u) {
          const float c = (d[u] + __uint_as_float(a[u].x)) + 0.0f;  // w1 (x) w2 = f32 add (tropical_weight.rs:60-70)
          if (i[u] < end[u] && c < INF) {                          // +inf never improves (shortest_path.rs:226)
            const uint32_t enc = enc_f32(c), t = a[u].y;
            if (hs[u] >= hop_cap) {
              bin_relax_direct(key, shadow, flags_next, t, enc, hs[u], tau, near_cnt, far_cnt);
              any = true;
            } else {
              const uint32_t b = t >> LOGD;
              const uint32_t sl = atomicAdd(&l_cur[b], 1u);
              msgs[l_base[b] + sl] = make_uint2((hs[u] << LOGD) | (t & (B - 1u)), enc);
            }
          }
          i[u] += 16;
          more |= i[u] < end[u] ? 1u : 0u;
        }
        if (!__any(more)) break;
        for (uint32_t u = 0; u < U; ++u) a[u] = wn[i[u] < end[u] ? i[u] : 0u];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the list is read before the next iteration overwrites it)
  }
  __syncthreads();
  // ---- publish: the region counts (the reader zeroes what it has read: only non-zero ones are written)
  for (uint32_t b = tid; b < nbin; b += BN_THREADS) {
    const uint32_t c = l_cur[b];
    if (c) {
      bv.cnt[(size_t)b * bv.G + g] = c;
      any = true;
    }
  }
  for (int dd = 32; dd >= 1; dd >>= 1) {
    near_cnt += __shfl_xor(near_cnt, dd);
    far_cnt += __shfl_xor(far_cnt, dd);
  }
  const bool wave_any = __any(any);
  if (lane == 0) {
    if (wave_any) s_any = 1u;
    if (near_cnt) atomicAdd(&s_near, near_cnt);
    if (far_cnt) atomicAdd(&s_far, far_cnt);
  }
  if (profile) {
    for (int dd = 32; dd >= 1; dd >>= 1) {
      p_arcs += __shfl_xor(p_arcs, dd);
      p_states += __shfl_xor(p_states, dd);
    }
    if (lane == 0 && (p_states | p_arcs)) {
      atomicAdd(&s_prof[0], p_arcs);
      atomicAdd(&s_prof[1], p_states);
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (s_any && *improved == 0u) *improved = 1u;
    if (s_near) atomicAdd(&ctl->near[sweep % NEAR_RING][(g % NEAR_SHARDS) * NEAR_STRIDE], s_near);
    if (s_far) atomicAdd(&ctl->far[sweep % NEAR_RING][(g % NEAR_SHARDS) * NEAR_STRIDE], s_far);
    if (profile && (s_prof[0] | s_prof[1])) {
      atomicAdd(&ctl->arcs[(g % PROF_SHARDS) * PROF_STRIDE], s_prof[0]);
      atomicAdd(&ctl->states[(g % PROF_SHARDS) * PROF_STRIDE], s_prof[1]);
    }
  }
}

// ---- APPLY: the messages of bin b -> its keys (LDS), the states that went down flagged for the next level.
// The bin's regions are contiguous (destination-major) and start on 128-byte lines; eight lanes read a region, sixteen
// bytes (two messages) per lane and load: every request is a whole line of one region.
template <uint32_t LOGD>
__global__ void __launch_bounds__(BN_THREADS) sssp_bin_apply_kernel(uint64_t* __restrict__ key, uint32_t* __restrict__ shadow,
                                                                    uint8_t* __restrict__ flags_next, uint32_t n,
                                                                    uint32_t* __restrict__ improved_ring, Ctl* __restrict__ ctl,
                                                                    uint32_t sweep_offset, BinView bv) {
  constexpr uint32_t B = 1u << LOGD, R = B / BN_THREADS, PW = B / 32, MU = 4, LPR = 8, RPP = BN_THREADS / LPR;
  constexpr uint32_t GR = 4;  // regions per 8-lane group (G <= 512)
  extern __shared__ __align__(16) unsigned char bn_dyn[];
  __shared__ uint32_t s_near, s_far;
  const uint32_t sweep = ctl->base + sweep_offset;
  const uint32_t slot_k = sweep % RING;
  if (ctl->mode[slot_k] != BN_MODE_DENSE) return;
  const float tau = __uint_as_float(ctl->tau[slot_k]);
  unsigned long long* const lkey = (unsigned long long*)bn_dyn;  // [B]
  uint32_t* const l_pend = (uint32_t*)(lkey + B);                // [B / 32] states whose key went down
  const uint32_t tid = threadIdx.x, lane = tid & 63u, b = blockIdx.x, G = bv.G;
  const uint32_t reg = tid / LPR, q = tid % LPR;
  const uint32_t s0 = b << LOGD;
  // ---- trip 1: the counts of my regions, and (speculatively: every bin of a dense level gets messages) the keys
  uint32_t c_in[GR], rb[GR];
  for (uint32_t k = 0; k < GR; ++k) {
    const uint32_t rg = reg + RPP * k;
    c_in[k] = rb[k] = 0;
    if (rg < G) {
      c_in[k] = bv.cnt[(size_t)b * G + rg];
      rb[k] = bv.roff[(size_t)b * G + rg];
    }
  }
  unsigned long long kreg[R];
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t s = s0 + tid + BN_THREADS * r;
    kreg[r] = s < n ? key[s] : KEY_INF;
  }
  if (tid == 0) {
    s_near = 0;
    s_far = 0;
  }
  uint32_t mine = 0;
  for (uint32_t k = 0; k < GR; ++k) mine |= c_in[k];
  if (!__syncthreads_or((int)(mine != 0u))) return;  // nothing arrived: the bin sleeps through this level
  for (uint32_t r = 0; r < R; ++r) lkey[tid + BN_THREADS * r] = kreg[r];
  for (uint32_t w = tid; w < PW; w += BN_THREADS) l_pend[w] = 0;
  __syncthreads();
  // ---- trip 2: the messages; candidates applied with LDS atomicMin, a key that went down marks its state
  const uint4* __restrict__ msgs2 = (const uint4*)bv.msgs;  // pairs of messages (regions start on even slots)
#define BN_APPLY(mx_, my_) do { const uint32_t tl__ = (mx_) & (B - 1u);                                              \
    const unsigned long long c__ = ((unsigned long long)(my_) << 32) | ((mx_) >> LOGD);                            \
    if (c__ < atomicMin(&lkey[tl__], c__)) atomicOr(&l_pend[tl__ >> 5], 1u << (tl__ & 31u)); } while (0)
  for (uint32_t k = 0; k < GR; ++k) {
    const uint32_t c = c_in[k], base2 = rb[k] >> 1;
    for (uint32_t p0 = q; 2u * p0 < c; p0 += LPR * MU) {
      uint4 m[MU];
      for (uint32_t u = 0; u < MU; ++u) {
        m[u] = make_uint4(0u, 0u, 0u, 0u);
        if (2u * (p0 + LPR * u) < c) m[u] = msgs2[base2 + p0 + LPR * u];
      }
      for (uint32_t u = 0; u < MU; ++u) {
        const uint32_t i0 = 2u * (p0 + LPR * u);
        if (i0 < c) BN_APPLY(m[u].x, m[u].y);
        if (i0 + 1u < c) BN_APPLY(m[u].z, m[u].w);
      }
    }
    if (q == 0 && c) bv.cnt[(size_t)b * G + reg + RPP * k] = 0;  // read: the region is empty again
  }
#undef BN_APPLY
  __syncthreads();
  // ---- the states that went down: key and shadow written back, flagged for the next level
  uint32_t near_cnt = 0, far_cnt = 0;
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t tl = tid + BN_THREADS * r, s = s0 + tl;
    if ((l_pend[tl >> 5] >> (tl & 31u)) & 1u) {
      const unsigned long long k = lkey[tl];
      key[s] = k;
      shadow[s] = (uint32_t)(k >> 32);
      flags_next[s] = 1;
      if (dec_f32((uint32_t)(k >> 32)) <= tau) near_cnt += 1u;
      else far_cnt += 1u;
    }
  }
  for (int dd = 32; dd >= 1; dd >>= 1) {
    near_cnt += __shfl_xor(near_cnt, dd);
    far_cnt += __shfl_xor(far_cnt, dd);
  }
  if (lane == 0) {
    if (near_cnt) atomicAdd(&s_near, near_cnt);
    if (far_cnt) atomicAdd(&s_far, far_cnt);
  }
  __syncthreads();
  if (tid == 0 && (s_near | s_far)) {
    uint32_t* improved = improved_ring + (sweep % IMP_RING);
    if (*improved == 0u) *improved = 1u;
    if (s_near) atomicAdd(&ctl->near[sweep % NEAR_RING][(b % NEAR_SHARDS) * NEAR_STRIDE], s_near);
    if (s_far) atomicAdd(&ctl->far[sweep % NEAR_RING][(b % NEAR_SHARDS) * NEAR_STRIDE], s_far);
  }
}
