// nbest_batch.hip — n > 1 shortest paths of MANY small FSTs in one launch (BASELINE configs[4]: the look-ahead compositions
// of a batch of acceptors, a few hundred states each, n = 10): one wavefront per FST does the whole of
//   shortest_distance (shortest_distance.rs:153-237)  -> forward distances, label correcting in LDS
//   reverse           (reverse.rs:33-87)              -> in-arcs in (source state, arc position) order + the super-initial arcs
//   n_shortest_path   (shortest_path.rs:409-518)      -> the best-first search with the reference's binary Heap (:340-407)
//                                                        under ShortestPathCompare (:288-338), replayed step for step
//   connect           (connect.rs:51-66)              -> the chains of the n selected paths, renumbered in creation order
// and writes the result tree into pinned host memory.  The search is inherently sequential (the order in which the heap
// pops tied elements decides which states exist): one lane runs it out of LDS (heap, keys, flags), the other 63 help with
// the phases around it.  An FST that does not fit the per-problem limits (states, arcs, tree entries, LDS), has negative
// weights or asks for more than 64 paths reports a status and goes through the host search of nshortest.hip instead;
// large inputs never come here.  Results are bit-identical to that path (and to the oracle): same distances (the exact
// (min,+) fixed point of f32 sums), same heap order, same numbering, same property word.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "fst_props.h"

namespace wfst {

namespace {

constexpr uint32_t NB_MAX_STATES = 4096;  // states of one input
constexpr uint32_t NB_MAX_ARCS = 8192;
constexpr uint32_t NB_MAX_PATHS = 64;     // (one lane per selected path in the last phase)
constexpr uint32_t NB_NONE = 0xFFFFFFFFu;
enum : uint32_t { NB_OK = 0, NB_TREE_FULL = 1, NB_OUT_FULL = 2 };

struct NbProb {
  const uint32_t* off;
  const wfst_tr* arcs;
  const float* finals;
  uint32_t n, n_arcs;
  int32_t start;
  uint32_t pad;
  uint64_t scratch;  // byte offset of the problem's slice of the scratch arena
};
struct NbOut {
  uint32_t status, n_states, n_arcs, facts;
  uint32_t has_start, tree_states, pops, pad;
  uint64_t payload;  // byte offset inside the pinned payload area: offsets[n_states + 1] | finals[n_states] | arcs[n_arcs]
};

__device__ __forceinline__ uint32_t nb_enc(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float nb_dec(uint32_t e) {
  uint32_t b = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
  return __uint_as_float(b);
}
// TropicalWeight with the reference's semantics (tropical_weight.rs:26-171, semiring.rs:159-168)
__device__ __forceinline__ float nb_plus(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float nb_times(float a, float b) { return a == INF ? a : (b == INF ? b : (a + b) + 0.0f); }
__device__ __forceinline__ bool nb_weq(float a, float b) { return a <= b + props::KDELTA && b <= a + props::KDELTA; }
__device__ __forceinline__ bool nb_is_zero(float w) { return nb_weq(w, INF); }
__device__ __forceinline__ bool nb_is_one(float w) { return nb_weq(w, 0.0f); }
__device__ __forceinline__ bool nb_approx_equal(float a, float b, float delta) { return fabsf(a - b) <= delta; }
__device__ __forceinline__ bool nb_natural_less(float w1, float w2) {  // shortest_path.rs:284-286
  return nb_weq(nb_plus(w1, w2), w1) && !nb_weq(w1, w2);
}
// add_tr's facts (fst_props.h add_trs_by_facts)
__device__ __forceinline__ uint32_t nb_facts(const wfst_tr& tr, uint32_t state) {
  const bool weighted = !nb_is_zero(tr.weight) && !nb_is_one(tr.weight);
  return (tr.ilabel != tr.olabel ? 1u : 0u) | (tr.ilabel == 0u ? 2u : 0u) | (tr.ilabel == 0u && tr.olabel == 0u ? 4u : 0u) |
         (tr.olabel == 0u ? 8u : 0u) | (weighted ? 64u : 0u) | (tr.nextstate <= state ? 128u : 0u);
}

// scratch slice of one problem (global memory), T = tree capacity:
//   dist2[n + 1] f32 | roff[n + 2] u32 | rarcs[n_arcs] wfst_tr | super[n] {state + 1, weight} | rcount[n + 2] u32 |
//   p_state[T] u32 | p_w[T] f32 | o_arc[T] wfst_tr | start_arc[NB_MAX_PATHS] u32
__host__ __device__ inline size_t nb_al(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline size_t nb_scratch_bytes(uint32_t n, uint32_t n_arcs, uint32_t T) {
  return nb_al(4 * ((size_t)n + 1)) + nb_al(4 * ((size_t)n + 2)) + nb_al(16 * (size_t)n_arcs) + nb_al(8 * (size_t)n) +
         nb_al(4 * ((size_t)n + 2)) + nb_al(4 * (size_t)T) + nb_al(4 * (size_t)T) + nb_al(16 * (size_t)T) + nb_al(4 * NB_MAX_PATHS);
}
// LDS: heap[T] u32 | hk[T] f32 | some[T] u8 (aliased by the distance phase: dist[n] u32)
__host__ __device__ inline size_t nb_lds_bytes(uint32_t n, uint32_t T) {
  const size_t a = 9 * (size_t)T + 64, b = 4 * ((size_t)n + 1);
  return nb_al(a > b ? a : b);
}

__global__ void __launch_bounds__(64) nbest_wave_kernel(const NbProb* __restrict__ probs, uint8_t* __restrict__ scratch,
                                                        NbOut* __restrict__ outs, uint8_t* __restrict__ payload,
                                                        unsigned long long* __restrict__ cursor, unsigned long long payload_cap,
                                                        uint32_t nshortest, float delta, uint32_t T) {
  extern __shared__ __align__(16) unsigned char nb_lds[];
  const uint32_t lane = threadIdx.x;
  const NbProb pr = probs[blockIdx.x];
  NbOut* const out = outs + blockIdx.x;
  const uint32_t n = pr.n;
  uint8_t* sp = scratch + pr.scratch;
  float* const dist2 = (float*)sp;
  sp += nb_al(4 * ((size_t)n + 1));
  uint32_t* const roff = (uint32_t*)sp;  // in-arcs of original state t: rarcs[roff[t] .. roff[t + 1])
  sp += nb_al(4 * ((size_t)n + 2));
  wfst_tr* const rarcs = (wfst_tr*)sp;
  sp += nb_al(16 * (size_t)pr.n_arcs);
  uint2* const super = (uint2*)sp;  // {rfst state = final state + 1, weight bits}
  sp += nb_al(8 * (size_t)n);
  uint32_t* const rcount = (uint32_t*)sp;
  sp += nb_al(4 * ((size_t)n + 2));
  uint32_t* const p_state = (uint32_t*)sp;
  sp += nb_al(4 * (size_t)T);
  float* const p_w = (float*)sp;
  sp += nb_al(4 * (size_t)T);
  wfst_tr* const o_arc = (wfst_tr*)sp;
  sp += nb_al(16 * (size_t)T);
  uint32_t* const start_arc = (uint32_t*)sp;

  uint32_t* const l_dist = (uint32_t*)nb_lds;  // phase 1
  uint32_t* const heap = (uint32_t*)nb_lds;    // phase 4
  float* const hk = (float*)(nb_lds + 4 * (size_t)T);
  uint8_t* const some = nb_lds + 8 * (size_t)T;
  __shared__ uint32_t s_n_super, s_tree, s_found, s_status, s_facts, s_pops;

  auto finish_empty = [&]() {  // FO::new(): no states, no start (shortest_path.rs:426-431)
    if (lane == 0) *out = NbOut{NB_OK, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0ull};
  };
  if (pr.start < 0 || n == 0) {
    finish_empty();
    return;
  }

  // ---- 1. forward distances: the exact (min,+) fixed point (f32 + is monotone: unique whatever the order)
  for (uint32_t s = lane; s < n; s += 64) l_dist[s] = s == (uint32_t)pr.start ? nb_enc(0.0f) : nb_enc(INF);
  __syncthreads();
  for (uint32_t round = 0;; ++round) {
    if (round > n + 1u) {  // more rounds than states: only a negative cycle does that (the driver excludes negative weights,
      if (lane == 0) *out = NbOut{NB_TREE_FULL, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0ull};  // but a wave must never spin): host search
      return;
    }
    bool changed = false;
    for (uint32_t s = lane; s < n; s += 64) {
      const float ds = nb_dec(l_dist[s]);
      if (!(ds < INF)) continue;
      for (uint32_t k = pr.off[s]; k < pr.off[s + 1]; ++k) {
        const wfst_tr a = pr.arcs[k];
        const float c = (ds + a.weight) + 0.0f;
        if (!(c < INF)) continue;
        const uint32_t e = nb_enc(c);
        if (e < l_dist[a.nextstate]) changed |= atomicMin(&l_dist[a.nextstate], e) > e;
      }
    }
    __syncthreads();
    if (!__any(changed)) break;
  }
  for (uint32_t s = lane; s < n; s += 64) dist2[s + 1] = nb_dec(l_dist[s]);

  // ---- 2. reverse: in-degrees, their exclusive scan, then the in-arcs in (source, position) order (one lane: the
  //         order IS the definition); the super-initial arcs: final states in state order
  for (uint32_t s = lane; s <= n + 1; s += 64) rcount[s] = 0;  // (in-degrees first, fill cursors next, then the search's counts)
  __syncthreads();
  for (uint32_t k = lane; k < pr.n_arcs; k += 64) atomicAdd(&rcount[pr.arcs[k].nextstate], 1u);
  __syncthreads();
  if (lane == 0) {
    uint32_t acc = 0;
    for (uint32_t t = 0; t < n; ++t) {  // in-arcs of original state t: rarcs[roff[t] .. roff[t + 1])
      const uint32_t c = rcount[t];
      roff[t] = acc;
      rcount[t] = acc;  // fill cursor
      acc += c;
    }
    roff[n] = acc;
    for (uint32_t s = 0; s < n; ++s)
      for (uint32_t k = pr.off[s]; k < pr.off[s + 1]; ++k) {
        const wfst_tr a = pr.arcs[k];
        rarcs[rcount[a.nextstate]++] = wfst_tr{a.ilabel, a.olabel, a.weight, s + 1};  // rfst: state i -> i + 1
      }
    for (uint32_t t = 0; t <= n + 1; ++t) rcount[t] = 0;
    uint32_t ns = 0;
    float d = INF;
    for (uint32_t s = 0; s < n; ++s) {
      const float fw = pr.finals[s];
      if (fw != INF) {  // a final weight that is Some (and not zero: +inf is the absence marker)
        super[ns++] = make_uint2(s + 1, __float_as_uint(fw));
        d = nb_plus(d, nb_times(fw, dist2[s + 1]));  // shortest_path.rs:143-153
      }
    }
    dist2[0] = d;
    s_n_super = ns;
    s_tree = 0;
    s_found = 0;
    s_status = NB_OK;
    s_facts = 0;
    s_pops = 0;
  }
  __syncthreads();
  if (nb_is_zero(dist2[0])) {  // the start state of rfst is unreachable: FO::new()
    finish_empty();
    return;
  }

  // ---- 4. n_shortest_path(rfst, distance_2, nshortest, delta): one lane, heap / keys / flags in LDS
  if (lane == 0) {
    uint32_t tree = 2, hsize = 0, facts = 0, status = NB_OK, found = 0, pops = 0;
    // ofst: state 0 = start, state 1 = final (weight One); pairs[1] = (Some(istart = 0), One)
    p_state[0] = NB_NONE;
    p_w[0] = INF;
    p_state[1] = 0;
    p_w[1] = 0.0f;
    auto key_of = [&](uint32_t st, float w) { return nb_times(st == NB_NONE ? 0.0f : dist2[st], w); };
    hk[1] = key_of(0, 0.0f);
    some[1] = 1;
    auto less = [&](uint32_t x, uint32_t y) {  // ShortestPathCompare::compare
      const float wx = hk[x], wy = hk[y];
      const bool sx = some[x] != 0, sy = some[y] != 0;
      const bool nl = nb_natural_less(wy, wx);
      if (!sx && sy) return nl || nb_approx_equal(wx, wy, delta);
      if (sx && !sy) return nl && !nb_approx_equal(wx, wy, delta);
      return nl;
    };
    auto push = [&](uint32_t v) {
      uint32_t idx = hsize++;
      heap[idx] = v;
      while (idx > 0) {
        const uint32_t parent = (idx - 1) / 2;
        if (!less(heap[parent], heap[idx])) break;
        const uint32_t t = heap[idx];
        heap[idx] = heap[parent];
        heap[parent] = t;
        idx = parent;
      }
    };
    auto pop = [&]() {
      const uint32_t top = heap[0];
      if (hsize == 1) {
        hsize = 0;
        return top;
      }
      heap[0] = heap[--hsize];
      uint32_t idx = 0;
      for (;;) {
        const uint32_t cur = heap[idx];
        const uint32_t c1 = 2 * idx + 1, c2 = 2 * idx + 2;
        uint32_t big;
        if (c1 >= hsize && c2 >= hsize) break;
        if (c1 < hsize && c2 >= hsize) big = c1;
        else if (less(heap[c1], heap[c2])) big = c2;
        else big = c1;
        if (less(heap[big], cur)) break;
        heap[idx] = heap[big];
        heap[big] = cur;
        idx = big;
      }
      return top;
    };
    auto new_state = [&](uint32_t st, float w, const wfst_tr& arc) -> bool {  // add_state + pairs.push + add_tr + heap.push
      if (tree >= T) {
        status = NB_TREE_FULL;
        return false;
      }
      const uint32_t next = tree++;
      p_state[next] = st;
      p_w[next] = w;
      hk[next] = key_of(st, w);
      some[next] = st != NB_NONE;
      o_arc[next] = arc;
      facts |= nb_facts(arc, next);
      push(next);
      return true;
    };
    push(1);
    const float limit = nb_times(dist2[0], INF);
    while (hsize && status == NB_OK) {
      const uint32_t state = pop();
      pops++;
      const uint32_t ps = p_state[state];
      const float pw = p_w[state];
      const uint32_t idx_r = ps == NB_NONE ? 0u : ps + 1u;  // p_first_real
      const float dd = ps == NB_NONE ? 0.0f : dist2[ps];
      if (nb_natural_less(limit, nb_times(dd, pw))) continue;
      const uint32_t rc = ++rcount[idx_r];
      if (ps == NB_NONE) {
        const wfst_tr sa{0u, 0u, 0.0f, state};
        facts |= nb_facts(sa, 0u);
        start_arc[found++] = state;
      }
      if (ps == NB_NONE && rc == nshortest) break;
      if (rc > nshortest) continue;
      if (ps == NB_NONE) continue;
      // the arcs of rfst state ps: the super-initial state's (one per final state), or the in-arcs of original state ps - 1
      if (ps == 0) {
        for (uint32_t i = 0; i < s_n_super && status == NB_OK; ++i) {
          const uint2 sa = super[i];
          const float w = __uint_as_float(sa.y);
          new_state(sa.x, nb_times(pw, w), wfst_tr{0u, 0u, w, state});
        }
      } else {
        for (uint32_t k = roff[ps - 1]; k < roff[ps] && status == NB_OK; ++k) {
          wfst_tr tr = rarcs[k];
          const uint32_t nxt = tr.nextstate;
          tr.nextstate = state;
          new_state(nxt, nb_times(pw, tr.weight), tr);
        }
      }
      // final weight of rfst state ps: only the original start state's image (weight One)
      if (status == NB_OK && ps == (uint32_t)pr.start + 1u) {
        const float fw = 0.0f;
        new_state(NB_NONE, nb_times(pw, fw), wfst_tr{0u, 0u, fw, state});
      }
    }
    s_tree = tree;
    s_found = found;
    s_status = status;
    s_facts = facts;
    s_pops = pops;
  }
  __syncthreads();
  const uint32_t tree = s_tree, found = s_found;
  if (s_status != NB_OK) {
    if (lane == 0) *out = NbOut{s_status, 0u, 0u, 0u, 0u, tree, s_pops, 0u, 0ull};
    return;
  }

  // ---- 5. connect: the accessible AND coaccessible states are the start state and the chains from the selected
  //         (None) states to the final state 1; stable renumbering = creation order.  keep / new ids live where the heap was.
  uint32_t* const newid = heap;
  __syncthreads();
  for (uint32_t k = lane; k < tree; k += 64) newid[k] = 0;
  __syncthreads();
  if (lane < found) {
    uint32_t k = start_arc[lane];
    while (k != 1u) {
      newid[k] = 1;  // (idempotent: chains share their tails)
      k = o_arc[k].nextstate;
    }
    newid[1] = 1;
    newid[0] = 1;
  }
  __syncthreads();
  // exclusive scan of the keep flags, 64 at a time
  uint32_t base = 0;
  for (uint32_t k0 = 0; k0 < tree; k0 += 64) {
    const uint32_t k = k0 + lane;
    const bool kp = k < tree && newid[k] != 0;
    const unsigned long long m = __ballot(kp);
    if (k < tree) newid[k] = kp ? base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) : NB_NONE;
    base += (uint32_t)__popcll(m);
  }
  __syncthreads();
  const uint32_t ns_out = base;                                   // 0 when no path was found (start not coaccessible)
  const uint32_t na_out = ns_out ? found + (ns_out - 2u) : 0u;    // start: `found` arcs; final: none; every other kept state: one
  unsigned long long po = 0;
  const size_t bytes = nb_al(4 * ((size_t)ns_out + 1)) + nb_al(4 * (size_t)ns_out) + nb_al(16 * (size_t)na_out);
  if (lane == 0) po = atomicAdd(cursor, (unsigned long long)bytes);
  po = __shfl(po, 0);
  if (po + bytes > payload_cap) {
    if (lane == 0) *out = NbOut{NB_OUT_FULL, 0u, 0u, 0u, 0u, tree, s_pops, 0u, 0ull};
    return;
  }
  uint32_t* const o_off = (uint32_t*)(payload + po);
  float* const o_fin = (float*)(payload + po + nb_al(4 * ((size_t)ns_out + 1)));
  wfst_tr* const o_arcs = (wfst_tr*)(payload + po + nb_al(4 * ((size_t)ns_out + 1)) + nb_al(4 * (size_t)ns_out));
  if (ns_out) {
    // state 0: its arcs in the order the paths were found; kept state k >= 2 with new id i: arc slot found + (i - 2)
    for (uint32_t i = lane; i < found; i += 64) o_arcs[i] = wfst_tr{0u, 0u, 0.0f, newid[start_arc[i]]};
    for (uint32_t k = lane; k < tree; k += 64) {
      const uint32_t i = newid[k];
      if (i == NB_NONE) continue;
      o_fin[i] = k == 1u ? 0.0f : INF;
      if (k == 0u) o_off[0] = 0;
      else if (k == 1u) o_off[1] = found;
      else {
        o_off[i] = found + (i - 2u);
        wfst_tr tr = o_arc[k];
        tr.nextstate = newid[tr.nextstate];
        o_arcs[found + (i - 2u)] = tr;
      }
    }
    if (lane == 0) {
      o_off[ns_out] = na_out;
      if (ns_out >= 3) o_off[2] = found;  // (state 1, the final state, has no arcs)
    }
  } else if (lane == 0) {
    o_off[0] = 0;
  }
  if (lane == 0) *out = NbOut{NB_OK, ns_out, na_out, s_facts, ns_out ? 1u : 0u, tree, s_pops, 0u, po};
}

}  // namespace

// shortest_path(nshortest > 1, unique = false) of n FSTs.  outs[i] = a new host-resident handle each.
void shortest_path_nbest_batch(wfst_ctx* ctx, const wfst_fst* const* fsts, size_t n, uint64_t nshortest, float delta, wfst_fst** outs) {
  for (size_t i = 0; i < n; ++i) outs[i] = nullptr;
  hipStream_t st = ctx->stream;
  int mode = 1;  // WFST_NBEST_DEVICE=0: always the host search; 1: the wave kernel where an input fits it
  if (const char* e = std::getenv("WFST_NBEST_DEVICE")) mode = std::atoi(e);
  if (ctx->batch_in_flight) mode = 0;  // (the fused batch in flight owns the context's pinned staging)
  std::vector<size_t> dev_idx;
  uint32_t max_n = 0, max_arcs = 0;
  if (mode && nshortest <= NB_MAX_PATHS)
    for (size_t i = 0; i < n; ++i) {
      const wfst_fst* f = fsts[i];
      if (f->n_states == 0 || f->start < 0) continue;  // (trivial: the host path returns the empty FST at once)
      if (f->n_states > NB_MAX_STATES || f->n_arcs > NB_MAX_ARCS) continue;
      ensure_device(const_cast<wfst_fst*>(f));  // (has_negative is worked out by the upload: a host-only handle reads false before it)
      if (f->has_negative) continue;
      dev_idx.push_back(i);
      max_n = std::max(max_n, f->n_states);
      max_arcs = std::max<uint32_t>(max_arcs, (uint32_t)f->n_arcs);
    }
  ctx->stats.nbest_device_problems = 0;
  if (!dev_idx.empty()) {
    // tree capacity: what n paths of about max_n states each can create, with room for side branches; the LDS it needs
    // (9 bytes per entry) bounds it
    uint32_t T = (uint32_t)std::min<uint64_t>(16384, std::max<uint64_t>(2048, 2 * nshortest * ((uint64_t)max_n + 8)));
    if (const char* e = std::getenv("WFST_NBEST_TREE")) T = (uint32_t)std::max(16, std::atoi(e));  // tests: force the fallback
    const size_t lds = nb_lds_bytes(max_n, T);
    if (lds <= 150 * 1024) {
      const size_t m = dev_idx.size();
      std::vector<NbProb> hp(m);
      size_t scratch_bytes = 0;
      for (size_t j = 0; j < m; ++j) {
        const wfst_fst* f = fsts[dev_idx[j]];
        hp[j] = NbProb{f->dev.offsets, f->dev.arcs, f->dev.finals, f->n_states, (uint32_t)f->n_arcs, (int32_t)f->start, 0u, scratch_bytes};
        scratch_bytes += nb_scratch_bytes(f->n_states, (uint32_t)f->n_arcs, T);
      }
      const size_t per_out = nb_al(4 * ((size_t)T + 1)) + nb_al(4 * (size_t)T) + nb_al(16 * (size_t)T);
      const size_t payload_cap = std::min<size_t>(m * per_out, (size_t)256 << 20);
      const size_t pin_bytes = nb_al(m * sizeof(NbProb)) + nb_al(m * sizeof(NbOut)) + payload_cap;
      char* pin = (char*)ctx->pinned_big.get(pin_bytes);
      NbProb* h_probs = (NbProb*)pin;
      NbOut* h_outs = (NbOut*)(pin + nb_al(m * sizeof(NbProb)));
      uint8_t* h_payload = (uint8_t*)(pin + nb_al(m * sizeof(NbProb)) + nb_al(m * sizeof(NbOut)));
      std::memcpy(h_probs, hp.data(), m * sizeof(NbProb));
      DBuf<uint8_t> scratch(*ctx->pool, scratch_bytes);
      DBuf<unsigned long long> cursor(*ctx->pool, 1);
      HIP_CHECK(hipMemsetAsync(cursor.p, 0, sizeof(unsigned long long), st));
      static std::once_flag lds_once[64];
      std::call_once(lds_once[(unsigned)ctx->device & 63u], [] {
        HIP_CHECK(hipFuncSetAttribute((const void*)nbest_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      });
      // (descriptors are read from, results written to, pinned host memory by the kernel itself: no copy commands)
      nbest_wave_kernel<<<(uint32_t)m, 64, lds, st>>>(h_probs, scratch.p, h_outs, h_payload, cursor.p, (unsigned long long)payload_cap,
                                                    (uint32_t)nshortest, delta, T);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(st));
      for (size_t j = 0; j < m; ++j) {
        const NbOut& o = h_outs[j];
        if (o.status != NB_OK) continue;  // does not fit: the host search below
        HostCsr h;
        const uint8_t* pl = h_payload + o.payload;
        const uint32_t* off = (const uint32_t*)pl;
        const float* fin = (const float*)(pl + nb_al(4 * ((size_t)o.n_states + 1)));
        const wfst_tr* arcs = (const wfst_tr*)(pl + nb_al(4 * ((size_t)o.n_states + 1)) + nb_al(4 * (size_t)o.n_states));
        if (o.n_states == 0) h.offsets.push_back(0);
        else h.offsets.assign(off, off + o.n_states + 1);
        h.finals.assign(fin, fin + o.n_states);
        h.arcs.assign(arcs, arcs + o.n_arcs);
        // the property word of the reference's mutation sequence (OutFst in nshortest.hip): add_state, set_start, add_state,
        // set_final(One), then every add_tr's facts folded (fst_props.h add_trs_by_facts: order-free), connect, and
        // shortest_path_properties(.., false).  An input without a reachable final state never gets that far: FO::new().
        uint64_t p = props::NULL_PROPS;
        if (o.tree_states) {
          p = props::add_state(p);
          p = props::set_start(p);
          p = props::add_state(p);
          const float one = 0.0f;
          p = props::set_final(p, nullptr, &one);
          if (o.tree_states > 2 || o.facts) p = props::add_state(props::add_trs_by_facts(p, o.facts));
          p = props::delete_states(p);
          p = (p & ~(props::ACCESSIBLE | props::COACCESSIBLE)) | props::ACCESSIBLE | props::COACCESSIBLE;
          p = props::shortest_path(p, false) & props::ALL;
        }
        outs[dev_idx[j]] = make_host_fst(ctx, o.n_states, o.has_start ? 0 : -1, p, std::move(h));
        ctx->stats.nbest_device_problems += 1;
      }
    }
  }
  for (size_t i = 0; i < n; ++i)
    if (!outs[i]) outs[i] = shortest_path_nbest(ctx, fsts[i], nshortest, delta);
}

namespace {

// ---------------------------------------------------------------- nshortest == 1 for many small FSTs
// single_shortest_path + backtrace (shortest_path.rs:173-282) of one small FST per wavefront, everything in one launch:
//   keys      (distance, hops) of every state in LDS, relaxed to the least fixed point (the canonical rule of sssp.hip:
//             among equal distances the fewer arcs) by label-correcting rounds over the arcs;
//   final     arg-min of d[s] (x) rho(s) over the final states, ties to the smaller state id;
//   parents   one pass over the arcs: the predecessor sssp_parent_kernel would select for every state (class, source,
//             position), kept in LDS next to the keys;
//   walk      one lane follows the parents from the final state; the arcs of the walk go to pinned memory in the order
//             build_path_fst expects (arc k enters the k-th created state).
// The result is what shortest_path_n1 returns for the same FST, bit for bit.  A lone solve through the relaxation kernels
// costs ~0.5 ms whatever the size (a dozen launches and three synchronisations): 64 composed lattices one after the
// other were 36 ms; here they are one launch.
constexpr uint32_t SP1_MAX_STATES = 4096;
constexpr unsigned long long SP1_KEY_INF = ~0ull;
constexpr unsigned long long SP1_PARENT_NONE = ~0ull;
struct Sp1Out {
  uint32_t has_path, hops, status, pad;
  float final_weight, total;
  uint64_t payload;  // byte offset of the walk's arcs in the pinned payload area
};
__device__ __forceinline__ unsigned long long sp1_parent_class(unsigned long long cand, unsigned long long key_s, unsigned long long key_t) {
  if (cand == key_t) return 0ull;  // (same rule as parent_class in sssp.hip)
  if ((cand >> 32) == (key_t >> 32) && key_s < key_t) return 1ull << 63;
  return SP1_PARENT_NONE;
}

// LDS: key[n] u64 | parent[n] u64 | (staged inputs) off[n + 1] u32 | wn[n_arcs] {weight bits, nextstate}
constexpr uint32_t SP1_STAGE_STATES = 2048, SP1_STAGE_ARCS = 4096;
__host__ __device__ inline bool sp1_staged(uint32_t n, uint32_t n_arcs) { return n <= SP1_STAGE_STATES && n_arcs <= SP1_STAGE_ARCS; }
__host__ __device__ inline size_t sp1_lds_bytes(uint32_t n, uint32_t n_arcs) {
  size_t b = 16 * (size_t)(n ? n : 1);
  if (sp1_staged(n, n_arcs)) b += nb_al(4 * ((size_t)n + 1)) + 8 * (size_t)n_arcs;
  return nb_al(b);
}

// mode 1 (export): stop after the relaxation and hand the distances AND the input's CSR to the host — dist[n] f32 |
// offsets[n + 1] | finals[n] | arcs[n_arcs], each 16-byte aligned, in the problem's payload slice — for host-side stages
// over many small inputs (the `unique` n-best branch: determinization) that would otherwise download them one by one.
__host__ __device__ inline size_t sp1_export_bytes(uint32_t n, uint32_t n_arcs) {
  return nb_al(4 * (size_t)n) + nb_al(4 * ((size_t)n + 1)) + nb_al(4 * (size_t)n) + 16 * (size_t)n_arcs;
}
__global__ void __launch_bounds__(64) sp1_wave_kernel(const NbProb* __restrict__ probs, Sp1Out* __restrict__ outs,
                                                      uint8_t* __restrict__ payload, uint32_t mode) {
  extern __shared__ __align__(16) unsigned char nb_lds[];
  const uint32_t lane = threadIdx.x;
  const NbProb pr = probs[blockIdx.x];
  const uint32_t n = pr.n;
  unsigned long long* const key = (unsigned long long*)nb_lds;  // [n]
  unsigned long long* const parent = key + n;                   // [n]  (during the relaxation: the key a state was last expanded with)
  uint32_t* const l_off = (uint32_t*)(parent + n);              // [n + 1]   staged inputs only
  uint2* const l_wn = (uint2*)((unsigned char*)l_off + nb_al(4 * ((size_t)n + 1)));  // [n_arcs]
  wfst_tr* const path = (wfst_tr*)(payload + pr.scratch);
  Sp1Out o{0u, 0u, 0u, 0u, INF, INF, pr.scratch};
  if (pr.start < 0 || n == 0) {
    if (lane == 0) outs[blockIdx.x] = o;
    return;
  }
  const bool staged = sp1_staged(n, pr.n_arcs);
  for (uint32_t s = lane; s < n; s += 64) {
    key[s] = s == (uint32_t)pr.start ? ((unsigned long long)nb_enc(0.0f) << 32) : SP1_KEY_INF;
    parent[s] = SP1_KEY_INF;
  }
  if (staged) {
    for (uint32_t s = lane; s <= n; s += 64) l_off[s] = pr.off[s];
    for (uint32_t k = lane; k < pr.n_arcs; k += 64) {
      const uint4 a = reinterpret_cast<const uint4*>(pr.arcs)[k];
      l_wn[k] = make_uint2(a.z, a.w);
    }
  }
  __syncthreads();
  if (staged) {
    // Offsets and arcs in LDS: the states are taken IN ORDER by the whole wave (lanes over the arcs of one state), so a
    // sweep carries an improvement along every forward arc at once — a lattice numbered in BFS order needs one sweep and
    // a second that finds nothing to do (a state is expanded again only if its key changed since it last was).
    for (uint32_t sweep = 0;; ++sweep) {
      if (sweep > n + 1u) {  // (more sweeps than states: only a negative cycle does that; the driver excludes negative weights)
        o.status = 2u;
        break;
      }
      bool changed = false;
      for (uint32_t s = 0; s < n; ++s) {
        const unsigned long long ks = key[s];
        if (ks != SP1_KEY_INF && ks != parent[s]) {  // (uniform: every lane reads the same words)
          if (lane == 0) parent[s] = ks;
          const float ds = nb_dec((uint32_t)(ks >> 32));
          const uint32_t h1 = (uint32_t)ks + 1u;
          for (uint32_t k = l_off[s] + lane; k < l_off[s + 1]; k += 64) {
            const uint2 a = l_wn[k];
            const float c = (ds + __uint_as_float(a.x)) + 0.0f;  // w1 (x) w2 (tropical_weight.rs:60-70)
            if (!(c < INF)) continue;                            // +inf never improves (shortest_path.rs:226)
            const unsigned long long ck = ((unsigned long long)nb_enc(c) << 32) | h1;
            if (ck < key[a.y]) changed |= atomicMin(&key[a.y], ck) > ck;
          }
        }
        __syncthreads();  // (one wave: this state's LDS atomics are done before the next state's key is read)
      }
      if (!__any(changed)) break;
    }
  } else {
    for (uint32_t round = 0;; ++round) {  // label-correcting rounds over the arcs in memory, lanes over the states
      if (round > n + 1u) {
        o.status = 2u;
        break;
      }
      bool changed = false;
      for (uint32_t s = lane; s < n; s += 64) {
        const unsigned long long ks = key[s];
        if (ks == SP1_KEY_INF) continue;
        const float ds = nb_dec((uint32_t)(ks >> 32));
        const uint32_t h1 = (uint32_t)ks + 1u;
        for (uint32_t k = pr.off[s]; k < pr.off[s + 1]; ++k) {
          const wfst_tr a = pr.arcs[k];
          const float c = (ds + a.weight) + 0.0f;
          if (!(c < INF)) continue;
          const unsigned long long ck = ((unsigned long long)nb_enc(c) << 32) | h1;
          if (ck < key[a.nextstate]) changed |= atomicMin(&key[a.nextstate], ck) > ck;
        }
      }
      __syncthreads();
      if (!__any(changed)) break;
    }
  }
  if (o.status != 0u) {  // did not converge: left to the single-FST path (which reports it)
    if (lane == 0) outs[blockIdx.x] = o;
    return;
  }
  if (mode == 1u) {
    uint8_t* pl = payload + pr.scratch;
    float* const e_dist = (float*)pl;
    pl += nb_al(4 * (size_t)n);
    uint32_t* const e_off = (uint32_t*)pl;
    pl += nb_al(4 * ((size_t)n + 1));
    float* const e_fin = (float*)pl;
    pl += nb_al(4 * (size_t)n);
    uint4* const e_arcs = (uint4*)pl;
    for (uint32_t s = lane; s < n; s += 64) {
      e_dist[s] = key[s] == SP1_KEY_INF ? INF : nb_dec((uint32_t)(key[s] >> 32));
      e_fin[s] = pr.finals[s];
    }
    for (uint32_t s = lane; s <= n; s += 64) e_off[s] = pr.off[s];
    for (uint32_t k = lane; k < pr.n_arcs; k += 64) e_arcs[k] = reinterpret_cast<const uint4*>(pr.arcs)[k];
    o.has_path = 1u;
    if (lane == 0) outs[blockIdx.x] = o;
    return;
  }
  // final state: d[s] (x) rho(s) (shortest_path.rs:214-220), ties to the smaller state
  unsigned long long best = SP1_KEY_INF;
  for (uint32_t s = lane; s < n; s += 64) {
    parent[s] = SP1_PARENT_NONE;
    const float f = pr.finals[s];
    const unsigned long long ks = key[s];
    if (!(f < INF) || ks == SP1_KEY_INF) continue;
    const float tot = (nb_dec((uint32_t)(ks >> 32)) + f) + 0.0f;
    if (!(tot < INF)) continue;
    const unsigned long long c = ((unsigned long long)nb_enc(tot) << 32) | s;
    best = c < best ? c : best;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long x = __shfl_xor(best, d);
    best = x < best ? x : best;
  }
  if (best == SP1_KEY_INF) {
    if (lane == 0) outs[blockIdx.x] = o;
    return;
  }
  __syncthreads();
  // parents: min (class, source, position) over the arcs the rule admits
  for (uint32_t s = lane; s < n; s += 64) {
    const unsigned long long ks = key[s];
    if (ks == SP1_KEY_INF) continue;
    const float ds = nb_dec((uint32_t)(ks >> 32));
    const uint32_t h1 = (uint32_t)ks + 1u, b = staged ? l_off[s] : pr.off[s], e = staged ? l_off[s + 1] : pr.off[s + 1];
    for (uint32_t k = b; k < e; ++k) {
      uint2 a;
      if (staged) {
        a = l_wn[k];
      } else {
        const wfst_tr t = pr.arcs[k];
        a = make_uint2(__float_as_uint(t.weight), t.nextstate);
      }
      const float c = (ds + __uint_as_float(a.x)) + 0.0f;
      if (!(c < INF)) continue;
      const unsigned long long ck = ((unsigned long long)nb_enc(c) << 32) | h1;
      const unsigned long long cls = sp1_parent_class(ck, ks, key[a.y]);
      if (cls != SP1_PARENT_NONE) atomicMin(&parent[a.y], cls | ((unsigned long long)s << 32) | (k - b));
    }
  }
  __syncthreads();
  const uint32_t fp = (uint32_t)best;
  o.has_path = 1u;
  o.final_weight = pr.finals[fp];
  o.total = nb_dec((uint32_t)(best >> 32));
  // the walk: one lane follows the parents (LDS) and notes the arc index of every step where the keys were (they are not
  // needed any more: the start state is where the walk ends); the arcs themselves are then fetched by all lanes at once
  // and go to pinned memory in the order make_path_fst expects (arc k enters the k-th created state)
  uint32_t* const steps = (uint32_t*)key;  // [<= n]
  uint32_t hops = 0, status = 0;
  if (lane == 0) {
    uint32_t cur = fp;
    while (cur != (uint32_t)pr.start) {
      const unsigned long long p = parent[cur];
      if (p == SP1_PARENT_NONE || hops >= n) {  // (cannot happen at a fixed point of non-negative weights: reported, not followed)
        status = 1u;
        break;
      }
      const uint32_t s = (uint32_t)(p >> 32) & 0x7FFFFFFFu, pos = (uint32_t)p;
      steps[hops] = (staged ? l_off[s] : pr.off[s]) + pos;
      cur = s;
      ++hops;
    }
  }
  hops = __shfl(hops, 0);
  status = __shfl(status, 0);
  __syncthreads();
  if (status == 0u)
    for (uint32_t k = lane; k < hops; k += 64) {
      wfst_tr tr = pr.arcs[steps[k]];
      tr.nextstate = k;
      path[k] = tr;
    }
  o.hops = hops;
  o.status = status;
  if (lane == 0) outs[blockIdx.x] = o;
}

}  // namespace

// Forward distances (the exact fixed point) and the CSR of the small inputs idx[..] of fsts, all in ONE launch and one
// synchronisation; false for an input the kernel handed back.  Inputs must have a start state and at most SP1_MAX_STATES
// states / 4 * SP1_MAX_STATES arcs, no negative weights, and a device copy.
void export_small_with_distances(wfst_ctx* ctx, const wfst_fst* const* fsts, const std::vector<size_t>& idx,
                                 std::vector<SmallFstExport>& out) {
  const size_t m = idx.size();
  out.assign(m, SmallFstExport{});
  if (m == 0) return;
  hipStream_t st = ctx->stream;
  std::vector<NbProb> hp(m);
  size_t pay = 0, lds = 0;
  for (size_t j = 0; j < m; ++j) {
    const wfst_fst* f = fsts[idx[j]];
    hp[j] = NbProb{f->dev.offsets, f->dev.arcs, f->dev.finals, f->n_states, (uint32_t)f->n_arcs, (int32_t)f->start, 0u, pay};
    pay += nb_al(sp1_export_bytes(f->n_states, (uint32_t)f->n_arcs));
    lds = std::max(lds, sp1_lds_bytes(f->n_states, (uint32_t)f->n_arcs));
  }
  const size_t pin_bytes = nb_al(m * sizeof(NbProb)) + nb_al(m * sizeof(Sp1Out)) + pay;
  char* pin = (char*)ctx->pinned_big.get(pin_bytes);
  NbProb* h_probs = (NbProb*)pin;
  Sp1Out* h_outs = (Sp1Out*)(pin + nb_al(m * sizeof(NbProb)));
  uint8_t* h_payload = (uint8_t*)(pin + nb_al(m * sizeof(NbProb)) + nb_al(m * sizeof(Sp1Out)));
  std::memcpy(h_probs, hp.data(), m * sizeof(NbProb));
  static std::once_flag lds_once[64];
  std::call_once(lds_once[(unsigned)ctx->device & 63u], [] {
    HIP_CHECK(hipFuncSetAttribute((const void*)sp1_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  });
  sp1_wave_kernel<<<(uint32_t)m, 64, lds, st>>>(h_probs, h_outs, h_payload, 1u);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(st));
  for (size_t j = 0; j < m; ++j) {
    const Sp1Out& o = h_outs[j];
    if (o.status != 0u || !o.has_path) continue;
    const wfst_fst* f = fsts[idx[j]];
    const uint32_t n = f->n_states, na = (uint32_t)f->n_arcs;
    const uint8_t* pl = h_payload + o.payload;
    SmallFstExport& e = out[j];
    e.dist.assign((const float*)pl, (const float*)pl + n);
    pl += nb_al(4 * (size_t)n);
    e.csr.offsets.assign((const uint32_t*)pl, (const uint32_t*)pl + n + 1);
    pl += nb_al(4 * ((size_t)n + 1));
    e.csr.finals.assign((const float*)pl, (const float*)pl + n);
    pl += nb_al(4 * (size_t)n);
    e.csr.arcs.assign((const wfst_tr*)pl, (const wfst_tr*)pl + na);
    e.ok = true;
  }
}

// A lone tiny FST (the two-step route on small lattices: compose, then shortest_path) is one wavefront's work too: ~0.1 ms
// instead of the ~0.5 ms of launches and synchronisations the relaxation kernels cost whatever the size.  Larger inputs
// would keep ONE wave busy for longer than the GPU-wide kernels take.  (Not when a test pins a relaxation kernel.)
bool shortest_path_n1_tiny(wfst_ctx* ctx, const wfst_fst* f, wfst_fst** out) {
  constexpr uint64_t TINY_ARCS = 2048;
  if (f->n_states > SP1_MAX_STATES || f->n_arcs > TINY_ARCS || std::getenv("WFST_SSSP_MAILBOX") || std::getenv("WFST_SSSP_DELTA")) return false;
  *out = nullptr;
  shortest_path_n1_batch(ctx, &f, 1, out, /*lone=*/true);
  return *out != nullptr;
}

void shortest_path_n1_batch(wfst_ctx* ctx, const wfst_fst* const* fsts, size_t n, wfst_fst** outs, bool lone) {
  hipStream_t st = ctx->stream;
  std::vector<size_t> idx;
  uint32_t max_n = 0;
  // (a fused batch in flight on this context owns its pinned staging: every input takes the general path then)
  const bool allow = (!std::getenv("WFST_SP1_DEVICE") || std::atoi(std::getenv("WFST_SP1_DEVICE")) != 0) && !ctx->batch_in_flight;
  for (size_t i = 0; i < n; ++i) {
    const wfst_fst* f = fsts[i];
    outs[i] = nullptr;
    // (the reference tie order is a property of the single-FST path; negative weights keep its convergence guard)
    if (!allow || ctx->tie_reference || f->n_states > SP1_MAX_STATES || f->n_arcs > 4ull * SP1_MAX_STATES || (n < 2 && !lone)) continue;
    ensure_device(const_cast<wfst_fst*>(f));
    if (f->has_negative) continue;
    idx.push_back(i);
    max_n = std::max(max_n, f->n_states);
  }
  if (!idx.empty()) {
    const size_t m = idx.size();
    size_t lds = 0;
    for (size_t i : idx) lds = std::max(lds, sp1_lds_bytes(fsts[i]->n_states, (uint32_t)fsts[i]->n_arcs));
    std::vector<NbProb> hp(m);
    size_t pay = 0;
    for (size_t j = 0; j < m; ++j) {
      const wfst_fst* f = fsts[idx[j]];
      hp[j] = NbProb{f->dev.offsets, f->dev.arcs, f->dev.finals, f->n_states, (uint32_t)f->n_arcs, (int32_t)f->start, 0u, pay};
      pay += nb_al(16 * (size_t)std::max<uint32_t>(f->n_states, 1));  // a shortest path visits a state once
    }
    const size_t pin_bytes = nb_al(m * sizeof(NbProb)) + nb_al(m * sizeof(Sp1Out)) + pay;
    char* pin = (char*)ctx->pinned_big.get(pin_bytes);
    NbProb* h_probs = (NbProb*)pin;
    Sp1Out* h_outs = (Sp1Out*)(pin + nb_al(m * sizeof(NbProb)));
    uint8_t* h_payload = (uint8_t*)(pin + nb_al(m * sizeof(NbProb)) + nb_al(m * sizeof(Sp1Out)));
    std::memcpy(h_probs, hp.data(), m * sizeof(NbProb));
    static std::once_flag lds_once[64];
    std::call_once(lds_once[(unsigned)ctx->device & 63u], [] {
      HIP_CHECK(hipFuncSetAttribute((const void*)sp1_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    });
    // (descriptors read from, results and path arcs written to pinned host memory by the kernel itself)
    sp1_wave_kernel<<<(uint32_t)m, 64, lds, st>>>(h_probs, h_outs, h_payload, 0u);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st));
    for (size_t j = 0; j < m; ++j) {
      const Sp1Out& o = h_outs[j];
      if (o.status != 0u) continue;  // the single-FST path below
      outs[idx[j]] = make_path_fst(ctx, o.has_path != 0, o.hops, o.final_weight, (const wfst_tr*)(h_payload + o.payload));
    }
  }
  if (lone) return;  // (the caller continues on the relaxation kernels)
  for (size_t i = 0; i < n; ++i)
    if (!outs[i]) outs[i] = shortest_path_n1(ctx, fsts[i]);
}

}  // namespace wfst
