// rm_epsilon.hip — epsilon removal of an HBM-resident FST.
//
// Replaces rustfst::algorithms::rm_epsilon with its default configuration (connect, no thresholds):
//   rm_epsilon / rm_epsilon_with_internal_config        algorithms/rm_epsilon/rm_epsilon_static.rs:50-163
//   RmEpsilonState::expand                              algorithms/rm_epsilon/rm_epsilon_state.rs:44-119
//   ShortestDistanceState (EpsilonTrFilter, retain)     algorithms/shortest_distance.rs:153-237
//   rmepsilon_properties                                fst_properties/mutate_properties.rs:646-660
//   connect                                             algorithms/connect.rs:51-66 (compose_wide.hip: connect_and_adopt)
//
// What the reference does: states are visited in reverse topological order of the epsilon graph (epsilon:epsilon arcs);
// every state s with a non-epsilon incoming arc (or the start state) is REWRITTEN in place:
//   * d[q] = (min,+) distance from s to every q of its epsilon closure;
//   * the closure is walked depth first with an explicit stack (pop, mark, push the unvisited epsilon targets in arc
//     order); every non-epsilon arc (il, ol, w, ns) of a visited q becomes (il, ol, d[q] (x) w, ns), arcs with the same
//     (il, ol, ns) are (+)-combined at the position of the first one; the list is then reversed;
//   * final(s) = (+) over the closure of d[q] (x) final(q).
// The walk reads the CURRENT arc lists: an epsilon successor that was rewritten earlier is seen through its new arcs (no
// epsilon arcs left, weights already combined, its own order), so results depend on that order.  What is independent:
// states of the same epsilon depth (longest epsilon path to a sink) outside epsilon cycles — one launch per depth, ONE
// THREAD PER STATE (closures are a handful of states), closure / stack / arc list in a slice of a scratch buffer, a state
// that outgrows its slice redone with four times the room.  States on an epsilon cycle are rewritten one by one in
// increasing id order inside their component, which is the reference's order (rm_epsilon_static.rs:108-127 reversed).
// The epsilon graph, its components and depths are computed on the host (one pass over the arcs, like the reference's
// own DFS).  States without a non-epsilon incoming arc lose their arcs; then connect.
// A state whose closure outgrows 64 states is redone by rm_expand_wave: ONE WAVE per state, closure membership and the
// (ilabel, olabel, nextstate) -> arc lookup through open-addressing tables in the state's scratch slice, the closure
// distances by wave-parallel label correcting, the depth-first walk itself sequential (it defines the arc order) with a
// visited state's arcs handled 64 at a time — no limit on the closure size, work linear in it (redone with four times
// the room on overflow, like the one-thread kernel).  Finished arcs are copied into compact per-batch arenas and the
// scratch of every attempt is released at once, so memory follows the size of the output.
// The reference relaxes only improvements larger than delta = 1e-6 (approx_equal, shortest_distance.rs:216); d[] here is
// the exact minimum of the left-folded f32 path sums, the same whenever weights differ by more than 1e-6 (any 1/512-grid
// input) — the deviation already documented for shortest_distance.
#include <algorithm>

#include <rocprim/device/device_scan.hpp>

#include "common.h"
#include "fst_props.h"

namespace wfst {

namespace {

struct RmCaps {
  uint32_t C;  // closure states
  uint32_t K;  // depth-first stack entries
  uint32_t A;  // arcs of the rewritten state
};
__host__ __device__ inline size_t rm_arcs_offset(const RmCaps& c) { return ((size_t)c.C * 12 + (size_t)c.K * 4 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t rm_slice_bytes(const RmCaps& c) { return rm_arcs_offset(c) + (size_t)c.A * 16; }

__device__ __forceinline__ float wtimes(float a, float b) { return a == INF ? a : (b == INF ? b : a + b); }
__device__ __forceinline__ bool is_eps(const wfst_tr& t) { return t.ilabel == 0u && t.olabel == 0u; }  // EpsilonTrFilter

struct RmView {  // the FST as the reference's loop sees it at this moment
  const uint32_t* offsets;
  const wfst_tr* arcs;
  const uint32_t* done;                 // state already rewritten
  const uint32_t* cnt;                  // its new arc count
  const unsigned long long* arc_ptr;    // and where its new arcs are
  __device__ const wfst_tr* trs(uint32_t q, uint32_t* n) const {
    if (done[q]) {
      *n = cnt[q];
      return (const wfst_tr*)arc_ptr[q];
    }
    *n = offsets[q + 1] - offsets[q];
    return arcs + offsets[q];
  }
};

// RmEpsilonState::expand + the rewrite of the listed states; status[i]: 0 done, 1 slice too small
__global__ void rm_expand(RmView v, const uint32_t* __restrict__ list, uint32_t n_list, RmCaps caps, char* __restrict__ scratch,
                          uint32_t* __restrict__ new_cnt, float* __restrict__ new_fin, unsigned long long* __restrict__ new_ptr,
                          const float* __restrict__ fin, uint32_t* __restrict__ status) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_list) return;
  const uint32_t s = list[i];
  char* slice = scratch + (size_t)i * rm_slice_bytes(caps);
  uint32_t* cl = (uint32_t*)slice;  // closure states, in discovery order
  float* dist = (float*)(slice + (size_t)caps.C * 4);
  uint32_t* vis = (uint32_t*)(slice + (size_t)caps.C * 8);
  uint32_t* stack = (uint32_t*)(slice + (size_t)caps.C * 12);
  wfst_tr* out = (wfst_tr*)(slice + rm_arcs_offset(caps));
  status[i] = 1u;  // (until done)
  // 1. closure and distances over the epsilon arcs as they are now
  uint32_t nc = 1;
  cl[0] = s;
  dist[0] = 0.0f;
  for (uint32_t iter = 0;; ++iter) {
    bool changed = false;
    for (uint32_t k = 0; k < nc; ++k) {
      const float dk = dist[k];
      uint32_t nq;
      const wfst_tr* tq = v.trs(cl[k], &nq);
      for (uint32_t a = 0; a < nq; ++a) {
        const wfst_tr tr = tq[a];
        if (!is_eps(tr)) continue;
        uint32_t j = 0;
        while (j < nc && cl[j] != tr.nextstate) ++j;
        if (j == nc) {
          if (nc == caps.C) return;
          cl[nc] = tr.nextstate;
          dist[nc] = INF;
          ++nc;
          changed = true;
        }
        const float cand = wtimes(dk, tr.weight);
        if (cand < dist[j]) {
          dist[j] = cand;
          changed = true;
        }
      }
    }
    if (!changed) break;
    if (iter > nc + 1u) break;  // a negative epsilon cycle: the reference would not terminate either; stop improving
  }
  // 2. the depth-first walk of the closure: arcs and the final weight in visiting order
  for (uint32_t k = 0; k < nc; ++k) vis[k] = 0u;
  uint32_t sp = 0, na = 0;
  stack[sp++] = 0u;  // (indices into cl)
  float final_w = INF;
  while (sp) {
    const uint32_t k = stack[--sp];
    if (vis[k]) continue;
    vis[k] = 1u;
    const uint32_t q = cl[k];
    const float dq = dist[k];
    uint32_t nq;
    const wfst_tr* tq = v.trs(q, &nq);
    for (uint32_t a = 0; a < nq; ++a) {
      wfst_tr tr = tq[a];
      tr.weight = wtimes(dq, tr.weight);
      if (is_eps(tr)) {
        uint32_t j = 0;
        while (cl[j] != tr.nextstate) ++j;  // (in the closure since step 1)
        if (!vis[j]) {
          if (sp == caps.K) return;
          stack[sp++] = j;
        }
      } else {
        uint32_t j = 0;
        while (j < na && !(out[j].ilabel == tr.ilabel && out[j].olabel == tr.olabel && out[j].nextstate == tr.nextstate)) ++j;
        if (j < na) {
          if (tr.weight < out[j].weight) out[j].weight = tr.weight;  // plus_assign at the first occurrence
        } else {
          if (na == caps.A) return;
          out[na++] = tr;
        }
      }
    }
    const float f = wtimes(dq, fin[q]);
    final_w = f < final_w ? f : final_w;
  }
  for (uint32_t a = 0; a < na / 2; ++a) {  // trs.into_iter().rev() (rm_epsilon_static.rs:125)
    const wfst_tr t = out[a];
    out[a] = out[na - 1 - a];
    out[na - 1 - a] = t;
  }
  // (published to later launches by rm_publish: nothing of this launch may read it)
  new_cnt[i] = na;
  new_fin[i] = final_w;
  new_ptr[i] = (unsigned long long)out;
  status[i] = 0u;
}

// ---- big closures: one wave per state -----------------------------------------------------------------------------
struct RmBigCaps {
  uint32_t C;   // closure states (table of 2C slots)
  uint32_t K;   // depth-first stack entries
  uint32_t A;   // arcs of the rewritten state (table of 2A slots)
};
__host__ __device__ inline uint32_t rm_pow2_at_least(uint32_t x) {
  uint32_t p = 16;
  while (p < x) p <<= 1;
  return p;
}
__host__ __device__ inline size_t rm_big_arcs_offset(const RmBigCaps& c) {
  const size_t words = (size_t)c.C * 3 + 2ull * rm_pow2_at_least(c.C) + c.K + 2ull * rm_pow2_at_least(c.A) + 16;
  return (words * 4 + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t rm_big_slice_bytes(const RmBigCaps& c) { return rm_big_arcs_offset(c) + (size_t)c.A * 16; }

__device__ __forceinline__ uint32_t rm_hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t rm_enc(float f) {  // order-preserving bits (weights may be negative)
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float rm_dec(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e); }

// closure index of state t (0xFFFFFFFF if absent); the table holds index + 1
__device__ __forceinline__ uint32_t rm_cl_find(const uint32_t* __restrict__ htab, uint32_t hmask, const uint32_t* __restrict__ cl, uint32_t t) {
  for (uint32_t p = rm_hash32(t) & hmask;; p = (p + 1) & hmask) {
    const uint32_t v = __hip_atomic_load(&htab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (v == 0u) return 0xFFFFFFFFu;
    if (v != 0xFFFFFFFFu && cl[v - 1] == t) return v - 1;
  }
}

// RmEpsilonState::expand for the listed states, one wave each; status[i]: 0 done, 1 slice too small
__global__ void __launch_bounds__(64) rm_expand_wave(RmView v, const uint32_t* __restrict__ list, uint32_t n_list, RmBigCaps caps,
                                                     char* __restrict__ scratch, uint32_t* __restrict__ new_cnt,
                                                     float* __restrict__ new_fin, unsigned long long* __restrict__ new_ptr,
                                                     const float* __restrict__ fin, uint32_t* __restrict__ status) {
  const uint32_t i = blockIdx.x, lane = threadIdx.x;
  if (i >= n_list) return;
  const uint32_t s = list[i];
  char* slice = scratch + (size_t)i * rm_big_slice_bytes(caps);
  const uint32_t H = 2u * rm_pow2_at_least(caps.C), hmask = H - 1u, HA = 2u * rm_pow2_at_least(caps.A), amask = HA - 1u;
  uint32_t* cl = (uint32_t*)slice;          // closure states
  uint32_t* dist = cl + caps.C;             // ordered bits of the closure distances
  uint32_t* vis = dist + caps.C;
  uint32_t* htab = vis + caps.C;            // state -> closure index + 1 (0 empty, ~0 being filled)
  uint32_t* stack = htab + H;
  uint32_t* otab = stack + caps.K;          // (ilabel, olabel, nextstate) -> arc index + 1
  uint32_t* hdr = otab + HA;                // [0] closure size, [1] "changed" flag
  wfst_tr* out = (wfst_tr*)(slice + rm_big_arcs_offset(caps));
  if (lane == 0) status[i] = 1u;  // (until done)
  for (uint32_t k = lane; k < caps.C; k += 64) {
    dist[k] = 0xFFFFFFFFu;
    vis[k] = 0u;
  }
  for (uint32_t k = lane; k < H; k += 64) htab[k] = 0u;
  for (uint32_t k = lane; k < HA; k += 64) otab[k] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  if (lane == 0) {
    cl[0] = s;
    dist[0] = rm_enc(0.0f);
    htab[rm_hash32(s) & hmask] = 1u;
    hdr[0] = 1u;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_s_barrier();
  // 1. closure and distances: label correcting over the epsilon arcs as they are now; every lane takes closure entries
  bool overflow = false;
  for (uint32_t iter = 0;; ++iter) {
    const uint32_t n0 = __hip_atomic_load(&hdr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    bool changed = false;
    for (uint32_t base = 0; base < n0; base += 64) {
      const uint32_t k = base + lane;
      uint32_t nq = 0;
      const wfst_tr* tq = nullptr;
      float dk = INF;
      if (k < n0) {
        tq = v.trs(cl[k], &nq);
        dk = rm_dec(__hip_atomic_load(&dist[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
      }
      uint32_t a = 0;
      // a lane that must insert a new closure state claims a table slot; the loop is wave-uniform so that nobody spins on
      // a lane that cannot run
      uint32_t pend_t = 0xFFFFFFFFu;  // target waiting for its closure index
      float pend_w = 0.0f;
      while (__any(a < nq || pend_t != 0xFFFFFFFFu)) {
        if (pend_t == 0xFFFFFFFFu && a < nq) {
          const wfst_tr tr = tq[a++];
          if (is_eps(tr)) {
            pend_t = tr.nextstate;
            pend_w = tr.weight;
          }
        }
        if (pend_t != 0xFFFFFFFFu) {
          // find or insert pend_t
          uint32_t j = 0xFFFFFFFFu;
          for (uint32_t p = rm_hash32(pend_t) & hmask;; p = (p + 1) & hmask) {
            uint32_t hv = __hip_atomic_load(&htab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (hv == 0u) {
              uint32_t expect = 0u;
              if (__hip_atomic_compare_exchange_strong(&htab[p], &expect, 0xFFFFFFFFu, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WAVEFRONT)) {
                const uint32_t idx = __hip_atomic_fetch_add(&hdr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (idx >= caps.C) {
                  overflow = true;
                  __hip_atomic_store(&htab[p], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                  j = 0xFFFFFFFEu;  // give up on this arc
                } else {
                  cl[idx] = pend_t;
                  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                  __hip_atomic_store(&htab[p], idx + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                  j = idx;
                  changed = true;
                }
                break;
              }
              hv = expect;  // somebody else took the slot: look at what is in it
            }
            if (hv == 0xFFFFFFFFu) break;  // being filled by another lane: retry on the next trip of the uniform loop
            if (cl[hv - 1u] == pend_t) {
              j = hv - 1u;
              break;
            }
          }
          if (j != 0xFFFFFFFFu) {
            if (j != 0xFFFFFFFEu) {
              const uint32_t cand = rm_enc(wtimes(dk, pend_w));
              if (cand < __hip_atomic_fetch_min(&dist[j], cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)) changed = true;
            }
            pend_t = 0xFFFFFFFFu;
          }
        }
      }
    }
    if (__any(overflow)) return;
    if (!__any(changed)) break;
    if (iter > __hip_atomic_load(&hdr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) + 1u) break;  // negative epsilon cycle
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  // 2. the depth-first walk of the closure (sequential: it defines the arc order); a visited state's arcs 64 at a time
  uint32_t sp = 0, na = 0;  // wave-uniform
  if (lane == 0) stack[0] = 0u;
  sp = 1;
  float final_w = INF;
  while (sp) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint32_t k = stack[sp - 1];
    --sp;
    if (vis[k]) continue;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0) vis[k] = 1u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint32_t q = cl[k];
    const float dq = rm_dec(dist[k]);
    uint32_t nq;
    const wfst_tr* tq = v.trs(q, &nq);
    for (uint32_t a0 = 0; a0 < nq; a0 += 64) {
      const bool live = a0 + lane < nq;
      wfst_tr tr{};
      if (live) {
        tr = tq[a0 + lane];
        tr.weight = wtimes(dq, tr.weight);
      }
      const bool eps = live && is_eps(tr);
      // epsilon arcs: unvisited targets go onto the stack in arc order
      uint32_t j = 0;
      bool push = false;
      if (eps) {
        j = rm_cl_find(htab, hmask, cl, tr.nextstate);  // (in the closure since step 1)
        push = vis[j] == 0u;
      }
      const unsigned long long pm = __ballot(push);
      if (pm) {
        const uint32_t cnt = (uint32_t)__popcll(pm);
        if (sp + cnt > caps.K) return;
        if (push) stack[sp + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull))] = j;
        sp += cnt;
      }
      // other arcs: (+)-combined at the first arc with the same (ilabel, olabel, nextstate), appended otherwise — one
      // distinct new key per trip of the loop, in arc order (the lowest waiting lane leads)
      bool wait = live && !eps;
      while (const unsigned long long wm = __ballot(wait)) {
        const int lead = __ffsll((unsigned long long)wm) - 1;
        const uint32_t il = __shfl(tr.ilabel, lead), ol = __shfl(tr.olabel, lead), ns = __shfl(tr.nextstate, lead);
        const bool same = wait && tr.ilabel == il && tr.olabel == ol && tr.nextstate == ns;
        // the group's weight: min over its lanes
        uint32_t wbits = same ? rm_enc(tr.weight) : 0xFFFFFFFFu;
        for (int d = 32; d >= 1; d >>= 1) wbits = min(wbits, (uint32_t)__shfl_xor(wbits, d));
        uint32_t found = 0xFFFFFFFFu;  // index of this key's arc after the leader's step
        if ((int)lane == lead) {
          uint32_t p = (rm_hash32(il) ^ rm_hash32(ol * 0x9E3779B9u + ns)) & amask;
          uint32_t ov;
          for (;; p = (p + 1) & amask) {
            ov = otab[p];
            if (ov == 0u) break;
            const wfst_tr o = out[ov - 1u];
            if (o.ilabel == il && o.olabel == ol && o.nextstate == ns) break;
          }
          const float w = rm_dec(wbits);
          if (ov) {
            if (w < out[ov - 1u].weight) out[ov - 1u].weight = w;  // plus_assign at the first occurrence
            found = ov - 1u;
          } else if (na < caps.A) {
            otab[p] = na + 1u;
            out[na] = wfst_tr{il, ol, w, ns};
            found = na;
          }
        }
        found = __shfl(found, lead);
        if (found == 0xFFFFFFFFu) return;  // no room for the arc: redone with a larger slice
        if (found == na) na += 1;          // (an existing arc has a smaller index)
        wait = wait && !same;
      }
    }
    const float f = wtimes(dq, fin[q]);
    final_w = f < final_w ? f : final_w;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (uint32_t a = lane; a < na / 2; a += 64) {  // trs.into_iter().rev() (rm_epsilon_static.rs:125)
    const wfst_tr t = out[a];
    out[a] = out[na - 1 - a];
    out[na - 1 - a] = t;
  }
  if (lane == 0) {
    new_cnt[i] = na;
    new_fin[i] = final_w;
    new_ptr[i] = (unsigned long long)out;
    status[i] = 0u;
  }
}

// finished states of an attempt: their arcs move from the scratch slices into a compact arena
__global__ void rm_compact(const uint32_t* __restrict__ status, const uint32_t* __restrict__ new_cnt,
                           unsigned long long* __restrict__ new_ptr, const uint32_t* __restrict__ arena_off,
                           wfst_tr* __restrict__ arena, uint32_t n_list) {
  const uint32_t i = blockIdx.x;
  if (i >= n_list || status[i]) return;
  const wfst_tr* src = (const wfst_tr*)new_ptr[i];
  wfst_tr* dst = arena + arena_off[i];
  for (uint32_t k = threadIdx.x; k < new_cnt[i]; k += blockDim.x) dst[k] = src[k];
  __syncthreads();
  if (threadIdx.x == 0) new_ptr[i] = (unsigned long long)dst;
}

// makes the rewritten states of a finished launch visible to the next ones
__global__ void rm_publish(const uint32_t* __restrict__ list, uint32_t n_list, const uint32_t* __restrict__ status,
                           const uint32_t* __restrict__ new_cnt, const float* __restrict__ new_fin,
                           const unsigned long long* __restrict__ new_ptr, uint32_t* __restrict__ done, uint32_t* __restrict__ cnt,
                           float* __restrict__ fin, unsigned long long* __restrict__ arc_ptr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_list || status[i]) return;
  const uint32_t s = list[i];
  cnt[s] = new_cnt[i];
  fin[s] = new_fin[i];
  arc_ptr[s] = new_ptr[i];
  done[s] = 1u;
}

// the new arcs into CSR order; facts of the added arcs for the property word:
// 1 some arc has ilabel != olabel | 2 some arc has nextstate <= its state | 4 some arc was added
__global__ void rm_write(const uint32_t* __restrict__ off, const uint32_t* __restrict__ cnt,
                         const unsigned long long* __restrict__ arc_ptr, wfst_tr* __restrict__ out, uint32_t n,
                         uint32_t* __restrict__ facts) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t f = 0;
  if (s < n && cnt[s]) {
    const wfst_tr* src = (const wfst_tr*)arc_ptr[s];
    const uint32_t na = cnt[s], o = off[s];
    for (uint32_t k = 0; k < na; ++k) {
      const wfst_tr tr = src[k];
      out[o + k] = tr;
      f |= 4u | (tr.ilabel != tr.olabel ? 1u : 0u) | (tr.nextstate <= s ? 2u : 0u);
    }
  }
  for (int d = 32; d >= 1; d >>= 1) f |= __shfl_xor(f, d);
  if ((threadIdx.x & 63) == 0 && f) atomicOr(facts, f);
}

// schedule of the rewrites: batches of mutually independent states, in an order in which every batch only depends on
// earlier ones (see the header)
void rm_schedule(const wfst_fst* f, const std::vector<uint8_t>& noneps_in, std::vector<std::vector<uint32_t>>& batches,
                 std::vector<uint32_t>& closure_hint) {
  const HostCsr& h = f->host;
  const uint32_t n = f->n_states;
  // epsilon graph
  std::vector<uint32_t> eoff(n + 1, 0), edst;
  for (uint32_t s = 0; s < n; ++s) {
    for (uint32_t a = h.offsets[s]; a < h.offsets[s + 1]; ++a)
      if (h.arcs[a].ilabel == 0 && h.arcs[a].olabel == 0) edst.push_back(h.arcs[a].nextstate);
    eoff[s + 1] = (uint32_t)edst.size();
  }
  // strongly connected components (Tarjan, explicit stack); components come out in reverse topological order (sinks first)
  std::vector<int32_t> comp(n, -1), low(n, 0), num(n, -1);
  std::vector<uint32_t> stk, pos(n, 0), call;
  std::vector<uint8_t> on(n, 0);
  int32_t counter = 0, n_comp = 0;
  for (uint32_t root = 0; root < n; ++root) {
    if (num[root] >= 0) continue;
    call.push_back(root);
    while (!call.empty()) {
      const uint32_t s = call.back();
      if (num[s] < 0) {
        num[s] = low[s] = counter++;
        stk.push_back(s);
        on[s] = 1;
        pos[s] = eoff[s];
      }
      bool descended = false;
      while (pos[s] < eoff[s + 1]) {
        const uint32_t t = edst[pos[s]++];
        if (num[t] < 0) {
          call.push_back(t);
          descended = true;
          break;
        }
        if (on[t]) low[s] = std::min(low[s], num[t]);
      }
      if (descended) continue;
      if (low[s] == num[s]) {
        uint32_t t;
        do {
          t = stk.back();
          stk.pop_back();
          on[t] = 0;
          comp[t] = n_comp;
        } while (t != s);
        ++n_comp;
      }
      call.pop_back();
      if (!call.empty()) low[call.back()] = std::min(low[call.back()], low[s]);
    }
  }
  // size / self loop of the components, epsilon depth (components are numbered sinks first: children have smaller ids)
  std::vector<uint32_t> size(n_comp, 0), depth(n_comp, 0);
  std::vector<uint8_t> cyclic(n_comp, 0);
  for (uint32_t s = 0; s < n; ++s) size[comp[s]]++;
  for (uint32_t s = 0; s < n; ++s)
    for (uint32_t a = eoff[s]; a < eoff[s + 1]; ++a)
      if (edst[a] == s) cyclic[comp[s]] = 1;
  for (int32_t c = 0; c < n_comp; ++c)
    if (size[c] > 1) cyclic[c] = 1;
  {
    // depth in component order: every epsilon arc goes to a component with a smaller id (or the same one)
    std::vector<uint32_t> order(n);
    for (uint32_t s = 0; s < n; ++s) order[s] = s;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return comp[a] < comp[b]; });
    for (uint32_t s : order)
      for (uint32_t a = eoff[s]; a < eoff[s + 1]; ++a) {
        const int32_t ct = comp[edst[a]];
        if (ct != comp[s]) depth[comp[s]] = std::max(depth[comp[s]], depth[ct] + 1);
      }
  }
  uint32_t max_depth = 0;
  for (int32_t c = 0; c < n_comp; ++c) max_depth = std::max(max_depth, depth[c]);
  std::vector<std::vector<uint32_t>> plain(max_depth + 1);
  std::vector<std::vector<int32_t>> cyc_at(max_depth + 1);
  std::vector<std::vector<uint32_t>> cyc_members(n_comp);
  std::vector<uint8_t> listed(n_comp, 0);
  for (uint32_t s = 0; s < n; ++s) {  // (increasing id: the order inside an epsilon cycle)
    const int32_t c = comp[s];
    if (cyclic[c]) {
      if (!listed[c]) {
        listed[c] = 1;
        cyc_at[depth[c]].push_back(c);
      }
      if (noneps_in[s]) cyc_members[c].push_back(s);
    } else if (noneps_in[s]) {
      plain[depth[c]].push_back(s);
    }
  }
  for (uint32_t d = 0; d <= max_depth; ++d) {
    if (!plain[d].empty()) {
      batches.push_back(std::move(plain[d]));
      closure_hint.push_back(0);
    }
    for (int32_t c : cyc_at[d])
      for (uint32_t s : cyc_members[c]) {
        batches.push_back(std::vector<uint32_t>{s});
        closure_hint.push_back(size[c]);  // the closure holds at least the state's own epsilon component
      }
  }
}

}  // namespace

wfst_fst* rm_epsilon_fst(wfst_ctx* ctx, const wfst_fst* f) {
  using namespace props;
  ensure_device(const_cast<wfst_fst*>(f));
  const uint32_t n = f->n_states;
  hipStream_t st = ctx->stream;
  if (f->start < 0)  // `None => return Ok(())`: the FST is returned as it is (rm_epsilon_static.rs:58-61)
    return adopt_device(ctx, n, f->n_arcs, -1, f->props, f->dev.offsets, f->dev.arcs, f->dev.finals);
  ensure_host(f);
  // noneps_in[s]: s is the start state or has an incoming arc that is not epsilon:epsilon (rm_epsilon_static.rs:64-75)
  std::vector<uint8_t> noneps_in(n, 0);
  noneps_in[f->start] = 1;
  for (const wfst_tr& tr : f->host.arcs)
    if (tr.ilabel != 0 || tr.olabel != 0) noneps_in[tr.nextstate] = 1;
  std::vector<std::vector<uint32_t>> batches;
  std::vector<uint32_t> closure_hint;  // per batch: a lower bound of its closures (0 = unknown, start small)
  rm_schedule(f, noneps_in, batches, closure_hint);

  DBuf<uint32_t> done(*ctx->pool, n), cnt(*ctx->pool, (size_t)n + 1), off(*ctx->pool, (size_t)n + 1), facts(*ctx->pool, 1);
  DBuf<float> fin(*ctx->pool, n);
  DBuf<unsigned long long> arc_ptr(*ctx->pool, n);
  HIP_CHECK(hipMemsetAsync(done.p, 0, (size_t)n * 4, st));
  HIP_CHECK(hipMemsetAsync(cnt.p, 0, ((size_t)n + 1) * 4, st));
  HIP_CHECK(hipMemsetAsync(facts.p, 0, 4, st));
  HIP_CHECK(hipMemcpyAsync(fin.p, f->dev.finals, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  const RmView view{f->dev.offsets, f->dev.arcs, done.p, cnt.p, arc_ptr.p};
  // the new arcs of every finished attempt live in a compact arena until rm_write has copied them into CSR order; the
  // scratch slices of an attempt (832 B per state at the first size, four times more per retry) are released at once
  std::vector<std::unique_ptr<DBuf<wfst_tr>>> arenas;
  uint64_t total_new_arcs = 0;
  for (size_t bi = 0; bi < batches.size(); ++bi) {
    const std::vector<uint32_t>& batch = batches[bi];
    std::vector<uint32_t> todo = batch;
    RmCaps caps{16, 32, 32};
    RmBigCaps big{1024, 1024, 1024};
    if (closure_hint[bi] > 64) {  // a member of a large epsilon cycle: straight to the wave kernel, sized for the component
      caps.C = 65;
      const uint32_t c = rm_pow2_at_least(2 * closure_hint[bi]);
      big = RmBigCaps{std::max(1024u, c), std::max(1024u, 2 * c), std::max(1024u, 4 * c)};
    }
    for (int attempt = 0; !todo.empty(); ++attempt) {
      // closures of more than 64 states: one wave per state, hashed lookups (rm_expand_wave) — the one-thread kernel
      // searches its closure linearly and would spend closure^2 dependent loads before giving up
      const bool wave = caps.C > 64;
      const size_t m = todo.size();
      const size_t slice = wave ? rm_big_slice_bytes(big) : rm_slice_bytes(caps);
      if (slice * m > (64ull << 30)) throw Error("rm_epsilon: scratch for the epsilon closures exceeds 64 GiB");
      DBuf<char> scratch(*ctx->pool, m * slice);
      DBuf<uint32_t> list(*ctx->pool, m), status(*ctx->pool, m), new_cnt(*ctx->pool, m), arena_off(*ctx->pool, m);
      DBuf<float> new_fin(*ctx->pool, m);
      DBuf<unsigned long long> new_ptr(*ctx->pool, m);
      HIP_CHECK(hipMemcpyAsync(list.p, todo.data(), m * 4, hipMemcpyHostToDevice, st));
      const uint32_t blocks = (uint32_t)((m + 63) / 64);
      if (wave)
        rm_expand_wave<<<(uint32_t)m, 64, 0, st>>>(view, list.p, (uint32_t)m, big, scratch.p, new_cnt.p, new_fin.p, new_ptr.p, fin.p,
                                                   status.p);
      else
        rm_expand<<<blocks, 64, 0, st>>>(view, list.p, (uint32_t)m, caps, scratch.p, new_cnt.p, new_fin.p, new_ptr.p, fin.p,
                                         status.p);
      HIP_CHECK(hipGetLastError());
      std::vector<uint32_t> h_status(m), h_cnt(m);
      HIP_CHECK(hipMemcpyAsync(h_status.data(), status.p, m * 4, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipMemcpyAsync(h_cnt.data(), new_cnt.p, m * 4, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      std::vector<uint32_t> again, h_off(m, 0);
      uint64_t arena_arcs = 0;
      for (size_t i = 0; i < m; ++i) {
        if (h_status[i]) {
          again.push_back(todo[i]);
        } else {
          h_off[i] = (uint32_t)arena_arcs;
          arena_arcs += h_cnt[i];
        }
      }
      total_new_arcs += arena_arcs;
      if (arena_arcs > 0xFFFFFFFFull || total_new_arcs > 0xFFFFFFFFull) throw Error("rm_epsilon: the result has more than 2^32 arcs");
      if (arena_arcs) {
        arenas.emplace_back(new DBuf<wfst_tr>(*ctx->pool, arena_arcs));
        HIP_CHECK(hipMemcpyAsync(arena_off.p, h_off.data(), m * 4, hipMemcpyHostToDevice, st));
        rm_compact<<<(uint32_t)m, 64, 0, st>>>(status.p, new_cnt.p, new_ptr.p, arena_off.p, arenas.back()->p, (uint32_t)m);
      }
      rm_publish<<<blocks, 64, 0, st>>>(list.p, (uint32_t)m, status.p, new_cnt.p, new_fin.p, new_ptr.p, done.p, cnt.p, fin.p,
                                        arc_ptr.p);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(st));  // the scratch and the per-attempt lists are released here
      todo.swap(again);
      if (wave) big = RmBigCaps{big.C * 4, big.K * 4, big.A * 4};
      else caps = RmCaps{caps.C * 4, caps.K * 4, caps.A * 4};
      if (attempt > 24) throw Error("rm_epsilon: scratch overflow after retries");
    }
  }
  // states that were not rewritten lose their arcs (rm_epsilon_static.rs:137-143): cnt is 0 for them already.
  // CSR of the result before connect
  size_t temp_bytes = 0;
  HIP_CHECK(rocprim::exclusive_scan(nullptr, temp_bytes, cnt.p, off.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st));
  DBuf<uint8_t> temp(*ctx->pool, temp_bytes);
  HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, cnt.p, off.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st));
  uint32_t h[2];
  HIP_CHECK(hipMemcpyAsync(&h[0], off.p + n, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  DBuf<wfst_tr> new_arcs(*ctx->pool, h[0]);
  rm_write<<<(n + 255) / 256, 256, 0, st>>>(off.p, cnt.p, arc_ptr.p, new_arcs.p, n, facts.p);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(&h[1], facts.p, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  // property word: every rewrite applies delete_trs_properties, add_tr over its new arcs and set_final; what
  // rmepsilon_properties(.., delayed = false) then reads are ACCEPTOR / ACYCLIC / INITIAL_ACYCLIC / TOP_SORTED, all of
  // them functions of the three facts; connect finishes with delete_states_properties | ACCESSIBLE | COACCESSIBLE
  const uint64_t in = f->props;
  const bool acceptor = (in & ACCEPTOR) && !(h[1] & 1u);
  const bool top = (in & TOP_SORTED) && !(h[1] & 2u);
  uint64_t out = NO_EPSILONS;
  if (acceptor) out |= ACCEPTOR | NO_I_EPSILONS | NO_O_EPSILONS;
  if (h[1] & 4u) {
    if (top) out |= ACYCLIC | INITIAL_ACYCLIC;  // add_tr keeps them only next to TOP_SORTED (mutate_properties.rs:93-99)
  } else {
    out |= (ACYCLIC | INITIAL_ACYCLIC) & in;
  }
  if (top) out |= TOP_SORTED;
  out = delete_states(out) | ACCESSIBLE | COACCESSIBLE;
  return connect_and_adopt(ctx, n, f->start, off.p, new_arcs.p, fin.p, /*all_accessible=*/false, out);
}

}  // namespace wfst
