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
// The reference relaxes only improvements larger than delta = 1e-6 (approx_equal, shortest_distance.rs:216); d[] here is
// the exact minimum of the left-folded f32 path sums, the same whenever weights differ by more than 1e-6 (any 1/512-grid
// input) — the deviation already documented for shortest_distance.
#include <algorithm>

#include <rocprim/device/device_scan.hpp>

#include "common.h"
#include "fst_props.h"

namespace wfst {

namespace {

constexpr uint32_t RM_MAX_CLOSURE = 1024;
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
void rm_schedule(const wfst_fst* f, const std::vector<uint8_t>& noneps_in, std::vector<std::vector<uint32_t>>& batches) {
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
    if (!plain[d].empty()) batches.push_back(std::move(plain[d]));
    for (int32_t c : cyc_at[d])
      for (uint32_t s : cyc_members[c]) batches.push_back(std::vector<uint32_t>{s});
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
  rm_schedule(f, noneps_in, batches);

  DBuf<uint32_t> done(*ctx->pool, n), cnt(*ctx->pool, (size_t)n + 1), off(*ctx->pool, (size_t)n + 1), facts(*ctx->pool, 1);
  DBuf<float> fin(*ctx->pool, n);
  DBuf<unsigned long long> arc_ptr(*ctx->pool, n);
  HIP_CHECK(hipMemsetAsync(done.p, 0, (size_t)n * 4, st));
  HIP_CHECK(hipMemsetAsync(cnt.p, 0, ((size_t)n + 1) * 4, st));
  HIP_CHECK(hipMemsetAsync(facts.p, 0, 4, st));
  HIP_CHECK(hipMemcpyAsync(fin.p, f->dev.finals, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  const RmView view{f->dev.offsets, f->dev.arcs, done.p, cnt.p, arc_ptr.p};
  std::vector<std::unique_ptr<DBuf<char>>> scratches;  // the new arcs live here until rm_write has copied them
  for (const std::vector<uint32_t>& batch : batches) {
    std::vector<uint32_t> todo = batch;
    RmCaps caps{16, 32, 32};
    for (int attempt = 0; !todo.empty(); ++attempt) {
      // one thread walks a closure with linear searches: fine for the handful of states closures have in practice,
      // quadratic beyond; a closure of more than RM_MAX_CLOSURE states is refused rather than ground through
      if (caps.C > RM_MAX_CLOSURE) throw Error("unsupported: rm_epsilon with an epsilon closure of more than 1024 states");
      const size_t m = todo.size();
      scratches.emplace_back(new DBuf<char>(*ctx->pool, m * rm_slice_bytes(caps)));
      DBuf<uint32_t> list(*ctx->pool, m), status(*ctx->pool, m), new_cnt(*ctx->pool, m);
      DBuf<float> new_fin(*ctx->pool, m);
      DBuf<unsigned long long> new_ptr(*ctx->pool, m);
      HIP_CHECK(hipMemcpyAsync(list.p, todo.data(), m * 4, hipMemcpyHostToDevice, st));
      const uint32_t blocks = (uint32_t)((m + 63) / 64);
      rm_expand<<<blocks, 64, 0, st>>>(view, list.p, (uint32_t)m, caps, scratches.back()->p, new_cnt.p, new_fin.p, new_ptr.p, fin.p,
                                       status.p);
      rm_publish<<<blocks, 64, 0, st>>>(list.p, (uint32_t)m, status.p, new_cnt.p, new_fin.p, new_ptr.p, done.p, cnt.p, fin.p,
                                        arc_ptr.p);
      HIP_CHECK(hipGetLastError());
      std::vector<uint32_t> h_status(m);
      HIP_CHECK(hipMemcpyAsync(h_status.data(), status.p, m * 4, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      std::vector<uint32_t> again;
      for (size_t i = 0; i < m; ++i)
        if (h_status[i]) again.push_back(todo[i]);
      todo.swap(again);
      caps = RmCaps{caps.C * 4, caps.K * 4, caps.A * 4};
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
