// compose_wide.hip — composition of ONE large pair of FSTs with the ComposeFilterEnum filters on the wide driver of
// compose_wide.h (8 / 16 / 64 lanes per composed state of a BFS level, level control on the device), plus connect() on
// the device.
//
// compose.hip runs one wavefront per (fst1, fst2) problem: right for batches of small lattices, hopeless for a single
// composition of a million states (5 s on MI355X, slower than the CPU).  compose() hands such a result over to this file
// once it has outgrown the wave kernel's first arena.  Same reference semantics as compose.hip (file:line there):
// SortedMatcher + EpsLoop (matchers/sorted_matcher.rs:124-184, matchers/mod.rs:98-105), the six filters as per-state
// outcome classes, ComposeFstOp::{compute_trs, compute_final_weight} (compose_fst_op.rs:406-449), first-touch state ids,
// connect + del_states (connect.rs:51-66, vector_fst/mutable_fst.rs:132-189).
#include "compose_filters.h"
#include "compose_wide.h"
#include "fst_props.h"

namespace wfst {
namespace {

constexpr uint32_t NO_LABEL = WFST_NO_LABEL;
constexpr uint32_t REJECT = 0xFFFFFFFFu;
enum : uint32_t { MODE_BOTH = 0, MODE_INPUT = 1, MODE_OUTPUT = 2 };

struct PlainView {
  const wfst_tr* arcs;
  const uint4* srec;  // {arc begin, arc count, final bits, SREC_* epsilon facts}
};

struct PlainExpand {
  bool mi;
  const wfst_tr* it_arcs;
  const wfst_tr* se_arcs;
  uint32_t n_it, n_se, sa, sb;
  uint32_t fs_nolabel, fs_eps, fsZ, fsM;  // filter_tr outcomes of the four pair classes (see compose.hip expand_state)
  float final_weight;
};

__device__ void plain_equal_range(const wfst_tr* arcs, uint32_t n, bool by_ilabel, uint32_t key, uint32_t* lo_out, uint32_t* cnt_out) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    const uint32_t k = by_ilabel ? arcs[mid].ilabel : arcs[mid].olabel;
    if (k < key) lo = mid + 1; else hi = mid;
  }
  const uint32_t first = lo;
  hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    const uint32_t k = by_ilabel ? arcs[mid].ilabel : arcs[mid].olabel;
    if (k <= key) lo = mid + 1; else hi = mid;
  }
  *lo_out = first;
  *cnt_out = lo - first;
}

struct PlainPolicy {
  PlainView f1, f2;
  uint32_t mode, filter;
  using Expand = PlainExpand;

  __device__ Expand make_expand(uint64_t tlo, uint64_t thi) const {
    Expand x;
    const uint32_t s1 = (uint32_t)(tlo >> 32), s2 = (uint32_t)tlo, fs = (uint32_t)thi;
    const uint4 r1 = f1.srec[s1], r2 = f2.srec[s2];
    const uint32_t n1 = r1.y, n2 = r2.y;
    const float fin1 = __uint_as_float(r1.z), fin2 = __uint_as_float(r2.z);
    x.final_weight = (fin1 != INF && fin2 != INF) ? wtimes(fin1, fin2) : INF;  // compute_final_weight :420-449
    const bool alleps1 = (r1.w & SREC_ALL_OEPS) && !(fin1 != INF);
    const bool noeps1 = (r1.w & SREC_NO_OEPS) != 0;
    const bool alleps2 = (r2.w & SREC_ALL_IEPS) && !(fin2 != INF);
    const bool noeps2 = (r2.w & SREC_NO_IEPS) != 0;
    x.mi = mode == MODE_BOTH ? (n1 <= n2) : (mode == MODE_INPUT);  // match_input :199-219
    x.it_arcs = x.mi ? f1.arcs + r1.x : f2.arcs + r2.x;
    x.se_arcs = x.mi ? f2.arcs + r2.x : f1.arcs + r1.x;
    x.n_it = x.mi ? n1 : n2;
    x.n_se = x.mi ? n2 : n1;
    x.sa = x.mi ? s2 : s1;
    x.sb = x.mi ? s1 : s2;
    const FilterOutcomes fo = filter_outcomes(filter, fs, alleps1, noeps1, alleps2, noeps2);  // compose_filters.h
    const uint32_t fsX = fo.fsX, fsY = fo.fsY, fsZ = fo.fsZ;
    x.fs_nolabel = x.mi ? fsX : fsY;  // the loop pseudo-arc of the iterated side against the searched side's epsilon arcs
    x.fs_eps = x.mi ? fsY : fsX;      // an epsilon of the iterated side against the matcher's EpsLoop
    x.fsZ = fsZ;
    x.fsM = 0u;
    return x;
  }

  // item 0 = the loop pseudo-arc (ordered_expand :229-233), item j = the j-th arc of the iterated side; the sorted matcher
  // yields for label 0 the EpsLoop first, then the real epsilon arcs; for NO_LABEL the epsilon arcs without the loop
  __device__ uint32_t eval_item(const Expand& x, uint32_t j, bool write, uint32_t write_pos, wfst_tr* arcs, uint64_t* a_lo,
                                uint64_t* a_hi, Emitted* first) const {
    const bool mi = x.mi;
    const ArcReg ab = j == 0 ? (mi ? ArcReg{0u, NO_LABEL, 0.0f, x.sb} : ArcReg{NO_LABEL, 0u, 0.0f, x.sb}) : load_arc(x.it_arcs + (j - 1));
    const uint32_t label = mi ? ab.ol : ab.il;
    uint32_t lo = 0, cnt = 0;
    plain_equal_range(x.se_arcs, x.n_se, mi, label == NO_LABEL ? 0u : label, &lo, &cnt);
    uint32_t loop1 = 0, fsn;
    if (label == NO_LABEL) {
      fsn = x.fs_nolabel;
    } else if (label == 0u) {
      loop1 = x.fs_eps != REJECT ? 1u : 0u;
      fsn = x.fsZ;
    } else {
      fsn = x.fsM;
    }
    const uint32_t n_real = fsn != REJECT ? cnt : 0u;
    uint32_t k = 0;
    for (uint32_t m = 0; m < loop1 + n_real; ++m) {
      const bool is_loop = m < loop1;
      const ArcReg aa = is_loop ? (mi ? ArcReg{NO_LABEL, 0u, 0.0f, x.sa} : ArcReg{0u, NO_LABEL, 0.0f, x.sa}) : load_arc(x.se_arcs + lo + (m - loop1));
      const ArcReg a1 = mi ? ab : aa;  // arc1 from fst1, arc2 from fst2 (match_tr_selected :301-319)
      const ArcReg a2 = mi ? aa : ab;
      const uint4 arc = make_uint4(a1.il, a2.ol, __float_as_uint(wtimes(a1.w, a2.w)), 0u);  // add_tr :267-285
      const uint64_t dlo = ((uint64_t)a1.ns << 32) | a2.ns, dhi = is_loop ? x.fs_eps : fsn;
      if (write) {
        *reinterpret_cast<uint4*>(arcs + write_pos + k) = arc;
        a_lo[write_pos + k] = dlo;
        a_hi[write_pos + k] = dhi;
      } else if (first && k == 0) {
        first->arc = arc;
        first->lo = dlo;
        first->hi = dhi;
      }
      k++;
    }
    return k;
  }
};

// ---------------------------------------------------------------- connect (connect.rs:51-66) on the finished CSR
// every composed state is accessible (it was discovered from the start state); coaccessible = reaches a final state
__global__ void coaccess_init(const float* __restrict__ fin, uint32_t* __restrict__ co, uint32_t n) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) co[s] = fin[s] != INF ? 1u : 0u;
}
// One backward sweep.  A thread walks SWEEP_RUN consecutive states in DESCENDING order and marks made in this very sweep
// count at once, so a chain numbered in path order (linear acceptors, lattices, reversed paths: the common deep inputs)
// advances SWEEP_RUN states per sweep and thread instead of one; the host looks at the result every SWEEP_BATCH sweeps.
constexpr uint32_t SWEEP_RUN = 16, SWEEP_BATCH = 4;
__global__ void coaccess_sweep(const uint32_t* __restrict__ off, const wfst_tr* __restrict__ arcs, uint32_t* __restrict__ co,
                               uint32_t n, uint32_t* __restrict__ changed) {
  bool any = false;
  const uint32_t n_runs = (n + SWEEP_RUN - 1) / SWEEP_RUN;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += gridDim.x * blockDim.x) {
    const uint32_t lo = r * SWEEP_RUN, hi = min(n, lo + SWEEP_RUN);
    for (uint32_t s = hi; s-- > lo;) {
      if (ld_l2(&co[s])) continue;
      for (uint32_t k = off[s]; k < off[s + 1]; ++k) {
        if (ld_l2(&co[arcs[k].nextstate])) {
          st_l2(&co[s], 1u);
          any = true;
          break;
        }
      }
    }
  }
  if (__any(any) && (threadIdx.x & 63) == 0) *changed = 1u;
}
// arcs a kept state keeps = those into kept states (del_states drops the others in order, mutable_fst.rs:160-176)
__global__ void kept_arc_counts(const uint32_t* __restrict__ off, const wfst_tr* __restrict__ arcs, const uint32_t* __restrict__ co,
                                const uint32_t* __restrict__ new_id, uint32_t* __restrict__ cnt, uint32_t n) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n || !co[s]) return;
  uint32_t c = 0;
  for (uint32_t k = off[s]; k < off[s + 1]; ++k) c += co[arcs[k].nextstate];
  cnt[new_id[s]] = c;
}
__global__ void compact_states(const uint32_t* __restrict__ off, const wfst_tr* __restrict__ arcs, const float* __restrict__ fin,
                               const uint32_t* __restrict__ co, const uint32_t* __restrict__ new_id,
                               const uint32_t* __restrict__ t_off, wfst_tr* __restrict__ t_arcs, float* __restrict__ t_fin, uint32_t n) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n || !co[s]) return;
  const uint32_t t = new_id[s];
  t_fin[t] = fin[s];
  uint32_t w = t_off[t];
  for (uint32_t k = off[s]; k < off[s + 1]; ++k) {
    const wfst_tr a = arcs[k];
    if (co[a.nextstate]) {
      t_arcs[w] = wfst_tr{a.ilabel, a.olabel, a.weight, new_id[a.nextstate]};
      ++w;
    }
  }
}

// accessibility from the start state (ConnectVisitor's `access`, connect.rs:106-115) by forward sweeps to a fixed point
__global__ void access_sweep(const uint32_t* __restrict__ off, const wfst_tr* __restrict__ arcs, uint32_t* __restrict__ acc,
                             uint32_t* __restrict__ expanded, uint32_t n, uint32_t* __restrict__ changed) {
  bool any = false;
  const uint32_t n_runs = (n + SWEEP_RUN - 1) / SWEEP_RUN;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += gridDim.x * blockDim.x) {
    const uint32_t lo = r * SWEEP_RUN, hi = min(n, lo + SWEEP_RUN);
    for (uint32_t s = lo; s < hi; ++s) {  // ascending: a forward chain advances SWEEP_RUN states per sweep
      if (!ld_l2(&acc[s]) || expanded[s]) continue;
      expanded[s] = 1u;
      for (uint32_t k = off[s]; k < off[s + 1]; ++k) {
        const uint32_t t = arcs[k].nextstate;
        if (!ld_l2(&acc[t])) {
          st_l2(&acc[t], 1u);
          any = true;
        }
      }
    }
  }
  if (__any(any) && (threadIdx.x & 63) == 0) *changed = 1u;
}
__global__ void and_flags(uint32_t* __restrict__ co, const uint32_t* __restrict__ acc, uint32_t n) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) co[s] = co[s] & acc[s];
}

}  // namespace

// connect (connect.rs:51-66) of a CSR in HBM: keep = accessible AND coaccessible (every state is accessible when
// `all_accessible`: a composition just built from its start state), stable renumbering, arcs into deleted states dropped
wfst_fst* connect_and_adopt(wfst_ctx* ctx, uint32_t n, int64_t start, const uint32_t* off, const wfst_tr* arcs, const float* fin,
                            bool all_accessible, uint64_t out_props) {
  hipStream_t st = ctx->stream;
  auto empty = [&] {
    HostCsr hc;
    hc.offsets.push_back(0);
    return make_host_fst(ctx, 0, -1, out_props, std::move(hc));
  };
  if (n == 0 || start < 0) return empty();  // no start state: nothing is accessible, del_states removes everything
  DBuf<uint32_t> co(*ctx->pool, (size_t)n + 1), new_id(*ctx->pool, (size_t)n + 1), cnt(*ctx->pool, (size_t)n + 1),
      t_off(*ctx->pool, (size_t)n + 1), changed(*ctx->pool, 1);
  const uint32_t blocks = (n + 255) / 256;
  const uint32_t sweep_blocks = std::max(1u, std::min<uint32_t>((n / SWEEP_RUN + 255) / 256, (uint32_t)ctx->n_cus * 16));
  uint32_t* h = (uint32_t*)ctx->pinned.get(4 * sizeof(uint32_t));
  coaccess_init<<<blocks, 256, 0, st>>>(fin, co.p, n);
  for (;;) {  // backward reachability to a fixed point: SWEEP_BATCH sweeps per look at the flag of the LAST one
    for (uint32_t k = 0; k < SWEEP_BATCH; ++k) {
      HIP_CHECK(hipMemsetAsync(changed.p, 0, sizeof(uint32_t), st));
      coaccess_sweep<<<sweep_blocks, 256, 0, st>>>(off, arcs, co.p, n, changed.p);
    }
    HIP_CHECK(hipMemcpyAsync(h, changed.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (!h[0]) break;
  }
  if (!all_accessible) {
    DBuf<uint32_t> acc(*ctx->pool, n), expanded(*ctx->pool, n);
    HIP_CHECK(hipMemsetAsync(acc.p, 0, (size_t)n * sizeof(uint32_t), st));
    HIP_CHECK(hipMemsetAsync(expanded.p, 0, (size_t)n * sizeof(uint32_t), st));
    const uint32_t one = 1;
    HIP_CHECK(hipMemcpyAsync(acc.p + start, &one, sizeof(uint32_t), hipMemcpyHostToDevice, st));
    for (;;) {
      for (uint32_t k = 0; k < SWEEP_BATCH; ++k) {
        HIP_CHECK(hipMemsetAsync(changed.p, 0, sizeof(uint32_t), st));
        access_sweep<<<sweep_blocks, 256, 0, st>>>(off, arcs, acc.p, expanded.p, n, changed.p);
      }
      HIP_CHECK(hipMemcpyAsync(h, changed.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      if (!h[0]) break;
    }
    and_flags<<<blocks, 256, 0, st>>>(co.p, acc.p, n);
    HIP_CHECK(hipStreamSynchronize(st));  // acc / expanded are released here
  }
  // stable renumbering of the survivors (del_states, mutable_fst.rs:132-158)
  HIP_CHECK(hipMemsetAsync(co.p + n, 0, sizeof(uint32_t), st));
  size_t temp_bytes = 0;
  HIP_CHECK(rocprim::exclusive_scan(nullptr, temp_bytes, co.p, new_id.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st));
  DBuf<uint8_t> temp(*ctx->pool, temp_bytes);
  HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, co.p, new_id.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st));
  HIP_CHECK(hipMemsetAsync(cnt.p, 0, ((size_t)n + 1) * sizeof(uint32_t), st));
  kept_arc_counts<<<blocks, 256, 0, st>>>(off, arcs, co.p, new_id.p, cnt.p, n);
  HIP_CHECK(hipMemcpyAsync(h + 1, new_id.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(h + 3, new_id.p + start, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(h, co.p + start, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  const uint32_t t_states = h[1];
  // the start state survives iff it reaches a final state; if it does not, nothing accessible does: everything goes
  if (t_states == 0 || !h[0]) return empty();
  const int64_t t_start = h[3];
  HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, cnt.p, t_off.p, 0u, (size_t)t_states + 1, rocprim::plus<uint32_t>(), st));
  HIP_CHECK(hipMemcpyAsync(h + 2, t_off.p + t_states, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  const uint32_t t_arcs_n = h[2];
  DBuf<wfst_tr> t_arcs(*ctx->pool, t_arcs_n);
  DBuf<float> t_fin(*ctx->pool, t_states);
  compact_states<<<blocks, 256, 0, st>>>(off, arcs, fin, co.p, new_id.p, t_off.p, t_arcs.p, t_fin.p, n);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(st));
  return adopt_device(ctx, t_states, t_arcs_n, t_start, out_props, t_off.p, t_arcs.p, t_fin.p);
}

wfst_fst* compose_wide(wfst_ctx* ctx, const wfst_fst* f1, const wfst_fst* f2, uint32_t mode, uint32_t filter, bool connect,
                       uint64_t out_props, uint64_t est_s) {
  const PlainPolicy pol{PlainView{f1->dev.arcs, f1->dev.srec}, PlainView{f2->dev.arcs, f2->dev.srec}, mode, filter};
  WideOutput w;
  const double d1 = (double)f1->n_arcs / std::max<uint32_t>(f1->n_states, 1), d2 = (double)f2->n_arcs / std::max<uint32_t>(f2->n_states, 1);
  const double items = 1.0 + (mode == MODE_INPUT ? d1 : mode == MODE_OUTPUT ? d2 : std::min(d1, d2));  // iterated side
  run_wide(ctx, pol, ((uint64_t)(uint32_t)f1->start << 32) | (uint32_t)f2->start, 0ull, est_s, 4 * est_s, items, w);
  ctx->stats.compose_states = w.n_states;
  ctx->stats.compose_arcs = w.n_arcs;
  if (!connect) return adopt_device(ctx, w.n_states, w.n_arcs, 0, out_props, w.off, w.arcs, w.fin);
  return connect_and_adopt(ctx, w.n_states, 0, w.off, w.arcs, w.fin, /*all_accessible=*/true, out_props);
}

// connect (algorithms/connect.rs:51-66) of a resident FST: a new handle with the accessible and coaccessible states
wfst_fst* connect_fst(wfst_ctx* ctx, const wfst_fst* f) {
  using namespace props;
  ensure_device(const_cast<wfst_fst*>(f));
  // del_states -> delete_states_properties, then ACCESSIBLE | COACCESSIBLE (connect.rs:61-64)
  const uint64_t out_props = (delete_states(f->props) & ~(ACCESSIBLE | NOT_ACCESSIBLE | COACCESSIBLE | NOT_COACCESSIBLE)) | ACCESSIBLE | COACCESSIBLE;
  return connect_and_adopt(ctx, f->n_states, f->start, f->dev.offsets, f->dev.arcs, f->dev.finals, /*all_accessible=*/false, out_props);
}

}  // namespace wfst
