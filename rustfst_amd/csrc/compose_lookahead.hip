// compose_lookahead.hip — look-ahead composition of HBM-resident CSR FSTs.
//
// Replaces the configuration the reference wires in rustfst-cli/src/cmds/compose.rs:77-181 (and tests in
// rustfst/src/tests_openfst/algorithms/compose.rs:118-254):
//   MatcherFst::new_with_relabeling                      compose/matcher_fst.rs:73-94       (host: lookahead.cpp)
//   LabelLookAheadMatcher::lookahead_fst / _label        lookahead_matchers/label_lookahead_matcher.rs:154-224
//   LabelReachable::{reach, reach_label, reach_final}    compose/label_reachable.rs:293-402
//   AltSequenceComposeFilter                             compose_filters/alt_sequence_compose_filter.rs:143-181
//   LookAheadComposeFilter (SMatchOutput)                lookahead_filters/lookahead_compose_filter.rs:191-290
//   PushWeightsComposeFilter                             lookahead_filters/push_weights_compose_filter.rs:131-189
//   PushLabelsComposeFilter + MultiEpsMatcher            lookahead_filters/push_labels_compose_filter.rs:160-400,
//                                                        matchers/multi_eps_matcher.rs:64-210
//   ComposeFstOp / StateTable / LazyFst::compute         compose_fst_op.rs:199-449, lazy/state_table.rs:49-59,
//                                                        lazy/lazy_fst.rs:226-269 (no connect: cmds/compose.rs:173-180)
//
// Same execution model as compose.hip: ONE wavefront walks the reference's FIFO BFS level by level, so state ids are the
// reference's first-touch ids.  What differs: the filter decision is per (arc1, arc2) PAIR and needs memory (the interval
// set of arc1's destination against the sorted arcs of arc2's destination), it rewrites arc2 (pushed weight, pushed label,
// jump over a unique prefix arc), and the composed-state tuple is (s1, s2, fs, pushed weight, pushed label): 128 bits.
// Lanes share the iterated side's arcs of one composed state (one item per lane); each lane walks its own matches, once
// to count and once to write (emission order = item order, then match order).
//
// Two drivers over the same per-state code:
//  * compose_lookahead_kernel: ONE wave does the whole composition (small results: no launch per level).  Destinations
//    are interned after each level in emission order, 64 at a time: duplicates inside a chunk are folded with shuffles,
//    the distinct keys then probe a table of two 64-bit words per slot (the claim is one CAS on the first word, the
//    winner writes the second; a lane that finds the first word equal but the second still "unset" retries the slot on the
//    next iteration of the wave-uniform loop).  One launch takes a whole batch (one wave per problem:
//    wfst_compose_lookahead_batch).  It gives up (LA_SWITCH_WIDE) once a BFS level adds more than WIDE_SWITCH_WIDTH states
//    or the result passes WIDE_SWITCH_STATES (a call with ONE composition hands over earlier: *_ONE).
//  * the wide driver of compose_wide.h (a few lanes per composed state of a BFS level, three launches per level, level
//    control on the device), with the look-ahead filter stack as its policy (LaPolicy below).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "common.h"
#include "fst_props.h"
#include "lookahead.h"
#include "compose_wide.h"

namespace wfst {

namespace {

constexpr uint32_t NO_LABEL = WFST_NO_LABEL;
constexpr uint32_t REJECT = 0xFFFFFFFFu;
constexpr float KDELTA_F = 1.0f / 1024.0f;  // lib.rs:269
constexpr uint32_t WIDE_SWITCH_STATES = 2048;  // results larger than this are redone on the wide path
constexpr uint32_t WIDE_SWITCH_WIDTH = 256;    // ... and so are results with a BFS level wider than this
constexpr uint32_t WIDE_SWITCH_STATES_ONE = 512;  // the same two limits for a call with ONE composition
constexpr uint32_t WIDE_SWITCH_WIDTH_ONE = 48;


struct LaView {
  const wfst_tr* arcs;
  const uint4* srec;  // {arc begin, arc count, final bits, SREC_* epsilon facts}
  uint32_t n_states;
  int32_t start;
};
struct LaArena {
  uint64_t* t_lo;   // [S] tuple: s1 << 32 | s2
  uint64_t* t_hi;   // [S]        fs << 63 | (pushed label + 1, 0 = none) << 32 | bits of the pushed weight
  uint64_t* klo;    // [H]
  uint64_t* khi;    // [H]
  uint32_t* hvals;  // [H] state id
  wfst_tr* arcs;    // [A]
  uint64_t* a_lo;   // [A] destination tuple of each arc until it is interned
  uint64_t* a_hi;   // [A]
  uint32_t* off;    // [S+1]
  float* fin;       // [S]
};
struct LaResult {
  uint32_t status, n_states, n_arcs, n_levels;
};
__host__ __device__ inline size_t la_al16(size_t x) { return (x + 15) & ~(size_t)15; }
// one problem's arena of the single-wave kernel (a batch = `stride` bytes apart)
__host__ __device__ inline LaArena la_carve(char* b, const LaCaps& c, size_t* bytes_out) {
  size_t o = 0;
  LaArena a;
  a.t_lo = (uint64_t*)(b + o); o += la_al16((size_t)c.S * 8);
  a.t_hi = (uint64_t*)(b + o); o += la_al16((size_t)c.S * 8);
  a.klo = (uint64_t*)(b + o); o += la_al16((size_t)c.H * 8);
  a.khi = (uint64_t*)(b + o); o += la_al16((size_t)c.H * 8);
  a.hvals = (uint32_t*)(b + o); o += la_al16((size_t)c.H * 4);
  a.arcs = (wfst_tr*)(b + o); o += la_al16((size_t)c.A * 16);
  a.a_lo = (uint64_t*)(b + o); o += la_al16((size_t)c.A * 8);
  a.a_hi = (uint64_t*)(b + o); o += la_al16((size_t)c.A * 8);
  a.off = (uint32_t*)(b + o); o += la_al16((size_t)(c.S + 1) * 4);
  a.fin = (float*)(b + o); o += la_al16((size_t)c.S * 4);
  if (bytes_out) *bytes_out = (o + 255) & ~(size_t)255;
  return a;
}

// filter state of PushLabels(PushWeights(LookAhead(AltSequence))): PairFilterState<PairFilterState<IntegerFilterState,
// WeightFilterState>, IntegerFilterState>
struct FState {
  uint32_t fs;      // AltSequence state (0 / 1)
  float fweight;    // pushed weight, quantized to KDELTA
  uint32_t flabel;  // pushed label, NO_LABEL = none
};
__host__ __device__ __forceinline__ uint64_t pack_hi(const FState& f) {
  uint32_t wbits;
  __builtin_memcpy(&wbits, &f.fweight, 4);
  return ((uint64_t)f.fs << 63) | ((uint64_t)(f.flabel == NO_LABEL ? 0u : f.flabel + 1u) << 32) | wbits;
}
__device__ __forceinline__ FState unpack_hi(uint64_t hi) {
  const uint32_t l = (uint32_t)(hi >> 32) & 0x7FFFFFFFu;
  return FState{(uint32_t)(hi >> 63), __uint_as_float((uint32_t)hi), l ? l - 1u : NO_LABEL};
}

struct Reach {  // device view of LabelReachData
  const uint32_t* iv_off;
  const uint32_t* iv;  // pairs begin, end
  uint32_t final_label;
};

// IntervalSet::member (interval_set.rs:138-145) on the normalized set of state s: the last interval that begins at or
// before `value` must end after it
__device__ bool iv_member(const Reach& r, uint32_t s, uint32_t value) {
  uint32_t lo = r.iv_off[s], hi = r.iv_off[s + 1];
  const uint32_t first = lo;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (r.iv[2 * mid] <= value) lo = mid + 1; else hi = mid;
  }
  return lo > first && r.iv[2 * (lo - 1) + 1] > value;
}

// first arc position in [lo, hi) of fst2's state whose ilabel is >= label (LabelReachable::lower_bound :375-402)
__device__ uint32_t arcs_lower_bound(const wfst_tr* arcs, uint32_t lo, uint32_t hi, uint32_t label) {
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (arcs[mid].ilabel < label) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// LabelLookAheadMatcher::lookahead_fst (label_lookahead_matcher.rs:154-213) with LOOKAHEAD_WEIGHT | LOOKAHEAD_PREFIX:
// can anything in fst2 at state s2 be read from fst1 at state s1 after output-epsilon moves?  *lweight = the (+)-sum
// (min) of what can, unless exactly one arc can and no final state is reachable: that arc is the prefix.
__device__ bool lookahead_fst(const Reach& r, const LaView& f2, uint32_t s1, uint32_t s2, float* lweight, bool* has_prefix,
                              ArcReg* prefix) {
  const uint4 rec = f2.srec[s2];
  const wfst_tr* arcs = f2.arcs + rec.x;
  const uint32_t n = rec.y;
  const float lfinal = __uint_as_float(rec.z);
  const uint32_t ib = r.iv_off[s1], ie = r.iv_off[s1 + 1];
  // LabelReachable::reach :312-373 (reach_fst_input: fst2's ilabels)
  uint32_t reach_begin = 0xFFFFFFFFu, reach_end = 0xFFFFFFFFu;
  float reach_weight = INF;
  if (2u * n < ie - ib) {
    uint32_t reach_label = NO_LABEL;
    for (uint32_t pos = 0; pos < n; ++pos) {
      const ArcReg a = load_arc(arcs + pos);
      if (a.il == reach_label || (a.il != 0u && iv_member(r, s1, a.il))) {
        reach_label = a.il;
        if (reach_begin == 0xFFFFFFFFu) reach_begin = pos;
        reach_end = pos + 1;
        reach_weight = a.w < reach_weight ? a.w : reach_weight;
      }
    }
  } else {
    uint32_t begin_low, end_low = 0;
    for (uint32_t k = ib; k < ie; ++k) {
      begin_low = arcs_lower_bound(arcs, end_low, n, r.iv[2 * k]);
      end_low = arcs_lower_bound(arcs, begin_low, n, r.iv[2 * k + 1]);
      if (end_low > begin_low) {
        if (reach_begin == 0xFFFFFFFFu) reach_begin = begin_low;
        reach_end = end_low;
        for (uint32_t i = begin_low; i < end_low; ++i) {
          const float w = arcs[i].weight;
          reach_weight = w < reach_weight ? w : reach_weight;
        }
      }
    }
  }
  const bool reach_tr = reach_begin != 0xFFFFFFFFu;
  const bool reach_final = lfinal != INF && iv_member(r, s1, r.final_label);
  *lweight = 0.0f;
  *has_prefix = false;
  bool compute_weight = true;
  if (reach_tr) {
    if (reach_end - reach_begin == 1u && !reach_final) {
      *prefix = load_arc(arcs + reach_begin);
      *has_prefix = true;
      compute_weight = false;
    } else {
      *lweight = reach_weight;
    }
  }
  if (reach_final && compute_weight) *lweight = reach_tr ? (lfinal < *lweight ? lfinal : *lweight) : lfinal;
  return reach_tr || reach_final;
}

// what the filter stack needs to know about the composed state being expanded (set_state of the four filters)
struct StateCtx {
  FState fs;
  bool alleps2, noeps2;  // AltSequence: fst2's state has only input-epsilon arcs and is not final / has none
  uint32_t ntrsa;        // PushLabels: number of arcs of fst1's state
};

// filter_tr of PushLabels(PushWeights(LookAhead(AltSequence))) on the pair (a1 from fst1, a2 from fst2); may rewrite a2.
// Returns false when the pair is rejected (FilterState::new_no_state()).
__device__ bool la_filter(const Reach& r, const LaView& f2, const StateCtx& c, const ArcReg& a1, ArcReg& a2, FState* out) {
  if (c.fs.flabel != NO_LABEL) {  // PushLabels::pushed_label_filter_tr :285-336: a label is owed to fst1's output side
    if (a2.il != NO_LABEL) return false;  // fst2 stands still until it is consumed
    if (a1.ol == c.fs.flabel) {
      *out = FState{0u, 0.0f, NO_LABEL};  // start()
      return true;
    }
    if (a1.ol == 0u && (c.ntrsa == 1u || iv_member(r, a1.ns, c.fs.flabel))) {  // lookahead_label :215-224
      *out = c.fs;
      return true;
    }
    return false;
  }
  uint32_t f;  // AltSequenceComposeFilter::filter_tr :160-181
  if (a2.il == NO_LABEL) f = c.alleps2 ? REJECT : (c.noeps2 ? 0u : 1u);
  else if (a1.ol == NO_LABEL) f = c.fs.fs == 1u ? REJECT : 0u;
  else if (a1.ol == 0u) f = REJECT;
  else f = 0u;
  if (f == REJECT) return false;
  // LookAheadComposeFilter::lookahead_filter_tr :191-228: only an epsilon on fst1's output side looks ahead
  // (LOOKAHEAD_EPSILONS set, LOOKAHEAD_NON_EPSILONS not)
  const bool la_tr = a1.ol == 0u;
  float lweight = 0.0f;
  bool has_prefix = false;
  ArcReg prefix{0, 0, 0.0f, 0};
  if (la_tr && !lookahead_fst(r, f2, a1.ns, a2.ns, &lweight, &has_prefix, &prefix)) return false;
  // PushWeightsComposeFilter::filter_tr :144-176
  if (lweight == INF) return false;  // zero() futures are not allowed
  a2.w = wtimes(a2.w, lweight);
  a2.w -= c.fs.fweight;                                           // divide_assign, tropical_weight.rs:128-131
  const float q = floorf((lweight / KDELTA_F) + 0.5f) * KDELTA_F;  // quantize, semiring.rs:132-145 (lweight is finite)
  // PushLabelsComposeFilter::filter_tr :193-222 + push_label_filter_tr :339-400
  if (!la_tr || a2.ol != 0u || !has_prefix) {
    *out = FState{f, q, NO_LABEL};
    return true;
  }
  a2.il = prefix.il;  // the unique arc fst2 can take: take it now, fst1 owes its label
  a2.ol = prefix.ol;
  a2.w = wtimes(a2.w, prefix.w);
  a2.ns = prefix.ns;
  *out = FState{f, q, prefix.il};
  return true;
}

// arcs of the searched side with key == label: [*lo, *lo + *cnt) relative to the state's first arc
__device__ void equal_range(const wfst_tr* arcs, uint32_t n, bool by_ilabel, uint32_t key, uint32_t* lo_out, uint32_t* cnt_out) {
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

// everything the expansion of ONE composed state needs (compute_trs, compose_fst_op.rs:406-418)
struct Expand {
  StateCtx c;
  bool mi;                 // match_input :199-219: both matchers report priority = num_trs
  const wfst_tr* it_arcs;  // iterated side's arcs, searched side's arcs
  const wfst_tr* se_arcs;
  uint32_t n_it, n_se, sa, sb;
  float final_weight;      // compute_final_weight :420-449 after filter_final of PushWeights (:178-189) / PushLabels (:224-238)
};
__device__ Expand make_expand(const LaView& f1, const LaView& f2, uint64_t tlo, uint64_t thi) {
  Expand x;
  const uint32_t s1 = (uint32_t)(tlo >> 32), s2 = (uint32_t)tlo;
  x.c.fs = unpack_hi(thi);
  const uint4 r1 = f1.srec[s1], r2 = f2.srec[s2];
  const uint32_t n1 = r1.y, n2 = r2.y;
  const float fin1 = __uint_as_float(r1.z), fin2 = __uint_as_float(r2.z);
  x.c.alleps2 = (r2.w & SREC_ALL_IEPS) && !(fin2 != INF);
  x.c.noeps2 = (r2.w & SREC_NO_IEPS) != 0;
  x.c.ntrsa = n1;
  x.final_weight = INF;
  if (fin1 != INF && fin2 != INF) {
    float w1 = fin1 - x.c.fs.fweight;
    if (x.c.fs.flabel != NO_LABEL) w1 = INF;
    x.final_weight = wtimes(w1, fin2);
  }
  x.mi = n1 <= n2;
  x.it_arcs = x.mi ? f1.arcs + r1.x : f2.arcs + r2.x;
  x.se_arcs = x.mi ? f2.arcs + r2.x : f1.arcs + r1.x;
  x.n_it = x.mi ? n1 : n2;
  x.n_se = x.mi ? n2 : n1;
  x.sa = x.mi ? s2 : s1;
  x.sb = x.mi ? s1 : s2;
  return x;
}

// One item (item 0 = the loop pseudo-arc of ordered_expand :229-233, item j = the j-th arc of the iterated side) against
// everything the (multi-epsilon) matcher of the searched side yields for its label (matchers/multi_eps_matcher.rs:160-210
// over sorted_matcher.rs:124-184): an optional EpsLoop, then the arcs carrying the pushed label (fst1 side only:
// MULTI_EPS_LIST), then the arcs with key == label (0 for NO_LABEL).  Returns the number of pairs the filter accepts;
// with `write`, the composed arcs (add_tr :267-285) and their destination tuples go to position write_pos onwards;
// without, the first accepted pair is returned in *first.
__device__ uint32_t eval_item(const Reach& reach, const LaView& f2, const Expand& x, uint32_t j, bool write, uint32_t write_pos,
                              wfst_tr* arcs, uint64_t* a_lo, uint64_t* a_hi, Emitted* first = nullptr) {
  const bool mi = x.mi;
  const ArcReg ab = j == 0 ? (mi ? ArcReg{0u, NO_LABEL, 0.0f, x.sb} : ArcReg{NO_LABEL, 0u, 0.0f, x.sb}) : load_arc(x.it_arcs + (j - 1));
  const uint32_t label = mi ? ab.ol : ab.il;
  const uint32_t flabel = x.c.fs.flabel;
  uint32_t has_loop = 0, loA = 0, cntA = 0, loB = 0, cntB = 0;
  if (label == 0u) {
    has_loop = 1;
    equal_range(x.se_arcs, x.n_se, mi, 0u, &loB, &cntB);
  } else if (label == NO_LABEL) {
    if (!mi && flabel != NO_LABEL) equal_range(x.se_arcs, x.n_se, false, flabel, &loA, &cntA);
    equal_range(x.se_arcs, x.n_se, mi, 0u, &loB, &cntB);
  } else if (mi && flabel != NO_LABEL && label == flabel) {
    has_loop = 1;  // MULTI_EPS_LOOP on fst2: the pushed label behaves like an epsilon self-loop
  } else {
    equal_range(x.se_arcs, x.n_se, mi, label, &loB, &cntB);
  }
  const uint32_t n_match = has_loop + cntA + cntB;
  uint32_t k = 0;
  for (uint32_t m = 0; m < n_match; ++m) {
    ArcReg aa;
    if (m < has_loop) aa = mi ? ArcReg{NO_LABEL, 0u, 0.0f, x.sa} : ArcReg{0u, NO_LABEL, 0.0f, x.sa};  // eps_loop, mod.rs:98-105
    else if (m - has_loop < cntA) aa = load_arc(x.se_arcs + loA + (m - has_loop));
    else aa = load_arc(x.se_arcs + loB + (m - has_loop - cntA));
    const ArcReg a1 = mi ? ab : aa;  // arc1 from fst1, arc2 from fst2 (match_tr_selected :301-319)
    ArcReg a2 = mi ? aa : ab;
    FState nfs;
    if (!la_filter(reach, f2, x.c, a1, a2, &nfs)) continue;
    const uint4 arc = make_uint4(a1.il, a2.ol, __float_as_uint(wtimes(a1.w, a2.w)), 0u);
    const uint64_t dlo = ((uint64_t)a1.ns << 32) | a2.ns, dhi = pack_hi(nfs);
    if (write) {
      const uint32_t e = write_pos + k;
      *reinterpret_cast<uint4*>(arcs + e) = arc;
      a_lo[e] = dlo;
      a_hi[e] = dhi;
    } else if (first && k == 0) {  // most items emit one arc: the counting pass keeps it, the writing pass is skipped
      first->arc = arc;
      first->lo = dlo;
      first->hi = dhi;
    }
    k++;
  }
  return k;
}

// one wave = one (fst1, fst2[p]) problem; fst1 and its reachability data are shared by the batch
__global__ void __launch_bounds__(64) compose_lookahead_kernel(LaView f1, const LaView* __restrict__ f2s, Reach reach, LaCaps caps,
                                                               char* __restrict__ arena_base, size_t arena_stride,
                                                               LaResult* __restrict__ results, uint32_t switch_states,
                                                               uint32_t switch_width) {
  const LaView f2 = f2s[blockIdx.x];
  const LaArena ar = la_carve(arena_base + (size_t)blockIdx.x * arena_stride, caps, nullptr);
  LaResult* result = results + blockIdx.x;
  const uint32_t lane = lane_id();
  const uint32_t hmask = caps.H - 1;
  LaResult res{LA_OK, 0, 0, 0};
  uint32_t n_states = 0, n_arcs = 0, n_levels = 0;
  bool ok = true;
#ifdef WFST_LA_PHASE
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, pt = 0;  // tuning: wall-clock per phase of a level (printed by problem 0)
#endif
  if (f1.start >= 0 && f2.start >= 0) {  // compute_start, compose_fst_op.rs:389-404
    for (uint32_t i = lane; i < caps.H; i += 64) {
      ar.klo[i] = K_EMPTY;
      ar.khi[i] = KHI_UNSET;
    }
    __syncthreads();
    if (lane == 0) {
      const uint64_t lo0 = ((uint64_t)(uint32_t)f1.start << 32) | (uint32_t)f2.start;
      const uint64_t hi0 = pack_hi(FState{0u, 0.0f, NO_LABEL});  // PushLabels::start :163-165
      const uint32_t slot = hash_128(lo0, hi0) & hmask;
      ar.klo[slot] = lo0;
      ar.khi[slot] = hi0;
      ar.hvals[slot] = 0;
      ar.t_lo[0] = lo0;
      ar.t_hi[0] = hi0;
    }
    __syncthreads();
    n_states = 1;
    uint32_t lo = 0, hi = 1;
#ifdef WFST_LA_PHASE
    pt = wall_clock64();
#define LA_PH(i) do { const unsigned long long n_ = wall_clock64(); ph[i] += n_ - pt; pt = n_; } while (0)
#else
#define LA_PH(i) do { } while (0)
#endif
    while (lo < hi && ok) {  // LazyFst::compute, lazy_fst.rs:235-259: level = ids [lo, hi)
      n_levels++;
      const uint32_t level_begin = n_arcs;
      for (uint32_t q = lo; q < hi && ok; ++q) {
        if (lane == 0) ar.off[q] = n_arcs;
        const Expand x = make_expand(f1, f2, ld_l2(&ar.t_lo[q]), ld_l2(&ar.t_hi[q]));
        if (lane == 0) ar.fin[q] = x.final_weight;
#ifdef WFST_LA_PHASE
        if (x.n_it == 0xFFFFFFFFu) ar.fin[q] = 0.0f;  // (consume x before the stamp)
#endif
        LA_PH(0);
        const uint32_t n_items = x.n_it + 1;
        for (uint32_t base = 0; base < n_items; base += 64) {
          const uint32_t j = base + lane;
          const bool have = j < n_items;
          Emitted em;
          const uint32_t cnt = have ? eval_item(reach, f2, x, j, false, 0, nullptr, nullptr, nullptr, &em) : 0u;
          uint32_t total, pos;
          if (!__any(cnt > 1u)) {  // (the usual case — every item emits at most one arc: positions from one ballot, no shuffles)
            const uint64_t m1 = __ballot(cnt == 1u);
            pos = lanes_below(m1);
            total = (uint32_t)__popcll(m1);
          } else {
            pos = wave_excl_scan(cnt, lane, &total);
          }
          if (total == 0) continue;
          if ((uint64_t)n_arcs + total > caps.A) {
            res.status = LA_OVERFLOW_ARCS;
            ok = false;
            break;
          }
          if (cnt == 1) {
            *reinterpret_cast<uint4*>(ar.arcs + n_arcs + pos) = em.arc;
            ar.a_lo[n_arcs + pos] = em.lo;
            ar.a_hi[n_arcs + pos] = em.hi;
          } else if (cnt) {
            eval_item(reach, f2, x, j, true, n_arcs + pos, ar.arcs, ar.a_lo, ar.a_hi);
          }
          n_arcs += total;
        }
        LA_PH(1);
      }
      if (!ok) break;
      if (lane == 0) ar.off[hi] = n_arcs;
      __syncthreads();
      LA_PH(2);
      // StateTable::find_id (state_table.rs:49-59) for this level's destinations, in emission order
      uint32_t n_new = 0;
      for (uint32_t base = level_begin; base < n_arcs && ok; base += 64) {
        const uint32_t e = base + lane;
        const bool have = e < n_arcs;
        uint64_t klo = K_EMPTY, khi = 0;
        if (have) {
          klo = ld_l2(&ar.a_lo[e]);
          khi = ld_l2(&ar.a_hi[e]);
        }
        uint32_t first = lane;  // lowest lane of the chunk holding the same tuple
        const uint32_t n_chunk = min(64u, n_arcs - base);  // (a decoding level emits two or three arcs: not 64 rounds of shuffles)
        for (uint32_t i = 0; i < n_chunk; ++i) {
          const uint64_t li = shfl64(klo, i), hi_i = shfl64(khi, i);
          if (have && i < first && li == klo && hi_i == khi) first = i;
        }
        const bool probe = have && first == lane;
        uint32_t slot = hash_128(klo, khi) & hmask;
        bool done = !probe, existed = false;
        while (__any(!done)) {
          uint64_t prev = 0;
          if (!done) prev = atomicCAS((unsigned long long*)&ar.klo[slot], (unsigned long long)K_EMPTY, (unsigned long long)klo);
          const bool won = !done && prev == K_EMPTY;
          if (won) st_l2(&ar.khi[slot], khi);  // (a lane that reads it too early sees KHI_UNSET and retries the slot)
          const bool same_lo = !done && !won && prev == klo;
          uint64_t h = KHI_UNSET;
          if (same_lo) h = ld_l2(&ar.khi[slot]);
          if (won) {
            done = true;
          } else if (same_lo && h == khi) {
            existed = true;
            done = true;
          } else if (!done && !(same_lo && h == KHI_UNSET)) {
            slot = (slot + 1) & hmask;
          }
        }
        const bool is_new = probe && !existed;
        const uint64_t nm = __ballot(is_new);
        const uint32_t cn = (uint32_t)__popcll(nm);
        if ((uint64_t)hi + n_new + cn > caps.S) {
          res.status = LA_OVERFLOW_STATES;
          ok = false;
          break;
        }
        uint32_t id = 0;
        if (is_new) {
          id = hi + n_new + lanes_below(nm);
          st_l2(&ar.hvals[slot], id);
          ar.t_lo[id] = klo;
          ar.t_hi[id] = khi;
        } else if (probe) {
          id = ld_l2(&ar.hvals[slot]);
        }
        const uint32_t id_all = __shfl(id, first);
        if (have) ar.arcs[e].nextstate = id_all;
        n_new += cn;
        __syncthreads();  // ids of this chunk are visible to the next one
      }
      if (!ok) break;
      LA_PH(3);
      lo = hi;
      hi += n_new;
      n_states = hi;
      // a big composition, or a frontier too wide for one wave: thousands of waves do it faster (deep and narrow results,
      // the decoding lattices, stay here: the wide path pays five launches and a read-back per level)
      if ((n_states > switch_states || (switch_states != 0xFFFFFFFFu && n_new > switch_width)) && lo < hi) {
        res.status = LA_SWITCH_WIDE;
        ok = false;
      }
      __syncthreads();
    }
  }
  res.n_states = n_states;
  res.n_arcs = n_arcs;
  res.n_levels = n_levels;
#ifdef WFST_LA_PHASE
  if (lane == 0 && blockIdx.x == 0)
    printf("la phases (us per level, %u levels): expand %.2f items %.2f sync %.2f hash %.2f\n", n_levels, ph[0] / 100.0 / n_levels,
           ph[1] / 100.0 / n_levels, ph[2] / 100.0 / n_levels, ph[3] / 100.0 / n_levels);
#endif
  if (lane == 0) *result = res;
}

size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }

uint32_t next_pow2(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return (uint32_t)p;
}

LaView view_of(const wfst_fst* f) {
  return LaView{f->dev.arcs, f->dev.srec, f->n_states, (int32_t)f->start};
}

// properties of the result: PushLabels(PushWeights(LookAhead(AltSequence))).properties(compose_properties(p1, p2))
// (push_labels_compose_filter.rs:261-268, push_weights :198-200, lookahead :308-314) then LazyFst::compute's
// set_properties (lazy_fst.rs:260); no start state: VectorFst::new()
uint64_t lookahead_result_props(uint64_t p1, uint64_t p2, bool has_start) {
  using namespace props;
  if (!has_start) return NULL_PROPS;
  const uint64_t weight_invariant = ACCEPTOR | NOT_ACCEPTOR | I_DETERMINISTIC | NOT_I_DETERMINISTIC | O_DETERMINISTIC |
                                    NOT_O_DETERMINISTIC | EPSILONS | NO_EPSILONS | I_EPSILONS | NO_I_EPSILONS | O_EPSILONS |
                                    NO_O_EPSILONS | I_LABEL_SORTED | NOT_I_LABEL_SORTED | O_LABEL_SORTED |
                                    NOT_O_LABEL_SORTED | CYCLIC | ACYCLIC | INITIAL_CYCLIC | INITIAL_ACYCLIC | TOP_SORTED |
                                    NOT_TOP_SORTED | ACCESSIBLE | NOT_ACCESSIBLE | COACCESSIBLE | NOT_COACCESSIBLE | STRING |
                                    NOT_STRING;  // properties.rs:436-465
  const uint64_t o_label_invariant = I_DETERMINISTIC | NOT_I_DETERMINISTIC | I_EPSILONS | NO_I_EPSILONS | I_LABEL_SORTED |
                                     NOT_I_LABEL_SORTED | WEIGHTED | UNWEIGHTED | CYCLIC | ACYCLIC | INITIAL_CYCLIC |
                                     INITIAL_ACYCLIC | TOP_SORTED | NOT_TOP_SORTED | ACCESSIBLE | NOT_ACCESSIBLE |
                                     COACCESSIBLE | NOT_COACCESSIBLE | STRING | NOT_STRING | WEIGHTED_CYCLES |
                                     UNWEIGHTED_CYCLES;  // :409-432
  return compose(p1, p2) & weight_invariant & o_label_invariant;
}

}  // namespace

// MatcherFst::new for an output look-ahead matcher (matcher_fst.rs:55-71): reachability data of fst1 on its output labels,
// fst1 relabelled and re-sorted by olabel (LabelLookAheadRelabeler::init, label_lookahead_relabeler.rs:11-25)
wfst_lookahead* lookahead_create(wfst_ctx* ctx, const wfst_fst* fst1) {
  ensure_host(fst1);
  std::unique_ptr<wfst_lookahead> la(new wfst_lookahead());
  la->ctx = ctx;
  const HostCsr& h = fst1->host;
  la->data.compute(fst1->n_states, h.offsets.data(), h.arcs.data(), h.finals.data(), /*reach_input=*/false);
  HostCsr r = h;
  const uint64_t p = la->data.relabel_fst(fst1->n_states, r.offsets.data(), r.arcs.data(), fst1->props, /*relabel_input=*/false);
  la->fst1 = make_host_fst(ctx, fst1->n_states, fst1->start, p, std::move(r));
  ensure_device(la->fst1);
  la->d_iv_off.reset(new DBuf<uint32_t>(*ctx->pool, la->data.iv_off.size()));
  la->d_iv.reset(new DBuf<uint32_t>(*ctx->pool, std::max<size_t>(la->data.iv.size(), 2)));
  HIP_CHECK(hipMemcpyAsync(la->d_iv_off->p, la->data.iv_off.data(), la->data.iv_off.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  if (!la->data.iv.empty())
    HIP_CHECK(hipMemcpyAsync(la->d_iv->p, la->data.iv.data(), la->data.iv.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return la.release();
}

// LabelLookAheadRelabeler::relabel(fst2, addon, relabel_input = true) (label_lookahead_relabeler.rs:27-41) followed by the
// caller's tr_sort(ILabelCompare) (cmds/compose.rs:151; a no-op after relabel_fst's own sort)
wfst_fst* lookahead_relabel(wfst_lookahead* la, const wfst_fst* fst2) {
  ensure_host(fst2);
  HostCsr r = fst2->host;
  uint64_t p = la->data.relabel_fst(fst2->n_states, r.offsets.data(), r.arcs.data(), fst2->props, /*relabel_input=*/true);
  p = tr_sort_props(p, true);
  wfst_fst* out = make_host_fst(la->ctx ? la->ctx : fst2->ctx, fst2->n_states, fst2->start, p, std::move(r));
  return out;
}

namespace {

// the look-ahead composition as a policy of the wide driver (compose_wide.h)
struct LaPolicy {
  LaView f1, f2;
  Reach reach;
  using Expand = wfst::Expand;
  __device__ Expand make_expand(uint64_t tlo, uint64_t thi) const { return wfst::make_expand(f1, f2, tlo, thi); }
  __device__ uint32_t eval_item(const Expand& x, uint32_t j, bool write, uint32_t write_pos, wfst_tr* arcs, uint64_t* a_lo,
                                uint64_t* a_hi, Emitted* first) const {
    return wfst::eval_item(reach, f2, x, j, write, write_pos, arcs, a_lo, a_hi, first);
  }
};

wfst_fst* compose_lookahead_wide(wfst_ctx* ctx, const wfst_lookahead* la, const wfst_fst* f1, const wfst_fst* fst2,
                                 uint64_t out_props, uint64_t est_s, uint64_t est_a) {
  const LaPolicy pol{view_of(f1), view_of(fst2), Reach{la->d_iv_off->p, la->d_iv->p, la->data.final_label}};
  WideOutput w;
  const double d1 = (double)f1->n_arcs / std::max<uint32_t>(f1->n_states, 1), d2 = (double)fst2->n_arcs / std::max<uint32_t>(fst2->n_states, 1);
  // A serving loop composes many inputs of one kind against the operand: the last result sizes the arena, + 1/8.  The hint
  // is best effort: it only applies to an input of about the size of the one it came from (within a factor of two in
  // arcs — one 90M-state result must not make every later small composition ask for gigabytes), and when the pool cannot
  // supply the hinted arena the composition runs from the unhinted estimate instead of failing.
  const uint64_t est_s0 = est_s, est_a0 = est_a;
  bool hinted = false;
  if (!std::getenv("WFST_WIDE_EST_STATES")) {
    const uint64_t ls = la->last_wide_states.load(std::memory_order_relaxed), lw = la->last_wide_arcs.load(std::memory_order_relaxed);
    const uint64_t li = la->last_wide_input_arcs.load(std::memory_order_relaxed), in = fst2->n_arcs;
    if (ls != 0 && in <= 2 * li + 16 && li <= 2 * in + 16 && ls + ls / 8 + 1024 < 0x7FFFFFF0ull && lw + lw / 4 + 65536 < 0x7FFFFFF0ull) {
      est_s = std::max<uint64_t>(est_s, ls + ls / 8 + 1024);
      est_a = std::max<uint64_t>(est_a, lw + lw / 4 + 65536);  // (the arc arena is cut into 64 slices: room for their skew)
      hinted = est_s != est_s0 || est_a != est_a0;
    }
  }
  const uint64_t t0 = ((uint64_t)(uint32_t)f1->start << 32) | (uint32_t)fst2->start;
  try {
    run_wide(ctx, pol, t0, pack_hi(FState{0u, 0.0f, NO_LABEL}), est_s, est_a, 1.0 + std::min(d1, d2), w);
  } catch (const Error&) {
    if (!hinted) throw;
    (void)hipGetLastError();  // (a refused allocation must not surface at the next launch check)
    w = WideOutput{};
    run_wide(ctx, pol, t0, pack_hi(FState{0u, 0.0f, NO_LABEL}), est_s0, est_a0, 1.0 + std::min(d1, d2), w);
  }
  la->last_wide_states.store(w.n_states, std::memory_order_relaxed);
  la->last_wide_arcs.store(w.n_arcs, std::memory_order_relaxed);
  la->last_wide_input_arcs.store(fst2->n_arcs, std::memory_order_relaxed);
  ctx->stats.compose_states = w.n_states;
  ctx->stats.compose_arcs = w.n_arcs;
  return adopt_device(ctx, w.n_states, w.n_arcs, 0, out_props, w.off, w.arcs, w.fin);
}

}  // namespace

namespace {
void check_lookahead_operands(const wfst_lookahead* la, const wfst_fst* fst2) {
  using namespace props;
  if (!la->fst1) throw Error("compose_lookahead: the look-ahead handle has no device FST (host-only handle)");
  // SortedMatcher(fst1, MatchOutput) and reach_init (label_reachable.rs:275-291) need the sorted bits
  if (!(la->fst1->props & O_LABEL_SORTED)) throw Error("compose_lookahead: the 1st FST is not sorted on output labels");
  if (!(fst2->props & I_LABEL_SORTED)) throw Error("LabelReachable::ReachInit: Fst is not sorted");
}
}  // namespace

// n problems (la's fst1, fst2s[i]) in one launch of the single-wave kernel, one wave each; results that outgrow it (or its
// arena) are redone one by one: bigger arena, then the wide path
void compose_lookahead_batch(wfst_ctx* ctx, const wfst_lookahead* la, const wfst_fst* const* fst2s, size_t n, wfst_fst** outs) {
  for (size_t i = 0; i < n; ++i) {
    outs[i] = nullptr;
    if (!fst2s[i]) throw Error("null pointer");
    check_lookahead_operands(la, fst2s[i]);
  }
  const wfst_fst* f1 = la->fst1;
  ensure_device(const_cast<wfst_fst*>(f1));
  hipStream_t st = ctx->stream;
  std::vector<size_t> todo;
  std::vector<LaView> views;
  std::vector<std::unique_ptr<wfst_fst>> done(n);  // (frees what was built if a later problem throws)
  {
    for (size_t i = 0; i < n; ++i) {
      ensure_device(const_cast<wfst_fst*>(fst2s[i]));
      const bool has_start = f1->start >= 0 && fst2s[i]->start >= 0;
      if (!has_start) {
        HostCsr h;
        h.offsets.push_back(0);
        done[i].reset(make_host_fst(ctx, 0, -1, lookahead_result_props(f1->props, fst2s[i]->props, false), std::move(h)));
      } else {
        todo.push_back(i);
        views.push_back(view_of(fst2s[i]));
      }
    }
    int force = 0;  // WFST_LOOKAHEAD_PATH=wave|wide pins one driver (tests)
    if (const char* e = std::getenv("WFST_LOOKAHEAD_PATH")) force = std::strcmp(e, "wide") == 0 ? 2 : (std::strcmp(e, "wave") == 0 ? 1 : 0);
    std::vector<size_t> redo;  // indices into todo
    if (!todo.empty() && force != 2) {
      const uint64_t est_s = 2ull * WIDE_SWITCH_STATES + 256;  // the single wave stops soon after WIDE_SWITCH_STATES
      const LaCaps caps{(uint32_t)est_s, (uint32_t)(4 * est_s), next_pow2(2 * est_s + 128)};
      size_t stride = 0;
      la_carve(nullptr, caps, &stride);
      const size_t m = todo.size();
      DBuf<char> arena(*ctx->pool, stride * m);
      DBuf<LaView> d_views(*ctx->pool, m);
      DBuf<LaResult> d_res(*ctx->pool, m);
      HIP_CHECK(hipMemcpyAsync(d_views.p, views.data(), m * sizeof(LaView), hipMemcpyHostToDevice, st));
      const Reach reach{la->d_iv_off->p, la->d_iv->p, la->data.final_label};
      // a lone composition hands over early (the wide driver costs ~15 us per narrow level, this kernel ~3 us per STATE);
      // in a batch the other waves are busy meanwhile, and the wide driver would take the problems one by one
      compose_lookahead_kernel<<<(uint32_t)m, 64, 0, st>>>(view_of(f1), d_views.p, reach, caps, arena.p, stride, d_res.p,
                                                            force == 1 ? 0xFFFFFFFFu : (m == 1 ? WIDE_SWITCH_STATES_ONE : WIDE_SWITCH_STATES),
                                                            m == 1 ? WIDE_SWITCH_WIDTH_ONE : WIDE_SWITCH_WIDTH);
      HIP_CHECK(hipGetLastError());
      std::vector<LaResult> res(m);
      HIP_CHECK(hipMemcpyAsync(res.data(), d_res.p, m * sizeof(LaResult), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      std::vector<AdoptDesc> ok;
      std::vector<size_t> ok_i;
      for (size_t k = 0; k < m; ++k) {
        const size_t i = todo[k];
        const LaResult& r = res[k];
        if (r.status != LA_OK) {
          redo.push_back(k);
          continue;
        }
        const LaArena ar = la_carve(arena.p + k * stride, caps, nullptr);
        ok.push_back(AdoptDesc{r.n_states, r.n_arcs, r.n_states ? 0 : -1, lookahead_result_props(f1->props, fst2s[i]->props, true),
                               ar.off, ar.arcs, ar.fin});
        ok_i.push_back(i);
        ctx->stats.compose_states = r.n_states;
        ctx->stats.compose_arcs = r.n_arcs;
      }
      if (!ok.empty()) {  // the finished results leave the problem arenas together (one allocation, one synchronisation)
        std::vector<wfst_fst*> made(ok.size(), nullptr);
        adopt_device_many(ctx, ok.size(), ok.data(), made.data());
        for (size_t q = 0; q < ok.size(); ++q) done[ok_i[q]].reset(made[q]);
      }
    } else {
      for (size_t k = 0; k < todo.size(); ++k) redo.push_back(k);
    }
    for (size_t k : redo) {
      const size_t i = todo[k];
      const wfst_fst* fst2 = fst2s[i];
      const uint64_t out_props = lookahead_result_props(f1->props, fst2->props, true);
      if (force == 1) {  // pinned to the single wave: grow its arena until the result fits
        uint64_t est_s = 16ull * WIDE_SWITCH_STATES;
        for (int attempt = 0;; ++attempt) {
          if (est_s > 0x1FFFFFF0ull) throw Error("compose_lookahead: composition too large");
          const LaCaps caps{(uint32_t)est_s, (uint32_t)(4 * est_s), next_pow2(2 * est_s + 128)};
          size_t stride = 0;
          la_carve(nullptr, caps, &stride);
          DBuf<char> arena(*ctx->pool, stride);
          DBuf<LaView> d_view(*ctx->pool, 1);
          DBuf<LaResult> d_res(*ctx->pool, 1);
          const LaView v = view_of(fst2);
          HIP_CHECK(hipMemcpyAsync(d_view.p, &v, sizeof(LaView), hipMemcpyHostToDevice, st));
          const Reach reach{la->d_iv_off->p, la->d_iv->p, la->data.final_label};
          compose_lookahead_kernel<<<1, 64, 0, st>>>(view_of(f1), d_view.p, reach, caps, arena.p, stride, d_res.p, 0xFFFFFFFFu, 0u);
          HIP_CHECK(hipGetLastError());
          LaResult r;
          HIP_CHECK(hipMemcpyAsync(&r, d_res.p, sizeof(r), hipMemcpyDeviceToHost, st));
          HIP_CHECK(hipStreamSynchronize(st));
          if (r.status == LA_OK) {
            const LaArena ar = la_carve(arena.p, caps, nullptr);
            done[i].reset(adopt_device(ctx, r.n_states, r.n_arcs, r.n_states ? 0 : -1, out_props, ar.off, ar.arcs, ar.fin));
            ctx->stats.compose_states = r.n_states;
            ctx->stats.compose_arcs = r.n_arcs;
            break;
          }
          ctx->stats.compose_retries++;
          if (attempt > 24) throw Error("compose_lookahead: arena overflow after retries");
          est_s *= 4;
        }
      } else {
        const uint64_t ws = 8ull * std::max<uint64_t>(std::min(f1->n_states, fst2->n_states), 4096);
        done[i].reset(compose_lookahead_wide(ctx, la, f1, fst2, out_props, ws, 4 * ws));
      }
    }
  }
  for (size_t i = 0; i < n; ++i) outs[i] = done[i].release();
}

wfst_fst* compose_lookahead(wfst_ctx* ctx, const wfst_lookahead* la, const wfst_fst* fst2) {
  wfst_fst* out = nullptr;
  compose_lookahead_batch(ctx, la, &fst2, 1, &out);
  return out;
}

}  // namespace wfst

wfst_lookahead::~wfst_lookahead() { delete fst1; }
