// Host side of look-ahead composition: the per-state sets of output labels reachable through
// output-epsilon paths, as sorted disjoint intervals over relabelled labels, and the relabelling of both operands.
//
// Mirrors rustfst::algorithms::compose::{LabelReachable (label_reachable.rs:135-273), StateReachable
// (state_reachable.rs:20-87), IntervalReachVisitor (interval_reach_visitor.rs:28-96), IntervalSet::normalize
// (interval_set.rs:153-190), condense (algorithms/condense.rs:15-55), LabelReachableData::{relabel, relabel_fst}
// (label_reachable.rs:52-93)}.  The label -> index map is the order in which a depth-first visit (dfs_visit.rs:97-187)
// of the transformed FST discovers the per-label sink states, so the visit order is reproduced exactly; everything
// works on flat CSR arrays with explicit stacks (decoding graphs have millions of states).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <thread>
#include <unordered_map>

#include "common.h"
#include "fst_props.h"
#include "lookahead.h"

namespace wfst {
namespace {

constexpr uint32_t NO_LABEL = WFST_NO_LABEL;
constexpr uint32_t UNASSIGNED = 0xFFFFFFFFu;

// successor lists of the transformed FST (label_reachable.rs:172-248): only the targets matter to the visit
struct Graph {
  std::vector<uint32_t> off, dst;
  std::vector<uint8_t> is_final;
  uint32_t start = 0;
  uint32_t n() const { return (uint32_t)is_final.size(); }
};

// the depth-first visit of dfs_visit.rs:97-187 (every state, roots: start, then 0, 1, 2, ... among the unvisited)
template <class V>
void depth_first(const Graph& g, V& v) {
  enum : uint8_t { White, Grey, Black };
  const uint32_t n = g.n();
  std::vector<uint8_t> color(n, White);
  std::vector<std::pair<uint32_t, uint32_t>> stack;  // {state, next arc position}
  uint32_t root = g.start;
  while (root < n) {
    color[root] = Grey;
    stack.push_back({root, g.off[root]});
    v.discover(root);
    while (!stack.empty()) {
      const uint32_t s = stack.back().first;
      const uint32_t pos = stack.back().second;
      if (pos >= g.off[s + 1]) {
        color[s] = Black;
        stack.pop_back();
        if (!stack.empty()) {
          v.finish(s, true, stack.back().first);
          stack.back().second++;
        } else {
          v.finish(s, false, 0);
        }
        continue;
      }
      const uint32_t t = g.dst[pos];
      if (color[t] == White) {
        color[t] = Grey;
        v.discover(t);
        stack.push_back({t, g.off[t]});
      } else if (color[t] == Grey) {
        v.back(s, t);
        stack.back().second++;
      } else {
        v.cross(s, t);
        stack.back().second++;
      }
    }
    root = root == g.start ? 0 : root + 1;
    while (root < n && color[root] != White) root++;
  }
}

using Intervals = std::vector<std::pair<uint32_t, uint32_t>>;  // half-open [begin, end)

// IntervalSet::normalize (interval_set.rs:156-190): order by begin (longer first on ties), merge overlapping and adjacent
void normalize(Intervals& iv) {
  auto less = [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) {
    return a.first != b.first ? a.first < b.first : a.second > b.second;
  };
  if (iv.size() <= 32) {  // the usual size (a state's own labels): insertion sort, no temporary buffer (equal keys are equal
                          // pairs, so stability is moot)
    for (size_t i = 1; i < iv.size(); ++i) {
      const std::pair<uint32_t, uint32_t> v = iv[i];
      size_t j = i;
      for (; j > 0 && less(v, iv[j - 1]); --j) iv[j] = iv[j - 1];
      iv[j] = v;
    }
  } else {
    std::sort(iv.begin(), iv.end(), less);
  }
  size_t w = 0;
  for (size_t i = 0; i < iv.size();) {
    std::pair<uint32_t, uint32_t> cur = iv[i];
    if (cur.first == cur.second) throw Error("IntervalSet: empty interval");  // (the reference does not terminate on it)
    size_t j = i + 1;
    for (; j < iv.size(); ++j) {
      if (iv[j].first > cur.second) break;
      if (iv[j].second > cur.second) cur.second = iv[j].second;
    }
    iv[w++] = cur;
    i = j;
  }
  iv.resize(w);
}

// IntervalReachVisitor (interval_reach_visitor.rs:28-96).  The reference keeps one IntervalSet per state and pours a
// child's set into its parent when the child is finished (tree arcs) or met again (cross arcs).  On an acyclic graph every
// successor of s is finished when s is, so the same set — the normalised union over s's arcs of the successors' sets,
// plus s's own interval if it is final — is assembled here in ONE go at finish(s), from a flat arena that holds every
// finished state's intervals back to back: no vector per state (five million small allocations were most of the visit),
// and the per-state CSR the caller wants is a copy out of the arena.
struct ReachVisitor {
  const Graph& g;
  Intervals arena;                     // intervals of the finished states, in finishing order
  std::vector<uint64_t> set_off;       // [n] first interval of the state in the arena
  std::vector<uint32_t> set_len;       // [n]
  std::vector<uint32_t> state2index, own_begin;
  Intervals scratch;
  uint32_t index = 1;
  bool cyclic = false;
  // the super-initial state (the visit's root) would collect the sets of every state nothing points at — millions of
  // intervals to sort for a set that find_intervals drops (label_reachable.rs:258): it is left empty
  uint32_t root;
  explicit ReachVisitor(const Graph& gr)
      : g(gr), set_off(gr.n(), 0), set_len(gr.n(), 0), state2index(gr.n(), UNASSIGNED), own_begin(gr.n(), 0), root(gr.start) {
    arena.reserve(2 * gr.dst.size() + gr.n());  // (a decoding graph's sets hold ~1.5 intervals per arc; regrowing 100+ MB is a copy)
  }
  void discover(uint32_t s) {
    if (g.is_final[s]) {
      own_begin[s] = index;
      state2index[s] = index;
      index++;
    }
  }
  void back(uint32_t, uint32_t) { cyclic = true; }  // (the caller falls back to the condensation)
  void cross(uint32_t, uint32_t) {}
  void finish(uint32_t s, bool, uint32_t) {
    if (cyclic) return;
    scratch.clear();
    if (g.is_final[s]) scratch.push_back({own_begin[s], index});  // every final state discovered below s has an index in [mine, index)
    if (s != root)
      for (uint32_t k = g.off[s]; k < g.off[s + 1]; ++k) {
        const uint32_t t = g.dst[k];
        const auto* b = arena.data() + set_off[t];
        scratch.insert(scratch.end(), b, b + set_len[t]);
      }
    normalize(scratch);
    set_off[s] = arena.size();
    set_len[s] = (uint32_t)scratch.size();
    arena.insert(arena.end(), scratch.begin(), scratch.end());
  }
};

// SccVisitor (visitors/scc_visitors.rs:10-180): Tarjan over the same visit; component ids are reversed at the end, so
// they are a topological numbering of the condensation
struct SccVisitor {
  std::vector<int32_t> scc, dfnumber, lowlink;
  std::vector<uint8_t> onstack;
  std::vector<uint32_t> stack;
  int32_t nstates = 0, nscc = 0;
  bool cyclic = false;
  explicit SccVisitor(uint32_t n) : scc(n, -1), dfnumber(n, -1), lowlink(n, -1), onstack(n, 0) {}
  void discover(uint32_t s) {
    stack.push_back(s);
    dfnumber[s] = lowlink[s] = nstates++;
    onstack[s] = 1;
  }
  void back(uint32_t s, uint32_t t) {
    cyclic = true;
    if (dfnumber[t] < lowlink[s]) lowlink[s] = dfnumber[t];
  }
  void cross(uint32_t s, uint32_t t) {
    if (dfnumber[t] < dfnumber[s] && onstack[t] && dfnumber[t] < lowlink[s]) lowlink[s] = dfnumber[t];
  }
  void finish(uint32_t s, bool has_parent, uint32_t parent) {
    if (dfnumber[s] == lowlink[s]) {
      uint32_t t;
      do {
        t = stack.back();
        stack.pop_back();
        scc[t] = nscc;
        onstack[t] = 0;
      } while (t != s);
      nscc++;
    }
    if (has_parent && lowlink[s] < lowlink[parent]) lowlink[parent] = lowlink[s];
  }
  void done() {
    for (auto& c : scc) c = nscc - 1 - c;
  }
};

// StateReachable::new (state_reachable.rs:26-76): interval sets + discovery index of the final states; cyclic inputs go
// through their condensation (no final state may lie on a cycle).  The sets come back as slices of one arena.
struct ReachSets {
  Intervals arena;
  std::vector<uint64_t> off;  // per state of g
  std::vector<uint32_t> len;
  std::vector<uint32_t> state2index;
};
void state_reachable(const Graph& g, ReachSets& out) {
  const bool timing = std::getenv("WFST_HOST_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto t0 = now();
  {
    // the visit that builds the sets also answers "acyclic?" (no back arc: compute_and_update_properties(ACYCLIC),
    // label_reachable.rs:146); decoding graphs are acyclic in their epsilon structure, so this is usually all there is
    ReachVisitor rv(g);
    depth_first(g, rv);
    if (timing) std::fprintf(stderr, "[wfst]   reach visit %.1f ms (cyclic %d)\n", ms(t0, now()), (int)rv.cyclic);
    if (!rv.cyclic) {
      out.arena = std::move(rv.arena);
      out.off = std::move(rv.set_off);
      out.len = std::move(rv.set_len);
      out.state2index = std::move(rv.state2index);
      return;
    }
  }
  const auto t1 = now();
  SccVisitor sv(g.n());
  depth_first(g, sv);
  sv.done();
  const uint32_t nc = (uint32_t)sv.nscc;
  Graph c;  // condense (condense.rs:15-55): arcs between different components, in source-state then arc order
  c.is_final.assign(nc, 0);
  c.start = (uint32_t)sv.scc[g.start];
  std::vector<uint32_t> members(nc, 0), deg(nc + 1, 0);
  for (uint32_t s = 0; s < g.n(); ++s) {
    const uint32_t cs = (uint32_t)sv.scc[s];
    members[cs]++;
    if (g.is_final[s]) c.is_final[cs] = 1;
    for (uint32_t k = g.off[s]; k < g.off[s + 1]; ++k)
      if ((uint32_t)sv.scc[g.dst[k]] != cs) deg[cs + 1]++;
  }
  c.off.assign(nc + 1, 0);
  for (uint32_t i = 0; i < nc; ++i) c.off[i + 1] = c.off[i] + deg[i + 1];
  c.dst.resize(c.off[nc]);
  std::vector<uint32_t> cur(c.off.begin(), c.off.end() - 1);
  for (uint32_t s = 0; s < g.n(); ++s) {
    const uint32_t cs = (uint32_t)sv.scc[s];
    for (uint32_t k = g.off[s]; k < g.off[s + 1]; ++k) {
      const uint32_t ct = (uint32_t)sv.scc[g.dst[k]];
      if (ct != cs) c.dst[cur[cs]++] = ct;
    }
  }
  const auto t2 = now();
  ReachVisitor rv(c);
  depth_first(c, rv);
  if (rv.cyclic) throw Error("IntervalReachVisitor: cyclic input");  // (a condensation is acyclic)
  const auto t3 = now();
  out.arena = std::move(rv.arena);
  out.off.assign(g.n(), 0);
  out.len.assign(g.n(), 0);
  out.state2index.assign(g.n(), UNASSIGNED);
  for (uint32_t s = 0; s < g.n(); ++s) {
    const uint32_t cs = (uint32_t)sv.scc[s];
    if (c.is_final[cs] && members[cs] > 1) throw Error("StateReachable: Final state contained in a cycle");
    out.off[s] = rv.set_off[cs];
    out.len[s] = rv.set_len[cs];
    out.state2index[s] = rv.state2index[cs];
  }
  if (timing)
    std::fprintf(stderr, "[wfst]   scc + condense %.1f ms, reach visit %.1f ms, copy back %.1f ms\n", ms(t1, t2), ms(t2, t3), ms(t3, now()));
}

}  // namespace

// LabelReachable::compute_data (label_reachable.rs:135-150) = transform_fst (:172-248) + find_intervals (:250-273)
void LabelReachData::compute(uint32_t n_states, const uint32_t* offsets, const wfst_tr* arcs, const float* finals,
                             bool reach_input_) {
  const bool timing = std::getenv("WFST_HOST_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto t_begin = now();
  reach_input = reach_input_;
  final_label = NO_LABEL;
  label2index.clear();
  const uint32_t ins = n_states;
  // labelled arcs go to a sink state per label (created in order of first appearance), final states get an arc to the
  // NO_LABEL sink, a super-initial state points at every state nothing points at
  std::unordered_map<uint32_t, uint32_t> label2state;
  std::vector<uint32_t> sink_label;  // label of sink state ins + k
  Graph g;
  g.off.assign(1, 0);
  g.dst.reserve((size_t)offsets[ins] + ins);
  auto sink = [&](uint32_t label) {
    auto it = label2state.find(label);
    if (it != label2state.end()) return it->second;
    const uint32_t st = ins + (uint32_t)sink_label.size();
    label2state.emplace(label, st);
    sink_label.push_back(label);
    return st;
  };
  for (uint32_t s = 0; s < ins; ++s) {
    for (uint32_t k = offsets[s]; k < offsets[s + 1]; ++k) {
      const uint32_t label = reach_input ? arcs[k].ilabel : arcs[k].olabel;
      g.dst.push_back(label != 0 ? sink(label) : arcs[k].nextstate);
    }
    if (finals[s] != INF) g.dst.push_back(sink(NO_LABEL));  // final_weight Some and not zero
    g.off.push_back((uint32_t)g.dst.size());
  }
  const uint32_t ons = ins + (uint32_t)sink_label.size();
  std::vector<uint32_t> indeg(ons, 0);
  for (uint32_t t : g.dst) indeg[t]++;
  for (uint32_t s = ins; s < ons; ++s) g.off.push_back((uint32_t)g.dst.size());  // sinks have no arcs
  const uint32_t start = ons;
  for (uint32_t s = 0; s < ons; ++s)
    if (indeg[s] == 0) g.dst.push_back(s);
  g.off.push_back((uint32_t)g.dst.size());
  g.is_final.assign((size_t)ons + 1, 0);
  for (uint32_t s = ins; s < ons; ++s) g.is_final[s] = 1;
  g.start = start;

  const auto t_graph = now();
  ReachSets sets;
  state_reachable(g, sets);
  const std::vector<uint32_t>& state2index = sets.state2index;
  const auto t_reach = now();

  iv_off.assign((size_t)ins + 1, 0);
  uint64_t total = 0;
  for (uint32_t s = 0; s < ins; ++s) {
    total += sets.len[s];
    if (total > 0xFFFFFFFFull) throw Error("LabelReachable: more than 2^32 intervals");
    iv_off[s + 1] = (uint32_t)total;
  }
  static_assert(sizeof(std::pair<uint32_t, uint32_t>) == 8, "an interval is two packed words");
  iv.resize(2 * (size_t)total);
  for (uint32_t s = 0; s < ins; ++s)
    if (sets.len[s]) std::memcpy(iv.data() + 2 * (size_t)iv_off[s], sets.arena.data() + sets.off[s], (size_t)sets.len[s] * 8);
  for (uint32_t k = 0; k < sink_label.size(); ++k) {
    const uint32_t idx = state2index[ins + k];
    label2index[sink_label[k]] = idx;
    if (sink_label[k] == NO_LABEL) final_label = idx;
  }
  if (timing)
    std::fprintf(stderr, "[wfst] label reachability of %u states: graph %.1f ms, intervals %.1f ms, flatten %.1f ms\n", ins,
                 ms(t_begin, t_graph), ms(t_graph, t_reach), ms(t_reach, now()));
}

uint32_t LabelReachData::relabel(uint32_t label) {  // label_reachable.rs:52-61
  if (label == 0) return 0;
  auto it = label2index.find(label);
  if (it != label2index.end()) return it->second;
  const uint32_t v = (uint32_t)label2index.size() + 1;
  label2index.emplace(label, v);
  return v;
}

// LabelReachableData::relabel_fst (label_reachable.rs:63-93): rewrite one label column, then tr_sort on it.  Returns the
// property word after the per-arc label bookkeeping (trs_iter_mut.rs:241-302) and tr_sort (tr_sort.rs:13-62).
uint64_t LabelReachData::relabel_fst(uint32_t n_states, const uint32_t* offsets, wfst_tr* arcs, uint64_t props_in,
                                     bool relabel_input) {
  uint64_t p = props_in;
  const uint64_t keep = props::ACCEPTOR | props::NOT_ACCEPTOR | props::EPSILONS | props::NO_EPSILONS | props::I_EPSILONS |
                        props::NO_I_EPSILONS | props::O_EPSILONS | props::NO_O_EPSILONS | props::WEIGHTED | props::UNWEIGHTED;
  // labels of decoding graphs are small integers: a table in front of the map (fifty million hash lookups were a third
  // of wfst_lookahead_create on the 5M-state graph); unseen labels still get their index in order of first appearance
  constexpr uint32_t LUT_MAX = 1u << 22;
  std::vector<uint32_t> lut;
  auto relabel_fast = [&](uint32_t label) -> uint32_t {
    if (label == 0) return 0;
    if (label >= LUT_MAX) return relabel(label);
    if (label >= lut.size()) lut.resize(std::min<size_t>(LUT_MAX, std::max<size_t>(2 * (size_t)label + 2, 1024)), UNASSIGNED);
    uint32_t& v = lut[label];
    if (v == UNASSIGNED) v = relabel(label);
    return v;
  };
  for (uint64_t k = 0; k < offsets[n_states]; ++k) {
    wfst_tr& tr = arcs[k];
    const uint32_t oi = tr.ilabel, oo = tr.olabel;
    if (relabel_input)
      tr.ilabel = relabel_fast(tr.ilabel);
    else
      tr.olabel = relabel_fast(tr.olabel);
    const uint32_t ni = tr.ilabel, no = tr.olabel;
    if (oi != oo) p &= ~props::NOT_ACCEPTOR;
    if (oi == 0) {
      p &= ~props::I_EPSILONS;
      if (oo == 0) p &= ~props::EPSILONS;
    }
    if (oo == 0) p &= ~props::O_EPSILONS;
    if (ni != no) {
      p |= props::NOT_ACCEPTOR;
      p &= ~props::ACCEPTOR;
    }
    if (ni == 0) {
      p |= props::I_EPSILONS;
      p &= ~props::NO_I_EPSILONS;
      if (no == 0) {
        p |= props::EPSILONS;
        p &= ~props::NO_EPSILONS;
      }
    }
    if (no == 0) {
      p |= props::O_EPSILONS;
      p &= ~props::NO_O_EPSILONS;
    }
    p &= keep;
  }
  // tr_sort (stable) of every state's arcs on the relabelled column: states are independent, so the range is cut over a
  // few host threads; short rows (the rule) by insertion sort — std::stable_sort allocates a buffer per call
  auto sort_range = [&](uint32_t s_lo, uint32_t s_hi) {
    auto keyof = [relabel_input](const wfst_tr& x) { return relabel_input ? x.ilabel : x.olabel; };
    for (uint32_t s = s_lo; s < s_hi; ++s) {
      wfst_tr* b = arcs + offsets[s];
      wfst_tr* e = arcs + offsets[s + 1];
      if (e - b <= 48) {
        for (wfst_tr* i = b + 1; i < e; ++i) {
          const wfst_tr v = *i;
          wfst_tr* j = i;
          for (; j > b && keyof(*(j - 1)) > keyof(v); --j) *j = *(j - 1);
          *j = v;
        }
      } else {
        std::stable_sort(b, e, [&](const wfst_tr& x, const wfst_tr& y) { return keyof(x) < keyof(y); });
      }
    }
  };
  const uint64_t n_arcs = offsets[n_states];
  unsigned n_thr = n_arcs >= (1u << 20) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
  if (n_thr <= 1) {
    sort_range(0, n_states);
  } else {
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errs(n_thr);
    for (unsigned t = 0; t < n_thr; ++t) {
      const uint32_t lo = (uint32_t)((uint64_t)n_states * t / n_thr), hi = (uint32_t)((uint64_t)n_states * (t + 1) / n_thr);
      pool.emplace_back([&, lo, hi, t] {
        try {
          sort_range(lo, hi);
        } catch (...) {
          errs[t] = std::current_exception();
        }
      });
    }
    for (auto& th : pool) th.join();
    for (auto& e : errs)
      if (e) std::rethrow_exception(e);
  }
  return tr_sort_props(p, relabel_input);
}

}  // namespace wfst
