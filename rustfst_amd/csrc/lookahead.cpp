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
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <thread>
#include <unordered_map>

#include "common.h"
#include "fst_props.h"
#include "host_parallel.h"
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
    arena.reserve(gr.dst.size() + gr.dst.size() / 2 + gr.n() / 4);  // (~1.5 intervals per arc on decoding graphs; grows geometrically beyond)
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
// `known_cyclic`: the caller has already found an epsilon cycle (the parallel path gave up): the visit of the whole graph
// whose sets would be thrown away at its first back arc is skipped
void state_reachable(const Graph& g, ReachSets& out, bool known_cyclic) {
  const bool timing = std::getenv("WFST_HOST_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto t0 = now();
  if (!known_cyclic) {
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

// ---- the parallel path for operands whose epsilon structure is acyclic (decoding graphs: the usual case)
//
// What the reference's visit produces, restated as two facts that need no 5M-state depth-first search:
//   * label2index[l] = 1 + the rank of label l's sink among the sinks in the order the visit DISCOVERS them.  The visit
//     order is dfs_visit's (root = the super-initial state, whose arcs go to the states without incoming epsilon arc in
//     increasing id, every state's arcs in stored order followed by the arc to the NO_LABEL sink if it is final; then the
//     unvisited states 0, 1, 2, ...) — replayed here over the operand's own arrays, and ABANDONED as soon as every label
//     of the operand has its index (a few hundred states into a decoding graph);
//   * the set of state s = normalize(its own labels' unit intervals  U  the sets of its epsilon successors): a pure
//     function of the index map on an acyclic epsilon graph, so the states are processed in rounds — every state whose
//     epsilon successors are done — by all host threads, each appending to an arena of its own.
// A round that finishes nothing means an epsilon cycle: the caller then takes the sequential path above (condensation).
struct LabelTable {  // label -> small value, a flat table in front of a map (labels of decoding graphs are small integers)
  static constexpr uint32_t LUT_MAX = 1u << 22;
  std::vector<uint32_t> lut;
  std::unordered_map<uint32_t, uint32_t> big;
  uint32_t get(uint32_t label) const {
    if (label < lut.size()) return lut[label];
    if (label < LUT_MAX) return UNASSIGNED;
    auto it = big.find(label);
    return it == big.end() ? UNASSIGNED : it->second;
  }
};

// returns false when the epsilon structure is cyclic (nothing is written then)
bool compute_acyclic_parallel(uint32_t ins, const uint32_t* offsets, const wfst_tr* arcs, const float* finals, bool reach_input,
                              LabelReachData& out, bool timing) {
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto t0 = now();
  const uint64_t n_arcs = offsets[ins];
  const unsigned n_thr = host_threads(n_arcs);
  auto label_of = [reach_input](const wfst_tr& a) { return reach_input ? a.ilabel : a.olabel; };

  // ---- pass 1: epsilon out-degrees, the largest label, which labels occur
  std::vector<uint32_t> eoff((size_t)ins + 1, 0);
  uint32_t max_small = 0;
  {
    std::vector<uint32_t> tmax(n_thr, 0);
    parallel_chunks(n_thr, ins, 1u << 14, [&](unsigned t, uint64_t b, uint64_t e) {
      uint32_t m = tmax[t];
      for (uint64_t s = b; s < e; ++s) {
        uint32_t ne = 0;
        for (uint32_t k = offsets[s]; k < offsets[s + 1]; ++k) {
          const uint32_t l = label_of(arcs[k]);
          ne += l == 0;
          if (l < LabelTable::LUT_MAX && l > m) m = l;
        }
        eoff[s + 1] = ne;
      }
      tmax[t] = m;
    });
    for (uint32_t m : tmax) max_small = std::max(max_small, m);
  }
  for (uint32_t s = 0; s < ins; ++s) eoff[s + 1] += eoff[s];
  const uint32_t n_eps = eoff[ins];
  // ---- pass 2: the epsilon successors (CSR), the labels seen
  std::vector<uint32_t> edst(n_eps);
  std::vector<uint8_t> seen((size_t)max_small + 1, 0);
  std::vector<std::vector<uint32_t>> big_seen(n_thr);
  std::vector<uint8_t> any_final(n_thr, 0);
  parallel_chunks(n_thr, ins, 1u << 14, [&](unsigned t, uint64_t b, uint64_t e) {
    for (uint64_t s = b; s < e; ++s) {
      uint32_t w = eoff[s];
      for (uint32_t k = offsets[s]; k < offsets[s + 1]; ++k) {
        const uint32_t l = label_of(arcs[k]);
        if (l == 0) edst[w++] = arcs[k].nextstate;
        else if (l < LabelTable::LUT_MAX) {
          if (!seen[l]) seen[l] = 1;  // (benign race: every writer stores 1; tested first, or the line ping-pongs)
        }
        else big_seen[t].push_back(l);
      }
      if (finals[s] != INF && !any_final[t]) any_final[t] = 1;
    }
  });
  uint32_t n_labels = 0;  // sinks the visit will discover
  for (uint32_t l = 1; l <= max_small; ++l) n_labels += seen[l];
  LabelTable idx;
  idx.lut.assign((size_t)max_small + 1, UNASSIGNED);
  for (auto& v : big_seen)
    for (uint32_t l : v)
      if (idx.big.emplace(l, UNASSIGNED).second) n_labels++;
  bool has_final = false;
  for (uint8_t f : any_final) has_final |= f != 0;
  if (has_final && idx.big.emplace(NO_LABEL, UNASSIGNED).second) n_labels++;
  const auto t1 = now();

  // ---- the sets, in rounds over the epsilon dependencies (needs no label indices yet: the sets are built from LABELS
  //      first?  no — intervals are over indices, so the index map comes first)
  // ---- the index map: dfs_visit's order, abandoned once every label has been met
  {
    std::vector<uint8_t> has_in((size_t)ins, 0);
    for (uint32_t k = 0; k < n_eps; ++k) has_in[edst[k]] = 1;
    enum : uint8_t { White, Grey, Black };
    std::vector<uint8_t> color((size_t)ins, White);
    std::vector<std::pair<uint32_t, uint32_t>> stack;  // {state, next arc position; == offsets[s + 1] : the final arc}
    uint32_t found = 0, index = 1;
    auto meet = [&](uint32_t label) {
      uint32_t* slot = label < LabelTable::LUT_MAX ? &idx.lut[label] : &idx.big[label];
      if (*slot == UNASSIGNED) {
        *slot = index++;
        found++;
      }
    };
    auto visit_from = [&](uint32_t root) {
      color[root] = Grey;
      stack.push_back({root, offsets[root]});
      while (!stack.empty() && found < n_labels) {
        const uint32_t s = stack.back().first;
        uint32_t& pos = stack.back().second;
        if (pos < offsets[s + 1]) {
          const wfst_tr& a = arcs[pos++];
          const uint32_t l = label_of(a);
          if (l != 0) {
            meet(l);
          } else if (color[a.nextstate] == White) {
            color[a.nextstate] = Grey;
            stack.push_back({a.nextstate, offsets[a.nextstate]});
          }
          continue;
        }
        if (pos == offsets[s + 1] && finals[s] != INF) {
          pos++;
          meet(NO_LABEL);
          continue;
        }
        color[s] = Black;
        stack.pop_back();
      }
      stack.clear();
    };
    // the super-initial state's arcs: states nothing points at, in increasing id; then dfs_visit's remaining roots
    for (uint32_t s = 0; s < ins && found < n_labels; ++s)
      if (!has_in[s] && color[s] == White) visit_from(s);
    for (uint32_t s = 0; s < ins && found < n_labels; ++s)
      if (color[s] == White) visit_from(s);
  }
  const auto t2 = now();

  // ---- the sets, in rounds: a state is ready when its epsilon successors are done
  // (a thread's finished sets are read by the others in later rounds while it appends new ones: the arenas are lists of
  // blocks that never move, and a set lies inside one block)
  using Iv = std::pair<uint32_t, uint32_t>;
  struct Rec {
    const Iv* p;
    uint32_t len;
  };
  struct BlockArena {
    std::vector<Iv*> blocks;  // (raw storage: `new Iv[]` would zero every block first)
    size_t used = 0, cap = 0;
    BlockArena() = default;
    BlockArena(const BlockArena&) = delete;
    BlockArena(BlockArena&& o) noexcept : blocks(std::move(o.blocks)), used(o.used), cap(o.cap) { o.blocks.clear(); }
    ~BlockArena() {
      for (Iv* b : blocks) std::free(b);
    }
    Iv* alloc(size_t n) {
      if (used + n > cap) {
        cap = std::max<size_t>(n, (size_t)1 << 20);
        Iv* b = (Iv*)std::malloc(cap * sizeof(Iv));
        if (!b) throw Error("out of memory");
        blocks.push_back(b);
        used = 0;
      }
      Iv* r = blocks.back() + used;
      used += n;
      return r;
    }
  };
  std::vector<Rec> rec((size_t)ins);
  std::vector<uint8_t> done((size_t)ins, 0);
  std::vector<BlockArena> arenas(n_thr);
  std::vector<uint32_t> todo, next_todo;
  std::vector<std::vector<uint32_t>> later(n_thr);
  std::vector<Intervals> scratch(n_thr);
  const uint32_t final_idx = has_final ? idx.get(NO_LABEL) : UNASSIGNED;
  auto build = [&](unsigned t, uint32_t s) {
    Intervals& sc = scratch[t];
    sc.clear();
    for (uint32_t k = offsets[s]; k < offsets[s + 1]; ++k) {
      const uint32_t l = label_of(arcs[k]);
      if (l != 0) {
        const uint32_t i = idx.get(l);
        sc.push_back({i, i + 1});
      }
    }
    if (finals[s] != INF) sc.push_back({final_idx, final_idx + 1});
    for (uint32_t k = eoff[s]; k < eoff[s + 1]; ++k) {
      const Rec& r = rec[edst[k]];
      sc.insert(sc.end(), r.p, r.p + r.len);
    }
    normalize(sc);
    Iv* dst = sc.empty() ? nullptr : arenas[t].alloc(sc.size());
    if (!sc.empty()) std::memcpy((void*)dst, sc.data(), sc.size() * sizeof(Iv));
    rec[s] = Rec{dst, (uint32_t)sc.size()};
  };
  // round 0 over all states (no list yet); later rounds over the states that were not ready
  {
    parallel_chunks(n_thr, ins, 1u << 12, [&](unsigned t, uint64_t b, uint64_t e) {
      for (uint64_t s = b; s < e; ++s) {
        if (eoff[s] == eoff[s + 1]) build(t, (uint32_t)s);
        else later[t].push_back((uint32_t)s);
      }
    });
    // (a state finished in this round becomes visible to the others only with the next one: `done` is written after the join)
    for (unsigned t = 0; t < n_thr; ++t) {
      todo.insert(todo.end(), later[t].begin(), later[t].end());
      later[t].clear();
    }
    parallel_chunks(n_thr, ins, 1u << 16, [&](unsigned, uint64_t b, uint64_t e) {
      for (uint64_t s = b; s < e; ++s) done[s] = eoff[s] == eoff[s + 1];
    });
  }
  std::vector<std::vector<uint32_t>> finished(n_thr);
  // every round rescans the whole list: depth x |todo| steps.  A long epsilon chain (a million states in a line) would make
  // that quadratic where the sequential visit is linear: past a few dozen rounds with most of the list still waiting, the
  // sequential path takes over.
  for (uint32_t round = 0; !todo.empty(); ++round) {
    if (round >= 64 && todo.size() > 4096) return false;
    parallel_chunks(n_thr, todo.size(), 1u << 10, [&](unsigned t, uint64_t b, uint64_t e) {
      for (uint64_t i = b; i < e; ++i) {
        const uint32_t s = todo[i];
        bool ready = true;
        for (uint32_t k = eoff[s]; k < eoff[s + 1] && ready; ++k) ready = done[edst[k]] != 0;
        if (ready) {
          build(t, s);
          finished[t].push_back(s);
        } else {
          later[t].push_back(s);
        }
      }
    });
    next_todo.clear();
    uint64_t fin = 0;
    for (unsigned t = 0; t < n_thr; ++t) {
      for (uint32_t s : finished[t]) done[s] = 1;
      fin += finished[t].size();
      finished[t].clear();
      next_todo.insert(next_todo.end(), later[t].begin(), later[t].end());
      later[t].clear();
    }
    if (fin == 0) return false;  // an epsilon cycle: the sequential path (condensation) takes over
    todo.swap(next_todo);
  }
  const auto t3 = now();

  // ---- flatten to the per-state CSR the kernels read
  out.iv_off.assign((size_t)ins + 1, 0);
  uint64_t total = 0;
  for (uint32_t s = 0; s < ins; ++s) {
    total += rec[s].len;
    if (total > 0xFFFFFFFFull) throw Error("LabelReachable: more than 2^32 intervals");
    out.iv_off[s + 1] = (uint32_t)total;
  }
  static_assert(sizeof(std::pair<uint32_t, uint32_t>) == 8, "an interval is two packed words");
  out.iv.resize(2 * (size_t)total);
  parallel_chunks(n_thr, ins, 1u << 14, [&](unsigned, uint64_t b, uint64_t e) {
    for (uint64_t s = b; s < e; ++s)
      if (rec[s].len)
        std::memcpy(out.iv.data() + 2 * (size_t)out.iv_off[s], rec[s].p, (size_t)rec[s].len * 8);
  });
  out.label2index.clear();
  out.final_label = NO_LABEL;
  for (uint32_t l = 1; l <= max_small; ++l)
    if (seen[l]) out.label2index[l] = idx.lut[l];
  for (auto& kv : idx.big) out.label2index[kv.first] = kv.second;
  if (has_final) out.final_label = final_idx;
  if (timing)
    std::fprintf(stderr, "[wfst] label reachability of %u states on %u host threads: scan %.1f ms, index map %.1f ms, sets %.1f ms, "
                         "flatten %.1f ms\n", ins, n_thr, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
  return true;
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
  // acyclic epsilon structure (decoding graphs): all host threads, no materialised transformed graph
  bool known_cyclic = false;
  if (!std::getenv("WFST_LOOKAHEAD_SEQUENTIAL") && ins > 0) {
    if (compute_acyclic_parallel(ins, offsets, arcs, finals, reach_input, *this, timing)) return;
    known_cyclic = true;
  }
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
  state_reachable(g, sets, known_cyclic);
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
  // The reference rewrites arc after arc (trs_iter_mut.rs:241-302): every arc clears and sets property bits from its old
  // and new labels.  Each arc's effect is p -> (p & ~C) | S, and such maps compose associatively, so the arcs are cut
  // into ranges whose (C, S) are worked out by the host threads and applied in order — bit-identical to the sequential
  // loop.  Labels the data has not seen yet get their index in order of first appearance (label_reachable.rs:52-61),
  // which IS sequential: an operand that brings such labels (checked first, in parallel) takes the one-thread loop.
  struct Cs {
    uint64_t c = 0, s = 0;
    void clear(uint64_t x) {
      c |= x;
      s &= ~x;
    }
    void set(uint64_t x) { s |= x; }
  };
  auto arc_effect = [&](Cs& e, uint32_t oi, uint32_t oo, uint32_t ni, uint32_t no) {
    if (oi != oo) e.clear(props::NOT_ACCEPTOR);
    if (oi == 0) {
      e.clear(props::I_EPSILONS);
      if (oo == 0) e.clear(props::EPSILONS);
    }
    if (oo == 0) e.clear(props::O_EPSILONS);
    if (ni != no) {
      e.set(props::NOT_ACCEPTOR);
      e.clear(props::ACCEPTOR);
    }
    if (ni == 0) {
      e.set(props::I_EPSILONS);
      e.clear(props::NO_I_EPSILONS);
      if (no == 0) {
        e.set(props::EPSILONS);
        e.clear(props::NO_EPSILONS);
      }
    }
    if (no == 0) {
      e.set(props::O_EPSILONS);
      e.clear(props::NO_O_EPSILONS);
    }
    e.clear(~keep);
  };
  const uint64_t n_arcs_all = offsets[n_states];
  const unsigned rl_thr = host_threads(n_arcs_all);
  bool all_known = rl_thr > 1;
  if (rl_thr > 1) {
    // read-only view of the map for the threads (flat table in front, as in relabel_fast)
    uint32_t max_small = 0;
    for (auto& kv : label2index)
      if (kv.first < LUT_MAX && kv.first > max_small) max_small = kv.first;
    lut.assign((size_t)max_small + 1, UNASSIGNED);
    for (auto& kv : label2index)
      if (kv.first < LUT_MAX) lut[kv.first] = kv.second;
    std::atomic<int> unknown{0};
    parallel_chunks(rl_thr, n_arcs_all, 1u << 18, [&](unsigned, uint64_t b, uint64_t e) {
      if (unknown.load(std::memory_order_relaxed)) return;
      for (uint64_t k = b; k < e; ++k) {
        const uint32_t l = relabel_input ? arcs[k].ilabel : arcs[k].olabel;
        if (l == 0) continue;
        const bool known = l < lut.size() ? lut[l] != UNASSIGNED : (l >= LUT_MAX && label2index.count(l) != 0);
        if (!known) {
          unknown.store(1, std::memory_order_relaxed);
          return;
        }
      }
    });
    all_known = unknown.load() == 0;
  }
  if (all_known) {
    const uint64_t per = (n_arcs_all + rl_thr - 1) / rl_thr;
    std::vector<Cs> eff(rl_thr);
    parallel_chunks(rl_thr, rl_thr, 1, [&](unsigned, uint64_t tb, uint64_t te) {
      for (uint64_t part = tb; part < te; ++part) {
        Cs e;
        const uint64_t b = part * per, en = std::min(n_arcs_all, b + per);
        for (uint64_t k = b; k < en; ++k) {
          wfst_tr& tr = arcs[k];
          const uint32_t oi = tr.ilabel, oo = tr.olabel;
          uint32_t& col = relabel_input ? tr.ilabel : tr.olabel;
          if (col != 0) col = col < lut.size() ? lut[col] : label2index.find(col)->second;
          arc_effect(e, oi, oo, tr.ilabel, tr.olabel);
        }
        eff[part] = e;
      }
    });
    for (const Cs& e : eff) p = (p & ~e.c) | e.s;
  } else {
    lut.clear();
    for (uint64_t k = 0; k < n_arcs_all; ++k) {
      wfst_tr& tr = arcs[k];
      const uint32_t oi = tr.ilabel, oo = tr.olabel;
      if (relabel_input)
        tr.ilabel = relabel_fast(tr.ilabel);
      else
        tr.olabel = relabel_fast(tr.olabel);
      Cs e;
      arc_effect(e, oi, oo, tr.ilabel, tr.olabel);
      p = (p & ~e.c) | e.s;
    }
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
  unsigned n_thr = n_arcs >= (1u << 20) || std::getenv("WFST_HOST_THREADS") ? host_threads(n_arcs) : 1u;
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
