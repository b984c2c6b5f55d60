// oracle.cpp — CPU ORACLE for compose -> shortest_path (test infrastructure, NOT product code).
//
// A single-threaded C++17 restatement of rustfst 1.3.1 (reference @ /root/reference) for
// VectorFst<TropicalWeight>.  Every block cites the reference file:line it follows.  It keeps
// the reference's data-structure choices (per-state arc vectors, hash-map state table keyed on
// (fs,s1,s2), FIFO BFS materialisation, DFS connect, AutoQueue with distance=None, approximate
// TropicalWeight ==) so it can also stand in as the CPU baseline ("port", not rustfst binaries).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load this library.
// PARITY PIN STATUS: see oracle.h.
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// ---------------------------------------------------------------- constants (lib.rs:236,269,292,298)
constexpr uint32_t EPS_LABEL = 0;
constexpr uint32_t NO_LABEL = 0xFFFFFFFFu;
constexpr uint32_t NO_STATE_ID = 0xFFFFFFFFu;
constexpr float KDELTA = 1.0f / 1024.0f;
constexpr float INF = std::numeric_limits<float>::infinity();

thread_local std::string t_err;
thread_local float t_delta = KDELTA;  // ORACLE_EQ_REF_KDELTA -> KDELTA, ORACLE_EQ_EXACT -> 0
thread_local const char* t_queue_kind = "";
// look-ahead composition bookkeeping for tools/lookahead_tuple_gap.py: tuples created, and how many of them had an
// already-present tuple equal in everything but a pushed weight exactly one KDELTA step away
thread_local uint64_t t_la_tuples = 0, t_la_adjacent = 0;

struct DeltaGuard {
  float saved;
  explicit DeltaGuard(int eq_mode) : saved(t_delta) { t_delta = eq_mode == ORACLE_EQ_EXACT ? 0.0f : KDELTA; }
  ~DeltaGuard() { t_delta = saved; }
};

// ---------------------------------------------------------------- TropicalWeight (T1)
// semirings/tropical_weight.rs:53-71 and the eq macro semirings/semiring.rs:159-168
inline bool weq(float a, float b) { return a <= b + t_delta && b <= a + t_delta; }
inline float wplus(float a, float b) { return b < a ? b : a; }  // plus_assign :53-58 (exact <)
inline float wtimes(float a, float b) {                          // times_assign :60-70
  if (a == INF) return a;
  if (b == INF) return b;
  return a + b;
}
inline bool wis_zero(float a) { return weq(a, INF); }  // semiring.rs:71-73
// quantize — semirings/semiring.rs:132-145
inline float quantize(float v, float delta) {
  if (std::isinf(v)) return v;
  return std::floor((v / delta) + 0.5f) * delta;
}
inline bool wis_one(float a) { return weq(a, 0.0f); }  // semiring.rs:68-70

// ---------------------------------------------------------------- FstProperties (properties.rs:22-103)
namespace P {
constexpr uint64_t ACCEPTOR = 0x0000000000010000ull, NOT_ACCEPTOR = 0x0000000000020000ull;
constexpr uint64_t I_DETERMINISTIC = 0x0000000000040000ull, NOT_I_DETERMINISTIC = 0x0000000000080000ull;
constexpr uint64_t O_DETERMINISTIC = 0x0000000000100000ull, NOT_O_DETERMINISTIC = 0x0000000000200000ull;
constexpr uint64_t EPSILONS = 0x0000000000400000ull, NO_EPSILONS = 0x0000000000800000ull;
constexpr uint64_t I_EPSILONS = 0x0000000001000000ull, NO_I_EPSILONS = 0x0000000002000000ull;
constexpr uint64_t O_EPSILONS = 0x0000000004000000ull, NO_O_EPSILONS = 0x0000000008000000ull;
constexpr uint64_t I_LABEL_SORTED = 0x0000000010000000ull, NOT_I_LABEL_SORTED = 0x0000000020000000ull;
constexpr uint64_t O_LABEL_SORTED = 0x0000000040000000ull, NOT_O_LABEL_SORTED = 0x0000000080000000ull;
constexpr uint64_t WEIGHTED = 0x0000000100000000ull, UNWEIGHTED = 0x0000000200000000ull;
constexpr uint64_t CYCLIC = 0x0000000400000000ull, ACYCLIC = 0x0000000800000000ull;
constexpr uint64_t INITIAL_CYCLIC = 0x0000001000000000ull, INITIAL_ACYCLIC = 0x0000002000000000ull;
constexpr uint64_t TOP_SORTED = 0x0000004000000000ull, NOT_TOP_SORTED = 0x0000008000000000ull;
constexpr uint64_t ACCESSIBLE = 0x0000010000000000ull, NOT_ACCESSIBLE = 0x0000020000000000ull;
constexpr uint64_t COACCESSIBLE = 0x0000040000000000ull, NOT_COACCESSIBLE = 0x0000080000000000ull;
constexpr uint64_t STRING = 0x0000100000000000ull, NOT_STRING = 0x0000200000000000ull;
constexpr uint64_t WEIGHTED_CYCLES = 0x0000400000000000ull, UNWEIGHTED_CYCLES = 0x0000800000000000ull;
constexpr uint64_t ALL = 0x0000ffffffff0000ull;  // all_properties(): binary bits are not flags (:520-534)
constexpr uint64_t STATIC_EXPANDED_MUTABLE = 0x3;  // properties.rs:5-6

// properties.rs:105-124
constexpr uint64_t NULL_PROPS = ACCEPTOR | I_DETERMINISTIC | O_DETERMINISTIC | NO_EPSILONS | NO_I_EPSILONS |
                                NO_O_EPSILONS | I_LABEL_SORTED | O_LABEL_SORTED | UNWEIGHTED | ACYCLIC |
                                INITIAL_ACYCLIC | TOP_SORTED | ACCESSIBLE | COACCESSIBLE | STRING |
                                UNWEIGHTED_CYCLES;
// properties.rs:166-194
constexpr uint64_t SET_START_MASK = ACCEPTOR | NOT_ACCEPTOR | I_DETERMINISTIC | NOT_I_DETERMINISTIC |
                                    O_DETERMINISTIC | NOT_O_DETERMINISTIC | EPSILONS | NO_EPSILONS | I_EPSILONS |
                                    NO_I_EPSILONS | O_EPSILONS | NO_O_EPSILONS | I_LABEL_SORTED |
                                    NOT_I_LABEL_SORTED | O_LABEL_SORTED | NOT_O_LABEL_SORTED | WEIGHTED |
                                    UNWEIGHTED | CYCLIC | ACYCLIC | TOP_SORTED | NOT_TOP_SORTED | COACCESSIBLE |
                                    NOT_COACCESSIBLE | WEIGHTED_CYCLES | UNWEIGHTED_CYCLES;
// properties.rs:197-225
constexpr uint64_t SET_FINAL_MASK = ACCEPTOR | NOT_ACCEPTOR | I_DETERMINISTIC | NOT_I_DETERMINISTIC |
                                    O_DETERMINISTIC | NOT_O_DETERMINISTIC | EPSILONS | NO_EPSILONS | I_EPSILONS |
                                    NO_I_EPSILONS | O_EPSILONS | NO_O_EPSILONS | I_LABEL_SORTED |
                                    NOT_I_LABEL_SORTED | O_LABEL_SORTED | NOT_O_LABEL_SORTED | CYCLIC | ACYCLIC |
                                    INITIAL_CYCLIC | INITIAL_ACYCLIC | TOP_SORTED | NOT_TOP_SORTED | ACCESSIBLE |
                                    NOT_ACCESSIBLE | WEIGHTED_CYCLES | UNWEIGHTED_CYCLES;
// properties.rs:228-259
constexpr uint64_t ADD_STATE_MASK = ACCEPTOR | NOT_ACCEPTOR | I_DETERMINISTIC | NOT_I_DETERMINISTIC |
                                    O_DETERMINISTIC | NOT_O_DETERMINISTIC | EPSILONS | NO_EPSILONS | I_EPSILONS |
                                    NO_I_EPSILONS | O_EPSILONS | NO_O_EPSILONS | I_LABEL_SORTED |
                                    NOT_I_LABEL_SORTED | O_LABEL_SORTED | NOT_O_LABEL_SORTED | WEIGHTED |
                                    UNWEIGHTED | CYCLIC | ACYCLIC | INITIAL_CYCLIC | INITIAL_ACYCLIC | TOP_SORTED |
                                    NOT_TOP_SORTED | NOT_ACCESSIBLE | NOT_COACCESSIBLE | NOT_STRING |
                                    WEIGHTED_CYCLES | UNWEIGHTED_CYCLES;
// properties.rs:262-278
constexpr uint64_t ADD_ARC_MASK = NOT_ACCEPTOR | NOT_I_DETERMINISTIC | NOT_O_DETERMINISTIC | EPSILONS | I_EPSILONS |
                                  O_EPSILONS | NOT_I_LABEL_SORTED | NOT_O_LABEL_SORTED | WEIGHTED | CYCLIC |
                                  INITIAL_CYCLIC | NOT_TOP_SORTED | ACCESSIBLE | COACCESSIBLE | WEIGHTED_CYCLES;
// properties.rs:286-300
constexpr uint64_t DELETE_STATES_MASK = ACCEPTOR | I_DETERMINISTIC | O_DETERMINISTIC | NO_EPSILONS | NO_I_EPSILONS |
                                        NO_O_EPSILONS | I_LABEL_SORTED | O_LABEL_SORTED | UNWEIGHTED | ACYCLIC |
                                        INITIAL_ACYCLIC | TOP_SORTED | UNWEIGHTED_CYCLES;

// mutate_properties.rs:7-13
inline uint64_t set_start_properties(uint64_t in) {
  uint64_t out = in & SET_START_MASK;
  if (in & ACYCLIC) out |= INITIAL_ACYCLIC;
  return out;
}
// mutate_properties.rs:15-37
inline uint64_t set_final_properties(uint64_t in, const float* old_w, const float* new_w) {
  uint64_t out = in;
  if (old_w && !wis_zero(*old_w) && !wis_one(*old_w)) out &= ~WEIGHTED;
  if (new_w && !wis_zero(*new_w) && !wis_one(*new_w)) {
    out |= WEIGHTED;
    out &= ~UNWEIGHTED;
  }
  out &= SET_FINAL_MASK | WEIGHTED | UNWEIGHTED;
  return out;
}
inline uint64_t add_state_properties(uint64_t in) { return in & ADD_STATE_MASK; }  // :39-41
// mutate_properties.rs:43-100
inline uint64_t add_tr_properties(uint64_t in, uint32_t state, const oracle_tr& tr, const oracle_tr* prev) {
  uint64_t out = in;
  if (tr.ilabel != tr.olabel) {
    out |= NOT_ACCEPTOR;
    out &= ~ACCEPTOR;
  }
  if (tr.ilabel == EPS_LABEL) {
    out |= I_EPSILONS;
    out &= ~NO_I_EPSILONS;
    if (tr.olabel == EPS_LABEL) {
      out |= EPSILONS;
      out &= ~NO_EPSILONS;
    }
  }
  if (tr.olabel == EPS_LABEL) {
    out |= O_EPSILONS;
    out &= ~NO_O_EPSILONS;
  }
  if (prev) {
    if (prev->ilabel > tr.ilabel) {
      out |= NOT_I_LABEL_SORTED;
      out &= ~I_LABEL_SORTED;
    }
    if (prev->olabel > tr.olabel) {
      out |= NOT_O_LABEL_SORTED;
      out &= ~O_LABEL_SORTED;
    }
  }
  if (!wis_zero(tr.weight) && !wis_one(tr.weight)) {
    out |= WEIGHTED;
    out &= ~UNWEIGHTED;
  }
  if (tr.nextstate <= state) {
    out |= NOT_TOP_SORTED;
    out &= ~TOP_SORTED;
  }
  out &= ADD_ARC_MASK | ACCEPTOR | NO_EPSILONS | NO_I_EPSILONS | NO_O_EPSILONS | I_LABEL_SORTED | O_LABEL_SORTED |
         UNWEIGHTED | TOP_SORTED;
  if (out & TOP_SORTED) out |= ACYCLIC | INITIAL_ACYCLIC;
  return out;
}
inline uint64_t delete_states_properties(uint64_t in) { return in & DELETE_STATES_MASK; }  // :102-104
// delete_arcs_properties(): properties.rs:300-316 (delete_trs_properties, mutate_properties.rs:110-112)
constexpr uint64_t DELETE_ARCS_MASK = ACCEPTOR | I_DETERMINISTIC | O_DETERMINISTIC | NO_EPSILONS | NO_I_EPSILONS | NO_O_EPSILONS |
                                      I_LABEL_SORTED | O_LABEL_SORTED | UNWEIGHTED | ACYCLIC | INITIAL_ACYCLIC | TOP_SORTED |
                                      NOT_ACCESSIBLE | NOT_COACCESSIBLE | UNWEIGHTED_CYCLES;
// mutate_properties.rs:151-184
inline uint64_t compose_properties(uint64_t p1, uint64_t p2) {
  uint64_t out = 0;
  if ((p1 & ACCEPTOR) && (p2 & ACCEPTOR)) {
    out |= ACCEPTOR | ACCESSIBLE;
    out |= (NO_EPSILONS | NO_I_EPSILONS | NO_O_EPSILONS | ACYCLIC | INITIAL_ACYCLIC) & p1 & p2;
    if ((p1 & NO_I_EPSILONS) && (p2 & NO_I_EPSILONS)) out |= (I_DETERMINISTIC | O_DETERMINISTIC) & p1 & p2;
  } else {
    out |= ACCESSIBLE;
    out |= (ACCEPTOR | NO_I_EPSILONS | ACYCLIC | INITIAL_ACYCLIC) & p1 & p2;
    if ((p1 & NO_I_EPSILONS) && (p2 & NO_I_EPSILONS)) out |= I_DETERMINISTIC & p1 & p2;
  }
  return out;
}
// mutate_properties.rs:662-672
inline uint64_t shortest_path_properties(uint64_t props, bool tree) {
  uint64_t out = props | ACYCLIC | INITIAL_ACYCLIC | ACCESSIBLE | UNWEIGHTED_CYCLES;
  if (!tree) out |= COACCESSIBLE;
  return out;
}
}  // namespace P

using Tr = oracle_tr;

// ---------------------------------------------------------------- VectorFst (T3)
// fst_impls/vector_fst/data_structure.rs:16-34
struct State {
  bool has_final = false;  // Option<W>
  float final_w = INF;
  std::vector<Tr> trs;  // TrsVec(Arc<Vec<Tr>>)
  size_t niepsilons = 0, noepsilons = 0;
};

}  // namespace

struct oracle_fst {
  std::vector<State> states;
  bool has_start = false;
  uint32_t start = 0;
  uint64_t properties = P::NULL_PROPS;  // VectorFst::new(): mutable_fst.rs:25-33

  size_t num_states() const { return states.size(); }
  // mutable_fst.rs:35-44
  bool set_start(uint32_t s) {
    if (s >= states.size()) {
      t_err = "The state " + std::to_string(s) + " doesn't exist";
      return false;
    }
    has_start = true;
    start = s;
    properties = P::set_start_properties(properties);
    return true;
  }
  // mutable_fst.rs:52-65
  bool set_final(uint32_t s, float w) {
    if (s >= states.size()) {
      t_err = "Stateid " + std::to_string(s) + " doesn't exist";
      return false;
    }
    State& st = states[s];
    properties = P::set_final_properties(properties, st.has_final ? &st.final_w : nullptr, &w);
    st.has_final = true;
    st.final_w = w;
    return true;
  }
  // mutable_fst.rs:82-87
  uint32_t add_state() {
    states.emplace_back();
    properties = P::add_state_properties(properties);
    return (uint32_t)(states.size() - 1);
  }
  // mutable_fst.rs:89-93
  void add_states(size_t n) {
    states.resize(states.size() + n);
    properties = P::add_state_properties(properties);
  }
  // mutable_fst.rs:235-244 + data_structure.rs:76-92
  bool add_tr(uint32_t source, const Tr& tr) {
    if (source >= states.size()) {
      t_err = "State " + std::to_string(source) + " doesn't exist";
      return false;
    }
    State& st = states[source];
    if (tr.ilabel == EPS_LABEL) st.niepsilons++;
    if (tr.olabel == EPS_LABEL) st.noepsilons++;
    st.trs.push_back(tr);
    const Tr* prev = st.trs.size() > 1 ? &st.trs[st.trs.size() - 2] : nullptr;
    properties = P::add_tr_properties(properties, source, st.trs.back(), prev);
    return true;
  }
  // mutable_fst.rs:255-281
  void set_trs_unchecked(uint32_t source, std::vector<Tr> trs) {
    uint64_t props = properties;
    State& st = states[source];
    st.trs = std::move(trs);
    size_t ni = 0, no = 0;
    for (size_t i = 0; i < st.trs.size(); ++i) {
      props = P::add_tr_properties(props, source, st.trs[i], i >= 1 ? &st.trs[i - 1] : nullptr);
      if (st.trs[i].ilabel == EPS_LABEL) ni++;
      if (st.trs[i].olabel == EPS_LABEL) no++;
    }
    st.niepsilons = ni;
    st.noepsilons = no;
    properties = props;
  }
  // mutable_fst.rs:132-189 (stable compaction; arcs into deleted states dropped in order)
  void del_states(const std::vector<uint32_t>& dstates) {
    std::vector<int32_t> new_id(states.size(), 0);
    for (uint32_t s : dstates) new_id[s] = -1;
    size_t nstates = 0;
    for (size_t s = 0; s < states.size(); ++s) {
      if (new_id[s] != -1) {
        new_id[s] = (int32_t)nstates;
        if (s != nstates) std::swap(states[nstates], states[s]);
        nstates++;
      }
    }
    states.resize(nstates);
    for (size_t s = 0; s < states.size(); ++s) {
      State& st = states[s];
      size_t w = 0;
      for (size_t i = 0; i < st.trs.size(); ++i) {
        int32_t t = new_id[st.trs[i].nextstate];
        if (t != -1) {
          st.trs[i].nextstate = (uint32_t)t;
          if (w != i) st.trs[w] = st.trs[i];
          w++;
        } else {
          if (st.trs[i].ilabel == EPS_LABEL) st.niepsilons--;
          if (st.trs[i].olabel == EPS_LABEL) st.noepsilons--;
        }
      }
      st.trs.resize(w);
    }
    if (has_start) {
      int32_t ns = new_id[start];
      if (ns == -1)
        has_start = false;
      else
        start = (uint32_t)ns;
    }
    properties = P::delete_states_properties(properties);
  }
  void set_properties(uint64_t p) { properties = p; }  // mutable_fst.rs:407-409
  void set_properties_with_mask(uint64_t p, uint64_t mask) {  // :411-414
    properties &= ~mask;
    properties |= p & mask;
  }
};

extern "C" void oracle_fst_tr_sort(oracle_fst* f, int by_olabel);

namespace {
using Fst = oracle_fst;

// ---------------------------------------------------------------- compose (A1-A9)
enum class MatchType { MatchInput, MatchOutput, MatchBoth, MatchNone, MatchUnknown };  // matchers/mod.rs:57-69

// SortedMatcher::match_type(test) — matchers/sorted_matcher.rs:56-85.
// Returns false + t_err when properties_check fails (fst_traits/fst.rs:166-176).
bool sorted_matcher_match_type(const Fst& fst, MatchType mt, bool test, MatchType* out) {
  const uint64_t true_prop = mt == MatchType::MatchInput ? P::I_LABEL_SORTED : P::O_LABEL_SORTED;
  const uint64_t false_prop = mt == MatchType::MatchInput ? P::NOT_I_LABEL_SORTED : P::NOT_O_LABEL_SORTED;
  const uint64_t props = fst.properties;
  if (test) {
    // FstProperties::knows: for each trinary pair in the mask at least one bit must be set
    if (!(props & (true_prop | false_prop))) {
      t_err = "Properties are not known";
      return false;
    }
  }
  if (props & true_prop)
    *out = mt;
  else if (props & false_prop)
    *out = MatchType::MatchNone;
  else
    *out = MatchType::MatchUnknown;
  return true;
}

// ComposeFstOp::match_type — compose/compose_fst_op.rs:169-197 (SortedMatcher flags are empty)
bool compose_match_type(const Fst& f1, const Fst& f2, MatchType* out) {
  MatchType type1, type2, t;
  sorted_matcher_match_type(f1, MatchType::MatchOutput, false, &type1);
  sorted_matcher_match_type(f2, MatchType::MatchInput, false, &type2);
  if (type1 == MatchType::MatchOutput && type2 == MatchType::MatchInput) {
    *out = MatchType::MatchBoth;
  } else if (type1 == MatchType::MatchOutput) {
    *out = MatchType::MatchOutput;
  } else if (type2 == MatchType::MatchInput) {
    *out = MatchType::MatchInput;
  } else {
    if (!sorted_matcher_match_type(f1, MatchType::MatchOutput, true, &t)) return false;
    if (t == MatchType::MatchOutput) {
      *out = MatchType::MatchOutput;
      return true;
    }
    if (!sorted_matcher_match_type(f2, MatchType::MatchInput, true, &t)) return false;
    if (t == MatchType::MatchInput) {
      *out = MatchType::MatchInput;
      return true;
    }
    t_err =
        "ComposeFst: 1st argument cannot match on output labels and 2nd argument cannot match on input labels "
        "(sort?).";
    return false;
  }
  return true;
}

// ComposeStateTuple — compose/compose_state_tuple.rs:10-15
struct Tuple {
  uint32_t fs, s1, s2;
  bool operator==(const Tuple& o) const { return fs == o.fs && s1 == o.s1 && s2 == o.s2; }
};
struct TupleHash {
  size_t operator()(const Tuple& t) const {
    uint64_t h = ((uint64_t)t.s1 << 32) | t.s2;
    h ^= (uint64_t)t.fs * 0x9E3779B97F4A7C15ull;
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return (size_t)h;
  }
};
// StateTable / BiHashMap — lazy/state_table.rs:20-64,102-125 (ids in first-lookup order)
struct StateTable {
  std::unordered_map<Tuple, uint32_t, TupleHash> tuple_to_id;
  std::vector<Tuple> id_to_tuple;
  uint32_t find_id(const Tuple& t) {
    auto it = tuple_to_id.find(t);
    if (it != tuple_to_id.end()) return it->second;
    uint32_t n = (uint32_t)id_to_tuple.size();
    id_to_tuple.push_back(t);
    tuple_to_id.emplace(t, n);
    return n;
  }
  Tuple find_tuple(uint32_t id) const { return id_to_tuple[id]; }
};

// IteratorSortedMatcher — matchers/sorted_matcher.rs:124-184
struct MatcherIter {
  const std::vector<Tr>* trs;
  uint32_t match_label;
  size_t pos;
  bool current_loop;
  bool by_ilabel;
  MatcherIter(const std::vector<Tr>& t, uint32_t label, bool match_input) : trs(&t), by_ilabel(match_input) {
    current_loop = label == EPS_LABEL;                   // :127
    match_label = label == NO_LABEL ? EPS_LABEL : label;  // :131-135
    if (current_loop) {
      pos = 0;  // :138-139
    } else {    // superslice lower_bound_by :141-142
      size_t lo = 0, hi = t.size();
      while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        uint32_t key = by_ilabel ? t[mid].ilabel : t[mid].olabel;
        if (key < match_label)
          lo = mid + 1;
        else
          hi = mid;
      }
      pos = lo;
    }
  }
  // next() :168-184.  Returns 0 = exhausted, 1 = EpsLoop, 2 = real arc (*out)
  int next(const Tr** out) {
    if (current_loop) {
      current_loop = false;
      return 1;
    }
    if (pos < trs->size()) {
      const Tr& tr = (*trs)[pos];
      uint32_t key = by_ilabel ? tr.ilabel : tr.olabel;
      if (key == match_label) {
        pos++;
        *out = &tr;
        return 2;
      }
    }
    return 0;
  }
};

// The compose filters reachable through ComposeFilterEnum (compose_static.rs:19-33): one struct, one branch per
// reference file.  Filter states: IntegerFilterState (u32, NO_STATE_ID = blocking) for Sequence / AltSequence /
// Match; TrivialFilterState (bool, false = blocking) for Null / Trivial / NoMatch — represented here as 0 = the
// single live state `true` and NO_STATE_ID = `false`.
enum FilterKind : int { F_AUTO = 0, F_NULL = 1, F_TRIVIAL = 2, F_SEQUENCE = 3, F_ALT_SEQUENCE = 4, F_MATCH = 5, F_NO_MATCH = 6 };

struct SequenceFilter {
  const Fst* fst1;
  const Fst* fst2 = nullptr;
  FilterKind kind = F_SEQUENCE;
  uint32_t s1 = NO_STATE_ID, s2 = NO_STATE_ID, fs = NO_STATE_ID;
  bool alleps1 = false, noeps1 = false, alleps2 = false, noeps2 = false;
  // set_state: sequence_compose_filter.rs:134-148, alt_sequence_compose_filter.rs:143-158,
  // match_compose_filter.rs:126-147 (fst1: OUTPUT epsilons, fst2: INPUT epsilons); no-op for the trivial-state filters
  void set_state(uint32_t s1_, uint32_t s2_, uint32_t fs_) {
    if (!(s1 == s1_ && s2 == s2_ && fs == fs_)) {
      s1 = s1_;
      s2 = s2_;
      fs = fs_;
      const State& st = fst1->states[s1];
      alleps1 = st.trs.size() == st.noepsilons && !st.has_final;
      noeps1 = st.noepsilons == 0;
      if (fst2) {
        const State& st2 = fst2->states[s2];
        alleps2 = st2.trs.size() == st2.niepsilons && !st2.has_final;
        noeps2 = st2.niepsilons == 0;
      }
    }
  }
  uint32_t filter_tr(const Tr& arc1, const Tr& arc2) const {
    switch (kind) {
      case F_NULL:  // null_compose_filter.rs:122-129
        return (arc1.olabel == NO_LABEL || arc2.ilabel == NO_LABEL) ? NO_STATE_ID : 0u;
      case F_TRIVIAL:  // trivial_compose_filter.rs:122-124
        return 0u;
      case F_NO_MATCH:  // no_match_compose_filter.rs:122-126
        return (arc1.olabel != EPS_LABEL || arc2.ilabel != EPS_LABEL) ? 0u : NO_STATE_ID;
      case F_ALT_SEQUENCE:  // alt_sequence_compose_filter.rs:160-181
        if (arc2.ilabel == NO_LABEL) {
          if (alleps2) return NO_STATE_ID;
          return noeps2 ? 0u : 1u;
        } else if (arc1.olabel == NO_LABEL) {
          return fs == 1 ? NO_STATE_ID : 0u;
        } else if (arc1.olabel == EPS_LABEL) {
          return NO_STATE_ID;
        }
        return 0u;
      case F_MATCH:  // match_compose_filter.rs:149-205
        if (arc2.ilabel == NO_LABEL) {  // epsilon in fst1
          if (fs == 0) return noeps2 ? 0u : (alleps2 ? NO_STATE_ID : 1u);
          return fs == 1 ? 1u : NO_STATE_ID;
        } else if (arc1.olabel == NO_LABEL) {  // epsilon in fst2
          if (fs == 0) return noeps1 ? 0u : (alleps1 ? NO_STATE_ID : 2u);
          return fs == 2 ? 2u : NO_STATE_ID;
        } else if (arc1.olabel == EPS_LABEL) {  // epsilon in both
          return fs == 0 ? 0u : NO_STATE_ID;
        }
        return 0u;
      default:  // F_AUTO / F_SEQUENCE: sequence_compose_filter.rs:150-171
        if (arc1.olabel == NO_LABEL) {
          if (alleps1) return NO_STATE_ID;
          return noeps1 ? 0u : 1u;
        } else if (arc2.ilabel == NO_LABEL) {
          return fs != 0 ? NO_STATE_ID : 0u;
        } else if (arc1.olabel == EPS_LABEL) {
          return NO_STATE_ID;
        }
        return 0u;
    }
  }
};

// ComposeFstOp — compose/compose_fst_op.rs
struct ComposeOp {
  const Fst& fst1;
  const Fst& fst2;
  MatchType match_type;
  uint64_t properties;  // compose_filter.properties(cprops) is the identity for all six filters
  StateTable table;
  FilterKind filter_kind = F_SEQUENCE;

  ComposeOp(const Fst& a, const Fst& b, MatchType mt, FilterKind fk = F_SEQUENCE)
      : fst1(a), fst2(b), match_type(mt), properties(P::compose_properties(a.properties, b.properties)), filter_kind(fk) {}

  // compute_start :389-404
  bool compute_start(uint32_t* out) {
    if (!fst1.has_start) return false;
    if (!fst2.has_start) return false;
    *out = table.find_id(Tuple{0u, fst1.start, fst2.start});
    return true;
  }
  // match_input :199-219 (SortedMatcher::priority = num_trs, sorted_matcher.rs:91-93)
  bool match_input(uint32_t s1, uint32_t s2) const {
    switch (match_type) {
      case MatchType::MatchInput: return true;
      case MatchType::MatchOutput: return false;
      default: return fst1.states[s1].trs.size() <= fst2.states[s2].trs.size();
    }
  }
  // add_tr :267-285
  Tr add_tr(Tr arc1, const Tr& arc2, uint32_t fs) {
    Tuple tuple{fs, arc1.nextstate, arc2.nextstate};
    arc1.weight = wtimes(arc1.weight, arc2.weight);
    return Tr{arc1.ilabel, arc2.olabel, arc1.weight, table.find_id(tuple)};
  }
  // match_tr :324-353 + match_tr_selected :287-322
  void match_tr(uint32_t sa, const Tr& tr, bool mi, SequenceFilter& filter, std::vector<Tr>& trs) {
    const uint32_t label = mi ? tr.olabel : tr.ilabel;
    // Fst1Matcher2 (mi): matcher2 on fst2 by ilabel; Fst2Matcher1: matcher1 on fst1 by olabel
    const std::vector<Tr>& sa_trs = mi ? fst2.states[sa].trs : fst1.states[sa].trs;
    MatcherIter it(sa_trs, label, /*by_ilabel=*/mi);
    const Tr* real = nullptr;
    for (;;) {
      int k = it.next(&real);
      if (k == 0) break;
      Tr arca;
      if (k == 1) {  // eps_loop(sa, match_type): matchers/mod.rs:98-105
        arca = mi ? Tr{NO_LABEL, EPS_LABEL, 0.0f, sa} : Tr{EPS_LABEL, NO_LABEL, 0.0f, sa};
      } else {
        arca = *real;
      }
      Tr arcb = tr;
      if (mi) {
        uint32_t fs = filter.filter_tr(arcb, arca);
        if (fs != NO_STATE_ID) trs.push_back(add_tr(arcb, arca, fs));
      } else {
        uint32_t fs = filter.filter_tr(arca, arcb);
        if (fs != NO_STATE_ID) trs.push_back(add_tr(arca, arcb, fs));
      }
    }
  }
  // ordered_expand :221-265
  std::vector<Tr> ordered_expand(uint32_t sa, uint32_t sb, bool mi, SequenceFilter& filter) {
    Tr tr_loop = mi ? Tr{EPS_LABEL, NO_LABEL, 0.0f, sb} : Tr{NO_LABEL, EPS_LABEL, 0.0f, sb};
    std::vector<Tr> trs;
    match_tr(sa, tr_loop, mi, filter, trs);
    const std::vector<Tr>& sb_trs = mi ? fst1.states[sb].trs : fst2.states[sb].trs;
    for (const Tr& tr : sb_trs) match_tr(sa, tr, mi, filter, trs);
    return trs;
  }
  // compute_trs :406-418
  std::vector<Tr> compute_trs(uint32_t state) {
    Tuple tuple = table.find_tuple(state);
    SequenceFilter filter{&fst1, &fst2, filter_kind};
    filter.set_state(tuple.s1, tuple.s2, tuple.fs);
    if (match_input(tuple.s1, tuple.s2)) return ordered_expand(tuple.s2, tuple.s1, true, filter);
    return ordered_expand(tuple.s1, tuple.s2, false, filter);
  }
  // compute_final_weight :420-449
  bool compute_final_weight(uint32_t state, float* out) const {
    Tuple tuple = table.find_tuple(state);
    const State& a = fst1.states[tuple.s1];
    if (!a.has_final) return false;
    const State& b = fst2.states[tuple.s2];
    if (!b.has_final) return false;
    float f = wtimes(a.final_w, b.final_w);
    if (wis_zero(f)) return false;
    *out = f;
    return true;
  }
};

void connect_impl(Fst& fst);

// compose_with_config (AutoFilter / SequenceFilter + SortedMatcher) — compose_static.rs:166-266;
// LazyFst::compute — lazy/lazy_fst.rs:226-269
bool compose_impl(const Fst& fst1, const Fst& fst2, bool connect, Fst& fst_out, uint64_t* arcs_pre_trim,
                  int filter = F_AUTO) {
  MatchType mt;
  if (!compose_match_type(fst1, fst2, &mt)) return false;
  if (filter < F_AUTO || filter > F_NO_MATCH) {
    t_err = "unknown compose filter";
    return false;
  }
  ComposeOp op(fst1, fst2, mt, filter == F_AUTO ? F_SEQUENCE : (FilterKind)filter);
  fst_out = Fst();
  uint32_t start_state;
  uint64_t n_arcs = 0;
  if (op.compute_start(&start_state)) {
    fst_out.add_states((size_t)start_state + 1);
    fst_out.set_start(start_state);
    std::deque<uint32_t> queue;
    std::vector<bool> visited((size_t)start_state + 1, false);
    visited[start_state] = true;
    queue.push_back(start_state);
    while (!queue.empty()) {
      uint32_t s = queue.front();
      queue.pop_front();
      std::vector<Tr> trs = op.compute_trs(s);
      for (const Tr& tr : trs) {
        if ((size_t)tr.nextstate >= visited.size()) visited.resize((size_t)tr.nextstate + 1, false);
        if (!visited[tr.nextstate]) {
          queue.push_back(tr.nextstate);
          visited[tr.nextstate] = true;
        }
        size_t n = fst_out.num_states();
        if ((size_t)tr.nextstate >= n) fst_out.add_states((size_t)tr.nextstate - n + 1);
      }
      n_arcs += trs.size();
      fst_out.set_trs_unchecked(s, std::move(trs));
      float fw;
      if (op.compute_final_weight(s, &fw)) fst_out.set_final(s, fw);
    }
    fst_out.set_properties(op.properties);  // lazy_fst.rs:260
  }
  // (start None: `return Ok(fst_out)` with F2::new() properties, lazy_fst.rs:229-232)
  if (arcs_pre_trim) *arcs_pre_trim = n_arcs;
  if (connect) connect_impl(fst_out);  // compose_static.rs:261-263
  return true;
}

// ---------------------------------------------------------------- dfs_visit (dfs_visit.rs:97-187)
struct KeepAll {  // AnyTrFilter (tr_filters.rs)
  bool operator()(const Tr&) const { return true; }
};
struct KeepEpsilon {  // EpsilonTrFilter (tr_filters.rs:25-31)
  bool operator()(const Tr& tr) const { return tr.ilabel == EPS_LABEL && tr.olabel == EPS_LABEL; }
};
template <class V, class K = KeepAll>
void dfs_visit(const Fst& fst, V& visitor, bool access_only, K keep = K()) {
  visitor.init_visit(fst);
  if (!fst.has_start) {
    visitor.finish_visit();
    return;
  }
  const uint32_t start = fst.start;
  const size_t nstates = fst.num_states();
  enum : uint8_t { White, Grey, Black };
  std::vector<uint8_t> color(nstates, White);
  struct DfsState {
    uint32_t state_id;
    size_t pos;
  };
  std::vector<DfsState> stack;
  bool dfs = true;
  uint32_t root = start;
  for (;;) {
    if (!dfs || (size_t)root >= nstates) break;
    color[root] = Grey;
    stack.push_back({root, 0});
    dfs = visitor.init_state(root, root);
    while (!stack.empty()) {
      DfsState& ds = stack.back();
      const uint32_t s = ds.state_id;
      const std::vector<Tr>& trs = fst.states[s].trs;
      if (!dfs || ds.pos >= trs.size()) {
        color[s] = Black;
        stack.pop_back();
        if (!stack.empty()) {
          DfsState& parent = stack.back();
          visitor.finish_state(s, true, parent.state_id);
          parent.pos++;
        } else {
          visitor.finish_state(s, false, 0);
        }
        continue;
      }
      const Tr& tr = trs[ds.pos];
      if (!keep(tr)) {  // dfs_visit.rs:146-149
        ds.pos++;
        continue;
      }
      switch (color[tr.nextstate]) {
        case White:
          dfs = visitor.tree_tr(s, tr);
          if (!dfs) break;
          color[tr.nextstate] = Grey;
          dfs = visitor.init_state(tr.nextstate, root);
          stack.push_back({tr.nextstate, 0});  // (ds is invalid from here)
          break;
        case Grey:
          dfs = visitor.back_tr(s, tr);
          ds.pos++;
          break;
        default:
          dfs = visitor.forward_or_cross_tr(s, tr);
          ds.pos++;
          break;
      }
    }
    if (access_only) break;
    root = root == start ? 0 : root + 1;
    while ((size_t)root < nstates && color[root] != White) root++;
  }
  visitor.finish_visit();
}
// NOTE on the `White` branch: the reference `break`s out of the inner while when tree_tr returns
// false (dfs_visit.rs:150-153); no visitor used on this path returns false from tree_tr.

// SccVisitor — visitors/scc_visitors.rs:10-180 (ConnectVisitor connect.rs:69-189 is the same
// machinery without scc ids; one struct serves both)
struct SccVisitor {
  const Fst* fst = nullptr;
  bool compute_scc;
  std::vector<int32_t> scc;
  std::vector<bool> access, coaccess;
  uint32_t start = NO_STATE_ID;
  size_t nstates = 0;
  std::vector<int32_t> dfnumber, lowlink;
  std::vector<bool> onstack;
  std::vector<uint32_t> scc_stack;
  int32_t nscc = 0;
  explicit SccVisitor(const Fst& f, bool compute_scc_) : fst(&f), compute_scc(compute_scc_) {
    size_t n = f.num_states();
    if (compute_scc) scc.assign(n, -1);
    access.assign(n, false);
    coaccess.assign(n, false);
    start = f.has_start ? f.start : NO_STATE_ID;
    dfnumber.assign(n, -1);
    lowlink.assign(n, -1);
    onstack.assign(n, false);
  }
  void init_visit(const Fst&) {}
  bool init_state(uint32_t s, uint32_t root) {  // connect.rs:106-115 / scc_visitors.rs:68-88
    scc_stack.push_back(s);
    dfnumber[s] = (int32_t)nstates;
    lowlink[s] = (int32_t)nstates;
    onstack[s] = true;
    access[s] = root == start;
    nstates++;
    return true;
  }
  bool tree_tr(uint32_t, const Tr&) { return true; }
  bool back_tr(uint32_t s, const Tr& tr) {  // connect.rs:121-131
    uint32_t t = tr.nextstate;
    if (dfnumber[t] < lowlink[s]) lowlink[s] = dfnumber[t];
    if (coaccess[t]) coaccess[s] = true;
    return true;
  }
  bool forward_or_cross_tr(uint32_t s, const Tr& tr) {  // connect.rs:133-146
    uint32_t t = tr.nextstate;
    if (dfnumber[t] < dfnumber[s] && onstack[t] && dfnumber[t] < lowlink[s]) lowlink[s] = dfnumber[t];
    if (coaccess[t]) coaccess[s] = true;
    return true;
  }
  void finish_state(uint32_t s, bool has_parent, uint32_t parent) {  // connect.rs:148-185 / scc_visitors.rs:128-170
    if (fst->states[s].has_final) coaccess[s] = true;
    if (dfnumber[s] == lowlink[s]) {
      bool scc_coaccess = false;
      size_t i = scc_stack.size();
      uint32_t t;
      do {
        i--;
        t = scc_stack[i];
        if (coaccess[t]) scc_coaccess = true;
      } while (s != t);
      do {
        t = scc_stack.back();
        if (compute_scc) scc[t] = nscc;
        if (scc_coaccess) coaccess[t] = true;
        onstack[t] = false;
        scc_stack.pop_back();
      } while (s != t);
      nscc++;
    }
    if (has_parent) {
      if (coaccess[s]) coaccess[parent] = true;
      if (lowlink[s] < lowlink[parent]) lowlink[parent] = lowlink[s];
    }
  }
  void finish_visit() {  // scc_visitors.rs:172-179
    if (compute_scc)
      for (auto& c : scc) c = nscc - 1 - c;
  }
};

// connect — connect.rs:51-66
void connect_impl(Fst& fst) {
  SccVisitor visitor(fst, false);
  dfs_visit(fst, visitor, false);
  std::vector<uint32_t> dstates;
  for (size_t s = 0; s < visitor.access.size(); ++s)
    if (!visitor.access[s] || !visitor.coaccess[s]) dstates.push_back((uint32_t)s);
  fst.del_states(dstates);
  fst.set_properties_with_mask(P::ACCESSIBLE | P::COACCESSIBLE, P::ACCESSIBLE | P::COACCESSIBLE);
}

// TopOrderVisitor — top_sort.rs:12-61
struct TopOrderVisitor {
  std::vector<uint32_t> order, finish;
  bool acyclic = true;
  void init_visit(const Fst&) {}
  bool init_state(uint32_t, uint32_t) { return true; }
  bool tree_tr(uint32_t, const Tr&) { return true; }
  bool back_tr(uint32_t, const Tr&) {
    acyclic = false;
    return false;
  }
  bool forward_or_cross_tr(uint32_t, const Tr&) { return true; }
  void finish_state(uint32_t s, bool, uint32_t) { finish.push_back(s); }
  void finish_visit() {
    if (acyclic) {
      order.assign(finish.size(), 0);
      for (size_t s = 0; s < finish.size(); ++s) order[finish[finish.size() - s - 1]] = (uint32_t)s;
    }
  }
};

// ---------------------------------------------------------------- queues (B2)
enum class QueueType { Trivial, Fifo, Lifo, ShortestFirst };  // queue.rs:6-26 (subset used)

struct Queue {  // queue.rs:30-38
  virtual ~Queue() = default;
  virtual void enqueue(uint32_t s) = 0;
  virtual bool dequeue(uint32_t* s) = 0;
  virtual void update(uint32_t) {}
  virtual bool is_empty() const = 0;
  virtual void clear() = 0;
};
struct FifoQueue : Queue {  // queues/fifo_queue.rs
  std::deque<uint32_t> q;
  void enqueue(uint32_t s) override { q.push_back(s); }
  bool dequeue(uint32_t* s) override {
    if (q.empty()) return false;
    *s = q.front();
    q.pop_front();
    return true;
  }
  bool is_empty() const override { return q.empty(); }
  void clear() override { q.clear(); }
};
struct LifoQueue : Queue {  // queues/lifo_queue.rs
  std::vector<uint32_t> q;
  void enqueue(uint32_t s) override { q.push_back(s); }
  bool dequeue(uint32_t* s) override {
    if (q.empty()) return false;
    *s = q.back();
    q.pop_back();
    return true;
  }
  bool is_empty() const override { return q.empty(); }
  void clear() override { q.clear(); }
};
struct TrivialQueue : Queue {  // queues/trivial_queue.rs
  bool has = false;
  uint32_t st = 0;
  void enqueue(uint32_t s) override {
    has = true;
    st = s;
  }
  bool dequeue(uint32_t* s) override {
    if (!has) return false;
    *s = st;
    has = false;
    return true;
  }
  bool is_empty() const override { return !has; }
  void clear() override { has = false; }
};
struct StateOrderQueue : Queue {  // queues/state_order_queue.rs
  size_t front = 0;
  bool has_back = false;
  size_t back = 0;
  std::vector<bool> enqueued;
  void enqueue(uint32_t s) override {
    size_t state = s;
    if (!has_back || front > back) {
      front = state;
      back = state;
      has_back = true;
    } else if (state > back) {
      back = state;
    } else if (state < front) {
      front = state;
    }
    while (enqueued.size() <= state) enqueued.push_back(false);
    enqueued[state] = true;
  }
  bool dequeue(uint32_t* s) override {
    if (is_empty()) return false;
    *s = (uint32_t)front;
    enqueued[front] = false;
    if (has_back)
      while (front <= back && !enqueued[front]) front++;
    return true;
  }
  bool is_empty() const override { return has_back ? front > back : true; }
  void clear() override {
    if (has_back)
      for (size_t i = front; i <= back; ++i) enqueued[i] = false;
    front = 0;
    has_back = false;
  }
};
struct TopOrderQueue : Queue {  // queues/top_order_queue.rs:12-94
  std::vector<uint32_t> order;
  std::vector<int64_t> state;  // Option<StateId>: -1 = None
  uint32_t front = 0;
  bool has_back = false;
  uint32_t back = 0;
  explicit TopOrderQueue(std::vector<uint32_t> o) : order(std::move(o)), state(order.size(), -1) {}
  void enqueue(uint32_t s) override {
    if (!has_back || front > back) {
      front = order[s];
      back = order[s];
      has_back = true;
    } else if (order[s] > back) {
      back = order[s];
    } else if (order[s] < front) {
      front = order[s];
    }
    state[order[s]] = s;
  }
  bool dequeue(uint32_t* s) override {
    if (is_empty()) return false;
    int64_t old_head = state[front];
    state[front] = -1;
    if (has_back)
      while (front <= back && state[front] < 0) front++;
    // (the reference returns Option; None cannot happen when !is_empty)
    if (old_head < 0) return false;
    *s = (uint32_t)old_head;
    return true;
  }
  bool is_empty() const override { return has_back ? front > back : true; }
  void clear() override {
    if (has_back)
      for (uint32_t i = front; i <= back; ++i) state[i] = -1;
    front = 0;
    has_back = false;
  }
};
struct SccQueue : Queue {  // queues/scc_queue.rs:6-84
  int32_t front = 0, back = -1;
  std::vector<std::unique_ptr<Queue>> queues;
  std::vector<uint32_t> sccs;
  void update_front() {
    while (front <= back && queues[front]->is_empty()) front++;
  }
  void enqueue(uint32_t s) override {
    int32_t c = (int32_t)sccs[s];
    if (front > back) {
      front = c;
      back = c;
    } else if (c > back) {
      back = c;
    } else if (c < front) {
      front = c;
    }
    queues[sccs[s]]->enqueue(s);
  }
  bool dequeue(uint32_t* s) override {
    if (is_empty()) return false;
    update_front();
    return queues[front]->dequeue(s);
  }
  void update(uint32_t s) override { queues[sccs[s]]->update(s); }
  bool is_empty() const override {
    if (front < back) return false;
    if (front > back) return true;
    return queues[front]->is_empty();
  }
  void clear() override {
    for (int32_t i = front; i <= back; ++i) queues[i]->clear();
    front = 0;
    back = -1;
  }
};

// AutoQueue::new(fst, distance=None, AnyTrFilter) — queues/auto_queue.rs:23-157.
// With distance=None `less` is None (:52-59) so every intra-SCC arc selects FifoQueue (:130-131).
template <class K>
std::unique_ptr<Queue> auto_queue_new(const Fst& fst, K keep) {
  const uint64_t props = fst.properties;
  if ((props & P::TOP_SORTED) || !fst.has_start) {
    t_queue_kind = "state_order";
    return std::make_unique<StateOrderQueue>();
  }
  if (props & P::ACYCLIC) {
    TopOrderVisitor v;
    dfs_visit(fst, v, false, keep);
    // (reference panics if !acyclic, top_order_queue.rs:24-26)
    t_queue_kind = "top_order";
    return std::make_unique<TopOrderQueue>(std::move(v.order));
  }
  if (props & P::UNWEIGHTED) {  // Tropical is IDEMPOTENT
    t_queue_kind = "lifo";
    return std::make_unique<LifoQueue>();
  }
  SccVisitor sv(fst, true);
  dfs_visit(fst, sv, false, keep);
  std::vector<uint32_t> sccs(sv.scc.size());
  for (size_t i = 0; i < sccs.size(); ++i) sccs[i] = (uint32_t)sv.scc[i];
  const size_t n_sccs = (size_t)sv.nscc;
  std::vector<QueueType> queue_types(n_sccs, QueueType::Trivial);
  // scc_queue_type :101-157 with compare = None
  bool all_trivial = true, unweighted = true;
  for (size_t state = 0; state < fst.num_states(); ++state) {
    for (const Tr& tr : fst.states[state].trs) {
      if (!keep(tr)) continue;  // auto_queue.rs:123-125
      if (sccs[state] == sccs[tr.nextstate]) {
        QueueType& qt = queue_types[sccs[state]];
        qt = QueueType::Fifo;  // compare.is_none()
        if (qt != QueueType::Trivial) all_trivial = false;
      }
      if (!wis_zero(tr.weight) && !wis_one(tr.weight)) unweighted = false;
    }
  }
  if (unweighted) {
    t_queue_kind = "lifo";
    return std::make_unique<LifoQueue>();
  }
  if (all_trivial) {
    t_queue_kind = "top_order_scc";
    return std::make_unique<TopOrderQueue>(std::move(sccs));
  }
  auto q = std::make_unique<SccQueue>();
  q->queues.reserve(n_sccs);
  for (size_t i = 0; i < n_sccs; ++i) {
    if (queue_types[i] == QueueType::Trivial)
      q->queues.push_back(std::make_unique<TrivialQueue>());
    else
      q->queues.push_back(std::make_unique<FifoQueue>());
  }
  q->sccs = std::move(sccs);
  t_queue_kind = "scc";
  return q;
}
inline std::unique_ptr<Queue> auto_queue_new(const Fst& fst) { return auto_queue_new(fst, KeepAll()); }

// ---------------------------------------------------------------- shortest path n=1 (B1,B3)
// single_shortest_path — shortest_path.rs:173-239
struct Parent {
  bool some = false;
  uint32_t state = 0;
  size_t pos = 0;
};
void single_shortest_path(const Fst& ifst, std::vector<float>& distance, bool& has_f_parent, uint32_t& f_parent,
                          std::vector<Parent>& parent) {
  parent.clear();
  has_f_parent = false;
  if (!ifst.has_start) return;
  std::vector<bool> enqueued;
  std::unique_ptr<Queue> queue = auto_queue_new(ifst);
  const uint32_t source = ifst.start;
  float f_distance = INF;
  distance.clear();
  queue->clear();
  distance.assign(ifst.num_states(), INF);
  enqueued.assign(ifst.num_states(), false);
  parent.assign(ifst.num_states(), Parent{});
  distance[source] = 0.0f;
  enqueued[source] = true;
  queue->enqueue(source);
  uint32_t s;
  while (queue->dequeue(&s)) {
    enqueued[s] = false;
    const float sd = distance[s];
    const State& st = ifst.states[s];
    if (st.has_final) {
      float plus = wplus(f_distance, wtimes(sd, st.final_w));
      if (!weq(f_distance, plus)) {
        f_distance = plus;
        has_f_parent = true;
        f_parent = s;
      }
    }
    for (size_t pos = 0; pos < st.trs.size(); ++pos) {
      const Tr& tr = st.trs[pos];
      const size_t nextstate = tr.nextstate;
      float& nd = distance[nextstate];
      const float weight = wtimes(sd, tr.weight);
      if (!weq(nd, wplus(nd, weight))) {
        nd = wplus(nd, weight);
        parent[nextstate] = Parent{true, s, pos};
        if (!enqueued[nextstate]) {
          queue->enqueue((uint32_t)nextstate);
          enqueued[nextstate] = true;
        } else {
          queue->update((uint32_t)nextstate);
        }
      }
    }
  }
}

// single_shortest_path_backtrace — shortest_path.rs:241-282
bool backtrace(const Fst& ifst, bool has_f_parent, uint32_t f_parent, const std::vector<Parent>& parent, Fst& ofst) {
  ofst = Fst();
  bool has_s_p = false, has_d_p = false, has_d = false;
  uint32_t s_p = 0, d_p = 0, d = 0;
  bool has_next = has_f_parent;
  uint32_t state = f_parent;
  size_t guard = 0;
  while (has_next) {
    if (++guard > ifst.num_states() + 1) {
      t_err = "backtrace: parent chain is cyclic";
      return false;
    }
    has_d_p = has_s_p;
    d_p = s_p;
    s_p = ofst.add_state();
    has_s_p = true;
    if (has_d) {
      size_t pos = parent[d].pos;
      Tr tr = ifst.states[state].trs[pos];
      tr.nextstate = d_p;
      (void)has_d_p;
      ofst.add_tr(s_p, tr);
    } else if (ifst.states[f_parent].has_final) {
      ofst.set_final(s_p, ifst.states[f_parent].final_w);
    }
    d = state;
    has_d = true;
    has_next = parent[state].some;
    state = parent[state].state;
  }
  if (has_s_p) ofst.set_start(s_p);
  ofst.set_properties_with_mask(P::shortest_path_properties(ofst.properties, true), P::ALL);
  return true;
}

// ---------------------------------------------------------------- shortest_distance (B4)
// ShortestDistanceState::shortest_distance with AutoQueue::new(fst, None, AnyTrFilter), first_path = false,
// retain = false — shortest_distance.rs:153-237,313-335.  Convergence test is approx_equal(delta), not ==.
inline bool approx_equal(float a, float b, float delta) { return std::fabs(a - b) <= delta; }  // utils_float.rs:1-3

std::vector<float> shortest_distance_impl(const Fst& fst, float delta) {
  std::vector<float> distance, adder, radder;
  std::vector<bool> enqueued;
  if (!fst.has_start) return distance;
  std::unique_ptr<Queue> queue = auto_queue_new(fst);
  queue->clear();
  auto ensure = [&](size_t index) {
    while (distance.size() <= index) {
      distance.push_back(INF);
      enqueued.push_back(false);
      adder.push_back(INF);
      radder.push_back(INF);
    }
  };
  const size_t source = fst.start;
  ensure(source);
  distance[source] = 0.0f;
  adder[source] = 0.0f;
  radder[source] = 0.0f;
  enqueued[source] = true;
  queue->enqueue((uint32_t)source);
  uint32_t st;
  while (queue->dequeue(&st)) {
    const size_t state = st;
    enqueued[state] = false;
    const float r = radder[state];
    radder[state] = INF;
    for (const Tr& tr : fst.states[state].trs) {
      const size_t nextstate = tr.nextstate;
      ensure(nextstate);
      const float weight = wtimes(r, tr.weight);
      if (!approx_equal(distance[nextstate], wplus(distance[nextstate], weight), delta)) {
        adder[nextstate] = wplus(adder[nextstate], weight);
        distance[nextstate] = adder[nextstate];
        radder[nextstate] = wplus(radder[nextstate], weight);
        if (!enqueued[state]) {  // (sic) the reference tests enqueued[state], shortest_distance.rs:224
          queue->enqueue((uint32_t)nextstate);
          enqueued[nextstate] = true;
        } else {
          queue->update((uint32_t)nextstate);
        }
      }
    }
  }
  return distance;
}

// ---------------------------------------------------------------- rm_epsilon (N4) — algorithms/rm_epsilon/*.rs
// ShortestDistanceState with EpsilonTrFilter, retain = true (shortest_distance.rs:100-237): one instance serves every
// source; entries of states not touched by the current source keep their old values and are reset lazily (`sources`).
struct EpsSdState {
  std::unique_ptr<Queue> queue;
  std::vector<float> distance, adder, radder;
  std::vector<bool> enqueued;
  std::vector<int64_t> sources;  // Option<StateId>: -1 = None
  int64_t source_id = 0;
  float delta;
  void ensure(size_t index) {
    while (distance.size() <= index) {
      distance.push_back(INF);
      enqueued.push_back(false);
      adder.push_back(INF);
      radder.push_back(INF);
    }
  }
  void ensure_source(size_t index) {
    while (sources.size() <= index) sources.push_back(-1);
  }
  const std::vector<float>& shortest_distance(const Fst& fst, uint32_t src) {
    KeepEpsilon keep;
    queue->clear();
    const size_t source = src;
    ensure(source);
    ensure_source(source);
    sources[source] = source_id;
    distance[source] = 0.0f;
    adder[source] = 0.0f;
    radder[source] = 0.0f;
    enqueued[source] = true;
    queue->enqueue((uint32_t)source);
    uint32_t st;
    while (queue->dequeue(&st)) {
      const size_t state = st;
      enqueued[state] = false;
      const float r = radder[state];
      radder[state] = INF;
      for (const Tr& tr : fst.states[state].trs) {
        const size_t nextstate = tr.nextstate;
        if (!keep(tr)) continue;
        ensure(nextstate);
        ensure_source(nextstate);
        if (sources[nextstate] != source_id) {
          distance[nextstate] = INF;
          adder[nextstate] = INF;
          radder[nextstate] = INF;
          enqueued[nextstate] = false;
          sources[nextstate] = source_id;
        }
        const float weight = wtimes(r, tr.weight);
        if (!approx_equal(distance[nextstate], wplus(distance[nextstate], weight), delta)) {
          adder[nextstate] = wplus(adder[nextstate], weight);
          distance[nextstate] = adder[nextstate];
          radder[nextstate] = wplus(radder[nextstate], weight);
          if (!enqueued[state]) {  // (sic) shortest_distance.rs:224
            queue->enqueue((uint32_t)nextstate);
            enqueued[nextstate] = true;
          } else {
            queue->update((uint32_t)nextstate);
          }
        }
      }
    }
    source_id += 1;
    return distance;
  }
};

// rm_epsilon with the default config (connect = true, no thresholds, delta = KSHORTESTDELTA): rm_epsilon_static.rs:50-163,
// RmEpsilonState::expand rm_epsilon_state.rs:44-119, rmepsilon_properties mutate_properties.rs:646-660
bool rm_epsilon_impl(Fst& fst) {
  if (!fst.has_start) return true;
  KeepEpsilon keep;
  const size_t n = fst.num_states();
  // noneps_in[s]: s has a non-epsilon incoming arc or is the start state
  std::vector<bool> noneps_in(n, false);
  noneps_in[fst.start] = true;
  for (const State& st : fst.states)
    for (const Tr& tr : st.trs)
      if (tr.ilabel != EPS_LABEL || tr.olabel != EPS_LABEL) noneps_in[tr.nextstate] = true;
  // states in (generic) topological order of the epsilon graph
  std::vector<uint32_t> states;
  if (fst.properties & P::TOP_SORTED) {
    for (size_t s = 0; s < n; ++s) states.push_back((uint32_t)s);
  } else if (fst.properties & P::ACYCLIC) {
    TopOrderVisitor v;
    dfs_visit(fst, v, false, keep);
    states.assign(v.order.size(), 0);
    for (size_t i = 0; i < v.order.size(); ++i) states[v.order[i]] = (uint32_t)i;
  } else {
    SccVisitor v(fst, true);
    dfs_visit(fst, v, false, keep);
    const std::vector<int32_t>& scc = v.scc;
    std::vector<int64_t> first(scc.size(), -1), next(scc.size(), -1);
    for (size_t i = 0; i < scc.size(); ++i) {
      if (first[scc[i]] >= 0) next[i] = first[scc[i]];
      first[scc[i]] = (int64_t)i;
    }
    for (size_t c = 0; c < first.size(); ++c)
      for (int64_t j = first[c]; j >= 0; j = next[j]) states.push_back((uint32_t)j);
  }
  EpsSdState sd;
  sd.queue = auto_queue_new(fst, keep);  // AutoQueue::new(fst, None, &EpsilonTrFilter), rm_epsilon_static.rs:52
  sd.delta = 1e-6f;                      // KSHORTESTDELTA (lib.rs:271)
  // RmEpsilonState
  std::vector<bool> visited;
  std::vector<uint32_t> visited_states;
  struct Elt {
    uint32_t il, ol, ns;
    bool operator==(const Elt& o) const { return il == o.il && ol == o.ol && ns == o.ns; }
  };
  struct EltHash {
    size_t operator()(const Elt& e) const {
      uint64_t h = ((uint64_t)e.il << 32) ^ ((uint64_t)e.ol << 11) ^ e.ns;
      h *= 0x9E3779B97F4A7C15ull;
      return (size_t)(h ^ (h >> 29));
    }
  };
  std::unordered_map<Elt, std::pair<uint64_t, size_t>, EltHash> element_map;
  uint64_t expand_id = 0;
  for (size_t idx = states.size(); idx-- > 0;) {
    const uint32_t state = states[idx];
    if (!noneps_in[state]) continue;  // (connect is set)
    // expand
    const std::vector<float>& distance = sd.shortest_distance(fst, state);
    std::vector<uint32_t> eps_queue{state};
    std::vector<Tr> trs;
    float final_weight = INF;
    while (!eps_queue.empty()) {
      const uint32_t q = eps_queue.back();
      eps_queue.pop_back();
      while (visited.size() <= q) visited.push_back(false);
      if (visited[q]) continue;
      visited[q] = true;
      visited_states.push_back(q);
      for (const Tr& tr0 : fst.states[q].trs) {
        Tr tr = tr0;
        tr.weight = wtimes(distance[q], tr.weight);
        if (keep(tr)) {
          while (visited.size() <= tr.nextstate) visited.push_back(false);
          if (!visited[tr.nextstate]) eps_queue.push_back(tr.nextstate);
        } else {
          const Elt elt{tr.ilabel, tr.olabel, tr.nextstate};
          auto it = element_map.find(elt);
          if (it == element_map.end()) {
            element_map.emplace(elt, std::make_pair(expand_id, trs.size()));
            trs.push_back(tr);
          } else if (it->second.first == expand_id) {
            trs[it->second.second].weight = wplus(trs[it->second.second].weight, tr.weight);
          } else {
            it->second.first = expand_id;
            it->second.second = trs.size();
            trs.push_back(tr);
          }
        }
      }
      const State& qs = fst.states[q];
      final_weight = wplus(final_weight, wtimes(distance[q], qs.has_final ? qs.final_w : INF));
    }
    while (!visited_states.empty()) {
      visited[visited_states.back()] = false;
      visited_states.pop_back();
    }
    expand_id += 1;
    // pop_trs_unchecked (mutable_fst.rs:325-331: delete_trs_properties), set_trs_unchecked(reversed), final weight
    fst.properties &= P::DELETE_ARCS_MASK;
    std::reverse(trs.begin(), trs.end());
    fst.set_trs_unchecked(state, std::move(trs));
    State& st = fst.states[state];
    if (!wis_zero(final_weight)) {  // `final_weight != zero`: the KDELTA-approximate != of the weight
      fst.properties = P::set_final_properties(fst.properties, st.has_final ? &st.final_w : nullptr, &final_weight);
      st.has_final = true;
      st.final_w = final_weight;
    } else {
      fst.properties = P::set_final_properties(fst.properties, st.has_final ? &st.final_w : nullptr, nullptr);
      st.has_final = false;
      st.final_w = INF;
    }
  }
  for (size_t s = 0; s < fst.num_states(); ++s) {  // (connect is set) rm_epsilon_static.rs:137-143
    if (!noneps_in[s]) {
      State& st = fst.states[s];
      st.trs.clear();
      st.niepsilons = st.noepsilons = 0;
      fst.properties &= P::DELETE_ARCS_MASK;
    }
  }
  {  // rmepsilon_properties(props, delayed = false)
    const uint64_t in = fst.properties;
    uint64_t out = P::NO_EPSILONS;
    out |= (P::ACCEPTOR | P::ACYCLIC | P::INITIAL_ACYCLIC) & in;
    if (in & P::ACCEPTOR) out |= P::NO_I_EPSILONS | P::NO_O_EPSILONS;
    out |= P::TOP_SORTED & in;
    out |= P::NOT_ACCEPTOR & in;
    fst.set_properties(out);
  }
  connect_impl(fst);
  return true;
}

// ---------------------------------------------------------------- reverse (B5) — reverse.rs:33-87
// mutate_properties.rs:622-638
inline uint64_t reverse_properties(uint64_t in, bool has_superinitial) {
  uint64_t out = (P::ACCEPTOR | P::NOT_ACCEPTOR | P::EPSILONS | P::I_EPSILONS | P::O_EPSILONS | P::UNWEIGHTED | P::CYCLIC |
                  P::ACYCLIC | P::WEIGHTED_CYCLES | P::UNWEIGHTED_CYCLES) & in;
  if (has_superinitial) out |= P::WEIGHTED & in;
  return out;
}
void reverse_impl(const Fst& ifst, Fst& ofst) {
  ofst = Fst();
  const uint32_t ostart = ofst.add_state();
  ofst.add_states(ifst.num_states());
  std::vector<std::vector<Tr>> states_trs(ifst.num_states() + 1);
  for (size_t is = 0; is < ifst.num_states(); ++is) {
    const uint32_t os = (uint32_t)is + 1;
    if (ifst.has_start && ifst.start == is) ofst.set_final(os, 0.0f);
    const State& st = ifst.states[is];
    if (st.has_final) states_trs[0].push_back(Tr{EPS_LABEL, EPS_LABEL, st.final_w, os});
    for (const Tr& itr : st.trs) states_trs[(size_t)itr.nextstate + 1].push_back(Tr{itr.ilabel, itr.olabel, itr.weight, os});
  }
  for (size_t s = 0; s < states_trs.size(); ++s) ofst.set_trs_unchecked((uint32_t)s, std::move(states_trs[s]));
  ofst.set_start(ostart);
  ofst.set_properties_with_mask(reverse_properties(ifst.properties, true) | ofst.properties, P::ALL);
}

// ---------------------------------------------------------------- n_shortest_path (B6) — shortest_path.rs:284-518
inline bool natural_less(float w1, float w2) { return weq(wplus(w1, w2), w1) && !weq(w1, w2); }  // :284-286

struct NPair {
  bool some;  // Option<StateId>
  uint32_t state;
  float w;
};
struct NHeap {  // Heap :340-407 with ShortestPathCompare :288-338 as `less`
  std::vector<uint32_t> data;
  const std::vector<NPair>* pairs;
  const std::vector<float>* distance;
  float delta;
  float pweight(const NPair& p) const {
    if (!p.some) return 0.0f;
    return p.state < distance->size() ? (*distance)[p.state] : INF;
  }
  bool less(uint32_t x, uint32_t y) const {
    const NPair& px = (*pairs)[x];
    const NPair& py = (*pairs)[y];
    const float wx = wtimes(pweight(px), px.w);
    const float wy = wtimes(pweight(py), py.w);
    if (!px.some && py.some) return natural_less(wy, wx) || approx_equal(wx, wy, delta);
    if (px.some && !py.some) return natural_less(wy, wx) && !approx_equal(wx, wy, delta);
    return natural_less(wy, wx);
  }
  void sift_up(size_t idx) {
    while (idx > 0) {
      size_t parent = (idx - 1) / 2;
      if (less(data[parent], data[idx])) {
        std::swap(data[idx], data[parent]);
        idx = parent;
      } else {
        break;
      }
    }
  }
  void push(uint32_t v) {
    data.push_back(v);
    sift_up(data.size() - 1);
  }
  void sift_down(size_t idx) {
    for (;;) {
      const uint32_t cur = data[idx];
      const size_t c1 = 2 * idx + 1, c2 = 2 * idx + 2;
      size_t big;
      if (c1 >= data.size() && c2 >= data.size()) return;
      if (c1 < data.size() && c2 >= data.size())
        big = c1;
      else if (less(data[c1], data[c2]))
        big = c2;
      else
        big = c1;
      if (!less(data[big], cur)) {
        std::swap(data[idx], data[big]);
        idx = big;
      } else {
        return;
      }
    }
  }
  uint32_t pop() {
    const uint32_t top = data[0];
    if (data.size() == 1) {
      data.erase(data.begin());
    } else {
      data[0] = data.back();
      data.pop_back();
      sift_down(0);
    }
    return top;
  }
};

// ifst = the reversed FST, distance indexed by its state ids.
void n_shortest_path_impl(const Fst& ifst, const std::vector<float>& distance, size_t nshortest, float delta, Fst& ofst) {
  ofst = Fst();
  if (nshortest == 0) return;
  if (!ifst.has_start || distance.size() <= ifst.start || wis_zero(distance[ifst.start])) return;
  const uint32_t istart = ifst.start;
  const uint32_t ostart = ofst.add_state();
  ofst.set_start(ostart);
  const uint32_t final_state = ofst.add_state();
  ofst.set_final(final_state, 0.0f);
  std::vector<NPair> pairs(final_state + 1, NPair{false, 0, INF});
  pairs[final_state] = NPair{true, istart, 0.0f};
  NHeap heap;
  heap.pairs = &pairs;
  heap.distance = &distance;
  heap.delta = delta;
  heap.push(final_state);
  const float limit = wtimes(distance[istart], INF);  // weight_threshold = zero()
  std::vector<size_t> r;
  while (!heap.data.empty()) {
    const uint32_t state = heap.pop();
    const NPair p = pairs[state];
    const int64_t p_first_real = (p.some ? (int64_t)p.state : -1) + 1;
    const float d = p.some ? (p.state < distance.size() ? distance[p.state] : INF) : 0.0f;
    if (natural_less(limit, wtimes(d, p.w))) continue;
    while ((int64_t)r.size() <= p_first_real) r.push_back(0);
    r[(size_t)p_first_real] += 1;
    if (!p.some) ofst.add_tr(ofst.start, Tr{0, 0, 0.0f, state});
    if (!p.some && r[(size_t)p_first_real] == nshortest) break;
    if (r[(size_t)p_first_real] > nshortest) continue;
    if (!p.some) continue;
    for (const Tr& rarc : ifst.states[p.state].trs) {
      Tr tr{rarc.ilabel, rarc.olabel, rarc.weight, rarc.nextstate};
      const float weight = wtimes(p.w, tr.weight);
      const uint32_t next = ofst.add_state();
      pairs.push_back(NPair{true, tr.nextstate, weight});
      tr.nextstate = state;
      ofst.add_tr(next, tr);
      heap.push(next);
    }
    const State& st = ifst.states[p.state];
    if (st.has_final && !wis_zero(st.final_w)) {
      const float weight = wtimes(p.w, st.final_w);
      const uint32_t next = ofst.add_state();
      pairs.push_back(NPair{false, 0, weight});
      ofst.add_tr(next, Tr{0, 0, st.final_w, state});
      heap.push(next);
    }
  }
  connect_impl(ofst);
  ofst.set_properties_with_mask(P::shortest_path_properties(ofst.properties, false), P::ALL);
}

// shortest_path_with_config, nshortest > 1, unique = false — shortest_path.rs:135-170
bool shortest_path_n_impl(const Fst& ifst, size_t nshortest, float delta, Fst& out) {
  std::vector<float> distance = shortest_distance_impl(ifst, delta);
  Fst rfst;
  reverse_impl(ifst, rfst);
  float d = INF;
  for (const Tr& rarc : rfst.states[0].trs) {
    const uint32_t state = rarc.nextstate - 1;
    if ((size_t)state < distance.size()) d = wplus(d, wtimes(rarc.weight, distance[state]));
  }
  std::vector<float> distance_2;
  distance_2.reserve(distance.size() + 1);
  distance_2.push_back(d);
  distance_2.insert(distance_2.end(), distance.begin(), distance.end());
  n_shortest_path_impl(rfst, distance_2, nshortest, delta, out);
  return true;
}

// ---------------------------------------------------------------- determinize_with_distance (the `unique` branch)
// DeterminizeFsa<DefaultCommonDivisor> materialised by LazyFst::compute, with the distance of every new state to the final
// states — determinize/determinize_static.rs:24-39, determinize_fsa_op.rs:43-196, state_table.rs:17-96,
// lazy/lazy_fst.rs:226-269.  One deliberate choice: after merging the duplicates of a destination subset the reference
// collects them from a std HashMap (determinize_fsa_op.rs:156-168), i.e. in an unspecified order that takes part in the
// identity of the tuple (Vec equality, element.rs:20-23) — the same weighted subset may become several states there, and
// which ones differs from run to run.  Here the elements stay in the order the sort above them left (ascending state id):
// every weighted subset is one state.  Strings and their weights are the same either way.
struct DetElt {
  uint32_t state;
  float w;
};
bool determinize_with_distance_impl(const Fst& ifst, const std::vector<float>& in_dist, float delta, Fst& dfst,
                                    std::vector<float>& out_dist) {
  dfst = Fst();
  out_dist.clear();
  if (!(ifst.properties & P::ACCEPTOR)) {  // determinize_fsa_op.rs:138-140
    t_err = "DeterminizeFsaImpl : expected acceptor as argument";
    return false;
  }
  std::vector<std::vector<DetElt>> tuples;  // DeterminizeStateTable: id <-> (subset, filter_state = start, always)
  std::unordered_map<uint64_t, std::vector<uint32_t>> by_states;  // candidates with the same state ids
  auto states_hash = [](const std::vector<DetElt>& t) {
    uint64_t h = 1469598103934665603ull;
    for (const DetElt& e : t) h = (h ^ e.state) * 1099511628211ull;
    return h;
  };
  auto find_state = [&](const std::vector<DetElt>& t) -> uint32_t {  // state_table.rs:79-96
    std::vector<uint32_t>& cand = by_states[states_hash(t)];
    for (uint32_t id : cand) {
      const std::vector<DetElt>& o = tuples[id];
      bool same = o.size() == t.size();
      for (size_t k = 0; same && k < t.size(); ++k) same = o[k].state == t[k].state && weq(o[k].w, t[k].w);
      if (same) return id;
    }
    const uint32_t n = (uint32_t)tuples.size();
    tuples.push_back(t);
    cand.push_back(n);
    float outd = INF;  // compute_distance, state_table.rs:25-39
    for (const DetElt& e : t) outd = wplus(outd, wtimes(e.w, e.state < in_dist.size() ? in_dist[e.state] : INF));
    out_dist.push_back(outd);
    return n;
  };
  if (!ifst.has_start) return true;  // compute_start -> None: the empty FST (lazy_fst.rs:229-232)
  find_state({DetElt{ifst.start, 0.0f}});  // determinize_fsa_op.rs:45-55
  dfst.add_state();
  dfst.set_start(0);
  // lazy_fst.rs:235-259: ids are handed out in the order find_state first sees a tuple, and the queue pops them in that order
  for (uint32_t s = 0; s < tuples.size(); ++s) {
    std::map<uint32_t, std::vector<DetElt>> label_map;  // BTreeMap: ascending label (determinize_fsa_op.rs:59-84)
    {
      const std::vector<DetElt> src = tuples[s];
      for (const DetElt& e : src)
        for (const Tr& tr : ifst.states[e.state].trs) label_map[tr.ilabel].push_back(DetElt{tr.nextstate, wtimes(e.w, tr.weight)});
    }
    std::vector<Tr> trs;
    for (auto& kv : label_map) {  // norm_tr, determinize_fsa_op.rs:149-179
      std::vector<DetElt>& pairs = kv.second;
      std::stable_sort(pairs.begin(), pairs.end(), [](const DetElt& a, const DetElt& b) { return a.state < b.state; });
      float weight = INF;
      for (const DetElt& e : pairs) weight = wplus(weight, e.w);  // DefaultCommonDivisor = plus (divisors.rs:17-21)
      std::vector<DetElt> merged;
      for (const DetElt& e : pairs) {
        if (!merged.empty() && merged.back().state == e.state) merged.back().w = wplus(merged.back().w, e.w);
        else merged.push_back(e);
      }
      for (DetElt& e : merged) e.w = quantize(e.w - weight, delta);  // divide (tropical_weight.rs:128-131), quantize
      trs.push_back(Tr{kv.first, kv.first, weight, find_state(merged)});
    }
    while (dfst.states.size() < tuples.size()) dfst.add_state();
    dfst.states[s].trs = trs;  // set_trs_unchecked: no per-arc property updates (lazy_fst.rs:255)
    float fw = INF;  // compute_final_weight, determinize_fsa_op.rs:101-118
    for (const DetElt& e : tuples[s]) {
      const State& st = ifst.states[e.state];
      fw = wplus(fw, wtimes(e.w, st.has_final ? st.final_w : INF));
    }
    if (!wis_zero(fw)) {
      dfst.states[s].has_final = true;
      dfst.states[s].final_w = fw;
    }
  }
  dfst.properties = 0;  // DeterminizeFsaOp::properties(): empty (determinize_fsa_op.rs:120-123)
  return true;
}

// shortest_path_with_config, nshortest > 1, unique = true — shortest_path.rs:135-170 (the else branch at :157-165)
bool shortest_path_n_unique_impl(const Fst& ifst, size_t nshortest, float delta, Fst& out) {
  std::vector<float> distance = shortest_distance_impl(ifst, delta);
  Fst rfst;
  reverse_impl(ifst, rfst);
  float d = INF;
  for (const Tr& rarc : rfst.states[0].trs) {
    const uint32_t state = rarc.nextstate - 1;
    if ((size_t)state < distance.size()) d = wplus(d, wtimes(rarc.weight, distance[state]));
  }
  std::vector<float> distance_2;
  distance_2.reserve(distance.size() + 1);
  distance_2.push_back(d);
  distance_2.insert(distance_2.end(), distance.begin(), distance.end());
  Fst dfst;
  std::vector<float> distance_3;  // (TropicalWeight::ReverseWeight = TropicalWeight: the conversions are identities)
  if (!determinize_with_distance_impl(rfst, distance_2, delta, dfst, distance_3)) return false;
  n_shortest_path_impl(dfst, distance_3, nshortest, delta, out);
  return true;
}

// ---------------------------------------------------------------- canonical (deterministic-tie) shortest path
struct Canon {
  std::vector<float> d;
  std::vector<uint32_t> h;
  bool has_final = false;
  uint32_t f_parent = 0;
  float total = INF;
  std::vector<Parent> parent;
};
inline bool key_less(float d1, uint32_t h1, float d2, uint32_t h2) { return d1 < d2 || (d1 == d2 && h1 < h2); }

void canonical_sssp(const Fst& f, Canon& c) {
  const size_t n = f.num_states();
  c.d.assign(n, INF);
  c.h.assign(n, 0xFFFFFFFFu);
  c.parent.assign(n, Parent{});
  if (!f.has_start) return;
  std::deque<uint32_t> q;
  std::vector<bool> inq(n, false);
  c.d[f.start] = 0.0f;
  c.h[f.start] = 0;
  q.push_back(f.start);
  inq[f.start] = true;
  while (!q.empty()) {
    uint32_t s = q.front();
    q.pop_front();
    inq[s] = false;
    const float sd = c.d[s];
    const uint32_t sh = c.h[s];
    for (const Tr& tr : f.states[s].trs) {
      float cand = (sd + tr.weight) + 0.0f;  // +0.0f canonicalises -0.0
      if (!(cand < INF)) continue;           // +inf / NaN never relax (plus(inf) == inf in the reference)
      uint32_t ch = sh + 1;
      uint32_t t = tr.nextstate;
      if (key_less(cand, ch, c.d[t], c.h[t])) {
        c.d[t] = cand;
        c.h[t] = ch;
        if (!inq[t]) {
          inq[t] = true;
          q.push_back(t);
        }
      }
    }
  }
  // final: min (d[s]+rho(s), s)
  for (size_t s = 0; s < n; ++s) {
    const State& st = f.states[s];
    if (!st.has_final || !(c.d[s] < INF)) continue;
    float tot = c.d[s] + st.final_w;
    if (!(tot < INF)) continue;
    if (!c.has_final || tot < c.total) {
      c.has_final = true;
      c.total = tot;
      c.f_parent = (uint32_t)s;
    }
  }
  // parents: min (s,pos) among layered tight arcs
  for (size_t s = 0; s < n; ++s) {
    if (!(c.d[s] < INF)) continue;
    const State& st = f.states[s];
    for (size_t pos = 0; pos < st.trs.size(); ++pos) {
      const Tr& tr = st.trs[pos];
      uint32_t t = tr.nextstate;
      float cand = (c.d[s] + tr.weight) + 0.0f;
      if (!(cand < INF)) continue;
      if (cand == c.d[t] && c.h[s] + 1 == c.h[t]) {
        Parent& p = c.parent[t];
        if (!p.some) p = Parent{true, (uint32_t)s, pos};  // s ascending, pos ascending => first is min
      }
    }
  }
  // Fallback (class 1 of the engine's predecessor rule, sssp.hip parent_class): with inexact f32 sums a state may keep
  // a hop count derived from a label of its predecessor that was later improved in the distance, so that no arc is
  // tight in the hop count.  Then: first arc (s, pos ascending) that is tight in the distance and whose source label
  // is lexicographically below the target's.
  for (size_t s = 0; s < n; ++s) {
    if (!(c.d[s] < INF)) continue;
    const State& st = f.states[s];
    for (size_t pos = 0; pos < st.trs.size(); ++pos) {
      const Tr& tr = st.trs[pos];
      uint32_t t = tr.nextstate;
      if (c.parent[t].some || (f.has_start && t == f.start)) continue;
      float cand = (c.d[s] + tr.weight) + 0.0f;
      if (!(cand < INF)) continue;
      if (cand == c.d[t] && key_less(c.d[s], c.h[s], c.d[t], c.h[t])) c.parent[t] = Parent{true, (uint32_t)s, pos};
    }
  }
}

// count path positions with >1 (unlayered) tight incoming arcs, plus a final-state tie
uint32_t canonical_count_ties(const Fst& f, const Canon& c) {
  if (!c.has_final) return 0;
  const size_t n = f.num_states();
  std::vector<uint8_t> on_path(n, 0);
  uint32_t cur = c.f_parent;
  for (size_t guard = 0; guard <= n; ++guard) {
    on_path[cur] = 1;
    if (!c.parent[cur].some) break;
    cur = c.parent[cur].state;
  }
  std::vector<uint32_t> tight(n, 0);
  for (size_t s = 0; s < n; ++s) {
    if (!(c.d[s] < INF)) continue;
    for (const Tr& tr : f.states[s].trs) {
      if (!on_path[tr.nextstate]) continue;
      float cand = (c.d[s] + tr.weight) + 0.0f;
      if (cand < INF && cand == c.d[tr.nextstate]) tight[tr.nextstate]++;
    }
  }
  uint32_t ties = 0;
  for (size_t s = 0; s < n; ++s) {
    if (!on_path[s]) continue;
    uint32_t expect = (f.has_start && s == f.start) ? 0u : 1u;
    if (tight[s] > expect) ties++;
  }
  uint32_t nfinal = 0;
  for (size_t s = 0; s < n; ++s) {
    const State& st = f.states[s];
    if (st.has_final && c.d[s] < INF && c.d[s] + st.final_w == c.total) nfinal++;
  }
  if (nfinal > 1) ties++;
  return ties;
}

// ---------------------------------------------------------------- binary I/O (L1)
// parsers/bin_fst/fst_header.rs:71-137, vector_fst/serializable_fst.rs:45-168, utils_parsing.rs:10-44
struct Reader {
  const uint8_t* p;
  size_t n, off = 0;
  bool ok = true;
  template <class T>
  T get() {
    T v{};
    if (off + sizeof(T) > n) {
      ok = false;
      return v;
    }
    std::memcpy(&v, p + off, sizeof(T));
    off += sizeof(T);
    return v;
  }
  std::string str() {
    int32_t len = get<int32_t>();
    if (!ok || len < 0 || off + (size_t)len > n) {
      ok = false;
      return {};
    }
    std::string s((const char*)p + off, (size_t)len);
    off += (size_t)len;
    return s;
  }
};
// symbol table skip: parsers/bin_symt/nom_parser.rs (magic i32, name, available_key i64, size i64, {symbol,key}*)
bool skip_symt(Reader& r) {
  int32_t magic = r.get<int32_t>();
  if (!r.ok || magic != 2125658996) return false;
  r.str();
  r.get<int64_t>();
  int64_t num = r.get<int64_t>();
  for (int64_t i = 0; i < num && r.ok; ++i) {
    r.str();
    r.get<int64_t>();
  }
  return r.ok;
}

Fst* load_vector_fst(const uint8_t* data, size_t len) {
  Reader r{data, len};
  int32_t magic = r.get<int32_t>();
  if (!r.ok || magic != 2125659606) {
    t_err = "bad magic number";
    return nullptr;
  }
  std::string fst_type = r.str(), arc_type = r.str();
  const bool is_const = fst_type == "const";
  if (!r.ok || (fst_type != "vector" && !is_const) || arc_type != "standard") {
    t_err = "expected fst_type=vector|const arc_type=standard, got '" + fst_type + "'/'" + arc_type + "'";
    return nullptr;
  }
  int32_t version = r.get<int32_t>();
  if (!r.ok || version < (is_const ? 1 : 2)) {  // VECTOR_MIN_FILE_VERSION = 2, CONST_MIN_FILE_VERSION = 1
    t_err = "unsupported fst version";
    return nullptr;
  }
  uint32_t flags = r.get<uint32_t>();
  uint64_t props = r.get<uint64_t>();
  int64_t start = r.get<int64_t>();
  int64_t num_states = r.get<int64_t>();
  int64_t num_trs_hdr = r.get<int64_t>();  // ignored by the vector parser (serializable_fst.rs:157), used by const
  if (!r.ok || (flags & ~7u)) {
    t_err = "bad header";
    return nullptr;
  }
  if ((flags & 1u) && !skip_symt(r)) {
    t_err = "bad input symbol table";
    return nullptr;
  }
  if ((flags & 2u) && !skip_symt(r)) {
    t_err = "bad output symbol table";
    return nullptr;
  }
  auto fst = std::make_unique<Fst>();
  fst->states.resize((size_t)num_states);
  if (is_const) {  // parse_const_fst: const_fst/serializable_fst.rs:176-237 (aligned when version == 1)
    const bool aligned = version == 1;
    auto align16 = [&]() {
      if (aligned && (r.off % 16) != 0) r.off += 16 - (r.off % 16);
    };
    if (num_states > 0) align16();
    std::vector<uint32_t> pos((size_t)num_states), ntrs((size_t)num_states);
    for (int64_t s = 0; s < num_states; ++s) {
      State& st = fst->states[(size_t)s];
      float fw = r.get<float>();
      pos[(size_t)s] = (uint32_t)r.get<int32_t>();
      ntrs[(size_t)s] = (uint32_t)r.get<int32_t>();
      st.niepsilons = (size_t)r.get<int32_t>();
      st.noepsilons = (size_t)r.get<int32_t>();
      float saved = t_delta;
      t_delta = KDELTA;
      if (!weq(fw, INF)) {
        st.has_final = true;
        st.final_w = fw;
      }
      t_delta = saved;
    }
    if (num_trs_hdr > 0) align16();
    std::vector<Tr> trs((size_t)std::max<int64_t>(num_trs_hdr, 0));
    for (auto& tr : trs) {
      tr.ilabel = (uint32_t)r.get<int32_t>();
      tr.olabel = (uint32_t)r.get<int32_t>();
      tr.weight = r.get<float>();
      tr.nextstate = (uint32_t)r.get<int32_t>();
    }
    if (!r.ok) {
      t_err = "Error while parsing binary ConstFst";
      return nullptr;
    }
    for (int64_t s = 0; s < num_states; ++s) {
      if ((uint64_t)pos[(size_t)s] + ntrs[(size_t)s] > trs.size()) {
        t_err = "Error while parsing binary ConstFst";
        return nullptr;
      }
      fst->states[(size_t)s].trs.assign(trs.begin() + pos[(size_t)s], trs.begin() + pos[(size_t)s] + ntrs[(size_t)s]);
    }
    fst->has_start = start != -1;
    fst->start = start == -1 ? 0u : (uint32_t)start;
    fst->properties = props & P::ALL;
    return fst.release();
  }
  for (int64_t s = 0; s < num_states; ++s) {
    State& st = fst->states[(size_t)s];
    float fw = r.get<float>();
    int64_t ntrs = r.get<int64_t>();
    if (!r.ok || ntrs < 0) {
      t_err = "truncated state";
      return nullptr;
    }
    // parse_final_weight: utils_parsing.rs:18-26 (approximate != zero)
    {
      float saved = t_delta;
      t_delta = KDELTA;
      if (!weq(fw, INF)) {
        st.has_final = true;
        st.final_w = fw;
      }
      t_delta = saved;
    }
    st.trs.resize((size_t)ntrs);
    for (int64_t i = 0; i < ntrs; ++i) {
      Tr& tr = st.trs[(size_t)i];
      tr.ilabel = (uint32_t)r.get<int32_t>();
      tr.olabel = (uint32_t)r.get<int32_t>();
      tr.weight = r.get<float>();
      tr.nextstate = (uint32_t)r.get<int32_t>();
      if (tr.ilabel == EPS_LABEL) st.niepsilons++;
      if (tr.olabel == EPS_LABEL) st.noepsilons++;
    }
    if (!r.ok) {
      t_err = "truncated arcs";
      return nullptr;
    }
  }
  fst->has_start = start != -1;
  fst->start = start == -1 ? 0u : (uint32_t)start;
  fst->properties = props & P::ALL;  // from_bits_truncate
  return fst.release();
}

size_t store_vector_fst(const Fst& f, uint8_t* out, size_t cap) {
  std::vector<uint8_t> buf;
  auto put = [&](const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    buf.insert(buf.end(), b, b + n);
  };
  auto put_i32 = [&](int32_t v) { put(&v, 4); };
  auto put_i64 = [&](int64_t v) { put(&v, 8); };
  auto put_str = [&](const char* s) {
    put_i32((int32_t)std::strlen(s));
    put(s, std::strlen(s));
  };
  int64_t num_trs = 0;
  for (const State& st : f.states) num_trs += (int64_t)st.trs.size();
  put_i32(2125659606);
  put_str("vector");
  put_str("standard");
  put_i32(2);
  uint32_t flags = 0;
  put(&flags, 4);
  uint64_t props = f.properties | P::STATIC_EXPANDED_MUTABLE;
  put(&props, 8);
  put_i64(f.has_start ? (int64_t)f.start : -1);
  put_i64((int64_t)f.states.size());
  put_i64(num_trs);
  for (const State& st : f.states) {
    float fw = st.has_final ? st.final_w : INF;
    put(&fw, 4);
    put_i64((int64_t)st.trs.size());
    for (const Tr& tr : st.trs) {
      put_i32((int32_t)tr.ilabel);
      put_i32((int32_t)tr.olabel);
      put(&tr.weight, 4);
      put_i32((int32_t)tr.nextstate);
    }
  }
  if (out && cap >= buf.size()) std::memcpy(out, buf.data(), buf.size());
  return buf.size();
}

// ConstFst::store (const_fst/serializable_fst.rs:41-89): version 2 (unaligned), static property EXPANDED,
// per state {final f32, pos i32, ntrs i32, niepsilons i32, noepsilons i32}, then all arcs.
size_t store_const_fst(const Fst& f, uint8_t* out, size_t cap) {
  std::vector<uint8_t> buf;
  auto put = [&](const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    buf.insert(buf.end(), b, b + n);
  };
  auto put_i32 = [&](int32_t v) { put(&v, 4); };
  auto put_i64 = [&](int64_t v) { put(&v, 8); };
  auto put_str = [&](const char* s) {
    put_i32((int32_t)std::strlen(s));
    put(s, std::strlen(s));
  };
  int64_t num_trs = 0;
  for (const State& st : f.states) num_trs += (int64_t)st.trs.size();
  put_i32(2125659606);
  put_str("const");
  put_str("standard");
  put_i32(2);
  uint32_t flags = 0;
  put(&flags, 4);
  uint64_t props = f.properties | 0x1ull;  // ConstFst::static_properties() = EXPANDED (data_structure.rs:33-35)
  put(&props, 8);
  put_i64(f.has_start ? (int64_t)f.start : -1);
  put_i64((int64_t)f.states.size());
  put_i64(num_trs);
  int32_t pos = 0;
  for (const State& st : f.states) {
    float fw = st.has_final ? st.final_w : INF;
    put(&fw, 4);
    put_i32(pos);
    put_i32((int32_t)st.trs.size());
    put_i32((int32_t)st.niepsilons);
    put_i32((int32_t)st.noepsilons);
    pos += (int32_t)st.trs.size();
  }
  for (const State& st : f.states)
    for (const Tr& tr : st.trs) {
      put_i32((int32_t)tr.ilabel);
      put_i32((int32_t)tr.olabel);
      put(&tr.weight, 4);
      put_i32((int32_t)tr.nextstate);
    }
  if (out && cap >= buf.size()) std::memcpy(out, buf.data(), buf.size());
  return buf.size();
}

// ---------------------------------------------------------------- helpers for invariants
float brute_rec(const Fst& f, uint32_t s, float acc, uint32_t depth, uint32_t max_len) {
  float best = INF;
  const State& st = f.states[s];
  if (st.has_final) best = wplus(best, wtimes(acc, st.final_w));
  if (depth == max_len) return best;
  for (const Tr& tr : st.trs) best = wplus(best, brute_rec(f, tr.nextstate, wtimes(acc, tr.weight), depth + 1, max_len));
  return best;
}

bool path_rec(const Fst& f, uint32_t s, const std::vector<Tr>& labels, size_t i, float acc, float target_final,
              float* w_out) {
  if (i == labels.size()) {
    const State& st = f.states[s];
    if (!st.has_final) return false;
    (void)target_final;
    *w_out = wtimes(acc, st.final_w);
    return true;
  }
  for (const Tr& tr : f.states[s].trs) {
    if (tr.ilabel == labels[i].ilabel && tr.olabel == labels[i].olabel && std::fabs(tr.weight - labels[i].weight) <= KDELTA) {
      if (path_rec(f, tr.nextstate, labels, i + 1, wtimes(acc, tr.weight), target_final, w_out)) return true;
    }
  }
  return false;
}

// ================================================================ look-ahead composition (A12 / N1)
// Restates the configuration the reference wires in rustfst-cli/src/cmds/compose.rs:77-181 and in its golden test
// rustfst/src/tests_openfst/algorithms/compose.rs:118-254:
//   fst1 -> MatcherFst::new_with_relabeling (LabelReachable on OUTPUT labels, both FSTs relabelled and re-sorted),
//   M1 = LabelLookAheadMatcher<SortedMatcher> (flags OUTPUT_LOOKAHEAD_MATCHER | LOOKAHEAD_WEIGHT | LOOKAHEAD_PREFIX |
//        LOOKAHEAD_EPSILONS | LOOKAHEAD_NON_EPSILON_PREFIX), M2 = SortedMatcher,
//   filter = PushLabels(PushWeights(LookAhead(AltSequence))) with SMatchOutput, result = dyn_fst.compute() (no connect).
// PIN STATUS: IntervalSet is pinned on the reference's unit tests (interval_set.rs:208-275); everything else in this
// section is UNPINNED (the reference's "lookahead" goldens are generated by OpenFST at test time) and is checked
// through invariants only (same weighted relation as the plain composition, reachability against brute force).
namespace la {

// IntInterval / IntervalSet — compose/interval_set.rs:8-190
struct IntInterval {
  size_t begin, end;
};
inline bool interval_less(const IntInterval& a, const IntInterval& b) {  // Ord :29-44: begin ascending, then end DESCENDING
  if (a.begin != b.begin) return a.begin < b.begin;
  return a.end > b.end;
}
struct IntervalSet {
  std::vector<IntInterval> iv;
  size_t count = 0;
  size_t len() const { return iv.size(); }
  void unite(const IntervalSet& o) { iv.insert(iv.end(), o.iv.begin(), o.iv.end()); }  // union :133-135 (not normalized)
  bool member(size_t value) const {  // :138-145 (requires normalized)
    const IntInterval x{value, value};
    size_t lo = 0, hi = iv.size();
    while (lo < hi) {
      size_t mid = lo + (hi - lo) / 2;
      if (interval_less(iv[mid], x))
        lo = mid + 1;
      else
        hi = mid;
    }
    if (lo == 0) return false;
    return iv[lo - 1].end > value;
  }
  bool normalize() {  // :156-190; false = an empty interval was met (the reference loops forever there)
    std::stable_sort(iv.begin(), iv.end(), interval_less);
    const size_t n = iv.size();
    std::vector<bool> keep(n, false);
    size_t cnt = 0, i = 0;
    while (i < n) {
      IntInterval& inti = iv[i];
      const size_t inti_index = i;
      if (inti.begin == inti.end) return false;
      for (size_t j = inti_index + 1; j < n; ++j) {
        const IntInterval& intj = iv[j];
        if (intj.begin > inti.end) break;
        if (intj.end > inti.end) inti.end = intj.end;
        i += 1;
      }
      cnt += inti.end - inti.begin;
      keep[inti_index] = true;
      i += 1;
    }
    size_t w = 0;
    for (size_t k = 0; k < n; ++k)
      if (keep[k]) iv[w++] = iv[k];
    iv.resize(w);
    count = cnt;
    return true;
  }
};

constexpr size_t UNASSIGNED = std::numeric_limits<size_t>::max();

// IntervalReachVisitor — compose/interval_reach_visitor.rs:10-96 (fresh visitor: index starts at 1)
struct IntervalReachVisitor {
  const Fst* fst;
  std::vector<IntervalSet> isets;
  std::vector<size_t> state2index;
  size_t index = 1;
  bool failed = false;
  explicit IntervalReachVisitor(const Fst& f) : fst(&f) {}
  void init_visit(const Fst&) {}
  bool init_state(uint32_t s, uint32_t) {  // :36-63
    while (isets.size() <= s) isets.emplace_back();
    while (state2index.size() <= s) state2index.push_back(UNASSIGNED);
    const State& st = fst->states[s];
    if (st.has_final && !wis_zero(st.final_w)) {
      isets[s].iv.push_back(IntInterval{index, index + 1});
      state2index[s] = index;
      index += 1;
    }
    return true;
  }
  bool tree_tr(uint32_t, const Tr&) { return true; }
  bool back_tr(uint32_t, const Tr&) {  // :71-73 panics
    failed = true;
    t_err = "IntervalReachVisitor: Cyclic input";
    return false;
  }
  bool forward_or_cross_tr(uint32_t s, const Tr& tr) {  // :76-79
    isets[s].unite(isets[tr.nextstate]);
    return true;
  }
  void finish_state(uint32_t s, bool has_parent, uint32_t parent) {  // :83-95
    const State& st = fst->states[s];
    if (st.has_final && !wis_zero(st.final_w)) isets[s].iv[0].end = index;
    if (!isets[s].normalize()) {
      failed = true;
      t_err = "IntervalSet::normalize: empty interval";
    }
    if (has_parent) isets[parent].unite(isets[s]);
  }
  void finish_visit() {}
};

// detects cycles the way compute_and_update_properties(ACYCLIC) does (a back arc, self loops included)
struct CycleVisitor {
  bool cyclic = false;
  void init_visit(const Fst&) {}
  bool init_state(uint32_t, uint32_t) { return true; }
  bool tree_tr(uint32_t, const Tr&) { return true; }
  bool back_tr(uint32_t, const Tr&) {
    cyclic = true;
    return true;
  }
  bool forward_or_cross_tr(uint32_t, const Tr&) { return true; }
  void finish_state(uint32_t, bool, uint32_t) {}
  void finish_visit() {}
};

// StateReachable — compose/state_reachable.rs:20-87
struct StateReachable {
  std::vector<IntervalSet> isets;
  std::vector<size_t> state2index;
};
bool state_reachable_acyclic(const Fst& fst, StateReachable& out) {  // :69-76
  IntervalReachVisitor v(fst);
  dfs_visit(fst, v, false);
  if (v.failed) return false;
  out.isets = std::move(v.isets);
  out.state2index = std::move(v.state2index);
  return true;
}
// condense — algorithms/condense.rs:15-55
void condense(const Fst& ifst, std::vector<int32_t>& scc, Fst& ofst) {
  SccVisitor visitor(ifst, true);
  dfs_visit(ifst, visitor, false);
  scc = visitor.scc;
  ofst = Fst();
  if (scc.empty()) return;
  int32_t mx = *std::max_element(scc.begin(), scc.end());
  ofst.add_states((size_t)mx + 1);
  for (size_t s = 0; s < scc.size(); ++s) {
    const uint32_t c = (uint32_t)scc[s];
    if (ifst.has_start && s == ifst.start) {
      ofst.has_start = true;
      ofst.start = c;
    }
    const State& st = ifst.states[s];
    if (st.has_final) {
      State& oc = ofst.states[c];
      oc.final_w = oc.has_final ? wplus(oc.final_w, st.final_w) : st.final_w;
      oc.has_final = true;
    }
    for (const Tr& tr : st.trs) {
      const uint32_t nextc = (uint32_t)scc[tr.nextstate];
      if (nextc != c) {
        Tr t = tr;
        t.nextstate = nextc;
        ofst.states[c].trs.push_back(t);
      }
    }
  }
}
bool state_reachable(const Fst& fst, StateReachable& out) {  // new :27-35, new_cyclic :37-67
  CycleVisitor cv;
  dfs_visit(fst, cv, false);
  if (!cv.cyclic) return state_reachable_acyclic(fst, out);
  std::vector<int32_t> scc;
  Fst cfst;
  condense(fst, scc, cfst);
  StateReachable reachable;
  if (!state_reachable_acyclic(cfst, reachable)) return false;
  std::vector<size_t> nscc;
  for (int32_t c : scc) {
    while ((size_t)c >= nscc.size()) nscc.push_back(0);
    nscc[c] += 1;
  }
  out.state2index.assign(scc.size(), UNASSIGNED);
  out.isets.assign(scc.size(), IntervalSet());
  for (size_t s = 0; s < scc.size(); ++s) {
    const size_t c = (size_t)scc[s];
    out.isets[s] = reachable.isets[c];
    out.state2index[s] = reachable.state2index[c];
    if (cfst.states[c].has_final && nscc[c] > 1) {
      t_err = "StateReachable: Final state contained in a cycle";
      return false;
    }
  }
  return true;
}

// LabelReachableData / LabelReachable — compose/label_reachable.rs:16-403
struct LabelReachableData {
  bool reach_input = false;
  uint32_t final_label = NO_LABEL;
  std::unordered_map<uint32_t, uint32_t> label2index;
  std::vector<IntervalSet> interval_sets;

  uint32_t relabel(uint32_t label) {  // :52-61
    if (label == EPS_LABEL) return EPS_LABEL;
    const size_t n = label2index.size();
    auto it = label2index.find(label);
    if (it != label2index.end()) return it->second;
    label2index.emplace(label, (uint32_t)n + 1);
    return (uint32_t)n + 1;
  }
};

// update_properties_labels + keep_only_relevant_properties — trs_iter_mut.rs:241-302 (set_arc_properties() is empty)
uint64_t new_properties_labels(uint64_t p, uint32_t oi, uint32_t oo, uint32_t ni, uint32_t no) {
  if (oi != oo) p &= ~P::NOT_ACCEPTOR;
  if (oi == EPS_LABEL) {
    p &= ~P::I_EPSILONS;
    if (oo == EPS_LABEL) p &= ~P::EPSILONS;
  }
  if (oo == EPS_LABEL) p &= ~P::O_EPSILONS;
  if (ni != no) {
    p |= P::NOT_ACCEPTOR;
    p &= ~P::ACCEPTOR;
  }
  if (ni == EPS_LABEL) {
    p |= P::I_EPSILONS;
    p &= ~P::NO_I_EPSILONS;
    if (no == EPS_LABEL) {
      p |= P::EPSILONS;
      p &= ~P::NO_EPSILONS;
    }
  }
  if (no == EPS_LABEL) {
    p |= P::O_EPSILONS;
    p &= ~P::NO_O_EPSILONS;
  }
  p &= P::ACCEPTOR | P::NOT_ACCEPTOR | P::EPSILONS | P::NO_EPSILONS | P::I_EPSILONS | P::NO_I_EPSILONS | P::O_EPSILONS |
       P::NO_O_EPSILONS | P::WEIGHTED | P::UNWEIGHTED;
  return p;
}

// LabelReachableData::relabel_fst — label_reachable.rs:63-93
// (relabel maps epsilon to epsilon and nothing else to it: the per-state epsilon counters stay as they are)
void relabel_fst(LabelReachableData& data, Fst& fst, bool relabel_input) {
  for (State& st : fst.states) {
    for (Tr& tr : st.trs) {
      if (relabel_input) {
        const uint32_t nl = data.relabel(tr.ilabel);
        fst.properties = new_properties_labels(fst.properties, tr.ilabel, tr.olabel, nl, tr.olabel);
        tr.ilabel = nl;
      } else {
        const uint32_t nl = data.relabel(tr.olabel);
        fst.properties = new_properties_labels(fst.properties, tr.ilabel, tr.olabel, tr.ilabel, nl);
        tr.olabel = nl;
      }
    }
  }
  oracle_fst_tr_sort(&fst, relabel_input ? 0 : 1);
}

// LabelReachable::transform_fst :172-248 + find_intervals :250-273 = compute_data :135-150
bool compute_data(const Fst& ifst, bool reach_input, LabelReachableData& data) {
  Fst fst = ifst;
  data = LabelReachableData();
  data.reach_input = reach_input;
  std::unordered_map<uint32_t, uint32_t> label2state;
  const uint32_t ins = (uint32_t)fst.num_states();
  std::vector<size_t> indeg(ins, 0);
  uint32_t ons = ins;
  auto label_state = [&](uint32_t label) {
    auto it = label2state.find(label);
    if (it != label2state.end()) return it->second;
    const uint32_t v = ons;
    label2state.emplace(label, v);
    indeg.push_back(0);
    ons += 1;
    return v;
  };
  for (uint32_t s = 0; s < ins; ++s) {
    State& st = fst.states[s];
    for (Tr& tr : st.trs) {
      const uint32_t label = reach_input ? tr.ilabel : tr.olabel;
      const uint32_t nextstate = label != EPS_LABEL ? label_state(label) : tr.nextstate;
      indeg[nextstate] += 1;
      tr.nextstate = nextstate;
    }
    if (st.has_final && !wis_zero(st.final_w)) {
      const uint32_t nextstate = label_state(NO_LABEL);
      st.trs.push_back(Tr{NO_LABEL, NO_LABEL, st.final_w, nextstate});
      indeg[nextstate] += 1;
      st.has_final = false;
      st.final_w = INF;
    }
  }
  while (fst.num_states() < (size_t)ons) {  // new final (label) states
    fst.states.emplace_back();
    fst.states.back().has_final = true;
    fst.states.back().final_w = 0.0f;
  }
  fst.states.emplace_back();  // super-initial state for all states with zero in-degree
  const uint32_t start = (uint32_t)fst.num_states() - 1;
  fst.has_start = true;
  fst.start = start;
  for (uint32_t s = 0; s < start; ++s)
    if (indeg[s] == 0) fst.states[start].trs.push_back(Tr{0, 0, 0.0f, s});

  StateReachable sr;
  if (!state_reachable(fst, sr)) return false;
  data.interval_sets = std::move(sr.isets);
  data.interval_sets.resize(ins);
  for (const auto& kv : label2state) {
    const size_t i = sr.state2index[kv.second];
    data.label2index[kv.first] = (uint32_t)i;
    if (kv.first == NO_LABEL) data.final_label = (uint32_t)i;
  }
  return true;
}

// LookAheadMatcherData — lookahead_matchers/mod.rs:27-69
struct LaData {
  float lookahead_weight = 0.0f;
  Tr prefix_tr{0, 0, 0.0f, NO_STATE_ID};
};

// LabelReachable::reach :312-373 with reach_fst_input = true (labels of the look-ahead FST are its ilabels)
bool reach(const LabelReachableData& data, uint32_t current_state, const std::vector<Tr>& trs, size_t aiter_begin,
           size_t aiter_end, bool compute_weight, size_t* rb, size_t* re, float* rw) {
  size_t reach_begin = UNASSIGNED, reach_end = UNASSIGNED;
  float reach_weight = INF;
  const IntervalSet& iset = data.interval_sets[current_state];
  auto lower_bound = [&](size_t lo, size_t hi, uint32_t match_label) {  // :375-402
    while (lo < hi) {
      size_t mid = lo + (hi - lo) / 2;
      if (trs[mid].ilabel < match_label)
        lo = mid + 1;
      else
        hi = mid;
    }
    return lo;
  };
  if (2 * (aiter_end - aiter_begin) < iset.len()) {
    uint32_t reach_label = NO_LABEL;
    for (size_t pos = aiter_begin; pos < aiter_end; ++pos) {
      const Tr& tr = trs[pos];
      const uint32_t label = tr.ilabel;
      // reach_label :287-296: epsilon is never reachable
      if (label == reach_label || (label != EPS_LABEL && iset.member(label))) {
        reach_label = label;
        if (reach_begin == UNASSIGNED) reach_begin = pos;
        reach_end = pos + 1;
        if (compute_weight) reach_weight = wplus(reach_weight, tr.weight);
      }
    }
  } else {
    size_t begin_low, end_low = aiter_begin;
    for (const IntInterval& interval : iset.iv) {
      begin_low = lower_bound(end_low, aiter_end, (uint32_t)interval.begin);
      end_low = lower_bound(begin_low, aiter_end, (uint32_t)interval.end);
      if (end_low - begin_low > 0) {
        if (reach_begin == UNASSIGNED) reach_begin = begin_low;
        reach_end = end_low;
        if (compute_weight)
          for (size_t i = begin_low; i < end_low; ++i) reach_weight = wplus(reach_weight, trs[i].weight);
      }
    }
  }
  if (reach_begin == UNASSIGNED) return false;
  *rb = reach_begin;
  *re = reach_end;
  *rw = reach_weight;
  return true;
}

// LabelLookAheadMatcher::lookahead_fst :154-213 (flags: LOOKAHEAD_WEIGHT and LOOKAHEAD_PREFIX set)
bool lookahead_fst(const LabelReachableData& data, uint32_t matcher_state, const Fst& lfst, uint32_t lfst_state, LaData* out) {
  LaData la;
  bool compute_weight = true;
  const bool compute_prefix = true;
  const State& ls = lfst.states[lfst_state];
  size_t rb = 0, re = 0;
  float rw = INF;
  const bool reach_tr = reach(data, matcher_state, ls.trs, 0, ls.trs.size(), compute_weight, &rb, &re, &rw);
  const bool reach_final =
      ls.has_final && !wis_zero(ls.final_w) && data.interval_sets[matcher_state].member(data.final_label);
  if (reach_tr) {
    if (compute_prefix && (re - rb) == 1 && !reach_final) {
      la.prefix_tr = ls.trs[rb];
      compute_weight = false;
    } else {
      la.lookahead_weight = rw;
    }
  }
  if (reach_final && compute_weight) {
    if (reach_tr)
      la.lookahead_weight = wplus(la.lookahead_weight, ls.final_w);
    else
      la.lookahead_weight = ls.final_w;
  }
  if (reach_tr || reach_final) {
    *out = la;
    return true;
  }
  return false;
}


// filter state of PushLabels(PushWeights(LookAhead(AltSequence))):
// PairFilterState<PairFilterState<IntegerFilterState, WeightFilterState>, IntegerFilterState>
struct FS {
  uint32_t fs1;     // AltSequence state
  float fweight;    // pushed weight (quantized)
  uint32_t flabel;  // pushed label, NO_LABEL = none
};
struct Tuple5 {
  uint32_t s1, s2;
  FS fs;
  // ComposeStateTuple equality: derived PartialEq; the weight compares within KDELTA in the reference while its hash
  // uses the exact bits, so tuples whose weights differ by less than KDELTA are merged or not depending on hash-bucket
  // collisions there.  Weights here are multiples of KDELTA after quantize; they are compared exactly.
  bool operator==(const Tuple5& o) const {
    uint32_t a, b;
    std::memcpy(&a, &fs.fweight, 4);
    std::memcpy(&b, &o.fs.fweight, 4);
    return s1 == o.s1 && s2 == o.s2 && fs.fs1 == o.fs.fs1 && a == b && fs.flabel == o.fs.flabel;
  }
};
struct Tuple5Hash {
  size_t operator()(const Tuple5& t) const {
    uint32_t wb;
    std::memcpy(&wb, &t.fs.fweight, 4);
    uint64_t h = ((uint64_t)t.s1 << 32) | t.s2;
    h ^= ((uint64_t)t.fs.fs1 + 0x9E3779B97F4A7C15ull) * 0xff51afd7ed558ccdull;
    h ^= ((uint64_t)wb << 32 | t.fs.flabel) * 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    return (size_t)h;
  }
};

struct LaComposeOp {
  const Fst& fst1;  // relabelled, olabel-sorted
  const Fst& fst2;  // relabelled, ilabel-sorted
  const LabelReachableData& data;
  uint64_t properties;
  std::unordered_map<Tuple5, uint32_t, Tuple5Hash> tuple_to_id;
  std::vector<Tuple5> id_to_tuple;

  // filter object state (rebuilt per compute_trs / compute_final_weight: compose_fst_op.rs:411,425)
  uint32_t c_s1 = 0, c_s2 = 0;
  FS c_fs{0, 0.0f, NO_LABEL};
  bool alleps2 = false, noeps2 = false;
  size_t ntrsa = 0;
  bool lookahead_tr = false;
  LaData la;

  LaComposeOp(const Fst& a, const Fst& b, const LabelReachableData& d) : fst1(a), fst2(b), data(d) {
    // PushLabels::properties(PushWeights::properties(LookAhead::properties(AltSequence::properties(cprops))))
    const uint64_t weight_invariant =
        P::ACCEPTOR | P::NOT_ACCEPTOR | P::I_DETERMINISTIC | P::NOT_I_DETERMINISTIC | P::O_DETERMINISTIC |
        P::NOT_O_DETERMINISTIC | P::EPSILONS | P::NO_EPSILONS | P::I_EPSILONS | P::NO_I_EPSILONS | P::O_EPSILONS |
        P::NO_O_EPSILONS | P::I_LABEL_SORTED | P::NOT_I_LABEL_SORTED | P::O_LABEL_SORTED | P::NOT_O_LABEL_SORTED |
        P::CYCLIC | P::ACYCLIC | P::INITIAL_CYCLIC | P::INITIAL_ACYCLIC | P::TOP_SORTED | P::NOT_TOP_SORTED |
        P::ACCESSIBLE | P::NOT_ACCESSIBLE | P::COACCESSIBLE | P::NOT_COACCESSIBLE | P::STRING | P::NOT_STRING;  // properties.rs:436-465
    const uint64_t o_label_invariant =
        P::I_DETERMINISTIC | P::NOT_I_DETERMINISTIC | P::I_EPSILONS | P::NO_I_EPSILONS | P::I_LABEL_SORTED |
        P::NOT_I_LABEL_SORTED | P::WEIGHTED | P::UNWEIGHTED | P::CYCLIC | P::ACYCLIC | P::INITIAL_CYCLIC |
        P::INITIAL_ACYCLIC | P::TOP_SORTED | P::NOT_TOP_SORTED | P::ACCESSIBLE | P::NOT_ACCESSIBLE | P::COACCESSIBLE |
        P::NOT_COACCESSIBLE | P::STRING | P::NOT_STRING | P::WEIGHTED_CYCLES | P::UNWEIGHTED_CYCLES;  // :409-432
    properties = P::compose_properties(a.properties, b.properties) & weight_invariant & o_label_invariant;
  }

  uint32_t find_id(const Tuple5& t) {
    auto it = tuple_to_id.find(t);
    if (it != tuple_to_id.end()) return it->second;
    const uint32_t n = (uint32_t)id_to_tuple.size();
    // the reference's PartialEq would call t equal to a tuple whose weight is one quantization step away (its Hash
    // would not); count how often such a neighbour exists so the deviation is measured, not only described
    ++t_la_tuples;
    if (!std::isinf(t.fs.fweight)) {
      for (float step : {-KDELTA, KDELTA}) {
        Tuple5 nb = t;
        nb.fs.fweight = t.fs.fweight + step;
        if (tuple_to_id.count(nb)) {
          ++t_la_adjacent;
          break;
        }
      }
    }
    id_to_tuple.push_back(t);
    tuple_to_id.emplace(t, n);
    return n;
  }

  // set_state: push_labels_compose_filter.rs:167-191 -> push_weights :139-142 -> lookahead :275-277 -> alt_sequence :143-158
  void set_state(uint32_t s1, uint32_t s2, const FS& fs) {
    c_s1 = s1;
    c_s2 = s2;
    c_fs = fs;
    const State& st2 = fst2.states[s2];
    alleps2 = st2.trs.size() == st2.niepsilons && !st2.has_final;
    noeps2 = st2.niepsilons == 0;
    ntrsa = fst1.states[s1].trs.size();  // lookahead_output(): num_trs(s1)
  }

  // AltSequenceComposeFilter::filter_tr :160-181
  uint32_t alt_sequence(const Tr& arc1, const Tr& arc2) const {
    if (arc2.ilabel == NO_LABEL) {
      if (alleps2) return NO_STATE_ID;
      return noeps2 ? 0u : 1u;
    } else if (arc1.olabel == NO_LABEL) {
      return c_fs.fs1 == 1 ? NO_STATE_ID : 0u;
    } else if (arc1.olabel == EPS_LABEL) {
      return NO_STATE_ID;
    }
    return 0u;
  }
  // LookAheadComposeFilter::filter_tr :279-290 + lookahead_filter_tr :191-228 (lookahead_output() = true, Fst2Matcher1)
  uint32_t lookahead_filter(Tr& arc1, Tr& arc2) {
    lookahead_tr = false;
    const uint32_t fs = alt_sequence(arc1, arc2);
    if (fs == NO_STATE_ID) return NO_STATE_ID;
    const uint32_t labela = arc1.olabel;
    if (labela != EPS_LABEL) return fs;  // LOOKAHEAD_NON_EPSILONS is not set
    lookahead_tr = true;                 // LOOKAHEAD_EPSILONS is set
    if (!lookahead_fst(data, arc1.nextstate, fst2, arc2.nextstate, &la)) return NO_STATE_ID;
    return fs;
  }
  // PushWeightsComposeFilter::filter_tr :144-176
  bool push_weights_filter(Tr& arc1, Tr& arc2, uint32_t* fs1, float* w) {
    const uint32_t f = lookahead_filter(arc1, arc2);
    if (f == NO_STATE_ID) return false;
    const float lweight = lookahead_tr ? la.lookahead_weight : 0.0f;
    const float fweight = c_fs.fweight;
    if (wis_zero(lweight)) return false;  // disallows zero() weight futures
    arc2.weight = wtimes(arc2.weight, lweight);
    arc2.weight -= fweight;  // divide_assign: tropical_weight.rs:128-131
    *fs1 = f;
    *w = quantize(lweight, KDELTA);
    return true;
  }
  // PushLabelsComposeFilter::filter_tr :193-222, pushed_label_filter_tr :285-336, push_label_filter_tr :339-400
  bool filter_tr(Tr& arc1, Tr& arc2, FS* out) {
    const uint32_t flabel = c_fs.flabel;
    if (flabel != NO_LABEL) {  // consumes an already pushed label
      const uint32_t labelb = arc2.ilabel;
      if (labelb != NO_LABEL) return false;
      if (arc1.olabel == flabel) {
        arc1.olabel = EPS_LABEL;
        *out = FS{0u, 0.0f, NO_LABEL};  // self.start()
        return true;
      }
      if (arc1.olabel == EPS_LABEL) {
        // ntrsa == 1 || matcher1.lookahead_label(arca.nextstate, flabel) (label_lookahead_matcher.rs:215-224)
        if (ntrsa == 1 || data.interval_sets[arc1.nextstate].member(flabel)) {
          *out = c_fs;
          return true;
        }
        return false;
      }
      return false;
    }
    uint32_t fs1;
    float w;
    if (!push_weights_filter(arc1, arc2, &fs1, &w)) return false;
    if (!lookahead_tr) {
      *out = FS{fs1, w, NO_LABEL};
      return true;
    }
    // pushes a label forward when possible
    const uint32_t labelb = arc2.olabel;
    if (labelb != EPS_LABEL) {
      *out = FS{fs1, w, NO_LABEL};
      return true;
    }
    // (labela == EPS here: the look-ahead only ran for an epsilon on fst1's output side; LOOKAHEAD_NON_EPSILON_PREFIX is set)
    if (arc1.olabel != EPS_LABEL) {
      *out = FS{fs1, w, NO_LABEL};
      return true;
    }
    if (la.prefix_tr.nextstate != NO_STATE_ID) {  // default_lookahead_prefix, lookahead_matchers/mod.rs:60-68
      const Tr& larc = la.prefix_tr;
      arc1.olabel = larc.ilabel;
      arc2.ilabel = larc.ilabel;
      arc2.olabel = larc.olabel;
      arc2.weight = wtimes(arc2.weight, larc.weight);
      arc2.nextstate = larc.nextstate;
      *out = FS{fs1, w, arc1.olabel};
      return true;
    }
    *out = FS{fs1, w, NO_LABEL};
    return true;
  }

  // items a matcher yields for (state, label): 0 = EpsLoop, else pointer to a real arc
  struct Item {
    bool loop;
    const Tr* tr;
  };
  void sorted_items(const std::vector<Tr>& trs, uint32_t label, bool by_ilabel, std::vector<Item>& out) const {
    MatcherIter it(trs, label, by_ilabel);
    const Tr* real = nullptr;
    for (;;) {
      int k = it.next(&real);
      if (k == 0) break;
      out.push_back(k == 1 ? Item{true, nullptr} : Item{false, real});
    }
  }
  // MultiEpsMatcher::iter + IteratorMultiEpsMatcher::next — matchers/multi_eps_matcher.rs:64-110,160-210.  The set of
  // multi-epsilon labels is {flabel} or empty (push_labels_compose_filter.rs:183-189); matcher1 (fst1, by olabel) has
  // MULTI_EPS_LIST, matcher2 (fst2, by ilabel) MULTI_EPS_LOOP (:115-137 with lookahead_output() = true).
  void multi_eps_items(bool side2, uint32_t state, uint32_t label, std::vector<Item>& out) const {
    const std::vector<Tr>& trs = side2 ? fst2.states[state].trs : fst1.states[state].trs;
    const bool by_ilabel = side2;
    const uint32_t flabel = c_fs.flabel;
    if (label == EPS_LABEL) {
      sorted_items(trs, EPS_LABEL, by_ilabel, out);
    } else if (label == NO_LABEL) {
      if (!side2 && flabel != NO_LABEL) {  // MULTI_EPS_LIST: arcs carrying the multi-epsilon label, then the epsilon arcs
        sorted_items(trs, flabel, by_ilabel, out);
        sorted_items(trs, NO_LABEL, by_ilabel, out);
      } else {
        sorted_items(trs, NO_LABEL, by_ilabel, out);
      }
    } else if (side2 && flabel != NO_LABEL && label == flabel) {  // MULTI_EPS_LOOP: the label behaves like an epsilon loop
      out.push_back(Item{true, nullptr});
    } else {
      sorted_items(trs, label, by_ilabel, out);
    }
  }

  Tr add_tr(Tr arc1, const Tr& arc2, const FS& fs) {  // compose_fst_op.rs:267-285
    Tuple5 t{arc1.nextstate, arc2.nextstate, fs};
    arc1.weight = wtimes(arc1.weight, arc2.weight);
    return Tr{arc1.ilabel, arc2.olabel, arc1.weight, find_id(t)};
  }
  void match_tr(uint32_t sa, const Tr& tr, bool mi, std::vector<Tr>& trs) {  // :287-353
    const uint32_t label = mi ? tr.olabel : tr.ilabel;
    std::vector<Item> items;
    multi_eps_items(/*side2=*/mi, sa, label, items);
    for (const Item& item : items) {
      Tr arca = item.loop ? (mi ? Tr{NO_LABEL, EPS_LABEL, 0.0f, sa} : Tr{EPS_LABEL, NO_LABEL, 0.0f, sa}) : *item.tr;
      Tr arcb = tr;
      FS fs;
      if (mi) {
        if (filter_tr(arcb, arca, &fs)) trs.push_back(add_tr(arcb, arca, fs));
      } else {
        if (filter_tr(arca, arcb, &fs)) trs.push_back(add_tr(arca, arcb, fs));
      }
    }
  }
  std::vector<Tr> compute_trs(uint32_t state) {  // :406-418, ordered_expand :221-265, match_input :199-219
    const Tuple5 tuple = id_to_tuple[state];
    set_state(tuple.s1, tuple.s2, tuple.fs);
    const bool mi = fst1.states[tuple.s1].trs.size() <= fst2.states[tuple.s2].trs.size();  // priorities = num_trs
    const uint32_t sa = mi ? tuple.s2 : tuple.s1, sb = mi ? tuple.s1 : tuple.s2;
    const Tr tr_loop = mi ? Tr{EPS_LABEL, NO_LABEL, 0.0f, sb} : Tr{NO_LABEL, EPS_LABEL, 0.0f, sb};
    std::vector<Tr> trs;
    match_tr(sa, tr_loop, mi, trs);
    const std::vector<Tr>& sb_trs = mi ? fst1.states[sb].trs : fst2.states[sb].trs;
    for (const Tr& tr : sb_trs) match_tr(sa, tr, mi, trs);
    return trs;
  }
  bool compute_final_weight(uint32_t state, float* out) {  // :420-449 + the filters' filter_final
    const Tuple5 tuple = id_to_tuple[state];
    const State& a = fst1.states[tuple.s1];
    if (!a.has_final) return false;
    const State& b = fst2.states[tuple.s2];
    if (!b.has_final) return false;
    float w1 = a.final_w, w2 = b.final_w;
    set_state(tuple.s1, tuple.s2, tuple.fs);
    if (!wis_zero(w1)) w1 -= tuple.fs.fweight;                              // push_weights :178-189
    if (!wis_zero(w1) && tuple.fs.flabel != NO_LABEL) w1 = INF;             // push_labels :224-238
    const float f = wtimes(w1, w2);
    if (wis_zero(f)) return false;
    *out = f;
    return true;
  }
};

// the whole pipeline of cmds/compose.rs:131-180; r1 / r2 receive the relabelled inputs
bool compose_lookahead(const Fst& in1, const Fst& in2, Fst& out, Fst& r1, Fst& r2, LabelReachableData& data) {
  r1 = in1;
  r2 = in2;
  // MatcherFst::new_with_relabeling(fst1, &mut fst2, true) — matcher_fst.rs:73-94: data for MatchOutput only (the
  // flags hold OUTPUT_LOOKAHEAD_MATCHER), init relabels fst1's olabels, relabel() fst2's ilabels
  if (!compute_data(r1, /*reach_input=*/false, data)) return false;
  relabel_fst(data, r1, /*relabel_input=*/false);
  relabel_fst(data, r2, /*relabel_input=*/true);
  oracle_fst_tr_sort(&r2, 0);  // cmds/compose.rs:151
  // matcher1 needs O_LABEL_SORTED on fst1 (SortedMatcher, MatchOutput), reach_init needs I_LABEL_SORTED on fst2
  // (label_reachable.rs:275-291): both hold after the sorts above.
  LaComposeOp op(r1, r2, data);
  out = Fst();
  if (r1.has_start && r2.has_start) {
    const uint32_t start_state = op.find_id(Tuple5{r1.start, r2.start, FS{0u, 0.0f, NO_LABEL}});
    out.add_states((size_t)start_state + 1);
    out.set_start(start_state);
    std::deque<uint32_t> queue;
    std::vector<bool> visited((size_t)start_state + 1, false);
    visited[start_state] = true;
    queue.push_back(start_state);
    while (!queue.empty()) {  // LazyFst::compute, lazy/lazy_fst.rs:226-269
      const uint32_t s = queue.front();
      queue.pop_front();
      std::vector<Tr> trs = op.compute_trs(s);
      for (const Tr& tr : trs) {
        if ((size_t)tr.nextstate >= visited.size()) visited.resize((size_t)tr.nextstate + 1, false);
        if (!visited[tr.nextstate]) {
          queue.push_back(tr.nextstate);
          visited[tr.nextstate] = true;
        }
        const size_t n = out.num_states();
        if ((size_t)tr.nextstate >= n) out.add_states((size_t)tr.nextstate - n + 1);
      }
      out.set_trs_unchecked(s, std::move(trs));
      float fw;
      if (op.compute_final_weight(s, &fw)) out.set_final(s, fw);
    }
    out.set_properties(op.properties);
  }
  return true;
}

}  // namespace la

}  // namespace

// ================================================================ C API
extern "C" {

const char* oracle_last_error(void) { return t_err.c_str(); }
const char* oracle_last_queue_kind(void) { return t_queue_kind; }
void oracle_last_lookahead_tuples(uint64_t* tuples, uint64_t* adjacent) {
  *tuples = t_la_tuples;
  *adjacent = t_la_adjacent;
}

oracle_fst* oracle_fst_new(void) { return new oracle_fst(); }
void oracle_fst_free(oracle_fst* f) { delete f; }
uint32_t oracle_fst_add_state(oracle_fst* f) { return f->add_state(); }
int oracle_fst_set_start(oracle_fst* f, uint32_t s) { return f->set_start(s) ? 0 : 1; }
int oracle_fst_set_final(oracle_fst* f, uint32_t s, float w) { return f->set_final(s, w) ? 0 : 1; }
int oracle_fst_add_tr(oracle_fst* f, uint32_t s, uint32_t il, uint32_t ol, float w, uint32_t ns) {
  return f->add_tr(s, Tr{il, ol, w, ns}) ? 0 : 1;
}

// tr_sort — algorithms/tr_sort.rs:50-62 (stable sort_by; sets the sorted bit, keeps arcsort-invariant bits)
void oracle_fst_tr_sort(oracle_fst* f, int by_olabel) {
  for (auto& st : f->states) {
    if (by_olabel)
      std::stable_sort(st.trs.begin(), st.trs.end(), [](const Tr& a, const Tr& b) { return a.olabel < b.olabel; });
    else
      std::stable_sort(st.trs.begin(), st.trs.end(), [](const Tr& a, const Tr& b) { return a.ilabel < b.ilabel; });
  }
  // tr_sort.rs: props & arcsort_properties() | (I|O)_LABEL_SORTED
  const uint64_t arcsort_mask = P::ALL & ~(P::I_LABEL_SORTED | P::NOT_I_LABEL_SORTED | P::O_LABEL_SORTED |
                                           P::NOT_O_LABEL_SORTED);
  const bool acceptor = (f->properties & P::ACCEPTOR) != 0;
  f->properties &= arcsort_mask;
  f->properties |= by_olabel ? P::O_LABEL_SORTED : P::I_LABEL_SORTED;
  if (acceptor) f->properties |= by_olabel ? P::I_LABEL_SORTED : P::O_LABEL_SORTED;  // tr_sort.rs:21-27,38-44
}

oracle_fst* oracle_fst_from_flat(uint32_t n_states, int64_t start, const uint32_t* offsets, const oracle_tr* arcs,
                                 const float* finals, uint64_t props) {
  auto* f = new oracle_fst();
  f->states.resize(n_states);
  for (uint32_t s = 0; s < n_states; ++s) {
    State& st = f->states[s];
    st.trs.assign(arcs + offsets[s], arcs + offsets[s + 1]);
    for (const Tr& tr : st.trs) {
      if (tr.ilabel == EPS_LABEL) st.niepsilons++;
      if (tr.olabel == EPS_LABEL) st.noepsilons++;
    }
    if (finals[s] != INF) {
      st.has_final = true;
      st.final_w = finals[s];
    }
  }
  f->has_start = start >= 0;
  f->start = start >= 0 ? (uint32_t)start : 0u;
  f->properties = props & P::ALL;
  return f;
}

void oracle_fst_info(const oracle_fst* f, uint32_t* n_states, uint64_t* n_arcs, int64_t* start, uint64_t* props) {
  if (n_states) *n_states = (uint32_t)f->states.size();
  if (n_arcs) {
    uint64_t n = 0;
    for (const State& st : f->states) n += st.trs.size();
    *n_arcs = n;
  }
  if (start) *start = f->has_start ? (int64_t)f->start : -1;
  if (props) *props = f->properties;
}

void oracle_fst_to_flat(const oracle_fst* f, uint32_t* offsets, oracle_tr* arcs, float* finals) {
  uint32_t off = 0;
  for (size_t s = 0; s < f->states.size(); ++s) {
    const State& st = f->states[s];
    offsets[s] = off;
    if (!st.trs.empty()) std::memcpy(arcs + off, st.trs.data(), st.trs.size() * sizeof(Tr));
    off += (uint32_t)st.trs.size();
    finals[s] = st.has_final ? st.final_w : INF;
  }
  offsets[f->states.size()] = off;
}

void oracle_fst_eps_counts(const oracle_fst* f, uint32_t* nieps, uint32_t* noeps) {
  for (size_t s = 0; s < f->states.size(); ++s) {
    nieps[s] = (uint32_t)f->states[s].niepsilons;
    noeps[s] = (uint32_t)f->states[s].noepsilons;
  }
}

oracle_fst* oracle_fst_load(const uint8_t* data, size_t len) { return load_vector_fst(data, len); }
size_t oracle_fst_store(const oracle_fst* f, uint8_t* out, size_t cap) { return store_vector_fst(*f, out, cap); }
size_t oracle_fst_store_const(const oracle_fst* f, uint8_t* out, size_t cap) { return store_const_fst(*f, out, cap); }

int oracle_compose(const oracle_fst* f1, const oracle_fst* f2, int connect, int eq_mode, oracle_fst** out) {
  DeltaGuard g(eq_mode);
  auto res = std::make_unique<oracle_fst>();
  if (!compose_impl(*f1, *f2, connect != 0, *res, nullptr)) return 1;
  *out = res.release();
  return 0;
}

int oracle_compose_filter(const oracle_fst* f1, const oracle_fst* f2, int connect, int eq_mode, int filter, oracle_fst** out) {
  DeltaGuard g(eq_mode);
  auto res = std::make_unique<oracle_fst>();
  if (!compose_impl(*f1, *f2, connect != 0, *res, nullptr, filter)) return 1;
  *out = res.release();
  return 0;
}

// look-ahead composition (A12): cmds/compose.rs:77-181
int oracle_compose_lookahead(const oracle_fst* f1, const oracle_fst* f2, oracle_fst** out, oracle_fst** relabeled1,
                             oracle_fst** relabeled2) {
  DeltaGuard g(ORACLE_EQ_REF_KDELTA);
  t_la_tuples = t_la_adjacent = 0;
  std::unique_ptr<oracle_fst> o(new oracle_fst()), r1(new oracle_fst()), r2(new oracle_fst());
  la::LabelReachableData data;
  if (!la::compose_lookahead(*f1, *f2, *o, *r1, *r2, data)) return 1;
  *out = o.release();
  if (relabeled1) *relabeled1 = r1.release();
  if (relabeled2) *relabeled2 = r2.release();
  return 0;
}

int64_t oracle_interval_set_normalize(uint64_t* pairs, size_t n, uint64_t* count) {
  la::IntervalSet s;
  for (size_t i = 0; i < n; ++i) s.iv.push_back(la::IntInterval{(size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]});
  if (!s.normalize()) return -1;
  for (size_t i = 0; i < s.iv.size(); ++i) {
    pairs[2 * i] = s.iv[i].begin;
    pairs[2 * i + 1] = s.iv[i].end;
  }
  if (count) *count = s.count;
  return (int64_t)s.iv.size();
}

int oracle_interval_set_member(const uint64_t* pairs, size_t n, uint64_t value) {
  la::IntervalSet s;
  for (size_t i = 0; i < n; ++i) s.iv.push_back(la::IntInterval{(size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]});
  return s.member((size_t)value) ? 1 : 0;
}

struct oracle_label_reachable {
  la::LabelReachableData data;
};
oracle_label_reachable* oracle_label_reachable_new(const oracle_fst* f, int reach_input) {
  DeltaGuard g(ORACLE_EQ_REF_KDELTA);
  std::unique_ptr<oracle_label_reachable> h(new oracle_label_reachable());
  if (!la::compute_data(*f, reach_input != 0, h->data)) return nullptr;
  return h.release();
}
void oracle_label_reachable_free(oracle_label_reachable* h) { delete h; }
uint32_t oracle_label_reachable_final_label(const oracle_label_reachable* h) { return h->data.final_label; }
size_t oracle_label_reachable_num_labels(const oracle_label_reachable* h) { return h->data.label2index.size(); }
void oracle_label_reachable_labels(const oracle_label_reachable* h, uint32_t* labels, uint32_t* indices) {
  std::vector<std::pair<uint32_t, uint32_t>> v(h->data.label2index.begin(), h->data.label2index.end());
  std::sort(v.begin(), v.end());
  for (size_t i = 0; i < v.size(); ++i) {
    labels[i] = v[i].first;
    indices[i] = v[i].second;
  }
}
size_t oracle_label_reachable_num_states(const oracle_label_reachable* h) { return h->data.interval_sets.size(); }
size_t oracle_label_reachable_num_intervals(const oracle_label_reachable* h, uint32_t state) {
  return h->data.interval_sets[state].iv.size();
}
void oracle_label_reachable_intervals(const oracle_label_reachable* h, uint32_t state, uint64_t* pairs) {
  const auto& iv = h->data.interval_sets[state].iv;
  for (size_t i = 0; i < iv.size(); ++i) {
    pairs[2 * i] = iv[i].begin;
    pairs[2 * i + 1] = iv[i].end;
  }
}

// project — algorithms/projection.rs:65-95 + project_properties (fst_properties/mutate_properties.rs:365-445); the per-arc
// label bookkeeping of set_{i,o}label_unchecked is overwritten by set_properties_with_mask(.., all_properties())
void oracle_fst_project(oracle_fst* f, int project_output) {
  const uint64_t in = f->properties;
  for (auto& st : f->states) {
    for (Tr& tr : st.trs) {
      if (project_output)
        tr.ilabel = tr.olabel;
      else
        tr.olabel = tr.ilabel;
    }
    if (project_output)
      st.niepsilons = st.noepsilons;
    else
      st.noepsilons = st.niepsilons;
  }
  uint64_t out = P::ACCEPTOR;
  out |= (P::WEIGHTED | P::UNWEIGHTED | P::WEIGHTED_CYCLES | P::UNWEIGHTED_CYCLES | P::CYCLIC | P::ACYCLIC | P::INITIAL_CYCLIC |
          P::INITIAL_ACYCLIC | P::TOP_SORTED | P::NOT_TOP_SORTED | P::ACCESSIBLE | P::NOT_ACCESSIBLE | P::COACCESSIBLE |
          P::NOT_COACCESSIBLE | P::STRING | P::NOT_STRING) & in;
  if (!project_output) {
    out |= (P::I_DETERMINISTIC | P::NOT_I_DETERMINISTIC | P::I_EPSILONS | P::NO_I_EPSILONS | P::I_LABEL_SORTED | P::NOT_I_LABEL_SORTED) & in;
    if (in & P::I_DETERMINISTIC) out |= P::O_DETERMINISTIC;
    if (in & P::NOT_I_DETERMINISTIC) out |= P::NOT_O_DETERMINISTIC;
    if (in & P::I_EPSILONS) out |= P::O_EPSILONS | P::EPSILONS;
    if (in & P::NO_I_EPSILONS) out |= P::NO_O_EPSILONS | P::NO_EPSILONS;
    if (in & P::I_LABEL_SORTED) out |= P::O_LABEL_SORTED;
    if (in & P::NOT_I_LABEL_SORTED) out |= P::NOT_O_LABEL_SORTED;
  } else {
    out |= (P::O_DETERMINISTIC | P::NOT_O_DETERMINISTIC | P::O_EPSILONS | P::NO_O_EPSILONS | P::O_LABEL_SORTED | P::NOT_O_LABEL_SORTED) & in;
    if (in & P::O_DETERMINISTIC) out |= P::I_DETERMINISTIC;
    if (in & P::NOT_O_DETERMINISTIC) out |= P::NOT_I_DETERMINISTIC;
    if (in & P::O_EPSILONS) out |= P::I_EPSILONS | P::EPSILONS;
    if (in & P::NO_O_EPSILONS) out |= P::NO_I_EPSILONS | P::NO_EPSILONS;
    if (in & P::O_LABEL_SORTED) out |= P::I_LABEL_SORTED;
    if (in & P::NOT_O_LABEL_SORTED) out |= P::NOT_I_LABEL_SORTED;
  }
  f->set_properties_with_mask(out, P::ALL);
}

// rm_epsilon (default config): algorithms/rm_epsilon/rm_epsilon_static.rs:50-163, in place
int oracle_rm_epsilon(oracle_fst* f) {
  DeltaGuard g(ORACLE_EQ_REF_KDELTA);
  return rm_epsilon_impl(*f) ? 0 : 1;
}

int oracle_connect(oracle_fst* f) {
  connect_impl(*f);
  return 0;
}

int oracle_shortest_path(const oracle_fst* f, int eq_mode, oracle_fst** out, float* distance, float* total_weight) {
  DeltaGuard g(eq_mode);
  std::vector<float> dist;
  std::vector<Parent> parent;
  bool has_fp = false;
  uint32_t fp = 0;
  t_queue_kind = "none";
  single_shortest_path(*f, dist, has_fp, fp, parent);
  auto res = std::make_unique<oracle_fst>();
  if (!backtrace(*f, has_fp, fp, parent, *res)) return 1;
  if (distance)
    for (size_t i = 0; i < f->states.size(); ++i) distance[i] = i < dist.size() ? dist[i] : INF;
  if (total_weight) *total_weight = has_fp ? wtimes(dist[fp], f->states[fp].final_w) : INF;
  *out = res.release();
  return 0;
}

int oracle_shortest_path_canonical(const oracle_fst* f, oracle_fst** out, float* distance, uint32_t* hops,
                                   float* total_weight, uint32_t* n_tied_choices) {
  Canon c;
  canonical_sssp(*f, c);
  auto res = std::make_unique<oracle_fst>();
  {
    DeltaGuard g(ORACLE_EQ_REF_KDELTA);  // property bits of the output follow the reference's is_one/is_zero
    if (!backtrace(*f, c.has_final, c.f_parent, c.parent, *res)) return 1;
  }
  if (distance) std::copy(c.d.begin(), c.d.end(), distance);
  if (hops) std::copy(c.h.begin(), c.h.end(), hops);
  if (total_weight) *total_weight = c.has_final ? c.total : INF;
  if (n_tied_choices) *n_tied_choices = canonical_count_ties(*f, c);
  *out = res.release();
  return 0;
}

int oracle_shortest_path_n(const oracle_fst* f, uint64_t nshortest, float delta, int eq_mode, oracle_fst** out) {
  DeltaGuard g(eq_mode);
  auto res = std::make_unique<oracle_fst>();
  if (nshortest == 0) {
    *out = res.release();
    return 0;
  }
  if (nshortest == 1) return oracle_shortest_path(f, eq_mode, out, nullptr, nullptr);
  if (!shortest_path_n_impl(*f, (size_t)nshortest, delta, *res)) return 1;
  *out = res.release();
  return 0;
}

// determinize_fsa::<DefaultCommonDivisor> (determinize_static.rs:41-53), what `determinize` does to an acceptor
// (:176-181) before it sets the property word (not restated: the reference's tests compare states and start only)
int oracle_determinize_fsa(const oracle_fst* f, float delta, int eq_mode, oracle_fst** out) {
  DeltaGuard g(eq_mode);
  auto res = std::make_unique<oracle_fst>();
  std::vector<float> no_dist, out_dist;
  if (!determinize_with_distance_impl(*f, no_dist, delta, *res, out_dist)) return 1;
  *out = res.release();
  return 0;
}

int oracle_shortest_path_n_unique(const oracle_fst* f, uint64_t nshortest, float delta, int eq_mode, oracle_fst** out) {
  DeltaGuard g(eq_mode);
  if (nshortest <= 1) return oracle_shortest_path_n(f, nshortest, delta, eq_mode, out);  // `unique` is not looked at
  auto res = std::make_unique<oracle_fst>();
  if (!shortest_path_n_unique_impl(*f, (size_t)nshortest, delta, *res)) return 1;
  *out = res.release();
  return 0;
}

uint64_t oracle_shortest_distance(const oracle_fst* f, float delta, float* distance, uint64_t cap) {
  DeltaGuard g(ORACLE_EQ_REF_KDELTA);
  std::vector<float> d = shortest_distance_impl(*f, delta);
  for (uint64_t i = 0; i < cap; ++i) distance[i] = i < d.size() ? d[i] : INF;
  return d.size();
}

int oracle_reverse(const oracle_fst* f, oracle_fst** out) {
  DeltaGuard g(ORACLE_EQ_REF_KDELTA);
  auto res = std::make_unique<oracle_fst>();
  reverse_impl(*f, *res);
  *out = res.release();
  return 0;
}

float oracle_bruteforce_min_weight(const oracle_fst* f, uint32_t max_len) {
  if (!f->has_start) return INF;
  DeltaGuard g(ORACLE_EQ_EXACT);
  return brute_rec(*f, f->start, 0.0f, 0, max_len);
}

int oracle_path_in_fst(const oracle_fst* path, const oracle_fst* f, float* weight_in_f) {
  if (!path->has_start || !f->has_start) return 0;
  std::vector<Tr> labels;
  uint32_t s = path->start;
  for (size_t guard = 0; guard <= path->states.size(); ++guard) {
    const State& st = path->states[s];
    if (st.trs.empty()) break;
    if (st.trs.size() != 1) return 0;
    labels.push_back(st.trs[0]);
    s = st.trs[0].nextstate;
  }
  if (!path->states[s].has_final) return 0;
  float w = INF;
  bool ok = path_rec(*f, f->start, labels, 0, 0.0f, path->states[s].final_w, &w);
  if (ok && weight_in_f) *weight_in_f = w;
  return ok ? 1 : 0;
}

int oracle_compose_shortest_path_batch(const oracle_fst* const* accs, size_t n, const oracle_fst* t, int n_threads,
                                       int eq_mode, oracle_fst** outs, uint64_t* composed_arcs_pre_trim,
                                       double* seconds) {
  if (n_threads < 1) n_threads = 1;
  std::atomic<size_t> next{0};
  std::atomic<int> failed{0};
  std::vector<uint64_t> arcs_per_thread((size_t)n_threads, 0);
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&](int tid) {
    DeltaGuard g(eq_mode);
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= n) break;
      oracle_fst composed;
      uint64_t na = 0;
      if (!compose_impl(*accs[i], *t, true, composed, &na)) {
        failed = 1;
        continue;
      }
      arcs_per_thread[(size_t)tid] += na;
      std::vector<float> dist;
      std::vector<Parent> parent;
      bool has_fp = false;
      uint32_t fp = 0;
      single_shortest_path(composed, dist, has_fp, fp, parent);
      auto res = std::make_unique<oracle_fst>();
      if (!backtrace(composed, has_fp, fp, parent, *res)) {
        failed = 1;
        continue;
      }
      if (outs)
        outs[i] = res.release();
    }
  };
  if (n_threads == 1) {
    worker(0);
  } else {
    std::vector<std::thread> th;
    for (int k = 0; k < n_threads; ++k) th.emplace_back(worker, k);
    for (auto& x : th) x.join();
  }
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  if (composed_arcs_pre_trim) {
    uint64_t tot = 0;
    for (uint64_t a : arcs_per_thread) tot += a;
    *composed_arcs_pre_trim = tot;
  }
  return failed.load();
}

}  // extern "C"
