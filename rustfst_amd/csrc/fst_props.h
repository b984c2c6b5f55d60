// fst_props.h — the 64-bit FstProperties word that crosses the C-ABI.
// Bit values: rustfst/src/fst_properties/properties.rs:22-103 (OpenFST-compatible).
// Update rules: rustfst/src/fst_properties/mutate_properties.rs (cited per function).
#pragma once
#include <cstdint>

#include "../../include/wfst.h"

namespace wfst::props {

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
constexpr uint64_t ALL = 0x0000ffffffff0000ull;  // trinary bits; EXPANDED|MUTABLE (0x3) are static (properties.rs:5-6)
constexpr uint64_t STATIC_BITS = 0x3;

// FstProperties::null_properties(): properties.rs:105-124 (VectorFst::new)
constexpr uint64_t NULL_PROPS = ACCEPTOR | I_DETERMINISTIC | O_DETERMINISTIC | NO_EPSILONS | NO_I_EPSILONS |
                                NO_O_EPSILONS | I_LABEL_SORTED | O_LABEL_SORTED | UNWEIGHTED | ACYCLIC |
                                INITIAL_ACYCLIC | TOP_SORTED | ACCESSIBLE | COACCESSIBLE | STRING | UNWEIGHTED_CYCLES;

constexpr uint64_t LABEL_INVARIANT_COMMON = 0;  // (unused; kept for readability of masks below)

// preserved-by masks: properties.rs:166-300
constexpr uint64_t SET_START_MASK = ACCEPTOR | NOT_ACCEPTOR | I_DETERMINISTIC | NOT_I_DETERMINISTIC | O_DETERMINISTIC |
                                    NOT_O_DETERMINISTIC | EPSILONS | NO_EPSILONS | I_EPSILONS | NO_I_EPSILONS |
                                    O_EPSILONS | NO_O_EPSILONS | I_LABEL_SORTED | NOT_I_LABEL_SORTED | O_LABEL_SORTED |
                                    NOT_O_LABEL_SORTED | WEIGHTED | UNWEIGHTED | CYCLIC | ACYCLIC | TOP_SORTED |
                                    NOT_TOP_SORTED | COACCESSIBLE | NOT_COACCESSIBLE | WEIGHTED_CYCLES |
                                    UNWEIGHTED_CYCLES;
constexpr uint64_t SET_FINAL_MASK = ACCEPTOR | NOT_ACCEPTOR | I_DETERMINISTIC | NOT_I_DETERMINISTIC | O_DETERMINISTIC |
                                    NOT_O_DETERMINISTIC | EPSILONS | NO_EPSILONS | I_EPSILONS | NO_I_EPSILONS |
                                    O_EPSILONS | NO_O_EPSILONS | I_LABEL_SORTED | NOT_I_LABEL_SORTED | O_LABEL_SORTED |
                                    NOT_O_LABEL_SORTED | CYCLIC | ACYCLIC | INITIAL_CYCLIC | INITIAL_ACYCLIC |
                                    TOP_SORTED | NOT_TOP_SORTED | ACCESSIBLE | NOT_ACCESSIBLE | WEIGHTED_CYCLES |
                                    UNWEIGHTED_CYCLES;
constexpr uint64_t ADD_STATE_MASK = ACCEPTOR | NOT_ACCEPTOR | I_DETERMINISTIC | NOT_I_DETERMINISTIC | O_DETERMINISTIC |
                                    NOT_O_DETERMINISTIC | EPSILONS | NO_EPSILONS | I_EPSILONS | NO_I_EPSILONS |
                                    O_EPSILONS | NO_O_EPSILONS | I_LABEL_SORTED | NOT_I_LABEL_SORTED | O_LABEL_SORTED |
                                    NOT_O_LABEL_SORTED | WEIGHTED | UNWEIGHTED | CYCLIC | ACYCLIC | INITIAL_CYCLIC |
                                    INITIAL_ACYCLIC | TOP_SORTED | NOT_TOP_SORTED | NOT_ACCESSIBLE | NOT_COACCESSIBLE |
                                    NOT_STRING | WEIGHTED_CYCLES | UNWEIGHTED_CYCLES;
constexpr uint64_t ADD_ARC_MASK = NOT_ACCEPTOR | NOT_I_DETERMINISTIC | NOT_O_DETERMINISTIC | EPSILONS | I_EPSILONS |
                                  O_EPSILONS | NOT_I_LABEL_SORTED | NOT_O_LABEL_SORTED | WEIGHTED | CYCLIC |
                                  INITIAL_CYCLIC | NOT_TOP_SORTED | ACCESSIBLE | COACCESSIBLE | WEIGHTED_CYCLES;
constexpr uint64_t DELETE_STATES_MASK = ACCEPTOR | I_DETERMINISTIC | O_DETERMINISTIC | NO_EPSILONS | NO_I_EPSILONS |
                                        NO_O_EPSILONS | I_LABEL_SORTED | O_LABEL_SORTED | UNWEIGHTED | ACYCLIC |
                                        INITIAL_ACYCLIC | TOP_SORTED | UNWEIGHTED_CYCLES;

// TropicalWeight::{is_zero,is_one} use the approximate == (semirings/semiring.rs:68-73,159-168); they
// only feed the WEIGHTED/UNWEIGHTED bits here, so the reference tolerance is kept for bit-parity.
constexpr float KDELTA = 1.0f / 1024.0f;
inline bool approx_eq(float a, float b) { return a <= b + KDELTA && b <= a + KDELTA; }
inline bool is_zero(float w) { return approx_eq(w, __builtin_huge_valf()); }
inline bool is_one(float w) { return approx_eq(w, 0.0f); }

inline uint64_t set_start(uint64_t in) {  // mutate_properties.rs:7-13
  uint64_t out = in & SET_START_MASK;
  if (in & ACYCLIC) out |= INITIAL_ACYCLIC;
  return out;
}
inline uint64_t set_final(uint64_t in, const float* old_w, const float* new_w) {  // :15-37
  uint64_t out = in;
  if (old_w && !is_zero(*old_w) && !is_one(*old_w)) out &= ~WEIGHTED;
  if (new_w && !is_zero(*new_w) && !is_one(*new_w)) {
    out |= WEIGHTED;
    out &= ~UNWEIGHTED;
  }
  return out & (SET_FINAL_MASK | WEIGHTED | UNWEIGHTED);
}
inline uint64_t add_state(uint64_t in) { return in & ADD_STATE_MASK; }  // :39-41
inline uint64_t add_tr(uint64_t in, uint32_t state, const wfst_tr& tr, const wfst_tr* prev) {  // :43-100
  uint64_t out = in;
  if (tr.ilabel != tr.olabel) out = (out | NOT_ACCEPTOR) & ~ACCEPTOR;
  if (tr.ilabel == WFST_EPS_LABEL) {
    out = (out | I_EPSILONS) & ~NO_I_EPSILONS;
    if (tr.olabel == WFST_EPS_LABEL) out = (out | EPSILONS) & ~NO_EPSILONS;
  }
  if (tr.olabel == WFST_EPS_LABEL) out = (out | O_EPSILONS) & ~NO_O_EPSILONS;
  if (prev) {
    if (prev->ilabel > tr.ilabel) out = (out | NOT_I_LABEL_SORTED) & ~I_LABEL_SORTED;
    if (prev->olabel > tr.olabel) out = (out | NOT_O_LABEL_SORTED) & ~O_LABEL_SORTED;
  }
  if (!is_zero(tr.weight) && !is_one(tr.weight)) out = (out | WEIGHTED) & ~UNWEIGHTED;
  if (tr.nextstate <= state) out = (out | NOT_TOP_SORTED) & ~TOP_SORTED;
  out &= ADD_ARC_MASK | ACCEPTOR | NO_EPSILONS | NO_I_EPSILONS | NO_O_EPSILONS | I_LABEL_SORTED | O_LABEL_SORTED |
         UNWEIGHTED | TOP_SORTED;
  if (out & TOP_SORTED) out |= ACYCLIC | INITIAL_ACYCLIC;
  return out;
}
// add_tr over a whole SET of arcs at once.  Every effect of add_tr is a sticky set/clear decided by one fact about the arc,
// followed by a mask that is the same for every arc, so folding add_tr over any number of arcs (in any order) equals one
// application with the union of their facts.  facts: 1 il != ol | 2 il == 0 | 4 il == 0 && ol == 0 | 8 ol == 0 |
// 16 ilabel below its predecessor's | 32 olabel below its predecessor's | 64 weight neither zero nor one | 128 nextstate <= state
inline uint64_t add_trs_by_facts(uint64_t in, uint32_t facts) {
  uint64_t out = in;
  if (facts & 1u) out = (out | NOT_ACCEPTOR) & ~ACCEPTOR;
  if (facts & 2u) out = (out | I_EPSILONS) & ~NO_I_EPSILONS;
  if (facts & 4u) out = (out | EPSILONS) & ~NO_EPSILONS;
  if (facts & 8u) out = (out | O_EPSILONS) & ~NO_O_EPSILONS;
  if (facts & 16u) out = (out | NOT_I_LABEL_SORTED) & ~I_LABEL_SORTED;
  if (facts & 32u) out = (out | NOT_O_LABEL_SORTED) & ~O_LABEL_SORTED;
  if (facts & 64u) out = (out | WEIGHTED) & ~UNWEIGHTED;
  if (facts & 128u) out = (out | NOT_TOP_SORTED) & ~TOP_SORTED;
  out &= ADD_ARC_MASK | ACCEPTOR | NO_EPSILONS | NO_I_EPSILONS | NO_O_EPSILONS | I_LABEL_SORTED | O_LABEL_SORTED |
         UNWEIGHTED | TOP_SORTED;
  if (out & TOP_SORTED) out |= ACYCLIC | INITIAL_ACYCLIC;
  return out;
}
// reverse_properties (mutate_properties.rs:622-638)
inline uint64_t reverse(uint64_t inprops, bool has_superinitial) {
  uint64_t out = (ACCEPTOR | NOT_ACCEPTOR | EPSILONS | I_EPSILONS | O_EPSILONS | UNWEIGHTED | CYCLIC | ACYCLIC |
                  WEIGHTED_CYCLES | UNWEIGHTED_CYCLES) & inprops;
  if (has_superinitial) out |= WEIGHTED & inprops;
  return out;
}

inline uint64_t delete_states(uint64_t in) { return in & DELETE_STATES_MASK; }  // :102-104
inline uint64_t compose(uint64_t p1, uint64_t p2) {                             // :151-184
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
inline uint64_t shortest_path(uint64_t p, bool tree) {  // :662-672
  uint64_t out = p | ACYCLIC | INITIAL_ACYCLIC | ACCESSIBLE | UNWEIGHTED_CYCLES;
  if (!tree) out |= COACCESSIBLE;
  return out;
}

// Property word of the linear FST single_shortest_path_backtrace builds (shortest_path.rs:241-282): state 0 final,
// state k >= 1 carries the single arc path_arcs[k-1] into state k-1, start = hops; the reference's incremental
// add_state / set_final / add_tr / set_start bookkeeping, then shortest_path_properties(.., true).
// add_tr(add_state(p), arc) is a pure function of (p, the five facts the arc contributes).  Arcs are processed in
// runs of equal facts; inside a run the word reaches a fixed point after a step or two (add_state(out) == in), and
// the rest of the run is skipped — the result is exactly the incremental one.
inline uint64_t linear_path_props(bool has_path, uint32_t hops, float final_weight, const wfst_tr* path_arcs) {
  uint64_t p = NULL_PROPS;
  if (has_path) {
    p = add_state(p);
    p = set_final(p, nullptr, &final_weight);
    auto facts_of = [&](uint32_t k) {  // arc of state k
      const wfst_tr& tr = path_arcs[k - 1];
      const bool weighted = !is_zero(tr.weight) && !is_one(tr.weight);
      return (tr.ilabel != tr.olabel ? 1u : 0u) | (tr.ilabel == WFST_EPS_LABEL ? 2u : 0u) |
             (tr.olabel == WFST_EPS_LABEL ? 4u : 0u) | (weighted ? 8u : 0u) | (tr.nextstate <= k ? 16u : 0u);
    };
    uint32_t k = 1;
    while (k <= hops) {
      const uint32_t f = facts_of(k);
      uint32_t end = k;
      while (end < hops && facts_of(end + 1) == f) ++end;
      for (uint32_t j = k; j <= end; ++j) {
        const uint64_t in = add_state(p);
        p = add_tr(in, j, path_arcs[j - 1], nullptr);
        if (add_state(p) == in) break;  // fixed point: every further arc of the run maps `in` to the same `p`
      }
      k = end + 1;
    }
    p = set_start(p);
  }
  return shortest_path(p, true) & ALL;
}

// The same word from the UNION of the arcs' facts, in add_trs_by_facts's encoding (1 il != ol | 2 il == 0 | 4 il == 0 and
// ol == 0 | 8 ol == 0 | 64 weighted | 128 nextstate <= state; a path's states have one arc each: no label-order facts).
// Every effect of add_tr is a sticky set / clear decided by one fact, followed by masks that are the same for every
// arc: the order of the arcs does not matter, and three applications of the union reach the fixed point.  The
// string o T kernel ORs the facts of a path's arcs while it writes them (compose.hip: Result::facts), so that the host
// does not read the arcs back to know the properties; tests/test_host.py (props_check) compares this function with
// linear_path_props on every fact sequence of up to four arcs.
constexpr uint32_t PATH_FACTS_NONE = 0xFFFFFFFFu;  // the kernel did not provide them: scan the arcs
inline uint32_t path_arc_facts(uint32_t ilabel, uint32_t olabel, float weight) {
  return (ilabel != olabel ? 1u : 0u) | (ilabel == WFST_EPS_LABEL ? 2u : 0u) |
         (ilabel == WFST_EPS_LABEL && olabel == WFST_EPS_LABEL ? 4u : 0u) | (olabel == WFST_EPS_LABEL ? 8u : 0u) |
         (!is_zero(weight) && !is_one(weight) ? 64u : 0u) | 128u;
}
inline uint64_t linear_path_props_from_facts(bool has_path, uint32_t hops, float final_weight, uint32_t facts_union) {
  uint64_t p = NULL_PROPS;
  if (has_path) {
    p = add_state(p);
    p = set_final(p, nullptr, &final_weight);
    for (uint32_t j = 0; j < (hops < 3u ? hops : 3u); ++j) p = add_trs_by_facts(add_state(p), facts_union);
    p = set_start(p);
  }
  return shortest_path(p, true) & ALL;
}

// project_properties (fst_properties/mutate_properties.rs:365-445) as applied by project() with the all_properties() mask
// (algorithms/projection.rs:91-94)
inline uint64_t project(uint64_t in, bool project_output) {
  uint64_t out = ACCEPTOR;
  out |= (WEIGHTED | UNWEIGHTED | WEIGHTED_CYCLES | UNWEIGHTED_CYCLES | CYCLIC | ACYCLIC | INITIAL_CYCLIC | INITIAL_ACYCLIC |
          TOP_SORTED | NOT_TOP_SORTED | ACCESSIBLE | NOT_ACCESSIBLE | COACCESSIBLE | NOT_COACCESSIBLE | STRING | NOT_STRING) & in;
  if (!project_output) {
    out |= (I_DETERMINISTIC | NOT_I_DETERMINISTIC | I_EPSILONS | NO_I_EPSILONS | I_LABEL_SORTED | NOT_I_LABEL_SORTED) & in;
    if (in & I_DETERMINISTIC) out |= O_DETERMINISTIC;
    if (in & NOT_I_DETERMINISTIC) out |= NOT_O_DETERMINISTIC;
    if (in & I_EPSILONS) out |= O_EPSILONS | EPSILONS;
    if (in & NO_I_EPSILONS) out |= NO_O_EPSILONS | NO_EPSILONS;
    if (in & I_LABEL_SORTED) out |= O_LABEL_SORTED;
    if (in & NOT_I_LABEL_SORTED) out |= NOT_O_LABEL_SORTED;
  } else {
    out |= (O_DETERMINISTIC | NOT_O_DETERMINISTIC | O_EPSILONS | NO_O_EPSILONS | O_LABEL_SORTED | NOT_O_LABEL_SORTED) & in;
    if (in & O_DETERMINISTIC) out |= I_DETERMINISTIC;
    if (in & NOT_O_DETERMINISTIC) out |= NOT_I_DETERMINISTIC;
    if (in & O_EPSILONS) out |= I_EPSILONS | EPSILONS;
    if (in & NO_O_EPSILONS) out |= NO_I_EPSILONS | NO_EPSILONS;
    if (in & O_LABEL_SORTED) out |= I_LABEL_SORTED;
    if (in & NOT_O_LABEL_SORTED) out |= NOT_I_LABEL_SORTED;
  }
  return (in & ~ALL) | out;
}

inline uint64_t compose_result(uint64_t p1, uint64_t p2, bool connected, bool has_start) {
  // start None: LazyFst::compute returns F2::new() untouched (lazy_fst.rs:229-232)
  uint64_t p = has_start ? compose(p1, p2) : NULL_PROPS;
  if (connected) {
    p = delete_states(p);
    p |= ACCESSIBLE | COACCESSIBLE;
  }
  return p;
}

}  // namespace wfst::props
