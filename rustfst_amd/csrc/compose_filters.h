// compose_filters.h — the ComposeFilterEnum filters (compose_static.rs:19-33) as per-composed-state outcome classes.
//
// With the default SortedMatcher pair a matcher can only produce four kinds of (arc1, arc2) pairs, and filter_tr's answer
// for each kind depends on the composed state alone (its filter state and the epsilon facts of s1 / s2 that set_state
// collects), never on the individual arcs:
//   X  arc1.olabel == NO_LABEL   fst1 stands still, fst2 takes an input-epsilon arc
//   Y  arc2.ilabel == NO_LABEL   fst2 stands still, fst1 takes an output-epsilon arc
//   Z  both arcs real, arc1.olabel == 0 == arc2.ilabel (epsilon:epsilon match)
//   M  both arcs real, matching non-epsilon label: every filter answers 0 (the start state)
// Shared by compose.hip (one wave per problem) and compose_wide.hip (one wave per composed state).
#pragma once
#include <cstdint>

namespace wfst {

constexpr uint32_t FILTER_REJECT = 0xFFFFFFFFu;  // FilterState::new_no_state()

struct FilterOutcomes {
  uint32_t fsX, fsY, fsZ;
};

// filter: ComposeFilterEnum value (0 Auto == 3 Sequence); fs: filter state of the composed state;
// alleps1 / noeps1: every arc / no arc of s1 has olabel 0 (alleps also needs s1 non-final); alleps2 / noeps2: same for
// s2's ilabels (set_state: sequence_compose_filter.rs:134-148, alt_sequence_compose_filter.rs:143-158,
// match_compose_filter.rs:126-147)
__host__ __device__ inline FilterOutcomes filter_outcomes(uint32_t filter, uint32_t fs, bool alleps1, bool noeps1, bool alleps2,
                                                          bool noeps2) {
  constexpr uint32_t R = FILTER_REJECT;
  FilterOutcomes o;
  switch (filter) {
    case 1:  // NullComposeFilter, null_compose_filter.rs:122-129
      o.fsX = o.fsY = R;
      o.fsZ = 0u;
      break;
    case 2:  // TrivialComposeFilter, trivial_compose_filter.rs:122-124
      o.fsX = o.fsY = o.fsZ = 0u;
      break;
    case 4:  // AltSequenceComposeFilter, alt_sequence_compose_filter.rs:160-181
      o.fsY = alleps2 ? R : (noeps2 ? 0u : 1u);
      o.fsX = fs == 1u ? R : 0u;
      o.fsZ = R;
      break;
    case 5:  // MatchComposeFilter, match_compose_filter.rs:149-205
      o.fsY = fs == 0u ? (noeps2 ? 0u : (alleps2 ? R : 1u)) : (fs == 1u ? 1u : R);
      o.fsX = fs == 0u ? (noeps1 ? 0u : (alleps1 ? R : 2u)) : (fs == 2u ? 2u : R);
      o.fsZ = fs == 0u ? 0u : R;
      break;
    case 6:  // NoMatchComposeFilter, no_match_compose_filter.rs:122-126
      o.fsX = o.fsY = 0u;
      o.fsZ = R;
      break;
    default:  // Auto / SequenceComposeFilter, sequence_compose_filter.rs:150-171
      o.fsX = alleps1 ? R : (noeps1 ? 0u : 1u);
      o.fsY = fs != 0u ? R : 0u;
      o.fsZ = R;
      break;
  }
  return o;
}

}  // namespace wfst
