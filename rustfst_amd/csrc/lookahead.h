// Look-ahead composition: shared declarations of lookahead.cpp (host precompute) and
// compose_lookahead.hip (the kernel and the C-ABI entry points' implementation).
#pragma once
#include <atomic>
#include <memory>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace wfst {

// (resize() of hundreds of MB that are overwritten at once should not zero them first, on one thread)
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = NoInitAlloc<U>;
  };
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
    else ::new ((void*)p) U(std::forward<A>(a)...);
  }
};

// LabelReachableData (compose/label_reachable.rs:16-21) with the interval sets flattened to CSR
struct LabelReachData {
  bool reach_input = false;
  uint32_t final_label = WFST_NO_LABEL;                 // index of the NO_LABEL sink (a final state is reachable)
  std::unordered_map<uint32_t, uint32_t> label2index;   // label -> relabelled label (grows in relabel())
  std::vector<uint32_t> iv_off;                         // [n+1] first interval of each state
  std::vector<uint32_t, NoInitAlloc<uint32_t>> iv;      // [2 * n_intervals] begin, end (half-open), sorted, disjoint

  void compute(uint32_t n_states, const uint32_t* offsets, const wfst_tr* arcs, const float* finals, bool reach_input);
  uint32_t relabel(uint32_t label);
  uint64_t relabel_fst(uint32_t n_states, const uint32_t* offsets, wfst_tr* arcs, uint64_t props_in, bool relabel_input);
};

uint64_t tr_sort_props(uint64_t in, bool ilabel_cmp);  // tr_sort.hip

}  // namespace wfst

// MatcherFst<.., LabelLookAheadMatcher, LabelReachableData> for an output look-ahead matcher (matcher_fst.rs:21-94)
struct wfst_lookahead {
  wfst_ctx* ctx = nullptr;  // null: host-only handle (wfst_label_reachable_compute)
  wfst::LabelReachData data;
  wfst_fst* fst1 = nullptr;  // relabelled, olabel-sorted copy of the first operand (owned)
  std::unique_ptr<wfst::DBuf<uint32_t>> d_iv_off, d_iv;
  // what the last composition on the wide driver came to (states, arcs before the gather): the next one against this
  // operand starts with an arena of that size instead of growing into it (eight growths, a quarter of a 90 M-state run)
  mutable std::atomic<uint64_t> last_wide_states{0}, last_wide_arcs{0}, last_wide_input_arcs{0};  // arena hint of the wide driver (best effort)
  ~wfst_lookahead();
};

namespace wfst {
wfst_lookahead* lookahead_create(wfst_ctx* ctx, const wfst_fst* fst1);
wfst_fst* lookahead_relabel(wfst_lookahead* la, const wfst_fst* fst2);
wfst_fst* compose_lookahead(wfst_ctx* ctx, const wfst_lookahead* la, const wfst_fst* fst2);
void compose_lookahead_batch(wfst_ctx* ctx, const wfst_lookahead* la, const wfst_fst* const* fst2s, size_t n, wfst_fst** outs);
}  // namespace wfst
