// vector_fst.cpp — host-side mutable VectorFst<TropicalWeight> mirror (wfst_vec_fst_*).
// Data structure: rustfst/src/fst_impls/vector_fst/data_structure.rs:16-34.
// Mutations + property bookkeeping: rustfst/src/fst_impls/vector_fst/mutable_fst.rs:25-281.
// This is plumbing for callers without a Rust VectorFst (Python mirror, C++ programs); the GPU
// engine itself only ever sees the flattened CSR (wfst_fst_upload).
#include <algorithm>

#include "common.h"
#include "fst_props.h"

using namespace wfst;

struct VecState {
  bool has_final = false;
  float final_w = INF;
  std::vector<wfst_tr> trs;
  size_t niepsilons = 0, noepsilons = 0;
};

struct wfst_vec_fst {
  std::vector<VecState> states;
  int64_t start = -1;
  uint64_t properties = props::NULL_PROPS;  // VectorFst::new(), mutable_fst.rs:25-33
};

namespace {
VecState& state_at(wfst_vec_fst* f, uint32_t s) {
  if (s >= f->states.size()) throw Error("State " + std::to_string(s) + " doesn't exist");
  return f->states[s];
}
const VecState& state_at(const wfst_vec_fst* f, uint32_t s) {
  if (s >= f->states.size()) throw Error("State " + std::to_string(s) + " doesn't exist");
  return f->states[s];
}
// Tr / TropicalWeight PartialEq: labels + nextstate exact, weight within KDELTA (semiring.rs:159-168)
bool tr_eq(const wfst_tr& a, const wfst_tr& b) {
  return a.ilabel == b.ilabel && a.olabel == b.olabel && a.nextstate == b.nextstate && props::approx_eq(a.weight, b.weight);
}
}  // namespace

extern "C" {

wfst_status wfst_vec_fst_new(wfst_vec_fst** out) {
  return wrap([&] {
    if (!out) throw Error("null out pointer");
    *out = new wfst_vec_fst();
  });
}
wfst_status wfst_vec_fst_destroy(wfst_vec_fst* f) {
  delete f;
  return WFST_OK;
}
wfst_status wfst_vec_fst_copy(const wfst_vec_fst* f, wfst_vec_fst** out) {
  return wrap([&] {
    if (!f || !out) throw Error("null pointer");
    *out = new wfst_vec_fst(*f);
  });
}
wfst_status wfst_vec_fst_add_state(wfst_vec_fst* f, uint32_t* state) {  // mutable_fst.rs:82-87
  return wrap([&] {
    if (!f) throw Error("null fst");
    f->states.emplace_back();
    f->properties = props::add_state(f->properties);
    if (state) *state = (uint32_t)(f->states.size() - 1);
  });
}
wfst_status wfst_vec_fst_add_tr(wfst_vec_fst* f, uint32_t state, const wfst_tr* tr) {  // mutable_fst.rs:235-244
  return wrap([&] {
    if (!f || !tr) throw Error("null pointer");
    VecState& st = state_at(f, state);
    if (tr->ilabel == WFST_EPS_LABEL) st.niepsilons++;
    if (tr->olabel == WFST_EPS_LABEL) st.noepsilons++;
    st.trs.push_back(*tr);
    const wfst_tr* prev = st.trs.size() > 1 ? &st.trs[st.trs.size() - 2] : nullptr;
    f->properties = props::add_tr(f->properties, state, st.trs.back(), prev);  // data_structure.rs:76-92
  });
}
wfst_status wfst_vec_fst_set_start(wfst_vec_fst* f, uint32_t state) {  // mutable_fst.rs:35-44
  return wrap([&] {
    if (!f) throw Error("null fst");
    if (state >= f->states.size()) throw Error("The state " + std::to_string(state) + " doesn't exist");
    f->start = state;
    f->properties = props::set_start(f->properties);
  });
}
wfst_status wfst_vec_fst_set_final(wfst_vec_fst* f, uint32_t state, float weight) {  // mutable_fst.rs:52-65
  return wrap([&] {
    if (!f) throw Error("null fst");
    if (state >= f->states.size()) throw Error("Stateid " + std::to_string(state) + " doesn't exist");
    VecState& st = f->states[state];
    f->properties = props::set_final(f->properties, st.has_final ? &st.final_w : nullptr, &weight);
    st.has_final = true;
    st.final_w = weight;
  });
}
wfst_status wfst_vec_fst_del_final_weight(wfst_vec_fst* f, uint32_t state) {  // mutable_fst.rs:283-291
  return wrap([&] {
    if (!f) throw Error("null fst");
    VecState& st = state_at(f, state);
    f->properties = props::set_final(f->properties, st.has_final ? &st.final_w : nullptr, nullptr);
    st.has_final = false;
    st.final_w = INF;
  });
}
wfst_status wfst_vec_fst_num_states(const wfst_vec_fst* f, uint32_t* n) {
  return wrap([&] {
    if (!f || !n) throw Error("null pointer");
    *n = (uint32_t)f->states.size();
  });
}
wfst_status wfst_vec_fst_start(const wfst_vec_fst* f, int64_t* start) {
  return wrap([&] {
    if (!f || !start) throw Error("null pointer");
    *start = f->start;
  });
}
wfst_status wfst_vec_fst_final_weight(const wfst_vec_fst* f, uint32_t state, float* weight, int* is_some) {
  return wrap([&] {
    if (!f || !weight || !is_some) throw Error("null pointer");
    const VecState& st = state_at(f, state);
    *is_some = st.has_final ? 1 : 0;
    *weight = st.has_final ? st.final_w : INF;
  });
}
wfst_status wfst_vec_fst_num_trs(const wfst_vec_fst* f, uint32_t state, uint64_t* n) {
  return wrap([&] {
    if (!f || !n) throw Error("null pointer");
    *n = state_at(f, state).trs.size();
  });
}
wfst_status wfst_vec_fst_get_trs(const wfst_vec_fst* f, uint32_t state, wfst_tr* out, uint64_t cap, uint64_t* n) {
  return wrap([&] {
    if (!f || !n) throw Error("null pointer");
    const VecState& st = state_at(f, state);
    *n = st.trs.size();
    if (out) std::memcpy(out, st.trs.data(), std::min<uint64_t>(cap, st.trs.size()) * sizeof(wfst_tr));
  });
}
wfst_status wfst_vec_fst_properties(const wfst_vec_fst* f, uint64_t* p) {
  return wrap([&] {
    if (!f || !p) throw Error("null pointer");
    *p = f->properties;
  });
}
// tr_sort with ILabelCompare / OLabelCompare: rustfst/src/algorithms/tr_sort.rs:13-62 (stable sort_by)
wfst_status wfst_vec_fst_tr_sort(wfst_vec_fst* f, int ilabel_cmp) {
  return wrap([&] {
    if (!f) throw Error("null fst");
    for (VecState& st : f->states) {
      if (ilabel_cmp)
        std::stable_sort(st.trs.begin(), st.trs.end(), [](const wfst_tr& a, const wfst_tr& b) { return a.ilabel < b.ilabel; });
      else
        std::stable_sort(st.trs.begin(), st.trs.end(), [](const wfst_tr& a, const wfst_tr& b) { return a.olabel < b.olabel; });
    }
    using namespace props;
    const uint64_t in = f->properties;
    const uint64_t arcsort_mask = ALL & ~(I_LABEL_SORTED | NOT_I_LABEL_SORTED | O_LABEL_SORTED | NOT_O_LABEL_SORTED);
    uint64_t out = (in & arcsort_mask) | (ilabel_cmp ? I_LABEL_SORTED : O_LABEL_SORTED);
    if (in & ACCEPTOR) out |= ilabel_cmp ? O_LABEL_SORTED : I_LABEL_SORTED;
    f->properties = out;
  });
}
// VectorFst PartialEq (data_structure.rs:36-41,28-34): states (final, trs, eps counters) + start; not props
wfst_status wfst_vec_fst_equals(const wfst_vec_fst* a, const wfst_vec_fst* b, int* equal) {
  return wrap([&] {
    if (!a || !b || !equal) throw Error("null pointer");
    *equal = 0;
    if (a->start != b->start || a->states.size() != b->states.size()) return;
    for (size_t s = 0; s < a->states.size(); ++s) {
      const VecState &x = a->states[s], &y = b->states[s];
      if (x.has_final != y.has_final) return;
      if (x.has_final && !props::approx_eq(x.final_w, y.final_w)) return;
      if (x.trs.size() != y.trs.size() || x.niepsilons != y.niepsilons || x.noepsilons != y.noepsilons) return;
      for (size_t i = 0; i < x.trs.size(); ++i)
        if (!tr_eq(x.trs[i], y.trs[i])) return;
    }
    *equal = 1;
  });
}

wfst_status wfst_vec_fst_to_device(wfst_ctx* ctx, const wfst_vec_fst* f, wfst_fst** out) {
  return wrap([&] {
    if (!ctx || !f || !out) throw Error("null pointer");
    std::vector<uint32_t> offsets;
    std::vector<wfst_tr> arcs;
    std::vector<float> finals;
    offsets.reserve(f->states.size() + 1);
    offsets.push_back(0);
    for (const VecState& st : f->states) {
      arcs.insert(arcs.end(), st.trs.begin(), st.trs.end());
      if (arcs.size() > 0xFFFFFFFFull) throw Error("FST too large: more than 2^32 arcs");
      offsets.push_back((uint32_t)arcs.size());
      finals.push_back(st.has_final ? st.final_w : INF);
    }
    *out = upload_from_host(ctx, (uint32_t)f->states.size(), f->start, offsets.data(), arcs.data(), finals.data(),
                            f->properties);
  });
}

wfst_status wfst_vec_fst_from_device(const wfst_fst* fst, wfst_vec_fst** out) {
  return wrap([&] {
    if (!fst || !out) throw Error("null pointer");
    ensure_host(fst);
    auto v = std::make_unique<wfst_vec_fst>();
    v->states.resize(fst->n_states);
    const HostCsr& h = fst->host;
    for (uint32_t s = 0; s < fst->n_states; ++s) {
      VecState& st = v->states[s];
      st.trs.assign(h.arcs.begin() + h.offsets[s], h.arcs.begin() + h.offsets[s + 1]);
      for (const wfst_tr& tr : st.trs) {  // set_trs_unchecked recount, mutable_fst.rs:262-279
        st.niepsilons += tr.ilabel == WFST_EPS_LABEL;
        st.noepsilons += tr.olabel == WFST_EPS_LABEL;
      }
      if (h.finals[s] != INF) {
        st.has_final = true;
        st.final_w = h.finals[s];
      }
    }
    v->start = fst->start;
    v->properties = fst->props;  // set_properties(op.properties) / the algorithm's property word
    *out = v.release();
  });
}

}  // extern "C"
