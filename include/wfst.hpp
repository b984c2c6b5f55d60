// wfst.hpp — header-only C++17 mirror of the part of rustfst's Rust interface that sits on the compose -> shortest-path
// path, over the C-ABI of wfst.h.  Same names, argument meaning and error behaviour as the reference, so that host code
// (and tests) read like rustfst's own:
//
//   rustfst                                                     here (namespace wfst_amd)
//   ----------------------------------------------------------  -------------------------------------------------
//   Tr::new(ilabel, olabel, weight, nextstate)      tr.rs:6-15   Tr{ilabel, olabel, weight, nextstate} (== wfst_tr)
//   VectorFst::<TropicalWeight>::new()                           VectorFst()
//   MutableFst::{add_state, add_states, set_start, set_final,    same member names (fst_traits/mutable_fst.rs)
//               add_tr, delete_final_weight}
//   CoreFst::{start, final_weight, num_trs, get_trs},            same member names (fst_traits/fst.rs:18-250);
//   ExpandedFst::num_states, Fst::properties                     Option<T> -> std::optional<T>
//   tr_sort(&mut fst, ILabelCompare{} | OLabelCompare{})         tr_sort(fst, ILabelCompare{} | OLabelCompare{})
//   project(&mut fst, ProjectType::ProjectInput)                 project(fst, ProjectType::ProjectInput)
//   connect(&mut fst) / rm_epsilon(&mut fst)                     connect(fst) / rm_epsilon(fst)
//   (look-ahead recipe of rustfst-cli/src/cmds/compose.rs)       LookAheadFst(fst1).compose(fst2) / compose_lookahead
//   compose(fst1, fst2) / compose_with_config(.., ComposeConfig) compose(..) / compose_with_config(..)   compose_static.rs:166-306
//   shortest_path(&fst) / shortest_path_with_config(..)          shortest_path(..) / shortest_path_with_config(..)  shortest_path.rs:76-133
//   anyhow::Result<T> Err(e)                                     throws wfst_amd::Error (what() = the library's message)
//
// The algorithms run on the GPU through a per-thread Context; a VectorFst lives on the host (wfst_vec_fst) and is
// flattened / rebuilt around each call exactly as the Rust shim of INTEGRATION.md does.
#ifndef WFST_AMD_HPP
#define WFST_AMD_HPP

#include <cmath>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "wfst.h"

namespace wfst_amd {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline void check(wfst_status st) {  // check_ffi_error (rustfst-python/rustfst/ffi_utils.py)
  if (st == WFST_OK) return;
  char* msg = nullptr;
  std::string text = "unknown error";
  if (wfst_last_error(&msg) == WFST_OK && msg) {
    text = msg;
    wfst_string_destroy(msg);
  }
  throw Error(text);
}

using Tr = wfst_tr;
using StateId = uint32_t;
using Label = uint32_t;
constexpr Label EPS_LABEL = 0;

enum class ComposeFilterEnum : uint32_t {  // compose_static.rs:19-33
  AutoFilter = 0, NullFilter = 1, TrivialFilter = 2, SequenceFilter = 3, AltSequenceFilter = 4, MatchFilter = 5, NoMatchFilter = 6
};
struct ComposeConfig {  // compose_static.rs:35-65 (default: AutoFilter, connect = true)
  ComposeFilterEnum compose_filter = ComposeFilterEnum::AutoFilter;
  bool connect = true;
};
struct ShortestPathConfig {  // shortest_path.rs:26-74 (default: delta 1e-6, nshortest 1, unique false)
  float delta = 1e-6f;
  size_t nshortest = 1;
  bool unique = false;
  ShortestPathConfig with_nshortest(size_t n) const { ShortestPathConfig c = *this; c.nshortest = n; return c; }
  ShortestPathConfig with_unique(bool u) const { ShortestPathConfig c = *this; c.unique = u; return c; }
};
struct ILabelCompare {};
struct OLabelCompare {};

class Context {  // one per (thread, GPU)
 public:
  explicit Context(int device = 0) { check(wfst_ctx_create(device, &h_)); }
  ~Context() { if (h_) wfst_ctx_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  wfst_ctx* get() const { return h_; }
  static Context& current() {
    static thread_local Context ctx(0);
    return ctx;
  }

 private:
  wfst_ctx* h_ = nullptr;
};

class VectorFst {
 public:
  VectorFst() { check(wfst_vec_fst_new(&h_)); }
  ~VectorFst() { if (h_) wfst_vec_fst_destroy(h_); }
  VectorFst(const VectorFst& o) { check(wfst_vec_fst_copy(o.h_, &h_)); }
  VectorFst(VectorFst&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  VectorFst& operator=(VectorFst o) noexcept { std::swap(h_, o.h_); return *this; }

  StateId add_state() { StateId s; check(wfst_vec_fst_add_state(h_, &s)); return s; }
  void add_states(size_t n) { for (size_t i = 0; i < n; ++i) add_state(); }
  void set_start(StateId s) { check(wfst_vec_fst_set_start(h_, s)); }
  void set_final(StateId s, float weight = 0.0f) { check(wfst_vec_fst_set_final(h_, s, weight)); }  // W::one() by default
  void delete_final_weight(StateId s) { check(wfst_vec_fst_del_final_weight(h_, s)); }
  void add_tr(StateId s, const Tr& tr) { check(wfst_vec_fst_add_tr(h_, s, &tr)); }

  size_t num_states() const { uint32_t n; check(wfst_vec_fst_num_states(h_, &n)); return n; }
  std::optional<StateId> start() const {
    int64_t s;
    check(wfst_vec_fst_start(h_, &s));
    return s < 0 ? std::nullopt : std::optional<StateId>((StateId)s);
  }
  std::optional<float> final_weight(StateId s) const {
    float w;
    int some;
    check(wfst_vec_fst_final_weight(h_, s, &w, &some));
    return some ? std::optional<float>(w) : std::nullopt;
  }
  bool is_final(StateId s) const { return final_weight(s).has_value(); }
  size_t num_trs(StateId s) const { uint64_t n; check(wfst_vec_fst_num_trs(h_, s, &n)); return (size_t)n; }
  std::vector<Tr> get_trs(StateId s) const {
    uint64_t n = 0;
    check(wfst_vec_fst_num_trs(h_, s, &n));
    std::vector<Tr> out((size_t)n);
    check(wfst_vec_fst_get_trs(h_, s, out.data(), n, &n));
    return out;
  }
  uint64_t properties() const { uint64_t p; check(wfst_vec_fst_properties(h_, &p)); return p; }
  // PartialEq of VectorFst (vector_fst/data_structure.rs:36-41): states, arcs, finals (approximate weights), start
  bool operator==(const VectorFst& o) const { int eq; check(wfst_vec_fst_equals(h_, o.h_, &eq)); return eq != 0; }
  bool operator!=(const VectorFst& o) const { return !(*this == o); }

  wfst_vec_fst* raw() const { return h_; }
  static VectorFst adopt(wfst_vec_fst* h) { VectorFst f(nullptr); f.h_ = h; return f; }

 private:
  explicit VectorFst(std::nullptr_t) {}
  wfst_vec_fst* h_ = nullptr;
};

inline void tr_sort(VectorFst& fst, ILabelCompare) { check(wfst_vec_fst_tr_sort(fst.raw(), 1)); }  // tr_sort.rs:13-62
inline void tr_sort(VectorFst& fst, OLabelCompare) { check(wfst_vec_fst_tr_sort(fst.raw(), 0)); }

namespace detail {
struct DeviceFst {  // owned wfst_fst handle
  wfst_fst* h = nullptr;
  DeviceFst() = default;
  DeviceFst(const DeviceFst&) = delete;
  ~DeviceFst() { if (h) wfst_fst_destroy(h); }
};
inline void upload(const VectorFst& f, DeviceFst& d) { check(wfst_vec_fst_to_device(Context::current().get(), f.raw(), &d.h)); }
inline VectorFst download(const DeviceFst& d) {
  wfst_vec_fst* out = nullptr;
  check(wfst_vec_fst_from_device(d.h, &out));
  return VectorFst::adopt(out);
}
}  // namespace detail

// compose_with_config (compose_static.rs:166-266): Err when neither side is known label-sorted ("... (sort?)")
inline VectorFst compose_with_config(const VectorFst& fst1, const VectorFst& fst2, const ComposeConfig& config) {
  detail::DeviceFst a, b, c;
  detail::upload(fst1, a);
  detail::upload(fst2, b);
  const wfst_compose_config cfg{(uint32_t)config.compose_filter, config.connect ? 1u : 0u};
  check(wfst_compose(Context::current().get(), a.h, b.h, &cfg, &c.h));
  return detail::download(c);
}
inline VectorFst compose(const VectorFst& fst1, const VectorFst& fst2) {  // compose_static.rs:293-303
  return compose_with_config(fst1, fst2, ComposeConfig{});
}

// shortest_path_with_config (shortest_path.rs:107-170)
inline VectorFst shortest_path_with_config(const VectorFst& ifst, const ShortestPathConfig& config) {
  detail::DeviceFst a, c;
  detail::upload(ifst, a);
  const wfst_shortest_path_config cfg{config.delta, (uint64_t)config.nshortest, config.unique ? 1u : 0u};
  check(wfst_shortest_path(Context::current().get(), a.h, &cfg, &c.h));
  return detail::download(c);
}
inline VectorFst shortest_path(const VectorFst& ifst) { return shortest_path_with_config(ifst, ShortestPathConfig{}); }

// project (algorithms/projection.rs:65-95): in place, like the reference
enum class ProjectType { ProjectInput, ProjectOutput };  // projection.rs:7-13
inline void project(VectorFst& fst, ProjectType project_type) {
  detail::DeviceFst a;
  detail::upload(fst, a);
  check(wfst_fst_project(Context::current().get(), a.h, project_type == ProjectType::ProjectOutput ? 1 : 0));
  fst = detail::download(a);
}

// connect (algorithms/connect.rs:51-66) and rm_epsilon (algorithms/rm_epsilon/rm_epsilon_static.rs:50-163): in place
inline void connect(VectorFst& fst) {
  detail::DeviceFst a, c;
  detail::upload(fst, a);
  check(wfst_connect(Context::current().get(), a.h, &c.h));
  fst = detail::download(c);
}
inline void rm_epsilon(VectorFst& fst) {
  detail::DeviceFst a, c;
  detail::upload(fst, a);
  check(wfst_rm_epsilon(Context::current().get(), a.h, &c.h));
  fst = detail::download(c);
}

// Look-ahead composition.  The reference has no single function for it: callers assemble MatcherFst::new_with_relabeling,
// a LabelLookAheadMatcher and the PushLabels(PushWeights(LookAhead(AltSequence))) filter by hand and call compute()
// (rustfst-cli/src/cmds/compose.rs:77-181).  LookAheadFst is the MatcherFst of that recipe (keep it while the same first
// operand meets further second operands); compose_lookahead is the whole recipe for one pair.  The result is not connected.
class LookAheadFst {
 public:
  explicit LookAheadFst(const VectorFst& fst1) {  // MatcherFst::new, compose/matcher_fst.rs:55-71
    detail::DeviceFst a;
    detail::upload(fst1, a);
    check(wfst_lookahead_create(Context::current().get(), a.h, &h_));
  }
  ~LookAheadFst() { if (h_) wfst_lookahead_destroy(h_); }
  LookAheadFst(const LookAheadFst&) = delete;
  LookAheadFst& operator=(const LookAheadFst&) = delete;
  // LabelLookAheadRelabeler::relabel(fst2, .., true) + tr_sort(ILabelCompare), then ComposeFst::new_with_options(..).compute()
  VectorFst compose(const VectorFst& fst2) const {
    detail::DeviceFst b, br, c;
    detail::upload(fst2, b);
    check(wfst_lookahead_relabel(h_, b.h, &br.h));
    check(wfst_compose_lookahead(Context::current().get(), h_, br.h, &c.h));
    return detail::download(c);
  }
  wfst_lookahead* raw() const { return h_; }

 private:
  wfst_lookahead* h_ = nullptr;
};
inline VectorFst compose_lookahead(const VectorFst& fst1, const VectorFst& fst2) { return LookAheadFst(fst1).compose(fst2); }

}  // namespace wfst_amd

#endif  // WFST_AMD_HPP
