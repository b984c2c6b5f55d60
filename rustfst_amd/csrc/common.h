// common.h — shared host-side definitions of libwfst_amd (context, FST handle, device pool, errors).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/wfst.h"

namespace wfst {

constexpr float INF = __builtin_huge_valf();

// one turn of a host spin loop on a word in pinned memory (the completion tickets)
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  asm volatile("" ::: "memory");
#endif
}

// ---------------------------------------------------------------- errors
struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
void set_last_error(const std::string& msg);

#define HIP_CHECK(expr)                                                                                  \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess)                                                                                \
      throw ::wfst::Error(std::string("HIP error ") + hipGetErrorString(_e) + " at " __FILE__ ":" +     \
                          std::to_string(__LINE__) + " in " #expr);                                      \
  } while (0)

template <class F>
wfst_status wrap(F&& f) noexcept {  // rustfst-ffi/src/lib.rs:43-56 `wrap`
  try {
    f();
    return WFST_OK;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return WFST_KO;
  } catch (...) {
    set_last_error("unknown error");
    return WFST_KO;
  }
}

// ---------------------------------------------------------------- device memory pool
// Size-bucketed caching allocator: hot paths never call hipMalloc/hipFree after warm-up.  Thread-safe: an FST shared
// by several contexts (threads) may grow its cached derived data from any of them.
class DevicePool {
 public:
  explicit DevicePool(int device) { (void)device; }
  ~DevicePool();
  void* alloc(size_t bytes);
  void free(void* p);
  void trim();

 private:
  static size_t bucket(size_t bytes);
  void trim_locked();
  std::mutex mu_;
  std::multimap<size_t, void*> free_;
  std::map<void*, size_t> live_;
};

// RAII device buffer from the pool
template <class T>
struct DBuf {
  DevicePool* pool = nullptr;
  T* p = nullptr;
  size_t n = 0;
  DBuf() = default;
  DBuf(DevicePool& pl, size_t count) : pool(&pl), p((T*)pl.alloc(std::max<size_t>(count, 1) * sizeof(T))), n(count) {}
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  DBuf(DBuf&& o) noexcept : pool(o.pool), p(o.p), n(o.n) { o.p = nullptr; }
  DBuf& operator=(DBuf&& o) noexcept {
    reset();
    pool = o.pool;
    p = o.p;
    n = o.n;
    o.p = nullptr;
    return *this;
  }
  void reset() {
    if (p) pool->free(p);
    p = nullptr;
  }
  ~DBuf() { reset(); }
};

// pinned host staging buffer (grow-only)
struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  void* get(size_t bytes);
  ~PinnedBuf();
};

// Pinned host blocks handed out by reference count: the kernel of a fused batch writes its results and path arcs into one,
// and the path FSTs the batch returns point into it (wfst_fst::path_form) until somebody asks for their arrays.  The
// block goes back to the ring when the job and every such result are gone (a serving loop alternates between two).
struct PinnedBlock {
  void* p = nullptr;
  size_t cap = 0;
};
class PinnedRing : public std::enable_shared_from_this<PinnedRing> {
 public:
  std::shared_ptr<PinnedBlock> take(size_t bytes);
  ~PinnedRing();

 private:
  std::mutex mu_;
  std::vector<PinnedBlock*> free_;
};

}  // namespace wfst

// ---------------------------------------------------------------- handles
struct wfst_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  // shared with every FST handle created on this context (and its arenas): the device memory of a handle that outlives
  // its context (interpreter shutdown destroys in any order) is still released to a live pool, which goes with the last owner
  std::shared_ptr<wfst::DevicePool> pool;
  wfst::PinnedBuf pinned;      // small D2H/H2D staging
  wfst::PinnedBuf pinned_big;  // batch descriptors / results
  bool profiling = false;
  bool tie_reference = false;  // wfst_ctx_set_tie_order: the reference's predecessor choice on acyclic inputs
  uint32_t resident_share = 0; // wfst_ctx_set_resident_share: 0 = resident launches may fill the device, 1 = at most half of it
  // A resident relaxation launch of this context gave up waiting (its grid was not resident as a whole: another tenant held
  // compute units): solves take one launch per level until `resident_retry_at`, then a resident launch is tried again;
  // the pause doubles with every abort in a row (50 ms .. 3.2 s) and is forgotten by the first resident solve that completes.
  int64_t resident_retry_at_ns = 0;  // steady clock; 0 = resident launches allowed
  bool resident_hold = false;        // the repeat of a solve whose resident launch gave up: never a resident one
  uint32_t resident_abort_streak = 0;
  bool resident_allowed() const;
  void resident_aborted();
  void resident_completed() { resident_abort_streak = 0; }
  // wfst_ctx_set_profiling(ctx, 2): no per-launch events; the sweeps of a repeated (predicted) shortest_path query are timed
  // as ONE chain between two events on the stream, without any synchronisation between launches
  bool chain_timing = false;
  hipEvent_t ev_chain[2] = {nullptr, nullptr};
  bool batch_in_flight = false;  // wfst_compose_shortest_path_batch_begin .. _end
  bool sp_in_flight = false;     // wfst_shortest_path_begin .. _end
  wfst_stats stats{};
  struct SweepSample {
    double ms;
    uint64_t arcs, states;
    uint32_t mode = 0;  // what ran the level (Ctl::mode of its slot): mailbox MODE_*, or 0 atomic sweep / 7 binned level
  };
  std::vector<SweepSample> sweep_trace;  // profiling only: one entry per relaxation launch of the last solve
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // cached HIP graph of one batch of relaxation sweeps (sssp.hip); rebuilt when any node argument changes
  struct SweepGraph {
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
    uint64_t key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  } sweep_graph[3];  // [0]: first batch of a solve (predicted length), [1]: 8 sweeps per replay, [2]: 64
  wfst::PinnedBuf pinned_flags;  // host mirror of the per-sweep activity flags (its address is baked into the graphs)
  std::shared_ptr<wfst::PinnedRing> pinned_ring = std::make_shared<wfst::PinnedRing>();  // result blocks of the fused batches
  int n_cus = 256;
};

// Device-resident CSR. All arrays live in one arena allocation that may be
// shared between several FSTs (wfst_fst_upload_many).
struct DeviceArena {
  std::shared_ptr<wfst::DevicePool> pool;
  void* base = nullptr;
  size_t bytes = 0;
  ~DeviceArena();
};

struct DeviceCsr {
  std::shared_ptr<DeviceArena> arena;
  const uint32_t* offsets = nullptr;  // [n+1]
  const wfst_tr* arcs = nullptr;      // [E] AoS 16 B == CTr == on-disk arc
  const float* finals = nullptr;      // [n], +inf = non-final
  const uint32_t* noeps = nullptr;    // [n] number of output-epsilon arcs (VectorFstState.noepsilons)
  const uint2* wn = nullptr;          // [E] packed {weight bits, nextstate}: the 8 B the relaxation needs
  const uint4* srec = nullptr;        // [n] {arc begin, arc count, final bits, SREC_* epsilon facts}: one 16-B load per state in compose
};

namespace wfst {
// reverse(fst) (reverse.rs:33-87) for the n-best search: state 0 = super-initial (one eps arc per final state of fst, in
// state order, kept on the host), state t+1 = the in-arcs of t in reverse()'s order.  The in-arc segments live on the
// host for small FSTs; for large ones they STAY in HBM and the host search fetches the few segments it visits.
struct RevFst {
  uint32_t n = 0;                 // states of the original FST
  std::vector<wfst_tr> super;     // arcs of state 0
  std::vector<float> finals;      // [n+1] final weights of rfst (+inf = none)
  bool on_host = true;
  std::vector<uint32_t> h_roff;   // [n+1] segment offsets (on_host)
  std::vector<wfst_tr> h_arcs;    // [E]
  DBuf<uint32_t> d_roff;          // same on the device (!on_host)
  DBuf<wfst_tr> d_arcs;
  uint64_t fetched_segments = 0;  // statistics
};
struct RevCsr {
  DBuf<uint32_t> off;  // [n+1]
  DBuf<uint4> arc;     // [E] {source state, position of the arc in the source's arc list, weight bits, 0}
};
// Message-region plan of the mailbox relaxation sweeps (sssp_mailbox.h): offsets of the region reserved for every
// (source block, destination block) pair, sized by the number of arcs between the two blocks.
struct MboxPlan {
  uint32_t nb = 0;         // blocks of 1 << log states
  uint32_t log = 12;
  DBuf<uint32_t> roff;     // [nb*nb + 1] destination-major
  DBuf<uint32_t> roff_t;   // [nb*nb]     source-major copy
  // regions of the resident kernel (sssp_resident.h): 16-byte header + one slot per arc, 64-byte aligned, in 8-byte units
  DBuf<uint32_t> roffh;    // [nb*nb + 1] destination-major
  DBuf<uint32_t> roffh_t;  // [nb*nb]     source-major copy
  uint64_t res_units = 0;  // units of one parity buffer
  uint32_t lps = 8;        // lanes per listed state of the resident kernel's expansion rounds (from the out-degrees)
  float mean_min_w = 0.0f; // mean over the states of their cheapest finite arc weight (0: unknown)
};
// Region plan of the binned levels of the atomic sweeps (sssp_binned.h): source states in G contiguous ranges of `sg`
// states, destination states in `nbin` bins of 1 << logd; region (g -> b) holds one slot per arc from range g to bin b.
struct BinPlan {
  uint32_t nbin = 0, G = 0, sg = 0, logd = 13;
  uint64_t slots = 0;     // message slots in all (regions rounded up to whole 128-byte lines)
  DBuf<uint32_t> roff;    // [nbin*G + 1] destination-major
  DBuf<uint32_t> roff_t;  // [G*nbin]     source-major copy
};
}  // namespace wfst

namespace wfst {
// 4th word of a device state record (DeviceCsr::srec): what the compose filters' set_state needs to know
constexpr uint32_t SREC_NO_OEPS = 1u;   // no arc with olabel 0
constexpr uint32_t SREC_ALL_OEPS = 2u;  // every arc has olabel 0 (vacuously true without arcs)
constexpr uint32_t SREC_NO_IEPS = 4u;   // no arc with ilabel 0
constexpr uint32_t SREC_ALL_IEPS = 8u;  // every arc has ilabel 0
}  // namespace wfst

struct HostCsr {
  std::vector<uint32_t> offsets;
  std::vector<wfst_tr> arcs;
  std::vector<float> finals;
};

struct wfst_fst {
  // (first member = destroyed last: the caches below release their buffers into it)
  std::shared_ptr<wfst::DevicePool> owner_pool;
  int device = 0;
  wfst_ctx* ctx = nullptr;  // the creating context: only dereferenced inside API calls, which need it alive anyway
  uint32_t n_states = 0;
  uint64_t n_arcs = 0;
  int64_t start = -1;
  uint64_t props = 0;
  bool has_host = false, has_dev = false;
  // arc-weight statistics gathered at upload (steer the near-far schedule of the relaxation)
  float mean_weight = 0.0f;   // mean of the finite arc weights
  bool has_negative = false;  // some arc weight < 0
  HostCsr host;
  DeviceCsr dev;
  // A linear path whose host arrays have not been built (has_host and has_dev both false): the result of a fused batch.
  // Its n_arcs arcs sit at path_arcs inside path_block — the pinned block the kernel wrote the whole batch's results into,
  // shared by them — and path_final is the final weight of the path's end (state 0 of the path FST, shortest_path.rs:257-272).
  // ensure_host() builds the three arrays on first use and lets go of the block.
  std::shared_ptr<wfst::PinnedBlock> path_block;
  const wfst_tr* path_arcs = nullptr;
  float path_final = 0.0f;
  bool path_form = false;
  // reverse(fst) (reverse.rs:33-87) as host CSR: built on the GPU on first use by the n>1 shortest-path search
  mutable std::shared_ptr<wfst::RevFst> rev_host;
  // transpose (in-arcs as {source state, arc position}) for the shortest-path backtrace; built on the second
  // shortest_path query of a large FST (sssp.hip reverse_csr)
  mutable std::shared_ptr<wfst::RevCsr> rev_dev;
  // region plan of the mailbox relaxation sweeps; depends on (source, target) pairs only, built on first use
  mutable std::shared_ptr<wfst::MboxPlan> mbox;
  mutable std::shared_ptr<wfst::MboxPlan> mbox13;  // the same with blocks of 8192 states (resident launches of 1M .. 2M-state FSTs)
  mutable std::shared_ptr<wfst::BinPlan> binplan;  // region plan of the binned levels (FSTs beyond the mailbox range)
  // a linear, epsilon-free, single-final acceptor ("string": utils::acceptor, labels_to_fst.rs:111-132), detected at
  // upload from the host arrays; such an fst1 takes the specialised string o T kernel of the fused batch
  bool is_string = false;
  // per-arc {arc begin, arc count} of the destination state (8 B per arc) and whether any arc has ilabel 0: derived
  // lazily when the FST first serves as fst2 of the string o T kernel (compose.hip), cached under cache_mu
  mutable std::shared_ptr<wfst::DBuf<uint2>> anext;
  mutable int ieps_state = 0;  // 0 unknown, 1 no input epsilons, 2 has input epsilons
  mutable std::atomic<uint32_t> sp_queries{0};
  mutable std::atomic<uint32_t> last_sweeps{0};  // sweeps the last relaxation of this FST needed (sizes the first graph replay)
  // mailbox sweeps: bit k set = launch k of the last solve was NOT a busy wide sweep (idle, hand-over or narrow launch):
  // the next solve gates that launch's bulk loads behind its mode / sleep decision
  mutable std::atomic<uint64_t> last_hint_mask{~0ull};
  mutable std::atomic<uint32_t> stable_sweeps{0};  // consecutive solves that needed exactly last_sweeps launches
  // the lazily built caches above may be requested from several contexts (threads) at once: built under this lock,
  // with buffers taken from the OWNER context's pool (this->ctx), which outlives the handle
  mutable std::mutex cache_mu;
};

namespace wfst {
// fst_store.hip
void ensure_device(wfst_fst* f);             // upload host copy if needed (mutates cache only)
void ensure_host(const wfst_fst* f);         // download if needed
wfst_fst* make_host_fst(wfst_ctx* ctx, uint32_t n_states, int64_t start, uint64_t props, HostCsr&& csr);
wfst_fst* upload_from_host(wfst_ctx* ctx, uint32_t n_states, int64_t start, const uint32_t* offsets,
                           const wfst_tr* arcs, const float* finals, uint64_t props);
wfst_fst* upload_from_device(wfst_ctx* ctx, uint32_t n_states, int64_t start, const uint32_t* d_offsets,
                             const wfst_tr* d_arcs, const float* d_finals, uint64_t props);
void upload_many(wfst_ctx* ctx, size_t n, const uint32_t* n_states, const int64_t* starts, const uint32_t* offsets_cat,
                 const wfst_tr* arcs_cat, const float* finals_cat, const uint64_t* props, wfst_fst** outs);
// adopt freshly produced device arrays (compose output) as a new FST; arrays are copied into one arena
wfst_fst* adopt_device(wfst_ctx* ctx, uint32_t n_states, uint64_t n_arcs, int64_t start, uint64_t props,
                       const uint32_t* d_offsets, const wfst_tr* d_arcs, const float* d_finals);
// ... of the m results of one batch in one allocation and one synchronisation (fst_store.hip)
struct AdoptDesc {
  uint32_t n_states;
  uint64_t n_arcs;
  int64_t start;
  uint64_t props;
  const uint32_t* off;
  const wfst_tr* arcs;
  const float* fin;
};
void adopt_device_many(wfst_ctx* ctx, size_t m, const AdoptDesc* descs, wfst_fst** outs);
// openfst_io.cpp
wfst_fst* fst_from_openfst_bytes(wfst_ctx* ctx, const uint8_t* data, size_t len);
void fst_to_openfst_bytes(const wfst_fst* f, std::vector<uint8_t>& out);
void fst_to_openfst_const_bytes(const wfst_fst* f, std::vector<uint8_t>& out);
// sssp.hip
wfst_fst* shortest_path_n1(wfst_ctx* ctx, const wfst_fst* f);
wfst_sp_job* shortest_path_n1_begin(wfst_ctx* ctx, const wfst_fst* f);
wfst_fst* shortest_path_n1_end(wfst_sp_job* job);
void shortest_path_n1_abandon(wfst_sp_job* job);
wfst_ctx* sp_job_ctx(wfst_sp_job* job);
void shortest_distance(wfst_ctx* ctx, const wfst_fst* f, float* distance, uint32_t* hops);
// nshortest.hip
wfst_fst* shortest_path_nbest(wfst_ctx* ctx, const wfst_fst* f, uint64_t nshortest, float delta, bool unique = false);
// nbest_batch.hip
void shortest_path_nbest_batch(wfst_ctx* ctx, const wfst_fst* const* fsts, size_t n, uint64_t nshortest, float delta, wfst_fst** outs);
wfst_fst* reverse_fst(wfst_ctx* ctx, const wfst_fst* f);
// per-arc destination ranges / input-epsilon check of an FST used as fst2 of the string o T kernel (cached on the handle)
const uint2* ensure_anext(wfst_ctx* ctx, const wfst_fst* f);
bool has_input_epsilons(wfst_ctx* ctx, const wfst_fst* f);
bool detect_string(uint32_t n_states, int64_t start, const uint32_t* offsets, const wfst_tr* arcs, const float* finals);
// tr_sort.hip
void tr_sort_device(wfst_ctx* ctx, wfst_fst* f, bool ilabel_cmp);
// fst_store.hip
void project_device(wfst_ctx* ctx, wfst_fst* f, bool project_output);
// compose.hip
wfst_fst* compose(wfst_ctx* ctx, const wfst_fst* f1, const wfst_fst* f2, bool connect, uint32_t filter = 0);
void compose_shortest_path_batch(wfst_ctx* ctx, const wfst_fst* const* accs, size_t n, const wfst_fst* t, bool connect,
                                 wfst_fst** outs, uint64_t* composed_arcs, uint32_t filter = 0);
wfst_batch_job* compose_shortest_path_batch_begin(wfst_ctx* ctx, const wfst_fst* const* accs, size_t n, const wfst_fst* t,
                                                  uint32_t filter = 0);
// `sink`: the results as fixed-size records (layout of wfst_fst_pack_paths) instead of handles (outs may then be null)
struct PackedSink {
  uint32_t max_arcs;
  uint32_t* out;
};
void compose_shortest_path_batch_end(wfst_batch_job* job, wfst_fst** outs, uint64_t* composed_arcs, const PackedSink* sink = nullptr);
// one record of wfst_fst_pack_paths: [n_arcs, final-weight bits, valid, 0] + max_arcs arcs, zero padded
void pack_path_record(uint32_t* rec, uint32_t max_arcs, bool valid, uint32_t n_arcs, float final_weight, const wfst_tr* arcs);
void pack_path_record(uint32_t* rec, uint32_t max_arcs, const wfst_fst* path);
wfst_fst* make_path_fst(wfst_ctx* ctx, bool has_path, uint32_t hops, float final_weight, const wfst_tr* path_arcs);  // fst_store.hip
// nbest_batch.hip: nshortest == 1 for many small FSTs in one launch (one wavefront each); the others one after the other
void shortest_path_n1_batch(wfst_ctx* ctx, const wfst_fst* const* fsts, size_t n, wfst_fst** outs, bool lone = false);
bool shortest_path_n1_tiny(wfst_ctx* ctx, const wfst_fst* f, wfst_fst** out);  // a lone tiny FST: one wavefront (false: not applicable)
// nbest_batch.hip: distances + CSR of many small device FSTs in one launch (for host-side stages over a batch)
struct SmallFstExport {
  bool ok = false;
  std::vector<float> dist;
  HostCsr csr;
};
constexpr uint32_t SMALL_FST_MAX_STATES = 4096, SMALL_FST_MAX_ARCS = 16384;
void export_small_with_distances(wfst_ctx* ctx, const wfst_fst* const* fsts, const std::vector<size_t>& idx,
                                 std::vector<SmallFstExport>& out);
// nshortest.hip: nshortest > 1 with unique = true for a batch (small inputs: one launch + host threads)
void shortest_path_nbest_unique_batch(wfst_ctx* ctx, const wfst_fst* const* fsts, size_t n, uint64_t nshortest, float delta, wfst_fst** outs);
void compose_shortest_path_batch_abandon(wfst_batch_job* job);
wfst_ctx* batch_job_ctx(const wfst_batch_job* job);
// compose_wide.hip
wfst_fst* connect_fst(wfst_ctx* ctx, const wfst_fst* f);
wfst_fst* connect_and_adopt(wfst_ctx* ctx, uint32_t n, int64_t start, const uint32_t* off, const wfst_tr* arcs, const float* fin,
                            bool all_accessible, uint64_t out_props);
// rm_epsilon.hip
wfst_fst* rm_epsilon_fst(wfst_ctx* ctx, const wfst_fst* f);
wfst_fst* compose_wide(wfst_ctx* ctx, const wfst_fst* f1, const wfst_fst* f2, uint32_t mode, uint32_t filter, bool connect,
                       uint64_t out_props, uint64_t est_s);
}  // namespace wfst
