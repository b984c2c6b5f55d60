// api.cpp — the extern "C" surface declared in include/wfst.h (conventions: rustfst-ffi/src/lib.rs:29-97).
#include <cstdlib>

#include "common.h"
#include "lookahead.h"
#include "fst_props.h"

namespace wfst {

static thread_local std::string t_last_error;  // LAST_ERROR, rustfst-ffi/src/lib.rs:39-41
static thread_local bool t_has_error = false;
void set_last_error(const std::string& msg) {
  t_last_error = msg;
  t_has_error = true;
  if (std::getenv("WFST_FFI_ERROR_STDERR")) std::fprintf(stderr, "%s\n", msg.c_str());  // cf. lib.rs:48-50
}

// ---------------------------------------------------------------- DevicePool
size_t DevicePool::bucket(size_t bytes) {
  if (bytes < 256) return 256;
  // round up to a multiple of 1/8 of the enclosing power of two: <= 12.5 % slack, few distinct sizes
  size_t p = 256;
  while (p < bytes) p <<= 1;
  size_t step = p >> 3;
  return (bytes + step - 1) / step * step;
}
void* DevicePool::alloc(size_t bytes) {
  std::lock_guard<std::mutex> lk(mu_);
  size_t b = bucket(bytes);
  // LIFO inside a bucket: the block freed LAST is handed out first.  A solve frees its buffers in the reverse order of their
  // allocation, so every buffer of the next solve gets the block it had in the last one — two buffers of one bucket (the
  // mailbox kernels' two message arrays are both 160 MB for the benched T) otherwise SWAP blocks from solve to solve, and a
  // resident launch whose region headers sit where the other array was last time finds none of them in the Infinity Cache:
  // 280 -> 265 us per solve of C3 (LAB_NOTEBOOK.md, round 5).
  auto range = free_.equal_range(b);
  if (range.first != range.second) {
    auto it = std::prev(range.second);
    void* p = it->second;
    free_.erase(it);
    live_[p] = b;
    return p;
  }
  // A LARGE request (>= 64 MB) with no block of its own bucket takes the smallest cached block of up to twice its size instead
  // of a fresh hipMalloc (0.1-0.7 s for the multi-GB arenas of the wide compose driver: its second call asks for "last result
  // + 1/8" while the pool holds the first call's grown arena, a little larger — profiles/r06d_wide_lookahead.md).  Small
  // requests keep the exact-bucket rule above: their placement is what the relaxation's launches are tuned on.
  if (b >= (64u << 20)) {
    auto it = free_.upper_bound(b);
    if (it != free_.end() && it->first <= 2 * b) {
      void* p = it->second;
      const size_t real = it->first;
      free_.erase(it);
      live_[p] = real;
      return p;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, b);
  if (e != hipSuccess) {
    trim_locked();
    HIP_CHECK(hipMalloc(&p, b));
  }
  live_[p] = b;
  return p;
}
void DevicePool::free(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(mu_);
  auto it = live_.find(p);
  if (it == live_.end()) return;
  free_.emplace(it->second, p);
  live_.erase(it);
}
void DevicePool::trim_locked() {
  for (auto& kv : free_) (void)hipFree(kv.second);
  free_.clear();
}
void DevicePool::trim() {
  std::lock_guard<std::mutex> lk(mu_);
  trim_locked();
}
DevicePool::~DevicePool() {
  trim();
  for (auto& kv : live_) (void)hipFree(kv.first);
}

void* PinnedBuf::get(size_t bytes) {
  if (bytes > cap) {
    if (p) (void)hipHostFree(p);
    size_t ncap = std::max<size_t>(bytes, cap * 2);
    ncap = std::max<size_t>(ncap, 4096);
    HIP_CHECK(hipHostMalloc(&p, ncap, hipHostMallocDefault));
    cap = ncap;
  }
  return p;
}
PinnedBuf::~PinnedBuf() {
  if (p) (void)hipHostFree(p);
}

std::shared_ptr<PinnedBlock> PinnedRing::take(size_t bytes) {
  PinnedBlock* blk = nullptr;
  {
    std::lock_guard<std::mutex> lk(mu_);
    size_t best = free_.size();
    for (size_t i = 0; i < free_.size(); ++i)
      if (free_[i]->cap >= bytes && (best == free_.size() || free_[i]->cap < free_[best]->cap)) best = i;
    if (best != free_.size()) {
      blk = free_[best];
      free_.erase(free_.begin() + (std::ptrdiff_t)best);
    }
  }
  if (!blk) {
    auto fresh = std::make_unique<PinnedBlock>();
    fresh->cap = (std::max<size_t>(bytes, 4096) + 65535) & ~(size_t)65535;
    HIP_CHECK(hipHostMalloc(&fresh->p, fresh->cap, hipHostMallocDefault));
    blk = fresh.release();
  }
  std::weak_ptr<PinnedRing> ring = weak_from_this();
  return std::shared_ptr<PinnedBlock>(blk, [ring](PinnedBlock* b) {
    if (auto r = ring.lock()) {
      std::lock_guard<std::mutex> lk(r->mu_);
      if (r->free_.size() < 8) {
        r->free_.push_back(b);
        return;
      }
    }
    (void)hipHostFree(b->p);
    delete b;
  });
}
PinnedRing::~PinnedRing() {
  for (PinnedBlock* b : free_) {
    (void)hipHostFree(b->p);
    delete b;
  }
}

}  // namespace wfst

DeviceArena::~DeviceArena() {
  if (base && pool) pool->free(base);
}

using namespace wfst;

static wfst_ctx* ctx_create(int device, void* stream, bool own, const uint32_t* cu_mask = nullptr, uint32_t mask_words = 0) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    throw Error("libwfst_amd: no HIP device available (this engine has no CPU fallback)");
  if (device < 0 || device >= count) throw Error("libwfst_amd: device index out of range");
  HIP_CHECK(hipSetDevice(device));
  auto ctx = std::make_unique<wfst_ctx>();
  ctx->device = device;
  if (own && cu_mask && mask_words) {
    HIP_CHECK(hipExtStreamCreateWithCUMask(&ctx->stream, mask_words, cu_mask));
    ctx->owns_stream = true;
  } else if (own) {
    HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->owns_stream = true;
  } else {
    ctx->stream = (hipStream_t)stream;
  }
  ctx->pool = std::make_shared<DevicePool>(device);
  HIP_CHECK(hipEventCreate(&ctx->ev0));
  HIP_CHECK(hipEventCreate(&ctx->ev1));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (cu_mask && mask_words) {  // grids are sized for the CUs this context's stream may use
    int bits = 0;
    for (uint32_t w = 0; w < mask_words; ++w) bits += __builtin_popcount(cu_mask[w]);
    if (bits <= 0) throw Error("empty CU mask");
    ctx->n_cus = std::min(ctx->n_cus, bits);
  }
  return ctx.release();
}

extern "C" {

uint32_t wfst_abi_version(void) { return WFST_ABI_VERSION; }

wfst_status wfst_last_error(char** msg) {
  return wrap([&] {
    if (!msg) throw Error("null out pointer");
    std::string s = t_has_error ? t_last_error : std::string("No error message");
    t_has_error = false;
    char* out = (char*)std::malloc(s.size() + 1);
    if (!out) throw Error("out of memory");
    std::memcpy(out, s.c_str(), s.size() + 1);
    *msg = out;
  });
}
wfst_status wfst_string_destroy(char* msg) {
  std::free(msg);
  return WFST_OK;
}
wfst_status wfst_bytes_destroy(uint8_t* data) {
  std::free(data);
  return WFST_OK;
}

wfst_status wfst_ctx_create(int device, wfst_ctx** out) {
  return wrap([&] {
    if (!out) throw Error("null out pointer");
    *out = ctx_create(device, nullptr, true);
  });
}
wfst_status wfst_ctx_create_on_stream(int device, void* hip_stream, wfst_ctx** out) {
  return wrap([&] {
    if (!out) throw Error("null out pointer");
    *out = ctx_create(device, hip_stream, false);
  });
}
wfst_status wfst_ctx_create_with_cu_mask(int device, const uint32_t* cu_mask, uint32_t mask_words, wfst_ctx** out) {
  return wrap([&] {
    if (!out || !cu_mask || !mask_words) throw Error("null pointer");
    *out = ctx_create(device, nullptr, true, cu_mask, mask_words);
  });
}
wfst_status wfst_ctx_destroy(wfst_ctx* ctx) {
  return wrap([&] {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& g : ctx->sweep_graph) {
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->ev_chain)
      if (e) (void)hipEventDestroy(e);
    ctx->pool.reset();
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
  });
}
wfst_status wfst_ctx_synchronize(wfst_ctx* ctx) {
  return wrap([&] {
    if (!ctx) throw Error("null ctx");
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
  });
}
wfst_status wfst_ctx_stream(wfst_ctx* ctx, void** hip_stream) {
  return wrap([&] {
    if (!ctx || !hip_stream) throw Error("null pointer");
    *hip_stream = (void*)ctx->stream;
  });
}

wfst_status wfst_fst_upload(wfst_ctx* ctx, uint32_t n_states, int64_t start, const uint32_t* offsets,
                            const wfst_tr* arcs, const float* finals, uint64_t props, wfst_fst** out) {
  return wrap([&] {
    if (!ctx || !out) throw Error("null pointer");
    if (n_states && (!offsets || !finals)) throw Error("null array");
    *out = upload_from_host(ctx, n_states, start, offsets, arcs, finals, props);
  });
}
wfst_status wfst_fst_upload_device(wfst_ctx* ctx, uint32_t n_states, int64_t start, const uint32_t* d_offsets,
                                   const wfst_tr* d_arcs, const float* d_finals, uint64_t props, wfst_fst** out) {
  return wrap([&] {
    if (!ctx || !out) throw Error("null pointer");
    *out = upload_from_device(ctx, n_states, start, d_offsets, d_arcs, d_finals, props);
  });
}
wfst_status wfst_fst_upload_many(wfst_ctx* ctx, size_t n, const uint32_t* n_states, const int64_t* starts,
                                 const uint32_t* offsets_cat, const wfst_tr* arcs_cat, const float* finals_cat,
                                 const uint64_t* props, wfst_fst** outs) {
  return wrap([&] {
    if (!ctx || !outs) throw Error("null pointer");
    upload_many(ctx, n, n_states, starts, offsets_cat, arcs_cat, finals_cat, props, outs);
  });
}
wfst_status wfst_fst_from_openfst_bytes(wfst_ctx* ctx, const uint8_t* data, size_t len, wfst_fst** out) {
  return wrap([&] {
    if (!ctx || !out || !data) throw Error("null pointer");
    *out = fst_from_openfst_bytes(ctx, data, len);
  });
}
wfst_status wfst_fst_to_openfst_bytes(const wfst_fst* fst, uint8_t** data, size_t* len) {
  return wrap([&] {
    if (!fst || !data || !len) throw Error("null pointer");
    std::vector<uint8_t> buf;
    fst_to_openfst_bytes(fst, buf);
    uint8_t* p = (uint8_t*)std::malloc(buf.size() ? buf.size() : 1);
    if (!p) throw Error("out of memory");
    std::memcpy(p, buf.data(), buf.size());
    *data = p;
    *len = buf.size();
  });
}
wfst_status wfst_fst_to_openfst_const_bytes(const wfst_fst* fst, uint8_t** data, size_t* len) {
  return wrap([&] {
    if (!fst || !data || !len) throw Error("null pointer");
    std::vector<uint8_t> buf;
    fst_to_openfst_const_bytes(fst, buf);
    uint8_t* p = (uint8_t*)std::malloc(buf.size() ? buf.size() : 1);
    if (!p) throw Error("out of memory");
    std::memcpy(p, buf.data(), buf.size());
    *data = p;
    *len = buf.size();
  });
}
wfst_status wfst_fst_info(const wfst_fst* fst, uint32_t* n_states, uint64_t* n_arcs, int64_t* start, uint64_t* props) {
  return wrap([&] {
    if (!fst) throw Error("null fst");
    if (n_states) *n_states = fst->n_states;
    if (n_arcs) *n_arcs = fst->n_arcs;
    if (start) *start = fst->start;
    if (props) *props = fst->props;
  });
}
wfst_status wfst_fst_download(const wfst_fst* fst, uint32_t* offsets, wfst_tr* arcs, float* finals) {
  return wrap([&] {
    if (!fst) throw Error("null fst");
    ensure_host(fst);
    const HostCsr& h = fst->host;
    if (offsets) std::memcpy(offsets, h.offsets.data(), h.offsets.size() * sizeof(uint32_t));
    if (arcs && !h.arcs.empty()) std::memcpy(arcs, h.arcs.data(), h.arcs.size() * sizeof(wfst_tr));
    if (finals && !h.finals.empty()) std::memcpy(finals, h.finals.data(), h.finals.size() * sizeof(float));
  });
}
wfst_status wfst_fst_destroy(wfst_fst* fst) {
  return wrap([&] {
    if (!fst) return;
    (void)hipSetDevice(fst->device);
    delete fst;
  });
}

wfst_status wfst_fst_destroy_many(wfst_fst* const* fsts, size_t n) {
  return wrap([&] {
    if (n && !fsts) throw Error("null pointer");
    for (size_t i = 0; i < n; ++i) delete fsts[i];
  });
}

wfst_status wfst_compose(wfst_ctx* ctx, const wfst_fst* fst1, const wfst_fst* fst2, const wfst_compose_config* cfg,
                         wfst_fst** out) {
  return wrap([&] {
    if (!ctx || !fst1 || !fst2 || !out) throw Error("null pointer");
    wfst_compose_config c = cfg ? *cfg : wfst_compose_config{0, 1};  // ComposeConfig::default(), compose_static.rs:56-65
    if (c.compose_filter > 6) throw Error("unknown compose_filter " + std::to_string(c.compose_filter));  // compose.rs:20-33
    HIP_CHECK(hipSetDevice(ctx->device));
    *out = compose(ctx, fst1, fst2, c.connect != 0, c.compose_filter);
  });
}

wfst_status wfst_shortest_path(wfst_ctx* ctx, const wfst_fst* fst, const wfst_shortest_path_config* cfg,
                               wfst_fst** out) {
  return wrap([&] {
    if (!ctx || !fst || !out) throw Error("null pointer");
    wfst_shortest_path_config c = cfg ? *cfg : wfst_shortest_path_config{1e-6f, 1, 0};  // shortest_path.rs:31-39
    HIP_CHECK(hipSetDevice(ctx->device));
    if (c.nshortest == 0) {  // shortest_path.rs:118-120: FO::new()
      HostCsr h;
      h.offsets.push_back(0);
      *out = make_host_fst(ctx, 0, -1, props::NULL_PROPS, std::move(h));
      return;
    }
    // nshortest == 1 returns before `unique` is looked at (shortest_path.rs:122-133)
    if (c.nshortest == 1) {
      *out = shortest_path_n1(ctx, fst);
      return;
    }
    // unique: the reversed FST is determinized on the host first (shortest_path.rs:157-165); acceptors only, as there
    *out = shortest_path_nbest(ctx, fst, c.nshortest, c.delta, c.unique != 0);
  });
}

wfst_status wfst_shortest_path_batch(wfst_ctx* ctx, const wfst_fst* const* fsts, size_t n, const wfst_shortest_path_config* cfg,
                                     wfst_fst** outs) {
  return wrap([&] {
    if (!ctx || (n && (!fsts || !outs))) throw Error("null pointer");
    for (size_t i = 0; i < n; ++i) {
      if (!fsts[i]) throw Error("null FST in batch");
      outs[i] = nullptr;
    }
    wfst_shortest_path_config c = cfg ? *cfg : wfst_shortest_path_config{1e-6f, 1, 0};
    HIP_CHECK(hipSetDevice(ctx->device));
    try {
      if (c.nshortest >= 2 && !c.unique) {
        shortest_path_nbest_batch(ctx, fsts, n, c.nshortest, c.delta, outs);
        return;
      }
      if (c.nshortest == 1) {  // small inputs in one launch (one wavefront each), the others one after the other
        shortest_path_n1_batch(ctx, fsts, n, outs);
        return;
      }
      if (c.nshortest >= 2 && c.unique) {  // small inputs: one launch for distances + arrays, the host stages on threads
        shortest_path_nbest_unique_batch(ctx, fsts, n, c.nshortest, c.delta, outs);
        return;
      }
      for (size_t i = 0; i < n; ++i) {
        if (c.nshortest == 0) {
          HostCsr h;
          h.offsets.push_back(0);
          outs[i] = make_host_fst(ctx, 0, -1, props::NULL_PROPS, std::move(h));
        } else if (c.nshortest == 1) {
          outs[i] = shortest_path_n1(ctx, fsts[i]);
        } else {
          outs[i] = shortest_path_nbest(ctx, fsts[i], c.nshortest, c.delta, true);  // unique: one after the other
        }
      }
    } catch (...) {
      for (size_t i = 0; i < n; ++i) {
        delete outs[i];
        outs[i] = nullptr;
      }
      throw;
    }
  });
}

wfst_status wfst_shortest_distance(wfst_ctx* ctx, const wfst_fst* fst, float* distance, uint32_t* hops) {
  return wrap([&] {
    if (!ctx || !fst || !distance) throw Error("null pointer");
    HIP_CHECK(hipSetDevice(ctx->device));
    shortest_distance(ctx, fst, distance, hops);
  });
}

wfst_status wfst_reverse(wfst_ctx* ctx, const wfst_fst* fst, wfst_fst** out) {
  return wrap([&] {
    if (!ctx || !fst || !out) throw Error("null pointer");
    HIP_CHECK(hipSetDevice(ctx->device));
    *out = reverse_fst(ctx, fst);
  });
}

wfst_status wfst_fst_tr_sort(wfst_ctx* ctx, wfst_fst* fst, int ilabel_cmp) {
  return wrap([&] {
    if (!ctx || !fst) throw Error("null pointer");
    tr_sort_device(ctx, fst, ilabel_cmp != 0);
  });
}

wfst_status wfst_fst_set_start(wfst_ctx* ctx, wfst_fst* fst, uint32_t state) {  // mutable_fst.rs:35-44
  return wrap([&] {
    if (!ctx || !fst) throw Error("null pointer");
    if (state >= fst->n_states) throw Error("The state " + std::to_string(state) + " doesn't exist");
    std::lock_guard<std::mutex> lk(fst->cache_mu);
    if (fst->start != (int64_t)state) fst->is_string = false;  // (the string o T kernel's input is linear FROM its start state)
    fst->start = (int64_t)state;
    fst->props = props::set_start(fst->props);
    fst->rev_host.reset();  // reverse(fst) marks the old start state final (reverse.rs:80-86)
    fst->stable_sweeps.store(0, std::memory_order_relaxed);  // (the next queries are queued with a spare launch again)
    // (what a solve learned about the launch pattern of the LAST source says little about this one: the first query predicts
    // from it all the same, and a solve that outruns its prediction is continued)
  });
}

wfst_status wfst_compose_shortest_path_batch(wfst_ctx* ctx, const wfst_fst* const* acceptors, size_t n,
                                             const wfst_fst* t, const wfst_compose_config* ccfg,
                                             const wfst_shortest_path_config* scfg, wfst_fst** outs,
                                             uint64_t* composed_arcs) {
  return wrap([&] {
    if (!ctx || !t || !outs || (n && !acceptors)) throw Error("null pointer");
    wfst_compose_config c = ccfg ? *ccfg : wfst_compose_config{0, 1};
    wfst_shortest_path_config s = scfg ? *scfg : wfst_shortest_path_config{1e-6f, 1, 0};
    if (c.compose_filter > 6) throw Error("unknown compose_filter");
    if (s.nshortest != 1) throw Error("unsupported: nshortest != 1 in the fused batch");
    HIP_CHECK(hipSetDevice(ctx->device));
    compose_shortest_path_batch(ctx, acceptors, n, t, c.connect != 0, outs, composed_arcs, c.compose_filter);
  });
}

wfst_status wfst_rm_epsilon(wfst_ctx* ctx, const wfst_fst* fst, wfst_fst** out) {
  return wrap([&] {
    if (!ctx || !fst || !out) throw Error("null pointer");
    *out = nullptr;
    HIP_CHECK(hipSetDevice(ctx->device));
    *out = rm_epsilon_fst(ctx, fst);
  });
}

wfst_status wfst_connect(wfst_ctx* ctx, const wfst_fst* fst, wfst_fst** out) {
  return wrap([&] {
    if (!ctx || !fst || !out) throw Error("null pointer");
    *out = nullptr;
    HIP_CHECK(hipSetDevice(ctx->device));
    *out = connect_fst(ctx, fst);
  });
}

wfst_status wfst_fst_project(wfst_ctx* ctx, wfst_fst* fst, int project_output) {
  return wrap([&] {
    if (!ctx || !fst) throw Error("null pointer");
    project_device(ctx, fst, project_output != 0);
  });
}

wfst_status wfst_lookahead_create(wfst_ctx* ctx, const wfst_fst* fst1, wfst_lookahead** out) {
  return wrap([&] {
    if (!ctx || !fst1 || !out) throw Error("null pointer");
    *out = nullptr;
    HIP_CHECK(hipSetDevice(ctx->device));
    *out = lookahead_create(ctx, fst1);
  });
}

wfst_status wfst_lookahead_relabel(wfst_lookahead* la, const wfst_fst* fst2, wfst_fst** out) {
  return wrap([&] {
    if (!la || !fst2 || !out) throw Error("null pointer");
    *out = nullptr;
    if (!la->ctx) throw Error("host-only look-ahead handle");
    HIP_CHECK(hipSetDevice(la->ctx->device));
    *out = lookahead_relabel(la, fst2);
  });
}

wfst_status wfst_lookahead_fst1(const wfst_lookahead* la, const wfst_fst** out) {
  return wrap([&] {
    if (!la || !out) throw Error("null pointer");
    if (!la->fst1) throw Error("host-only look-ahead handle");
    *out = la->fst1;
  });
}

wfst_status wfst_compose_lookahead(wfst_ctx* ctx, const wfst_lookahead* la, const wfst_fst* relabeled_fst2, wfst_fst** out) {
  return wrap([&] {
    if (!ctx || !la || !relabeled_fst2 || !out) throw Error("null pointer");
    *out = nullptr;
    HIP_CHECK(hipSetDevice(ctx->device));
    *out = compose_lookahead(ctx, la, relabeled_fst2);
  });
}

wfst_status wfst_compose_lookahead_batch(wfst_ctx* ctx, const wfst_lookahead* la, const wfst_fst* const* relabeled_fst2s,
                                         size_t n, wfst_fst** outs) {
  return wrap([&] {
    if (!ctx || !la || (n && (!relabeled_fst2s || !outs))) throw Error("null pointer");
    HIP_CHECK(hipSetDevice(ctx->device));
    compose_lookahead_batch(ctx, la, relabeled_fst2s, n, outs);
  });
}

wfst_status wfst_lookahead_destroy(wfst_lookahead* la) {
  return wrap([&] {
    if (!la) return;
    if (la->ctx) (void)hipSetDevice(la->ctx->device);
    delete la;
  });
}

wfst_status wfst_lookahead_info(const wfst_lookahead* la, uint32_t* n_states, uint64_t* n_intervals, uint32_t* n_labels,
                                uint32_t* final_label) {
  return wrap([&] {
    if (!la) throw Error("null pointer");
    if (n_states) *n_states = (uint32_t)(la->data.iv_off.size() - 1);
    if (n_intervals) *n_intervals = la->data.iv.size() / 2;
    if (n_labels) *n_labels = (uint32_t)la->data.label2index.size();
    if (final_label) *final_label = la->data.final_label;
  });
}

wfst_status wfst_lookahead_download(const wfst_lookahead* la, uint32_t* interval_offsets, uint32_t* intervals,
                                    uint32_t* labels, uint32_t* indices) {
  return wrap([&] {
    if (!la) throw Error("null pointer");
    if (interval_offsets) std::copy(la->data.iv_off.begin(), la->data.iv_off.end(), interval_offsets);
    if (intervals) std::copy(la->data.iv.begin(), la->data.iv.end(), intervals);
    if (labels || indices) {
      std::vector<std::pair<uint32_t, uint32_t>> v(la->data.label2index.begin(), la->data.label2index.end());
      std::sort(v.begin(), v.end());
      for (size_t i = 0; i < v.size(); ++i) {
        if (labels) labels[i] = v[i].first;
        if (indices) indices[i] = v[i].second;
      }
    }
  });
}

wfst_status wfst_label_reachable_compute(uint32_t n_states, const uint32_t* offsets, const wfst_tr* arcs, const float* finals,
                                         int reach_input, wfst_lookahead** out) {
  return wrap([&] {
    if (!offsets || !out || (n_states && !finals) || (offsets[n_states] && !arcs)) throw Error("null pointer");
    *out = nullptr;
    for (uint32_t s = 0; s < n_states; ++s) {
      if (offsets[s] > offsets[s + 1]) throw Error("offsets are not monotone");
      for (uint32_t k = offsets[s]; k < offsets[s + 1]; ++k)
        if (arcs[k].nextstate >= n_states) throw Error("arc to a state that does not exist");
    }
    std::unique_ptr<wfst_lookahead> la(new wfst_lookahead());
    la->data.compute(n_states, offsets, arcs, finals, reach_input != 0);
    *out = la.release();
  });
}

wfst_status wfst_shortest_path_begin(wfst_ctx* ctx, const wfst_fst* fst, const wfst_shortest_path_config* cfg,
                                     wfst_sp_job** job) {
  return wrap([&] {
    if (!ctx || !fst || !job) throw Error("null pointer");
    *job = nullptr;
    wfst_shortest_path_config c = cfg ? *cfg : wfst_shortest_path_config{1e-6f, 1, 0};
    if (c.nshortest != 1) throw Error("unsupported: nshortest != 1 in the asynchronous shortest_path");
    if (ctx->sp_in_flight) throw Error("a shortest_path job is already in flight on this context");
    HIP_CHECK(hipSetDevice(ctx->device));
    *job = shortest_path_n1_begin(ctx, fst);
    ctx->sp_in_flight = true;
  });
}

wfst_status wfst_shortest_path_end(wfst_sp_job* job, wfst_fst** out) {
  return wrap([&] {
    if (!job) throw Error("null job");
    wfst_ctx* ctx = sp_job_ctx(job);
    ctx->sp_in_flight = false;
    if (!out) {
      shortest_path_n1_abandon(job);
      throw Error("null pointer");
    }
    *out = nullptr;
    HIP_CHECK(hipSetDevice(ctx->device));
    *out = shortest_path_n1_end(job);
  });
}

wfst_status wfst_compose_shortest_path_batch_begin(wfst_ctx* ctx, const wfst_fst* const* acceptors, size_t n,
                                                   const wfst_fst* t, const wfst_compose_config* ccfg,
                                                   const wfst_shortest_path_config* scfg, wfst_batch_job** job) {
  return wrap([&] {
    if (!ctx || !t || !job || (n && !acceptors)) throw Error("null pointer");
    *job = nullptr;
    wfst_compose_config c = ccfg ? *ccfg : wfst_compose_config{0, 1};
    wfst_shortest_path_config s = scfg ? *scfg : wfst_shortest_path_config{1e-6f, 1, 0};
    if (c.compose_filter > 6) throw Error("unknown compose_filter");
    if (s.nshortest != 1) throw Error("unsupported: nshortest != 1 in the fused batch");
    if (ctx->batch_in_flight) throw Error("a fused batch is already in flight on this context");
    HIP_CHECK(hipSetDevice(ctx->device));
    *job = compose_shortest_path_batch_begin(ctx, acceptors, n, t, c.compose_filter);
    ctx->batch_in_flight = true;
  });
}

wfst_status wfst_compose_shortest_path_batch_end(wfst_batch_job* job, wfst_fst** outs, uint64_t* composed_arcs) {
  return wrap([&] {
    if (!job) throw Error("null job");
    wfst_ctx* ctx = batch_job_ctx(job);
    ctx->batch_in_flight = false;
    if (!outs) {
      compose_shortest_path_batch_abandon(job);
      throw Error("null pointer");
    }
    HIP_CHECK(hipSetDevice(ctx->device));
    compose_shortest_path_batch_end(job, outs, composed_arcs);
  });
}

wfst_status wfst_fst_pack_paths(const wfst_fst* const* paths, size_t n, uint32_t max_arcs, uint32_t* out) {
  return wrap([&] {
    if ((n && !paths) || !out) throw Error("null pointer");
    const size_t rec = 4 + 4 * (size_t)max_arcs;
    for (size_t i = 0; i < n; ++i) {
      if (!paths[i]) throw Error("null path in batch");
      pack_path_record(out + i * rec, max_arcs, paths[i]);
    }
  });
}

wfst_status wfst_compose_shortest_path_batch_packed(wfst_ctx* ctx, const wfst_fst* const* acceptors, size_t n, const wfst_fst* t,
                                                    const wfst_compose_config* ccfg, const wfst_shortest_path_config* scfg,
                                                    uint32_t max_arcs, uint32_t* out, uint64_t* composed_arcs) {
  return wrap([&] {
    if (!ctx || !t || (n && (!acceptors || !out))) throw Error("null pointer");
    wfst_compose_config c = ccfg ? *ccfg : wfst_compose_config{0, 1};
    wfst_shortest_path_config s = scfg ? *scfg : wfst_shortest_path_config{1e-6f, 1, 0};
    if (c.compose_filter > 6) throw Error("unknown compose_filter");
    if (s.nshortest != 1) throw Error("unsupported: nshortest != 1 in the fused batch");
    if (ctx->batch_in_flight) throw Error("a batch is in flight on this context");
    HIP_CHECK(hipSetDevice(ctx->device));
    const PackedSink sink{max_arcs, out};
    compose_shortest_path_batch_end(compose_shortest_path_batch_begin(ctx, acceptors, n, t, c.compose_filter), nullptr, composed_arcs,
                                    &sink);
  });
}

wfst_status wfst_ctx_set_tie_order(wfst_ctx* ctx, int reference_order) {
  return wrap([&] {
    if (!ctx) throw Error("null ctx");
    ctx->tie_reference = reference_order != 0;
  });
}
wfst_status wfst_ctx_set_resident_share(wfst_ctx* ctx, uint32_t share) {
  return wrap([&] {
    if (!ctx) throw Error("null ctx");
    if (share > 1) throw Error("wfst_ctx_set_resident_share: 0 (whole device) or 1 (half)");
    ctx->resident_share = share;
  });
}
wfst_status wfst_ctx_set_profiling(wfst_ctx* ctx, int on) {
  return wrap([&] {
    if (!ctx) throw Error("null ctx");
    ctx->profiling = on == 1;
    ctx->chain_timing = on == 2;
    if (ctx->chain_timing && !ctx->ev_chain[0]) {
      HIP_CHECK(hipEventCreate(&ctx->ev_chain[0]));
      HIP_CHECK(hipEventCreate(&ctx->ev_chain[1]));
    }
  });
}
wfst_status wfst_ctx_get_stats(wfst_ctx* ctx, wfst_stats* out) {
  return wrap([&] {
    if (!ctx || !out) throw Error("null pointer");
    *out = ctx->stats;
  });
}
wfst_status wfst_ctx_get_sweep_trace(wfst_ctx* ctx, double* ms, uint64_t* arcs, uint64_t* states, size_t cap, size_t* n) {
  return wrap([&] {
    if (!ctx || !n) throw Error("null pointer");
    *n = ctx->sweep_trace.size();
    for (size_t i = 0; i < std::min(cap, *n); ++i) {
      if (ms) ms[i] = ctx->sweep_trace[i].ms;
      if (arcs) arcs[i] = ctx->sweep_trace[i].arcs;
      if (states) states[i] = ctx->sweep_trace[i].states;
    }
  });
}
wfst_status wfst_ctx_get_sweep_modes(wfst_ctx* ctx, uint32_t* modes, size_t cap, size_t* n) {
  return wrap([&] {
    if (!ctx || !n) throw Error("null pointer");
    *n = ctx->sweep_trace.size();
    for (size_t i = 0; i < std::min(cap, *n); ++i)
      if (modes) modes[i] = ctx->sweep_trace[i].mode;
  });
}
wfst_status wfst_ctx_reset_stats(wfst_ctx* ctx) {
  return wrap([&] {
    if (!ctx) throw Error("null ctx");
    ctx->stats = wfst_stats{};
  });
}

}  // extern "C"
