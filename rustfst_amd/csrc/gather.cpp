// gather.cpp — the multi-GPU step of the batch path behind the C-ABI: acceptor i -> GPU i mod G, T
// replicated, NO collective during compute; the only exchange is an all-gather of the finished results over RCCL/xGMI.
//
// The reference has no distributed code (rustfst is single-process), so there is no reference interface to mirror: these
// entry points are what a Rust host with one process (or thread) per GPU binds next to compose / shortest_path
// (INTEGRATION.md "8 GPUs from Rust").  librccl is opened at first use (dlopen), so the library still loads — and every
// single-GPU entry point works — where RCCL is not installed; a gather there fails loudly.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.h"

namespace {

using namespace wfst;

// the few RCCL entry points used, by their stable C signatures (rccl.h: ncclUniqueId is 128 opaque bytes, ncclComm_t an
// opaque pointer, ncclResult_t / ncclDataType_t ints with ncclSuccess = 0 and ncclInt8 = ncclChar = 0)
struct UniqueId {
  char internal[128];
};
using Comm = void*;
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // (a process that has PyTorch-ROCm loaded already maps an RCCL of the same SONAME: dlopen returns that copy)
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.so) break;
    }
    if (!r.so) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.so, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
  });
  if (!r.so || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather)
    throw Error("libwfst_amd: librccl is not available (multi-GPU gather needs RCCL; every single-GPU entry point works without it)");
  return r;
}

void rccl_check(int rc, const char* what) {
  if (rc == 0) return;
  Rccl& r = rccl();
  throw Error(std::string("RCCL error in ") + what + ": " + (r.GetErrorString ? r.GetErrorString(rc) : std::to_string(rc).c_str()));
}

}  // namespace

// One communicator per (process or thread, GPU): an RCCL communicator, a stream of its own (the exchange must not queue
// behind the relaxation sweeps of the context's stream, nor in front of the next solve's), an event, and two sets of
// staging buffers (pinned host in / out, device in / out) that alternate, so that one exchange may be in flight while
// the next is being packed.
struct wfst_comm {
  wfst_ctx* ctx = nullptr;
  int device = 0;
  uint32_t rank = 0, world = 1;
  Comm comm = nullptr;
  hipStream_t stream = nullptr;
  struct Set {
    void* h_in = nullptr;
    void* h_out = nullptr;
    void* d_in = nullptr;
    void* d_out = nullptr;
    size_t cap = 0;  // bytes per rank the set is sized for
    hipEvent_t done = nullptr;
    size_t bytes = 0;     // bytes per rank of the exchange in flight
    bool in_flight = false;
  } sets[2];
  hipEvent_t order_ev = nullptr;  // wfst_comm_order_after
  // host transport (wfst_comm_create_host): the exchange is the caller's all-gather of host buffers (MPI, gloo, a test
  // harness) — no RCCL, no device, no stream; everything above the transport (staging sets, record layout, ragged
  // gathers) is the same code
  wfst_allgather_fn host_fn = nullptr;
  void* host_user = nullptr;
  int next = 0, pending = -1;
  uint32_t paths_n = 0, paths_max_arcs = 0;  // shape of a wfst_gather_paths_begin in flight
};

namespace {

void set_reserve(wfst_comm* c, wfst_comm::Set& s, size_t bytes) {
  if (bytes <= s.cap) return;
  const size_t cap = std::max<size_t>(bytes, std::max<size_t>(2 * s.cap, 4096));
  if (c->host_fn) {
    std::free(s.h_in);
    std::free(s.h_out);
    s.h_in = s.h_out = nullptr;
    s.cap = 0;
    s.h_in = std::malloc(cap);
    s.h_out = std::malloc(cap * c->world);
    if (!s.h_in || !s.h_out) throw Error("out of memory");
    s.cap = cap;
    return;
  }
  if (s.h_in) (void)hipHostFree(s.h_in);
  if (s.h_out) (void)hipHostFree(s.h_out);
  if (s.d_in) (void)hipFree(s.d_in);
  if (s.d_out) (void)hipFree(s.d_out);
  s.h_in = s.h_out = s.d_in = s.d_out = nullptr;
  s.cap = 0;
  HIP_CHECK(hipHostMalloc(&s.h_in, cap, hipHostMallocDefault));
  HIP_CHECK(hipHostMalloc(&s.h_out, cap * c->world, hipHostMallocDefault));
  HIP_CHECK(hipMalloc(&s.d_in, cap));
  HIP_CHECK(hipMalloc(&s.d_out, cap * c->world));
  s.cap = cap;
}

// queues H2D -> all-gather -> D2H of `bytes` bytes per rank (already in s.h_in) on the communicator's stream
void queue_exchange(wfst_comm* c, wfst_comm::Set& s, size_t bytes) {
  if (c->host_fn) {  // (synchronous: the transport returns with every rank's block in place)
    if (c->host_fn(c->host_user, s.h_in, s.h_out, bytes) != 0) throw Error("wfst_comm: the host transport's all-gather failed");
    s.bytes = bytes;
    s.in_flight = true;
    return;
  }
  Rccl& r = rccl();
  HIP_CHECK(hipMemcpyAsync(s.d_in, s.h_in, bytes, hipMemcpyHostToDevice, c->stream));
  rccl_check(r.AllGather(s.d_in, s.d_out, bytes, /*ncclInt8*/ 0, c->comm, c->stream), "ncclAllGather");
  HIP_CHECK(hipMemcpyAsync(s.h_out, s.d_out, bytes * c->world, hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipEventRecord(s.done, c->stream));
  s.bytes = bytes;
  s.in_flight = true;
}

wfst_comm::Set& begin_set(wfst_comm* c, size_t bytes) {
  if (c->pending >= 0) throw Error("wfst_comm: an exchange is already in flight (call the matching _end first)");
  wfst_comm::Set& s = c->sets[c->next];
  set_reserve(c, s, bytes);
  return s;
}

const void* end_set(wfst_comm* c, size_t* bytes) {
  if (c->pending < 0) throw Error("wfst_comm: no exchange in flight");
  wfst_comm::Set& s = c->sets[c->pending];
  if (!c->host_fn) HIP_CHECK(hipEventSynchronize(s.done));
  s.in_flight = false;
  c->pending = -1;
  *bytes = s.bytes;
  return s.h_out;
}

}  // namespace

extern "C" {

wfst_status wfst_comm_unique_id(uint8_t* id) {
  return wrap([&] {
    if (!id) throw Error("null pointer");
    static_assert(WFST_COMM_ID_BYTES == sizeof(UniqueId), "ncclUniqueId is 128 bytes");
    UniqueId u;
    rccl_check(rccl().GetUniqueId(&u), "ncclGetUniqueId");
    std::memcpy(id, &u, sizeof(u));
  });
}

wfst_status wfst_comm_create(wfst_ctx* ctx, const uint8_t* id, uint32_t rank, uint32_t world, wfst_comm** out) {
  return wrap([&] {
    if (!ctx || !id || !out) throw Error("null pointer");
    if (world == 0 || rank >= world) throw Error("wfst_comm_create: rank out of range");
    HIP_CHECK(hipSetDevice(ctx->device));
    auto c = std::make_unique<wfst_comm>();
    c->ctx = ctx;
    c->device = ctx->device;
    c->rank = rank;
    c->world = world;
    UniqueId u;
    std::memcpy(&u, id, sizeof(u));
    rccl_check(rccl().CommInitRank(&c->comm, (int)world, u, (int)rank), "ncclCommInitRank");
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto& s : c->sets) HIP_CHECK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    *out = c.release();
  });
}

wfst_status wfst_comm_create_host(uint32_t rank, uint32_t world, wfst_allgather_fn fn, void* user, wfst_comm** out) {
  return wrap([&] {
    if (!fn || !out) throw Error("null pointer");
    if (world == 0 || rank >= world) throw Error("wfst_comm_create_host: rank out of range");
    auto c = std::make_unique<wfst_comm>();
    c->rank = rank;
    c->world = world;
    c->host_fn = fn;
    c->host_user = user;
    *out = c.release();
  });
}

wfst_status wfst_comm_info(const wfst_comm* comm, uint32_t* rank, uint32_t* world) {
  return wrap([&] {
    if (!comm) throw Error("null comm");
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
  });
}

wfst_status wfst_comm_destroy(wfst_comm* c) {
  return wrap([&] {
    if (!c) return;
    if (c->host_fn) {
      for (auto& s : c->sets) {
        std::free(s.h_in);
        std::free(s.h_out);
      }
      delete c;
      return;
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    for (auto& s : c->sets) {
      if (s.h_in) (void)hipHostFree(s.h_in);
      if (s.h_out) (void)hipHostFree(s.h_out);
      if (s.d_in) (void)hipFree(s.d_in);
      if (s.d_out) (void)hipFree(s.d_out);
      if (s.done) (void)hipEventDestroy(s.done);
    }
    if (c->order_ev) (void)hipEventDestroy(c->order_ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
  });
}

wfst_status wfst_comm_order_after(wfst_comm* c, wfst_ctx* ctx) {
  return wrap([&] {
    if (!c || !ctx) throw Error("null pointer");
    if (c->host_fn) return;  // (a host transport queues nothing on a stream)
    if (ctx->device != c->device) throw Error("wfst_comm_order_after: the context lives on another GPU");
    HIP_CHECK(hipSetDevice(c->device));
    if (!c->order_ev) HIP_CHECK(hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(c->order_ev, ctx->stream));
    HIP_CHECK(hipStreamWaitEvent(c->stream, c->order_ev, 0));
  });
}

wfst_status wfst_comm_allgather_begin(wfst_comm* c, const void* send, size_t bytes) {
  return wrap([&] {
    if (!c || (bytes && !send)) throw Error("null pointer");
    if (!c->host_fn) HIP_CHECK(hipSetDevice(c->device));
    wfst_comm::Set& s = begin_set(c, std::max<size_t>(bytes, 1));
    if (bytes) std::memcpy(s.h_in, send, bytes);
    queue_exchange(c, s, bytes);
    c->pending = c->next;
    c->next ^= 1;
    c->paths_n = 0;
  });
}

wfst_status wfst_comm_allgather_end(wfst_comm* c, void* recv) {
  return wrap([&] {
    if (!c || !recv) throw Error("null pointer");
    size_t bytes = 0;
    const void* h = end_set(c, &bytes);
    std::memcpy(recv, h, bytes * c->world);
  });
}

wfst_status wfst_gather_paths_begin(wfst_comm* c, const wfst_fst* const* paths, size_t n, uint32_t max_arcs) {
  return wrap([&] {
    if (!c || (n && !paths)) throw Error("null pointer");
    if (!c->host_fn) HIP_CHECK(hipSetDevice(c->device));
    const size_t bytes = n * (4 + 4 * (size_t)max_arcs) * sizeof(uint32_t);
    wfst_comm::Set& s = begin_set(c, std::max<size_t>(bytes, 1));
    // the records are packed straight into the pinned send buffer (same layout as wfst_fst_pack_paths)
    if (wfst_fst_pack_paths(paths, n, max_arcs, (uint32_t*)s.h_in) != WFST_OK) {
      char* msg = nullptr;
      (void)wfst_last_error(&msg);
      std::string m = msg ? msg : "wfst_fst_pack_paths failed";
      if (msg) (void)wfst_string_destroy(msg);
      throw Error(m);
    }
    queue_exchange(c, s, bytes);
    c->pending = c->next;
    c->next ^= 1;
    c->paths_n = (uint32_t)n;
    c->paths_max_arcs = max_arcs;
  });
}

wfst_status wfst_gather_records_begin(wfst_comm* c, const uint32_t* records, size_t n, uint32_t max_arcs) {
  return wrap([&] {
    if (!c || (n && !records)) throw Error("null pointer");
    if (!c->host_fn) HIP_CHECK(hipSetDevice(c->device));
    const size_t bytes = n * (4 + 4 * (size_t)max_arcs) * sizeof(uint32_t);
    wfst_comm::Set& s = begin_set(c, std::max<size_t>(bytes, 1));
    if (bytes) std::memcpy(s.h_in, records, bytes);
    queue_exchange(c, s, bytes);
    c->pending = c->next;
    c->next ^= 1;
    c->paths_n = (uint32_t)n;
    c->paths_max_arcs = max_arcs;
  });
}

wfst_status wfst_gather_paths_end(wfst_comm* c, uint32_t* out) {
  return wrap([&] {
    if (!c || !out) throw Error("null pointer");
    size_t bytes = 0;
    const void* h = end_set(c, &bytes);
    std::memcpy(out, h, bytes * c->world);
  });
}

// Ragged payloads (general FSTs serialised in the OpenFST binary format: n-best trees, look-ahead compositions): two
// exchanges — the byte counts, then the payloads padded to the largest rank.  *recv is one allocation holding the ranks'
// payloads back to back in rank order (release with wfst_bytes_destroy); sizes[r] = bytes of rank r.
wfst_status wfst_comm_allgatherv(wfst_comm* c, const void* send, size_t bytes, uint64_t* sizes, void** recv, size_t* total) {
  return wrap([&] {
    if (!c || (bytes && !send) || !sizes || !recv || !total) throw Error("null pointer");
    if (!c->host_fn) HIP_CHECK(hipSetDevice(c->device));
    uint64_t mine = bytes;
    {
      wfst_comm::Set& s = begin_set(c, sizeof(uint64_t));
      std::memcpy(s.h_in, &mine, sizeof(mine));
      queue_exchange(c, s, sizeof(mine));
      c->pending = c->next;
      c->next ^= 1;
      size_t b = 0;
      const void* h = end_set(c, &b);
      std::memcpy(sizes, h, sizeof(uint64_t) * c->world);
    }
    uint64_t max_b = 0, sum = 0;
    for (uint32_t r = 0; r < c->world; ++r) {
      max_b = std::max<uint64_t>(max_b, sizes[r]);
      sum += sizes[r];
    }
    // (handed out like wfst_fst_to_openfst_bytes' buffers: a malloc'ed block released by wfst_bytes_destroy)
    uint8_t* blk = (uint8_t*)std::malloc(std::max<size_t>((size_t)sum, 1));
    if (!blk) throw Error("out of memory");
    try {
      if (max_b) {
        wfst_comm::Set& s = begin_set(c, (size_t)max_b);
        if (bytes) std::memcpy(s.h_in, send, bytes);
        queue_exchange(c, s, (size_t)max_b);
        c->pending = c->next;
        c->next ^= 1;
        size_t b = 0;
        const uint8_t* h = (const uint8_t*)end_set(c, &b);
        size_t o = 0;
        for (uint32_t r = 0; r < c->world; ++r) {
          if (sizes[r]) std::memcpy(blk + o, h + (size_t)r * max_b, (size_t)sizes[r]);
          o += (size_t)sizes[r];
        }
      }
    } catch (...) {
      std::free(blk);
      throw;
    }
    *total = (size_t)sum;
    *recv = blk;
  });
}

}  // extern "C"
