// fst_store.hip — FST handles: HBM CSR arenas, derived arrays, upload / download.
//
// HBM layout of one FST, all in one arena allocation:
//   offsets[n+1] u32 | arcs[E] {ilabel,olabel,weight,nextstate} 16 B (== CTr == on-disk arc,
//   rustfst-ffi/src/tr.rs:8-21) | finals[n] f32 (+inf = non-final) | noeps[n] u32 (number of
//   output-epsilon arcs: VectorFstState.noepsilons, vector_fst/data_structure.rs:28-34, needed by
//   the sequence filter) | wn[E] {weight bits, nextstate} 8 B (the only bytes the relaxation reads) |
//   srec[n] {arc begin, arc count, final bits, noeps} 16 B (compose reads one record per state).
#include "common.h"
#include "fst_props.h"

namespace wfst {

namespace {

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

struct WeightStats {
  double sum;                // sum of the finite arc weights
  unsigned long long count;  // number of finite arc weights
  uint32_t negative;         // some weight < 0
  uint32_t pad;
};

// one thread per arc: pack {w,next}; validate nextstate; accumulate weight statistics
__global__ void derive_wn_kernel(const wfst_tr* __restrict__ arcs, uint2* __restrict__ wn, uint64_t n_arcs,
                                 uint32_t n_states_max, uint32_t* __restrict__ err, WeightStats* __restrict__ ws) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  bool bad = false, neg = false;
  double sum = 0.0;
  unsigned long long cnt = 0;
  for (; i < n_arcs; i += stride) {
    const uint4 a = reinterpret_cast<const uint4*>(arcs)[i];  // one 16-B load per arc
    wn[i] = make_uint2(a.z, a.w);
    bad |= a.w >= n_states_max;
    const float w = __uint_as_float(a.z);
    if (w < INF && w > -INF) {
      sum += (double)w;
      cnt++;
      neg |= w < 0.0f;
    }
  }
  if (bad) atomicOr(err, 1u);
  if (ws) {
    __shared__ double s_sum[4];
    __shared__ unsigned long long s_cnt[4];
    for (int d = 32; d >= 1; d >>= 1) {
      sum += __shfl_xor(sum, d);
      cnt += __shfl_xor(cnt, d);
    }
    if ((threadIdx.x & 63) == 0) {
      s_sum[threadIdx.x >> 6] = sum;
      s_cnt[threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // one pair of atomics per workgroup (same-address atomics serialise)
      double bs = 0.0;
      unsigned long long bc = 0;
      for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) {
        bs += s_sum[w];
        bc += s_cnt[w];
      }
      if (bc) {
        atomicAdd(&ws->sum, bs);
        atomicAdd(&ws->count, bc);
      }
    }
    if (neg) ws->negative = 1u;
  }
}

// one thread per arc: {arc begin, arc count} of the destination state
__global__ void derive_anext_kernel(const wfst_tr* __restrict__ arcs, const uint32_t* __restrict__ offsets,
                                    uint2* __restrict__ anext, uint64_t n_arcs, uint32_t n_states) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_arcs; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t ns = arcs[i].nextstate;
    uint2 r = make_uint2(0u, 0u);
    if (ns < n_states) {
      const uint32_t b = offsets[ns];
      r = make_uint2(b, offsets[ns + 1] - b);
    }
    anext[i] = r;
  }
}
// does any state have an arc with ilabel 0?  (the state records already carry the fact)
__global__ void any_ieps_kernel(const uint4* __restrict__ srec, uint32_t n_states, uint32_t* __restrict__ flag) {
  bool any = false;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_states; s += gridDim.x * blockDim.x)
    any |= srec[s].y != 0u && (srec[s].w & SREC_NO_IEPS) == 0u;
  if (__any(any) && (threadIdx.x & 63) == 0) *flag = 1u;
}

// one thread per state (of the concatenation): count olabel == 0 arcs; validate offsets.
// seg_* describe the FSTs packed in the arena so that nextstate bounds are per FST.
__global__ void derive_noeps_kernel(const uint32_t* __restrict__ offsets, const wfst_tr* __restrict__ arcs,
                                    const float* __restrict__ finals, uint32_t* __restrict__ noeps,
                                    uint4* __restrict__ srec, uint32_t n_states, uint32_t* __restrict__ err) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_states) return;
  const uint32_t b = offsets[s], e = offsets[s + 1];
  if (e < b) {
    atomicOr(err, 2u);
    noeps[s] = 0;
    srec[s] = make_uint4(b, 0u, __float_as_uint(finals[s]), 0u);
    return;
  }
  uint32_t c = 0, ci = 0;
  for (uint32_t i = b; i < e; ++i) {
    c += arcs[i].olabel == WFST_EPS_LABEL;
    ci += arcs[i].ilabel == WFST_EPS_LABEL;
  }
  noeps[s] = c;
  const uint32_t facts = (c == 0 ? SREC_NO_OEPS : 0u) | (c == e - b ? SREC_ALL_OEPS : 0u) | (ci == 0 ? SREC_NO_IEPS : 0u) |
                         (ci == e - b ? SREC_ALL_IEPS : 0u);
  srec[s] = make_uint4(b, e - b, __float_as_uint(finals[s]), facts);
}

struct Layout {
  size_t off_offsets, off_arcs, off_finals, off_noeps, off_wn, off_srec, total;
};
Layout make_layout(size_t n_offsets, size_t n_states, size_t n_arcs) {
  Layout l;
  size_t o = 0;
  l.off_offsets = o;
  o = align_up(o + n_offsets * sizeof(uint32_t));
  l.off_arcs = o;
  o = align_up(o + n_arcs * sizeof(wfst_tr));
  l.off_finals = o;
  o = align_up(o + n_states * sizeof(float));
  l.off_noeps = o;
  o = align_up(o + n_states * sizeof(uint32_t));
  l.off_wn = o;
  o = align_up(o + n_arcs * sizeof(uint2));
  l.off_srec = o;
  o = align_up(o + n_states * sizeof(uint4));
  l.total = std::max<size_t>(o, ALIGN);
  return l;
}

std::shared_ptr<DeviceArena> make_arena(wfst_ctx* ctx, size_t bytes) {
  auto a = std::make_shared<DeviceArena>();
  a->pool = ctx->pool;
  a->base = ctx->pool->alloc(bytes);
  a->bytes = bytes;
  return a;
}

void check_header(uint32_t n_states, int64_t start) {
  if (n_states >= 0x7FFFFFFFu) throw Error("FST too large: state ids must fit in 31 bits");
  if (start < -1 || (start >= 0 && (uint64_t)start >= n_states)) throw Error("start state out of range");
}

}  // namespace

const uint2* ensure_anext(wfst_ctx* ctx, const wfst_fst* f) {
  std::lock_guard<std::mutex> lk(f->cache_mu);
  if (f->anext) return f->anext->p;
  if (!f->has_dev || f->n_arcs == 0) return nullptr;
  DevicePool& owner_pool = f->owner_pool ? *f->owner_pool : *ctx->pool;
  auto buf = std::make_shared<DBuf<uint2>>(owner_pool, f->n_arcs);
  int blocks = (int)std::min<uint64_t>((f->n_arcs + 255) / 256, (uint64_t)ctx->n_cus * 16);
  derive_anext_kernel<<<blocks, 256, 0, ctx->stream>>>(f->dev.arcs, f->dev.offsets, buf->p, f->n_arcs, f->n_states);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  f->anext = buf;
  return buf->p;
}

bool has_input_epsilons(wfst_ctx* ctx, const wfst_fst* f) {
  std::lock_guard<std::mutex> lk(f->cache_mu);
  if (f->ieps_state == 0) {
    if (f->props & props::NO_I_EPSILONS) {
      f->ieps_state = 1;
    } else if (f->props & props::I_EPSILONS) {
      f->ieps_state = 2;
    } else if (!f->has_dev || f->n_states == 0) {
      f->ieps_state = 1;
      if (f->path_form) ensure_host(f);
      if (f->has_host)
        for (const wfst_tr& a : f->host.arcs)
          if (a.ilabel == WFST_EPS_LABEL) f->ieps_state = 2;
    } else {
      DBuf<uint32_t> flag(*ctx->pool, 1);
      HIP_CHECK(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), ctx->stream));
      const uint32_t blocks = std::min<uint32_t>((f->n_states + 255) / 256, (uint32_t)ctx->n_cus * 8);
      any_ieps_kernel<<<blocks, 256, 0, ctx->stream>>>(f->dev.srec, f->n_states, flag.p);
      uint32_t h = 0;
      HIP_CHECK(hipMemcpyAsync(&h, flag.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      f->ieps_state = h ? 2 : 1;
    }
  }
  return f->ieps_state == 2;
}

// utils::acceptor shape (labels_to_fst.rs:111-132): states 0..L in a chain, state s < L has exactly one arc, to s + 1,
// ilabel == olabel != 0; only state L is final; start = 0
bool detect_string(uint32_t n_states, int64_t start, const uint32_t* offsets, const wfst_tr* arcs, const float* finals) {
  if (n_states == 0 || start != 0) return false;
  const uint32_t L = n_states - 1;
  if (offsets[n_states] != L) return false;
  for (uint32_t s = 0; s < L; ++s) {
    if (offsets[s] != s) return false;
    const wfst_tr& a = arcs[s];
    if (a.nextstate != s + 1 || a.ilabel != a.olabel || a.ilabel == WFST_EPS_LABEL) return false;
    if (finals[s] != INF) return false;
  }
  return offsets[L] == L && finals[L] != INF;
}

namespace {

// Runs the derive kernels for a single FST laid out at `l` in `arena` and validates it.
void derive_single(wfst_ctx* ctx, const DeviceCsr& d, uint32_t n_states, uint64_t n_arcs, float* mean_weight,
                   bool* has_negative) {
  DBuf<uint32_t> err(*ctx->pool, 1);
  DBuf<WeightStats> ws(*ctx->pool, 1);
  HIP_CHECK(hipMemsetAsync(err.p, 0, sizeof(uint32_t), ctx->stream));
  HIP_CHECK(hipMemsetAsync(ws.p, 0, sizeof(WeightStats), ctx->stream));
  if (n_arcs) {
    int blocks = (int)std::min<uint64_t>((n_arcs + 255) / 256, (uint64_t)ctx->n_cus * 16);
    derive_wn_kernel<<<blocks, 256, 0, ctx->stream>>>(d.arcs, const_cast<uint2*>(d.wn), n_arcs, n_states, err.p, ws.p);
  }
  if (n_states) {
    derive_noeps_kernel<<<(n_states + 255) / 256, 256, 0, ctx->stream>>>(d.offsets, d.arcs, d.finals,
                                                                        const_cast<uint32_t*>(d.noeps),
                                                                        const_cast<uint4*>(d.srec), n_states, err.p);
  }
  HIP_CHECK(hipGetLastError());
  uint32_t* h = (uint32_t*)ctx->pinned.get(64 + sizeof(WeightStats));
  WeightStats* hws = (WeightStats*)((char*)h + 64);
  HIP_CHECK(hipMemcpyAsync(h, err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipMemcpyAsync(hws, ws.p, sizeof(WeightStats), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (*h & 1u) throw Error("invalid FST: an arc's nextstate is >= num_states");
  if (*h & 2u) throw Error("invalid FST: offsets are not non-decreasing");
  *mean_weight = hws->count ? (float)(hws->sum / (double)hws->count) : 0.0f;
  *has_negative = hws->negative != 0;
}

DeviceCsr carve(const std::shared_ptr<DeviceArena>& arena, const Layout& l) {
  DeviceCsr d;
  d.arena = arena;
  char* b = (char*)arena->base;
  d.offsets = (const uint32_t*)(b + l.off_offsets);
  d.arcs = (const wfst_tr*)(b + l.off_arcs);
  d.finals = (const float*)(b + l.off_finals);
  d.noeps = (const uint32_t*)(b + l.off_noeps);
  d.wn = (const uint2*)(b + l.off_wn);
  d.srec = (const uint4*)(b + l.off_srec);
  return d;
}

}  // namespace

wfst_fst* make_host_fst(wfst_ctx* ctx, uint32_t n_states, int64_t start, uint64_t props, HostCsr&& csr) {
  auto f = std::make_unique<wfst_fst>();
  f->ctx = ctx;
  f->owner_pool = ctx->pool;
  f->device = ctx->device;
  f->n_states = n_states;
  f->n_arcs = csr.arcs.size();
  f->start = start;
  f->props = props & props::ALL;
  f->host = std::move(csr);
  f->has_host = true;
  return f.release();
}

static wfst_fst* upload_generic(wfst_ctx* ctx, uint32_t n_states, int64_t start, const uint32_t* offsets,
                                const wfst_tr* arcs, const float* finals, uint64_t props, hipMemcpyKind kind,
                                uint64_t n_arcs) {
  check_header(n_states, start);
  HIP_CHECK(hipSetDevice(ctx->device));
  Layout l = make_layout((size_t)n_states + 1, n_states, n_arcs);
  auto arena = make_arena(ctx, l.total);
  DeviceCsr d = carve(arena, l);
  if (n_states) {
    HIP_CHECK(hipMemcpyAsync(const_cast<uint32_t*>(d.offsets), offsets, ((size_t)n_states + 1) * sizeof(uint32_t), kind,
                             ctx->stream));
    HIP_CHECK(hipMemcpyAsync(const_cast<float*>(d.finals), finals, (size_t)n_states * sizeof(float), kind, ctx->stream));
  } else {
    HIP_CHECK(hipMemsetAsync(const_cast<uint32_t*>(d.offsets), 0, sizeof(uint32_t), ctx->stream));
  }
  if (n_arcs) HIP_CHECK(hipMemcpyAsync(const_cast<wfst_tr*>(d.arcs), arcs, n_arcs * sizeof(wfst_tr), kind, ctx->stream));
  float mean_w = 0.0f;
  bool has_neg = false;
  derive_single(ctx, d, n_states, n_arcs, &mean_w, &has_neg);
  auto f = std::make_unique<wfst_fst>();
  f->mean_weight = mean_w;
  f->has_negative = has_neg;
  f->ctx = ctx;
  f->owner_pool = ctx->pool;
  f->device = ctx->device;
  f->n_states = n_states;
  f->n_arcs = n_arcs;
  f->start = start;
  f->props = props & props::ALL;  // FstProperties::from_bits_truncate (vector_fst/serializable_fst.rs:165)
  f->dev = d;
  f->has_dev = true;
  return f.release();
}

// A small FST uploaded from host arrays keeps them as its host mirror: host-side steps on it (the look-ahead relabelling
// of an acceptor, packing, downloads) then cost no device read-back and no synchronisation.  (In-place device operations
// drop the mirror: project_device, tr_sort_device.)
static void keep_small_host_copy(wfst_fst* f, const uint32_t* offsets, const wfst_tr* arcs, const float* finals) {
  constexpr uint64_t SMALL_ARCS = 4096;
  if (f->n_arcs > SMALL_ARCS || f->n_states > SMALL_ARCS) return;
  if (f->n_states) f->host.offsets.assign(offsets, offsets + f->n_states + 1);
  else f->host.offsets.assign(1, 0u);
  if (f->n_arcs) f->host.arcs.assign(arcs, arcs + f->n_arcs);
  if (f->n_states) f->host.finals.assign(finals, finals + f->n_states);
  f->has_host = true;
}

wfst_fst* upload_from_host(wfst_ctx* ctx, uint32_t n_states, int64_t start, const uint32_t* offsets, const wfst_tr* arcs,
                           const float* finals, uint64_t props) {
  uint64_t n_arcs = n_states ? offsets[n_states] : 0;
  if (n_states && offsets[0] != 0) throw Error("invalid FST: offsets[0] must be 0");
  if (n_arcs && !arcs) throw Error("null arcs array");
  wfst_fst* f = upload_generic(ctx, n_states, start, offsets, arcs, finals, props, hipMemcpyHostToDevice, n_arcs);
  f->is_string = n_states <= 65536 && detect_string(n_states, start, offsets, arcs, finals);
  keep_small_host_copy(f, offsets, arcs, finals);
  return f;
}

wfst_fst* upload_from_device(wfst_ctx* ctx, uint32_t n_states, int64_t start, const uint32_t* d_offsets,
                             const wfst_tr* d_arcs, const float* d_finals, uint64_t props) {
  HIP_CHECK(hipSetDevice(ctx->device));
  uint32_t last = 0;
  if (n_states) {
    HIP_CHECK(hipMemcpyAsync(&last, d_offsets + n_states, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  return upload_generic(ctx, n_states, start, d_offsets, d_arcs, d_finals, props, hipMemcpyDeviceToDevice, last);
}

wfst_fst* adopt_device(wfst_ctx* ctx, uint32_t n_states, uint64_t n_arcs, int64_t start, uint64_t props,
                       const uint32_t* d_offsets, const wfst_tr* d_arcs, const float* d_finals) {
  return upload_generic(ctx, n_states, start, d_offsets, d_arcs, d_finals, props, hipMemcpyDeviceToDevice, n_arcs);
}

namespace {
struct AdoptJob {
  const uint32_t* off;
  const wfst_tr* arcs;
  const float* fin;
  uint32_t* d_off;
  wfst_tr* d_arcs;
  float* d_fin;
  uint32_t n_states, n_arcs;
};
// block k copies result k (offsets, finals, arcs) out of its problem arena into the batch's shared allocation
__global__ void __launch_bounds__(256) adopt_gather_kernel(const AdoptJob* __restrict__ jobs) {
  const AdoptJob j = jobs[blockIdx.x];
  for (uint32_t i = threadIdx.x; i <= j.n_states; i += 256) j.d_off[i] = j.n_states ? j.off[i] : 0u;
  for (uint32_t i = threadIdx.x; i < j.n_states; i += 256) j.d_fin[i] = j.fin[i];
  const uint4* src = reinterpret_cast<const uint4*>(j.arcs);
  uint4* dst = reinterpret_cast<uint4*>(j.d_arcs);
  for (uint32_t i = threadIdx.x; i < j.n_arcs; i += 256) dst[i] = src[i];
}
}  // namespace

// adopt_device for the m results of one batch: ONE allocation, one gather launch, the derive kernels queued without a
// host round trip in between, ONE synchronisation (per result adopt_device costs ~40 us: three copy commands, two small
// kernels and a read-back of the weight statistics each).
void adopt_device_many(wfst_ctx* ctx, size_t m, const AdoptDesc* descs, wfst_fst** outs) {
  if (m == 0) return;
  HIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  size_t tot_states = 0, tot_arcs = 0;
  std::vector<size_t> state_base(m), arc_base(m);
  for (size_t i = 0; i < m; ++i) {
    check_header(descs[i].n_states, descs[i].start);
    if (descs[i].n_arcs >= 0xFFFFFFFFull) throw Error("adopt_device_many: a result with 2^32 arcs");
    state_base[i] = tot_states;
    arc_base[i] = tot_arcs;
    tot_states += descs[i].n_states;
    tot_arcs += descs[i].n_arcs;
  }
  Layout l = make_layout(tot_states + m, tot_states, tot_arcs);
  auto arena = make_arena(ctx, l.total);
  DeviceCsr all = carve(arena, l);
  std::vector<AdoptJob> jobs(m);
  for (size_t i = 0; i < m; ++i)
    jobs[i] = AdoptJob{descs[i].off, descs[i].arcs, descs[i].fin, const_cast<uint32_t*>(all.offsets) + state_base[i] + i,
                       const_cast<wfst_tr*>(all.arcs) + arc_base[i], const_cast<float*>(all.finals) + state_base[i],
                       descs[i].n_states, (uint32_t)descs[i].n_arcs};
  DBuf<AdoptJob> d_jobs(*ctx->pool, m);
  DBuf<uint32_t> err(*ctx->pool, 1);
  DBuf<WeightStats> ws(*ctx->pool, m);
  HIP_CHECK(hipMemcpyAsync(d_jobs.p, jobs.data(), m * sizeof(AdoptJob), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemsetAsync(err.p, 0, sizeof(uint32_t), st));
  HIP_CHECK(hipMemsetAsync(ws.p, 0, m * sizeof(WeightStats), st));
  adopt_gather_kernel<<<(uint32_t)m, 256, 0, st>>>(d_jobs.p);
  for (size_t i = 0; i < m; ++i) {
    const uint32_t ns = descs[i].n_states;
    const uint64_t na = descs[i].n_arcs;
    if (na)
      derive_wn_kernel<<<(int)std::min<uint64_t>((na + 255) / 256, 64), 256, 0, st>>>(all.arcs + arc_base[i],
                                                                                         const_cast<uint2*>(all.wn) + arc_base[i], na, ns,
                                                                                         err.p, ws.p + i);
    if (ns)
      derive_noeps_kernel<<<(ns + 255) / 256, 256, 0, st>>>(all.offsets + state_base[i] + i, all.arcs + arc_base[i],
                                                            all.finals + state_base[i], const_cast<uint32_t*>(all.noeps) + state_base[i],
                                                            const_cast<uint4*>(all.srec) + state_base[i], ns, err.p);
  }
  HIP_CHECK(hipGetLastError());
  std::vector<WeightStats> hws(m);
  uint32_t herr = 0;
  HIP_CHECK(hipMemcpyAsync(&herr, err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(hws.data(), ws.p, m * sizeof(WeightStats), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));  // (also: jobs / the sources may go now)
  if (herr & 1u) throw Error("invalid FST: an arc's nextstate is >= num_states");
  if (herr & 2u) throw Error("invalid FST: offsets are not non-decreasing");
  for (size_t i = 0; i < m; ++i) {
    auto f = std::make_unique<wfst_fst>();
    f->mean_weight = hws[i].count ? (float)(hws[i].sum / (double)hws[i].count) : 0.0f;
    f->has_negative = hws[i].negative != 0;
    f->ctx = ctx;
    f->owner_pool = ctx->pool;
    f->device = ctx->device;
    f->n_states = descs[i].n_states;
    f->n_arcs = descs[i].n_arcs;
    f->start = descs[i].start;
    f->props = descs[i].props & props::ALL;
    f->dev.arena = arena;
    f->dev.offsets = all.offsets + state_base[i] + i;
    f->dev.arcs = all.arcs + arc_base[i];
    f->dev.finals = all.finals + state_base[i];
    f->dev.noeps = all.noeps + state_base[i];
    f->dev.wn = all.wn + arc_base[i];
    f->dev.srec = all.srec + state_base[i];
    f->has_dev = true;
    outs[i] = f.release();
  }
}

void upload_many(wfst_ctx* ctx, size_t n, const uint32_t* n_states, const int64_t* starts, const uint32_t* offsets_cat,
                 const wfst_tr* arcs_cat, const float* finals_cat, const uint64_t* props, wfst_fst** outs) {
  HIP_CHECK(hipSetDevice(ctx->device));
  size_t tot_states = 0, tot_arcs = 0;
  std::vector<size_t> state_base(n), arc_base(n);
  for (size_t i = 0; i < n; ++i) {
    check_header(n_states[i], starts[i]);
    state_base[i] = tot_states;
    arc_base[i] = tot_arcs;
    const uint32_t* off = offsets_cat + tot_states + i;
    if (off[0] != 0) throw Error("invalid FST in batch: offsets[0] must be 0");
    tot_arcs += off[n_states[i]];
    tot_states += n_states[i];
  }
  Layout l = make_layout(tot_states + n, tot_states, tot_arcs);
  auto arena = make_arena(ctx, l.total);
  DeviceCsr all = carve(arena, l);
  HIP_CHECK(hipMemcpyAsync(const_cast<uint32_t*>(all.offsets), offsets_cat, (tot_states + n) * sizeof(uint32_t),
                           hipMemcpyHostToDevice, ctx->stream));
  if (tot_states)
    HIP_CHECK(hipMemcpyAsync(const_cast<float*>(all.finals), finals_cat, tot_states * sizeof(float), hipMemcpyHostToDevice,
                             ctx->stream));
  if (tot_arcs)
    HIP_CHECK(hipMemcpyAsync(const_cast<wfst_tr*>(all.arcs), arcs_cat, tot_arcs * sizeof(wfst_tr), hipMemcpyHostToDevice,
                             ctx->stream));
  // validation on the host (these are small FSTs: acceptors), derive arrays on the device
  for (size_t i = 0; i < n; ++i) {
    const uint32_t* off = offsets_cat + state_base[i] + i;
    for (uint32_t s = 0; s < n_states[i]; ++s)
      if (off[s + 1] < off[s]) throw Error("invalid FST in batch: offsets are not non-decreasing");
    const wfst_tr* a = arcs_cat + arc_base[i];
    for (uint32_t e = 0; e < off[n_states[i]]; ++e)
      if (a[e].nextstate >= n_states[i]) throw Error("invalid FST in batch: an arc's nextstate is >= num_states");
  }
  {
    DBuf<uint32_t> err(*ctx->pool, 1);
    HIP_CHECK(hipMemsetAsync(err.p, 0, sizeof(uint32_t), ctx->stream));
    if (tot_arcs) {
      int blocks = (int)std::min<uint64_t>((tot_arcs + 255) / 256, (uint64_t)ctx->n_cus * 16);
      derive_wn_kernel<<<blocks, 256, 0, ctx->stream>>>(all.arcs, const_cast<uint2*>(all.wn), tot_arcs, 0xFFFFFFFFu, err.p,
                                                        nullptr);
    }
    // noeps: per FST the offsets are relative, so run per FST on its slice (tiny launches; upload path only)
    for (size_t i = 0; i < n; ++i) {
      if (!n_states[i]) continue;
      derive_noeps_kernel<<<(n_states[i] + 255) / 256, 256, 0, ctx->stream>>>(
          all.offsets + state_base[i] + i, all.arcs + arc_base[i], all.finals + state_base[i],
          const_cast<uint32_t*>(all.noeps) + state_base[i], const_cast<uint4*>(all.srec) + state_base[i], n_states[i],
          err.p);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  for (size_t i = 0; i < n; ++i) {
    auto f = std::make_unique<wfst_fst>();
    f->ctx = ctx;
    f->owner_pool = ctx->pool;
    f->device = ctx->device;
    f->n_states = n_states[i];
    f->n_arcs = (offsets_cat + state_base[i] + i)[n_states[i]];
    f->start = starts[i];
    f->props = props[i] & props::ALL;
    f->dev.arena = arena;
    f->dev.offsets = all.offsets + state_base[i] + i;
    f->dev.arcs = all.arcs + arc_base[i];
    f->dev.finals = all.finals + state_base[i];
    f->dev.noeps = all.noeps + state_base[i];
    f->dev.wn = all.wn + arc_base[i];
    f->dev.srec = all.srec + state_base[i];
    f->has_dev = true;
    f->is_string = n_states[i] <= 65536 && detect_string(n_states[i], starts[i], offsets_cat + state_base[i] + i,
                                                         arcs_cat + arc_base[i], finals_cat + state_base[i]);
    keep_small_host_copy(f.get(), offsets_cat + state_base[i] + i, arcs_cat + arc_base[i], finals_cat + state_base[i]);
    outs[i] = f.release();
  }
}

namespace {
__global__ void project_kernel(wfst_tr* __restrict__ arcs, uint64_t n_arcs, int project_output) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_arcs; k += (uint64_t)gridDim.x * blockDim.x) {
    if (project_output)
      arcs[k].ilabel = arcs[k].olabel;
    else
      arcs[k].olabel = arcs[k].ilabel;
  }
}
}  // namespace

// project (algorithms/projection.rs:65-95) in place on the device-resident arcs: one label column copied over the other,
// the per-state epsilon facts (noeps, srec) derived again, the property word from project_properties
void project_device(wfst_ctx* ctx, wfst_fst* f, bool project_output) {
  HIP_CHECK(hipSetDevice(ctx->device));
  if (f->n_arcs) {
    ensure_device(f);
    const int blocks = (int)std::min<uint64_t>((f->n_arcs + 255) / 256, (uint64_t)ctx->n_cus * 16);
    project_kernel<<<blocks, 256, 0, ctx->stream>>>(const_cast<wfst_tr*>(f->dev.arcs), f->n_arcs, project_output ? 1 : 0);
    DBuf<uint32_t> err(*ctx->pool, 1);
    HIP_CHECK(hipMemsetAsync(err.p, 0, sizeof(uint32_t), ctx->stream));
    derive_noeps_kernel<<<(f->n_states + 255) / 256, 256, 0, ctx->stream>>>(f->dev.offsets, f->dev.arcs, f->dev.finals,
                                                                             const_cast<uint32_t*>(f->dev.noeps),
                                                                             const_cast<uint4*>(f->dev.srec), f->n_states, err.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (f->has_host) {  // the host mirror described the old labels
      f->host = HostCsr{};
      f->has_host = false;
    }
    f->rev_host.reset();  // reverse() carries labels
    f->ieps_state = 0;
    if (f->is_string) {  // still linear; epsilon-free on the input side only if the copied column was
      ensure_host(f);
      f->is_string = detect_string(f->n_states, f->start, f->host.offsets.data(), f->host.arcs.data(), f->host.finals.data());
    }
  }
  f->props = props::project(f->props, project_output);
}

void ensure_device(wfst_fst* f) {
  if (f->has_dev) return;
  if (f->path_form) ensure_host(f);
  if (!f->has_host) throw Error("FST handle holds no data");
  wfst_ctx* ctx = f->ctx;
  std::unique_ptr<wfst_fst> tmp(upload_from_host(ctx, f->n_states, f->start, f->host.offsets.data(), f->host.arcs.data(),
                                                 f->host.finals.data(), f->props));
  f->dev = tmp->dev;
  f->mean_weight = tmp->mean_weight;
  f->has_negative = tmp->has_negative;
  f->has_dev = true;
}

// The reference's linear shortest-path FST (single_shortest_path_backtrace, shortest_path.rs:241-282) from the walk's arcs:
// hops + 1 states numbered backwards (state 0 final, start = hops), arc k enters state k; no path = the empty FST.
wfst_fst* make_path_fst(wfst_ctx* ctx, bool has_path, uint32_t hops, float final_weight, const wfst_tr* path_arcs) {
  HostCsr h;
  uint32_t n_states = 0;
  int64_t start = -1;
  if (has_path) {
    n_states = hops + 1;
    h.finals.assign(n_states, INF);
    h.finals[0] = final_weight;
    if (hops) h.arcs.assign(path_arcs, path_arcs + hops);
    h.offsets.resize((size_t)n_states + 1);
    h.offsets[0] = 0;
    for (uint32_t k = 0; k <= hops; ++k) h.offsets[k + 1] = k;  // state 0 has no arc, state k >= 1 one
    start = hops;
  } else {
    h.offsets.push_back(0);
  }
  return make_host_fst(ctx, n_states, start, props::linear_path_props(has_path, hops, final_weight, path_arcs), std::move(h));
}

void pack_path_record(uint32_t* rec, uint32_t max_arcs, bool valid, uint32_t n_arcs, float final_weight, const wfst_tr* arcs) {
  if (valid && n_arcs > max_arcs) throw Error("wfst_fst_pack_paths: path longer than the record");
  const float w = valid ? final_weight : INF;
  rec[0] = valid ? n_arcs : 0u;
  std::memcpy(&rec[1], &w, 4);
  rec[2] = valid ? 1u : 0u;
  rec[3] = 0;
  const size_t used = valid ? n_arcs : 0;
  if (used) std::memcpy(&rec[4], arcs, used * sizeof(wfst_tr));
  if (used < max_arcs) std::memset(&rec[4 + 4 * used], 0, (max_arcs - used) * sizeof(wfst_tr));
}

void pack_path_record(uint32_t* rec, uint32_t max_arcs, const wfst_fst* f) {
  if (f->path_form) {  // straight from the batch's block
    pack_path_record(rec, max_arcs, f->n_states != 0, (uint32_t)f->n_arcs, f->n_states ? f->path_final : INF, f->n_arcs ? f->path_arcs : nullptr);
    return;
  }
  ensure_host(f);
  if (f->n_states && f->n_arcs + 1 != f->n_states) throw Error("wfst_fst_pack_paths: not a linear path FST");
  pack_path_record(rec, max_arcs, f->n_states != 0, (uint32_t)f->n_arcs, f->n_states ? f->host.finals[0] : INF,
                   f->n_arcs ? f->host.arcs.data() : nullptr);
}

void ensure_host(const wfst_fst* cf) {
  if (cf->has_host) return;
  wfst_fst* f = const_cast<wfst_fst*>(cf);  // cache fill only
  if (f->path_form) {  // a batch result still pointing into its block: the reference's linear path FST (shortest_path.rs:257-272)
    const uint32_t hops = (uint32_t)f->n_arcs;
    f->host.finals.assign(f->n_states, INF);
    f->host.offsets.assign(1, 0u);
    if (f->n_states) {
      f->host.finals[0] = f->path_final;
      if (hops) f->host.arcs.assign(f->path_arcs, f->path_arcs + hops);
      f->host.offsets.resize((size_t)f->n_states + 1);
      for (uint32_t k = 0; k <= hops; ++k) f->host.offsets[k + 1] = k;
    }
    f->has_host = true;
    f->path_form = false;
    f->path_arcs = nullptr;
    f->path_block.reset();
    return;
  }
  if (!f->has_dev) throw Error("FST handle holds no data");
  wfst_ctx* ctx = f->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  f->host.offsets.resize((size_t)f->n_states + 1);
  f->host.arcs.resize(f->n_arcs);
  f->host.finals.resize(f->n_states);
  HIP_CHECK(hipMemcpyAsync(f->host.offsets.data(), f->dev.offsets, f->host.offsets.size() * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, ctx->stream));
  if (f->n_arcs)
    HIP_CHECK(hipMemcpyAsync(f->host.arcs.data(), f->dev.arcs, f->n_arcs * sizeof(wfst_tr), hipMemcpyDeviceToHost,
                             ctx->stream));
  if (f->n_states)
    HIP_CHECK(hipMemcpyAsync(f->host.finals.data(), f->dev.finals, (size_t)f->n_states * sizeof(float),
                             hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  f->has_host = true;
}

}  // namespace wfst
