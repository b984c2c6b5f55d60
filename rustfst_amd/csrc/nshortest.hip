// nshortest.hip — shortest_path_with_config for nshortest > 1, unique = false
// (rustfst/src/algorithms/shortest_path.rs:135-170):
//   1. distance = shortest_distance(ifst)           shortest_distance.rs:153-237   -> sssp.hip relaxation (GPU)
//   2. rfst = reverse(ifst)                         reverse.rs:33-87               -> CSR transpose kernels below (GPU)
//   3. d = (+) over rfst.trs(0) of w (x) distance   shortest_path.rs:143-153       -> host, O(#finals)
//   4. n_shortest_path(rfst, [d]++distance, n)      shortest_path.rs:284-518       -> host: an inherently sequential
//      best-first search (custom binary heap) that pops at most n times per state; it touches a few
//      thousand arcs, so it runs on the host against the reversed CSR (downloaded once per FST handle).
//   5. connect + shortest_path_properties(.., false)  shortest_path.rs:512-517     -> host, on the small result
// Steps 3-5 keep the reference's TropicalWeight semantics verbatim (approximate ==, approx_equal(delta)):
// they decide heap order and must not deviate.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstring>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <cstdlib>
#include <unordered_map>

#include "common.h"
#include "fst_props.h"
#include "host_parallel.h"

namespace wfst {

namespace {

// ---------------------------------------------------------------- reverse(): counting sort by nextstate
__global__ void rev_count_kernel(const wfst_tr* __restrict__ arcs, uint64_t n_arcs, uint32_t* __restrict__ counts) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_arcs; i += (uint64_t)gridDim.x * blockDim.x)
    atomicAdd(&counts[arcs[i].nextstate], 1u);
}
// slot reservation in arbitrary order; the arc index is kept so that each target's segment can be put back
// into (source state, arc position) order = the order reverse() pushes arcs (reverse.rs:62-67)
__global__ void rev_place_kernel(const uint32_t* __restrict__ offsets, const wfst_tr* __restrict__ arcs, uint32_t n_states,
                                 const uint32_t* __restrict__ roff, uint32_t* __restrict__ cursor,
                                 uint2* __restrict__ tmp) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = tid & 15u;
  for (uint32_t s = tid >> 4; s < n_states; s += (gridDim.x * blockDim.x) >> 4)
    for (uint32_t i = offsets[s] + lane; i < offsets[s + 1]; i += 16) {
      const uint32_t t = arcs[i].nextstate;
      const uint32_t slot = roff[t] + atomicAdd(&cursor[t], 1u);
      tmp[slot] = make_uint2(i, s);
    }
}
// per target: put the segment back into arc-index order and emit the reversed arcs.  Segments of up to REV_SMALL in-arcs
// (all but hub states) are insertion-sorted by one lane; a longer one — a hub with 1e5..1e6 in-arcs would cost
// in-degree^2 dependent memory operations in that lane — is only LISTED here and sorted by rocPRIM's segmented radix
// sort on the arc index (the low word of the {arc index, source} pair), then emitted by rev_emit_big_kernel.
constexpr uint32_t REV_SMALL = 48;
__global__ void rev_emit_kernel(const wfst_tr* __restrict__ arcs, const uint32_t* __restrict__ roff, uint2* __restrict__ tmp,
                                uint32_t n_states, wfst_tr* __restrict__ rarcs, uint32_t* __restrict__ big_count,
                                uint32_t* __restrict__ big_begin, uint32_t* __restrict__ big_end) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_states) return;
  const uint32_t b = roff[t], e = roff[t + 1];
  if (e - b > REV_SMALL) {
    const uint32_t k = atomicAdd(big_count, 1u);
    big_begin[k] = b;
    big_end[k] = e;
    return;
  }
  for (uint32_t i = b + 1; i < e; ++i) {
    const uint2 v = tmp[i];
    uint32_t j = i;
    while (j > b && tmp[j - 1].x > v.x) {
      tmp[j] = tmp[j - 1];
      --j;
    }
    tmp[j] = v;
  }
  for (uint32_t i = b; i < e; ++i) {
    const uint2 v = tmp[i];
    wfst_tr a = arcs[v.x];
    a.nextstate = v.y + 1;  // state i -> i + 1 (super-initial state 0)
    rarcs[i] = a;
  }
}
// the listed segments, already sorted: one workgroup per segment
__global__ void __launch_bounds__(256) rev_emit_big_kernel(const wfst_tr* __restrict__ arcs, const uint2* __restrict__ sorted,
                                                          const uint32_t* __restrict__ big_begin,
                                                          const uint32_t* __restrict__ big_end, uint32_t n_big,
                                                          wfst_tr* __restrict__ rarcs) {
  for (uint32_t k = blockIdx.x; k < n_big; k += gridDim.x)
    for (uint32_t i = big_begin[k] + threadIdx.x; i < big_end[k]; i += blockDim.x) {
      const uint2 v = sorted[i];
      wfst_tr a = arcs[v.x];
      a.nextstate = v.y + 1;
      rarcs[i] = a;
    }
}

// one in-arc segment (state t of the original FST) -> pinned host memory: {count, arcs[min(count, cap)]}
__global__ void rev_fetch_kernel(const uint32_t* __restrict__ roff, const wfst_tr* __restrict__ rarcs, uint32_t t,
                                 uint32_t* __restrict__ out_count, wfst_tr* __restrict__ out_arcs, uint32_t cap) {
  const uint32_t b = roff[t], c = roff[t + 1] - b;
  if (threadIdx.x == 0) *out_count = c;
  for (uint32_t i = threadIdx.x; i < c && i < cap; i += blockDim.x) out_arcs[i] = rarcs[b + i];
}

std::shared_ptr<RevFst> build_reverse(wfst_ctx* ctx, const wfst_fst* f) {
  ensure_device(const_cast<wfst_fst*>(f));
  const uint32_t n = f->n_states;
  const uint64_t E = f->n_arcs;
  hipStream_t st = ctx->stream;
  DevicePool& pool = *ctx->pool;
  auto rev = std::make_shared<RevFst>();
  rev->n = n;
  // the in-arc segments of large FSTs stay in HBM: the search visits a few hundred of them, downloading all (16 B per
  // arc) would dominate (0.42 s for 50M arcs)
  rev->on_host = E < (1ull << 22);
  if (const char* e = std::getenv("WFST_NBEST_LAZY")) rev->on_host = std::atoi(e) == 0;
  std::vector<float> finals(n);
  DBuf<uint32_t> d_counts(pool, (size_t)n + 1);
  HIP_CHECK(hipMemsetAsync(d_counts.p, 0, ((size_t)n + 1) * sizeof(uint32_t), st));
  const int blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((E + 255) / 256, (uint64_t)ctx->n_cus * 8));
  if (E) rev_count_kernel<<<blocks, 256, 0, st>>>(f->dev.arcs, E, d_counts.p);
  if (n) HIP_CHECK(hipMemcpyAsync(finals.data(), f->dev.finals, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
  // offsets of the in-arc segments (targets 0..n-1): exclusive scan of the in-degrees on the device
  DevicePool& owner_pool = f->owner_pool ? *f->owner_pool : *ctx->pool;  // cached with the handle: the owner's pool outlives it
  rev->d_roff = DBuf<uint32_t>(owner_pool, (size_t)n + 1);
  rev->d_arcs = DBuf<wfst_tr>(owner_pool, E);
  {
    size_t temp_bytes = 0;
    HIP_CHECK(rocprim::exclusive_scan(nullptr, temp_bytes, d_counts.p, rev->d_roff.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st));
    DBuf<uint8_t> temp(pool, temp_bytes);
    HIP_CHECK(rocprim::exclusive_scan(temp.p, temp_bytes, d_counts.p, rev->d_roff.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st));
    uint32_t total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, rev->d_roff.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (total != E) throw Error("reverse: inconsistent arc count");
  }
  std::vector<uint32_t> roff;
  if (rev->on_host) {
    roff.resize((size_t)n + 1);
    HIP_CHECK(hipMemcpyAsync(roff.data(), rev->d_roff.p, ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  }
  if (E) {
    DBuf<uint32_t> d_cursor(pool, n);
    DBuf<uint2> d_tmp(pool, E);
    HIP_CHECK(hipMemsetAsync(d_cursor.p, 0, (size_t)n * sizeof(uint32_t), st));
    rev_place_kernel<<<blocks, 256, 0, st>>>(f->dev.offsets, f->dev.arcs, n, rev->d_roff.p, d_cursor.p, d_tmp.p);
    // (a segment longer than REV_SMALL exists at most E / REV_SMALL times)
    const size_t big_cap = (size_t)(E / (REV_SMALL + 1)) + 1;
    DBuf<uint32_t> d_big(pool, 1 + 2 * big_cap);
    HIP_CHECK(hipMemsetAsync(d_big.p, 0, sizeof(uint32_t), st));
    rev_emit_kernel<<<(n + 255) / 256, 256, 0, st>>>(f->dev.arcs, rev->d_roff.p, d_tmp.p, n, rev->d_arcs.p, d_big.p, d_big.p + 1,
                                                     d_big.p + 1 + big_cap);
    HIP_CHECK(hipGetLastError());
    uint32_t n_big = 0;
    HIP_CHECK(hipMemcpyAsync(&n_big, d_big.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (n_big) {  // hub states: segmented radix sort of the {arc index, source} pairs on the arc index
      if (E >= 0x7FFFFFFFull) throw Error("reverse: too many arcs for the segmented sort");
      DBuf<uint2> d_sorted(pool, E);
      uint64_t* kin = (uint64_t*)d_tmp.p;
      uint64_t* kout = (uint64_t*)d_sorted.p;
      size_t temp_bytes = 0;
      HIP_CHECK(rocprim::segmented_radix_sort_keys(nullptr, temp_bytes, kin, kout, (unsigned)E, n_big, d_big.p + 1,
                                                   d_big.p + 1 + big_cap, 0u, 32u, st));
      DBuf<uint8_t> temp(pool, temp_bytes);
      HIP_CHECK(rocprim::segmented_radix_sort_keys(temp.p, temp_bytes, kin, kout, (unsigned)E, n_big, d_big.p + 1,
                                                   d_big.p + 1 + big_cap, 0u, 32u, st));
      rev_emit_big_kernel<<<std::min<uint32_t>(n_big, (uint32_t)ctx->n_cus * 8), 256, 0, st>>>(
          f->dev.arcs, d_sorted.p, d_big.p + 1, d_big.p + 1 + big_cap, n_big, rev->d_arcs.p);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(st));  // d_sorted / temp are released here
    }
    if (rev->on_host) {
      rev->h_arcs.resize(E);
      HIP_CHECK(hipMemcpyAsync(rev->h_arcs.data(), rev->d_arcs.p, E * sizeof(wfst_tr), hipMemcpyDeviceToHost, st));
    }
    HIP_CHECK(hipStreamSynchronize(st));  // d_cursor / d_tmp are released here
  } else {
    HIP_CHECK(hipStreamSynchronize(st));
  }
  if (rev->on_host) {
    rev->h_roff = std::move(roff);
    rev->d_roff.reset();
    rev->d_arcs.reset();
  }
  // state 0 = super-initial with one eps:eps arc per final state, in state order (reverse.rs:56-60)
  for (uint32_t s = 0; s < n; ++s)
    if (finals[s] != INF) rev->super.push_back(wfst_tr{0u, 0u, finals[s], s + 1});
  rev->finals.assign((size_t)n + 1, INF);
  if (f->start >= 0) rev->finals[(size_t)f->start + 1] = 0.0f;  // reverse.rs:53-55
  return rev;
}

// ---------------------------------------------------------------- reverse() as a public operation (wfst_reverse)
// union of add_tr's facts (fst_props.h add_trs_by_facts) over the arcs of reversed states 1..n
__global__ void __launch_bounds__(256) rev_facts_kernel(const uint32_t* __restrict__ roff, const wfst_tr* __restrict__ rarcs,
                                                       uint32_t n, uint32_t* __restrict__ facts_out) {
  uint32_t facts = 0;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const uint32_t b = roff[t], e = roff[t + 1];
    uint32_t pil = 0, pol = 0;
    for (uint32_t i = b; i < e; ++i) {
      const wfst_tr a = rarcs[i];
      facts |= (a.ilabel != a.olabel ? 1u : 0u) | (a.ilabel == 0u ? 2u : 0u) | (a.ilabel == 0u && a.olabel == 0u ? 4u : 0u) |
               (a.olabel == 0u ? 8u : 0u) | (i > b && pil > a.ilabel ? 16u : 0u) | (i > b && pol > a.olabel ? 32u : 0u) |
               (a.nextstate <= t + 1u ? 128u : 0u);
      const float w = a.weight;
      const bool is_zero = INF <= w + props::KDELTA;                                  // approx == +inf (semiring.rs:159-168)
      const bool is_one = w <= props::KDELTA && 0.0f <= w + props::KDELTA;            // approx == 0
      if (!is_zero && !is_one) facts |= 64u;
      pil = a.ilabel;
      pol = a.olabel;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) facts |= __shfl_xor(facts, d);
  if ((threadIdx.x & 63) == 0 && facts) atomicOr(facts_out, facts);
}
// offsets of the output: [0, F, F + roff[1], ..., F + roff[n]]; finals: one() at start + 1
__global__ void rev_assemble_kernel(const uint32_t* __restrict__ roff, uint32_t n, uint32_t n_super, int64_t start,
                                    uint32_t* __restrict__ off_out, float* __restrict__ fin_out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k <= n + 1) off_out[k] = k == 0 ? 0u : n_super + roff[k - 1];
  if (k <= n) fin_out[k] = (k >= 1 && (int64_t)(k - 1) == start) ? 0.0f : INF;
}

// arcs of state `rs` of the reversed FST (a pointer valid until the next call)
const wfst_tr* rev_arcs_of(wfst_ctx* ctx, RevFst& r, uint32_t rs, uint32_t* count, std::vector<wfst_tr>& scratch) {
  if (rs == 0) {
    *count = (uint32_t)r.super.size();
    return r.super.data();
  }
  const uint32_t t = rs - 1;
  if (r.on_host) {
    *count = r.h_roff[t + 1] - r.h_roff[t];
    return r.h_arcs.data() + r.h_roff[t];
  }
  constexpr uint32_t CAP = 1024;  // arcs per fetch through the small pinned buffer (in-degrees are ~fan-out)
  char* pin = (char*)ctx->pinned.get(64 + CAP * sizeof(wfst_tr));
  uint32_t* h_cnt = (uint32_t*)pin;
  wfst_tr* h_arcs = (wfst_tr*)(pin + 64);
  rev_fetch_kernel<<<1, 64, 0, ctx->stream>>>(r.d_roff.p, r.d_arcs.p, t, h_cnt, h_arcs, CAP);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  r.fetched_segments += 1;
  const uint32_t c = *h_cnt;
  *count = c;
  if (c <= CAP) return h_arcs;
  // a hub state: copy its whole segment
  scratch.resize(c);
  uint32_t b = 0;
  HIP_CHECK(hipMemcpyAsync(&b, r.d_roff.p + t, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  HIP_CHECK(hipMemcpyAsync(scratch.data(), r.d_arcs.p + b, (size_t)c * sizeof(wfst_tr), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return scratch.data();
}

// ---------------------------------------------------------------- TropicalWeight with the reference's semantics
inline float wplus(float a, float b) { return b < a ? b : a; }
inline float wtimes(float a, float b) { return a == INF ? a : (b == INF ? b : a + b); }
inline bool weq(float a, float b) { return props::approx_eq(a, b); }  // KDELTA (semiring.rs:159-168)
inline bool approx_equal(float a, float b, float delta) { return std::fabs(a - b) <= delta; }
inline bool natural_less(float w1, float w2) { return weq(wplus(w1, w2), w1) && !weq(w1, w2); }  // shortest_path.rs:284-286

struct Pair {
  bool some;
  uint32_t state;
  float w;
};
// Heap (:340-407) ordered by ShortestPathCompare (:288-338)
struct Heap {
  std::vector<uint32_t> data;
  const std::vector<Pair>* pairs;
  const std::vector<float>* distance;
  float delta;
  float pweight(const Pair& p) const {
    if (!p.some) return 0.0f;
    return p.state < distance->size() ? (*distance)[p.state] : INF;
  }
  bool less(uint32_t x, uint32_t y) const {
    const Pair& px = (*pairs)[x];
    const Pair& py = (*pairs)[y];
    const float wx = wtimes(pweight(px), px.w);
    const float wy = wtimes(pweight(py), py.w);
    if (!px.some && py.some) return natural_less(wy, wx) || approx_equal(wx, wy, delta);
    if (px.some && !py.some) return natural_less(wy, wx) && !approx_equal(wx, wy, delta);
    return natural_less(wy, wx);
  }
  void push(uint32_t v) {
    data.push_back(v);
    size_t idx = data.size() - 1;
    while (idx > 0) {
      const size_t parent = (idx - 1) / 2;
      if (!less(data[parent], data[idx])) break;
      std::swap(data[idx], data[parent]);
      idx = parent;
    }
  }
  uint32_t pop() {
    const uint32_t top = data[0];
    if (data.size() == 1) {
      data.clear();
      return top;
    }
    data[0] = data.back();
    data.pop_back();
    size_t idx = 0;
    for (;;) {
      const uint32_t cur = data[idx];
      const size_t c1 = 2 * idx + 1, c2 = 2 * idx + 2;
      size_t big;
      if (c1 >= data.size() && c2 >= data.size()) break;
      if (c1 < data.size() && c2 >= data.size())
        big = c1;
      else if (less(data[c1], data[c2]))
        big = c2;
      else
        big = c1;
      if (less(data[big], cur)) break;
      std::swap(data[idx], data[big]);
      idx = big;
    }
    return top;
  }
};

// small mutable output FST with the reference's property bookkeeping (mutable_fst.rs)
struct OutFst {
  struct St {
    bool has_final = false;
    float final_w = INF;
    std::vector<wfst_tr> trs;
  };
  std::vector<St> states;
  int64_t start = -1;
  uint64_t p = props::NULL_PROPS;
  uint32_t add_state() {
    states.emplace_back();
    p = props::add_state(p);
    return (uint32_t)states.size() - 1;
  }
  void set_start(uint32_t s) {
    start = s;
    p = props::set_start(p);
  }
  void set_final(uint32_t s, float w) {
    p = props::set_final(p, states[s].has_final ? &states[s].final_w : nullptr, &w);
    states[s].has_final = true;
    states[s].final_w = w;
  }
  void add_tr(uint32_t s, const wfst_tr& tr) {
    states[s].trs.push_back(tr);
    const size_t k = states[s].trs.size();
    p = props::add_tr(p, s, states[s].trs.back(), k > 1 ? &states[s].trs[k - 2] : nullptr);
  }
  // connect (connect.rs:51-66): keep access & coaccess, stable renumbering (mutable_fst.rs:132-189)
  void connect() {
    const size_t n = states.size();
    std::vector<uint8_t> access(n, 0), coaccess(n, 0);
    std::vector<uint32_t> stack;
    if (start >= 0) {
      access[(size_t)start] = 1;
      stack.push_back((uint32_t)start);
      while (!stack.empty()) {
        const uint32_t s = stack.back();
        stack.pop_back();
        for (const wfst_tr& tr : states[s].trs)
          if (!access[tr.nextstate]) {
            access[tr.nextstate] = 1;
            stack.push_back(tr.nextstate);
          }
      }
    }
    std::vector<std::vector<uint32_t>> rev(n);
    for (size_t s = 0; s < n; ++s)
      for (const wfst_tr& tr : states[s].trs) rev[tr.nextstate].push_back((uint32_t)s);
    for (size_t s = 0; s < n; ++s)
      if (states[s].has_final) {
        coaccess[s] = 1;
        stack.push_back((uint32_t)s);
      }
    while (!stack.empty()) {
      const uint32_t s = stack.back();
      stack.pop_back();
      for (uint32_t q : rev[s])
        if (!coaccess[q]) {
          coaccess[q] = 1;
          stack.push_back(q);
        }
    }
    std::vector<int64_t> new_id(n, -1);
    size_t k = 0;
    for (size_t s = 0; s < n; ++s)
      if (access[s] && coaccess[s]) new_id[s] = (int64_t)k++;
    std::vector<St> kept(k);
    for (size_t s = 0; s < n; ++s) {
      if (new_id[s] < 0) continue;
      St& dst = kept[(size_t)new_id[s]];
      dst.has_final = states[s].has_final;
      dst.final_w = states[s].final_w;
      for (const wfst_tr& tr : states[s].trs)
        if (new_id[tr.nextstate] >= 0) {
          wfst_tr t2 = tr;
          t2.nextstate = (uint32_t)new_id[tr.nextstate];
          dst.trs.push_back(t2);
        }
    }
    states.swap(kept);
    start = (start >= 0 && new_id[(size_t)start] >= 0) ? new_id[(size_t)start] : -1;
    p = props::delete_states(p);
    p = (p & ~(props::ACCESSIBLE | props::COACCESSIBLE)) | props::ACCESSIBLE | props::COACCESSIBLE;
  }
};

}  // namespace

// reverse (reverse.rs:33-87): state 0 = super-initial with one eps:eps arc per final state (weight = its final weight),
// state s + 1 = the arcs INTO s, turned around, in (source state, arc position) order; start + 1 is final with one().
wfst_fst* reverse_fst(wfst_ctx* ctx, const wfst_fst* f) {
  wfst_fst* mf = const_cast<wfst_fst*>(f);
  std::shared_ptr<RevFst> rev;
  {
    std::lock_guard<std::mutex> lk(f->cache_mu);
    if (!mf->rev_host) mf->rev_host = build_reverse(ctx, f);
    rev = mf->rev_host;
  }
  const RevFst& r = *rev;
  const uint32_t n = r.n;
  const uint32_t n_super = (uint32_t)r.super.size();
  const uint64_t E = f->n_arcs;
  hipStream_t st = ctx->stream;
  DevicePool& pool = *ctx->pool;
  // the in-arc segments on the device (small FSTs keep them on the host only)
  DBuf<uint32_t> tmp_roff;
  DBuf<wfst_tr> tmp_arcs;
  const uint32_t* d_roff = r.d_roff.p;
  const wfst_tr* d_arcs = r.d_arcs.p;
  if (r.on_host) {
    tmp_roff = DBuf<uint32_t>(pool, (size_t)n + 1);
    tmp_arcs = DBuf<wfst_tr>(pool, E);
    HIP_CHECK(hipMemcpyAsync(tmp_roff.p, r.h_roff.data(), ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if (E) HIP_CHECK(hipMemcpyAsync(tmp_arcs.p, r.h_arcs.data(), E * sizeof(wfst_tr), hipMemcpyHostToDevice, st));
    d_roff = tmp_roff.p;
    d_arcs = tmp_arcs.p;
  }
  DBuf<uint32_t> off_out(pool, (size_t)n + 2), d_facts(pool, 1);
  DBuf<float> fin_out(pool, (size_t)n + 1);
  DBuf<wfst_tr> arcs_out(pool, (size_t)n_super + E);
  HIP_CHECK(hipMemsetAsync(d_facts.p, 0, sizeof(uint32_t), st));
  if (n_super) HIP_CHECK(hipMemcpyAsync(arcs_out.p, r.super.data(), (size_t)n_super * sizeof(wfst_tr), hipMemcpyHostToDevice, st));
  if (E) HIP_CHECK(hipMemcpyAsync(arcs_out.p + n_super, d_arcs, E * sizeof(wfst_tr), hipMemcpyDeviceToDevice, st));
  rev_assemble_kernel<<<(n + 2 + 255) / 256, 256, 0, st>>>(d_roff, n, n_super, f->start, off_out.p, fin_out.p);
  if (n && E) {
    const uint32_t blocks = std::min<uint32_t>((n + 255) / 256, (uint32_t)ctx->n_cus * 8);
    rev_facts_kernel<<<blocks, 256, 0, st>>>(d_roff, d_arcs, n, d_facts.p);
  }
  HIP_CHECK(hipGetLastError());
  uint32_t facts = 0;
  HIP_CHECK(hipMemcpyAsync(&facts, d_facts.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  // property word: the mutations reverse() performs, in its order (add_state x (n+1), set_final(start+1, one), the arcs
  // state by state, set_start(0)), then reverse_properties(iprops, true) | oprops
  uint64_t p = props::NULL_PROPS;
  for (uint32_t k = 0; k < std::min<uint32_t>(n + 1, 2u); ++k) p = props::add_state(p);  // (idempotent after the first)
  if (f->start >= 0) {
    const float one = 0.0f;
    p = props::set_final(p, nullptr, &one);
  }
  for (const wfst_tr& a : r.super) {  // state 0: eps:eps arcs into states >= 1
    if (!props::is_zero(a.weight) && !props::is_one(a.weight)) facts |= 64u;
    facts |= 2u | 4u | 8u;
  }
  if (n_super + E) p = props::add_trs_by_facts(p, facts);
  p = props::set_start(p);
  p = (props::reverse(f->props, true) | p) & props::ALL;
  return adopt_device(ctx, n + 1, (uint64_t)n_super + E, 0, p, off_out.p, arcs_out.p, fin_out.p);
}

// n_shortest_path (shortest_path.rs:409-518) over an FST given by two accessors (its start state is 0): the reversed input
// (arcs fetched lazily from the GPU transpose), or its determinization (`unique`).  Leaves the un-trimmed tree in ofst.
template <class ArcsOf, class FinalOf>
void nbest_search(ArcsOf&& arcs_of, FinalOf&& final_of, const std::vector<float>& distance_2, uint64_t nshortest, float delta,
                  OutFst& ofst) {
  const uint32_t istart = 0;  // rfst.start()
  if (distance_2.size() <= istart || props::is_zero(distance_2[istart])) return;
  const uint32_t ostart = ofst.add_state();
  ofst.set_start(ostart);
  const uint32_t final_state = ofst.add_state();
  ofst.set_final(final_state, 0.0f);
  std::vector<Pair> pairs(final_state + 1, Pair{false, 0, INF});
  pairs[final_state] = Pair{true, istart, 0.0f};
  Heap heap;
  heap.pairs = &pairs;
  heap.distance = &distance_2;
  heap.delta = delta;
  heap.push(final_state);
  const float limit = wtimes(distance_2[istart], INF);
  std::vector<uint64_t> rcount;
  while (!heap.data.empty()) {
    const uint32_t state = heap.pop();
    const Pair p = pairs[state];
    const int64_t p_first_real = (p.some ? (int64_t)p.state : -1) + 1;
    const float dd = p.some ? (p.state < distance_2.size() ? distance_2[p.state] : INF) : 0.0f;
    if (natural_less(limit, wtimes(dd, p.w))) continue;
    while ((int64_t)rcount.size() <= p_first_real) rcount.push_back(0);
    rcount[(size_t)p_first_real] += 1;
    if (!p.some) ofst.add_tr((uint32_t)ofst.start, wfst_tr{0u, 0u, 0.0f, state});
    if (!p.some && rcount[(size_t)p_first_real] == nshortest) break;
    if (rcount[(size_t)p_first_real] > nshortest) continue;
    if (!p.some) continue;
    uint32_t n_in = 0;
    const wfst_tr* in = arcs_of(p.state, &n_in);
    for (uint32_t i = 0; i < n_in; ++i) {
      wfst_tr tr = in[i];
      const float weight = wtimes(p.w, tr.weight);
      const uint32_t next = ofst.add_state();
      pairs.push_back(Pair{true, tr.nextstate, weight});
      tr.nextstate = state;
      ofst.add_tr(next, tr);
      heap.push(next);
    }
    const float fw = final_of(p.state);
    if (fw != INF && !props::is_zero(fw)) {
      const float weight = wtimes(p.w, fw);
      const uint32_t next = ofst.add_state();
      pairs.push_back(Pair{false, 0, weight});
      ofst.add_tr(next, wfst_tr{0u, 0u, fw, state});
      heap.push(next);
    }
  }
}

// determinize_with_distance (determinize/determinize_static.rs:24-39): DeterminizeFsa with the default common divisor
// (determinize_fsa_op.rs:43-196) materialised in discovery order (lazy/lazy_fst.rs:226-269), with the distance of every
// new state to the final states (state_table.rs:25-39,79-96).  Host code: the `unique` branch of shortest_path determinizes
// the REVERSED input, a lattice in the decoding case; subsets are a handful of states.  After merging duplicate
// destinations the reference collects a subset from a std HashMap — an order that is unspecified yet part of the tuple's
// identity (the same weighted subset can become several states, differently from run to run); here a subset is kept in
// ascending state order, so every weighted subset is one state.  Strings and weights are the same either way.
struct DetElt {
  uint32_t state;
  float w;
};
inline float quantize(float v, float delta) {  // semirings/semiring.rs:132-145
  if (std::isinf(v)) return v;
  return std::floor((v / delta) + 0.5f) * delta;
}
void determinize_with_distance(const HostCsr& in, int64_t start, uint64_t in_props, const std::vector<float>& in_dist, float delta,
                               HostCsr& out, std::vector<float>& out_dist) {
  if (!(in_props & props::ACCEPTOR)) throw Error("DeterminizeFsaImpl : expected acceptor as argument");  // determinize_fsa_op.rs:138-140
  out = HostCsr{};
  out.offsets.push_back(0);
  out_dist.clear();
  if (start < 0) return;
  std::vector<std::vector<DetElt>> tuples;
  uint64_t n_elts = 0;
  std::unordered_map<uint64_t, std::vector<uint32_t>> by_states;  // tuples with the same state ids (weights compare by ==, KDELTA)
  auto find_state = [&](const std::vector<DetElt>& t) -> uint32_t {
    uint64_t h = 1469598103934665603ull;
    for (const DetElt& e : t) h = (h ^ e.state) * 1099511628211ull;
    std::vector<uint32_t>& cand = by_states[h];
    for (uint32_t id : cand) {
      const std::vector<DetElt>& o = tuples[id];
      bool same = o.size() == t.size();
      for (size_t k = 0; same && k < t.size(); ++k) same = o[k].state == t[k].state && weq(o[k].w, t[k].w);
      if (same) return id;
    }
    const uint32_t id = (uint32_t)tuples.size();
    // (a cyclic weighted acceptor without the twins property has no finite determinization: the reference runs out of
    // memory on it; this library stops at 16 M states / 256 M subset elements)
    n_elts += t.size();
    if (id >= (1u << 24) || n_elts > (1ull << 28)) throw Error("determinize: more than 16 M states (the input does not determinize?)");
    tuples.push_back(t);
    cand.push_back(id);
    float outd = INF;
    for (const DetElt& e : t) outd = wplus(outd, wtimes(e.w, e.state < in_dist.size() ? in_dist[e.state] : INF));
    out_dist.push_back(outd);
    return id;
  };
  find_state({DetElt{(uint32_t)start, 0.0f}});
  std::vector<std::pair<uint32_t, DetElt>> cand;  // (label, destination element) of the state being expanded
  std::vector<DetElt> merged;
  for (uint32_t s = 0; s < tuples.size(); ++s) {
    cand.clear();
    {
      const std::vector<DetElt>& src = tuples[s];
      for (const DetElt& e : src)
        for (uint32_t k = in.offsets[e.state]; k < in.offsets[e.state + 1]; ++k)
          cand.push_back({in.arcs[k].ilabel, DetElt{in.arcs[k].nextstate, wtimes(e.w, in.arcs[k].weight)}});
    }
    // ascending label (the BTreeMap), then ascending state (norm_tr's stable sort): one stable sort on (label, state)
    std::stable_sort(cand.begin(), cand.end(), [](const auto& a, const auto& b) {
      return a.first != b.first ? a.first < b.first : a.second.state < b.second.state;
    });
    float fw = INF;  // compute_final_weight (determinize_fsa_op.rs:101-118)
    for (const DetElt& e : tuples[s]) fw = wplus(fw, wtimes(e.w, in.finals[e.state]));
    for (size_t i = 0; i < cand.size();) {
      size_t j = i;
      float weight = INF;  // common divisor = plus over the label's candidates
      while (j < cand.size() && cand[j].first == cand[i].first) weight = wplus(weight, cand[j++].second.w);
      merged.clear();
      for (size_t k = i; k < j; ++k) {
        if (!merged.empty() && merged.back().state == cand[k].second.state) merged.back().w = wplus(merged.back().w, cand[k].second.w);
        else merged.push_back(cand[k].second);
      }
      for (DetElt& e : merged) e.w = quantize(e.w - weight, delta);  // divide left (tropical_weight.rs:128-131), quantize
      const uint32_t label = cand[i].first;
      out.arcs.push_back(wfst_tr{label, label, weight, find_state(merged)});
      i = j;
    }
    out.offsets.push_back((uint32_t)out.arcs.size());
    out.finals.push_back(props::is_zero(fw) ? INF : fw);
  }
}

// reverse (reverse.rs:33-87) of a small FST on the host: super-initial state 0 with an epsilon arc (weight = the final
// weight) to every final state + 1, state s becomes s + 1, arc (s -> t) becomes (t + 1 -> s + 1) in the order the input's
// arcs are met; the old start state + 1 is final with weight One.  Same arrays as reverse_fst() builds on the GPU.
void reverse_small_host(const HostCsr& in, uint32_t n, int64_t start, HostCsr& out) {
  std::vector<uint32_t> deg((size_t)n + 2, 0);
  for (uint32_t s = 0; s < n; ++s) {
    if (in.finals[s] != INF) deg[1] += 1;  // (arcs of state 0)
    for (uint32_t k = in.offsets[s]; k < in.offsets[s + 1]; ++k) deg[(size_t)in.arcs[k].nextstate + 2] += 1;
  }
  out.offsets.assign((size_t)n + 2, 0);
  for (size_t i = 1; i <= (size_t)n + 1; ++i) out.offsets[i] = out.offsets[i - 1] + deg[i];
  out.arcs.resize(out.offsets[(size_t)n + 1]);
  std::vector<uint32_t> cur(out.offsets.begin(), out.offsets.end() - 1);
  for (uint32_t s = 0; s < n; ++s) {
    if (in.finals[s] != INF) out.arcs[cur[0]++] = wfst_tr{0u, 0u, in.finals[s], s + 1};
    for (uint32_t k = in.offsets[s]; k < in.offsets[s + 1]; ++k) {
      const wfst_tr& a = in.arcs[k];
      out.arcs[cur[(size_t)a.nextstate + 1]++] = wfst_tr{a.ilabel, a.olabel, a.weight, s + 1};
    }
  }
  out.finals.assign((size_t)n + 1, INF);
  if (start >= 0) out.finals[(size_t)start + 1] = 0.0f;
}

// The `unique` branch after the distances and the reversal (shortest_path.rs:143-165,170): distance of the super-initial
// state, determinization of the reversed acceptor, the search on it, connect.
wfst_fst* nbest_unique_from_reversed(wfst_ctx* ctx, const HostCsr& rh, uint64_t rprops, const std::vector<float>& distance,
                                     uint64_t nshortest, float delta) {
  OutFst ofst;
  auto finish = [&]() {
    HostCsr h;
    h.offsets.push_back(0);
    for (const OutFst::St& st : ofst.states) {
      h.arcs.insert(h.arcs.end(), st.trs.begin(), st.trs.end());
      h.offsets.push_back((uint32_t)h.arcs.size());
      h.finals.push_back(st.has_final ? st.final_w : INF);
    }
    return make_host_fst(ctx, (uint32_t)ofst.states.size(), ofst.start, ofst.p, std::move(h));
  };
  float d0 = INF;
  for (uint32_t k = rh.offsets[0]; k < rh.offsets[1]; ++k) {
    const uint32_t state = rh.arcs[k].nextstate - 1;
    if (state < distance.size()) d0 = wplus(d0, wtimes(rh.arcs[k].weight, distance[state]));
  }
  std::vector<float> distance_2;
  distance_2.reserve(distance.size() + 1);
  distance_2.push_back(d0);
  distance_2.insert(distance_2.end(), distance.begin(), distance.end());
  HostCsr dh;
  std::vector<float> distance_3;
  determinize_with_distance(rh, 0, rprops, distance_2, delta, dh, distance_3);
  nbest_search([&](uint32_t st, uint32_t* cnt) {
                 *cnt = dh.offsets[st + 1] - dh.offsets[st];
                 return dh.arcs.data() + dh.offsets[st];
               },
               [&](uint32_t st) { return dh.finals[st]; }, distance_3, nshortest, delta, ofst);
  if (ofst.states.empty()) return finish();
  ofst.connect();
  ofst.p = props::shortest_path(ofst.p, false) & props::ALL;
  return finish();
}

wfst_fst* shortest_path_nbest(wfst_ctx* ctx, const wfst_fst* f, uint64_t nshortest, float delta, bool unique) {
  OutFst ofst;
  auto finish = [&]() {
    HostCsr h;
    h.offsets.push_back(0);
    for (const OutFst::St& st : ofst.states) {
      h.arcs.insert(h.arcs.end(), st.trs.begin(), st.trs.end());
      h.offsets.push_back((uint32_t)h.arcs.size());
      h.finals.push_back(st.has_final ? st.final_w : INF);
    }
    return make_host_fst(ctx, (uint32_t)ofst.states.size(), ofst.start, ofst.p, std::move(h));
  };
  const uint32_t n = f->n_states;
  if (f->start < 0 || n == 0) {  // shortest_distance -> [] ; istart check fails -> FO::new()
    if (unique) {  // the reference determinizes reverse(ifst) whatever its start state is: "expected acceptor" comes first
      std::unique_ptr<wfst_fst> rf(reverse_fst(ctx, f));
      if (!(rf->props & props::ACCEPTOR)) throw Error("DeterminizeFsaImpl : expected acceptor as argument");  // determinize_fsa_op.rs:138-140
    }
    return finish();
  }
  // 1. forward distances (GPU relaxation; exact fixed point == the reference's on grid weights)
  const bool timing = std::getenv("WFST_HOST_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_0 = tnow();
  std::vector<float> distance(n);
  shortest_distance(ctx, f, distance.data(), nullptr);
  auto t_1 = tnow();
  if (unique) {
    // shortest_path.rs:157-165: the reversed FST is determinized (with the distances to its final states) first, so that
    // every string has ONE path; the search then runs on that acceptor.  Acceptors only, as in the reference.
    std::unique_ptr<wfst_fst> rf(reverse_fst(ctx, f));
    ensure_host(rf.get());
    return nbest_unique_from_reversed(ctx, rf->host, rf->props, distance, nshortest, delta);
  }
  // 2. reversed FST (GPU transpose, cached on the handle)
  wfst_fst* mf = const_cast<wfst_fst*>(f);
  std::shared_ptr<RevFst> rev_keep;
  {
    std::lock_guard<std::mutex> lk(f->cache_mu);
    if (!mf->rev_host) mf->rev_host = build_reverse(ctx, f);
    rev_keep = mf->rev_host;
  }
  RevFst& r = *rev_keep;
  std::vector<wfst_tr> scratch;
  auto t_2 = tnow();
  // 3. distance of the super-initial state (shortest_path.rs:143-153)
  float d = INF;
  for (const wfst_tr& a : r.super) {
    const uint32_t state = a.nextstate - 1;
    if (state < distance.size()) d = wplus(d, wtimes(a.weight, distance[state]));
  }
  std::vector<float> distance_2;
  distance_2.reserve((size_t)n + 1);
  distance_2.push_back(d);
  distance_2.insert(distance_2.end(), distance.begin(), distance.end());
  // 4. n_shortest_path(rfst, distance_2, nshortest, delta)   shortest_path.rs:409-518
  nbest_search([&](uint32_t st, uint32_t* cnt) { return rev_arcs_of(ctx, r, st, cnt, scratch); },
               [&](uint32_t st) { return r.finals[st]; }, distance_2, nshortest, delta, ofst);
  if (ofst.states.empty()) return finish();
  auto t_3 = tnow();
  if (timing) {
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    std::fprintf(stderr, "[nbest] distances %.0f us | reverse %.0f us | search %.0f us (%llu segments fetched, %zu states)\n",
                 us(t_0, t_1), us(t_1, t_2), us(t_2, t_3), (unsigned long long)r.fetched_segments, ofst.states.size());
  }
  // 5. connect + property word (shortest_path.rs:512-517)
  ofst.connect();
  ofst.p = props::shortest_path(ofst.p, false) & props::ALL;
  return finish();
}

// nshortest > 1 with unique = true for a batch: the distances and the arrays of all small inputs come to the host in ONE
// launch (export_small_with_distances), then reversal, determinization and the search of every input run on host threads
// (64 lattices one after the other were ~0.8 ms each: a relaxation, a GPU reversal and two read-backs per input).
void shortest_path_nbest_unique_batch(wfst_ctx* ctx, const wfst_fst* const* fsts, size_t n, uint64_t nshortest, float delta,
                                      wfst_fst** outs) {
  std::vector<size_t> idx;
  for (size_t i = 0; i < n; ++i) {
    const wfst_fst* f = fsts[i];
    outs[i] = nullptr;
    if (n < 2 || ctx->batch_in_flight || f->start < 0 || f->n_states == 0 || f->n_states > SMALL_FST_MAX_STATES ||
        f->n_arcs > SMALL_FST_MAX_ARCS)
      continue;  // (a fused batch in flight owns the context's pinned staging: one by one then)
    ensure_device(const_cast<wfst_fst*>(f));
    if (f->has_negative) continue;
    idx.push_back(i);
  }
  std::vector<SmallFstExport> ex;
  export_small_with_distances(ctx, fsts, idx, ex);
  const unsigned n_thr = std::getenv("WFST_HOST_THREADS") ? host_threads(idx.size())
                                                           : (unsigned)std::min<size_t>(std::min(16u, host_threads(1u << 16)), idx.size() / 4);
  parallel_chunks(n_thr, idx.size(), 1, [&](unsigned, uint64_t b, uint64_t e) {
    for (uint64_t j = b; j < e; ++j) {
      if (!ex[j].ok) continue;
      const wfst_fst* f = fsts[idx[j]];
      HostCsr rh;
      reverse_small_host(ex[j].csr, f->n_states, f->start, rh);
      // the reversed FST's ACCEPTOR bit (reverse.rs:80-86): from the input's word, or from its arcs (every arc added to the
      // fresh output keeps or clears it, mutable_fst.rs:235-244)
      bool all_same = true;
      for (const wfst_tr& a : ex[j].csr.arcs) all_same = all_same && a.ilabel == a.olabel;
      const uint64_t rprops = ((f->props & props::ACCEPTOR) || all_same) ? props::ACCEPTOR : props::NOT_ACCEPTOR;
      outs[idx[j]] = nbest_unique_from_reversed(ctx, rh, rprops, ex[j].dist, nshortest, delta);
    }
  });
  for (size_t i = 0; i < n; ++i)
    if (!outs[i]) outs[i] = shortest_path_nbest(ctx, fsts[i], nshortest, delta, true);
}

}  // namespace wfst
