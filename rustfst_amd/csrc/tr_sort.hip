// tr_sort.hip — tr_sort(fst, ILabelCompare | OLabelCompare) on the device, in place.
// Reference: rustfst/src/algorithms/tr_sort.rs:13-62 (per-state STABLE sort_by on one label, then the
// property rule of tr_sort_properties: mutable_properties.rs arcsort rule, restated in vector_fst.cpp).
//
// HBM-bound integer work: every arc is read once (16 B) and written once (16 B + 8 B of the derived
// {weight,next} array).  Stability is obtained by making the key unique: key = (label, position).
//   * states with <= RANK_MAX arcs: rank sort.  GROUP=16 lanes own one state; the group's keys travel by
//     __shfl inside the 16-lane group, rank_i = #{j : key_j < key_i}; arcs are scattered to begin + rank.
//     O(deg^2 / 16) shuffles per lane — deg is ~10 for lexicon/grammar transducers.
//   * larger states: listed by the same kernel and sorted by rocPRIM's segmented radix sort over exactly
//     those segments, on the packed (label, position) keys restricted to the bits in use; a gather kernel
//     then moves the 16-byte arcs.
#include <cstring>

#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "common.h"
#include "fst_props.h"

namespace wfst {
namespace {

constexpr int GROUP = 16;
constexpr uint32_t RANK_MAX = 256;

__device__ __forceinline__ uint64_t key_of(const wfst_tr& a, uint32_t pos, int by_olabel) {
  return ((uint64_t)(by_olabel ? a.olabel : a.ilabel) << 32) | pos;
}

// one 16-lane group per state
__global__ __launch_bounds__(256) void trsort_rank_kernel(const uint32_t* __restrict__ offsets,
                                                         const wfst_tr* __restrict__ in, wfst_tr* __restrict__ out,
                                                         uint2* __restrict__ wn, uint32_t n_states, int by_olabel,
                                                         uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count,
                                                         uint32_t* __restrict__ big_max_deg) {
  const uint32_t gl = threadIdx.x & (GROUP - 1);
  const uint32_t groups = (gridDim.x * blockDim.x) / GROUP;
  for (uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) / GROUP; s < n_states; s += groups) {
    const uint32_t b = offsets[s], e = offsets[s + 1], deg = e - b;
    if (deg > RANK_MAX) {
      if (gl == 0) {  // rare: a handful of hub states per FST
        big_list[atomicAdd(big_count, 1u)] = s;
        atomicMax(big_max_deg, deg);
      }
      continue;
    }
    if (deg <= GROUP) {  // common case: the whole state lives in one register per lane
      const bool live = gl < deg;
      wfst_tr a{};
      if (live) a = in[b + gl];
      const uint64_t k = live ? key_of(a, gl, by_olabel) : ~0ull;
      uint32_t rank = 0;
#pragma unroll
      for (int t = 0; t < GROUP; ++t) rank += __shfl(k, t, GROUP) < k;
      if (live) {
        out[b + rank] = a;
        wn[b + rank] = make_uint2(__float_as_uint(a.weight), a.nextstate);
      }
      continue;
    }
    for (uint32_t i0 = 0; i0 < deg; i0 += GROUP) {
      const bool live = i0 + gl < deg;
      wfst_tr a{};
      if (live) a = in[b + i0 + gl];
      const uint64_t k = live ? key_of(a, i0 + gl, by_olabel) : ~0ull;
      uint32_t rank = 0;
      for (uint32_t j0 = 0; j0 < deg; j0 += GROUP) {
        uint64_t kj = ~0ull;
        if (j0 + gl < deg) {
          const wfst_tr& o = in[b + j0 + gl];
          kj = ((uint64_t)(by_olabel ? o.olabel : o.ilabel) << 32) | (j0 + gl);
        }
#pragma unroll
        for (int t = 0; t < GROUP; ++t) rank += __shfl(kj, t, GROUP) < k;
      }
      if (live) {
        out[b + rank] = a;
        wn[b + rank] = make_uint2(__float_as_uint(a.weight), a.nextstate);
      }
    }
  }
}

// big states only: packed keys (label << pos_bits | position), one workgroup per state
__global__ __launch_bounds__(256) void trsort_bigkeys_kernel(const uint32_t* __restrict__ offsets,
                                                            const wfst_tr* __restrict__ in,
                                                            const uint32_t* __restrict__ big_list, uint32_t n_big,
                                                            uint64_t* __restrict__ keys, uint32_t* __restrict__ seg_begin,
                                                            uint32_t* __restrict__ seg_end, int by_olabel, int pos_bits) {
  for (uint32_t i = blockIdx.x; i < n_big; i += gridDim.x) {
    const uint32_t s = big_list[i], b = offsets[s], e = offsets[s + 1];
    if (threadIdx.x == 0) {
      seg_begin[i] = b;
      seg_end[i] = e;
    }
    for (uint32_t p = b + threadIdx.x; p < e; p += blockDim.x)
      keys[p] = ((uint64_t)(by_olabel ? in[p].olabel : in[p].ilabel) << pos_bits) | (p - b);
  }
}

__global__ __launch_bounds__(256) void trsort_biggather_kernel(const uint32_t* __restrict__ offsets,
                                                              const wfst_tr* __restrict__ in, wfst_tr* __restrict__ out,
                                                              uint2* __restrict__ wn,
                                                              const uint32_t* __restrict__ big_list, uint32_t n_big,
                                                              const uint64_t* __restrict__ sorted_keys, int pos_bits) {
  for (uint32_t i = blockIdx.x; i < n_big; i += gridDim.x) {
    const uint32_t s = big_list[i], b = offsets[s], e = offsets[s + 1];
    for (uint32_t p = b + threadIdx.x; p < e; p += blockDim.x) {
      const wfst_tr a = in[b + (uint32_t)(sorted_keys[p] & ((1ull << pos_bits) - 1))];
      out[p] = a;
      wn[p] = make_uint2(__float_as_uint(a.weight), a.nextstate);
    }
  }
}

}  // namespace

// tr_sort_properties + set_properties_with_mask as applied by tr_sort (tr_sort.rs:21-33)
uint64_t tr_sort_props(uint64_t in, bool ilabel_cmp) {
  using namespace props;
  const uint64_t arcsort_mask = ALL & ~(I_LABEL_SORTED | NOT_I_LABEL_SORTED | O_LABEL_SORTED | NOT_O_LABEL_SORTED);
  uint64_t out = (in & arcsort_mask) | (ilabel_cmp ? I_LABEL_SORTED : O_LABEL_SORTED);
  if (in & ACCEPTOR) out |= ilabel_cmp ? O_LABEL_SORTED : I_LABEL_SORTED;
  return out;
}

void tr_sort_device(wfst_ctx* ctx, wfst_fst* f, bool ilabel_cmp) {
  using namespace props;
  HIP_CHECK(hipSetDevice(ctx->device));
  const uint64_t sorted_bit = ilabel_cmp ? I_LABEL_SORTED : O_LABEL_SORTED;
  const bool already = (f->props & sorted_bit) != 0;  // a stable sort of a sorted list is the identity
  if (!already && f->n_arcs > 1) {
    ensure_device(f);
    const uint32_t n = f->n_states;
    const int by_olabel = ilabel_cmp ? 0 : 1;
    wfst_tr* arcs = const_cast<wfst_tr*>(f->dev.arcs);
    uint2* wn = const_cast<uint2*>(f->dev.wn);
    DBuf<wfst_tr> tmp(*ctx->pool, f->n_arcs);
    DBuf<uint32_t> big_list(*ctx->pool, n);
    DBuf<uint32_t> big_count(*ctx->pool, 2);  // {number of big states, their maximum degree}
    HIP_CHECK(hipMemcpyAsync(tmp.p, arcs, f->n_arcs * sizeof(wfst_tr), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_CHECK(hipMemsetAsync(big_count.p, 0, 2 * sizeof(uint32_t), ctx->stream));
    const int blocks = (int)std::min<uint64_t>(((uint64_t)n * GROUP + 255) / 256, (uint64_t)ctx->n_cus * 32);
    trsort_rank_kernel<<<blocks, 256, 0, ctx->stream>>>(f->dev.offsets, tmp.p, arcs, wn, n, by_olabel, big_list.p,
                                                        big_count.p, big_count.p + 1);
    HIP_CHECK(hipGetLastError());
    uint32_t* h_big = (uint32_t*)ctx->pinned.get(2 * sizeof(uint32_t));
    HIP_CHECK(hipMemcpyAsync(h_big, big_count.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const uint32_t n_big = h_big[0];
    if (n_big) {
      int pos_bits = 1;  // positions inside a state need ceil(log2(max degree)) bits; labels sit right above them
      while ((1ull << pos_bits) < h_big[1]) ++pos_bits;
      const unsigned end_bit = (unsigned)pos_bits + 32u;
      DBuf<uint64_t> keys(*ctx->pool, f->n_arcs), keys_out(*ctx->pool, f->n_arcs);
      DBuf<uint32_t> seg_b(*ctx->pool, n_big), seg_e(*ctx->pool, n_big);
      const int bblocks = (int)std::min<uint32_t>(n_big, (uint32_t)ctx->n_cus * 8);
      trsort_bigkeys_kernel<<<bblocks, 256, 0, ctx->stream>>>(f->dev.offsets, tmp.p, big_list.p, n_big, keys.p, seg_b.p,
                                                              seg_e.p, by_olabel, pos_bits);
      HIP_CHECK(hipGetLastError());
      size_t temp_bytes = 0;
      HIP_CHECK(rocprim::segmented_radix_sort_keys(nullptr, temp_bytes, keys.p, keys_out.p, (unsigned)f->n_arcs, n_big,
                                                   seg_b.p, seg_e.p, 0u, end_bit, ctx->stream));
      DBuf<uint8_t> temp(*ctx->pool, temp_bytes);
      HIP_CHECK(rocprim::segmented_radix_sort_keys(temp.p, temp_bytes, keys.p, keys_out.p, (unsigned)f->n_arcs, n_big,
                                                   seg_b.p, seg_e.p, 0u, end_bit, ctx->stream));
      trsort_biggather_kernel<<<bblocks, 256, 0, ctx->stream>>>(f->dev.offsets, tmp.p, arcs, wn, big_list.p, n_big,
                                                                keys_out.p, pos_bits);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    if (f->has_host) {  // the host mirror (and the cached reverse) described the old arc order
      f->host = HostCsr{};
      f->has_host = false;
    }
    f->rev_host.reset();
    f->rev_dev.reset();  // arc positions changed
    f->anext.reset();
  }
  f->props = tr_sort_props(f->props, ilabel_cmp);
}

}  // namespace wfst
