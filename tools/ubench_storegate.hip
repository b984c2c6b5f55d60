// Does an in-flight store delay a later dependent load's s_waitcnt? (single wave, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k(uint32_t* chain, uint32_t* outbuf, uint64_t* cyc, int iters, uint32_t stride_words) {
  uint32_t idx = threadIdx.x == 0 ? 0 : 0;
  unsigned long long t0 = clock64();
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 1) outbuf[(size_t)i * stride_words + threadIdx.x] = i;                 // plain store, new line each iter
    if (MODE == 2) outbuf[(size_t)(i & 7) * 64 + threadIdx.x] = i;                     // plain store, same few lines
    if (MODE == 3) __hip_atomic_store(&outbuf[(size_t)i * stride_words + threadIdx.x], (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 4) atomicMin(&outbuf[(size_t)i * stride_words + threadIdx.x], (uint32_t)i);   // no-return atomic
    idx = chain[idx];   // dependent load chain (L2 resident)
    acc += idx;
  }
  unsigned long long t1 = clock64();
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = acc; }
}

int main() {
  const int iters = 2000; const uint32_t stride_words = 4096;  // 16 KB apart
  uint32_t *chain, *outbuf; uint64_t* cyc;
  CK(hipMalloc(&chain, 1 << 20)); CK(hipMalloc(&outbuf, (size_t)iters * stride_words * 4 + 4096)); CK(hipMalloc(&cyc, 16));
  uint32_t h[1 << 18];
  for (int i = 0; i < (1 << 18); ++i) h[i] = (uint32_t)((i * 2654435761u + 12345u) % (1 << 18));
  CK(hipMemcpy(chain, h, 1 << 20, hipMemcpyHostToDevice));
  const char* names[] = {"load chain only", "+ plain store to a fresh line", "+ plain store to warm lines", "+ sc1 atomic store fresh line", "+ atomicMin(noret) fresh line"};
  for (int rep = 0; rep < 2; ++rep)
  for (int m = 0; m < 5; ++m) {
    CK(hipMemset(outbuf, 0xFF, (size_t)iters * stride_words * 4));
    CK(hipDeviceSynchronize());
    switch (m) {
      case 0: k<0><<<1, 64>>>(chain, outbuf, cyc, iters, stride_words); break;
      case 1: k<1><<<1, 64>>>(chain, outbuf, cyc, iters, stride_words); break;
      case 2: k<2><<<1, 64>>>(chain, outbuf, cyc, iters, stride_words); break;
      case 3: k<3><<<1, 64>>>(chain, outbuf, cyc, iters, stride_words); break;
      case 4: k<4><<<1, 64>>>(chain, outbuf, cyc, iters, stride_words); break;
    }
    CK(hipDeviceSynchronize());
    uint64_t c[2]; CK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
    if (rep == 1) printf("%-36s %8.1f cycles / iteration\n", names[m], (double)c[0] / iters);
  }
  return 0;
}
