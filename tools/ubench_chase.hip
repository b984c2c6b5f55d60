// Micro-benchmark: the latency of ONE dependent global load seen by a lone wavefront (what a level of the string o T kernel
// costs at least: the next state's arc row can only be asked for once the current row is there).  A random cyclic
// permutation is chased through arrays of several sizes: inside one XCD's L2, inside the Infinity Cache, beyond it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
// one lane chases; the 16-byte "row" it reads stands for the row of a state (the kernel reads 16 B per lane of an 80-byte row)
__global__ void chase(const uint4* __restrict__ a, uint32_t hops, uint32_t start, uint32_t* out) {
  uint32_t i = start;
  for (uint32_t h = 0; h < hops; ++h) i = a[i].x;
  out[0] = i;
}
int main() {
  std::mt19937_64 rng(7);
  for (size_t mb : {1, 16, 64, 160, 1024}) {
    const size_t n = mb * (1u << 20) / 16;
    std::vector<uint32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<uint4> h(n);
    for (size_t k = 0; k < n; ++k) h[perm[k]] = make_uint4(perm[(k + 1) % n], 0, 0, 0);
    uint4* d; uint32_t* out;
    CK(hipMalloc(&d, n * 16)); CK(hipMalloc(&out, 4));
    CK(hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice));
    const uint32_t hops = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      chase<<<1, 64>>>(d, hops, perm[rep * 977 % n], out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    printf("array %5zu MB: %7.1f ns per dependent load (lone wave, %u hops)\n", mb, best * 1e6 / hops, hops);
    CK(hipFree(d)); CK(hipFree(out));
  }
  return 0;
}
