// Micro-benchmark: gathering 80-byte arc rows (10 x 8 B, 16 lanes per row) at random from an array larger than the
// Infinity Cache / inside it: rows at arbitrary 8-byte offsets vs rows that never straddle a 128-byte line vs 64-byte-aligned.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(1024) gather(const uint2* __restrict__ wn, const uint32_t* __restrict__ starts, uint32_t n_rows, uint32_t deg, unsigned long long* sink) {
  const uint32_t sub = threadIdx.x & 15u;
  const uint32_t groups = gridDim.x * (blockDim.x / 16);
  unsigned long long acc = 0;
  for (uint32_t r = blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4); r < n_rows; r += groups * 8) {
    uint2 a[8];
    for (int u = 0; u < 8; ++u) {
      a[u] = make_uint2(0, 0);
      const uint32_t rr = r + groups * u;
      if (rr < n_rows && sub < deg) a[u] = wn[(size_t)starts[rr] + sub];
    }
    for (int u = 0; u < 8; ++u) acc += a[u].x + a[u].y;
  }
  if (acc == 0x123456789ull) sink[0] = acc;
}
// the same rows with 16 bytes per lane: 8 lanes per row (5 of them carry the 10 arcs), 8 rows per 64-lane instruction
__global__ void __launch_bounds__(1024) gather16(const uint2* __restrict__ wn, const uint32_t* __restrict__ starts, uint32_t n_rows, uint32_t deg, unsigned long long* sink) {
  const uint32_t sub = threadIdx.x & 7u;
  const uint32_t groups = gridDim.x * (blockDim.x / 8);
  unsigned long long acc = 0;
  for (uint32_t r = blockIdx.x * (blockDim.x / 8) + (threadIdx.x >> 3); r < n_rows; r += groups * 4) {
    uint4 a[4];
    for (int u = 0; u < 4; ++u) {
      a[u] = make_uint4(0, 0, 0, 0);
      const uint32_t rr = r + groups * u;
      if (rr < n_rows && 2 * sub < deg) {
        const uint2* p = wn + (size_t)starts[rr] + 2 * sub;
        const uint2 x = p[0], y = p[1];  // (8-byte aligned rows: two dwordx2 loads that the compiler may not merge)
        a[u] = make_uint4(x.x, x.y, y.x, y.y);
      }
    }
    for (int u = 0; u < 4; ++u) acc += a[u].x + a[u].y + a[u].z + a[u].w;
  }
  if (acc == 0x123456789ull) sink[0] = acc;
}
int main() {
  const uint32_t deg = 10;
  for (int big = 0; big < 2; ++big) {
    const size_t n_units = big ? (size_t)96 << 20 : (size_t)10 << 20;  // 8-byte units: 768 MB / 80 MB
    uint2* wn; unsigned long long* sink; uint32_t* starts;
    CK(hipMalloc(&wn, n_units * 8)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(wn, 1, n_units * 8));
    const uint32_t n_rows = 500000;
    CK(hipMalloc(&starts, n_rows * 4));
    const char* names[] = {"rows at random 8-byte offsets", "rows inside one 128-byte line (start % 16 <= 6)", "rows 64-byte aligned", "rows 128-byte aligned", "rows in ascending order, random 8-byte offsets (sorted gather)", "8 lanes x 2 arcs per row, random offsets", "8 lanes x 2 arcs per row, 128-byte aligned rows"};
    for (int mode = 0; mode < 7; ++mode) {
      std::vector<uint32_t> h(n_rows);
      uint64_t s = 12345 + mode;
      for (uint32_t i = 0; i < n_rows; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        uint64_t u = (s >> 20) % (n_units - 32);
        if (mode == 1) { u = (u & ~15ull) | ((s >> 7) % 7); }
        if (mode == 2) u &= ~7ull;
        if (mode == 3 || mode == 6) u &= ~15ull;
        h[i] = (uint32_t)u;
      }
      if (mode == 4) std::sort(h.begin(), h.end());
      CK(hipMemcpy(starts, h.data(), n_rows * 4, hipMemcpyHostToDevice));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        if (mode >= 5) gather16<<<245, 1024>>>(wn, starts, n_rows, deg, sink);
        else gather<<<245, 1024>>>(wn, starts, n_rows, deg, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("%s array %4zu MB: %-70s %7.2f us  %6.1f G rows/s  %6.1f GB/s useful\n", big ? "large" : "small", n_units * 8 >> 20, names[mode], best * 1e3, n_rows / best * 1e-6, n_rows * 80.0 / best * 1e-6);
    }
    CK(hipFree(wn)); CK(hipFree(starts)); CK(hipFree(sink));
  }
  return 0;
}
