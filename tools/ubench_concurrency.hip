// Do a long, narrow kernel (64 single-wave workgroups, like the fused compose->shortest-path batch kernel)
// and a chain of short, GPU-wide kernels (like the relaxation sweeps) overlap when issued on two HIP streams?
// hipcc --offload-arch=gfx950 -O3 tools/ubench_concurrency.hip -o /tmp/ubench_conc && /tmp/ubench_conc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void long_narrow(unsigned long long ticks, unsigned* sink) {
  const unsigned long long t0 = wall_clock64();
  unsigned acc = 0;
  while (wall_clock64() - t0 < ticks) acc += 1;  // 100 MHz constant clock
  if (acc == 0xFFFFFFFFu) *sink = acc;
}
__global__ void __launch_bounds__(256) short_wide(const unsigned* in, unsigned* out, unsigned n) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i] + 1;
}

int main() {
  const unsigned n = 1u << 20;
  unsigned *a, *b, *sink;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, n * 4));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  const unsigned long long ticks = 60000;  // 600 us
  const int chain = 100;
  for (int grid : {2048, 1024, 256}) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = now();
      long_narrow<<<64, 64, 0, s1>>>(ticks, sink);
      CK(hipStreamSynchronize(s1));
      auto t1 = now();
      for (int i = 0; i < chain; ++i) short_wide<<<grid, 256, 0, s2>>>(a, b, n);
      CK(hipStreamSynchronize(s2));
      auto t2 = now();
      long_narrow<<<64, 64, 0, s1>>>(ticks, sink);
      for (int i = 0; i < chain; ++i) short_wide<<<grid, 256, 0, s2>>>(a, b, n);
      CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
      auto t3 = now();
      // chain first, then the long one
      for (int i = 0; i < chain; ++i) short_wide<<<grid, 256, 0, s2>>>(a, b, n);
      long_narrow<<<64, 64, 0, s1>>>(ticks, sink);
      CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
      auto t4 = now();
      if (rep) printf("grid %4d: long alone %.0f us, chain(%d) alone %.0f us, both (long first) %.0f us, both (chain first) %.0f us\n", grid,
                      us(t0, t1), chain, us(t1, t2), us(t2, t3), us(t3, t4));
    }
  }
  return 0;
}
