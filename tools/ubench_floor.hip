// What a launch of the mailbox kernel's SHAPE costs before it does anything useful (MI355X): 245 workgroups of 1024 threads
// with ~157 KB of LDS, then one, two, three dependent memory trips with a barrier after each, then a 32 KB write-back.
// Run under rocprofv3 --kernel-trace --stats: one kernel name per variant.  hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_floor tools/ubench_floor.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int TRIPS, bool STORE, int THREADS, bool BIGLDS>
__global__ void __launch_bounds__(THREADS) shape_kernel(const uint64_t* __restrict__ a, const uint32_t* __restrict__ idx, uint64_t* __restrict__ out, uint32_t n_words) {
  extern __shared__ __align__(16) unsigned char dyn[];
  __shared__ uint64_t l[4096];
  constexpr int R = 4096 / THREADS;
  const uint32_t tid = threadIdx.x, j = blockIdx.x;
  if (BIGLDS && tid == 0) dyn[j & 1023] = 1;
  if (TRIPS == 0) return;
  uint64_t k[R];
  for (int r = 0; r < R; ++r) k[r] = a[(size_t)j * 4096 + tid + THREADS * r];
  uint32_t o[R];
  for (int r = 0; r < R; ++r) o[r] = idx[(size_t)j * 4096 + tid + THREADS * r];
  for (int r = 0; r < R; ++r) l[tid + THREADS * r] = k[r] + o[r];
  __syncthreads();
  uint64_t acc = l[(tid * 7) & 4095];
  if (TRIPS >= 2) {  // a dependent trip: 8 loads per thread at addresses that come out of LDS
    uint64_t m[8];
    for (int u = 0; u < 8; ++u) m[u] = a[(acc + (uint64_t)u * 977 + tid * 13) % n_words];
    for (int u = 0; u < 8; ++u) atomicMin((unsigned long long*)&l[m[u] & 4095], (unsigned long long)m[u]);
    __syncthreads();
    acc += l[(tid * 5) & 4095];
  }
  if (TRIPS >= 3) {
    uint64_t m[8];
    for (int u = 0; u < 8; ++u) m[u] = a[(acc + (uint64_t)u * 1013 + tid * 29) % n_words];
    for (int u = 0; u < 8; ++u) atomicMin((unsigned long long*)&l[m[u] & 4095], (unsigned long long)m[u]);
    __syncthreads();
    acc += l[(tid * 3) & 4095];
  }
  if (STORE) {
    for (int r = 0; r < R; ++r) out[(size_t)j * 4096 + tid + THREADS * r] = l[tid + THREADS * r] + acc;
  } else if (acc == 0x1234567812345678ull) {
    out[0] = acc;
  }
}

template <int TRIPS, bool STORE, int THREADS, bool BIGLDS>
int run(const char* name, int nb, const uint64_t* a, const uint32_t* idx, uint64_t* out, uint32_t n_words, hipStream_t st) {
  auto k = shape_kernel<TRIPS, STORE, THREADS, BIGLDS>;
  const size_t dyn = BIGLDS ? 100 * 1024 : 0;
  if (BIGLDS) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  for (int i = 0; i < 20; ++i) k<<<nb, THREADS, dyn, st>>>(a, idx, out, n_words);
  CK(hipStreamSynchronize(st));
  const int reps = 300;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) k<<<nb, THREADS, dyn, st>>>(a, idx, out, n_words);
  CK(hipStreamSynchronize(st));
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("%-44s %7.2f us launch to launch (%d back to back)\n", name, us, reps);
  return 0;
}

int main() {
  const int nb = 245;
  const uint32_t n_words = nb * 4096;
  uint64_t *a, *out;
  uint32_t* idx;
  CK(hipMalloc(&a, (size_t)n_words * 8));
  CK(hipMalloc(&out, (size_t)n_words * 8));
  CK(hipMalloc(&idx, (size_t)n_words * 4));
  std::vector<uint64_t> h(n_words);
  uint64_t x = 88172645463325252ull;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x; }
  CK(hipMemcpy(a, h.data(), (size_t)n_words * 8, hipMemcpyHostToDevice));
  CK(hipMemset(idx, 1, (size_t)n_words * 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  if (run<0, false, 1024, false>("empty, 1024 threads", nb, a, idx, out, n_words, st)) return 1;
  if (run<0, false, 1024, true>("empty, 1024 threads, 157 KB LDS", nb, a, idx, out, n_words, st)) return 1;
  if (run<0, false, 512, true>("empty, 512 threads, 157 KB LDS", nb, a, idx, out, n_words, st)) return 1;
  if (run<0, false, 256, true>("empty, 256 threads, 157 KB LDS", nb, a, idx, out, n_words, st)) return 1;
  if (run<1, false, 1024, true>("1 trip (48 KB per workgroup)", nb, a, idx, out, n_words, st)) return 1;
  if (run<2, false, 1024, true>("2 trips", nb, a, idx, out, n_words, st)) return 1;
  if (run<3, false, 1024, true>("3 trips", nb, a, idx, out, n_words, st)) return 1;
  if (run<3, true, 1024, true>("3 trips + 32 KB stored per workgroup", nb, a, idx, out, n_words, st)) return 1;
  if (run<3, true, 512, true>("3 trips + store, 512 threads", nb, a, idx, out, n_words, st)) return 1;
  if (run<3, true, 256, true>("3 trips + store, 256 threads", nb, a, idx, out, n_words, st)) return 1;
  if (run<1, true, 1024, true>("1 trip + store", nb, a, idx, out, n_words, st)) return 1;
  return 0;
}
