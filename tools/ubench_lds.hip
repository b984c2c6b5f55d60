// LDS atomic throughput of one full workgroup per CU (1024 threads), the access patterns of the mailbox kernel's phases:
//   slot allocation  atomicAdd with return on one of 245 counters chosen at random (expansion: one per candidate)
//   apply            atomicMin on a 64-bit key of 4096 chosen at random (inbox messages), with and without the old value
// Reported: nanoseconds per wave-level instruction (64 lanes), from the kernel's duration with K operations per thread.
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_lds tools/ubench_lds.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t rng(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <int MODE>
__global__ void __launch_bounds__(1024) lds_kernel(uint32_t* out, int k_ops, uint32_t n_cnt) {
  __shared__ unsigned long long lkey[4096];
  __shared__ uint32_t cnt[2048];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < 4096; i += 1024) lkey[i] = ~0ull;
  for (uint32_t i = tid; i < 2048; i += 1024) cnt[i] = 0;
  __syncthreads();
  uint32_t s = tid * 2654435761u + blockIdx.x * 97u + 12345u, acc = 0;
  for (int i = 0; i < k_ops; ++i) {
    const uint32_t r = rng(s);
    if (MODE == 0) acc += atomicAdd(&cnt[r % n_cnt], 1u);                      // slot allocation (returned)
    if (MODE == 1) atomicAdd(&cnt[r % n_cnt], 1u);                             // counting only
    if (MODE == 2) atomicMin(&lkey[r & 4095u], ((unsigned long long)r << 32) | tid);          // apply, old value dropped
    if (MODE == 3) acc += (uint32_t)atomicMin(&lkey[r & 4095u], ((unsigned long long)r << 32) | tid);  // apply, old value used
    if (MODE == 4) lkey[r & 4095u] = ((unsigned long long)r << 32) | tid;      // plain 8-byte store
    if (MODE == 5) acc += (uint32_t)lkey[r & 4095u];                           // plain 8-byte load
    if (MODE == 6) acc += atomicAdd(&cnt[(r % n_cnt) * 8u + (tid & 7u)], 1u);  // slot allocation, 8 sub-counters per destination
    if (MODE == 7) acc += r;                                                   // the loop alone
  }
  __syncthreads();
  if (acc == 0x12345u) out[blockIdx.x] = acc + (uint32_t)lkey[tid] + cnt[tid];
}

template <int MODE>
int run(const char* name, uint32_t* out, uint32_t n_cnt) {
  const int k = 256, nb = 245;
  lds_kernel<MODE><<<nb, 1024>>>(out, k, n_cnt);
  CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    lds_kernel<MODE><<<nb, 1024>>>(out, k, n_cnt);
    CK(hipDeviceSynchronize());
    best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  // one CU: 16 waves x k wave-instructions
  printf("%-58s %8.1f us per launch, %6.1f ns per wave instruction per CU\n", name, best, (best - 8.0) * 1e3 / (16.0 * k));
  return 0;
}

int main() {
  uint32_t* out;
  CK(hipMalloc(&out, 4096));
  if (run<7>("loop alone", out, 245)) return 1;
  if (run<0>("atomicAdd returned, 245 counters", out, 245)) return 1;
  if (run<1>("atomicAdd not returned, 245 counters", out, 245)) return 1;
  if (run<6>("atomicAdd returned, 245 x 8 sub-counters", out, 245)) return 1;
  if (run<0>("atomicAdd returned, 2048 counters", out, 2048)) return 1;
  if (run<2>("atomicMin u64, 4096 keys, not returned", out, 245)) return 1;
  if (run<3>("atomicMin u64, 4096 keys, returned", out, 245)) return 1;
  if (run<4>("store 8 B, 4096 keys", out, 245)) return 1;
  if (run<5>("load 8 B, 4096 keys", out, 245)) return 1;
  return 0;
}
