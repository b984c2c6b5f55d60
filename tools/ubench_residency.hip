// How many 64-thread workgroups run concurrently per CU, as a function of scratch / LDS / VGPR use?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LDS_BYTES, int SCRATCH_WORDS, int BLOCK>
__global__ void __launch_bounds__(BLOCK) spin(unsigned long long cycles, uint32_t* sink, int idx) {
  __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 1];
  volatile uint32_t scratch[SCRATCH_WORDS > 0 ? SCRATCH_WORDS : 1];
  if (SCRATCH_WORDS > 0) for (int i = 0; i < SCRATCH_WORDS; ++i) scratch[(i * 7 + idx) % SCRATCH_WORDS] = i;
  if (LDS_BYTES > 0) lds[threadIdx.x] = (char)idx;
  unsigned long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  uint32_t v = LDS_BYTES > 0 ? lds[(threadIdx.x + 1) % 64] : 0;
  if (SCRATCH_WORDS > 0) v += scratch[idx % SCRATCH_WORDS];
  if (v == 0x12345) sink[0] = v;
}

template <int L, int S, int B>
int run(const char* name, uint32_t* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned long long cyc = 200000;  // ~0.1 ms
  for (int nb : {256, 2048, 8192, 16384}) {
    spin<L, S, B><<<nb, B>>>(cyc, sink, 3);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    spin<L, S, B><<<nb, B>>>(cyc, sink, 3);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s blocks %6d  %7.3f ms\n", name, nb, ms);
  }
  return 0;
}

int main() {
  uint32_t* sink; CK(hipMalloc(&sink, 64));
  run<0, 0, 64>("64 thr, no lds, no scratch", sink);
  run<4864, 0, 64>("64 thr, 4.8KB lds", sink);
  run<0, 9, 64>("64 thr, 36B scratch", sink);
  run<4864, 9, 64>("64 thr, lds + scratch", sink);
  run<0, 0, 256>("256 thr, none", sink);
  return 0;
}
