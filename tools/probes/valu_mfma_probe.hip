// valu_mfma_probe.hip -- micro-benchmark: how much VALU work fits beside a saturated MFMA stream on one SIMD (gfx950)?
//   mode 0: every wave runs  [8 MFMA + K VALU ops] x N   (self-interleaved)
//   mode 1: waves 0-3 of the workgroup run MFMAs only, waves 4-7 run VALU only (the co-resident wave of each SIMD)
// 512-thread workgroups, one per CU.  Prints cycles per 8-MFMA group and VALU ops per group.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int MODE>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, int iters, float seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x16 acc[8];
  for (int r = 0; r < 8; ++r) for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + lane * 0.01f + j); b[j] = (__bf16)(seed * 0.5f + j); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = seed + lane + j;
  const bool do_mfma = MODE == 0 || wave < 4, do_valu = MODE == 0 || wave >= 4;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[r], 0, 0, 0);
    }
    if (do_valu) {
      // K "elements" of the GN+SiLU prologue: fma, mul, exp, add, rcp, mul (2 transcendental + 4 plain)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float x = v[k & 7] * 1.0001f + 0.5f;
        float e = __expf(-x);
        v[k & 7] = x * __builtin_amdgcn_rcpf(1.0f + e);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 8; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int K, int MODE>
void run(float* out, unsigned long long* cyc, int iters) {
  hipLaunchKernelGGL((probe<K, MODE>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<K, MODE>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double mf = MODE == 0 ? 8.0 : 4.0;  // waves issuing MFMAs
  const double tf = 256.0 * mf * iters * 8 * 32768.0 / (ms * 1e-3) / 1e12;
  printf("mode %d K=%2d: %.3f ms  MFMA rate %.0f TFLOP/s | ticks per iteration: wave0 %.1f wave4 %.1f | prologue elements per MFMA-wave-iteration %d\n",
         MODE, K, ms, tf, (double)h[0] / iters, (double)h[4] / iters, K);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  const int iters = 20000;
  run<0, 0>(out, cyc, iters); run<2, 0>(out, cyc, iters); run<4, 0>(out, cyc, iters); run<8, 0>(out, cyc, iters); run<16, 0>(out, cyc, iters);
  run<0, 1>(out, cyc, iters); run<4, 1>(out, cyc, iters); run<8, 1>(out, cyc, iters); run<16, 1>(out, cyc, iters); run<32, 1>(out, cyc, iters);
  return 0;
}
