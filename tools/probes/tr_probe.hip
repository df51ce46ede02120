// ds_read_b64_tr_b16 semantics probe (gfx950): LDS holds element index i at element i; every lane reads with the address pattern of
// wgrad_dma_kernel's fragment reads (lane l: chunk (l & 15) of block (l >> 4), 8 bytes per chunk, 128 bytes per block) and the four
// received elements are printed -- lane l, element j should be element (l >> 4) * 64 + j * 16 + (l & 15).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (l >> 4) * 64 + (l & 15) * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d;
  short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      const int want = (l >> 4) * 64 + j * 16 + (l & 15);
      printf(" %4d%s", h[l * 4 + j], h[l * 4 + j] == want ? "" : "!");
      bad += h[l * 4 + j] != want;
    }
    printf("\n");
  }
  printf("mismatches against the assumed layout: %d\n", bad);
  return 0;
}
