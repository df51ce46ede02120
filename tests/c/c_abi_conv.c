/* c_abi_conv.c -- the C ABI of libcvvae_hip.so used from plain C (no Python, no torch): pack a 1x1x1 weight, run
 * cvvae_conv_fwd on device buffers, check against a host-side dot product.  Built and run by tests/test_gpu_c_abi.py:
 *   gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c/c_abi_conv.c -Lcvvae_amd -lcvvae_hip
 *       -L/opt/rocm/lib -lamdhip64 -lm -o /tmp/c_abi_conv
 * (the HIP runtime is only here for hipMalloc / hipMemcpy; every library entry point is plain C.) */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cvvae.h"

static uint16_t f2bf(float f) { /* round to nearest even */
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(void) {
  enum { NPIX = 600, CIN = 128, COUT = 64 };
  if (cvvae_abi_version() != CVVAE_ABI_VERSION) { printf("ABI mismatch\n"); return 2; }
  uint16_t* hx = malloc(NPIX * CIN * 2); uint16_t* hw = malloc(COUT * CIN * 2); float hb[COUT];
  srand(7);
  for (int i = 0; i < NPIX * CIN; ++i) hx[i] = f2bf((float)rand() / RAND_MAX * 2.f - 1.f);
  for (int i = 0; i < COUT * CIN; ++i) hw[i] = f2bf(((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f);
  for (int i = 0; i < COUT; ++i) hb[i] = 0.01f * i;
  void *dx, *dw, *dpk, *dy; float* db;
  const size_t pk = cvvae_packed_weight_bytes(COUT, CIN, 1);
  if (hipMalloc(&dx, NPIX * CIN * 2) || hipMalloc(&dw, COUT * CIN * 2) || hipMalloc(&dpk, pk) || hipMalloc(&dy, NPIX * COUT * 2) ||
      hipMalloc((void**)&db, 64 * 4)) { printf("hipMalloc failed\n"); return 2; }
  hipMemcpy(dx, hx, NPIX * CIN * 2, hipMemcpyHostToDevice);
  hipMemcpy(dw, hw, COUT * CIN * 2, hipMemcpyHostToDevice);
  hipMemcpy(db, hb, COUT * 4, hipMemcpyHostToDevice);
  hipMemset(dpk, 0, pk);
  /* nn.Linear-style weight [Cout][Cin] -> MFMA fragment order */
  int rc = cvvae_pack_weights(CVVAE_BF16, dw, COUT, CIN, 1, CIN, 1, 0, CIN, cvvae_conv_kchunk(1, 1, 1), dpk, NULL);
  if (rc) { printf("cvvae_pack_weights -> %d\n", rc); return 1; }
  cvvae_conv_desc d; memset(&d, 0, sizeof(d));
  d.dtype = CVVAE_BF16; d.B = 1; d.Ti = 1; d.Hi = 1; d.Wi = NPIX; d.Cin = CIN; d.in_pix_stride = CIN;
  d.kT = d.kH = d.kW = 1; d.sT = d.sH = d.sW = 1; d.gn_rows_per_batch = 1;
  d.To = 1; d.Ho = 1; d.Wo = NPIX; d.Cout = COUT; d.out_mode = CVVAE_OUT_NDHWC; d.out_pix_stride = COUT; d.alpha = 1.0f;
  printf("kernel: %s\n", cvvae_conv_kernel_name(&d));
  rc = cvvae_conv_fwd(&d, dx, dpk, db, NULL, NULL, NULL, dy, NULL);
  if (rc) { printf("cvvae_conv_fwd -> %d\n", rc); return 1; }
  if (hipDeviceSynchronize()) { printf("launch failed\n"); return 1; }
  uint16_t* hy = malloc(NPIX * COUT * 2);
  hipMemcpy(hy, dy, NPIX * COUT * 2, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int p = 0; p < NPIX; ++p)
    for (int co = 0; co < COUT; ++co) {
      double acc = hb[co];
      for (int ci = 0; ci < CIN; ++ci) acc += (double)bf2f(hx[p * CIN + ci]) * (double)bf2f(hw[co * CIN + ci]);
      const double e = fabs(acc - (double)bf2f(hy[p * COUT + co]));
      if (e > maxerr) maxerr = e;
      if (fabs(acc) > maxref) maxref = fabs(acc);
    }
  printf("max |err| %.3e (max |ref| %.3f)\n", maxerr, maxref);
  /* error path: a bad descriptor is refused, nothing is launched */
  d.Cin = 100;
  if (cvvae_conv_fwd(&d, dx, dpk, db, NULL, NULL, NULL, dy, NULL) != CVVAE_EINVAL) { printf("bad descriptor accepted\n"); return 1; }
  if (maxerr > maxref * 0.00390625 + 1e-6) { printf("MISMATCH\n"); return 1; }  /* one bf16 output rounding */
  printf("C_ABI_OK\n");
  return 0;
}
