// fp6_probe.hip -- what gfx950's fp6 ("bf6" = e3m2) conversion and block-scaled MFMA instructions do, bit for bit (a measurement
// aid: the fast-fp32 correction MFMA of conv_kernel.h relies on exactly these facts).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/fp6_probe.hip -o /tmp/fp6_probe && /tmp/fp6_probe
// Part 1: v_cvt_scalef32_2xpk16_bf6_f32 / v_cvt_scalef32_pk32_bf6_f16 -- which input element lands in which 6-bit slot, the
//         direction of the scale, rounding, saturation, subnormals.
// Part 2: v_mfma_scale_f32_32x32x64_f8f6f4 with bf6 operands -- lane/slot -> (row, k) map and the per-lane E8M0 scale bytes,
//         checked against a host product.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void cvt_f32(const float* in, unsigned* out, float scale) {
  v16f a, b;
  for (int i = 0; i < 16; ++i) {
    a[i] = in[threadIdx.x * 32 + i];
    b[i] = in[threadIdx.x * 32 + 16 + i];
  }
  v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, scale);
  for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}
__global__ void cvt_f16(const float* in, unsigned* out, float scale) {
  v32h a;
  for (int i = 0; i < 32; ++i) a[i] = (_Float16)in[threadIdx.x * 32 + i];
  v6u r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(a, scale);
  for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}
__global__ void mfma6(const unsigned* A, const unsigned* B, const int* SA, const int* SB, float* out) {
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 6; ++i) {
    a[i] = (int)A[threadIdx.x * 6 + i];
    b[i] = (int)B[threadIdx.x * 6 + i];
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 3, 3, 0, SA[threadIdx.x], 0, SB[threadIdx.x]);
  for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = c[i];
}

static float dec(unsigned c) {
  const int s = (c >> 5) & 1, e = (c >> 2) & 7, m = c & 3;
  const float v = e == 0 ? m * 0.0625f : (1.0f + 0.25f * m) * std::ldexp(1.0f, e - 3);
  return s ? -v : v;
}
static unsigned slot(const unsigned* w6, int i) {  // 6-bit field i of a 192-bit little-endian string
  const int bit = 6 * i, d = bit >> 5, o = bit & 31;
  unsigned long long v = w6[d];
  if (d + 1 < 6) v |= (unsigned long long)w6[d + 1] << 32;
  return (unsigned)(v >> o) & 63u;
}

int main() {
  float* din;
  unsigned* dout;
  hipMalloc(&din, 64 * 32 * 4);
  hipMalloc(&dout, 64 * 6 * 4);
  std::vector<float> in(64 * 32, 0.f);
  std::vector<unsigned> out(64 * 6);
  // lane 0: element i = dec(i); lane 1: negative; lane 2: 4 * dec(i); lane 3: rounding / saturation cases
  const float cases[32] = {1.124f, 1.125f, 1.126f, 1.374f, 1.375f, 1.376f, 27.f, 28.f, 30.f, 1000.f, INFINITY, 0.03f,
                           0.031f, 0.0313f, 0.032f, 0.0625f, 0.09f, 0.0937f, 0.0938f, 0.1f, 0.2f, 0.22f, 0.24f, 0.25f,
                           -0.24f, -30.f, 1e-9f, -1e-9f, 0.28f, 0.29f, NAN, 0.f};
  for (int i = 0; i < 32; ++i) {
    in[0 * 32 + i] = dec(i);
    in[1 * 32 + i] = -dec(i);
    in[2 * 32 + i] = 4.f * dec(i);
    in[3 * 32 + i] = cases[i];
  }
  hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  for (int pass = 0; pass < 4; ++pass) {
    const float scale = pass < 2 ? 1.0f : 4.0f;
    const bool f16 = pass & 1;
    if (f16) hipLaunchKernelGGL(cvt_f16, dim3(1), dim3(64), 0, 0, din, dout, scale);
    else hipLaunchKernelGGL(cvt_f32, dim3(1), dim3(64), 0, 0, din, dout, scale);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    printf("== %s scale %.1f\n", f16 ? "v_cvt_scalef32_pk32_bf6_f16" : "v_cvt_scalef32_2xpk16_bf6_f32(a = in[0..15], b = in[16..31])", scale);
    for (int lane = 0; lane < 4; ++lane) {
      printf(" lane %d slots:", lane);
      for (int i = 0; i < 32; ++i) {
        const unsigned c = slot(&out[lane * 6], i);
        if (lane < 3) printf(" %u", c);
        else printf(" [%g->%g]", cases[i], dec(c));
      }
      printf("\n");
    }
  }
  // ---- Part 2
  std::vector<unsigned> A(64 * 6), B(64 * 6);
  std::vector<int> SA(64), SB(64);
  std::vector<unsigned> ca(64 * 32), cb(64 * 32);
  srand(7);
  for (int l = 0; l < 64; ++l) {
    for (int d = 0; d < 6; ++d) A[l * 6 + d] = B[l * 6 + d] = 0;
    for (int i = 0; i < 32; ++i) {
      ca[l * 32 + i] = rand() & 63;
      cb[l * 32 + i] = rand() & 63;
      for (int which = 0; which < 2; ++which) {
        unsigned* w = which ? &B[l * 6] : &A[l * 6];
        const unsigned long long c = which ? cb[l * 32 + i] : ca[l * 32 + i];
        const int bit = 6 * i, d = bit >> 5, o = bit & 31;
        w[d] |= (unsigned)(c << o);
        if (o > 26) w[d + 1] |= (unsigned)(c >> (32 - o));
      }
    }
    // scale VGPR: byte 0 = the value we believe is used; bytes 1-3 = decoys
    SA[l] = (127 + (rand() % 5) - 2) | (140 << 8) | (100 << 16) | (90 << 24);
    SB[l] = (127 + (rand() % 5) - 2) | (141 << 8) | (101 << 16) | (91 << 24);
  }
  unsigned *dA, *dB;
  int *dSA, *dSB;
  float* dC;
  hipMalloc(&dA, 64 * 6 * 4);
  hipMalloc(&dB, 64 * 6 * 4);
  hipMalloc(&dSA, 64 * 4);
  hipMalloc(&dSB, 64 * 4);
  hipMalloc(&dC, 64 * 16 * 4);
  hipMemcpy(dA, A.data(), 64 * 6 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), 64 * 6 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dSA, SA.data(), 64 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dSB, SB.data(), 64 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma6, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
  std::vector<float> C(64 * 16);
  hipMemcpy(C.data(), dC, 64 * 16 * 4, hipMemcpyDeviceToHost);
  // hypothesis: lane l holds row (l & 31) of A (and column (l & 31) of B), k = 32 * (l >> 5) + slot; scale = 2^(byte0 - 127) of
  // the lane; accumulator register i of lane l = C[row 8 * (i / 4) + 4 * (l >> 5) + i % 4][col l & 31]  (rows from A, cols from B)
  double worst = 0, worst_noscale = 0, mag = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 16; ++i) {
      const int row = 8 * (i / 4) + 4 * (l >> 5) + (i % 4), col = l & 31;
      double s = 0, s0 = 0;
      for (int h = 0; h < 2; ++h)
        for (int k = 0; k < 32; ++k) {
          const int la = row + 32 * h, lb = col + 32 * h;
          const double p = (double)dec(ca[la * 32 + k]) * dec(cb[lb * 32 + k]);
          s += p * std::ldexp(1.0, ((SA[la] & 255) - 127) + ((SB[lb] & 255) - 127));
          s0 += p;
        }
      worst = fmax(worst, fabs(s - C[l * 16 + i]));
      worst_noscale = fmax(worst_noscale, fabs(s0 - C[l * 16 + i]));
      mag = fmax(mag, fabs(s));
    }
  printf("== v_mfma_scale_f32_32x32x64_f8f6f4 cbsz:3 blgp:3 (bf6 x bf6), per-lane scale bytes (byte 0, op_sel 0)\n");
  printf(" max |C - host| under the hypothesis: %.6g   (without scales: %.6g; max |C| %.6g)\n", worst, worst_noscale, mag);
  printf(" C[lane 0][0..3] = %g %g %g %g\n", C[0], C[1], C[2], C[3]);
  return 0;
}
