// misc_kernels.hip -- the HBM-bound kernels around the conv: weight packing, GroupNorm statistics, LayerNorm,
// row softmax, transpose, temporal attention, NCDHW<->NDHWC, tile blending.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cvvae.h"
#include "conv_kernel.h"

namespace cvvae {

// ---------------------------------------------------------------------------------------------------------
// weight packing: dst[(((nb*nchunks + chunk)*taps + tap)*KSUB + ks)*512 + lane*8 + j]
//   = src(co = nb*32 + sigma(lane&31), ci = chunk*CK + ks*16 + (lane>>5)*8 + j, tap)
// ---------------------------------------------------------------------------------------------------------
// MFMA row i of a 32-output-channel block carries output channel sigma(i) = i with bits 2 and 3 swapped: the accumulator
// quads 2p, 2p+1 of a lane are then 8 consecutive channels (conv_kernel.h, store tail) -- the order is private to the
// packed format (cvvae_pack_weights* write it, conv_fwd_kernel reads it).
__device__ __forceinline__ int sigma_row(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <typename T>
__global__ void pack_weights_kernel(const T* __restrict__ src, int Cout_src, int Cin_src, int taps, long long s_co,
                                    long long s_ci, long long s_tap, int nchunks, int ksub, T* __restrict__ dst,
                                    long long nfrag_lanes, int fold_n, long long s_fold, long long s_batch,
                                    long long d_batch, int dst_taps, int dst_tap0) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nfrag_lanes) return;
  src += (long long)blockIdx.y * s_batch;  // batch item (grid.y)
  dst += (long long)blockIdx.y * d_batch;
  const int lane = (int)(gid & 63);
  long long f = gid >> 6;
  const int ks = (int)(f % ksub);
  f /= ksub;
  const int tap = (int)(f % taps);
  f /= taps;
  const int chunk = (int)(f % nchunks);
  const int nb = (int)(f / nchunks);
  const int co = nb * 32 + sigma_row(lane & 31);
  const int ci0 = chunk * (16 * ksub) + ks * 16 + (lane >> 5) * 8;
  typename Tr<T>::v8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = ci0 + j;
    T x = (T)0.f;
    if (co < Cout_src && ci < Cin_src) {
      const T* e = src + (long long)co * s_co + (long long)ci * s_ci + (long long)tap * s_tap;
      if (fold_n <= 1) {
        x = e[0];
      } else {  // taps that read the same input element are summed (fp32) and rounded once
        float a = 0.f;
        for (int f = 0; f < fold_n; ++f) a += (float)e[(long long)f * s_fold];
        x = (T)a;
      }
    }
    v[j] = x;
  }
  // dst_taps / dst_tap0: this launch fills taps [dst_tap0, dst_tap0 + taps) of a layout with dst_taps taps per k16 record
  // group (the time-fold slots of cvvae_pack_weights_tfolds); the plain packers pass dst_taps = taps, dst_tap0 = 0
  const long long rec = ((long long)nb * nchunks + chunk) * dst_taps + dst_tap0 + tap;
  *reinterpret_cast<typename Tr<T>::v8*>(dst + (rec * ksub + ks) * 512 + lane * 8) = v;
}

// Split-precision records (dtype CVVAE_F32: fp32 source, conv_fwd_kernel<..., XP>): THREE fp16 records per (k16, tap),
//   part 0: Wlo(c)  at k = c            (c = 0..15 of the sub-chunk)      x  B = hi(c)            -> Wlo.hi
//   part 1: Whi(c)  at k = c and c + 8  (c = 0..7)                        x  B = [hi(c) | lo(c)]  -> Whi.hi + Whi.lo
//   part 2: Whi(c)  at k = c-8 and c    (c = 8..15)                       x  B = [hi(c) | lo(c)]
// with Whi = fp16(w), Wlo = fp16(w - Whi); w = the (folded, fp32) source element.  The host pre-scales the weights by a power
// of two (ops.py) so that Wlo stays in fp16's normal range; the conv's alpha undoes it.
__device__ __forceinline__ void xp_store(_Float16* dst, long long rec3, int lane, const float (&w)[16]) {
  // w[0..15]: the 16 channels of the sub-chunk for my output row.  lane >> 5 selects k half of the MFMA A operand.
  const int kh = lane >> 5;
  f16x8 p0, p1, p2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a0 = w[kh * 8 + j];
    p0[j] = (_Float16)(a0 - (float)(_Float16)a0);  // Wlo of channel kh*8 + j
    p1[j] = (_Float16)w[j];                        // Whi of channel j      (both k halves)
    p2[j] = (_Float16)w[8 + j];                    // Whi of channel 8 + j  (both k halves)
  }
  *reinterpret_cast<f16x8*>(dst + (rec3 + 0) * 512 + lane * 8) = p0;
  *reinterpret_cast<f16x8*>(dst + (rec3 + 1) * 512 + lane * 8) = p1;
  *reinterpret_cast<f16x8*>(dst + (rec3 + 2) * 512 + lane * 8) = p2;
}

// Fast-fp32 records (dtype CVVAE_F32Q, conv_fwd_kernel<..., XP = 2>): the same three 1-KiB records per (k16, tap), holding
//   [0]: Whi(c) at k = c (c = 0..15)                                        x B = hi(c)  on the fp16 MFMA
//   [1], [2] (only at the FIRST tap of a pair; pairs = taps (0,1), (2,3), ... of every run of `run` taps): bytes 0-15 / 16-31 of
//            a lane's operand of the K = 64 bf8 MFMA: lanes 0-31 bf8(Whi(c)), lanes 32-63 bf8(Wlo(c)), c = 0..15, of tap a in
//            [1] and of tap b in [2] (zero when the run ends on tap a)     x B = [bf8(lo) | bf8(hi)] of taps a, b
__device__ __forceinline__ uint4 xq_half(int kh, const float (&w)[16]) {
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float hi = (float)(_Float16)w[j];
    v[j] = kh ? w[j] - hi : hi;
  }
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = v[j];
    b[j] = v[8 + j];
  }
  const uint2 lo = pack8_bf8(a), hi = pack8_bf8(b);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
// e3m2 ("bf6": sign, 3 exponent bits of bias 3, 2 mantissa bits; normals 0.25 .. 28, subnormals k/16), round to nearest even,
// saturating -- the rounding of v_cvt_scalef32_pk32_bf6_f16 (tools/probes/fp6_probe.hip)
__device__ __forceinline__ unsigned enc_e3m2(float v) {
  const unsigned sgn = (__float_as_uint(v) >> 31) << 5;
  const float a = fminf(fabsf(v), 28.0f);
  if (!(a >= 0.25f)) return sgn | (unsigned)(int)rintf(a * 16.0f);  // subnormals (and 0.25 itself from below: code 4)
  int e;
  const float fr = frexpf(a, &e);  // a = fr * 2^e, fr in [0.5, 1)  ->  a = (2 fr) * 2^(e-1)
  int m = (int)rintf((2.0f * fr - 1.0f) * 4.0f), ex = e - 1;
  if (m == 4) {
    m = 0;
    ++ex;
  }
  return sgn | (unsigned)((ex + 3) << 2) | (unsigned)m;
}
// Fast-fp32 records with fp6 corrections (dtype CVVAE_F32Q6, conv_fwd_kernel<..., XP = 3>): [0] as above; [1], [2] = the 32 bytes of a
// lane's operand of the K = 64 bf6 MFMA: lanes 0-31 tap a, lanes 32-63 tap b of the pair (zero when the run ends on tap a), row =
// output channel; 32 six-bit codes, slot 16 i + 8 t + c = channel 8 i + c of term t:  t = 0: Whi (x the activation's lo * 2^11),
// t = 1: Wlo * 2^11 (x the activation's hi) -- both of the magnitude of the weight, so that ONE power of two per lane (2^sh, the
// largest with max |value| 2^sh <= 28) puts them into e3m2's range; bytes 0-23 = the codes, byte 24 = the lane's E8M0 block scale
// 127 - sh - 11 (it also undoes the 2^11 of either term).
__device__ __forceinline__ void xq6_lane(const float (&w)[16], uint4& q1, uint4& q2) {
  float v[32];
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float hi = (float)(_Float16)w[j];
    const int sl = 16 * (j >> 3) + (j & 7);
    v[sl] = hi;
    v[sl + 8] = (w[j] - hi) * 2048.0f;
    mx = fmaxf(mx, fmaxf(fabsf(v[sl]), fabsf(v[sl + 8])));
  }
  int sh = 0;
  if (mx > 0.f) {
    int e;
    (void)frexpf(28.0f / mx, &e);
    sh = e - 1;  // floor(log2(28 / mx))
    sh = sh < -100 ? -100 : (sh > 100 ? 100 : sh);
  }
  const float mul = ldexpf(1.0f, sh);
  unsigned d[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const unsigned long long c = enc_e3m2(v[i] * mul);
    const int bit = 6 * i, dw = bit >> 5, o = bit & 31;
    d[dw] |= (unsigned)(c << o);
    if (o > 26) d[dw + 1] |= (unsigned)(c >> (32 - o));
  }
  q1 = make_uint4(d[0], d[1], d[2], d[3]);
  q2 = make_uint4(d[4], d[5], (unsigned)(127 - sh - 11), 0u);
}
__device__ __forceinline__ void xq_store(_Float16* dst, long long rec3, int lane, const float (&wa)[16], const float (&wb)[16],
                                         bool pair_start, bool has_b, bool q6) {
  const int kh = lane >> 5;
  f16x8 p0;
#pragma unroll
  for (int j = 0; j < 8; ++j) p0[j] = (_Float16)wa[kh * 8 + j];
  uint4 q1 = make_uint4(0, 0, 0, 0), q2 = make_uint4(0, 0, 0, 0);
  if (pair_start && q6) {
    if (kh == 0) xq6_lane(wa, q1, q2);
    else if (has_b) xq6_lane(wb, q1, q2);
    else q2 = make_uint4(0u, 0u, 127u, 0u);
  } else if (pair_start) {
    q1 = xq_half(kh, wa);
    if (has_b) q2 = xq_half(kh, wb);
  }
  *reinterpret_cast<f16x8*>(dst + (rec3 + 0) * 512 + lane * 8) = p0;
  *reinterpret_cast<uint4*>(dst + (rec3 + 1) * 512 + lane * 8) = q1;
  *reinterpret_cast<uint4*>(dst + (rec3 + 2) * 512 + lane * 8) = q2;
}

// qrun: 0 = split-precision (three fp16 MFMAs) records; > 0 = fast-fp32 records with tap pairs inside runs of qrun taps
__global__ void pack_weights_xp_kernel(const float* __restrict__ src, int Cout_src, int Cin_src, int taps, long long s_co,
                                       long long s_ci, long long s_tap, int nchunks, _Float16* __restrict__ dst,
                                       long long nfrag_lanes, int fold_n, long long s_fold, long long s_batch,
                                       long long d_batch, int dst_taps, int dst_tap0, int qrun, int q6) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nfrag_lanes) return;
  src += (long long)blockIdx.y * s_batch;
  dst += (long long)blockIdx.y * d_batch;
  const int lane = (int)(gid & 63);
  long long f = gid >> 6;
  const int tap = (int)(f % taps);
  f /= taps;
  const int chunk = (int)(f % nchunks);  // k16 index
  const int nb = (int)(f / nchunks);
  const int co = nb * 32 + sigma_row(lane & 31);
  auto load16 = [&](int tp, float (&w)[16]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ci = chunk * 16 + j;
      float a = 0.f;
      if (co < Cout_src && ci < Cin_src) {
        const float* e = src + (long long)co * s_co + (long long)ci * s_ci + (long long)tp * s_tap;
        for (int q = 0; q < (fold_n <= 1 ? 1 : fold_n); ++q) a += e[(long long)q * s_fold];
      }
      w[j] = a;
    }
  };
  float w[16];
  load16(tap, w);
  const long long rec = ((long long)nb * nchunks + chunk) * dst_taps + dst_tap0 + tap;
  if (qrun > 0) {
    const int tr = tap % qrun;
    const bool start = (tr & 1) == 0, has_b = start && tr + 1 < qrun;
    float wb[16];
    if (has_b) load16(tap + 1, wb);
    else {
#pragma unroll
      for (int j = 0; j < 16; ++j) wb[j] = 0.f;
    }
    xq_store(dst, rec * 3, lane, w, wb, start, has_b, q6 != 0);
  } else {
    xp_store(dst, rec * 3, lane, w);
  }
}

// Nearest-2x upsample folded into the conv weights (Upsample3D: F.interpolate(scale (1,2,2)) then a 3x3x3 conv,
// models/vae_blocks3d_sd3.py:342-356, models/vae_models.py:218-229).  Output row 2y+py reads upsampled rows 2y+py-1..2y+py+1
// = stored rows {y-1, y, y} (py = 0) or {y, y, y+1} (py = 1): two stored rows per phase, with the weights of the taps that
// coincide summed.  Same along x.  So phase (py, px) is a 3x2x2 convolution over the stored input with
//   a = 0: ky in {0}      (py = 0) | {0, 1} (py = 1);      a = 1: ky in {1, 2} (py = 0) | {2} (py = 1)
// (likewise b / kx / px), summed in fp32 and rounded once to the storage dtype.  Packed layout per phase as above with
// taps = 12, tap = (kt*2 + a)*2 + b.
template <typename T>
__global__ void pack_upfold_kernel(const T* __restrict__ src, int Cout, int Cin, int nchunks, T* __restrict__ dst,
                                   long long per_phase_lanes, long long phase_stride_elems, int tfold, int dst_taps,
                                   int dst_tap0) {
  const long long gid0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid0 >= 4 * per_phase_lanes) return;
  const int phase = (int)(gid0 / per_phase_lanes);
  const long long gid = gid0 - (long long)phase * per_phase_lanes;
  const int py = phase >> 1, px = phase & 1;
  const int lane = (int)(gid & 63);
  long long f = gid >> 6;
  // tfold: 0 = the three time taps kept (12 taps per phase); 1 = all three summed, 2 = centre tap only (single-frame inputs);
  // 3 = taps {0,1} summed, 4 = taps {1,2} summed (the boundary-frame slots of cvvae_pack_weights_upfold_tfolds): 4 taps each
  const int ntap = tfold ? 4 : 12;
  const int tap = (int)(f % ntap);
  f /= ntap;
  const int chunk = (int)(f % nchunks);
  const int nb = (int)(f / nchunks);
  const int kt = tap >> 2, a = (tap >> 1) & 1, b = tap & 1;
  const int kt_lo = tfold == 0 ? kt : (tfold == 1 || tfold == 3 ? 0 : 1), kt_hi = tfold == 0 ? kt : (tfold == 2 || tfold == 3 ? 1 : 2);
  // folded tap sets [lo, hi] along y and x
  const int y_lo = a == 0 ? 0 : (py == 0 ? 1 : 2), y_hi = a == 0 ? (py == 0 ? 0 : 1) : 2;
  const int x_lo = b == 0 ? 0 : (px == 0 ? 1 : 2), x_hi = b == 0 ? (px == 0 ? 0 : 1) : 2;
  const int co = nb * 32 + sigma_row(lane & 31);
  const int ci0 = chunk * 16 + (lane >> 5) * 8;
  typename Tr<T>::v8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = ci0 + j;
    float acc = 0.f;
    if (co < Cout && ci < Cin) {
      const T* w = src + ((long long)co * Cin + ci) * 27;
      for (int k = kt_lo; k <= kt_hi; ++k)
        for (int ky = y_lo; ky <= y_hi; ++ky)
          for (int kx = x_lo; kx <= x_hi; ++kx) acc += (float)w[k * 9 + ky * 3 + kx];
    }
    v[j] = (T)acc;
  }
  const long long rec = ((long long)nb * nchunks + chunk) * dst_taps + dst_tap0 + tap;
  *reinterpret_cast<typename Tr<T>::v8*>(dst + (long long)phase * phase_stride_elems + rec * 512 + lane * 8) = v;
}

__global__ void pack_upfold_xp_kernel(const float* __restrict__ src, int Cout, int Cin, int nchunks, _Float16* __restrict__ dst,
                                      long long per_phase_lanes, long long phase_stride_elems, int tfold, int dst_taps,
                                      int dst_tap0, int qrun, int q6) {
  const long long gid0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid0 >= 4 * per_phase_lanes) return;
  const int phase = (int)(gid0 / per_phase_lanes);
  const long long gid = gid0 - (long long)phase * per_phase_lanes;
  const int py = phase >> 1, px = phase & 1;
  const int lane = (int)(gid & 63);
  long long f = gid >> 6;
  const int ntap = tfold ? 4 : 12;
  const int tap = (int)(f % ntap);
  f /= ntap;
  const int chunk = (int)(f % nchunks);
  const int nb = (int)(f / nchunks);
  const int co = nb * 32 + sigma_row(lane & 31);
  auto load16 = [&](int tp, float (&w16)[16]) {
    const int kt = tp >> 2, a = (tp >> 1) & 1, b = tp & 1;
    const int kt_lo = tfold == 0 ? kt : (tfold == 1 || tfold == 3 ? 0 : 1), kt_hi = tfold == 0 ? kt : (tfold == 2 || tfold == 3 ? 1 : 2);
    const int y_lo = a == 0 ? 0 : (py == 0 ? 1 : 2), y_hi = a == 0 ? (py == 0 ? 0 : 1) : 2;
    const int x_lo = b == 0 ? 0 : (px == 0 ? 1 : 2), x_hi = b == 0 ? (px == 0 ? 0 : 1) : 2;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ci = chunk * 16 + j;
      float acc = 0.f;
      if (co < Cout && ci < Cin) {
        const float* w = src + ((long long)co * Cin + ci) * 27;
        for (int k = kt_lo; k <= kt_hi; ++k)
          for (int ky = y_lo; ky <= y_hi; ++ky)
            for (int kx = x_lo; kx <= x_hi; ++kx) acc += w[k * 9 + ky * 3 + kx];
      }
      w16[j] = acc;
    }
  };
  float w16[16];
  load16(tap, w16);
  const long long rec = ((long long)nb * nchunks + chunk) * dst_taps + dst_tap0 + tap;
  if (qrun > 0) {  // (runs of 4 taps = the 2x2 spatial taps of one time tap: pairs (0,1), (2,3))
    const bool start = (tap & 1) == 0;
    float wb[16];
    if (start) load16(tap + 1, wb);
    else {
#pragma unroll
      for (int j = 0; j < 16; ++j) wb[j] = 0.f;
    }
    xq_store(dst + (long long)phase * phase_stride_elems, rec * 3, lane, w16, wb, start, start, q6 != 0);
  } else {
    xp_store(dst + (long long)phase * phase_stride_elems, rec * 3, lane, w16);
  }
}

// ---------------------------------------------------------------------------------------------------------
// GroupNorm statistics.  Stage 1: grid (nsplit, rows); each block reduces a slab of pixels for all channels
// with Welford/Chan updates on 4-channel quads; stage 2 merges the slabs and emits the affine table.
// ---------------------------------------------------------------------------------------------------------
struct WStat {
  float n, mean, m2;
};
__device__ __forceinline__ void chan_merge(WStat& a, const WStat& b) {
  if (b.n == 0.f) return;
  const float n = a.n + b.n;
  const float d = b.mean - a.mean;
  const float f = b.n / n;
  a.mean += d * f;
  a.m2 += b.m2 + d * d * a.n * f;
  a.n = n;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, long long S, int C, long long ps,
                                                         int G, int nsplit, float* __restrict__ ws) {
  const int split = blockIdx.x, row = blockIdx.y;
  const int cv = C >> 3;         // 16-byte channel vectors per pixel
  const int ppp = 256 / cv;      // pixels per pass
  const int tid = threadIdx.x;
  const int myv = tid % cv, mypl = tid / cv;
  const long long per = (S + nsplit - 1) / nsplit;
  const long long p0 = (long long)split * per;
  long long p1 = p0 + per;
  if (p1 > S) p1 = S;
  WStat st[2] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  if (tid < cv * ppp) {
    const T* base = x + ((long long)row * S) * ps + myv * 8;
    long long px = p0 + mypl;
    // main loop: 8 independent 16-byte loads in flight per thread, then an exact two-pass (sum, then squared
    // deviations) over the 32 values of each channel quad held in registers, and ONE Chan merge per batch
    constexpr int U = 8;
    for (; px + (long long)(U - 1) * ppp < p1; px += (long long)U * ppp) {
      Raw8<T> u[U];
#pragma unroll
      for (int i = 0; i < U; ++i) u[i] = ldraw8<T>(base + (px + (long long)i * ppp) * ps);
      float f[U][8];
#pragma unroll
      for (int i = 0; i < U; ++i) unraw8<T>(u[i], f[i]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < U; ++i) sum += (f[i][h * 4 + 0] + f[i][h * 4 + 1]) + (f[i][h * 4 + 2] + f[i][h * 4 + 3]);
        const float mb = sum * (1.0f / (4 * U));
        float m2b = 0.f;
#pragma unroll
        for (int i = 0; i < U; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = f[i][h * 4 + j] - mb;
            m2b += d * d;
          }
        WStat q = {(float)(4 * U), mb, m2b};
        chan_merge(st[h], q);
      }
    }
    for (; px < p1; px += ppp) {
      float f[8];
      ld8<T>(base + px * ps, f);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float a = f[h * 4 + 0], b = f[h * 4 + 1], c = f[h * 4 + 2], d = f[h * 4 + 3];
        const float mb = (a + b + c + d) * 0.25f;
        const float m2b = (a - mb) * (a - mb) + (b - mb) * (b - mb) + (c - mb) * (c - mb) + (d - mb) * (d - mb);
        WStat q = {4.f, mb, m2b};
        chan_merge(st[h], q);
      }
    }
  }
  __shared__ WStat sh[256 * 2];
  sh[tid * 2 + 0] = st[0];
  sh[tid * 2 + 1] = st[1];
  __syncthreads();
  if (tid < G) {
    const int cpg = C / G;        // channels per group (multiple of 4)
    const int qpg = cpg >> 2;     // quads per group
    WStat acc = {0.f, 0.f, 0.f};
    const int q0 = tid * qpg;     // first quad index (over channels) of this group
    for (int pl = 0; pl < ppp; ++pl)
      for (int q = q0; q < q0 + qpg; ++q) {
        const int v = q >> 1, h = q & 1;
        chan_merge(acc, sh[(pl * cv + v) * 2 + h]);
      }
    float* o = ws + (((long long)row * nsplit + split) * G + tid) * 3;
    o[0] = acc.n;
    o[1] = acc.mean;
    o[2] = acc.m2;
  }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ ws, int nsplit, int G, int C, float eps,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ scale, float* __restrict__ shift) {
  const int row = blockIdx.x, tid = threadIdx.x;
  __shared__ float s_mean[64], s_rstd[64];
  // 8 lanes per group (G <= 32)
  const int g = tid >> 3, l8 = tid & 7;
  WStat acc = {0.f, 0.f, 0.f};
  if (g < G) {
    for (int s = l8; s < nsplit; s += 8) {
      const float* o = ws + (((long long)row * nsplit + s) * G + g) * 3;
      WStat q = {o[0], o[1], o[2]};
      chan_merge(acc, q);
    }
  }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    WStat q;
    q.n = __shfl_xor(acc.n, off);
    q.mean = __shfl_xor(acc.mean, off);
    q.m2 = __shfl_xor(acc.m2, off);
    chan_merge(acc, q);
  }
  if (g < G && l8 == 0) {
    s_mean[g] = acc.mean;
    s_rstd[g] = rsqrtf(acc.m2 / acc.n + eps);
  }
  __syncthreads();
  const int cpg = C / G;
  for (int c = tid; c < C; c += 256) {
    const int gg = c / cpg;
    const float sc = gamma[c] * s_rstd[gg];
    scale[(long long)row * C + c] = sc;
    shift[(long long)row * C + c] = beta[c] - s_mean[gg] * sc;
  }
}

// Finalize for the statistics fused into the conv epilogue: grid (G, rows); the block merges the `slabs` records of
// its (row, group) -- ws[(row*G + g)*slabs + s] = (n, mean, M2), contiguous -- in a fixed order (thread-strided, then a Chan tree)
// and writes the affine table of the group's channels.
// frames > 1: per-FRAME statistics (the per-frame GroupNorm of the attention blocks) from the records of a per-frame producer
// (kT = 1 convs: one-frame tiles in frame-major order): table row = sample * frames + frame, merging that frame's slabs / frames
// consecutive records.
__global__ __launch_bounds__(256) void gn_finalize_slabs_kernel(const float* __restrict__ ws, long long slabs_total, int G, int C,
                                                                float eps, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ scale,
                                                                float* __restrict__ shift, int frames) {
  const int g = blockIdx.x, tid = threadIdx.x;
  const int orow = blockIdx.y;                      // output table row
  const int row = orow / frames, fr = orow - row * frames;
  const long long slabs = slabs_total / frames;
  const float* base = ws + (((long long)row * G + g) * slabs_total + (long long)fr * slabs) * 3;  // my (row, group, frame) records
  WStat acc = {0.f, 0.f, 0.f};
  // keep 8 independent loads in flight per thread and merge them in index order afterwards (the merge order, hence the
  // result, does not depend on the batching); a wave's loads cover 768 contiguous bytes
  constexpr int U = 8;
  long long s = tid;
  for (; s + (long long)(U - 1) * 256 < slabs; s += (long long)U * 256) {
    WStat q[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const float* o = base + (s + (long long)i * 256) * 3;
      q[i].n = o[0];
      q[i].mean = o[1];
      q[i].m2 = o[2];
    }
#pragma unroll
    for (int i = 0; i < U; ++i) chan_merge(acc, q[i]);
  }
  for (; s < slabs; s += 256) {
    const float* o = base + s * 3;
    WStat q = {o[0], o[1], o[2]};
    chan_merge(acc, q);
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    WStat q;
    q.n = __shfl_xor(acc.n, off);
    q.mean = __shfl_xor(acc.mean, off);
    q.m2 = __shfl_xor(acc.m2, off);
    // both partners must end with the same value: merge in a fixed (lower lane first) order
    WStat lo = (tid & off) ? q : acc, hi = (tid & off) ? acc : q;
    chan_merge(lo, hi);
    acc = lo;
  }
  __shared__ WStat sh[4];
  if ((tid & 63) == 0) sh[tid >> 6] = acc;
  __syncthreads();
  WStat t = sh[0];
  chan_merge(t, sh[1]);
  chan_merge(t, sh[2]);
  chan_merge(t, sh[3]);
  const float mean = t.mean, rstd = rsqrtf(t.m2 / t.n + eps);
  const int cpg = C / G;
  for (int c = g * cpg + tid; c < (g + 1) * cpg; c += 256) {
    const float sc = gamma[c] * rstd;
    scale[(long long)orow * C + c] = sc;
    shift[(long long)orow * C + c] = beta[c] - mean * sc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// GroupNorm-apply (+ SiLU) pass: the conv prologue's arithmetic (gn_affine + silu_f of conv_kernel.h), once per element
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_silu_apply_kernel(const T* __restrict__ x, long long S, int C, long long ps,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            int silu, T* __restrict__ out, long long nvec) {
  const int cv = C >> 3;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < nvec; g += (long long)gridDim.x * 256) {
    const long long pix = g / cv;
    const int c0 = (int)(g - pix * cv) * 8;
    const long long row = pix / S;
    float f[8];
    ld8<T>(x + pix * ps + c0, f);
    const float4 a0 = *reinterpret_cast<const float4*>(scale + row * C + c0), a1 = *reinterpret_cast<const float4*>(scale + row * C + c0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(shift + row * C + c0), b1 = *reinterpret_cast<const float4*>(shift + row * C + c0 + 4);
    const float sc[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, sh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = gn_affine(f[j], sc[j], sh[j]);
      f[j] = silu ? silu_f(v) : v;
    }
    st8<T>(out + pix * C + c0, f);
  }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over C per pixel: one wave per pixel, C multiple of 8, C <= 64*8*4
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, long long P, int C, float eps,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        T* __restrict__ out) {
  const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= P) return;
  const int lane = threadIdx.x & 63;
  const int nv = C >> 3;
  float sum = 0.f;
  for (int v = lane; v < nv; v += 64) {
    float f[8];
    ld8<T>(x + pix * C + v * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += f[j];
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) sum += __shfl_xor(sum, off);
  const float mean = sum / (float)C;
  float var = 0.f;
  for (int v = lane; v < nv; v += 64) {
    float f[8];
    ld8<T>(x + pix * C + v * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) var += (f[j] - mean) * (f[j] - mean);
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) var += __shfl_xor(var, off);
  const float rstd = rsqrtf(var / (float)C + eps);
  for (int v = lane; v < nv; v += 64) {
    float f[8];
    ld8<T>(x + pix * C + v * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (f[j] - mean) * rstd * gamma[v * 8 + j] + beta[v * 8 + j];
    st8<T>(out + pix * C + v * 8, f);
  }
}

// ---------------------------------------------------------------------------------------------------------
// row softmax: fp32 scores -> T probabilities; one block per row
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
#pragma unroll
  for (int off = 32; off; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int n_valid, long long ld_s,
                                                           T* __restrict__ p, long long ld_p) {
  __shared__ float sh[4];
  const float* sr = s + (long long)blockIdx.x * ld_s;
  T* pr = p + (long long)blockIdx.x * ld_p;
  float mx = -3.0e38f;
  for (int i = threadIdx.x; i < n_valid; i += 256) mx = fmaxf(mx, sr[i]);
  mx = block_reduce(mx, true, sh);
  float sum = 0.f;
  for (int i = threadIdx.x; i < n_valid; i += 256) sum += __expf(sr[i] - mx);
  sum = block_reduce(sum, false, sh);
  const float inv = 1.0f / sum;
  for (int i = threadIdx.x; i < (int)ld_p; i += 256) pr[i] = (T)(i < n_valid ? __expf(sr[i] - mx) * inv : 0.f);
}

// ---------------------------------------------------------------------------------------------------------
// Input gradient of GroupNorm (+ SiLU) -- the frozen-module backward of the training path (cvvae_amd/grad.py).
//   xh = (x - mean) * rstd = x * rs[c] + nm[c]      (rs = rstd, nm = -mean * rstd: cvvae_gn_finalize with gamma 1, beta 0)
//   a  = xh * gamma[c] + beta[c],  y = act(a)       (the forward value the next conv consumed)
//   gh = gy * act'(a) * gamma[c]                    (gradient w.r.t. xh)
//   gx = rs * (gh - mean_grp(gh) - xh * mean_grp(gh * xh))  [+ add]
// Pass 1 (gn_bwd_reduce_kernel) writes, per (row, pixel split, group), sum(gh) and sum(gh * xh); pass 2 (gn_bwd_apply_kernel) sums
// the splits in index order (deterministic) and applies.  Channel quads never straddle a group (C / G a multiple of 4).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_grad_f(float a) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-a));
  return sg * (1.0f + a * (1.0f - sg));
}

struct GnBwdTab {  // the per-channel tables of my 8 channels
  float rs[8], nm[8], ga[8], be[8];
};
__device__ __forceinline__ void gn_bwd_load_tab(GnBwdTab& t, const float* __restrict__ rs, const float* __restrict__ nm,
                                                const float* __restrict__ gamma, const float* __restrict__ beta, long long row,
                                                int C, int c0) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    t.rs[j] = rs[row * C + c0 + j];
    t.nm[j] = nm[row * C + c0 + j];
    t.ga[j] = gamma[c0 + j];
    t.be[j] = beta[c0 + j];
  }
}

// PARAMS: also the per-channel sums of the affine gradients  d beta = sum gy act'(a),  d gamma = sum gy act'(a) xh  (the kernel forms
// gy act'(a) anyway: training the norm costs no extra pass over x and gy), one [C][2] table per (row, split) in `wsp`
template <typename T, bool PARAMS = false>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ gy, long long S, int C,
                                                            int G, int nsplit, const float* __restrict__ rs,
                                                            const float* __restrict__ nm, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int silu, float* __restrict__ ws,
                                                            float* __restrict__ wsp = nullptr) {
  const int split = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
  const int cv = C >> 3, ppp = 256 / cv;
  const int myv = tid % cv, mypl = tid / cv;
  const long long per = (S + nsplit - 1) / nsplit;
  const long long p0 = (long long)split * per;
  long long p1 = p0 + per;
  if (p1 > S) p1 = S;
  GnBwdTab t;
  gn_bwd_load_tab(t, rs, nm, gamma, beta, row, C, myv * 8);
  float a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};
  float pb[PARAMS ? 8 : 1], pg[PARAMS ? 8 : 1];
#pragma unroll
  for (int j = 0; j < (PARAMS ? 8 : 1); ++j) pb[j] = pg[j] = 0.f;
  const T* xb = x + (long long)row * S * C + myv * 8;
  const T* gb = gy + (long long)row * S * C + myv * 8;
  // four pixels per trip: their 16-byte loads are issued together (one pixel per trip was latency bound: ~2 TB/s on the
  // million-pixel tensors of a training step); lanes past the end re-read pixel px with a zeroed gradient
  constexpr int U = 4;
  for (long long px = p0 + mypl; px < p1; px += U * ppp) {
    float f[U][8], g[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = px + u * ppp < p1;
      const long long q = ok ? px + u * ppp : px;
      ld8<T>(xb + q * C, f[u]);
      ld8<T>(gb + q * C, g[u]);
      if (!ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) g[u][j] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = __builtin_fmaf(f[u][j], t.rs[j], t.nm[j]);
        const float a = __builtin_fmaf(xh, t.ga[j], t.be[j]);
        const float gact = g[u][j] * (silu ? silu_grad_f(a) : 1.0f);
        const float gh = gact * t.ga[j];
        a1[j >> 2] += gh;
        a2[j >> 2] += gh * xh;
        if constexpr (PARAMS) {
          pb[j] += gact;
          pg[j] += gact * xh;
        }
      }
    }
  }
  if constexpr (PARAMS) {  // per channel: the pixel lanes of its vector, in index order
    __shared__ float shp[256][17];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      shp[tid][j] = pb[j];
      shp[tid][8 + j] = pg[j];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      float s1 = 0.f, s2 = 0.f;
      for (int pl = 0; pl < ppp; ++pl) {
        s1 += shp[pl * cv + (c >> 3)][c & 7];
        s2 += shp[pl * cv + (c >> 3)][8 + (c & 7)];
      }
      float* o = wsp + (((long long)row * nsplit + split) * C + c) * 2;
      o[0] = s1;
      o[1] = s2;
    }
  }
  __shared__ float sh1[256][2], sh2[256][2];
  sh1[tid][0] = a1[0]; sh1[tid][1] = a1[1];
  sh2[tid][0] = a2[0]; sh2[tid][1] = a2[1];
  __syncthreads();
  if (tid < G) {  // group tid: its channel quads q, every pixel lane, in index order
    const int qpg = (C / G) >> 2;
    float s1 = 0.f, s2 = 0.f;
    for (int q = tid * qpg; q < (tid + 1) * qpg; ++q)
      for (int pl = 0; pl < ppp; ++pl) {
        const int th = pl * cv + (q >> 1);
        s1 += sh1[th][q & 1];
        s2 += sh2[th][q & 1];
      }
    float* o = ws + (((long long)row * nsplit + split) * G + tid) * 2;
    o[0] = s1;
    o[1] = s2;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                           const T* __restrict__ add, T* __restrict__ out, long long S, int C, int G,
                                                           int nsplit, const float* __restrict__ rs, const float* __restrict__ nm,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                                           const float* __restrict__ ws) {
  const int row = blockIdx.y, tid = threadIdx.x;
  __shared__ float c1[64], c2[64], q1[256], q2[256];
  {  // group g = tid % G, split lane l = tid / G sums splits l, l + lanes, ...; the lanes are then added in index order
    const int lanes = 256 / G, g = tid % G, l = tid / G;
    float s1 = 0.f, s2 = 0.f;
    if (l < lanes)
      for (int sp = l; sp < nsplit; sp += lanes) {
        const float2 o = *reinterpret_cast<const float2*>(ws + (((long long)row * nsplit + sp) * G + g) * 2);
        s1 += o.x;
        s2 += o.y;
      }
    q1[tid] = s1;
    q2[tid] = s2;
    __syncthreads();
    if (tid < G) {
      s1 = 0.f; s2 = 0.f;
      for (int k = 0; k < lanes; ++k) {
        s1 += q1[k * G + tid];
        s2 += q2[k * G + tid];
      }
      const float invd = 1.0f / ((float)(C / G) * (float)S);
      c1[tid] = s1 * invd;
      c2[tid] = s2 * invd;
    }
  }
  __syncthreads();
  const int cv = C >> 3;
  const int myv = tid % cv;  // 256 % cv == 0 and the stride below is a multiple of 256: my channel vector is fixed
  GnBwdTab t;
  gn_bwd_load_tab(t, rs, nm, gamma, beta, row, C, myv * 8);
  const int cpg = C / G;
  const float m1a = c1[(myv * 8) / cpg], m2a = c2[(myv * 8) / cpg], m1b = c1[(myv * 8 + 4) / cpg], m2b = c2[(myv * 8 + 4) / cpg];
  const long long nvec = S * cv, base = (long long)row * S * C;
  for (long long v = (long long)blockIdx.x * 256 + tid; v < nvec; v += (long long)gridDim.x * 256) {
    const long long off = base + v * 8;  // v = pixel * cv + myv
    float f[8], g[8], ad[8];
    ld8<T>(x + off, f);
    ld8<T>(gy + off, g);
    if (add) ld8<T>(add + off, ad);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = __builtin_fmaf(f[j], t.rs[j], t.nm[j]);
      const float a = __builtin_fmaf(xh, t.ga[j], t.be[j]);
      const float gh = g[j] * (silu ? silu_grad_f(a) : 1.0f) * t.ga[j];
      const float r = t.rs[j] * (gh - (j < 4 ? m1a : m1b) - xh * (j < 4 ? m2a : m2b));
      f[j] = add ? r + ad[j] : r;
    }
    st8<T>(out + off, f);
  }
}

// gradient of the row softmax: gs = alpha * p * (gp - sum_j p_j gp_j); columns >= n_valid (up to ld_o) are written as 0
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const T* __restrict__ p, long long ld_p, const float* __restrict__ gp,
                                                               long long ld_g, int n_valid, float alpha, T* __restrict__ out,
                                                               long long ld_o) {
  __shared__ float sh[4];
  const T* pr = p + (long long)blockIdx.x * ld_p;
  const float* gr = gp + (long long)blockIdx.x * ld_g;
  T* orow = out + (long long)blockIdx.x * ld_o;
  float dot = 0.f;
  for (int i = threadIdx.x; i < n_valid; i += 256) dot += (float)pr[i] * gr[i];
  dot = block_reduce(dot, false, sh);
  for (int i = threadIdx.x; i < (int)ld_o; i += 256) orow[i] = (T)(i < n_valid ? alpha * (float)pr[i] * (gr[i] - dot) : 0.f);
}

// gradient of the nearest-neighbour x2 upsample (H and W): out[n][y][x][c] = sum of the 2x2 block of g [n][2H][2W][C]
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_sum_kernel(const T* __restrict__ g, long long N, int H, int W, int C,
                                                             T* __restrict__ out) {
  const int cv = C >> 3;
  const long long nvec = N * H * W * cv;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
    const int c0 = (int)(v % cv) * 8;
    long long pix = v / cv;
    const int xo = (int)(pix % W);
    pix /= W;
    const int yo = (int)(pix % H);
    const long long n = pix / H;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float f[8];
      ld8<T>(g + (((n * 2 * H + 2 * yo + (d >> 1)) * 2 * W) + 2 * xo + (d & 1)) * C + c0, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
    st8<T>(out + ((n * H + yo) * W + xo) * C + c0, acc);
  }
}

static inline int gn_bwd_splits(long long S) {  // >= 512 pixels per split (a 147 k-pixel tensor at 2048 had 72 workgroups for 256 CUs)
  // (up to 1024 splits: a 5-D GroupNorm has ONE row per sample, and 64 workgroups left three quarters of the CUs idle on the
  //  million-pixel tensors of a training step -- 372 us per call, profiles/r4_train_step_kernel_stats_v2.txt)
  const long long n = (S + 511) / 512;
  return (int)(n < 1 ? 1 : (n > 1024 ? 1024 : n));
}
// sum1[c], sum2[c] = sums over `nparts` [C][2] tables (a block = 32 channels x 8 part lanes, parts summed in a fixed order)
__global__ __launch_bounds__(256) void gn_params_final_kernel(const float* __restrict__ ws, int nparts, int C, float* __restrict__ o1,
                                                              float* __restrict__ o2) {
  __shared__ float sh[8][32][2];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a1 = 0.f, a2 = 0.f;
  if (c < C)
    for (int i = pl; i < nparts; i += 8) {
      const float2 v = *reinterpret_cast<const float2*>(ws + ((long long)i * C + c) * 2);
      a1 += v.x;
      a2 += v.y;
    }
  sh[pl][cl][0] = a1;
  sh[pl][cl][1] = a2;
  __syncthreads();
  if (pl == 0 && c < C) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s1 += sh[k][cl][0];
      s2 += sh[k][cl][1];
    }
    o1[c] = s1;
    o2[c] = s2;
  }
}

template <typename T>
static void gn_bwd_launch(const void* x, const void* gy, const void* add, int rows, long long S, int C, int G, const float* rs,
                          const float* nm, const float* gamma, const float* beta, int silu, void* gx, float* ws, hipStream_t s,
                          float* dgamma = nullptr, float* dbeta = nullptr) {
  const int nsplit = gn_bwd_splits(S);
  if (dgamma) {  // training the norm: the affine sums ride on the reduction pass (tables behind the group sums in the workspace)
    float* wsp = ws + (long long)rows * nsplit * G * 2;
    hipLaunchKernelGGL((gn_bwd_reduce_kernel<T, true>), dim3(nsplit, rows), dim3(256), 0, s, (const T*)x, (const T*)gy, S, C, G, nsplit,
                       rs, nm, gamma, beta, silu, ws, wsp);
    hipLaunchKernelGGL(gn_params_final_kernel, dim3((C + 31) / 32), dim3(256), 0, s, (const float*)wsp, rows * nsplit, C, dbeta, dgamma);
  } else
  hipLaunchKernelGGL(gn_bwd_reduce_kernel<T>, dim3(nsplit, rows), dim3(256), 0, s, (const T*)x, (const T*)gy, S, C, G, nsplit, rs, nm,
                     gamma, beta, silu, ws);
  long long blocks = (S * (C / 8) + 255) / 256;
  if (blocks > 2048) blocks = 2048;  // grid-stride beyond 8 blocks per CU and row
  hipLaunchKernelGGL(gn_bwd_apply_kernel<T>, dim3((unsigned)blocks, rows), dim3(256), 0, s, (const T*)x, (const T*)gy, (const T*)add,
                     (T*)gx, S, C, G, nsplit, rs, nm, gamma, beta, silu, ws);
}

// ---------------------------------------------------------------------------------------------------------
// transpose of 2-byte elements, 32x32 LDS tiles
// ---------------------------------------------------------------------------------------------------------
template <typename E>  // E = uint16_t (fp16 / bf16 elements) or uint32_t (float)
__global__ __launch_bounds__(256) void transpose16_kernel(const E* __restrict__ in, int R, int C, long long ld_in,
                                                          long long bs_in, E* __restrict__ out, long long ld_out,
                                                          long long bs_out) {
  __shared__ E t[32][33];
  const E* ib = in + (long long)blockIdx.z * bs_in;
  E* ob = out + (long long)blockIdx.z * bs_out;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + k * 8, c = c0 + tx;
    t[ty + k * 8][tx] = (r < R && c < C) ? ib[(long long)r * ld_in + c] : (E)0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + k * 8, r = r0 + tx;
    if (c < C && r < R) ob[(long long)c * ld_out + r] = t[tx][ty + k * 8];
  }
}

// ---------------------------------------------------------------------------------------------------------
// temporal attention over T <= 8 frames per pixel, directly on NDHWC: token (b, t, s) lives at
// ((b*T + t)*S + s)*C.  One wave per pixel (b, s).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                            const T* __restrict__ v, long long P, int Tn, long long S,
                                                            int C, float scale, T* __restrict__ out) {
  const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // over B*S
  if (pix >= P) return;
  const int lane = threadIdx.x & 63;
  const int nv = C >> 3;  // 16-byte vectors per token
  const long long b = pix / S, s = pix - b * S;
  const long long base = (b * Tn * S + s) * C;  // token 0 of this pixel
  const long long tstride = S * C;
  const T* qb = q + base;
  const T* kb = k + base;
  const T* vb = v + base;
  float sc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[i][j] = 0.f;
  for (int vv = lane; vv < nv; vv += 64) {
    float qf[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < Tn) ld8<T>(qb + i * tstride + vv * 8, qf[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < Tn) {
        float kf[8];
        ld8<T>(kb + j * tstride + vv * 8, kf);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < Tn) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += qf[i][e] * kf[e];
            sc[i][j] += d;
          }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = sc[i][j];
#pragma unroll
      for (int off = 32; off; off >>= 1) x += __shfl_xor(x, off);
      sc[i][j] = x * scale;
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < Tn) {
      float mx = -3.0e38f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < Tn) mx = fmaxf(mx, sc[i][j]);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < Tn) {
          sc[i][j] = __expf(sc[i][j] - mx);
          sum += sc[i][j];
        }
      const float inv = 1.0f / sum;
#pragma unroll
      for (int j = 0; j < 8; ++j) sc[i][j] = (j < Tn) ? sc[i][j] * inv : 0.f;
    }
  }
  for (int vv = lane; vv < nv; vv += 64) {
    float o[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) o[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < Tn) {
        float vf[8];
        ld8<T>(vb + j * tstride + vv * 8, vf);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < Tn) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[i][e] += sc[i][j] * vf[e];
          }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < Tn) st8<T>(out + base + i * tstride + vv * 8, o[i]);
  }
}

// General frame count (T' > 8: en_de_n_frames_a_time = None or > 28): one wave per (pixel, query frame), online softmax over
// the key frames, a lane owns up to 4 x 8 channels (C <= 2048).  Same arithmetic as above: fp32 scores / probabilities /
// accumulation, one rounding at the store.
template <typename T>
__global__ __launch_bounds__(256) void temporal_attn_general_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                    const T* __restrict__ v, long long P, int Tn, long long S,
                                                                    int C, float scale, T* __restrict__ out) {
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // over B*S*Tn
  if (wid >= P * Tn) return;
  const int lane = threadIdx.x & 63;
  const int nv = C >> 3;
  const long long pix = wid / Tn;
  const int i = (int)(wid - pix * Tn);
  const long long b = pix / S, s = pix - b * S;
  const long long base = (b * Tn * S + s) * C;
  const long long tstride = S * C;
  float qf[4][8], o[4][8];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int vv = lane + u * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) { qf[u][e] = 0.f; o[u][e] = 0.f; }
    if (vv < nv) ld8<T>(q + base + i * tstride + vv * 8, qf[u]);
  }
  float mx = -3.0e38f, sum = 0.f;
  for (int j = 0; j < Tn; ++j) {
    float d = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int vv = lane + u * 64;
      if (vv < nv) {
        float kf[8];
        ld8<T>(k + base + j * tstride + vv * 8, kf);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += qf[u][e] * kf[e];
      }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) d += __shfl_xor(d, off);
    d *= scale;
    const float nmx = fmaxf(mx, d);
    const float corr = __expf(mx - nmx), pj = __expf(d - nmx);
    sum = sum * corr + pj;
    mx = nmx;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int vv = lane + u * 64;
      if (vv < nv) {
        float vf[8];
        ld8<T>(v + base + j * tstride + vv * 8, vf);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[u][e] = o[u][e] * corr + pj * vf[e];
      }
    }
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int vv = lane + u * 64;
    if (vv < nv) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[u][e] *= inv;
      st8<T>(out + base + i * tstride + vv * 8, o[u]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// layout at the boundary
// ---------------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void ncdhw_to_ndhwc_kernel(const TS* __restrict__ in, int C, long long THW, int Cpad,
                                                             long long npix, TD* __restrict__ out) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;  // over B*T*H*W
  if (pix >= npix) return;
  const long long b = pix / THW, s = pix - b * THW;
  for (int c0 = 0; c0 < Cpad; c0 += 8) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      f[j] = (c < C) ? (float)in[(b * C + c) * THW + s] : 0.f;
    }
    st8<TD>(out + pix * Cpad + c0, f);
  }
}

// [B,C,T,H,W] -> row-packed [B,T,H,W+3,4] (include/cvvae.h, in_overlap): one thread per stored pixel (8 / 16 bytes)
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void ncdhw_to_rowpack_kernel(const TS* __restrict__ in, int C, long long THW, int W, int replicate,
                                                               long long npix_out, TD* __restrict__ out) {
  const long long op = (long long)blockIdx.x * 256 + threadIdx.x;  // over B*T*H*(W+3)
  if (op >= npix_out) return;
  const int Wp = W + 3;
  const long long row = op / Wp;            // over B*T*H
  const int xp = (int)(op - row * Wp);
  int xs = xp - 1;
  bool zero = xp == W + 2;
  if (xs < 0 || xs >= W) {
    if (replicate) xs = xs < 0 ? 0 : W - 1;
    else zero = true;
  }
  const long long TH = THW / W;             // rows per batch item
  const long long b = row / TH, r = row - b * TH;
  TD v[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = (TD)((c < C && !zero) ? (float)in[(b * C + c) * THW + r * W + xs] : 0.f);
#pragma unroll
  for (int c = 0; c < 4; ++c) out[op * 4 + c] = v[c];
}

template <typename T>
__global__ __launch_bounds__(256) void ndhwc_to_rowpack_kernel(const T* __restrict__ in, int C, long long ps, int W, int replicate,
                                                               long long npix_out, T* __restrict__ out) {
  const long long op = (long long)blockIdx.x * 256 + threadIdx.x;  // over B*T*H*(W+3)
  if (op >= npix_out) return;
  const int Wp = W + 3;
  const long long row = op / Wp;
  const int xp = (int)(op - row * Wp);
  int xs = xp - 1;
  bool zero = xp == W + 2;
  if (xs < 0 || xs >= W) {
    if (replicate) xs = xs < 0 ? 0 : W - 1;
    else zero = true;
  }
  const T* src = in + (row * W + xs) * ps;
#pragma unroll
  for (int c = 0; c < 4; ++c) out[op * 4 + c] = (c < C && !zero) ? src[c] : (T)0.f;
}

template <typename T>
__global__ __launch_bounds__(256) void ndhwc_to_ncdhw_kernel(const T* __restrict__ in, int C, long long THW, long long ps,
                                                             long long npix, T* __restrict__ out) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= npix) return;
  const long long b = pix / THW, s = pix - b * THW;
  for (int c = 0; c < C; ++c) out[(b * C + c) * THW + s] = in[pix * ps + c];
}

// ---------------------------------------------------------------------------------------------------------
// pixel pre/post-processing of the inference scripts, on the device (cvvae_inference_video.py:24-38, 47-50).
// The arithmetic is evaluated op by op in the storage dtype, as the script's half tensors do (fp32 op, one rounding each).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void frames_u8_to_ndhwc_kernel(const uint8_t* __restrict__ f, long long npix, int Cpad,
                                                                 T* __restrict__ out) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= npix) return;
  float v[8];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const T h = (T)(float)f[pix * 3 + c];        // .half()
    const T d = (T)((float)h / 127.5f);          // / 127.5
    v[c] = (float)(T)((float)d - 1.0f);          // - 1.0
  }
#pragma unroll
  for (int c = 3; c < 8; ++c) v[c] = 0.f;
  st8<T>(out + pix * Cpad, v);
  const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c0 = 8; c0 < Cpad; c0 += 8) st8<T>(out + pix * Cpad + c0, z);
}

template <typename T>
__global__ __launch_bounds__(256) void ncdhw_to_frames_u8_kernel(const T* __restrict__ in, long long thw, uint8_t* __restrict__ f) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= thw) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float x = (float)in[(long long)c * thw + pix];
    x = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);   // torch.clamp(results, -1.0, 1.0)
    const T a = (T)(x + 1.0f);                         // + 1.0
    const T m = (T)((float)a * 127.5f);                // * 127.5
    f[pix * 3 + c] = (uint8_t)(float)m;                // .to(torch.uint8): truncation
  }
}

// ---------------------------------------------------------------------------------------------------------
// one axis of the uint8 antialiased bilinear frame resize (include/cvvae.h cvvae_resize_u8_axis)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_u8_axis_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long long n_out,
                                                             int in_size, int out_size, long long inner,
                                                             const int* __restrict__ xmin, const int* __restrict__ xsize,
                                                             const int* __restrict__ w, int ksize, int precision) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // over outer * out_size * inner
  if (e >= n_out) return;
  const long long r = e % inner, q = e / inner;
  const int i = (int)(q % out_size);
  const long long o = q / out_size;
  const uint8_t* src = in + (o * in_size + xmin[i]) * inner + r;
  const int* wi = w + (long long)i * ksize;
  int acc = 1 << (precision - 1);
  const int n = xsize[i];
  for (int j = 0; j < n; ++j) acc += wi[j] * (int)src[(long long)j * inner];
  acc >>= precision;
  out[e] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
}

// ---------------------------------------------------------------------------------------------------------
// the last layer's spatial taps, gathered from the (3,1,1) conv's per-input-pixel columns (include/cvvae.h cvvae_conv_out_gather)
// ---------------------------------------------------------------------------------------------------------
template <typename T, int CO>
__global__ __launch_bounds__(256) void conv_out_gather_kernel(const float* __restrict__ V, int T_, int H, int W, long long ldv,
                                                              const float* __restrict__ bias, int replicate, long long npix,
                                                              T* __restrict__ out, uint8_t* __restrict__ u8) {
  // XCD-aware block order: consecutive blocks run on different XCDs (8, round-robin), each with its own L2; neighbouring patches
  // share their halo records, so each XCD gets a CONTIGUOUS band of patches
  const long long nblk = gridDim.x, q8 = nblk / 8, bid = blockIdx.x;
  const long long lb = bid < q8 * 8 ? (bid % 8) * q8 + bid / 8 : bid;
  // A block is an 8 x 32 pixel patch.  The 10 x 34 records around it are staged ONCE into LDS with coalesced 16-byte loads (every
  // thread reading its nine neighbours' 12 bytes straight from memory touched nine 128-byte lines per pixel: 3.8 GB fetched for
  // a 0.57 GB V, 0.56 ms); the nine-neighbour sums then come from LDS.  Pixel pitch in LDS: 9*CO + 2 floats (odd: no bank conflicts
  // between the lanes of a row).  Halo pixels outside the frame are staged from the clamped coordinate (replicate padding) or as
  // zeros (zero padding), so the sum below needs no bounds logic.
  constexpr int PW = 34, PH = 10, NV = 9 * CO, PITCH = NV + 2;
  __shared__ float sv[PH * PW * PITCH];
  const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
  const int tx = (int)(lb % tiles_x);
  const long long r0 = lb / tiles_x;
  const int ty = (int)(r0 % tiles_y);
  const long long bt = r0 / tiles_y;  // b*T + t
  const int x0 = tx * 32 - 1, y0 = ty * 8 - 1;
  // stage: one thread per (pixel, 16-byte quarter-record): NV = 27 floats = 6.75 quarters -> 7 loads per pixel
  constexpr int QP = (NV + 3) / 4;
  for (int i = threadIdx.x; i < PH * PW * QP; i += 256) {
    const int q = i % QP, pp = i / QP;
    const int px = pp % PW, py = pp / PW;
    int ys = y0 + py, xs = x0 + px;
    const bool outside = ys < 0 || ys >= H || xs < 0 || xs >= W;
    ys = ys < 0 ? 0 : (ys >= H ? H - 1 : ys);
    xs = xs < 0 ? 0 : (xs >= W ? W - 1 : xs);
    float4 v = *reinterpret_cast<const float4*>(V + ((bt * H + ys) * W + xs) * ldv + q * 4);  // (ldv % 4 == 0: 16-byte aligned)
    if (outside && !replicate) v = make_float4(0.f, 0.f, 0.f, 0.f);
    float* d = sv + pp * PITCH + q * 4;
    d[0] = v.x;
    if (q * 4 + 1 < NV) d[1] = v.y;
    if (q * 4 + 2 < NV) d[2] = v.z;
    if (q * 4 + 3 < NV) d[3] = v.w;
  }
  __syncthreads();
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  const int x = tx * 32 + lx, y = ty * 8 + ly;
  if (x >= W || y >= H) return;
  const long long pix = (bt * H + y) * W + x;
  (void)npix;
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = bias[c];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const float* v = sv + ((ly + dy) * PW + (lx + dx)) * PITCH + (dy * 3 + dx) * CO;
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] += v[c];
    }
  const long long thw = (long long)T_ * H * W;
  const long long b = bt / T_, s = pix - b * thw;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    const T q = (T)acc[c];  // the layer's output, rounded to the model dtype as the stored tensor would be
    if (u8) {
      float xf = (float)q;
      xf = xf < -1.0f ? -1.0f : (xf > 1.0f ? 1.0f : xf);   // torch.clamp(results, -1.0, 1.0)
      const T a = (T)(xf + 1.0f);                            // + 1.0
      const T m = (T)((float)a * 127.5f);                    // * 127.5
      u8[pix * CO + c] = (uint8_t)(float)m;                  // .to(torch.uint8): truncation   ('t h w c', B = 1)
    } else {
      out[(b * CO + c) * thw + s] = q;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// tile blending (in place on b): b[.., :o] = (1-w)*a[.., -o:] + w*b[.., :o], w = i/o in fp32
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void blend_kernel(const T* __restrict__ a, int Ha, int Wa, T* __restrict__ b, int Hb,
                                                    int Wb, long long rows, int o, int axis) {
  const long long per_row = axis == 0 ? (long long)o * Wb : (long long)Hb * o;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= rows * per_row) return;
  const long long r = gid / per_row, e = gid - r * per_row;
  int y, x, i;
  long long ai;
  if (axis == 0) {  // blend_v: first o rows of b with last o rows of a
    y = (int)(e / Wb);
    x = (int)(e - (long long)y * Wb);
    i = y;
    ai = (r * Ha + (Ha - o + y)) * (long long)Wa + x;
  } else {          // blend_h: first o columns of b with last o columns of a
    y = (int)(e / o);
    x = (int)(e - (long long)y * o);
    i = x;
    ai = (r * Ha + y) * (long long)Wa + (Wa - o + x);
  }
  const long long bi = (r * Hb + y) * (long long)Wb + x;
  const float w = (float)i / (float)o;
  b[bi] = (T)((1.0f - w) * (float)a[ai] + w * (float)b[bi]);
}

}  // namespace cvvae

using namespace cvvae;

#define CHECK_LAUNCH() return (int)hipGetLastError()

extern "C" {

size_t cvvae_packed_weight_bytes(int32_t Cout, int32_t Cin, int32_t taps) {
  const size_t co = (size_t)((Cout + 31) / 32) * 32;
  return co * (size_t)Cin * (size_t)taps * 2 + WEIGHT_TAIL_BYTES;  // + read-ahead tail for the prefetch ring
}

int cvvae_pack_weights(int32_t dtype, const void* src, int32_t Cout_src, int32_t Cin_src, int32_t taps, int64_t s_co,
                       int64_t s_ci, int64_t s_tap, int32_t Cin_pad, int32_t kchunk, void* dst, void* stream) {
  return cvvae_pack_weights_fold(dtype, src, Cout_src, Cin_src, taps, s_co, s_ci, s_tap, 1, 0, Cin_pad, kchunk, dst, stream);
}

static int pack_impl(int32_t dtype, const void* src, int32_t batch, int64_t s_batch, int32_t Cout_src, int32_t Cin_src,
                     int32_t taps, int64_t s_co, int64_t s_ci, int64_t s_tap, int32_t fold_n, int64_t s_fold, int32_t Cin_pad,
                     int32_t kchunk, void* dst, int64_t d_batch_bytes, void* stream, int32_t dst_taps = 0, int32_t dst_tap0 = 0);

int cvvae_pack_weights_fold(int32_t dtype, const void* src, int32_t Cout_src, int32_t Cin_src, int32_t taps, int64_t s_co,
                            int64_t s_ci, int64_t s_tap, int32_t fold_n, int64_t s_fold, int32_t Cin_pad, int32_t kchunk,
                            void* dst, void* stream) {
  return pack_impl(dtype, src, 1, 0, Cout_src, Cin_src, taps, s_co, s_ci, s_tap, fold_n, s_fold, Cin_pad, kchunk, dst, 0, stream);
}

int cvvae_pack_weights_batched(int32_t dtype, const void* src, int32_t batch, int64_t s_batch, int32_t Cout_src,
                               int32_t Cin_src, int32_t taps, int64_t s_co, int64_t s_ci, int64_t s_tap, int32_t Cin_pad,
                               int32_t kchunk, void* dst, int64_t dst_batch_stride, void* stream) {
  if (dtype == CVVAE_F32Q || dtype == CVVAE_F32Q6) return CVVAE_EUNSUPPORTED;  // per-item (attention) weights are 1x1: they stay in the three-MFMA form
  if (batch <= 0 || dst_batch_stride % 16 || (size_t)dst_batch_stride < cvvae_packed_weight_bytes(Cout_src, Cin_pad, taps * (dtype == CVVAE_F32 ? 3 : 1)))
    return CVVAE_EINVAL;
  return pack_impl(dtype, src, batch, s_batch, Cout_src, Cin_src, taps, s_co, s_ci, s_tap, 1, 0, Cin_pad, kchunk, dst,
                   dst_batch_stride, stream);
}

static int pack_impl(int32_t dtype, const void* src, int32_t batch, int64_t s_batch, int32_t Cout_src, int32_t Cin_src,
                     int32_t taps, int64_t s_co, int64_t s_ci, int64_t s_tap, int32_t fold_n, int64_t s_fold, int32_t Cin_pad,
                     int32_t kchunk, void* dst, int64_t d_batch_bytes, void* stream, int32_t dst_taps, int32_t dst_tap0) {
  if (dst_taps <= 0) dst_taps = taps;
  if (!src || !dst || Cout_src <= 0 || Cin_src <= 0 || taps <= 0 || kchunk <= 0 || kchunk % 16 || Cin_pad % kchunk ||
      Cin_pad < Cin_src || fold_n < 1)
    return CVVAE_EINVAL;
  // the packed layout is the same for every kernel family: [Cout/32][Cin_pad/16][tap][64 lanes][8] (k16-major);
  // kchunk only states the granularity Cin_pad was rounded to (the K-chunk of the consuming kernel instance)
  const int nb = (Cout_src + 31) / 32, nchunks = Cin_pad / 16, ksub = 1;
  const long long n = (long long)nb * nchunks * taps * ksub * 64;
  const int grid = (int)((n + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(pack_weights_kernel<__bf16>, dim3(grid, batch), dim3(256), 0, s, (const __bf16*)src, Cout_src, Cin_src,
                       taps, (long long)s_co, (long long)s_ci, (long long)s_tap, nchunks, ksub, (__bf16*)dst, n, fold_n,
                       (long long)s_fold, (long long)s_batch, (long long)(d_batch_bytes / 2), dst_taps, dst_tap0);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(pack_weights_kernel<_Float16>, dim3(grid, batch), dim3(256), 0, s, (const _Float16*)src, Cout_src,
                       Cin_src, taps, (long long)s_co, (long long)s_ci, (long long)s_tap, nchunks, ksub, (_Float16*)dst, n,
                       fold_n, (long long)s_fold, (long long)s_batch, (long long)(d_batch_bytes / 2), dst_taps, dst_tap0);
  else if (dtype == CVVAE_F32 || dtype == CVVAE_F32Q || dtype == CVVAE_F32Q6) {  // fp32 source -> split-precision records (3 per (k16, tap); one thread per record TRIPLE)
    // fast-fp32 pairs stay inside a run of kH*kW taps (the kernel walks 3-tap time kernels as time groups of kH*kW steps):
    // 27 / 54 (time-fold slots) / 9 -> runs of 9, 12 / 24 / 4 -> runs of 4 (the folded-upsample phases come through upfold_launch)
    int qrun = 0;
    if (dtype == CVVAE_F32Q || dtype == CVVAE_F32Q6) {
      qrun = taps % 9 == 0 ? 9 : (taps % 4 == 0 ? 4 : 0);
      if (!qrun || dst_taps % qrun || dst_tap0 % qrun) return CVVAE_EUNSUPPORTED;  // (1x1x1 weights: use CVVAE_F32)
    }
    hipLaunchKernelGGL(pack_weights_xp_kernel, dim3(grid, batch), dim3(256), 0, s, (const float*)src, Cout_src, Cin_src, taps,
                       (long long)s_co, (long long)s_ci, (long long)s_tap, nchunks, (_Float16*)dst, n, fold_n, (long long)s_fold,
                       (long long)s_batch, (long long)(d_batch_bytes / 2), dst_taps, dst_tap0, qrun, dtype == CVVAE_F32Q6 ? 1 : 0);
  }
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

// Time-fold slots.  A 3-tap time kernel at a clip boundary reads the SAME stored frame through two or three of its taps
// (replicate padding: CausalConv3d front 2, Conv3d(padding_mode="replicate") 1+1 -- vae_blocks3d_sd3.py:16-104,
// vae_models.py:266-328): W0.x + W1.x = (W0+W1).x.  The packed buffer therefore carries, after the 3 original time slots
// (taps [0, 3*nsp)), the slots W0+W1, W1+W2 and W0+W1+W2 (fp32 sums, rounded once), nsp taps each: 6*nsp taps per record
// group.  conv_fwd_kernel picks, per output frame, the slots whose input frames are distinct (conv_kernel.h, "time folds").
int cvvae_pack_weights_tfolds(int32_t dtype, const void* src, int32_t Cout_src, int32_t Cin_src, int32_t nsp, int64_t s_co,
                              int64_t s_ci, int64_t s_tap, int32_t Cin_pad, int32_t kchunk, void* dst, void* stream) {
  if (nsp <= 0) return CVVAE_EINVAL;
  const int es = dtype >= CVVAE_F32 ? 4 : 2;  // bytes per source element
  const char* sp = (const char*)src;
  int rc = pack_impl(dtype, src, 1, 0, Cout_src, Cin_src, 3 * nsp, s_co, s_ci, s_tap, 1, 0, Cin_pad, kchunk, dst, 0, stream, 6 * nsp, 0);
  if (rc) return rc;
  rc = pack_impl(dtype, sp, 1, 0, Cout_src, Cin_src, nsp, s_co, s_ci, s_tap, 2, nsp * s_tap, Cin_pad, kchunk, dst, 0, stream, 6 * nsp, 3 * nsp);
  if (rc) return rc;
  rc = pack_impl(dtype, sp + (long long)nsp * s_tap * es, 1, 0, Cout_src, Cin_src, nsp, s_co, s_ci, s_tap, 2, nsp * s_tap, Cin_pad, kchunk,
                 dst, 0, stream, 6 * nsp, 4 * nsp);
  if (rc) return rc;
  return pack_impl(dtype, sp, 1, 0, Cout_src, Cin_src, nsp, s_co, s_ci, s_tap, 3, nsp * s_tap, Cin_pad, kchunk, dst, 0, stream, 6 * nsp,
                   5 * nsp);
}

static int upfold_launch(int32_t dtype, const void* src, int32_t Cout, int32_t Cin, int32_t Cin_pad, int32_t tfold, void* dst,
                         int32_t dst_taps, int32_t dst_tap0, void* stream) {
  const int nb = (Cout + 31) / 32, nchunks = Cin_pad / 16, ntap = tfold ? 4 : 12;
  const long long per_phase = (long long)nb * nchunks * ntap * 64;
  const bool f32 = dtype == CVVAE_F32 || dtype == CVVAE_F32Q || dtype == CVVAE_F32Q6;
  const long long stride = (long long)(cvvae_packed_weight_bytes(Cout, Cin_pad, dst_taps * (f32 ? 3 : 1)) / 2);
  const int grid = (int)((4 * per_phase + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(pack_upfold_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)src, Cout, Cin, nchunks,
                       (__bf16*)dst, per_phase, stride, tfold, dst_taps, dst_tap0);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(pack_upfold_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)src, Cout, Cin, nchunks,
                       (_Float16*)dst, per_phase, stride, tfold, dst_taps, dst_tap0);
  else if (f32)
    hipLaunchKernelGGL(pack_upfold_xp_kernel, dim3(grid), dim3(256), 0, s, (const float*)src, Cout, Cin, nchunks, (_Float16*)dst,
                       per_phase, stride, tfold, dst_taps, dst_tap0, (dtype == CVVAE_F32Q || dtype == CVVAE_F32Q6) ? 4 : 0,
                       dtype == CVVAE_F32Q6 ? 1 : 0);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

// the four folded 3x2x2 phase kernels of cvvae_pack_weights_upfold with the time-fold slots appended: 24 taps per phase
int cvvae_pack_weights_upfold_tfolds(int32_t dtype, const void* src, int32_t Cout, int32_t Cin, int32_t Cin_pad, void* dst,
                                     void* stream) {
  if (!src || !dst || Cout <= 0 || Cin <= 0 || Cin_pad < Cin || Cin_pad % 16) return CVVAE_EINVAL;
  int rc = upfold_launch(dtype, src, Cout, Cin, Cin_pad, 0, dst, 24, 0, stream);
  if (!rc) rc = upfold_launch(dtype, src, Cout, Cin, Cin_pad, 3, dst, 24, 12, stream);
  if (!rc) rc = upfold_launch(dtype, src, Cout, Cin, Cin_pad, 4, dst, 24, 16, stream);
  if (!rc) rc = upfold_launch(dtype, src, Cout, Cin, Cin_pad, 1, dst, 24, 20, stream);
  return rc;
}

int cvvae_pack_weights_upfold(int32_t dtype, const void* src, int32_t Cout, int32_t Cin, int32_t Cin_pad, int32_t tfold,
                              void* dst, void* stream) {
  if (!src || !dst || Cout <= 0 || Cin <= 0 || Cin_pad < Cin || Cin_pad % 16 || tfold < 0 || tfold > 2) return CVVAE_EINVAL;
  return upfold_launch(dtype, src, Cout, Cin, Cin_pad, tfold, dst, tfold ? 4 : 12, 0, stream);
}

static int gn_nsplit(int64_t S) {
  int64_t n = (S + 8191) / 8192;  // >= 8192 pixels per block; <= 512 slabs keeps the finalize merge short
  if (n < 1) n = 1;
  if (n > 512) n = 512;
  return (int)n;
}

size_t cvvae_gn_workspace_bytes(int32_t rows, int32_t groups, int64_t S) {
  return (size_t)rows * (size_t)gn_nsplit(S) * (size_t)groups * 3 * sizeof(float);
}

int cvvae_gn_stats(int32_t dtype, const void* x, int32_t rows, int64_t S, int32_t C, int64_t pix_stride, int32_t groups,
                   float eps, const float* gamma, const float* beta, float* scale, float* shift, void* workspace,
                   void* stream) {
  if (!x || !gamma || !beta || !scale || !shift || !workspace || rows <= 0 || S <= 0) return CVVAE_EINVAL;
  if (groups <= 0 || groups > 32 || C % groups || (C / groups) % 4 || C % 8 || C > 2048 || pix_stride % 8) return CVVAE_EUNSUPPORTED;
  const int nsplit = gn_nsplit(S);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(gn_partial_kernel<__bf16>, dim3(nsplit, rows), dim3(256), 0, s, (const __bf16*)x, (long long)S, C,
                       (long long)pix_stride, groups, nsplit, (float*)workspace);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(gn_partial_kernel<_Float16>, dim3(nsplit, rows), dim3(256), 0, s, (const _Float16*)x, (long long)S, C,
                       (long long)pix_stride, groups, nsplit, (float*)workspace);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(gn_partial_kernel<float>, dim3(nsplit, rows), dim3(256), 0, s, (const float*)x, (long long)S, C,
                       (long long)pix_stride, groups, nsplit, (float*)workspace);
  else
    return CVVAE_EINVAL;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(rows), dim3(256), 0, s, (const float*)workspace, nsplit, groups, C, eps, gamma,
                     beta, scale, shift);
  CHECK_LAUNCH();
}

int cvvae_gn_silu_apply(int32_t dtype, const void* x, int32_t rows, int64_t S, int32_t C, int64_t pix_stride,
                        const float* scale, const float* shift, int32_t silu, void* out, void* stream) {
  if (!x || !scale || !shift || !out || rows <= 0 || S <= 0 || C <= 0 || C % 8 || pix_stride < C || pix_stride % 8) return CVVAE_EINVAL;
  const long long nvec = (long long)rows * S * (C / 8);
  long long blocks = (nvec + 255) / 256;
  if (blocks > 256LL * 64) blocks = 256LL * 64;  // grid-stride beyond 64 blocks per CU
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(gn_silu_apply_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)x, (long long)S, C,
                       (long long)pix_stride, scale, shift, silu, (__bf16*)out, nvec);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(gn_silu_apply_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, (const _Float16*)x, (long long)S, C,
                       (long long)pix_stride, scale, shift, silu, (_Float16*)out, nvec);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(gn_silu_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, (long long)S, C,
                       (long long)pix_stride, scale, shift, silu, (float*)out, nvec);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_gn_finalize(const float* partials, int32_t rows, int64_t slabs, int32_t C, int32_t groups, float eps,
                      const float* gamma, const float* beta, float* scale, float* shift, void* stream) {
  return cvvae_gn_finalize_frames(partials, rows, 1, slabs, C, groups, eps, gamma, beta, scale, shift, stream);
}

int cvvae_gn_finalize_frames(const float* partials, int32_t rows, int32_t frames, int64_t slabs, int32_t C, int32_t groups, float eps,
                             const float* gamma, const float* beta, float* scale, float* shift, void* stream) {
  if (!partials || !gamma || !beta || !scale || !shift || rows <= 0 || frames <= 0 || slabs <= 0 || slabs % frames || groups <= 0 ||
      C <= 0 || C % groups)
    return CVVAE_EINVAL;
  hipLaunchKernelGGL(gn_finalize_slabs_kernel, dim3(groups, rows * frames), dim3(256), 0, (hipStream_t)stream, partials,
                     (long long)slabs, groups, C, eps, gamma, beta, scale, shift, frames);
  CHECK_LAUNCH();
}

int64_t cvvae_gn_bwd_workspace_bytes(int32_t rows, int32_t groups, int64_t S) {
  if (rows <= 0 || groups <= 0 || S <= 0) return 0;
  return (int64_t)rows * gn_bwd_splits(S) * groups * 2 * (int64_t)sizeof(float);
}

int64_t cvvae_gn_bwd_params_workspace_bytes(int32_t rows, int32_t groups, int64_t S, int32_t C) {
  if (rows <= 0 || groups <= 0 || S <= 0 || C <= 0) return 0;
  return (int64_t)rows * gn_bwd_splits(S) * (groups + C) * 2 * (int64_t)sizeof(float);
}

int cvvae_gn_bwd_input_params(int32_t dtype, const void* x, const void* gy, const void* add, int32_t rows, int64_t S, int32_t C,
                              int32_t groups, const float* rstd, const float* nmean, const float* gamma, const float* beta,
                              int32_t silu, void* gx, float* dgamma, float* dbeta, void* workspace, void* stream) {
  if (!x || !gy || !gx || !rstd || !nmean || !gamma || !beta || !workspace || !dgamma || !dbeta || rows <= 0 || S <= 0 || C <= 0)
    return CVVAE_EINVAL;
  if (groups <= 0 || groups > 64 || C % groups || (C / groups) % 4 || C % 8 || C > 2048 || 256 % (C / 8)) return CVVAE_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    gn_bwd_launch<__bf16>(x, gy, add, rows, S, C, groups, rstd, nmean, gamma, beta, silu, gx, (float*)workspace, s, dgamma, dbeta);
  else if (dtype == CVVAE_F16)
    gn_bwd_launch<_Float16>(x, gy, add, rows, S, C, groups, rstd, nmean, gamma, beta, silu, gx, (float*)workspace, s, dgamma, dbeta);
  else if (dtype == CVVAE_F32)
    gn_bwd_launch<float>(x, gy, add, rows, S, C, groups, rstd, nmean, gamma, beta, silu, gx, (float*)workspace, s, dgamma, dbeta);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_gn_bwd_input(int32_t dtype, const void* x, const void* gy, const void* add, int32_t rows, int64_t S, int32_t C,
                       int32_t groups, const float* rstd, const float* nmean, const float* gamma, const float* beta, int32_t silu,
                       void* gx, void* workspace, void* stream) {
  if (!x || !gy || !gx || !rstd || !nmean || !gamma || !beta || !workspace || rows <= 0 || S <= 0 || C <= 0) return CVVAE_EINVAL;
  // 8-channel vectors that tile a 256-thread block; channel quads inside one group
  if (groups <= 0 || groups > 64 || C % groups || (C / groups) % 4 || C % 8 || C > 2048 || 256 % (C / 8)) return CVVAE_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    gn_bwd_launch<__bf16>(x, gy, add, rows, S, C, groups, rstd, nmean, gamma, beta, silu, gx, (float*)workspace, s);
  else if (dtype == CVVAE_F16)
    gn_bwd_launch<_Float16>(x, gy, add, rows, S, C, groups, rstd, nmean, gamma, beta, silu, gx, (float*)workspace, s);
  else if (dtype == CVVAE_F32)
    gn_bwd_launch<float>(x, gy, add, rows, S, C, groups, rstd, nmean, gamma, beta, silu, gx, (float*)workspace, s);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_softmax_bwd_rows(int32_t dtype, const void* p, int64_t ld_p, const float* gp, int64_t ld_g, int64_t rows, int32_t n_valid,
                           float alpha, void* gs, int64_t ld_o, void* stream) {
  if (!p || !gp || !gs || rows <= 0 || n_valid <= 0 || ld_p < n_valid || ld_g < n_valid || ld_o < n_valid) return CVVAE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(softmax_bwd_rows_kernel<__bf16>, dim3((unsigned)rows), dim3(256), 0, s, (const __bf16*)p, (long long)ld_p, gp,
                       (long long)ld_g, n_valid, alpha, (__bf16*)gs, (long long)ld_o);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(softmax_bwd_rows_kernel<_Float16>, dim3((unsigned)rows), dim3(256), 0, s, (const _Float16*)p, (long long)ld_p,
                       gp, (long long)ld_g, n_valid, alpha, (_Float16*)gs, (long long)ld_o);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(softmax_bwd_rows_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, (const float*)p, (long long)ld_p, gp,
                       (long long)ld_g, n_valid, alpha, (float*)gs, (long long)ld_o);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_upsample2x_sum(int32_t dtype, const void* g, int64_t N, int32_t H, int32_t W, int32_t C, void* out, void* stream) {
  if (!g || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return CVVAE_EINVAL;
  long long blocks = ((long long)N * H * W * (C / 8) + 255) / 256;
  if (blocks > 256LL * 64) blocks = 256LL * 64;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(upsample2x_sum_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)g, (long long)N, H, W, C,
                       (__bf16*)out);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(upsample2x_sum_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, (const _Float16*)g, (long long)N, H, W,
                       C, (_Float16*)out);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(upsample2x_sum_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)g, (long long)N, H, W, C,
                       (float*)out);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_layernorm(int32_t dtype, const void* x, int64_t P, int32_t C, float eps, const float* gamma, const float* beta,
                    void* out, void* stream) {
  if (!x || !out || !gamma || !beta || P <= 0 || C <= 0 || C % 8) return CVVAE_EINVAL;
  const int grid = (int)((P + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(layernorm_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)x, (long long)P, C, eps, gamma,
                       beta, (__bf16*)out);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(layernorm_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)x, (long long)P, C, eps,
                       gamma, beta, (_Float16*)out);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(layernorm_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (long long)P, C, eps,
                       gamma, beta, (float*)out);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_softmax_rows(int32_t dtype, const float* sc, int64_t rows, int32_t n_valid, int64_t ld_s, void* p, int64_t ld_p,
                       void* stream) {
  if (!sc || !p || rows <= 0 || n_valid <= 0 || ld_s < n_valid || ld_p < n_valid) return CVVAE_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<__bf16>, dim3((unsigned)rows), dim3(256), 0, s, sc, n_valid, (long long)ld_s,
                       (__bf16*)p, (long long)ld_p);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(softmax_rows_kernel<_Float16>, dim3((unsigned)rows), dim3(256), 0, s, sc, n_valid, (long long)ld_s,
                       (_Float16*)p, (long long)ld_p);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, sc, n_valid, (long long)ld_s,
                       (float*)p, (long long)ld_p);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_transpose(int32_t dtype, const void* in, int32_t batch, int32_t R, int32_t C, int64_t ld_in, int64_t bs_in,
                    void* out, int64_t ld_out, int64_t bs_out, void* stream) {
  if (!in || !out || batch <= 0 || R <= 0 || C <= 0) return CVVAE_EINVAL;
  if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(transpose16_kernel<uint32_t>, dim3((C + 31) / 32, (R + 31) / 32, batch), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)in, R, C, (long long)ld_in, (long long)bs_in, (uint32_t*)out, (long long)ld_out,
                       (long long)bs_out);
  else if (dtype == CVVAE_BF16 || dtype == CVVAE_F16)
    hipLaunchKernelGGL(transpose16_kernel<uint16_t>, dim3((C + 31) / 32, (R + 31) / 32, batch), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)in, R, C, (long long)ld_in, (long long)bs_in, (uint16_t*)out, (long long)ld_out,
                       (long long)bs_out);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_temporal_attention(int32_t dtype, const void* q, const void* k, const void* v, int32_t B, int32_t T, int64_t S,
                             int32_t C, void* out, void* stream) {
  if (!q || !k || !v || !out || B <= 0 || S <= 0 || T <= 0 || C <= 0 || C % 8) return CVVAE_EINVAL;
  const float scale = 1.0f / sqrtf((float)C);
  const long long P = (long long)B * S;
  hipStream_t s = (hipStream_t)stream;
  if (T > 8) {  // general frame count: one wave per (pixel, query frame)
    if (C > 2048 || P * T >= (1LL << 33)) return CVVAE_EUNSUPPORTED;
    const int gridg = (int)((P * T + 3) / 4);
    if (dtype == CVVAE_BF16)
      hipLaunchKernelGGL(temporal_attn_general_kernel<__bf16>, dim3(gridg), dim3(256), 0, s, (const __bf16*)q, (const __bf16*)k,
                         (const __bf16*)v, P, T, (long long)S, C, scale, (__bf16*)out);
    else if (dtype == CVVAE_F16)
      hipLaunchKernelGGL(temporal_attn_general_kernel<_Float16>, dim3(gridg), dim3(256), 0, s, (const _Float16*)q,
                         (const _Float16*)k, (const _Float16*)v, P, T, (long long)S, C, scale, (_Float16*)out);
    else if (dtype == CVVAE_F32)
      hipLaunchKernelGGL(temporal_attn_general_kernel<float>, dim3(gridg), dim3(256), 0, s, (const float*)q,
                         (const float*)k, (const float*)v, P, T, (long long)S, C, scale, (float*)out);
    else
      return CVVAE_EINVAL;
    CHECK_LAUNCH();
  }
  const int grid = (int)((P + 3) / 4);
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(temporal_attn_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)q, (const __bf16*)k,
                       (const __bf16*)v, P, T, (long long)S, C, scale, (__bf16*)out);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(temporal_attn_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)q, (const _Float16*)k,
                       (const _Float16*)v, P, T, (long long)S, C, scale, (_Float16*)out);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(temporal_attn_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)q, (const float*)k,
                       (const float*)v, P, T, (long long)S, C, scale, (float*)out);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_ncdhw_to_ndhwc(int32_t src_dtype, int32_t dst_dtype, const void* in, int32_t B, int32_t C, int32_t T, int32_t H,
                         int32_t W, int32_t Cpad, void* out, void* stream) {
  if (!in || !out || B <= 0 || C <= 0 || T <= 0 || H <= 0 || W <= 0 || Cpad < C || Cpad % 8) return CVVAE_EINVAL;
  const long long THW = (long long)T * H * W, npix = THW * B;
  const int grid = (int)((npix + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
#define L(TS, TD) \
  hipLaunchKernelGGL((ncdhw_to_ndhwc_kernel<TS, TD>), dim3(grid), dim3(256), 0, s, (const TS*)in, C, THW, Cpad, npix, (TD*)out)
  if (dst_dtype == CVVAE_BF16) {
    if (src_dtype == CVVAE_BF16) L(__bf16, __bf16);
    else if (src_dtype == CVVAE_F16) L(_Float16, __bf16);
    else if (src_dtype == 2) L(float, __bf16);
    else return CVVAE_EINVAL;
  } else if (dst_dtype == CVVAE_F16) {
    if (src_dtype == CVVAE_BF16) L(__bf16, _Float16);
    else if (src_dtype == CVVAE_F16) L(_Float16, _Float16);
    else if (src_dtype == 2) L(float, _Float16);
    else return CVVAE_EINVAL;
  } else if (dst_dtype == CVVAE_F32) {
    if (src_dtype == CVVAE_BF16) L(__bf16, float);
    else if (src_dtype == CVVAE_F16) L(_Float16, float);
    else if (src_dtype == 2) L(float, float);
    else return CVVAE_EINVAL;
  } else
    return CVVAE_EINVAL;
#undef L
  CHECK_LAUNCH();
}

int cvvae_ncdhw_to_rowpack(int32_t src_dtype, int32_t dst_dtype, const void* in, int32_t B, int32_t C, int32_t T, int32_t H,
                           int32_t W, int32_t pad_mode_w, void* out, void* stream) {
  if (!in || !out || B <= 0 || C <= 0 || C > 4 || T <= 0 || H <= 0 || W <= 0 || (pad_mode_w != 0 && pad_mode_w != 1)) return CVVAE_EINVAL;
  const long long THW = (long long)T * H * W, npo = (long long)B * T * H * (W + 3);
  const int grid = (int)((npo + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  // the 16 read-ahead elements behind the last pixel are zeroed (they are only ever multiplied by zero weights)
  const size_t es = dst_dtype == CVVAE_F32 ? 4 : 2;
  hipError_t me = hipMemsetAsync((char*)out + (size_t)npo * 4 * es, 0, 16 * es, s);
  if (me != hipSuccess) return (int)me;
#define L(TS, TD) \
  hipLaunchKernelGGL((ncdhw_to_rowpack_kernel<TS, TD>), dim3(grid), dim3(256), 0, s, (const TS*)in, C, THW, W, pad_mode_w, npo, (TD*)out)
  if (dst_dtype == CVVAE_BF16) {
    if (src_dtype == CVVAE_BF16) L(__bf16, __bf16);
    else if (src_dtype == CVVAE_F16) L(_Float16, __bf16);
    else if (src_dtype == CVVAE_F32) L(float, __bf16);
    else return CVVAE_EINVAL;
  } else if (dst_dtype == CVVAE_F16) {
    if (src_dtype == CVVAE_BF16) L(__bf16, _Float16);
    else if (src_dtype == CVVAE_F16) L(_Float16, _Float16);
    else if (src_dtype == CVVAE_F32) L(float, _Float16);
    else return CVVAE_EINVAL;
  } else
    return CVVAE_EUNSUPPORTED;  // (fp32 models keep the channel-padded first layer: no split-precision (3,3,1) instance)
#undef L
  CHECK_LAUNCH();
}

int cvvae_ndhwc_to_rowpack(int32_t dtype, const void* in, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W, int64_t pix_stride,
                           int32_t pad_mode_w, void* out, void* stream) {
  if (!in || !out || B <= 0 || C <= 0 || C > 4 || T <= 0 || H <= 0 || W <= 0 || pix_stride < C || (pad_mode_w != 0 && pad_mode_w != 1))
    return CVVAE_EINVAL;
  if (dtype != CVVAE_F16 && dtype != CVVAE_BF16) return CVVAE_EUNSUPPORTED;
  const long long npo = (long long)B * T * H * (W + 3);
  const int grid = (int)((npo + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  hipError_t me = hipMemsetAsync((char*)out + (size_t)npo * 4 * 2, 0, 16 * 2, s);
  if (me != hipSuccess) return (int)me;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(ndhwc_to_rowpack_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)in, C, (long long)pix_stride, W,
                       pad_mode_w, npo, (__bf16*)out);
  else
    hipLaunchKernelGGL(ndhwc_to_rowpack_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)in, C, (long long)pix_stride, W,
                       pad_mode_w, npo, (_Float16*)out);
  CHECK_LAUNCH();
}

int cvvae_ndhwc_to_ncdhw(int32_t dtype, const void* in, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                         int64_t pix_stride, void* out, void* stream) {
  if (!in || !out || B <= 0 || C <= 0 || T <= 0 || H <= 0 || W <= 0 || pix_stride < C) return CVVAE_EINVAL;
  const long long THW = (long long)T * H * W, npix = THW * B;
  const int grid = (int)((npix + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)in, C, THW,
                       (long long)pix_stride, npix, (__bf16*)out);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)in, C, THW,
                       (long long)pix_stride, npix, (_Float16*)out);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)in, C, THW,
                       (long long)pix_stride, npix, (float*)out);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_frames_u8_to_ndhwc(int32_t dtype, const uint8_t* frames, int64_t npix, int32_t Cpad, void* out, void* stream) {
  if (!frames || !out || npix <= 0 || Cpad < 8 || Cpad % 8) return CVVAE_EINVAL;
  const int grid = (int)((npix + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(frames_u8_to_ndhwc_kernel<__bf16>, dim3(grid), dim3(256), 0, s, frames, (long long)npix, Cpad, (__bf16*)out);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(frames_u8_to_ndhwc_kernel<_Float16>, dim3(grid), dim3(256), 0, s, frames, (long long)npix, Cpad,
                       (_Float16*)out);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(frames_u8_to_ndhwc_kernel<float>, dim3(grid), dim3(256), 0, s, frames, (long long)npix, Cpad,
                       (float*)out);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_ncdhw_to_frames_u8(int32_t dtype, const void* in, int64_t thw, uint8_t* frames, void* stream) {
  if (!in || !frames || thw <= 0) return CVVAE_EINVAL;
  const int grid = (int)((thw + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(ncdhw_to_frames_u8_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)in, (long long)thw, frames);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(ncdhw_to_frames_u8_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)in, (long long)thw,
                       frames);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(ncdhw_to_frames_u8_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)in, (long long)thw,
                       frames);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_resize_u8_axis(const uint8_t* in, uint8_t* out, int64_t outer, int32_t in_size, int32_t out_size, int64_t inner,
                         const int32_t* xmin, const int32_t* xsize, const int32_t* w, int32_t ksize, int32_t precision, void* stream) {
  if (!in || !out || !xmin || !xsize || !w || outer <= 0 || in_size <= 0 || out_size <= 0 || inner <= 0 || ksize <= 0 || precision < 1 ||
      precision > 22)
    return CVVAE_EINVAL;
  const long long n = (long long)outer * out_size * inner;
  if (n >= (1LL << 31) * 256) return CVVAE_EUNSUPPORTED;
  hipLaunchKernelGGL(resize_u8_axis_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, n, in_size, out_size,
                     (long long)inner, xmin, xsize, w, ksize, precision);
  CHECK_LAUNCH();
}

int cvvae_conv_out_gather(int32_t dtype, const float* V, int32_t B, int32_t T, int32_t H, int32_t W, int32_t Cout, int64_t ldv,
                          const float* bias, int32_t pad_mode_hw, void* out_ncdhw, uint8_t* out_u8, void* stream) {
  if (!V || !bias || B <= 0 || T <= 0 || H <= 0 || W <= 0 || ldv < ((9LL * Cout + 3) & ~3LL) || (ldv & 3) || (pad_mode_hw != 0 && pad_mode_hw != 1))
    return CVVAE_EINVAL;
  if ((out_ncdhw != nullptr) == (out_u8 != nullptr)) return CVVAE_EINVAL;
  if (Cout != 3) return CVVAE_EUNSUPPORTED;  // the shipped decoders: RGB
  if (out_u8 && B != 1) return CVVAE_EINVAL;
  const long long npix = (long long)B * T * H * W;
  const long long nblk = (long long)B * T * ((H + 7) / 8) * ((W + 31) / 32);
  if (nblk >= (1LL << 31)) return CVVAE_EUNSUPPORTED;
  const int grid = (int)nblk;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL((conv_out_gather_kernel<__bf16, 3>), dim3(grid), dim3(256), 0, s, V, T, H, W, (long long)ldv, bias, pad_mode_hw,
                       npix, (__bf16*)out_ncdhw, out_u8);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL((conv_out_gather_kernel<_Float16, 3>), dim3(grid), dim3(256), 0, s, V, T, H, W, (long long)ldv, bias, pad_mode_hw,
                       npix, (_Float16*)out_ncdhw, out_u8);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL((conv_out_gather_kernel<float, 3>), dim3(grid), dim3(256), 0, s, V, T, H, W, (long long)ldv, bias, pad_mode_hw,
                       npix, (float*)out_ncdhw, out_u8);
  else
    return CVVAE_EUNSUPPORTED;
  CHECK_LAUNCH();
}

int cvvae_blend(int32_t dtype, const void* a, int32_t Ha, int32_t Wa, void* b, int32_t Hb, int32_t Wb, int64_t rows,
                int32_t overlap, int32_t axis, void* stream) {
  if (!a || !b || rows <= 0 || overlap <= 0 || (axis != 0 && axis != 1)) return CVVAE_EINVAL;
  if (axis == 0 && (overlap > Ha || overlap > Hb || Wa != Wb)) return CVVAE_EINVAL;
  if (axis == 1 && (overlap > Wa || overlap > Wb || Ha != Hb)) return CVVAE_EINVAL;
  const long long n = rows * (axis == 0 ? (long long)overlap * Wb : (long long)Hb * overlap);
  const int grid = (int)((n + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(blend_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)a, Ha, Wa, (__bf16*)b, Hb, Wb,
                       (long long)rows, overlap, axis);
  else if (dtype == CVVAE_F16)
    hipLaunchKernelGGL(blend_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)a, Ha, Wa, (_Float16*)b, Hb, Wb,
                       (long long)rows, overlap, axis);
  else if (dtype == CVVAE_F32)
    hipLaunchKernelGGL(blend_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)a, Ha, Wa, (float*)b, Hb, Wb,
                       (long long)rows, overlap, axis);
  else
    return CVVAE_EINVAL;
  CHECK_LAUNCH();
}

int cvvae_abi_version(void) { return CVVAE_ABI_VERSION; }

}  // extern "C"
