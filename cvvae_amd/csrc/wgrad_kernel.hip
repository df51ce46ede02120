// wgrad_kernel.hip -- weight gradient of the codec's convolutions on MFMA, and the small reductions of the training-side backward
// (bias / GroupNorm-affine gradients, the adjoint of replicate padding).  SURVEY.md 8(f) rank 4: training the 3-D networks
// (/root/reference/lvdm/models/autoencoder.py:1057-1090 runs `z, xrec = self(x)` through the TRAINABLE encoder / decoder).
//
//   dW[co][ci][tap] = sum over output pixels p of  gy[p][co] * a[src(p, tap)][ci]
//
// a = the operand the forward conv multiplied (after GroupNorm + SiLU; padding applied by coordinate mapping exactly as the
// forward's staging does: replicate = clamp, zero = skip), gy = dL/d(conv output).  The contraction runs over PIXELS, which are
// the slow axis of NDHWC tensors, so both MFMA operands are transposed on their way into LDS:
//
//   * workgroup = 8 waves; tile = 128 output channels x 64 input channels x the KH*KW spatial taps of ONE time tap
//     (blockIdx.z = dt): 4 x 2 x 9 accumulator fragments of 32 x 32, nine per wave (144 registers);
//   * K panel = KP consecutive output pixels of one output row (b, t, y).  Staging: ONE task per thread and panel -- the input
//     pixels under 8 consecutive output pixels x 4 channels of a (7 sW + kW eight-byte global loads) or 8 pixels x 8 channels of
//     gy (sixteen-byte loads): 384 + 128 tasks = every thread -- whose (pixel, channel) block is transposed IN REGISTERS (static
//     indices) and written as 16-byte rows
//     into CHANNEL-major LDS copies  GS[co][k]  and  XS[dy][dx][ci][k] = a[.., (x0+k)*sW + dx - pw][ci]  -- one copy per kW tap, so
//     that every MFMA operand is an ALIGNED ds_read_b128 of 8 consecutive k (row pitch KP*2+16 bytes: conflict-free for the reads
//     and for the 16-byte writes); strides and both padding flavours live in the staging's coordinate map.  (The first version
//     wrote 2-byte elements -- 88 ds_write_b16 per thread and panel -- and ran at 0.09 of the MFMA peak: profiles/r4_train_step_v1_*.json);
//   * the loads of panel i+1 are issued before the MFMAs of panel i (register prefetch), LDS is single-buffered;
//   * MFMA: v_mfma_f32_32x32x16 with A = gy^T fragment (rows = output channels), B = a fragment (columns = input channels): an
//     accumulator lane holds one input channel and 16 output channels of one tap;
//   * the output pixels are cut into `nslab` slabs of rows; every workgroup writes its fp32 partial tile
//     part[slab][tap][co][ci]; wgrad_reduce_kernel sums the slabs in index order (deterministic) into PyTorch's [Cout][Cin][taps].
//   * fp32 models (XP): both operands are split bf16 hi + lo (gradients have fp32's range, so bf16 rather than fp16) and every
//     product runs as three MFMAs  g_hi a_hi + g_hi a_lo + g_lo a_hi  (~2^-16 relative); KP = 32 keeps the doubled LDS in budget.
//
// Bound: MFMA for the 3x3x3 layers (same FLOPs as the forward).  LDS is single-buffered (101 KB per panel set), so a panel's
// staging and its 36 MFMAs per wave alternate; the global loads of panel i+1 fly under the MFMAs of panel i.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cvvae.h"
#include "conv_kernel.h"

namespace cvvae {

struct WgradArgs {
  const void* a;
  const void* g;
  float* part;
  float* bias_part;  // wgrad_dma_kernel: per-slab sums of gy over the pixels, [nslab][Coutp] (nullptr: not wanted)
  int B, Ti, Hi, Wi, Cin;
  long long a_ps;
  int To, Ho, Wo, Cout;
  long long g_ps;
  int kT, sT, sH, sW, pt, ph, pw, mode_t, mode_hw;
  int nslab, rows_total, n_ci_blk, n_coci, n_members;
  int Coutp, Cinp;  // padded to multiples of 128 / 64: the partial buffer's channel extents
};

// 16-bit pair (a = element of the earlier pixel, b = of the later one) from channel j of two pixels' packed 8-channel vectors
__device__ __forceinline__ unsigned pair16(const uint4& pa, const uint4& pb, int j) {
  const unsigned wa = j < 2 ? pa.x : (j < 4 ? pa.y : (j < 6 ? pa.z : pa.w));
  const unsigned wb = j < 2 ? pb.x : (j < 4 ? pb.y : (j < 6 ? pb.z : pb.w));
  return (j & 1) ? ((wa >> 16) | (wb & 0xffff0000u)) : ((wa & 0xffffu) | (wb << 16));
}

template <typename T, int KHW, bool XP, int SW>
__global__ __launch_bounds__(512, 1) void wgrad_kernel(const WgradArgs p) {
  using TIO = std::conditional_t<XP, float, T>;
  using v8 = typename Tr<T>::v8;
  // output pixels per K panel (single-tap kernels -- shortcuts, linear layers -- take longer panels: one accumulator fragment per
  // wave is little MFMA work per staged panel)
  constexpr int KP = KHW == 1 ? (XP ? 64 : 128) : (XP ? 32 : 64);
  constexpr int ROWP = KP * 2 + 16;       // LDS row pitch (bytes): odd multiple of 16
  constexpr int NSP = KHW * KHW;          // spatial taps
  constexpr int CO = 128, CI = 64;
  constexpr int NPART = XP ? 2 : 1;       // hi (and lo) copies
  constexpr int XS_BYTES = NSP * CI * ROWP, GS_BYTES = CO * ROWP;
  constexpr int KB = KP / 8;              // 8-pixel k blocks per panel
  constexpr int NL = 7 * SW + KHW;        // input pixels one x task loads: the 8 outputs' taps along W
  // channels per staging task.  x tasks take 4 (one 8-byte load per pixel for 16-bit models, 16 bytes of floats for fp32 ones): with 3
  // input rows that makes 384 x tasks + 128 g tasks (8 channels, 16-byte loads) = one task for EVERY thread of the workgroup
  // (8-channel x tasks left three of the eight waves idle while the others staged)
  constexpr int CHX = 4, CHG = XP ? 4 : 8;
  constexpr int NCX = CI / CHX, NCG = CO / CHG;       // channel groups of the x / g tile
  constexpr int NXT = KHW * KB * NCX, NGT = KB * NCG;  // staging tasks per panel: x (dy, k block, channel group), g (k block, group)
  static_assert(NXT + NGT <= 512, "one staging task per thread");
  static_assert(NPART * (XS_BYTES + GS_BYTES) <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem[NPART * (XS_BYTES + GS_BYTES)];
  char* const xs = smem;                       // [part][tap][ci][ROWP]
  char* const gs = smem + NPART * XS_BYTES;    // [part][co][ROWP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware block map (consecutive block ids run on the 8 XCDs round-robin, each with its own L2): the n = n_co * n_ci * kT
  // workgroups that share a slab -- they read the same gy rows and the same / neighbouring a rows -- are consecutive ON ONE XCD
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int member = idx % p.n_members, slab = (idx / p.n_members) * 8 + xcd;
  if (slab >= p.nslab) return;
  const int coci = member % p.n_coci, dt = member / p.n_coci;
  const int co_blk = coci / p.n_ci_blk, ci_blk = coci % p.n_ci_blk;
  const int co0 = co_blk * CO, ci0 = ci_blk * CI;
  // the panels of all output rows, cut into nslab contiguous runs (a flattened 1x1 layer is ONE long row)
  const int npanel_row = (p.Wo + KP - 1) / KP;
  const long long ptotal = (long long)p.rows_total * npanel_row;
  const long long pbeg = ptotal * slab / p.nslab, npanels = ptotal * (slab + 1) / p.nslab - pbeg;

  const TIO* __restrict__ ap = reinterpret_cast<const TIO*>(p.a);
  const TIO* __restrict__ gp = reinterpret_cast<const TIO*>(p.g);

  f32x16 acc[NSP];
#pragma unroll
  for (int t = 0; t < NSP; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // ---- staging: ONE task per thread and panel.  Threads [0, NXT): x task (dy, k block kb, channel group): the NL input pixels
  //      under the 8 output pixels of the block, CHX channels each; the 8 x CH (pixel, channel) block is transposed IN REGISTERS
  //      (static indices: free) and written as 16-byte rows  XS[dy][dx][channel][8 consecutive k]  -- one row per channel and kW tap.
  //      Threads [NXT, NXT + NGT): g task (k block, channel group) the same way into GS[channel][k].
  //      Task -> lane order: 16 consecutive lanes are 4 k blocks x 4 channel groups (4-channel tasks) or 8 k blocks x 2 groups
  //      (8-channel tasks) -- rows 4 (8) apart are 16 (32) banks apart at this row pitch, so the 16 sixteen-byte writes of a quarter
  //      wave tile all 64 banks (channel-group-fastest order was 4- to 8-way conflicted).
  const bool is_x = tid < NXT, is_g = tid >= NXT && tid < NXT + NGT;
  auto decode = [](int t, int ch, int nc, int& kb, int& cg, int& rest) {
    const int kl = ch == 4 ? 2 : 3, cl = ch == 4 ? 2 : 1;   // low bits of kb / cg inside a 16-lane group
    const int kbl = t & ((1 << kl) - 1), cgl = (t >> kl) & ((1 << cl) - 1);
    int r = t >> 4;
    const int nkh = KB >> kl > 0 ? KB >> kl : 1, nch = nc >> cl;
    const int kbh = r % nkh;
    r /= nkh;
    kb = kbh * (1 << kl) + kbl;
    cg = (r % nch) * (1 << cl) + cgl;
    rest = r / nch;
  };
  static_assert(KB >= 8 || (KB == 4 && CHG == 4), "k blocks per panel vs the lane order of the 8-channel tasks");
  int xt_kb, xt_cg, xt_dy, gt_kb, gt_cg, gt_rest;
  decode(tid, CHX, NCX, xt_kb, xt_cg, xt_dy);
  decode(tid - NXT, CHG, NCG, gt_kb, gt_cg, gt_rest);
  (void)gt_rest;
  constexpr int NREG = NL > 8 ? NL : 8;
  uint4 raw[NREG];
  auto load_panel = [&](long long pi) {
    const long long pg_ = pbeg + pi;
    const int row = (int)(pg_ / npanel_row), x0 = (int)(pg_ % npanel_row) * KP;
    const int yo = row % p.Ho, to = (row / p.Ho) % p.To, b = row / (p.Ho * p.To);
    if (is_x) {
      bool zrow = false;
      const int ts = map_coord(to * p.sT + dt - p.pt, p.Ti, p.mode_t, zrow);
      const int ys = map_coord(yo * p.sH + xt_dy - p.ph, p.Hi, p.mode_hw, zrow);
      const int c = ci0 + xt_cg * CHX;
      const TIO* rowp = ap + (((long long)b * p.Ti + ts) * p.Hi + ys) * p.Wi * p.a_ps + c;
      const int xb = (x0 + xt_kb * 8) * SW - p.pw;  // unpadded input column of pixel 0 of my block
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        bool zero = zrow;
        const int xsrc = map_coord(xb + i, p.Wi, p.mode_hw, zero);
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (!zero && c < p.Cin) {
          if constexpr (XP) {
            r = *reinterpret_cast<const uint4*>(rowp + (long long)xsrc * p.a_ps);   // 4 floats
          } else {
            const uint2 v2 = *reinterpret_cast<const uint2*>(rowp + (long long)xsrc * p.a_ps);  // 4 x 16 bit
            r = make_uint4(v2.x, v2.y, 0u, 0u);
          }
        }
        raw[i] = r;
      }
    } else if (is_g) {
      const int c = co0 + gt_cg * CHG;
      const TIO* rowp = gp + ((((long long)b * p.To + to) * p.Ho + yo) * p.Wo) * p.g_ps + c;
      const int kx = x0 + gt_kb * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (kx + i < p.Wo && c < p.Cout) r = *reinterpret_cast<const uint4*>(rowp + (long long)(kx + i) * p.g_ps);
        raw[i] = r;
      }
    }
  };
  // my task's pixels -> the storage type T, packed per pixel: `hi` (16-bit models: the loaded 8-channel vector itself) and, for
  // fp32 models (4 floats per pixel), hi = T(x) and lo = T(x - hi) in the low halves of the vectors
  auto convert = [&](auto n_tag, uint4 (&hi)[NREG], uint4 (&lo)[NREG]) {
    constexpr int N = decltype(n_tag)::value;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if constexpr (XP) {
        const float f[4] = {__uint_as_float(raw[i].x), __uint_as_float(raw[i].y), __uint_as_float(raw[i].z), __uint_as_float(raw[i].w)};
        float fh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, fl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          fh[j] = (float)(T)f[j];
          fl[j] = f[j] - fh[j];
        }
        hi[i] = pack8<T>(fh);
        lo[i] = pack8<T>(fl);
      } else {
        hi[i] = raw[i];
      }
    }
  };
  // pixels first, first + step, ..., first + 7 step of px -> CH channel rows of 8 consecutive k each (16-byte LDS writes)
  auto put_block = [&](auto nch_tag, const uint4 (&px)[NREG], int first, int step, char* dst) {
    constexpr int NCH = decltype(nch_tag)::value;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      uint4 o;
      o.x = pair16(px[first], px[first + step], j);
      o.y = pair16(px[first + 2 * step], px[first + 3 * step], j);
      o.z = pair16(px[first + 4 * step], px[first + 5 * step], j);
      o.w = pair16(px[first + 6 * step], px[first + 7 * step], j);
      *reinterpret_cast<uint4*>(dst + j * ROWP) = o;
    }
  };
  auto store_panel = [&]() {
    uint4 hi[NREG], lo[NREG];
    if (is_x) {
      convert(std::integral_constant<int, NL>{}, hi, lo);
#pragma unroll
      for (int dx = 0; dx < KHW; ++dx) {
        char* dst = xs + ((xt_dy * KHW + dx) * CI + xt_cg * CHX) * ROWP + xt_kb * 16;
        put_block(std::integral_constant<int, CHX>{}, hi, dx, SW, dst);
        if constexpr (XP) put_block(std::integral_constant<int, CHX>{}, lo, dx, SW, dst + XS_BYTES);
      }
    } else if (is_g) {
      convert(std::integral_constant<int, 8>{}, hi, lo);
      char* dst = gs + (gt_cg * CHG) * ROWP + gt_kb * 16;
      put_block(std::integral_constant<int, CHG>{}, hi, 0, 1, dst);
      if constexpr (XP) put_block(std::integral_constant<int, CHG>{}, lo, 0, 1, dst + GS_BYTES);
    }
  };

  const int cof = wave & 3, cif = wave >> 2;
  const unsigned a_off = (unsigned)((cof * 32 + (lane & 31)) * ROWP + (lane >> 5) * 16);
  const unsigned b_off = (unsigned)((cif * 32 + (lane & 31)) * ROWP + (lane >> 5) * 16);

  // 16-bit models: the loads of panel i+1 fly under the MFMAs of panel i (40-68 registers of prefetch).  fp32 models: no room
  // beside the 144 accumulators and the three-MFMA operand sets -- the loads are issued right before they are staged
  constexpr bool PREFETCH = !XP;
  if (PREFETCH && npanels > 0) load_panel(0);
  for (long long pi = 0; pi < npanels; ++pi) {
    if (!PREFETCH) load_panel(pi);
    store_panel();
    __syncthreads();
    if (PREFETCH && pi + 1 < npanels) load_panel(pi + 1);
#pragma unroll
    for (int s = 0; s < KP / 16; ++s) {
      const v8 ah = *reinterpret_cast<const v8*>(gs + a_off + s * 32);
      v8 al;
      if constexpr (XP) al = *reinterpret_cast<const v8*>(gs + GS_BYTES + a_off + s * 32);
#pragma unroll
      for (int t = 0; t < NSP; ++t) {
        const v8 bh = *reinterpret_cast<const v8*>(xs + t * (CI * ROWP) + b_off + s * 32);
        acc[t] = Tr<T>::mfma(ah, bh, acc[t]);
        if constexpr (XP) {
          const v8 bl = *reinterpret_cast<const v8*>(xs + XS_BYTES + t * (CI * ROWP) + b_off + s * 32);
          acc[t] = Tr<T>::mfma(ah, bl, acc[t]);
          acc[t] = Tr<T>::mfma(al, bh, acc[t]);
        }
      }
    }
    __syncthreads();
  }
  // partial tile: part[slab][co block][ci block][128 co][tap][64 ci] (see wgrad_reduce_kernel); accumulator register i of a lane =
  // output channel 8*(i/4) + 4*(lane/32) + i%4 of the fragment, input channel lane % 32
  const int ntaps = p.kT * NSP;
#pragma unroll
  for (int t = 0; t < NSP; ++t) {
    float* o = p.part + (((((long long)slab * (p.Coutp / CO) + co_blk) * p.n_ci_blk + ci_blk) * CO + cof * 32) * ntaps + dt * NSP + t) * CI +
               cif * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) o[(long long)(8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)) * ntaps * CI] = acc[t][i];
  }
}

// ---------------------------------------------------------------------------------------------------------
// wgrad_dma_kernel (round 5; 16-bit models, 3x3 spatial taps): the same tile and the same partial-tile output as wgrad_kernel, with
// NOTHING staged through registers.  The contraction runs over pixels, the slow axis of both operands, and wgrad_kernel pays for
// that with a register transpose of every staged block and 96 KB of transposing LDS writes per panel, alternating with its 36
// MFMAs per wave behind two barriers (0.23 of the MFMA peak, the matrix pipe 0.29 busy at 2.2 GHz: not power-limited).  gfx950 has
// the two instructions that remove both: global_load_lds_dwordx4 (a wave-load puts lane i's 16 bytes at LDS base + 16 i: the tiles
// go global -> LDS in their NATURAL pixel-major layout) and ds_read_b64_tr_b16 (a 16-lane group reads a [4 rows][16 columns] block of
// 16-bit elements through per-lane 8-byte chunk addresses and every lane receives one COLUMN: 4 consecutive k of one channel -- the
// K-major MFMA operand out of a pixel-major image).
//   * LDS images per panel of KP = 64 output pixels of one output row:  GS[k][128 co] (row pitch 256 B) and, per kernel row dy,
//     XS[dy][input pixel][64 ci] (pitch 128 B) holding the (KP-1) sW + 3 input pixels under the panel ONCE -- the three kW taps read
//     the same rows at per-lane row offsets (the transpose read takes any 8-byte-aligned chunk address), where wgrad_kernel kept one
//     transposed copy per tap.  44 KB per panel (sW = 1): THREE panels in flight; 69 KB (sW = 2): two.
//   * 16-byte pieces of a row are stored XOR-swizzled (by row bits, chosen when the wave-load's lanes pick their source addresses:
//     the LDS side of a wave-load is lane-linear) so that the 32 lanes of a transpose-read half cover all 64 banks (sW = 1).
//   * padding = the source address of a lane: replicate clamps it, zero padding / pixels past the row end / channels past the stored
//     ones read a 512-byte zero page at the end of the workspace (cleared by every workgroup before its first wave-load).
//   * the wave-loads of a panel are dealt to the waves statically, the same number per panel for a wave (waves 0-2: one more), so
//     "panel i has landed" is an exact s_waitcnt vmcnt(my loads per panel) -- nothing else of this loop touches vector memory -- and
//     ONE barrier per panel orders landing and reuse.
//   * a wave owns a pair of co-fragments x four taps (+ tap 8 for one of them): 9 MFMAs per k16 step from 14 transpose reads; the
//     bias gradient (sum of gy over the pixels) is one more MFMA per step against ones in the workgroups that own it (ABI 12).
//   * FAST (per launch): wave-loads as scalar base + 32-bit lane offset; the per-lane 64-bit form serves ragged tiles and far zero pages.
// ---------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
#ifndef CVVAE_WGRAD_DEPTH
#define CVVAE_WGRAD_DEPTH 3  // input fragments requested ahead of their first MFMA (tuning aid)
#endif
#ifndef CVVAE_WGRAD_STAGGER
#define CVVAE_WGRAD_STAGGER 1  // the two waves of a SIMD request / multiply in opposite orders (tuning aid)
#endif
#ifndef CVVAE_WGRAD_ABLATE
// timing aid, scratch builds only (the results are WRONG when nonzero): leave out  1 the transpose reads,  2 the wave-load instruction
// (its address arithmetic stays),  4 the whole request code,  8 the MFMAs -- what the loop costs without each of its parts
#define CVVAE_WGRAD_ABLATE 0
#endif

template <typename T, int SW, bool FAST>
__global__ __launch_bounds__(512, 1) void wgrad_dma_kernel(const WgradArgs p, void* __restrict__ zero_page) {
  using v8 = typename Tr<T>::v8;
  constexpr int KP = 64, CO = 128, CI = 64, NSP = 9;
  constexpr int XR = (KP - 1) * SW + 3;               // input pixels under a panel, per kernel row
  constexpr int XRP = (XR + 7) / 8 * 8;               // ... in whole wave-loads of 8 rows x 128 bytes
  constexpr int XS_B = 3 * XRP * 128, GS_B = KP * 256, BUF_B = XS_B + GS_B;
  constexpr int NBUF = 3 * BUF_B <= 160 * 1024 ? 3 : 2;
  static_assert(NBUF * BUF_B <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) char smem[NBUF * BUF_B];
  typedef __attribute__((address_space(3))) void* lptr_t;
  typedef __attribute__((address_space(3))) s16x4* lrd_t;

  const int tid = threadIdx.x, lane_ = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = lane_;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int member = idx % p.n_members, slab = (idx / p.n_members) * 8 + xcd;
  if (slab >= p.nslab) return;
  const int coci = member % p.n_coci, dt = member / p.n_coci;
  const int co_blk = coci / p.n_ci_blk, ci_blk = coci % p.n_ci_blk;
  const int co0 = co_blk * CO, ci0 = ci_blk * CI;
  const int npanel_row = (p.Wo + KP - 1) / KP;
  const long long ptotal = (long long)p.rows_total * npanel_row;
  const long long pbeg = ptotal * slab / p.nslab, npanels = ptotal * (slab + 1) / p.nslab - pbeg;
  const T* __restrict__ ap = reinterpret_cast<const T*>(p.a);
  const T* __restrict__ gp = reinterpret_cast<const T*>(p.g);
  const char* const zp = reinterpret_cast<const char*>(zero_page);
  // The zero page is cleared HERE, by every workgroup (they all store the same zeros), not by a memset launched in front of the
  // kernel.  The stores are acknowledged by this XCD's L2 (vmcnt) before the barrier; the wave-loads that read the page come
  // after it and go through the same L2.
  if (tid < 128) __atomic_store_n(reinterpret_cast<unsigned*>(zero_page) + tid, 0u, __ATOMIC_RELAXED);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 acc[NSP];
#pragma unroll
  for (int t = 0; t < NSP; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // the BIAS gradient (sum of gy over the pixels) as a by-product: the gy^T fragments are in registers anyway, and one more MFMA per
  // k16 step against a matrix of ones sums them over the pixels.  Done once per output channel: by the workgroups of input-channel
  // block 0 and time tap 0, in the four waves (cif = 0) whose first co-fragments cover the 128 channels.  (A separate pass over gy
  // per convolution was 70 launches = 2.7 ms of a training step.)
  f32x16 accb;
#pragma unroll
  for (int i = 0; i < 16; ++i) accb[i] = 0.f;

  // ---- my wave-loads of a panel.  An input image row dy is C = XRP / 8 wave-loads (8 pixels x 128 bytes each), the gradient image
  //      KP / 4 = 16 (4 pixels x 256 bytes).  Wave w issues, in this order: for dy = 0, 1, 2 the input chunks w, w + 8, .. of row dy
  //      (C = 8 JX + 1); the gradient chunks w and w + 8; and -- waves 0, 1, 2 only -- the last chunk C - 1 of row dy = w.  WHICH row
  //      and chunk a load fetches is therefore known at compile time except for the one extra load (the first form dealt the loads
  //      round-robin, ii = wave + 8 k: every load selected its row base, its LDS slot and its lane constants at run time -- ~35 scalar
  //      and ~25 vector instructions per wave-load), and a lane's (pixel, 16-byte piece) inside a chunk is the same for all of them.
  //      Waves 0-2 wait for one more load per panel than the others (NLW_LO + 1).
  constexpr int C = XRP / 8, JX = C / 8, NLW_LO = 3 * JX + 2;
  static_assert(C % 8 == 1 && KP / 4 == 16, "wave-load assignment");
  const int xr8 = lane >> 3;                                             // my pixel inside an input chunk
  const int xq = (lane & 7) ^ (((xr8 >> 1) & 1) << 2);                   // my 16-byte piece (swizzled by bit 1 of the row = of xr8)
  const bool cx_ok = ci0 + xq * 8 < (int)p.a_ps;
  const unsigned lcx = (unsigned)((ci0 + xq * 8) * (int)sizeof(T));
  const int gr4 = wave * 4 + (lane >> 4);                                // my pixel inside the gradient image (first chunk)
  const int gq = (lane & 15) ^ ((gr4 & 3) << 2);
  const bool cg_ok = co0 + gq * 8 < (int)p.g_ps;
  const unsigned lcg = (unsigned)((co0 + gq * 8) * (int)sizeof(T));
  const unsigned long long zlane = (unsigned long long)(size_t)(zp + (lane & 31) * 16);
  const unsigned apitch = (unsigned)(p.a_ps * sizeof(T)), gpitch = (unsigned)(p.g_ps * sizeof(T));
  // FAST (chosen by the launcher: whole gradient panels, every channel of the tile stored, and the input tensor and the zero page
  // within 4 GB of each other): a wave-load is SCALAR base + 32-bit lane offset.  The ablated builds of the first form
  // (profiles/r5_ab_wgrad_parts.log) showed the requests costing their issue time in full ON TOP of the MFMA time (vector ALU work
  // of the other wave of a SIMD is not hidden under MFMAs: the round-2 probe, profiles/r2_peak_probe.txt).  Here an input load is
  // add / clamp / multiply-add (+ compare / select under zero padding) relative to `lo` = the lower of the two addresses, a
  // gradient load is no vector instruction at all.
  // (a layer with replicate padding on every axis never reads the zero page: its base is the tensor's, wherever the workspace lies --
  //  in a full training step the allocator puts most workspaces more than 4 GB from the activations)
  const bool needs_zero = p.mode_hw == 0 || p.mode_t == 0;
  const unsigned long long lo64 = !FAST ? 0ull : (!needs_zero || (size_t)ap < (size_t)zp) ? (unsigned long long)(size_t)ap : (unsigned long long)(size_t)zp;
  const unsigned zvoff = (unsigned)((unsigned long long)(size_t)zp - lo64) + (unsigned)(lane & 31) * 16u;
  const unsigned gvoff = (unsigned)gr4 * gpitch + lcg;                   // (FAST) my offset inside the gradient panel
  constexpr unsigned NONE = 0xffffffffu;
  unsigned xrel[3];
  // (panels are requested in order: the (batch, frame, row, panel-in-row) position advances by counters -- the divisions of the
  //  first build were a hundred instructions per panel)
  int q_xp = (int)(pbeg % npanel_row), q_yo, q_to, q_b;
  {
    const int row0 = (int)(pbeg / npanel_row);
    q_yo = row0 % p.Ho;
    q_to = (row0 / p.Ho) % p.To;
    q_b = row0 / (p.Ho * p.To);
  }
  const int wi1 = p.Wi - 1;
  const bool zero_hw = p.mode_hw == 0;
  // (wave-uniform) bases of the three input rows under the current output row -- 0: none (zero padding in time / height) -- and of the
  // gradient row: recomputed when the row changes, i.e. every npanel_row panels
  unsigned long long xrow[3], grow;
  auto row_bases = [&]() __attribute__((always_inline)) {
    bool zt = false;
    const int ts = map_coord(q_to * p.sT + dt - p.pt, p.Ti, p.mode_t, zt);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      bool z = zt;
      const int ys = map_coord(q_yo * p.sH + dy - p.ph, p.Hi, p.mode_hw, z);
      xrow[dy] = z ? 0ull : (unsigned long long)(size_t)ap + (unsigned long long)((((long long)q_b * p.Ti + ts) * p.Hi + ys) * p.Wi) * apitch;
      if (FAST) xrel[dy] = z ? NONE : (unsigned)(xrow[dy] - lo64);
    }
    grow = (unsigned long long)(size_t)gp + (unsigned long long)((((long long)q_b * p.To + q_to) * p.Ho + q_yo) * p.Wo) * gpitch;
  };
  row_bases();
  bool q_new_row = false;
  const unsigned sm0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)smem) + (unsigned)wave * 1024u;
  // The wave-load is issued BY HAND: with a global_load_lds it knows of still pending, hipcc puts s_waitcnt vmcnt(0) in front of
  // the next LDS read that may alias it -- i.e. in front of this panel's first fragment read, which would wait for the panels just
  // requested.  Landing and reuse are ordered by the explicit vmcnt + barrier of the panel loop instead.
  // (m0 is a reserved register: nothing else of this kernel uses it)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  auto dma = [&](unsigned long long src, unsigned voff, unsigned long long sbase, unsigned la) __attribute__((always_inline)) {
    if (CVVAE_WGRAD_ABLATE & 2) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0" ::"v"(src), "s"(la), "v"(voff), "s"(sbase) : "memory", "m0");
    else if (FAST) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(la) : "memory", "m0");
    else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(la) : "memory", "m0");
  };
#pragma clang diagnostic pop
  // one input chunk: pixels xs .. xs + 7 (xs wave-uniform) of the row whose base is rel (FAST) / rowbase, to LDS address la
  auto xload = [&](unsigned rel, unsigned long long rowbase, int xs, unsigned la) __attribute__((always_inline)) {
    const int xi = xs + xr8;
    const int xc = min(max(xi, 0), wi1);
    if (FAST) {
      unsigned voff = zvoff;
      if (rel != NONE) {                      // (wave-uniform)
        voff = __umul24((unsigned)xc, apitch) + lcx + rel;
        if (zero_hw) voff = xi == xc ? voff : zvoff;
      }
      dma(0ull, voff, lo64, la);
    } else {
      // branch-free per lane: clamp, one 32 x 32 -> 64 multiply-add, a mask select between the source and the zero page
      const bool ok = cx_ok & (rowbase != 0ull) & (!zero_hw | (xi == xc));
      const unsigned long long m = ok ? ~0ull : 0ull;
      const unsigned long long a = rowbase + (unsigned long long)(unsigned)xc * apitch + lcx;
      dma((a & m) | (zlane & ~m), 0u, 0ull, la);
    }
  };
  auto issue = [&](unsigned buf_off) __attribute__((always_inline)) {
    if (CVVAE_WGRAD_ABLATE & 4) return;
    if (q_new_row) row_bases();
    const int x0 = q_xp * KP;
    q_new_row = false;
    if (++q_xp == npanel_row) {
      q_xp = 0;
      q_new_row = true;
      if (++q_yo == p.Ho) {
        q_yo = 0;
        if (++q_to == p.To) {
          q_to = 0;
          ++q_b;
        }
      }
    }
    const int xbase = x0 * SW - p.pw + wave * 8;     // first pixel of my first chunk
    const unsigned lbase = sm0 + buf_off;            // LDS address of chunk `wave` of input row 0
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int j = 0; j < JX; ++j) xload(xrel[dy], xrow[dy], xbase + 64 * j, lbase + (unsigned)(dy * (XRP * 128) + j * 8192));
    const unsigned long long gbase = grow + (unsigned long long)((unsigned)x0 * gpitch);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (FAST) dma(0ull, gvoff, gbase + (unsigned long long)((unsigned)(32 * j) * gpitch), lbase + (unsigned)(XS_B + j * 8192));
      else {
        const int kx = x0 + gr4 + 32 * j;
        const bool ok = cg_ok & (kx < p.Wo);
        const unsigned long long m = ok ? ~0ull : 0ull;
        const unsigned long long a = grow + (unsigned long long)(unsigned)kx * gpitch + lcg;
        dma((a & m) | (zlane & ~m), 0u, 0ull, lbase + (unsigned)(XS_B + j * 8192));
      }
    }
    if (wave < 3) {   // the last chunk of input row dy = wave
      const unsigned rel = wave == 0 ? xrel[0] : (wave == 1 ? xrel[1] : xrel[2]);
      const unsigned long long rb = wave == 0 ? xrow[0] : (wave == 1 ? xrow[1] : xrow[2]);
      xload(rel, rb, xbase - wave * 8 + (C - 1) * 8, sm0 - (unsigned)wave * 1024u + buf_off + (unsigned)(wave * (XRP * 128) + (C - 1) * 1024));
    }
  };

  // ---- which products a wave owns.  The tile is 4 co-fragments x 2 ci-fragments x 9 taps of 32 x 32 products.  The first working form
  //      gave a wave ONE (co, ci) fragment pair and all nine taps: 2 transpose reads of gy^T and 18 of the input per k16 step for 9
  //      MFMAs.  Now a wave owns a PAIR of co-fragments of one ci-fragment and four taps (wave >> 2 picks taps 0-3 or 4-7), plus
  //      tap 8 for its first co-fragment: still 9 MFMAs and 9 accumulator tiles, but every input fragment read feeds two MFMAs --
  //      4 + 10 = 14 reads per step instead of 20 (the LDS pipe moved 327 KB per panel per CU; now 229 KB).
  const int tg = wave >> 2, cif = (wave >> 1) & 1, cp = wave & 1;
  const int cof0 = 2 * cp + tg, cof1 = 2 * cp + (1 - tg);   // (my first co-fragment is the one whose tap 8 is mine)
  const bool do_bias = p.bias_part != nullptr && ci_blk == 0 && dt == 0 && cif == 0;   // (wave-uniform)
  v8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (T)1.0f;
  // ---- fragment addresses.  A transpose read: the 16 lanes of a group give the 8-byte chunks of a [4 rows][16 columns] block in
  //      row-major chunk order (lane i: row i / 4, columns 4 (i % 4) .. +3) and lane i receives column i of the four rows.  Group g of
  //      an operand: 16-channel half g & 1 of the wave's 32-channel fragment, k half g >> 1 (the MFMA's lanes 32-63 carry k 8..15).
  const int li = lane & 15, lg = lane >> 4;
  const int rsub = li >> 2;                          // row of my chunk inside the 4-row block
  const int khalf8 = (lg >> 1) * 8;
  // gy^T: row (k16 step * 16 + khalf8 + 4 ksub + rsub): row & 3 = rsub, so the swizzle is a lane constant
  auto ga_of = [&](int cof) __attribute__((always_inline)) -> unsigned {
    const int c8 = (cof * 32 + (lg & 1) * 16) / 4 + (li & 3);       // 8-byte chunk inside the 256-byte row
    return (unsigned)(XS_B + (khalf8 + rsub) * 256 + (((c8 >> 1) ^ (rsub << 2)) * 16) + (c8 & 1) * 8);
  };
  const unsigned ga[2] = {ga_of(cof0), ga_of(cof1)};
  // a: row (k sW + dx) of kernel row dy; per kW tap the row's swizzle bit differs per lane
  const int xb_c8 = (cif * 32 + (lg & 1) * 16) / 4 + (li & 3);
  auto xb_of = [&](int t) __attribute__((always_inline)) -> unsigned {
    const int dy = t / 3, dx = t - 3 * dy;
    const int r = (khalf8 + rsub) * SW + dx;         // (the rest of the row index is a multiple of 4: bit 1 of the row is bit 1 of r)
    return (unsigned)(dy * (XRP * 128) + r * 128 + (((xb_c8 >> 1) ^ (((r >> 1) & 1) << 2)) * 16) + (xb_c8 & 1) * 8);
  };
  constexpr int NTW = 5;                              // taps a wave reads: 4 tg .. 4 tg + 3, and 8
  unsigned xb[NTW];
#pragma unroll
  for (int j = 0; j < 4; ++j) xb[j] = xb_of(4 * tg + j);
  xb[4] = xb_of(8);
  auto tr8 = [&](unsigned off, int imm0, int imm1) __attribute__((always_inline)) -> v8 {
    if (CVVAE_WGRAD_ABLATE & 1) {
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      s16x8 v = (short)(off + imm0);
      asm volatile("" : "+v"(v));
      return __builtin_bit_cast(v8, v);
    }
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lrd_t)(smem + off + imm0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lrd_t)(smem + off + imm1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(v8, v);
  };

  // ---- pipeline: NBUF - 1 panels in flight ahead of the one being multiplied.  (The buffer offsets are rotated as scalars and the
  //      panel loop is not unrolled: with `pi % 3` hipcc unrolled it three times and spilled 51 registers.)
#pragma unroll
  for (int j = 0; j < NBUF - 1; ++j)
    if (j < npanels) issue((unsigned)(j * BUF_B));
  unsigned cur_off = 0, fill_off = (unsigned)((NBUF - 1) * BUF_B);
#pragma unroll 1
  for (long long pi = 0; pi < npanels; ++pi) {
    // panel pi has landed once at most the loads of the panels issued after it are outstanding
    if (NBUF == 3 && pi + 1 < npanels) {
      if (wave < 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLW_LO + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLW_LO) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // ... for every wave; and panel pi - 1's buffer is free
    // The barrier puts every wave at the same point, and a panel's requests are ~300 scalar / vector instructions: issued by all
    // waves at once they leave the matrix pipe idle meanwhile (the first build: 0.39 busy, waves 0.5 of their cycles waiting).  The two
    // waves of a SIMD therefore run opposite orders -- waves 0-3 request, then multiply; waves 4-7 multiply, then request -- so that
    // one's requests issue under the other's MFMAs.  (Both orders are legal: the buffer being filled was last read by panel pi - 1,
    // which every wave finished before this barrier; the late requests still have a whole panel to land.)
    const bool request_first = NBUF == 2 || wave < 4 || !CVVAE_WGRAD_STAGGER;
    const bool more = pi + NBUF - 1 < npanels;
    if (more && request_first) issue(fill_off);
    const unsigned lb = cur_off, fill_off_now = fill_off;
    cur_off = cur_off + BUF_B == (unsigned)(NBUF * BUF_B) ? 0u : cur_off + BUF_B;
    fill_off = fill_off + BUF_B == (unsigned)(NBUF * BUF_B) ? 0u : fill_off + BUF_B;
    asm volatile("" : "+s"(cur_off), "+s"(fill_off));
    // Input fragments are requested DEPTH taps (2 DEPTH MFMAs) ahead of their first MFMA, the next step's two gy^T fragments during
    // the current step; the fences keep hipcc from hoisting more (left alone it requested a whole step's fragments and spilled).
    constexpr int DEPTH = CVVAE_WGRAD_DEPTH < 6 ? CVVAE_WGRAD_DEPTH : 5, NS = KP / 16, NU = NS * NTW;
    auto b_of = [&](int u) __attribute__((always_inline)) -> v8 {
      const int st = u / NTW, j = u % NTW;
      return tr8(lb + xb[j], (st * 16) * SW * 128, (st * 16 + 4) * SW * 128);
    };
    auto a_of = [&](int st, int c) __attribute__((always_inline)) -> v8 { return tr8(lb + ga[c], (st * 16) * 256, (st * 16 + 4) * 256); };
    v8 aq[2][2], bq[DEPTH];
    aq[0][0] = a_of(0, 0);
    aq[0][1] = a_of(0, 1);
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) bq[u] = b_of(u);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int st = u / NTW, j = u % NTW;
      const v8 bcur = bq[u % DEPTH];
      if (u + DEPTH < NU) bq[u % DEPTH] = b_of(u + DEPTH);
      if (st + 1 < NS && (j == 1 || j == 2)) aq[(st + 1) & 1][j - 1] = a_of(st + 1, j - 1);
      if (CVVAE_WGRAD_ABLATE & 8) asm volatile("" ::"v"(aq[st & 1][0]), "v"(bcur));
      else acc[2 * j] = Tr<T>::mfma(aq[st & 1][0], bcur, acc[2 * j]);       // (tap 8: j = 4 -> acc[8])
      __builtin_amdgcn_sched_barrier(0);
      if (j == 0 && do_bias) {
        accb = Tr<T>::mfma(aq[st & 1][0], ones, accb);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (j < 4) {
        if (CVVAE_WGRAD_ABLATE & 8) asm volatile("" ::"v"(aq[st & 1][1]), "v"(bcur));
        else acc[2 * j + 1] = Tr<T>::mfma(aq[st & 1][1], bcur, acc[2 * j + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more && !request_first) issue(fill_off_now);
  }
  const int ntaps = p.kT * NSP;
#pragma unroll
  for (int a = 0; a < NSP; ++a) {
    const int t = a == 8 ? 8 : 4 * tg + (a >> 1), cof = (a == 8 || !(a & 1)) ? cof0 : cof1;
    float* o = p.part + (((((long long)slab * (p.Coutp / CO) + co_blk) * p.n_ci_blk + ci_blk) * CO + cof * 32) * ntaps + dt * NSP + t) * CI +
               cif * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) o[(long long)(8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)) * ntaps * CI] = acc[a][i];
  }
  if (do_bias && (lane & 31) == 0) {   // (every column of the 32 x 32 product holds the same sums)
    float* o = p.bias_part + (long long)slab * p.Coutp + co0 + cof0 * 32;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)] = accb[i];
  }
}

// dW[co][ci][tap] = sum over slabs (index order) of the partial tiles.  part[slab][co block][ci block][128 co][tap][64 ci]: for one
// output channel and one block of 64 input channels, all taps are one contiguous run of ntaps x 256 bytes -- in the partial buffer AND
// (transposed: [64 ci][taps]) in the [Cout][Cin][taps] result.  One block per (co, ci block): 16-byte coalesced slab reads, the sums
// transposed through LDS, one contiguous store.  (Until round 5 a thread stored its four sums as four dwords 4 x ntaps bytes apart:
// 70 launches = 3.4 ms of a training step for 8 GB of partial tiles, profiles/r5_train_step_kernel_breakdown.txt.)
constexpr int WGRAD_RED_MAXTAPS = 27;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nslab, int ntaps, int Coutp, int Cinp,
                                                           int Cout, int Cin, float* __restrict__ dw,
                                                           const float* __restrict__ bias_part, float* __restrict__ db) {
  __shared__ float sm[64 * (WGRAD_RED_MAXTAPS + 1)];
  const int n_ci = Cinp >> 6;
  if ((int)blockIdx.x >= Coutp * n_ci) {   // the blocks behind the weight blocks: db[co] = sum over slabs of the fused bias sums
    const int co = ((int)blockIdx.x - Coutp * n_ci) * 256 + (int)threadIdx.x;
    if (co < Cout) {
      float acc = 0.f;
      for (int sl = 0; sl < nslab; ++sl) acc += bias_part[(long long)sl * Coutp + co];
      db[co] = acc;
    }
    return;
  }
  const int ci_blk = blockIdx.x % n_ci, co = blockIdx.x / n_ci;       // co = co block * 128 + row
  if (co >= Cout) return;
  const int ci0 = ci_blk * 64;
  const long long slab_stride4 = (long long)Coutp * Cinp * ntaps / 4;  // float4 units
  const float4* __restrict__ src = reinterpret_cast<const float4*>(part) +
                                   ((((long long)(co >> 7) * n_ci + ci_blk) * 128 + (co & 127)) * ntaps) * 16;
  const int n4 = ntaps * 16, pitch = ntaps + 1;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const float4* s = src + i;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    int sl = 0;
    for (; sl + 8 <= nslab; sl += 8) {  // eight loads in flight, summed in slab order (deterministic)
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s[(sl + u) * slab_stride4];
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; sl < nslab; ++sl) {
      const float4 v = s[sl * slab_stride4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const int tap = i >> 4, c = (i & 15) * 4;
    sm[(c + 0) * pitch + tap] = acc.x;
    sm[(c + 1) * pitch + tap] = acc.y;
    sm[(c + 2) * pitch + tap] = acc.z;
    sm[(c + 3) * pitch + tap] = acc.w;
  }
  __syncthreads();
  const int nci = Cin - ci0 < 64 ? Cin - ci0 : 64;                    // stored input channels of this block
  float* o = dw + ((long long)co * Cin + ci0) * ntaps;
  for (int e = threadIdx.x; e < nci * ntaps; e += 256) o[e] = sm[(e / ntaps) * pitch + e % ntaps];
}

// does the weight-gradient launch of this convolution also produce the bias gradient?  (wgrad_dma_kernel: 16-bit operands, 3x3 taps)
static bool wgrad_fuses_bias(const cvvae_conv_desc* d) {
  static const bool dma_off = getenv("CVVAE_WGRAD_DMA") && atoi(getenv("CVVAE_WGRAD_DMA")) == 0;
  return !dma_off && d->dtype < CVVAE_F32 && d->kH == 3;
}

static void wgrad_plan(const cvvae_conv_desc* d, int& nslab, int& n_co, int& n_ci) {
  n_co = (d->Cout + 127) / 128;
  n_ci = (d->Cin + 63) / 64;
  const bool xp = d->dtype >= CVVAE_F32;
  const int kp = d->kH == 1 ? (xp ? 64 : 128) : (xp ? 32 : 64);      // the kernel's K panel (wgrad_kernel)
  const long long rows = (long long)d->B * d->To * d->Ho * ((d->Wo + kp - 1) / kp);  // panels in all
  const long long per_slab = (long long)n_co * n_ci * d->kT;
  static const int target = getenv("CVVAE_WGRAD_WGS") ? atoi(getenv("CVVAE_WGRAD_WGS")) : 0;   // (tuning aid, read once)
  static const bool dma_off = getenv("CVVAE_WGRAD_DMA") && atoi(getenv("CVVAE_WGRAD_DMA")) == 0;
  long long s = (768 + per_slab - 1) / per_slab;                      // ~3 workgroups per CU in all
  if (target > 0) s = target >= 512 ? (target + per_slab - 1) / per_slab : target / per_slab;
  else if (!xp && d->kH == 3 && !dma_off) {
    // wgrad_dma_kernel: ONE workgroup per CU at a time (its LDS), the workgroups of a slab share an XCD (slab % 8), and every
    // workgroup pays a pipeline fill and a 295 KB partial-tile store the CU waits for -- all of them at once, against HBM.  So the
    // fewest ROUNDS r of 32 workgroups per XCD that fill >= 90 % of the r x 32 places with whole slabs: 128 -> 128 (6 members per
    // slab): 5 slabs per XCD in one round; 2 members: 16 in one; 24 or 96 members: three rounds (the 768 of the line above).
    // Measured (profiles/r5_ab_wgrad_rounds.log): one round instead of three is -11 % on the per-frame 128-channel layers, -24 % on
    // the strided ones, even on 128 -> 128 3x3x3 -- and a third of the partial tiles to write and to reduce.
    for (int r = 1; r <= 3; ++r) {
      const long long spx = 32ll * r / per_slab;
      if (spx >= 1 && spx * per_slab * 10 >= 9 * 32ll * r) {
        s = spx * 8;
        break;
      }
    }
  }
  const long long tile_bytes = (long long)d->kT * d->kH * d->kW * n_co * 128 * n_ci * 64 * 4;
  const long long cap = (512ll << 20) / (tile_bytes > 0 ? tile_bytes : 1);  // <= 512 MB of partials
  if (s > cap) s = cap;
  if (s > rows) s = rows;
  if (s < 1) s = 1;
  nslab = (int)s;
}

static long long wgrad_partial_bytes(const cvvae_conv_desc* d, int nslab, int n_co, int n_ci) {
  const long long b = (long long)nslab * d->kT * d->kH * d->kW * n_co * 128 * n_ci * 64 * (long long)sizeof(float);
  return (b + 255) / 256 * 256;
}

template <typename T, int KHW, bool XP, int SW>
static int wgrad_launch_sw(const cvvae_conv_desc* d, const void* a, const void* gy, int64_t g_ps, float* dw, float* db, void* ws, hipStream_t s) {
  int nslab, n_co, n_ci;
  wgrad_plan(d, nslab, n_co, n_ci);
  WgradArgs p{};
  p.a = a; p.g = gy; p.part = (float*)ws; p.bias_part = nullptr;
  p.B = d->B; p.Ti = d->Ti; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.a_ps = d->in_pix_stride;
  p.To = d->To; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.g_ps = g_ps;
  p.kT = d->kT; p.sT = d->sT; p.sH = d->sH; p.sW = d->sW; p.pt = d->pad_t; p.ph = d->pad_h; p.pw = d->pad_w;
  p.mode_t = d->pad_mode_t; p.mode_hw = d->pad_mode_hw;
  p.nslab = nslab; p.rows_total = d->B * d->To * d->Ho; p.n_ci_blk = n_ci;
  p.Coutp = n_co * 128; p.Cinp = n_ci * 64;
  p.n_coci = n_co * n_ci;
  p.n_members = n_co * n_ci * d->kT;
  const int nslab8 = (nslab + 7) / 8 * 8;  // (blocks of the padding slabs exit at once)
  int rc = 0;
  if (db && !wgrad_fuses_bias(d)) return CVVAE_EUNSUPPORTED;
  // CVVAE_WGRAD_DMA=0: the register-staged kernel for every layer (A/B aid; read once)
  static const bool dma_off = getenv("CVVAE_WGRAD_DMA") && atoi(getenv("CVVAE_WGRAD_DMA")) == 0;
  if constexpr (KHW == 3 && !XP) {
    if (!dma_off) {
      // the zero page behind the partial tiles (cvvae_conv_wgrad_workspace_bytes reserves it)
      char* zero = reinterpret_cast<char*>(ws) + wgrad_partial_bytes(d, nslab, n_co, n_ci);
      if (db) p.bias_part = reinterpret_cast<float*>(zero + 512);   // [nslab][Coutp] behind the zero page
      // the scalar-base form of the wave-loads (FAST, see the kernel): whole gradient panels, every channel of the tiles stored, the
      // input tensor and the zero page inside one 4 GB window
      const unsigned long long a0 = (unsigned long long)(size_t)a, z0 = (unsigned long long)(size_t)zero;
      const unsigned long long a1 = a0 + (unsigned long long)d->B * d->Ti * d->Hi * d->Wi * d->in_pix_stride * sizeof(T);
      const bool needs_zero = d->pad_mode_hw == 0 || d->pad_mode_t == 0;   // (else the zero page is never read: see the kernel)
      const unsigned long long lo = !needs_zero || a0 < z0 ? a0 : z0, hi = !needs_zero || a1 > z0 + 512 ? a1 : z0 + 512;
      const char* fenv = getenv("CVVAE_WGRAD_FAST");   // (read per call: tests compare the two address forms inside one process)
      const bool fast_off = fenv && atoi(fenv) == 0;
      const bool fast = !fast_off && d->Wo % 64 == 0 && d->in_pix_stride >= (long long)n_ci * 64 && g_ps >= (long long)n_co * 128 &&
                        hi - lo < (1ull << 32) && (long long)d->Wo * g_ps * (long long)sizeof(T) < (1ll << 31);
      // (a form that keeps the input rows in LDS across consecutive output rows -- column-major panels, one new row per panel, 25.6 instead
      //  of 44 KB of wave-loads -- was built and measured 1-3 % slower: commit aa16cd7, profiles/r5_ab_wgrad_rounds.log)
      if (fast) hipLaunchKernelGGL((wgrad_dma_kernel<T, SW, true>), dim3((unsigned)(nslab8 * p.n_members)), dim3(512), 0, s, p, (void*)zero);
      else hipLaunchKernelGGL((wgrad_dma_kernel<T, SW, false>), dim3((unsigned)(nslab8 * p.n_members)), dim3(512), 0, s, p, (void*)zero);
    } else {
      hipLaunchKernelGGL((wgrad_kernel<T, KHW, XP, SW>), dim3((unsigned)(nslab8 * p.n_members)), dim3(512), 0, s, p);
    }
  } else {
    hipLaunchKernelGGL((wgrad_kernel<T, KHW, XP, SW>), dim3((unsigned)(nslab8 * p.n_members)), dim3(512), 0, s, p);
  }
  rc = (int)hipGetLastError();
  if (rc) return rc;
  const int ntaps = d->kT * d->kH * d->kW;
  if (ntaps > WGRAD_RED_MAXTAPS) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(p.Coutp * n_ci + (p.bias_part ? (d->Cout + 255) / 256 : 0))), dim3(256), 0, s,
                     (const float*)ws, nslab, ntaps, p.Coutp, p.Cinp, d->Cout, d->Cin, dw, (const float*)p.bias_part, db);
  return (int)hipGetLastError();
}

template <typename T, int KHW, bool XP>
static int wgrad_launch(const cvvae_conv_desc* d, const void* a, const void* gy, int64_t g_ps, float* dw, float* db, void* ws, hipStream_t s) {
  return d->sW == 2 ? wgrad_launch_sw<T, KHW, XP, 2>(d, a, gy, g_ps, dw, db, ws, s) : wgrad_launch_sw<T, KHW, XP, 1>(d, a, gy, g_ps, dw, db, ws, s);
}

// ---------------------------------------------------------------------------------------------------------
// per-channel sums over pixels: bias gradients (x = null: sum g) and GroupNorm affine gradients
//   d beta[c] = sum g * act'(a),  d gamma[c] = sum g * act'(a) * xh      (xh, a as in gn_bwd_*: misc_kernels.hip)
// Two passes: per (row, split) partial sums [C][2] written in place, then summed in index order.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_grad2_f(float a) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-a));
  return sg * (1.0f + a * (1.0f - sg));
}

template <typename T>
__global__ __launch_bounds__(256) void chan_sums_kernel(const T* __restrict__ x, const T* __restrict__ g, long long S, int C,
                                                        long long g_ps, int nsplit, const float* __restrict__ rs,
                                                        const float* __restrict__ nm, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int silu, float* __restrict__ ws) {
  const int split = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
  const int cv = (C + 7) >> 3;
  const int ppp = 256 / cv > 0 ? 256 / cv : 1;
  const int myv = tid % cv, mypl = tid / cv;
  const long long per = (S + nsplit - 1) / nsplit;
  const long long p0 = (long long)split * per;
  long long p1 = p0 + per;
  if (p1 > S) p1 = S;
  float s1[8], s2[8], trs[8], tnm[8], tga[8], tbe[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s1[j] = s2[j] = 0.f;
    const int c = myv * 8 + j;
    const bool ok = x != nullptr && c < C;
    trs[j] = ok ? rs[(long long)row * C + c] : 0.f;
    tnm[j] = ok ? nm[(long long)row * C + c] : 0.f;
    tga[j] = ok ? gamma[c] : 0.f;
    tbe[j] = ok ? beta[c] : 0.f;
  }
  if (mypl < ppp && myv * 8 < C) {
    // four pixels per trip: their 16-byte loads are issued together (one pixel per trip left ~4 MB in flight on the chip --
    // latency bound at ~2 TB/s); lanes past the end re-read pixel px with a zeroed gradient
    constexpr int U = 4;
    for (long long px = p0 + mypl; px < p1; px += U * ppp) {
      float f[U][8], gg[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = px + u * ppp < p1;
        const long long q = ok ? px + u * ppp : px;
        ld8<T>(g + ((long long)row * S + q) * g_ps + myv * 8, gg[u]);
        if (x != nullptr) ld8<T>(x + ((long long)row * S + q) * (long long)C + myv * 8, f[u]);
        if (!ok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) gg[u][j] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (x != nullptr) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xh = __builtin_fmaf(f[u][j], trs[j], tnm[j]);
            const float a = __builtin_fmaf(xh, tga[j], tbe[j]);
            const float ga = gg[u][j] * (silu ? silu_grad2_f(a) : 1.0f);
            s1[j] += ga;
            s2[j] += ga * xh;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) s1[j] += gg[u][j];
        }
      }
    }
  }
  __shared__ float sh[256][17];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sh[tid][j] = s1[j];
    sh[tid][8 + j] = s2[j];
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {  // channel c: its vector c/8, every pixel lane, in index order
    float a1 = 0.f, a2 = 0.f;
    for (int pl = 0; pl < ppp; ++pl) {
      const int th = pl * cv + (c >> 3);
      if (th < 256) {
        a1 += sh[th][c & 7];
        a2 += sh[th][8 + (c & 7)];
      }
    }
    float* o = ws + (((long long)row * nsplit + split) * C + c) * 2;
    o[0] = a1;
    o[1] = a2;
  }
}

__global__ __launch_bounds__(256) void chan_sums_final_kernel(const float* __restrict__ ws, int nparts, int C, float* __restrict__ o1,
                                                              float* __restrict__ o2) {
  // a block = 32 channels x 8 part lanes: lane pl sums parts pl, pl + 8, ... (coalesced 256-byte rows), then the 8 lanes are
  // added in index order -- a fixed summation order, hence bit-reproducible
  __shared__ float sh[8][32][2];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a1 = 0.f, a2 = 0.f;
  if (c < C) {
    int i = pl;
    for (; i + 24 < nparts; i += 32) {  // four loads in flight, added in index order
      float2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float2*>(ws + ((long long)(i + 8 * u) * C + c) * 2);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a1 += v[u].x; a2 += v[u].y; }
    }
    for (; i < nparts; i += 8) {
      const float2 v = *reinterpret_cast<const float2*>(ws + ((long long)i * C + c) * 2);
      a1 += v.x;
      a2 += v.y;
    }
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
    if (o1) o1[c] = s1;
    if (o2) o2[c] = s2;
  }
}

static inline int chan_sums_splits(long long S, int rows, int C) {
  // ~32 pixels per thread (8 trips of 4): a workgroup covers 256 / (C / 8) pixels per step, so wide tensors take short splits
  // (at a fixed 1024 pixels per split a 512-channel tensor gave each thread 256 dependent trips and a 147 k-pixel one 144
  // workgroups for 256 CUs)
  const int cv = (C + 7) >> 3;
  const int ppp = 256 / cv > 0 ? 256 / cv : 1;
  long long per = 32ll * ppp;
  if (per < 128) per = 128;
  long long n = (S + per - 1) / per;
  const long long cap = rows > 0 ? (2048 + rows - 1) / rows : 2048;
  if (n > cap) n = cap;
  return (int)(n < 1 ? 1 : n);
}

// ---------------------------------------------------------------------------------------------------------
// adjoint of replicate / zero padding: gp [B][Tp][Hp][Wp][C] is the gradient w.r.t. the PADDED input (front pads pt, ph, pw);
// out[b][t][y][x] = sum of gp over every padded position that the forward's coordinate map sends to (t, y, x)
// (replicate: clamp -- border elements collect their whole pad region; zero: only the interior copy).  `add` is summed in.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pad_fold_kernel(const T* __restrict__ gp, int B, int T_, int H, int W, int C, int Tp, int Hp,
                                                       int Wp, int pt, int ph, int pw, int mode_t, int mode_hw,
                                                       const T* __restrict__ add, T* __restrict__ out) {
  const int cv = C >> 3;
  const long long nvec = (long long)B * T_ * H * W * cv;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
    const int c0 = (int)(v % cv) * 8;
    long long pix = v / cv;
    const int x = (int)(pix % W);
    pix /= W;
    const int y = (int)(pix % H);
    pix /= H;
    const int t = (int)(pix % T_);
    const int b = (int)(pix / T_);
    // padded index range [lo, hi] that maps to coordinate c of an axis of length L
    auto range = [](int c, int L, int pf, int Lp, int mode, int& lo, int& hi) {
      lo = hi = c + pf;
      if (mode) {
        if (c == 0) lo = 0;
        if (c == L - 1) hi = Lp - 1;
      }
    };
    int t0, t1, y0, y1, x0, x1;
    range(t, T_, pt, Tp, mode_t, t0, t1);
    range(y, H, ph, Hp, mode_hw, y0, y1);
    range(x, W, pw, Wp, mode_hw, x0, x1);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int tp = t0; tp <= t1; ++tp)
      for (int yp = y0; yp <= y1; ++yp)
        for (int xp = x0; xp <= x1; ++xp) {
          float f[8];
          ld8<T>(gp + ((((long long)b * Tp + tp) * Hp + yp) * Wp + xp) * C + c0, f);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
    const long long off = ((((long long)b * T_ + t) * H + y) * W + x) * C + c0;
    if (add) {
      float f[8];
      ld8<T>(add + off, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
    st8<T>(out + off, acc);
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward of cvvae_temporal_attention (MemoryEfficientAttnVideoBlock.attention_t, models/vae_models.py:573-587): per pixel,
// softmax(q k^T * scale) v over the Tn <= 8 frames of a clip.  One wave per pixel, as the forward: pass 1 over the channels forms
// S = q k^T and dP = go v^T (two 8 x 8 tables per lane, reduced over the wave), the softmax and its gradient
// dS = scale * P o (dP - rowsum(P o dP)) are per lane; pass 2 writes dq = dS k, dk = dS^T q, dv = P^T go.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void temporal_attn_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                const T* __restrict__ v, const T* __restrict__ go, long long P,
                                                                int Tn, long long S, int C, float scale, T* __restrict__ gq,
                                                                T* __restrict__ gk, T* __restrict__ gv) {
  const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // over B*S
  if (pix >= P) return;
  const int lane = threadIdx.x & 63;
  const int nv = C >> 3;
  const long long b = pix / S, s = pix - b * S;
  const long long base = (b * Tn * S + s) * C, tstride = S * C;
  float sc[8][8], dp[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[i][j] = dp[i][j] = 0.f;
  for (int vv = lane; vv < nv; vv += 64) {
    float qf[8][8], gf[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < Tn) {
        ld8<T>(q + base + i * tstride + vv * 8, qf[i]);
        ld8<T>(go + base + i * tstride + vv * 8, gf[i]);
      }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < Tn) {
        float kf[8], vf[8];
        ld8<T>(k + base + j * tstride + vv * 8, kf);
        ld8<T>(v + base + j * tstride + vv * 8, vf);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < Tn) {
            float d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              d1 += qf[i][e] * kf[e];
              d2 += gf[i][e] * vf[e];
            }
            sc[i][j] += d1;
            dp[i][j] += d2;
          }
      }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = sc[i][j], y = dp[i][j];
#pragma unroll
      for (int off = 32; off; off >>= 1) {
        x += __shfl_xor(x, off);
        y += __shfl_xor(y, off);
      }
      sc[i][j] = x * scale;
      dp[i][j] = y;
    }
  // P (as the forward computes it), then dS in place of dp
#pragma unroll
  for (int i = 0; i < 8; ++i)
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
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sc[i][j] = (j < Tn) ? sc[i][j] * inv : 0.f;
        dot += sc[i][j] * dp[i][j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) dp[i][j] = (j < Tn) ? scale * sc[i][j] * (dp[i][j] - dot) : 0.f;
    }
  for (int vv = lane; vv < nv; vv += 64) {
    float qf[8][8], gf[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < Tn) {
        ld8<T>(q + base + i * tstride + vv * 8, qf[i]);
        ld8<T>(go + base + i * tstride + vv * 8, gf[i]);
      }
    float dq[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) dq[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < Tn) {
        float kf[8], dk[8], dv[8];
        ld8<T>(k + base + j * tstride + vv * 8, kf);
#pragma unroll
        for (int e = 0; e < 8; ++e) dk[e] = dv[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < Tn) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              dq[i][e] += dp[i][j] * kf[e];
              dk[e] += dp[i][j] * qf[i][e];
              dv[e] += sc[i][j] * gf[i][e];
            }
          }
        st8<T>(gk + base + j * tstride + vv * 8, dk);
        st8<T>(gv + base + j * tstride + vv * 8, dv);
      }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < Tn) st8<T>(gq + base + i * tstride + vv * 8, dq[i]);
  }
}

}  // namespace cvvae

using namespace cvvae;

extern "C" {

int64_t cvvae_conv_wgrad_workspace_bytes(const cvvae_conv_desc* d) {
  if (!d || d->Cout <= 0 || d->Cin <= 0 || d->kT <= 0) return CVVAE_EINVAL;
  int nslab, n_co, n_ci;
  wgrad_plan(d, nslab, n_co, n_ci);
  // + the zero page of wgrad_dma_kernel + its per-slab bias sums
  return (int64_t)wgrad_partial_bytes(d, nslab, n_co, n_ci) + 512 + ((int64_t)nslab * n_co * 128 * 4 + 255) / 256 * 256;
}

int cvvae_conv_wgrad_fuses_bias(const cvvae_conv_desc* d) {
  if (!d) return CVVAE_EINVAL;
  return wgrad_fuses_bias(d) ? 1 : 0;
}

int cvvae_conv_wgrad(const cvvae_conv_desc* d, const void* a, const void* gy, int64_t gy_pix_stride, float* dw, void* workspace,
                     void* stream_) {
  return cvvae_conv_wgrad_bias(d, a, gy, gy_pix_stride, dw, nullptr, workspace, stream_);
}

int cvvae_conv_wgrad_bias(const cvvae_conv_desc* d, const void* a, const void* gy, int64_t gy_pix_stride, float* dw, float* dbias,
                          void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d || !a || !gy || !dw || !workspace) return CVVAE_EINVAL;
  if (d->B <= 0 || d->Ti <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->To <= 0 || d->Ho <= 0 || d->Wo <= 0) return CVVAE_EINVAL;
  if (d->Cin <= 0 || d->Cin % 8 || d->in_pix_stride < d->Cin || d->in_pix_stride % 8) return CVVAE_EINVAL;
  if (d->Cout <= 0 || d->Cout % 8 || gy_pix_stride < d->Cout || gy_pix_stride % 8) return CVVAE_EINVAL;
  if (d->upsample2x || d->in_overlap) return CVVAE_EUNSUPPORTED;
  if (!(d->kH == d->kW && (d->kH == 3 || d->kH == 1) && (d->kT == 3 || d->kT == 1))) return CVVAE_EUNSUPPORTED;
  if (d->sT < 1 || d->sT > 2 || d->sH < 1 || d->sH > 2 || d->sW < 1 || d->sW > 2 || d->sH != d->sW) return CVVAE_EUNSUPPORTED;
  const bool k3 = d->kH == 3;
  switch (d->dtype) {
    case CVVAE_BF16:
      return k3 ? wgrad_launch<__bf16, 3, false>(d, a, gy, gy_pix_stride, dw, dbias, workspace, stream)
                : wgrad_launch<__bf16, 1, false>(d, a, gy, gy_pix_stride, dw, dbias, workspace, stream);
    case CVVAE_F16:
      return k3 ? wgrad_launch<_Float16, 3, false>(d, a, gy, gy_pix_stride, dw, dbias, workspace, stream)
                : wgrad_launch<_Float16, 1, false>(d, a, gy, gy_pix_stride, dw, dbias, workspace, stream);
    case CVVAE_F32:
    case CVVAE_F32Q:
    case CVVAE_F32Q6:
      return k3 ? wgrad_launch<__bf16, 3, true>(d, a, gy, gy_pix_stride, dw, dbias, workspace, stream)
                : wgrad_launch<__bf16, 1, true>(d, a, gy, gy_pix_stride, dw, dbias, workspace, stream);
    default:
      return CVVAE_EINVAL;
  }
}

int64_t cvvae_channel_sums_workspace_bytes(int32_t rows, int64_t S, int32_t C) {
  if (rows <= 0 || S <= 0 || C <= 0) return CVVAE_EINVAL;
  return (int64_t)rows * chan_sums_splits(S, rows, C) * C * 2 * (int64_t)sizeof(float);
}

// x == NULL: sum1[c] = sum over rows x S pixels of g (a bias gradient); else the GroupNorm affine gradients
// sum1 = d beta, sum2 = d gamma with (rstd, -mean*rstd) tables [rows][C] as in cvvae_gn_bwd_input
int cvvae_channel_sums(int32_t dtype, const void* x, const void* g, int64_t g_pix_stride, int32_t rows, int64_t S, int32_t C,
                       const float* rstd, const float* nmean, const float* gamma, const float* beta, int32_t silu, float* sum1,
                       float* sum2, void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!g || !sum1 || !workspace || rows <= 0 || S <= 0 || C <= 0 || C % 8 || C > 2048 || g_pix_stride < C || g_pix_stride % 8)
    return CVVAE_EINVAL;
  if (x && (!rstd || !nmean || !gamma || !beta || !sum2 || g_pix_stride != C)) return CVVAE_EINVAL;
  if ((C >> 3) > 256) return CVVAE_EUNSUPPORTED;
  const int nsplit = chan_sums_splits(S, rows, C);
  float* ws = (float*)workspace;
  switch (dtype) {
    case CVVAE_BF16:
      hipLaunchKernelGGL(chan_sums_kernel<__bf16>, dim3(nsplit, rows), dim3(256), 0, stream, (const __bf16*)x, (const __bf16*)g, S, C,
                         g_pix_stride, nsplit, rstd, nmean, gamma, beta, silu, ws);
      break;
    case CVVAE_F16:
      hipLaunchKernelGGL(chan_sums_kernel<_Float16>, dim3(nsplit, rows), dim3(256), 0, stream, (const _Float16*)x, (const _Float16*)g,
                         S, C, g_pix_stride, nsplit, rstd, nmean, gamma, beta, silu, ws);
      break;
    case CVVAE_F32:
      hipLaunchKernelGGL(chan_sums_kernel<float>, dim3(nsplit, rows), dim3(256), 0, stream, (const float*)x, (const float*)g, S, C,
                         g_pix_stride, nsplit, rstd, nmean, gamma, beta, silu, ws);
      break;
    default:
      return CVVAE_EINVAL;
  }
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  hipLaunchKernelGGL(chan_sums_final_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, (const float*)ws, rows * nsplit, C, sum1,
                     sum2);
  return (int)hipGetLastError();
}

int cvvae_pad_fold(int32_t dtype, const void* gp, int32_t B, int32_t T, int32_t H, int32_t W, int32_t C, int32_t pad_t_front,
                   int32_t pad_t_back, int32_t pad_h, int32_t pad_w, int32_t pad_mode_t, int32_t pad_mode_hw, const void* add, void* out,
                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!gp || !out || B <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return CVVAE_EINVAL;
  if (pad_t_front < 0 || pad_t_back < 0 || pad_h < 0 || pad_w < 0) return CVVAE_EINVAL;
  const int Tp = T + pad_t_front + pad_t_back, Hp = H + 2 * pad_h, Wp = W + 2 * pad_w;
  long long blocks = ((long long)B * T * H * W * (C / 8) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
#define CVVAE_FOLD(TY)                                                                                                           \
  hipLaunchKernelGGL(pad_fold_kernel<TY>, dim3((unsigned)blocks), dim3(256), 0, stream, (const TY*)gp, B, T, H, W, C, Tp, Hp, Wp, \
                     pad_t_front, pad_h, pad_w, pad_mode_t, pad_mode_hw, (const TY*)add, (TY*)out)
  switch (dtype) {
    case CVVAE_BF16: CVVAE_FOLD(__bf16); break;
    case CVVAE_F16: CVVAE_FOLD(_Float16); break;
    case CVVAE_F32: CVVAE_FOLD(float); break;
    default: return CVVAE_EINVAL;
  }
#undef CVVAE_FOLD
  return (int)hipGetLastError();
}

int cvvae_temporal_attention_bwd(int32_t dtype, const void* q, const void* k, const void* v, const void* go, int32_t B, int32_t Tn,
                                 int64_t S, int32_t C, void* gq, void* gk, void* gv, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!q || !k || !v || !go || !gq || !gk || !gv || B <= 0 || Tn <= 0 || S <= 0 || C <= 0 || C % 8) return CVVAE_EINVAL;
  if (Tn > 8) return CVVAE_EUNSUPPORTED;  // (the windows of the reference's wrapper give T' <= 5; longer sequences: not built)
  const long long P = (long long)B * S;
  const float scale = 1.0f / sqrtf((float)C);
  const unsigned grid = (unsigned)((P + 3) / 4);
#define CVVAE_TAB(TY)                                                                                                          \
  hipLaunchKernelGGL(temporal_attn_bwd_kernel<TY>, dim3(grid), dim3(256), 0, stream, (const TY*)q, (const TY*)k, (const TY*)v, \
                     (const TY*)go, P, Tn, (long long)S, C, scale, (TY*)gq, (TY*)gk, (TY*)gv)
  switch (dtype) {
    case CVVAE_BF16: CVVAE_TAB(__bf16); break;
    case CVVAE_F16: CVVAE_TAB(_Float16); break;
    case CVVAE_F32: CVVAE_TAB(float); break;
    default: return CVVAE_EINVAL;
  }
#undef CVVAE_TAB
  return (int)hipGetLastError();
}

}  // extern "C"
