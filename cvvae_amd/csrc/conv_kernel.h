// conv_kernel.h -- implicit-GEMM convolution for CDNA4 (gfx950): the hot kernel of the CV-VAE codec.
//
// One workgroup = 512 threads = 8 wave64 (2 per SIMD, one workgroup per CU).  It owns an output tile of
// BM = TT*TH*TW pixels x BN = 32*WN output channels and walks K = taps x Cin in chunks of CK = 16*KSUB input channels:
//
//   * the INPUT HALO TILE of the chunk ((TT-1)*sT+kT) x ((TH-1)*sH+kH) x ((TW-1)*sW+kW) pixels x CK channels
//     is staged ONCE into LDS (register-staged: pad-mode / nearest-2x coordinate mapping, GroupNorm affine +
//     SiLU are applied on the way, so neither F.pad, F.interpolate, GroupNorm nor SiLU ever touch HBM);
//     im2col is then pure LDS addressing: tap (dt,dy,dx) is an immediate offset on the ds_read_b128.
//     Pixel stride in LDS is CK*2+16 bytes, which makes every 16-lane ds_read_b128 group conflict-free.
//   * the halo tile is DOUBLE BUFFERED and the 8 waves are split in two groups with opposite phase order
//     (X: stage next chunk -> MFMA; Y: MFMA -> stage next chunk).  One s_barrier per chunk.
//   * WEIGHTS never go through LDS: they are pre-packed in MFMA-fragment order (cvvae_pack_weights*) and each
//     wave streams its own 32-output-channel slice straight HBM/L2 -> VGPR (1 KiB per wave per k16 step,
//     contiguous, software-prefetched PF steps ahead).
//   * MFMA is v_mfma_f32_32x32x16 with SWAPPED operands (A = weights, B = activations): the accumulator
//     lane then holds, for ONE pixel, 4 consecutive output channels per register quad.  The packers store output
//     channel sigma(i) in MFMA row i (sigma swaps bits 2 and 3 of i), which makes quads 2p, 2p+1 of a lane 8
//     CONSECUTIVE channels: every lane stores 16 bytes at a time with no cross-lane exchange.
//   * WM x WN x KG = 8 waves: WM pixel slabs x WN 32-channel blocks x KG K-groups (KG = 2: the two wave groups
//     split the chunk's channels and reduce their accumulators through LDS).
//   * UPS = 1: nearest-2x gather fused into the staging; UPS = 2: the upsample folded into four 3x2x2 phase
//     convolutions over the stored input (phase = 2 extra bits of the logical tile index).
//   * the epilogue optionally emits GroupNorm statistics of what it stores (one (n, mean, M2) record per pixel tile,
//     wave slab and 4-channel slot; no atomics).
//   * logical tile order: N-tile (and phase) fastest, then time, x, y, batch; XCD-aware bijective block remap.
//
// Reference semantics implemented here (file:line in /root/reference): see include/cvvae.h.  Design notes and
// measurements: DESIGN.md section 3.1.  -DCVVAE_CONV_PROBE adds s_memtime stamps (tools/probes/conv_probe.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "tile_map.h"

namespace cvvae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <typename T>
struct Tr;
template <>
struct Tr<__bf16> {
  using v8 = bf16x8;
  using v4 = bf16x4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Tr<_Float16> {
  using v8 = f16x8;
  using v4 = f16x4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// v_mfma_f32_32x32x64_f8f6f4 with both operands bf8 (OCP e5m2) and no block scales: K = 64 at TWICE the fp16 rate per K
// element.  An operand is 32 bytes per lane, given here as its two 16-byte halves (bytes 0-15 | 16-31 of the lane): lanes 0-31
// carry k = 0..31, lanes 32-63 k = 32..63 of row / column (lane & 31) -- the same map for A and B.
__device__ __forceinline__ f32x16 mfma_bf8_k64(f16x8 a_lo, f16x8 a_hi, f16x8 b_lo, f16x8 b_hi, f32x16 c) {
  const i32x4 al = __builtin_bit_cast(i32x4, a_lo), ah = __builtin_bit_cast(i32x4, a_hi);
  const i32x4 bl = __builtin_bit_cast(i32x4, b_lo), bh = __builtin_bit_cast(i32x4, b_hi);
  const i32x8 a = __builtin_shufflevector(al, ah, 0, 1, 2, 3, 4, 5, 6, 7);
  const i32x8 b = __builtin_shufflevector(bl, bh, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, 0, 0, 0);  // cbsz = blgp = 1: bf8; scales 0: unscaled form
}
// v_mfma_scale_f32_32x32x64_f8f6f4 with both operands "bf6" (OCP MX e3m2: 6 bits, normals 0.25 .. 28, subnormal quantum 1/16):
// K = 64 at FOUR times the fp16 rate per K element.  An operand is 32 six-bit codes per lane (24 bytes, element i at bits 6i..6i+5
// of the lane's first six dwords), same lane -> (row, k) map as above; each LANE carries one E8M0 scale byte (value 2^(byte-127),
// byte 0 of the scale operand) for its 32 elements -- the hardware's block scale (tools/probes/fp6_probe.hip, profiles/r3_fp6_probe.txt).
typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x2_t __attribute__((ext_vector_type(2)));
// (plain ext-vector types throughout: arrays of HIP's struct-based uint4 / uint2 inside the K loop's nested lambdas were not scalarised --
//  the backend promoted them to LDS and the loop ran 2.4x slower)
__device__ __forceinline__ f32x16 mfma_bf6_k64(i32x4_t a03, i32x4_t a47, i32x4_t b03, i32x2_t b45, int scale_b, f32x16 c) {
  const i32x8_t a = {a03[0], a03[1], a03[2], a03[3], a47[0], a47[1], 0, 0};
  const i32x8_t b = {b03[0], b03[1], b03[2], b03[3], b45[0], b45[1], 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 3, 3, 0, a47[2], 0, scale_b);  // cbsz = blgp = 3: bf6
}
// 16 halves (8 dwords) -> 16 bf6 codes = 96 bits, code i at bits 6i: v_cvt_scalef32_pk32_bf6_f16 (code = round-to-nearest-even of
// value / scale, saturating at +-28; only the exponent of `scale` is used) on a 32-half operand whose upper half is don't-care
typedef _Float16 f16x16_t __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32_t __attribute__((ext_vector_type(32)));
typedef unsigned u32x6_t __attribute__((ext_vector_type(6)));
__device__ __forceinline__ uint3 cvt16_bf6(uint4 h07, uint4 h8f, float scale) {
  typedef unsigned u32x8_t __attribute__((ext_vector_type(8)));
  const u32x8_t w = {h07.x, h07.y, h07.z, h07.w, h8f.x, h8f.y, h8f.z, h8f.w};
  const f16x16_t v = __builtin_bit_cast(f16x16_t, w);
  const f16x32_t v32 = __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, -1, -1, -1, -1, -1, -1, -1, -1,
                                               -1, -1, -1, -1, -1, -1, -1, -1);
  const u32x6_t r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v32, scale);
  return make_uint3(r[0], r[1], r[2]);
}
// 8 floats -> 8 bf8 (e5m2, round to nearest even), packed in two dwords
__device__ __forceinline__ uint2 pack8_bf8(const float (&f)[8]) {
  int a = 0, b = 0;
  a = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], a, false);
  a = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], a, true);
  b = __builtin_amdgcn_cvt_pk_bf8_f32(f[4], f[5], b, false);
  b = __builtin_amdgcn_cvt_pk_bf8_f32(f[6], f[7], b, true);
  return make_uint2((unsigned)a, (unsigned)b);
}

struct ConvArgs {
  const void* in;
  const void* w;
  const float* bias;
  const void* res;
  const float* gsc;
  const float* gsh;
  void* out;
  int B, Ti, Hi, Wi;  // stored input dims
  int Tl, Hl, Wl;     // logical input dims seen by the taps (Hl = 2*Hi when upsample2x)
  int Cin;
  long long in_ps;
  int To, Ho, Wo, Cout;
  long long out_ps;
  int pt, ph, pw;
  int mode_t, mode_hw;
  int tiles_t, tiles_h, tiles_w, ntiles_n;
  int nchunks;   // Cin / CK
  int nblk32;    // ceil(Cout/32): number of packed 32-channel weight blocks
  int out_mode, out_f32;
  // UPS == 2 (nearest-2x upsample folded into four 3x2x2 phase convolutions over the stored input): To/Ho/Wo are the
  // per-phase output dims (= Ti/Hi/Wi); output pixel (y, x) of phase (py, px) is stored at (2y+py, 2x+px) of a
  // 2Ho x 2Wo frame; phase weights are w + phase * w_phase_stride (elements)
  long long w_phase_stride;
  long long w_bstride;  // elements between the packed weights of consecutive batch items (0: shared)
  // fused 1x1 shortcut (1x3x3 instances): after the K loop over `in`, nchunks2 more chunks over `in2` (same B,T,H,W, no
  // prologue) through the centre tap only, with the 1x1 weights w2 -- ResnetBlock conv2 + conv_shortcut in one accumulator
  const void* in2;
  const void* w2;
  long long in2_ps;
  int nchunks2;
  // sub-range launches (an odd frame count is covered by a two-frame-tile launch over [0, To-1) plus a one-frame-tile
  // launch for the last frame): first output frame of this launch, and its first pixel-tile index inside a batch row
  int t_begin, tile_base;
  int order;     // logical tile order: 1 = (ntile, t, x, y, b) fastest-first, 0 = (ntile, x, y, t, b)
  int gn_rpb;
  float alpha;
  // fused GroupNorm statistics of the OUTPUT (NDHWC / time-shuffle modes): partial (n, mean, M2) records
  float* gnp;        // [B][gn_G][gn_slabs][3], or null
  int gn_slabs;      // records per batch row and group = pixel tiles * WM * KG * (4-channel slots per group)
  int gn_G;          // groups of the stored tensor
  int gn_sh;         // log2(channels per group), >= 2
  // tuning probe (tools/probes/conv_probe.hip, built with -DCVVAE_CONV_PROBE): s_memtime stamps of workgroup dbg_block
  unsigned long long* dbg;
  int dbg_block;
  int t_short_lo, t_short_hi;  // leading / trailing time tiles that are short (time folds): scheduled after the long ones
  int w_taps;    // taps per k16 record group of the packed weights: NTAPS, or 2*NTAPS with the time-fold slots (KT == 3)
  int res_pre;   // residual is accumulated during the K loop instead of in the store tail (needs alpha == 1, 16-bit NDHWC output)
  // XP == 3 (fp6 corrections): the activations' static power-of-two scale -- codes = value / q6_scale, MFMA scale byte q6_eb
  float q6_scale;
  int q6_eb;
  // ... or, when the bound is only known on the device (cvvae_conv_desc.act_bound_dev: an operand without a GroupNorm in front, whose
  // bound is a reduction the producer's stream computes), the kernel derives both from *q6_bound itself
  const float* q6_bound;
  int probe_nostore;  // probe builds (-DCVVAE_CONV_PROBE) only: run the store tail without its stores
  int stats_noshift;  // debug aid (CVVAE_STATS_NOSHIFT=1): fused statistics as plain sums (shift K = 0)
  int ws_window;      // UPS == 2: weight-stationary window of the tile order (tile_map.h), 0 / 1 = off
  int phase_sync;     // 1: every wave multiplies chunk c, THEN stages chunk c+1 (no wave stages beside another's MFMA stream);
                      // 0: the two wave groups run opposite phase orders (X: stage -> MFMA, Y: MFMA -> stage)
};

#ifdef CVVAE_CONV_PROBE
#define CVVAE_PROBE_MARK()                                                                     \
  do {                                                                                         \
    if (p.dbg && (int)blockIdx.x == p.dbg_block && lane == 0 && probe_n < 128)                 \
      p.dbg[wave * 128 + probe_n] = __builtin_amdgcn_s_memtime();                              \
    ++probe_n;                                                                                 \
  } while (0)
#else
#define CVVAE_PROBE_MARK() do { } while (0)
#endif

// Ablation builds (scratch libraries loaded through CVVAE_LIB; results meaningless, times not -- the method of round 5's weight-gradient
// work): -DCVVAE_ABLATE_STAGE=1 stages every other pass of the register-staged halo (HALF the loads, the GroupNorm + SiLU arithmetic
// and the LDS writes: what a kernel that re-uses two of three halo frames across time steps would still have to do), =2 stages
// chunk 0 only (no staging at all inside the K loop).  An upper bound on what any scheme that cuts the staging can buy (round 6).
#ifndef CVVAE_ABLATE_STAGE
#define CVVAE_ABLATE_STAGE 0
#endif
#ifndef CVVAE_LD_PF
#define CVVAE_LD_PF 0  // tuning aid: weight-ring depth of the DMA-staged instances (0: as deep as divides the time group)
#endif
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// Every wave keeps PF packed-weight fragments (1 KiB each) in flight and never branches on "last step", so it reads up
// to PF KiB past the last real fragment: cvvae_packed_weight_bytes() appends this many readable bytes.
constexpr int WEIGHT_TAIL_BYTES = 16 * 1024;

// XP: 0 = 16-bit model; 1 = fp32 model, three fp16 MFMAs per product ("exact"); 2 / 3 = fp32 model, one fp16 MFMA + bf8 / bf6 correction
// terms on the K = 64 fp8 MFMA ("fast", see conv_fwd_kernel)
// NB: 32-channel N-blocks per wave.  NB = 2: a wave multiplies every activation fragment it reads from LDS with TWO weight
// fragments (a 2 x MREP register block): half the LDS operand reads per MFMA -- the LDS pipe (128 B/clk/CU = one 1-KiB fragment
// per 8 clocks) is otherwise exactly saturated at the full MFMA rate of four SIMDs -- for twice the (L2-resident) weight stream.
// LD = 1 ("DMA-staged"): the halo tile goes global -> LDS by global_load_lds_dwordx4 (no VGPRs, no VALU: the instruction writes
// lane i's 16 bytes at LDS base + 16 i), into a CHANNEL-PLANE layout -- plane q holds channels 8q..8q+7 of every halo pixel at a
// 16-byte pitch, so a wave-load of 64 consecutive halo pixels of one plane is exactly one DMA instruction, the fragment reads of 16
// consecutive pixels are 256 contiguous bytes (conflict-free without the 16 padding bytes per pixel of the register-staged
// layout: a third less LDS at 16-channel chunks), and the tap offsets stay immediates.  Nothing stages in registers, so EVERY wave
// multiplies all the time (the register-staged kernel alternates: one wave of a SIMD stages while the other multiplies, and a
// lone wave issues an MFMA every ~38 instead of 32 cycles).  For instances WITHOUT a prologue (PRO = 0: the operand is consumed as
// stored -- upsample / downsample convs, conv_in, 1x1 layers, and any conv whose GroupNorm + SiLU was applied by cvvae_gn_silu_apply).
template <int KT, int KH, int KW, int ST, int SH, int SW, int TT, int TH, int TW, int WM, int WN, int KG, int KSUB, int XP = 0, int NB = 1, int LD = 0>
struct Geo {
  static constexpr int NTAPS = KT * KH * KW;
  static constexpr int BM = TT * TH * TW;
  static constexpr int BN = 32 * WN * NB;
  static constexpr int MREP = BM / 32 / WM;
  static constexpr int FT = (TT - 1) * ST + KT, FH = (TH - 1) * SH + KH, FW = (TW - 1) * SW + KW;
  static constexpr int NPIX = FT * FH * FW;
  static constexpr int CK = 16 * KSUB;
  // XP (fp32 activations, split-fp16 MFMA): a pixel holds, per 16 channels, hi[0..7] hi[8..15] lo[0..7] lo[8..15] (64 bytes);
  // XP == 2: hi[0..15] (fp16, 32 bytes) | lo[0..15] (bf8, 16 bytes) | hi[0..15] (bf8, 16 bytes)
  // XP == 3: hi[0..15] (fp16, 32 bytes) | bf6 codes of [lo*2^11 (8) | hi (8)] of channels 0..7, then of channels 8..15 (24 bytes) | pad
  // LD: planes of NPIXP = NPIX rounded up to whole 64-pixel wave-loads, 16 bytes per pixel; plane q of a buffer at q * PLB
  static constexpr int NQ = 2 * KSUB;                 // 8-channel planes per K chunk
  static constexpr int NGRP = (NPIX + 63) / 64;       // 64-pixel groups = DMA wave-loads per plane
  // PL ("planar" fast-fp32 layout, XP == 3 instances with eight fragments per wave): the register-staged XP pixel is 80 bytes per 16
  // channels (hi fp16 32 B + 24 B of bf6 codes, padded to the next conflict-free pitch), which confines the fast-fp32 instances to
  // 256- / 128-pixel tiles with MREP = 4.  Here the three fields live in separate LDS regions, each a plane with its own pitch and NO
  // padding: hi as the two 8-channel planes of the DMA layout (16 B per pixel), the codes as a 16-byte plane (dwords 0-3 of the
  // pixel's 24-byte field) and an 8-byte plane (dwords 4-5) -- 56 bytes per pixel and 16 channels.  A fragment row is 32 consecutive
  // pixels, so every operand read (ds_read_b128 / ds_read_b64) covers contiguous LDS: conflict-free without padding, tap offsets stay
  // immediates.  The two-frame 512-pixel tile of the 128-channel 3x3x3 layers (2 x 76 KB) and the all-waves-in-N 256-pixel tile of the
  // 256- / 512-channel ones (2 x 57 KB) then exist for the fp6 form, with MREP = 8: every weight record feeds 8 MFMAs instead of 4.
  // ... or as a 2 x 4 REGISTER BLOCK (NB = 2: two 32-channel blocks x four fragments per wave -- every operand read from LDS feeds TWO
  // MFMAs with two weight records): half the LDS operand bytes per MFMA.  The fp6 K loop reads 3.5 KiB per fragment and pair of
  // taps for 96 MFMA cycles: 149 B/clk per CU at the full matrix rate against the LDS pipe's 128 -- the 16-bit kernels' 128 B/clk
  // sit exactly ON that limit, which is why the same register block bought them nothing in round 3 (section 3.1)
  static constexpr bool PL = (XP == 3 && (MREP * NB >= 8 || NB == 2) && SW == 1 && TW >= 32);  // (NB = 2 with fewer fragments: the odd-frame sibling)
  // bytes per plane.  PL: + 256 / (2 KSUB) so that the 16 lanes of a staging write (8 or 4 consecutive pixels x the chunk's 2 or 4 hi
  // planes) spread over all 64 banks
  static constexpr int PLB = PL ? (NPIX * 16 + 255) / 256 * 256 + 256 / (2 * KSUB) : NGRP * 64 * 16;
  static constexpr int NDMA = NQ * NGRP;              // wave-loads per chunk, dealt to the 8 waves round-robin
  static constexpr int NDI = (NDMA + 7) / 8;          // ... per wave
  static constexpr int PIXB = (LD || PL) ? 16 : (XP ? CK * 4 : CK * 2) + 16;
  static constexpr int KSB = (LD || PL) ? 2 * PLB : (XP ? 64 : 32);  // byte offset of k16 sub-chunk ks inside a buffer: ks * KSB
  static constexpr int KHB = (LD || PL) ? PLB : 16;                  // ... of the upper k half (lanes 32-63 of an operand): + KHB
  // PL: code plane A (16 B per pixel) of sub-chunk ks at CQA + ks * PLB, code plane B (8 B per pixel) at CQB + ks * PLB / 2
  static constexpr int CQA = 2 * KSUB * PLB, CQB = 3 * KSUB * PLB;
  static constexpr int XPM = XP ? 3 : 1;  // weight records per k16 sub-chunk and tap (XP == 1: one MFMA each)
  static constexpr int BUFB = LD ? NQ * PLB : (PL ? KSUB * (PLB / 2) * 7 : NPIX * PIXB);
  static constexpr int LDSB = 2 * BUFB;
  static constexpr int NWV = WM * WN * KG;  // waves per workgroup: 8 (one workgroup per CU) or 4 (TWO workgroups per CU)
  static constexpr int IPP = 2 * KSUB;   // 16-byte items per pixel
  static constexpr int PPP = 256 / IPP;  // pixels per staging pass of one 256-thread group
  // pixels staged by group X (NWV == 8: the two wave groups split the halo; NWV == 4: the single group stages all of it)
  static constexpr int NPH = NWV == 4 ? (NPIX + PPP - 1) / PPP * PPP : ((NPIX + 1) / 2 + PPP - 1) / PPP * PPP;
  static constexpr int NPASS = NWV == 4 ? NPH / PPP : (cmax(NPH, NPIX - NPH) + PPP - 1) / PPP;
  // B fragments: two register sets (the reads of step st+1 fly under the MFMAs of step st), or -- MREP >= 8 -- ONE set updated in
  // place: the read for (st+1, r) is issued right after the MFMA of (st, r) and is needed MREP MFMAs (>= 256 cycles) later.  The
  // 32 registers this frees hold ALL staging loads of a chunk at once (one exposed memory latency per chunk instead of two).
  static constexpr int NAB = MREP >= 8 ? 1 : 2;
  // (PL: fp32 loads are eight registers per pass beside 128 accumulators, the weight ring and the 16 GroupNorm terms: half the
  //  passes per batch, or the staging spills its LDS addresses and reloads them -- with a full wait -- per item)
#ifndef CVVAE_PL_SBATCH_FULL
#define CVVAE_PL_SBATCH_FULL 0  // tuning aid: all staging passes of a planar tile in one batch
#endif
  static constexpr int SBATCH = (PL && !CVVAE_PL_SBATCH_FULL) ? (NPASS + 1) / 2
                                   : (NAB == 1 && NPASS <= 6) ? NPASS : (NPASS <= 4 ? NPASS : (NPASS <= 8 ? (NPASS + 1) / 2 : 4));
  static constexpr int STEPS = NTAPS * KSUB * XPM;   // k16 steps per chunk, ordered ks-major: st = (ks * NTAPS + tap) * XPM + part
  static constexpr int STEPS_W = STEPS / KG;   // steps one wave executes per chunk (K-group g takes ks in [g*KSUB/KG, ..))
  // weight fragments kept in flight per wave: deeper when a wave issues few MFMAs per fragment (small MREP)
  static constexpr int MPS = NB * MREP;  // MFMAs per k16 step of a wave
  // LD: the staging registers are gone, and the DMA wave-loads of the next chunk -- issued at the top of a chunk -- sit in the same
  // in-order vmcnt queue as the weight records: the first record requested AFTER them cannot be consumed before they have landed,
  // so the ring is as deep as divides the time group (9 / 8 / 6 records: the DMA has that many k16 steps to arrive)
  static constexpr int GSL = (KT == 3 && KG == 1 ? KH * KW : NTAPS) * KSUB;  // steps of a time group (or of the chunk)
  static constexpr int PFL = GSL % 9 == 0 ? 9 : (GSL % 8 == 0 ? 8 : (GSL % 6 == 0 ? 6 : (GSL % 4 == 0 ? 4 : (GSL % 3 == 0 ? 3 : (GSL % 2 == 0 ? 2 : 1)))));
#ifdef CVVAE_PF_OVERRIDE   // (probe builds only: tools/probes/conv_probe.hip)
  static constexpr int PF = CVVAE_PF_OVERRIDE;
#else
  static constexpr int PF = LD ? (CVVAE_LD_PF ? (GSL % CVVAE_LD_PF == 0 ? CVVAE_LD_PF : PFL) : PFL) : XP ? 3 : (KT == 3 && KH * KW == 1) ? KSUB : (STEPS_W % 9 == 0) ? ((MPS >= 8 || KH * KW < 9) ? 3 : 9) : (STEPS_W % 8 == 0 ? (MPS >= 8 ? 4 : 8) : (STEPS_W % 4 == 0 ? 4 : 3));
#endif
  // K-group reduction through LDS (KG == 2): each wave parks half of its accumulators (MREP/2 fragments x 4 KiB)
  static constexpr int REDB = KG == 2 ? 8 * (MREP / 2) * 4096 : 0;
  static constexpr int SMEMB = cmax(LDSB, REDB);
  static_assert(LD == 0 || (XP == 0 && NB == 1 && KG == 1 && NWV == 8 && TW >= 16), "DMA-staged instances: 16-bit, 8 waves, no K-group split, "
                "fragment rows of >= 16 consecutive pixels");
  static_assert(XP == 0 || KG == 1, "split-precision instances: no K-group split");
  static_assert(NB == 1 || (NB == 2 && KG == 1 && (XP == 0 || PL)), "two N-blocks per wave: 16-bit or planar fast-fp32 instances without K-group split");
  static_assert(NWV == 8 || (NWV == 4 && KG == 1), "8 waves per workgroup, or 4 (two workgroups per CU; no K-group split)");
  static_assert(NWV == 8 || LDSB <= 80 * 1024, "two resident workgroups share the CU's 160 KiB of LDS");
  static_assert(KG == 1 || (KG == 2 && KSUB % 2 == 0 && MREP % 2 == 0 && WM * WN == 4), "K-group split");
  static_assert(BM % (32 * WM) == 0, "tile rows must split into 32-row MFMA fragments per wave");
  static_assert(STEPS_W % PF == 0, "weight prefetch ring must divide the steps of a chunk");
  static_assert(SMEMB <= 160 * 1024, "LDS budget");
#ifndef CVVAE_PF_OVERRIDE
  static_assert(PF * 1024 <= WEIGHT_TAIL_BYTES, "weight prefetch ring reads past the packed buffer's tail");
#endif
  static_assert(NWV == 4 || NPH <= NPIX || NPIX <= PPP, "split");
};

template <typename T>
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  typename Tr<T>::v8 x = __builtin_bit_cast(typename Tr<T>::v8, u);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (float)x[j];
}
template <typename T>
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  typename Tr<T>::v8 x;
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = (T)f[j];
  return __builtin_bit_cast(uint4, x);
}

// 8 consecutive elements <-> fp32 registers, for the 16-bit storage types (one 16-byte access) and for float (two)
template <typename T>
struct Raw8 {
  uint4 a;
};
template <>
struct Raw8<float> {
  uint4 a, b;
};
template <typename T>
__device__ __forceinline__ Raw8<T> ldraw8(const T* p) {
  Raw8<T> r;
  r.a = *reinterpret_cast<const uint4*>(p);
  return r;
}
template <>
__device__ __forceinline__ Raw8<float> ldraw8<float>(const float* p) {
  Raw8<float> r;
  r.a = reinterpret_cast<const uint4*>(p)[0];
  r.b = reinterpret_cast<const uint4*>(p)[1];
  return r;
}
template <typename T>
__device__ __forceinline__ void unraw8(const Raw8<T>& r, float (&f)[8]) {
  unpack8<T>(r.a, f);
}
template <>
__device__ __forceinline__ void unraw8<float>(const Raw8<float>& r, float (&f)[8]) {
  f[0] = __uint_as_float(r.a.x); f[1] = __uint_as_float(r.a.y); f[2] = __uint_as_float(r.a.z); f[3] = __uint_as_float(r.a.w);
  f[4] = __uint_as_float(r.b.x); f[5] = __uint_as_float(r.b.y); f[6] = __uint_as_float(r.b.z); f[7] = __uint_as_float(r.b.w);
}
template <typename T>
__device__ __forceinline__ void ld8(const T* p, float (&f)[8]) {
  unraw8<T>(ldraw8<T>(p), f);
}
template <typename T>
__device__ __forceinline__ void st8(T* p, const float (&f)[8]) {
  *reinterpret_cast<uint4*>(p) = pack8<T>(f);
}
template <>
__device__ __forceinline__ void st8<float>(float* p, const float (&f)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}

// sum over the 32 lanes of each half-wave, delivered in lanes 31 and 63 (other lanes hold partial sums): four DPP row
// shifts (zero fill) give lane 15 of every 16-lane row its row sum, row_bcast:15 adds it into the next row (rows 1 and 3)
__device__ __forceinline__ float half_wave_sum(float x) {
  auto shr = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  x += shr(x, std::integral_constant<int, 0x111>{});  // row_shr:1
  x += shr(x, std::integral_constant<int, 0x112>{});  // row_shr:2
  x += shr(x, std::integral_constant<int, 0x114>{});  // row_shr:4
  x += shr(x, std::integral_constant<int, 0x118>{});  // row_shr:8
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xa, 0xf, false));  // row_bcast:15
  return x;
}

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// GroupNorm affine as ONE fused multiply-add wherever it is evaluated (conv prologue, cvvae_gn_silu_apply): the two forms
// must round identically
__device__ __forceinline__ float gn_affine(float x, float sc, float sh) { return __builtin_fmaf(x, sc, sh); }

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// helpers of the planar fast-fp32 K loop (conv_fwd_kernel, Geo::PL): one-instruction address sums the compiler can neither hoist nor
// merge (volatile), and a wave-uniform 64-bit offset made visibly so (an SGPR pair: the weight records are then addressed as the
// kernel argument's pointer + scalar offset + the lane's 32-bit offset -- offsets, not laundered pointers: a pointer rebuilt from
// integers is a FLAT pointer, and flat loads count in lgkmcnt, i.e. every weight request would drain the LDS reads in flight)
__device__ __forceinline__ unsigned pl_addv(unsigned a, unsigned b) {  // a + b
  unsigned d;
  asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ unsigned pl_add2x(unsigned a, unsigned b) {  // 2 a + b
  unsigned d;
  asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ long long pl_uniform(long long q) {
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// out-of-range tap handling: replicate = clamp, zero = flag
__device__ __forceinline__ int map_coord(int c, int L, int mode, bool& zero) {
  if (mode) return c < 0 ? 0 : (c >= L ? L - 1 : c);
  if (c < 0 || c >= L) {
    zero = true;
    return 0;
  }
  return c;
}

// XP ("extended precision", T = _Float16): the model is fp32 -- activations, residuals and outputs are float tensors, the
// packed weights are split-fp16 records (cvvae_pack_weights* with dtype CVVAE_F32).  Every fp32 operand x is staged as
// hi = fp16(x), lo = fp16(x - hi) and every product runs as THREE fp16 MFMAs into the same fp32 accumulator:
//   Whi.hi + Whi.lo + Wlo.hi   (Wlo.lo ~ 2^-22 relative is dropped)
// arranged so that the K = 16 of one MFMA holds [Whi(c0..7) | Whi(c0..7)] x [hi(c0..7) | lo(c0..7)]  (parts 1, 2: channels
// 0..7 / 8..15 of the sub-chunk) and [Wlo(c0..15)] x [hi(c0..15)] (part 0).  ~fp32 results (relative error ~1e-6) at 3x the
// MFMA work: the reference's fp32 model path (from_pretrained without torch_dtype) and north_star's |delta| <= 1e-3 bound.
//
// XP == 2 ("fast" fp32, dtype CVVAE_F32Q): the two correction terms only have to be right to 3-4 bits, so they run on the fp8
// matrix pipe, which has twice the rate per K element:   Whi.hi (fp16 MFMA)  +  bf8(Whi).bf8(lo) + bf8(Wlo).bf8(hi)  (ONE
// v_mfma_f32_32x32x64_f8f6f4 per PAIR of taps: its K = 64 holds 16 channels x 2 taps x 2 terms -- lanes 0-31 carry
// bf8(Whi) x bf8(lo), lanes 32-63 bf8(Wlo) x bf8(hi)).  e5m2 has fp16's exponent range, so nothing is scaled: bf8(v) is v with
// its mantissa rounded to 2 bits.  Cost per pair of taps: 2 fp16 MFMAs + 1 fp8 MFMA of twice the duration = 4 units instead of
// 6; error ~2^-14 relative per product (oracle/precision_ladder.py: latent max |delta| 1.3e-4 where the fp16 model has 2.6e-3
// and the three-MFMA form 1.9e-5).  The packed weights keep three 1-KiB records per (k16, tap): [0] Whi (fp16), [1], [2] the
// two halves of the pair's bf8 record (at the pair's FIRST tap; pairs never cross a run of KH*KW taps).
//
// XP == 3 (dtype CVVAE_F32Q6): the same with the correction terms in "bf6" (e3m2) on the block-scaled form of that instruction, at four
// times the fp16 rate per K element.  A lane's 32 codes are [lo * 2^11 | hi] of ONE tap's 16 channels (lanes 0-31: tap a, 32-63: tap
// b) against [Whi | Wlo * 2^11]: both halves have the operand's own magnitude, so ONE power of two per lane (the E8M0 block scale)
// serves both terms -- per (output channel, tap) for the weights (stored in the record), one per launch for the activations
// (ConvArgs.q6_scale / q6_eb from the caller's bound of the operand).  An LDS pixel holds hi (fp16, 32 B) | 24 bytes of codes.
template <typename T, int KT, int KH, int KW, int ST, int SH, int SW, int TT, int TH, int TW, int WM, int WN, int KG, int KSUB,
          int PRO, int UPS, int XP = 0, int NB = 1, int LD = 0>
__global__ __launch_bounds__(WM * WN * KG * 64, (WM * WN * KG == 4 ? 2 : 1)) void conv_fwd_kernel(const ConvArgs p) {
  using G = Geo<KT, KH, KW, ST, SH, SW, TT, TH, TW, WM, WN, KG, KSUB, XP, NB, LD>;
  static_assert(LD == 0 || (PRO == 0 && UPS != 1 && !(KT == 3 && KH == 3 && KW == 1)), "DMA staging: no prologue, no nearest-2x gather, not the row-packed layer");
  using TIO = std::conditional_t<XP != 0, float, T>;  // element type of the activation tensors in HBM
  static_assert(XP == 0 || std::is_same<T, _Float16>::value, "split precision runs on fp16 MFMA");
  constexpr int XPM = G::XPM;
  // NWV == 4 (WM x WN = 4 waves, 256 threads, <= 256 VGPRs, <= 80 KiB of LDS): TWO workgroups are resident per CU, one wave of
  // each per SIMD.  Nothing couples them, so one workgroup's serial parts (first halo chunk, store tail, barrier waits, the
  // GroupNorm + SiLU VALU work of the staging, which runs ~4x slower beside a saturated MFMA stream) run under the other's MFMAs.
  constexpr int NWV = G::NWV;
  using v8 = typename Tr<T>::v8;
  constexpr int MREP = G::MREP, PIXB = G::PIXB, NPASS = G::NPASS, STEPS = G::STEPS, STEPS_W = G::STEPS_W, PF = G::PF,
                CK = G::CK, NTAPS = G::NTAPS;

  __shared__ __attribute__((aligned(16))) char smem[G::SMEMB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = NWV == 4 ? 0 : wave >> 2;  // 0: stage-then-compute, 1: compute-then-stage
  // XP == 3: the activations' power-of-two scale (codes = value / q6_scale) and its E8M0 byte -- from the host's bound (ConvArgs), or
  // from a bound the device holds: floor(log2(28 / bound)) is the exponent field of the quotient (the host does the same with frexpf)
  float q6_scale = p.q6_scale;
  int q6_eb = p.q6_eb;
  if constexpr (XP == 3) {
    if (p.q6_bound != nullptr) {
      const float bnd = *p.q6_bound;
      const float v = 28.0f / (bnd > 1.0e-30f ? bnd : 1.0e-30f);
      int sft = (int)((__float_as_uint(v) >> 23) & 0xffu) - 127;
      sft = sft < -100 ? -100 : (sft > 100 ? 100 : sft);
      q6_eb = __builtin_amdgcn_readfirstlane(127 - sft);
      q6_scale = __uint_as_float((unsigned)q6_eb << 23);  // 2^-sft
    }
  }
  const int wave_n = wave % WN;
  const int wave_m = (wave / WN) % WM;
  const int kgrp = KG == 2 ? grp : 0;  // K-group: which half of the chunk's k16 sub-chunks this wave multiplies

  // ---- XCD-aware, bijective block -> tile map (tile_map.h): each XCD (bid % 8) gets a contiguous run of logical tiles, so
  //      neighbouring halo tiles and all N-tiles of one M-tile share one L2; short (time-folded) tiles run last on every XCD
  const int logical = logical_tile_of_block((int)gridDim.x, (int)blockIdx.x, p.ntiles_n * (UPS == 2 ? 4 : 1), p.tiles_t,
                                            p.t_short_lo, p.t_short_hi, UPS == 2 ? p.ws_window : 0);
  // logical order: N-tile fastest (the N-tiles of one pixel tile re-use its halo from L2), then TIME, then x, y, b.
  // Time-adjacent tiles share 2 of their 3 halo frames; with time next-fastest they run at the same moment on CUs of
  // the same XCD, so the K-chunk passes over the halo stay inside that XCD's 4 MiB L2 instead of thrashing it.
  const int ntile = logical % p.ntiles_n;
  int mt = logical / p.ntiles_n;
  int phase = 0;  // UPS == 2: the four (py, px) phases of a pixel tile are consecutive logical tiles (their halos coincide)
  if (UPS == 2) {
    phase = mt & 3;
    mt >>= 2;
  }
  const int py = phase >> 1, px = phase & 1;
  int tw_i, th_i, tt_i, b;
  if (p.order == 1) {
    tt_i = mt % p.tiles_t;
    mt /= p.tiles_t;
    tw_i = mt % p.tiles_w;
    mt /= p.tiles_w;
    th_i = mt % p.tiles_h;
    b = mt / p.tiles_h;
  } else {
    tw_i = mt % p.tiles_w;
    mt /= p.tiles_w;
    th_i = mt % p.tiles_h;
    mt /= p.tiles_h;
    tt_i = mt % p.tiles_t;
    b = mt / p.tiles_t;
  }
  const int t0 = tt_i * TT + p.t_begin, y0 = th_i * TH, x0 = tw_i * TW;
  const int tile_in_b = p.tile_base + (tt_i * p.tiles_h + th_i) * p.tiles_w + tw_i;  // pixel-tile index inside batch row b
  // time-shuffle stores drop output frame -1 = channels [0, Cout/2) of conv frame 0: a workgroup whose whole tile is that
  // is done before it starts (the host zero-fills the GroupNorm records, so the missing ones read as empty)
  if (UPS != 1 && TT == 1 && p.out_mode == 2 && t0 == 0 && (ntile + 1) * G::BN <= (p.Cout >> 1)) return;

  // ---- time folds (KT == 3).  At a clip boundary two or three time taps of an output frame read the SAME stored frame
  //      (replicate padding) or a zero frame (zero padding).  The chunk's steps are therefore walked as up to three TIME GROUPS
  //      of KH*KW*KSUB steps, each with a weight slot and an LDS frame chosen per wave (a wave's fragments lie in one output
  //      frame): distinct frames -> slots W0, W1, W2 on frames 0, 1, 2; frames 0 = 1 -> (W0+W1 on frame 1), W2; frames 1 = 2 ->
  //      W0, (W1+W2 on frame 1); all equal -> W0+W1+W2 on frame 1 (the summed slots come from cvvae_pack_weights_tfolds:
  //      p.w_taps == 2*NTAPS); zero frames are simply skipped.  Halo frames no wave of the workgroup reads are not staged
  //      either.  cfg 3: 6-20 % fewer MFMAs on the causal encoder convs, 4-13 % on the decoder's.
  constexpr bool TFOLD = (KT == 3 && KG == 1);
  constexpr int NSP = KH * KW, GS = NSP * KSUB * XPM;  // steps of one time group
  // a wave's fragments lie in ONE output frame -- or (WM == 1 with a multi-frame tile: the 4-wave instances) span all of the
  // tile's frames, in which case the tile takes a fold only when every frame has the same plan (else: plain three groups)
  static_assert(!TFOLD || (GS % PF == 0 && (TT == 1 || WM == 1 || (TH * TW) % (MREP * 32) == 0)), "time-group plan");
  const long long w_ks = (long long)(TFOLD ? p.w_taps : NTAPS) * 512;  // elements between consecutive k16 record groups
  const long long w_cs = w_ks * KSUB;                                   // ... between consecutive K chunks
  int tf_ng = 3;
  long long tf_w0 = 0, tf_w1 = NSP * XPM * 512, tf_w2 = 2 * NSP * XPM * 512;  // weight slot of each time group (element offsets)
  unsigned tf_l0 = 0, tf_l1 = G::FH * G::FW * PIXB, tf_l2 = 2 * G::FH * G::FW * PIXB;  // LDS frame of each time group
  int hf_a = 0, hf_b = G::FT - 1;  // halo frames [hf_a, hf_b] some wave of this workgroup reads
  if constexpr (TFOLD) {
    // plan of output frame `to`: number of time groups, (weight slot, tap frame) of groups 0 and 1 (group 2 is always W2 on 2)
    // (the same table as time_fold_plan() in tile_map.h, which the host uses and tests/c/tile_map_test.cpp checks against the
    //  27-tap sum for every clip length, stride and padding)
    auto tf_variant = [&](int to, int& ng, int& slot0, int& slot1, int& dt0, int& dt1) {
      const int f0 = to * ST - p.pt;  // input frame of time tap 0 before padding
      auto clampT = [&](int v) { return v < 0 ? 0 : (v >= p.Tl ? p.Tl - 1 : v); };
      ng = 3; slot0 = 0; slot1 = 1; dt0 = 0; dt1 = 1;
      if (p.mode_t != 0) {  // replicate
        if (p.w_taps == 2 * NTAPS * XPM) {
          const bool eq01 = clampT(f0) == clampT(f0 + 1), eq12 = clampT(f0 + 1) == clampT(f0 + 2);
          if (eq01 && eq12) { ng = 1; slot0 = 5; dt0 = 1; }
          else if (eq01) { ng = 2; slot0 = 3; dt0 = 1; slot1 = 2; dt1 = 2; }
          else if (eq12) { ng = 2; slot0 = 0; dt0 = 0; slot1 = 4; dt1 = 1; }
        }
      } else {  // zero padding: a tap on a padding frame contributes exactly nothing
        const bool z0 = f0 < 0 || f0 >= p.Tl, z2 = f0 + 2 < 0 || f0 + 2 >= p.Tl;
        if (z0 && z2) { ng = 1; slot0 = 1; dt0 = 1; }
        else if (z0) { ng = 2; slot0 = 1; dt0 = 1; slot1 = 2; dt1 = 2; }
        else if (z2) { ng = 2; slot0 = 0; dt0 = 0; slot1 = 1; dt1 = 1; }
      }
    };
    int slot0, slot1, dt0, dt1;
    tf_variant(t0 + (TT > 1 && WM > 1 ? (wave_m * MREP * 32) / (TH * TW) : 0), tf_ng, slot0, slot1, dt0, dt1);
    bool plain = false;  // WM == 1, TT > 1: the wave covers every frame of the tile -> one plan for all of them, or none
    if (TT > 1 && WM == 1) {
#pragma unroll
      for (int tt = 1; tt < TT; ++tt) {
        int ng, s0_, s1_, d0_, d1_;
        tf_variant(t0 + tt, ng, s0_, s1_, d0_, d1_);
        plain = plain || ng != tf_ng || s0_ != slot0 || s1_ != slot1 || d0_ != dt0 || d1_ != dt1;
      }
      if (plain) { tf_ng = 3; slot0 = 0; slot1 = 1; dt0 = 0; dt1 = 1; }
    }
    tf_w0 = (long long)slot0 * (NSP * XPM * 512);
    tf_w1 = (long long)slot1 * (NSP * XPM * 512);
    tf_l0 = (unsigned)dt0 * (G::FH * G::FW * PIXB);
    tf_l1 = (unsigned)dt1 * (G::FH * G::FW * PIXB);
    hf_a = G::FT;
    hf_b = -1;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {  // union over the output frames of the tile
      int ng, s0_, s1_, d0_, d1_;
      tf_variant(t0 + tt, ng, s0_, s1_, d0_, d1_);
      if (plain) { ng = 3; d0_ = 0; d1_ = 1; }
      const int lo = tt * ST + d0_, hi = tt * ST + (ng == 3 ? 2 : (ng == 2 ? d1_ : d0_));
      hf_a = lo < hf_a ? lo : hf_a;
      hf_b = hi > hf_b ? hi : hf_b;
    }
  }
  // ---- staging plan (chunk independent): which stored pixel feeds each of my halo slots (the halo pixels of the frames
  //      [hf_a, hf_b] are split between the two wave groups)
  const int tl = tid & 255;
  const int sq = tl % G::IPP;  // my 16-byte slice (8 channels) inside the chunk -- fixed for the whole kernel
  const int spl = tl / G::IPP;
  const int hbase = hf_a * (G::FH * G::FW), hcnt = (hf_b - hf_a + 1) * (G::FH * G::FW);
  const int nph_x = NWV == 4 ? (hcnt + G::PPP - 1) / G::PPP * G::PPP   // the single group stages all of it
                             : (TFOLD ? (((hcnt + 1) / 2 + G::PPP - 1) / G::PPP * G::PPP) : G::NPH);  // pixels staged by group X
  const int pstart = hbase + (grp ? nph_x : 0);
  const int pend = hbase + (grp ? hcnt : (nph_x < hcnt ? nph_x : hcnt));
  // stored pixel that feeds halo slot hp (frame-major inside the halo tile), or -1: zero padding
  auto halo_src = [&](int hp) -> int {
    const int f = hp / (G::FH * G::FW);
    const int rem = hp - f * (G::FH * G::FW);
    const int hy = rem / G::FW;
    const int hx = rem - hy * G::FW;
    bool zero = false;
    int ts = map_coord(t0 * ST + f - p.pt, p.Tl, p.mode_t, zero);
    // UPS == 2: phase 0 along an axis reads rows {y-1, y} (front pad 1), phase 1 reads {y, y+1} (front pad 0)
    int ys = map_coord(y0 * SH + hy - (UPS == 2 ? p.ph - py : p.ph), p.Hl, p.mode_hw, zero);
    int xs = map_coord(x0 * SW + hx - (UPS == 2 ? p.pw - px : p.pw), p.Wl, p.mode_hw, zero);
    if (UPS == 1) {
      ys >>= 1;
      xs >>= 1;
    }
    return zero ? -1 : ((b * p.Ti + ts) * p.Hi + ys) * p.Wi + xs;
  };
  int srcpix[NPASS];
  unsigned passmask = 0;  // wave-uniform: passes in which some lane of this wave has a slot
  if constexpr (!LD) {
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
      const int hp = pstart + spl + k * G::PPP;
      const int sp = hp < pend ? halo_src(hp) : -2;  // -2: slot not mine / beyond tile, -1: zero padding
      srcpix[k] = sp;
      if (__builtin_amdgcn_ballot_w64(sp != -2) != 0) passmask |= 1u << k;
    }
  }
  // ---- LD: the DMA plan (chunk independent).  Wave-load ii = wave + 8 k of a chunk fills plane ii % NQ, halo pixels
  //      64 (ii / NQ) .. +63: my lane's slot of it is fed by stored pixel dpix[k] (NOSRC: zero padding, a halo frame nobody reads, or
  //      beyond the halo tile -- such lanes are masked out of the load; zero-padding slots are zeroed ONCE below, in both buffers)
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr unsigned NOSRC = 0xffffffffu;
  unsigned dpix[LD ? G::NDI : 1];
  if constexpr (LD) {
#pragma unroll
    for (int k = 0; k < G::NDI; ++k) {
      const int ii = wave + 8 * k;
      const int hp = (ii / G::NQ) * 64 + lane;
      unsigned sp = NOSRC;
      if (ii < G::NDMA && hp >= hbase && hp < hbase + hcnt) {
        const int v = halo_src(hp);
        if (v >= 0) sp = (unsigned)v;
        else {  // zero padding: the DMA never writes this slot
          const uint4 z = make_uint4(0, 0, 0, 0);
          char* d = smem + (ii % G::NQ) * G::PLB + hp * 16;
          *reinterpret_cast<uint4*>(d) = z;
          *reinterpret_cast<uint4*>(d + G::BUFB) = z;
        }
      }
      dpix[k] = sp;
    }
  }
  auto dma_from = [&](const TIO* __restrict__ src, size_t src_ps, int chunk, int bufsel) {
    if constexpr (LD) {
      const char* cbase = reinterpret_cast<const char*>(src) + (size_t)chunk * (CK * sizeof(TIO));  // wave-uniform
      const unsigned pitch = (unsigned)(src_ps * sizeof(TIO));
#pragma unroll
      for (int k = 0; k < G::NDI; ++k) {
        const int ii = wave + 8 * k;
        if (ii >= G::NDMA) continue;
        const int q = ii % G::NQ, j = ii / G::NQ;
        if (j * 64 + 63 < hbase || j * 64 >= hbase + hcnt) continue;  // (wave-uniform: a halo frame no wave reads)
        if (dpix[k] != NOSRC)
          __builtin_amdgcn_global_load_lds((gptr_t)(cbase + (size_t)dpix[k] * pitch + q * 16),
                                           (lptr_t)(smem + bufsel * G::BUFB + q * G::PLB + j * 1024), 16, 0, 0);
      }
    }
  };
  // (PL: my hi slice is pixel slot (pstart + spl) of hi plane sq)
  const int lds_w0 = (pstart + spl) * PIXB + (G::PL ? sq * G::PLB : (XP ? (sq >> 1) * 64 + (sq & 1) * 16 : sq * 16));  // XP: my hi slice; lo = +32
  // (XP == 2: my 8 bf8 lo values at +32 + (sq & 1) * 8 and my 8 bf8 hi values at +48 + (sq & 1) * 8 of the k16 group)
  const int lds_q8 = (pstart + spl) * PIXB + (sq >> 1) * 64 + 32 + (sq & 1) * 8;
  const TIO* __restrict__ inp = reinterpret_cast<const TIO*>(p.in);
  const size_t gn_row = (size_t)(b * p.gn_rpb + (p.gn_rpb > 1 ? t0 : 0)) * (size_t)p.Cin;

#ifdef CVVAE_CONV_PROBE
  int probe_n = 0;
#endif
  auto stage_from = [&](auto pro_tag, const TIO* __restrict__ src, size_t src_ps, int chunk, int bufsel) {
    constexpr int PRO_ = decltype(pro_tag)::value;  // prologue applied to THIS source (the shortcut input has none)
    const int c0 = chunk * CK + sq * 8;
    float sc[8], sh[8];
    if (PRO_ != 0) {
      const float4* ps = reinterpret_cast<const float4*>(p.gsc + gn_row + c0);
      const float4* pb = reinterpret_cast<const float4*>(p.gsh + gn_row + c0);
      float4 a0 = ps[0], a1 = ps[1], b0 = pb[0], b1 = pb[1];
      sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
      sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
    }
    char* dst = smem + bufsel * G::BUFB + lds_w0;
    constexpr int SB = G::SBATCH;  // passes in flight together (bounds the staging registers)
#pragma unroll
    for (int k0 = 0; k0 < NPASS; k0 += SB) {
      Raw8<TIO> raw[SB];
#pragma unroll
      for (int kk = 0; kk < SB; ++kk) {
        const int k = k0 + kk;
        // (fast-fp32 instances: EVERY pass loads, needed or not -- straight-line code.  With a wave-uniform branch per pass hipcc merged each load's
        //  block with the same pass's block of the processing loop below: load, wait, GroupNorm arithmetic, next load ... -- one
        //  exposed memory latency per pass; in-kernel stamps: 7-10 k cycles "issuing" six passes against 0.9 k in the 16-bit kernel,
        //  profiles/r6_probe_fast_fp32_timelines.log)
        if (CVVAE_ABLATE_STAGE == 1 && (k & 1)) continue;
        if (k < NPASS && (XP >= 2 || ((passmask >> k) & 1))) {
          // unconditional load (slot 0 of the tensor for padding / foreign slots) keeps the loads branch-free
          const int sp = srcpix[k] < 0 ? 0 : srcpix[k];
          if constexpr (KT == 3 && KH == 3 && KW == 1 && !std::is_same<TIO, float>::value) {
            // row-packed first layer (cvvae_conv_desc.in_overlap): the 16 virtual channels of a pixel start at an 8-byte boundary
            const uint2* q2 = reinterpret_cast<const uint2*>(src + (size_t)sp * src_ps + c0);
            const uint2 lo2 = q2[0], hi2 = q2[1];
            raw[kk].a = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
          } else {
            raw[kk] = ldraw8<TIO>(src + (size_t)sp * src_ps + c0);
          }
        }
      }
#ifdef CVVAE_CONV_PROBE
      if (k0 == 0) {
        CVVAE_PROBE_MARK();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CVVAE_PROBE_MARK();
      }
#endif
      if constexpr (XP >= 2) __builtin_amdgcn_sched_barrier(0);  // (all loads of the batch are requested before the first is consumed)
#pragma unroll
      for (int kk = 0; kk < SB; ++kk) {
        const int k = k0 + kk;
        if (CVVAE_ABLATE_STAGE == 1 && (k & 1)) continue;
        if (k < NPASS && ((passmask >> k) & 1)) {
          if (srcpix[k] == -2) continue;
          if constexpr (XP != 0) {  // fp32 source -> (GroupNorm affine, SiLU in fp32) -> hi = fp16(x), lo = fp16(x - hi)
            float f[8], fl[8];
            unraw8<float>(raw[kk], f);
            if (PRO_ != 0) {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                float v = gn_affine(f[j], sc[j], sh[j]);
                f[j] = (PRO_ == 1) ? silu_f(v) : v;
              }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float h = (float)(T)f[j];
              fl[j] = f[j] - h;
              f[j] = h;
            }
            if constexpr (XP == 3) {  // hi fp16 (32 bytes) | 24 bytes of bf6 codes: my 8 channels' [lo * 2^11 (8) | hi (8)] at +12 * (sq & 1)
#pragma unroll
              for (int j = 0; j < 8; ++j) fl[j] *= 2048.0f;
              uint4 oh = pack8<T>(f);
              const uint4 ol = pack8<T>(fl);
              uint3 q = cvt16_bf6(ol, oh, q6_scale);
              if (srcpix[k] < 0) {
                oh = make_uint4(0, 0, 0, 0);
                q = make_uint3(0, 0, 0);
              }
              // dense 24-byte field at +32: item 0 = dwords 0-2, item 1 = dwords 3-5 -> an 8-byte and a 4-byte store each
              const bool odd = (sq & 1) != 0;
              *reinterpret_cast<uint4*>(dst + k * (G::PPP * PIXB)) = oh;
              if constexpr (G::PL) {
                // the same 24-byte field, split over code plane A (dwords 0-3, 16 B per pixel) and code plane B (dwords 4-5, 8 B per
                // pixel) of my k16 sub-chunk: item 0 -> A[0..11]; item 1 -> A[12..15] and B[0..7]
                // (branch-free: every thread issues one 8-byte and one 4-byte store; parity selects addresses and data)
                const int hp = pstart + spl + k * G::PPP;
                const int oA = bufsel * G::BUFB + G::CQA + (sq >> 1) * G::PLB + hp * 16;
                const int oB = bufsel * G::BUFB + G::CQB + (sq >> 1) * (G::PLB / 2) + hp * 8;
                *reinterpret_cast<uint2*>(smem + (odd ? oB : oA)) = odd ? make_uint2(q.y, q.z) : make_uint2(q.x, q.y);
                *reinterpret_cast<unsigned*>(smem + oA + (odd ? 12 : 8)) = odd ? q.x : q.z;
              } else {
              char* d6 = smem + bufsel * G::BUFB + lds_q8 - (sq & 1) * 8 + k * (G::PPP * PIXB);
              *reinterpret_cast<uint2*>(d6 + (odd ? 16 : 0)) = odd ? make_uint2(q.y, q.z) : make_uint2(q.x, q.y);
              *reinterpret_cast<unsigned*>(d6 + (odd ? 12 : 8)) = odd ? q.x : q.z;
              }
            } else if constexpr (XP == 2) {  // hi fp16 | bf8(lo) | bf8(hi)
              uint4 oh = pack8<T>(f);
              uint2 l8 = pack8_bf8(fl), h8 = pack8_bf8(f);
              if (srcpix[k] < 0) {
                oh = make_uint4(0, 0, 0, 0);
                l8 = h8 = make_uint2(0, 0);
              }
              char* d8 = smem + bufsel * G::BUFB + lds_q8 + k * (G::PPP * PIXB);
              *reinterpret_cast<uint4*>(dst + k * (G::PPP * PIXB)) = oh;
              *reinterpret_cast<uint2*>(d8) = l8;
              *reinterpret_cast<uint2*>(d8 + 16) = h8;
            } else {
            uint4 oh = pack8<T>(f), ol = pack8<T>(fl);
            if (srcpix[k] < 0) oh = ol = make_uint4(0, 0, 0, 0);  // zero padding is applied AFTER GroupNorm+SiLU
            *reinterpret_cast<uint4*>(dst + k * (G::PPP * PIXB)) = oh;
            *reinterpret_cast<uint4*>(dst + k * (G::PPP * PIXB) + 32) = ol;
            }
          } else {
          uint4 o = raw[kk].a;
          if (PRO_ != 0) {
            float f[8];
            unpack8<T>(raw[kk].a, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float v = gn_affine(f[j], sc[j], sh[j]);
              f[j] = (PRO_ == 1) ? silu_f(v) : v;
            }
            o = pack8<T>(f);
          }
          if (srcpix[k] < 0) o = make_uint4(0, 0, 0, 0);  // zero padding is applied AFTER GroupNorm+SiLU
          *reinterpret_cast<uint4*>(dst + k * (G::PPP * PIXB)) = o;
          }
        }
      }
    }
  };
  auto stage = [&](int chunk, int bufsel) {
    if (CVVAE_ABLATE_STAGE == 2 && chunk != 0) return;
    stage_from(std::integral_constant<int, PRO>{}, inp, (size_t)p.in_ps, chunk, bufsel);
  };

  // ---- MFMA plan
  const int nb0 = (ntile * WN + wave_n) * NB;  // my first 32-output-channel block (NB consecutive ones; the host admits NB = 2
  const int nb = nb0;                          // instances only for Cout % 64 == 0, so the blocks of a wave are all real or none)
  const bool active = nb0 < p.nblk32;
  unsigned aoff[MREP];
#pragma unroll
  for (int r = 0; r < MREP; ++r) {
    const int m = (wave_m * MREP + r) * 32 + (lane & 31);
    const int tx = m % TW, ty = (m / TW) % TH, tt = m / (TW * TH);
    aoff[r] = (unsigned)((((tt * ST) * G::FH + ty * SH) * G::FW + tx * SW) * PIXB + (lane >> 5) * G::KHB +
                         kgrp * (KSUB / KG) * G::KSB);
  }
  // PL: ONE per-fragment array -- the fragment's pixel in 8-byte units (the pitch of code plane B; the 16-byte planes are addressed
  // as 2 x + base) -- instead of aoff[] (dead in those instances: the shortcut / 16-bit paths that read it are not instantiated)
  unsigned aoff8[G::PL ? MREP : 1];
  if constexpr (G::PL) {
#pragma unroll
    for (int r = 0; r < MREP; ++r) aoff8[r] = (aoff[r] - (unsigned)((lane >> 5) * G::KHB)) >> 1;
  }
  // step i of a time group: k16 sub-chunk i / NSP, spatial tap i % NSP
  auto tf_rec = [&](int i) -> long long {
    const int j = i / XPM, part = i % XPM;
    return (long long)(j / NSP) * w_ks + ((j % NSP) * XPM + part) * 512;
  };
  // XP: LDS byte offset of step (k16 sub-chunk ks, part) inside a pixel, and the lane term of parts 1, 2: lanes 32-63 read the
  // lo half 32 bytes above the hi half (aoff[] carries (lane >> 5) * 16, the standard k split of part 0)
  const unsigned lhi16 = XP ? (unsigned)((lane >> 5) * 16) : 0u;
  const T* wq = reinterpret_cast<const T*>(p.w) + (size_t)b * (size_t)p.w_bstride +
                (UPS == 2 ? (size_t)phase * (size_t)p.w_phase_stride : 0) +
                (TFOLD ? (size_t)(active ? nb : 0) * (size_t)p.nchunks * (size_t)w_cs
                       : (size_t)(active ? nb : 0) * (size_t)p.nchunks * (STEPS * 512) + kgrp * (STEPS_W * 512)) + lane * 8;
  // XP == 2: the ring holds the FOUR records of one pair of taps -- [0] Whi of tap a, [1], [2] the halves of the pair's bf8
  // record, [3] Whi of tap b -- each refilled with the next pair's right after its last use
  // (XQ_DEPTH pairs in flight: with four fragments per wave a pair lasts ~400-500 clocks -- less than an L2 round trip under load,
  //  and the MFMA phase then runs at the latency of its weight stream: measured 60 ticks per MFMA, tools/probes/conv_probe.hip CFG 11/12)
#ifndef CVVAE_XQ_DEPTH
#define CVVAE_XQ_DEPTH 2
#endif
#ifndef CVVAE_XQ_DEPTH8
#define CVVAE_XQ_DEPTH8 1
#endif
  // (eight fragments per wave: a pair of taps lasts 768 cycles -- one pair in flight covers an L2 round trip)
  constexpr int XQD = G::PL ? CVVAE_XQ_DEPTH8 : CVVAE_XQ_DEPTH;
  static_assert(XQD == 1 || XQD == 2, "fast-fp32 weight ring: one or two pairs of taps in flight");
  constexpr int NWF = XP >= 2 ? 4 * XQD : PF;
  static_assert(XP < 2 || TFOLD || KT == 1, "fast-fp32 instances: 3-tap time kernels walk time groups, the others have KT = 1");
  const long long wq_ks = (long long)(TFOLD ? p.w_taps : NTAPS * 3) * 512, wq_cs = wq_ks * KSUB;  // (XP == 2)
  const T* wqx = reinterpret_cast<const T*>(p.w) + (size_t)b * (size_t)p.w_bstride +
                 (UPS == 2 ? (size_t)phase * (size_t)p.w_phase_stride : 0) +
                 (size_t)(active ? nb : 0) * (size_t)p.nchunks * (size_t)wq_cs + lane * 8;
  // (PL: the same as a wave-uniform base -- SGPRs -- plus the lane's byte offset)
  const long long wqu = (long long)b * p.w_bstride + (UPS == 2 ? (long long)phase * p.w_phase_stride : 0) +
                        (long long)(active ? nb : 0) * (long long)p.nchunks * wq_cs;  // (element offset from p.w)
  const unsigned wlane = (unsigned)lane * 16u;
  // elements between the packed weights of consecutive 32-channel blocks (NB = 2: the wave's second block)
  const long long wq_nbs = (long long)p.nchunks * (TFOLD ? w_cs : (long long)(STEPS * 512));
  v8 wf[NB][NWF];
  // LD: the weight records are requested and awaited BY HAND.  hipcc's own s_waitcnt placement treats a pending global_load_lds
  // as an access that may touch both memories and turns the next vmcnt dependency into a full drain -- of the wave-loads just
  // issued and of the ring's whole read-ahead.  With the request and the wait as asm statements the ring keeps its depth: at the
  // start of step i the PF - 1 youngest requests are the records of steps i+1 .. i+PF-1, so "at most PF - 1 outstanding" means record
  // i has arrived (any other request in the queue -- the wave-loads, a residual run -- only makes that wait stricter, never laxer).
  auto wload = [&](v8& dst, const T* ptr) __attribute__((always_inline)) {
    if constexpr (LD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr));
    else dst = *reinterpret_cast<const v8*>(ptr);
  };
  auto wwait = [&](v8& rec) __attribute__((always_inline)) {
    if constexpr (LD) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rec) : "n"(PF - 1));
  };
  const long long wqx_nbs = (long long)p.nchunks * wq_cs;  // (fast fp32: elements between a wave's two 32-channel blocks, NB = 2)
  if constexpr (XP >= 2) {
#pragma unroll
    for (int n = 1; n < NB; ++n) {  // (NB = 2, planar instances: one pair of taps in flight -- XQD = 1)
      static_assert(NB == 1 || XQD == 1, "two N-blocks per wave: one pair of taps in flight");
      const T* e2 = wqx + (TFOLD ? tf_w0 : 0) + n * wqx_nbs;
      wf[n][0] = *reinterpret_cast<const v8*>(e2);
      wf[n][1] = *reinterpret_cast<const v8*>(e2 + 512);
      wf[n][2] = *reinterpret_cast<const v8*>(e2 + 1024);
      wf[n][3] = *reinterpret_cast<const v8*>(e2 + (KH * KW > 1 ? 3 * 512 : 0));
    }
    const T* e = wqx + (TFOLD ? tf_w0 : 0);  // chunk 0, time group 0, pair 0 (taps 0 and 1 of k16 sub-chunk 0)
    wf[0][0] = *reinterpret_cast<const v8*>(e);
    wf[0][1] = *reinterpret_cast<const v8*>(e + 512);
    wf[0][2] = *reinterpret_cast<const v8*>(e + 1024);
    wf[0][3] = *reinterpret_cast<const v8*>(e + (KH * KW > 1 ? 3 * 512 : 0));
    if constexpr (XQD == 2) {  // ... and pair 1 (a group holds at least two pairs)
      constexpr int R_ = KH * KW, PR_ = (R_ + 1) / 2;
      static_assert(PR_ * KSUB >= 2, "two pairs per time group");
      const T* e1 = e + (long long)(1 / PR_) * wq_ks + 2 * (1 % PR_) * (3 * 512);
      wf[0][4] = *reinterpret_cast<const v8*>(e1);
      wf[0][5] = *reinterpret_cast<const v8*>(e1 + 512);
      wf[0][6] = *reinterpret_cast<const v8*>(e1 + 1024);
      wf[0][7] = *reinterpret_cast<const v8*>(e1 + ((2 * (1 % PR_) + 1 < R_) ? 3 * 512 : 0));
    }
  } else {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int i = 0; i < PF; ++i)
        wload(wf[n][i], (TFOLD ? wq + tf_w0 + tf_rec(i) : wq + i * 512) + n * wq_nbs);
  }

  // The accumulators start from the BIAS (alpha == 1, i.e. every layer but the attention score product): 128 v_add per lane
  // leave the store tail -- which is VALU-issue bound -- for nothing (the zero fill cost the same moves).  K-group 1 of a
  // K-group split starts from zero (the halves are summed).
  // Register quad g of a lane holds channels nb*32 + 16(g>>1) + 8(lane>>5) + 4(g&1) + j: the weight packers put output channel
  // sigma(i) into MFMA row i (sigma swaps bits 2 and 3 of the index inside the 32-channel block), so quads 2p and 2p+1 of a
  // lane are 8 CONSECUTIVE channels and the store tail writes 16-byte runs without any cross-lane exchange.
  const bool bias_pre = p.alpha == 1.0f;
  f32x16 acc[NB * MREP];  // fragment r of N-block n: acc[n * MREP + r]
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    float bq[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias_pre && active && kgrp == 0)
        bv = *reinterpret_cast<const float4*>(p.bias + (nb0 + n) * 32 + (g >> 1) * 16 + (lane >> 5) * 8 + (g & 1) * 4);
      bq[g * 4] = bv.x; bq[g * 4 + 1] = bv.y; bq[g * 4 + 2] = bv.z; bq[g * 4 + 3] = bv.w;
    }
#pragma unroll
    for (int r = 0; r < MREP; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[n * MREP + r][i] = bq[i];
  }

  // ---- residual pre-accumulation (per-frame convs = the ResnetBlock tails).  The residual add in the store tail is a chain
  //      of 16 latency-bound 16-byte loads per lane with nothing to overlap.  Instead, K chunk c (c < MREP/2) requests the
  //      residual of fragments 2c, 2c+1 before its MFMAs (16 VGPRs) and adds it to those accumulators afterwards: the loads
  //      fly under ~150 MFMAs (a 16-byte run = quads 2pr, 2pr+1 of the lane).  Measured: +1.3 % (128 ch) ... +2.7 %
  //      (512 ch) on the conv2 layers.
  constexpr bool RES_PRE = (KT == 1 && KG == 1 && UPS == 0 && MREP % 2 == 0 && XP == 0);
  constexpr int NFR = NB * MREP;  // accumulator fragments of a wave; unit u = n * MREP + r
  const bool res_pre = RES_PRE && p.res != nullptr && p.res_pre != 0 && p.nchunks >= NFR / 2;
  uint4 rpre[2][2];

  // ---- pipeline
  CVVAE_PROBE_MARK();
  if constexpr (LD) {
    dma_from(inp, (size_t)p.in_ps, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    stage(0, 0);
  }
  CVVAE_PROBE_MARK();
  __syncthreads();
  for (int c = 0; c < p.nchunks; ++c) {
    const int cur = c & 1;
    const bool more = (c + 1) < p.nchunks;
    CVVAE_PROBE_MARK();
    const bool stage_first = !LD && grp == 0 && !p.phase_sync;
    if (stage_first && more) stage(c + 1, cur ^ 1);
    // LD: every wave requests its share of the next chunk's halo and goes on multiplying this one.  WHERE in the chunk: vector
    // memory operations complete in order, so the first weight record requested after the wave-loads cannot be consumed before they
    // have landed -- and the wait for this chunk's first records sits in front of its first MFMA (with a ring as deep as a time
    // group it is a wait for everything outstanding).  The wave-loads therefore go out right AFTER the MFMAs of the chunk's first
    // k16 step (LD_ISSUE below), not here; a wave without output channels has no such step
    if (LD && more && !active) dma_from(inp, (size_t)p.in_ps, c + 1, cur ^ 1);
#define CVVAE_LD_ISSUE(first_step)                                             \
    if constexpr (LD) {                                                          \
      if (first_step) {                                                          \
        __builtin_amdgcn_sched_barrier(0);                                       \
        if (more) dma_from(inp, (size_t)p.in_ps, c + 1, cur ^ 1);                \
      }                                                                          \
    }
    CVVAE_PROBE_MARK();
    if constexpr (XP >= 2) {
      // ---- fast fp32: per pair of taps (a, b) of a run of R = KH*KW taps:  Whi.hi (a), Whi.hi (b) on the fp16 MFMA, then both
      //      correction terms of both taps on ONE bf8 K = 64 MFMA.  A run with an odd tap count ends with a half-empty pair (the
      //      packer zero-fills its second half; the B operand repeats tap a so that 0 x finite = 0).  LDS fragments and weight
      //      records of the next sub-step are requested while the MFMAs of this one issue, as in the loops below.
      if constexpr (G::PL) {
        // ---- the same walk on the PLANAR layout with eight fragments per wave (Geo::PL).  One operand slot per fragment, updated IN
        //      PLACE: the LDS read for the next sub-step's fragment r is issued right after the MFMA that consumed fragment r and is
        //      needed eight MFMAs (>= 256 cycles) later -- fp16 tap a -> fp16 tap b -> the 24 bytes of codes -> next pair's tap a all
        //      pass through the same six registers, so the three operand kinds of a pair cost 48 registers per wave instead of 112
        //      beside the 128 accumulators.  With that few registers to spare, hipcc has to be held to the plan: (1) a slot is ONE
        //      six-dword value whose parts are replaced (and which an empty asm keeps whole), so a refill cannot be given fresh
        //      registers; (2) a scheduling fence after every MFMA + refill keeps the refill behind the MFMA that reads the slot;
        //      (3) every LDS address is formed right at its read by a volatile one-instruction asm from ONE per-fragment array (the
        //      pixel's offset in 8-byte units; the 16-byte planes use (x << 1) + base) -- left to itself the compiler hoists the
        //      loop-invariant sums (fragment offset + the lane's tap select) of all pairs out of the chunk loop, ~80 registers;
        //      (4) the weight records are addressed as a wave-uniform base (SGPRs) + the lane's 32-bit byte offset.
        if (active) {
          constexpr int R = KH * KW, PR = (R + 1) / 2, NPAIR = PR * KSUB;
          const unsigned lb = (unsigned)(cur * G::BUFB);
          const long long wcb = wqu + (long long)c * wq_cs;  // (element offsets from p.w)
          const long long wnx = more ? wcb + wq_cs : wcb;    // the last chunk's read-ahead re-reads its own records
          const int ngq = TFOLD ? tf_ng : 1;
          i32x4_t s4[MREP];
          i32x2_t s2[MREP];
#define CVVAE_LDW(eo) (*reinterpret_cast<const v8*>(reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.w) + (eo)) + wlane))  /* eo: wave-uniform element offset */
          // refill the low four dwords (an fp16 fragment) / all six (codes) of slot r  (macros, not lambdas: one more level of closures
          // inside the time-group loop and the optimiser leaves every captured array -- the accumulators included -- in scratch)
#define CVVAE_PUT4(r, ADDR) s4[r] = *reinterpret_cast<const i32x4_t*>(&smem[ADDR])
#define CVVAE_PUT6(r, ADDRA, ADDRB)                                                              \
          do {                                                                                   \
            s4[r] = *reinterpret_cast<const i32x4_t*>(&smem[ADDRA]);                             \
            s2[r] = *reinterpret_cast<const i32x2_t*>(&smem[ADDRB]);                             \
          } while (0)
#define CVVAE_LO4(r) s4[r]
          const unsigned khb = (unsigned)((lane >> 5) * G::KHB);  // the fp16 k split: lanes 32-63 read the next plane
          // the lane's tap inside a pair: lanes 0-31 read the codes of tap a, lanes 32-63 of tap b = a fixed pixel distance further
          // (+1 in a row, or -- the pair that wraps to the next kernel row -- + FW - (KW - 1)): two per-lane constants
          const unsigned hi1 = (unsigned)((lane >> 5) * 16), hiw = (unsigned)((lane >> 5) * (G::FW - (KW - 1)) * 16);
          {
            const unsigned b0 = lb + (TFOLD ? tf_l0 : 0u) + khb;
#pragma unroll
            for (int r = 0; r < MREP; ++r) {
              CVVAE_PUT4(r, pl_add2x(aoff8[r], b0));
            }
          }
          // one time group: LDS frame offset fo / weight slot ws of this group, fon / wsn of the next one (values, not `g == 0 ? .. : ..`
          // over the captured plan: hipcc turns that into a load through a SELECTED ADDRESS inside the closure object, which then
          // -- and with it every array it refers to, the accumulators included -- stays in scratch memory)
          auto group_body = [&](const unsigned fo, const long long ws, const unsigned fon, const long long wsn, const bool lastg) __attribute__((always_inline)) {
            const unsigned lbg = lb + fo + khb;                       // hi planes: 16 bytes per pixel
            const unsigned lba = lb + fo + (unsigned)G::CQA;          // code plane A: 16 bytes per pixel, read by the lane's own tap
            const unsigned lbh = lb + (fo >> 1) + (unsigned)G::CQB;   // code plane B: 8 bytes per pixel
            const unsigned lba1 = lba + hi1, lbaw = lba + hiw, lbh1 = lbh + (hi1 >> 1), lbhw = lbh + (hiw >> 1);
            const long long wg = pl_uniform(wcb + ws);
            const long long wn = pl_uniform(lastg ? wnx + wsn : wcb + wsn);  // pair 0 of the next group / of the next chunk's first group
            const unsigned lbn = lb + fon + khb;
            static_for<NPAIR>([&](auto q_tag) __attribute__((always_inline)) {
              constexpr int q = decltype(q_tag)::value;
              constexpr int ks = q / PR, ta = 2 * (q % PR), tb = ta + 1;
              constexpr bool hasb = tb < R, lastq = q + 1 == NPAIR;
              constexpr unsigned pa = (unsigned)((ta / KW) * G::FW + (ta % KW)), pb = (unsigned)((tb / KW) * G::FW + (tb % KW));
              static_assert(!hasb || pb - pa == 1 || pb - pa == (unsigned)(G::FW - (KW - 1)), "tap b of a pair: next in the row, or first of the next row");
              constexpr unsigned ob = pb * 16 + ks * G::KSB;  // fp16 fragment of tap b
              constexpr unsigned oca = pa * 16 + ks * G::PLB, och = pa * 8 + ks * (G::PLB / 2);  // codes of tap a
              constexpr int nks = (q + 1) / PR, nta = 2 * ((q + 1) % PR);
              constexpr unsigned ona = (unsigned)(((nta / KW) * G::FW + (nta % KW)) * 16 + nks * G::KSB);
              constexpr int qf = q + XQD;
              constexpr bool wrapf = qf >= NPAIR;
              constexpr int qw = qf - NPAIR;
              constexpr int qq = !wrapf ? qf : ((XQD == 2 && (NPAIR & 1)) ? 1 - qw : qw);
              constexpr int fks = qq / PR, fta = 2 * (qq % PR);
              const long long ne = (wrapf ? wn : wg) + (long long)fks * wq_ks + fta * (3 * 512);  // its record [0]
              constexpr bool nhasb = fta + 1 < R;
              constexpr int S = XQD == 2 ? 4 * (q & 1) : 0;  // this pair's ring set
              // my tap's pixel: tap a + (lanes 32-63 of a full pair) the distance to tap b
              const unsigned ba = !hasb ? lba : (pb - pa == 1 ? lba1 : lbaw), bh = !hasb ? lbh : (pb - pa == 1 ? lbh1 : lbhw);
#define CVVAE_CODES(r) CVVAE_PUT6(r, pl_add2x(aoff8[r], ba) + oca, pl_addv(aoff8[r], bh) + och)
              // Whi.hi of tap a
#pragma unroll
              for (int r = 0; r < MREP; ++r) {
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[n * MREP + r] = Tr<T>::mfma(wf[n][S + 0], __builtin_bit_cast(v8, CVVAE_LO4(r)), acc[n * MREP + r]);
                if (hasb) CVVAE_PUT4(r, pl_add2x(aoff8[r], lbg) + ob);
                else CVVAE_CODES(r);
                if (r == MREP - 1) {
#pragma unroll
                  for (int n = 0; n < NB; ++n) wf[n][S + 0] = CVVAE_LDW(ne + n * wqx_nbs);
                }
                __builtin_amdgcn_sched_barrier(0);
              }
              if (hasb) {  // Whi.hi of tap b
#pragma unroll
                for (int r = 0; r < MREP; ++r) {
#pragma unroll
                  for (int n = 0; n < NB; ++n) acc[n * MREP + r] = Tr<T>::mfma(wf[n][S + 3], __builtin_bit_cast(v8, CVVAE_LO4(r)), acc[n * MREP + r]);
                  CVVAE_CODES(r);
                  if (r == MREP - 1) {
#pragma unroll
                    for (int n = 0; n < NB; ++n) wf[n][S + 3] = CVVAE_LDW(ne + (nhasb ? 3 * 512 : 0) + n * wqx_nbs);
                  }
                  __builtin_amdgcn_sched_barrier(0);
                }
              } else {
#pragma unroll
                for (int n = 0; n < NB; ++n) wf[n][S + 3] = CVVAE_LDW(ne + (nhasb ? 3 * 512 : 0) + n * wqx_nbs);
              }
              // q(Whi).q(lo) + q(Wlo).q(hi) of both taps on the block-scaled bf6 K = 64 MFMA
#pragma unroll
              for (int r = 0; r < MREP; ++r) {
#pragma unroll
                for (int n = 0; n < NB; ++n)
                  acc[n * MREP + r] = mfma_bf6_k64(__builtin_bit_cast(i32x4_t, wf[n][S + 1]), __builtin_bit_cast(i32x4_t, wf[n][S + 2]), CVVAE_LO4(r), s2[r], q6_eb, acc[n * MREP + r]);
                if (!lastq) CVVAE_PUT4(r, pl_add2x(aoff8[r], lbg) + ona);
                else if (!lastg) CVVAE_PUT4(r, pl_add2x(aoff8[r], lbn));
                if (r == MREP - 1) {
#pragma unroll
                  for (int n = 0; n < NB; ++n) {
                    wf[n][S + 1] = CVVAE_LDW(ne + 512 + n * wqx_nbs);
                    wf[n][S + 2] = CVVAE_LDW(ne + 1024 + n * wqx_nbs);
                  }
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            });
          };
          if constexpr (TFOLD) {
            unsigned fo = tf_l0, fon = tf_l1;
            long long ws = tf_w0, wsn = tf_w1;
            for (int g = 0; g < ngq; ++g) {
              const bool lastg = g + 1 == ngq;
              group_body(fo, ws, fon, lastg ? tf_w0 : wsn, lastg);
              fo = fon; ws = wsn;
              fon = tf_l2; wsn = tf_w2;
            }
          } else {
            group_body(0u, 0ll, 0u, 0ll, true);
          }
#undef CVVAE_PUT4
#undef CVVAE_LDW
#undef CVVAE_PUT6
#undef CVVAE_LO4
#undef CVVAE_CODES
        }
      } else
      if (active) {
        constexpr int R = KH * KW, PR = (R + 1) / 2, NPAIR = PR * KSUB;
        const unsigned lb = (unsigned)(cur * G::BUFB);
        const T* wcb = wqx + (size_t)c * (size_t)wq_cs;
        const T* wnx = more ? wcb + wq_cs : wcb;  // the last chunk's read-ahead re-reads its own records
        const int ngq = TFOLD ? tf_ng : 1;
        v8 fa[MREP], fb[MREP], qa[MREP], qb[MREP];
        // XP == 3: ONE bf6 operand per fragment -- 24 bytes of the LANE's tap (lanes 0-31: tap a, lanes 32-63: tap b of the pair):
        // [lo * 2^11 | hi] codes of the pixel's 16 channels
        i32x4_t q6a[XP == 3 ? MREP : 1];
        i32x2_t q6b[XP == 3 ? MREP : 1];
        const unsigned aq6 = XP == 3 ? 32u - (unsigned)((lane >> 5) * 16) : 0u;  // (aoff[] carries the fp16 k split: undo it)
        const bool lhi = lane >= 32;
#pragma unroll
        for (int r = 0; r < MREP; ++r) fa[r] = *reinterpret_cast<const v8*>(&smem[lb + (TFOLD ? tf_l0 : 0u) + aoff[r]]);
        auto group_body = [&](int g) __attribute__((always_inline)) {
          const unsigned lbg = lb + (TFOLD ? (g == 0 ? tf_l0 : (g == 1 ? tf_l1 : tf_l2)) : 0u);
          const T* wg = wcb + (TFOLD ? (g == 0 ? tf_w0 : (g == 1 ? tf_w1 : tf_w2)) : 0);
          const bool lastg = g + 1 == ngq;
          const T* wn = lastg ? wnx + (TFOLD ? tf_w0 : 0) : wcb + (g == 0 ? tf_w1 : tf_w2);  // pair 0 of the next group / chunk
          const unsigned lbn = lb + (g == 0 ? tf_l1 : tf_l2);
          static_for<NPAIR>([&](auto q_tag) __attribute__((always_inline)) {
            constexpr int q = decltype(q_tag)::value;
            constexpr int ks = q / PR, ta = 2 * (q % PR), tb = ta + 1;
            constexpr bool hasb = tb < R, lastq = q + 1 == NPAIR;
            const unsigned oa = (unsigned)(((ta / KW) * G::FW + (ta % KW)) * PIXB + ks * 64);
            const unsigned ob = (unsigned)(((tb / KW) * G::FW + (tb % KW)) * PIXB + ks * 64);
            const int nks = (q + 1) / PR, nta = 2 * ((q + 1) % PR);
            const unsigned ona = (unsigned)(((nta / KW) * G::FW + (nta % KW)) * PIXB + nks * 64);
            // The pair whose records refill this pair's ring slots: XQD pairs ahead, in this group or the next one.  Two pairs in
            // flight: even pairs live in ring set 0, odd pairs in set 1.  A group with an ODD pair count ends on an even pair, and
            // the next group starts on one: its last two pairs swap their refills (the second-to-last fetches the next group's
            // pair 1, three pairs ahead; the last one the next group's pair 0, one pair ahead) -- the sets stay static.
            constexpr int qf = q + XQD;
            constexpr bool wrapf = qf >= NPAIR;
            constexpr int qw = qf - NPAIR;
            constexpr int qq = !wrapf ? qf : ((XQD == 2 && (NPAIR & 1)) ? 1 - qw : qw);
            constexpr int fks = qq / PR, fta = 2 * (qq % PR);
            const T* ne = (wrapf ? wn : wg) + (long long)fks * wq_ks + fta * (3 * 512);  // its record [0]
            constexpr bool nhasb = fta + 1 < R;
            constexpr int S = XQD == 2 ? 4 * (q & 1) : 0;  // this pair's ring set
            const unsigned oq6 = (hasb && lhi ? ob : oa) + aq6;  // XP == 3: my tap's pixel (an unpaired tap: both halves read tap a)
            // (requesting the LDS fragments two sub-steps ahead instead of one -- correction operands during the tap-a products, the
            //  next pair's tap a during tap b, its tap b during the corrections -- measured 3-6 % SLOWER: the loop does not wait on LDS)
            // Whi.hi of tap a
#pragma unroll
            for (int r = 0; r < MREP; ++r) {
              acc[r] = Tr<T>::mfma(wf[0][S + 0], fa[r], acc[r]);
              if (hasb) fb[r] = *reinterpret_cast<const v8*>(&smem[lbg + aoff[r] + ob]);
              else if constexpr (XP == 3) {
                q6a[r] = *reinterpret_cast<const i32x4_t*>(&smem[lbg + aoff[r] + oq6]);
                q6b[r] = *reinterpret_cast<const i32x2_t*>(&smem[lbg + aoff[r] + oq6 + 16]);
              } else qa[r] = *reinterpret_cast<const v8*>(&smem[lbg + aoff[r] + oa + 32]);
            }
            wf[0][S + 0] = *reinterpret_cast<const v8*>(ne);
            __builtin_amdgcn_sched_barrier(0);
            if (hasb) {  // Whi.hi of tap b
#pragma unroll
              for (int r = 0; r < MREP; ++r) {
                acc[r] = Tr<T>::mfma(wf[0][S + 3], fb[r], acc[r]);
                if constexpr (XP == 3) {
                  q6a[r] = *reinterpret_cast<const i32x4_t*>(&smem[lbg + aoff[r] + oq6]);
                  q6b[r] = *reinterpret_cast<const i32x2_t*>(&smem[lbg + aoff[r] + oq6 + 16]);
                } else {
                  qa[r] = *reinterpret_cast<const v8*>(&smem[lbg + aoff[r] + oa + 32]);
                  qb[r] = *reinterpret_cast<const v8*>(&smem[lbg + aoff[r] + ob + 32]);
                }
              }
            }
            wf[0][S + 3] = *reinterpret_cast<const v8*>(ne + (nhasb ? 3 * 512 : 0));
            __builtin_amdgcn_sched_barrier(0);
            // q(Whi).q(lo) + q(Wlo).q(hi) of both taps (bf8: K = 64 at twice the fp16 rate; bf6: four times)
#pragma unroll
            for (int r = 0; r < MREP; ++r) {
              if constexpr (XP == 3)
                acc[r] = mfma_bf6_k64(__builtin_bit_cast(i32x4_t, wf[0][S + 1]), __builtin_bit_cast(i32x4_t, wf[0][S + 2]), q6a[r], q6b[r], q6_eb, acc[r]);
              else
                acc[r] = mfma_bf8_k64(wf[0][S + 1], wf[0][S + 2], qa[r], hasb ? qb[r] : qa[r], acc[r]);
              if (!lastq) fa[r] = *reinterpret_cast<const v8*>(&smem[lbg + aoff[r] + ona]);
              else if (!lastg) fa[r] = *reinterpret_cast<const v8*>(&smem[lbn + aoff[r]]);
            }
            wf[0][S + 1] = *reinterpret_cast<const v8*>(ne + 512);
            wf[0][S + 2] = *reinterpret_cast<const v8*>(ne + 1024);
            __builtin_amdgcn_sched_barrier(0);
          });
        };
        for (int g = 0; g < ngq; ++g) group_body(g);
      }
    } else if constexpr (TFOLD) {
      if (active) {
        const unsigned lb = (unsigned)(cur * G::BUFB);
        const T* wcb = wq + (size_t)c * (size_t)w_cs;
        const T* wnx = more ? wcb + w_cs : wcb;  // the last chunk's read-ahead re-reads its own records (never past the buffer)
        constexpr int NAB = G::NAB;
        v8 ab[NAB][MREP];
#pragma unroll
        for (int r = 0; r < MREP; ++r) ab[0][r] = *reinterpret_cast<const v8*>(&smem[lb + tf_l0 + aoff[r]]);
        for (int g = 0; g < tf_ng; ++g) {
          const unsigned lbg = lb + (g == 0 ? tf_l0 : (g == 1 ? tf_l1 : tf_l2));
          const T* wg = wcb + (g == 0 ? tf_w0 : (g == 1 ? tf_w1 : tf_w2));
          const bool lastg = g + 1 == tf_ng;
          const T* wn = lastg ? wnx + tf_w0 : wcb + (g == 0 ? tf_w1 : tf_w2);  // where the weight ring continues
          const unsigned lbn = lb + (g == 0 ? tf_l1 : tf_l2);                   // LDS frame of the next group
#pragma unroll
          for (int i = 0; i < GS; ++i) {
            const int nj = (i + 1) / XPM, npart = (i + 1) % XPM;
            const int nsp = nj % NSP, nks = nj / NSP;
            const int noff = ((nsp / KW) * G::FW + (nsp % KW)) * PIXB + nks * G::KSB + (npart == 2 ? 16 : 0);
#pragma unroll
            for (int n = 0; n < NB; ++n) wwait(wf[n][i % PF]);
#pragma unroll
            for (int r = 0; r < MREP; ++r) {
#pragma unroll
              for (int n = 0; n < NB; ++n) acc[n * MREP + r] = Tr<T>::mfma(wf[n][i % PF], ab[i & (NAB - 1)][r], acc[n * MREP + r]);
              if (i + 1 < GS)
                ab[(i + 1) & (NAB - 1)][r] = *reinterpret_cast<const v8*>(&smem[lbg + aoff[r] + (npart ? lhi16 : 0u) + (unsigned)noff]);
              else if (NAB == 1 && !lastg)  // (in place) first fragments of the next time group
                ab[0][r] = *reinterpret_cast<const v8*>(&smem[lbn + aoff[r]]);
            }
            if (NAB == 2 && i + 1 == GS && !lastg) {  // first fragments of the next time group (set 0: its step 0)
#pragma unroll
              for (int r = 0; r < MREP; ++r) ab[0][r] = *reinterpret_cast<const v8*>(&smem[lbn + aoff[r]]);
            }
            CVVAE_LD_ISSUE(i == 0 && g == 0)
#pragma unroll
            for (int n = 0; n < NB; ++n)
              wload(wf[n][i % PF], (i + PF < GS ? wg + tf_rec(i + PF) : wn + tf_rec(i + PF - GS)) + n * wq_nbs);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    } else if (active) {
      const unsigned lb = (unsigned)(cur * G::BUFB);  // 32-bit LDS offsets throughout (no 64-bit address math)
      const T* wc = wq + (size_t)c * (STEPS * 512);
      // software-pipelined LDS reads: the activation fragments of step st+1 are requested while the MFMAs of
      // step st issue (two register sets), so an MFMA never waits on the ds_read issued right before it.
      constexpr int NAB = G::NAB;
      v8 ab[NAB][MREP];
#pragma unroll
      for (int r = 0; r < MREP; ++r) ab[0][r] = *reinterpret_cast<const v8*>(&smem[lb + aoff[r]]);
      if constexpr (RES_PRE) {
        if (res_pre && c < NFR / 2) {
          int lane_p = lane;  // opaque copy: keeps the address math inside this branch (not hoisted into loop-long VGPRs)
          asm volatile("" : "+v"(lane_p));
          const int nbr = nb0 + (2 * c) / MREP;  // (MREP is even: both fragments of chunk c lie in one N-block)
#pragma unroll
          for (int ri = 0; ri < 2; ++ri) {
            const int m = (wave_m * MREP + (2 * c) % MREP + ri) * 32 + (lane_p & 31);
            const int tx = m % TW, ty = (m / TW) % TH, tt = m / (TW * TH);
            const int to = t0 + tt, yo = y0 + ty, xo = x0 + tx;
            const bool in = to < p.To && yo < p.Ho && xo < p.Wo;  // lanes outside read pixel 0; their sums are never stored
            const long long pix = in ? (((long long)b * p.To + to) * p.Ho + yo) * p.Wo + xo : 0;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
              rpre[ri][pr] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.res) + pix * (long long)p.out_ps +
                                                             (nbr * 32 + pr * 16 + (lane_p >> 5) * 8));
          }
        }
      }
#pragma unroll
      for (int st = 0; st < STEPS_W; ++st) {
        const int nq = (st + 1) / XPM, npart = (st + 1) % XPM;
        const int nks = nq / NTAPS, nt = nq % NTAPS;  // next step's k-sub-chunk (within my K-group) / tap
        const int ndt = nt / (KH * KW), ndy = (nt / KW) % KH, ndx = nt % KW;
        const int noff = ((ndt * G::FH + ndy) * G::FW + ndx) * PIXB + nks * G::KSB + (npart == 2 ? 16 : 0);
#pragma unroll
        for (int n = 0; n < NB; ++n) wwait(wf[n][st % PF]);
#pragma unroll
        for (int r = 0; r < MREP; ++r) {
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[n * MREP + r] = Tr<T>::mfma(wf[n][st % PF], ab[st & (NAB - 1)][r], acc[n * MREP + r]);
          if (st + 1 < STEPS_W)
            ab[(st + 1) & (NAB - 1)][r] = *reinterpret_cast<const v8*>(&smem[lb + aoff[r] + (npart ? lhi16 : 0u) + (unsigned)noff]);
        }
        CVVAE_LD_ISSUE(st == 0)
        // ring refill: my record st+PF of this chunk, or (wrapping) record st+PF-STEPS_W of the next chunk
#pragma unroll
        for (int n = 0; n < NB; ++n)
          wload(wf[n][st % PF], wc + (st + PF < STEPS_W ? st + PF : st + PF - STEPS_W + STEPS) * 512 + n * wq_nbs);
        // Fence per step: keeps the next step's ds_reads and the weight prefetch inside THIS step.  hipcc otherwise
        // sinks every load to just before its first use, which exposes the LDS / L2 latency once per MFMA
        // (measured on MI355X: 1158 -> 1240 TFLOP/s on 256->256 @9x256^2; pinning a strict MFMA/ds_read
        // alternation with sched_group_barrier instead was 4 % slower than letting hipcc order the step).
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (RES_PRE) {
        if (res_pre && c < NFR / 2) {
#pragma unroll
          for (int cc = 0; cc < NFR / 2; ++cc) {  // static accumulator indices (a runtime-indexed array would go to scratch)
            if (c != cc) continue;
#pragma unroll
            for (int ri = 0; ri < 2; ++ri)
#pragma unroll
              for (int pr = 0; pr < 2; ++pr) {
                float rf[8];
                unpack8<T>(rpre[ri][pr], rf);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  acc[2 * cc + ri][(2 * pr) * 4 + j] += rf[j];
                  acc[2 * cc + ri][(2 * pr + 1) * 4 + j] += rf[4 + j];
                }
              }
          }
        }
      }
    }
#undef CVVAE_LD_ISSUE
    CVVAE_PROBE_MARK();
    if (!LD && !stage_first && more) stage(c + 1, cur ^ 1);
    if constexpr (LD) {
      // my wave-loads of the next chunk were issued BEFORE this chunk's weight-record requests, and vector memory operations
      // complete in order: once all but the PF youngest requests (the ring's read-ahead) are back, so are they.  A wave without
      // output channels requested nothing after them: it waits for everything.
      if (more) {
        if (active) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    CVVAE_PROBE_MARK();
    // (LD: the plain barrier instruction.  __syncthreads() carries workgroup fences, and with LDS-DMA writes possibly pending hipcc
    //  turns them into s_waitcnt vmcnt(0) -- which would also drain the weight ring's read-ahead at every chunk.  What the barrier
    //  needs is exactly what the two waits give: my wave-loads have landed, my LDS reads of this chunk are back)
    if constexpr (LD) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
  }
  // LD: the ring's read-ahead of the last chunk is still in flight, requested by asm statements hipcc knows nothing about: once the
  // loop is over it would hand the ring's registers to the epilogue (addresses, bias) and the late records would land on top of them
  if constexpr (LD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- fused 1x1 shortcut: more K chunks over the second input through the centre tap (the final barrier of the loop
  //      above has released both halo buffers)
  if constexpr (KT == 1 && KH == 3 && KW == 3 && ST == 1 && SH == 1 && SW == 1 && KG == 1 && UPS == 0 && XP < 2) {
    if (p.in2 != nullptr) {
      const TIO* __restrict__ inp2 = reinterpret_cast<const TIO*>(p.in2);
      auto stage2 = [&](int chunk, int bufsel) {
        if constexpr (LD) dma_from(inp2, (size_t)p.in2_ps, chunk, bufsel);
        else stage_from(std::integral_constant<int, 0>{}, inp2, (size_t)p.in2_ps, chunk, bufsel);
      };
      constexpr unsigned ctr = (unsigned)((1 * G::FW + 1) * PIXB);  // centre tap (dy = dx = 1)
      const T* w2q = reinterpret_cast<const T*>(p.w2) + (size_t)(active ? nb : 0) * (size_t)p.nchunks2 * (KSUB * XPM * 512) + lane * 8;
      stage2(0, 0);
      if constexpr (LD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int c = 0; c < p.nchunks2; ++c) {
        const int cur = c & 1;
        const bool more = (c + 1) < p.nchunks2;
        const bool stage_first = LD || (grp == 0 && !p.phase_sync);
        if (stage_first && more) stage2(c + 1, cur ^ 1);
        if (active) {
          const unsigned lb = (unsigned)(cur * G::BUFB);
          v8 wv[NB][KSUB * XPM];
#pragma unroll
          for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int ks = 0; ks < KSUB * XPM; ++ks)
              wv[n][ks] = *reinterpret_cast<const v8*>(w2q + (size_t)n * (size_t)p.nchunks2 * (KSUB * XPM * 512) +
                                                       ((size_t)c * KSUB * XPM + ks) * 512);
#pragma unroll
          for (int ks = 0; ks < KSUB; ++ks)
#pragma unroll
            for (int part = 0; part < XPM; ++part)
#pragma unroll
              for (int r = 0; r < MREP; ++r) {
                const v8 bf = *reinterpret_cast<const v8*>(&smem[lb + aoff[r] + (part ? lhi16 : 0u) + ctr +
                                                                 ks * G::KSB + (part == 2 ? 16 : 0)]);
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[n * MREP + r] = Tr<T>::mfma(wv[n][ks * XPM + part], bf, acc[n * MREP + r]);
              }
        }
        if (!stage_first && more) stage2(c + 1, cur ^ 1);
        if constexpr (LD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the 1x1 weights of a chunk are requested at its top)
        __syncthreads();
      }
    }
  }
  CVVAE_PROBE_MARK();
  // ---- K-group reduction (KG == 2): group 0 keeps fragments [0,H) and parks [H,MREP) in LDS, group 1 the opposite;
  //      after one barrier each adds its partner's parked half (same (wave_m, wave_n) slot of the other group).
  //      The K loop ended on a barrier, so the halo buffers are dead.  Static fragment indices only (a runtime-indexed
  //      accumulator array would go to scratch), hence the two wave-uniform branches.
  if constexpr (KG == 2) {
    constexpr int H = MREP / 2;
    const int wslot = wave & 3;
    float4* park = reinterpret_cast<float4*>(smem) + (size_t)((kgrp * 4 + wslot) * H * 4) * 64 + lane;
    if (active) {
      if (kgrp == 0) {
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            park[(h * 4 + q) * 64] = make_float4(acc[H + h][q * 4], acc[H + h][q * 4 + 1], acc[H + h][q * 4 + 2], acc[H + h][q * 4 + 3]);
      } else {
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            park[(h * 4 + q) * 64] = make_float4(acc[h][q * 4], acc[h][q * 4 + 1], acc[h][q * 4 + 2], acc[h][q * 4 + 3]);
      }
    }
    __syncthreads();
    const float4* take = reinterpret_cast<const float4*>(smem) + (size_t)(((kgrp ^ 1) * 4 + wslot) * H * 4) * 64 + lane;
    if (active) {
      if (kgrp == 0) {
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = take[(h * 4 + q) * 64];
            acc[h][q * 4] += v.x; acc[h][q * 4 + 1] += v.y; acc[h][q * 4 + 2] += v.z; acc[h][q * 4 + 3] += v.w;
          }
      } else {
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = take[(h * 4 + q) * 64];
            acc[H + h][q * 4] += v.x; acc[H + h][q * 4 + 1] += v.y; acc[H + h][q * 4 + 2] += v.z; acc[H + h][q * 4 + 3] += v.w;
          }
      }
    }
  }
  if (!active) return;
  // The epilogue runs once per N-block of the wave (NB = 2: the second pass re-derives the channel terms; the pixel terms fold)
  auto epilogue = [&](auto ni_tag) {
    constexpr int NI = decltype(ni_tag)::value;
    const int nb = nb0 + NI;

  // ---- epilogue.  Accumulator layout: lane = one pixel, register quad g = 4 consecutive output channels
  //      (lanes 0-31: channels 8g..8g+3, lanes 32-63: channels 8g+4..8g+7 of the wave's 32-channel block).
  // (the opaque copy of `lane` stops hipcc from hoisting the output addresses above the K loop, where they
  //  would sit in VGPRs for the whole kernel)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int C2 = p.Cout >> 1;
  if (p.out_mode == 1) {  // NCDHW, dtype T (conv_out: few channels, scattered 2-byte stores)
#pragma unroll
    for (int r = 0; r < MREP; ++r) {
      if (KG == 2 && ((r < MREP / 2) != (kgrp == 0))) continue;  // each K-group stores the half it reduced
      const int m = (wave_m * MREP + r) * 32 + (lane_e & 31);
      const int tx = m % TW, ty = (m / TW) % TH, tt = m / (TW * TH);
      const int to = t0 + tt, yo = y0 + ty, xo = x0 + tx;
      if (to >= p.To || yo >= p.Ho || xo >= p.Wo) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = nb * 32 + (g >> 1) * 16 + (lane_e >> 5) * 8 + (g & 1) * 4;
        if (cb >= p.Cout) continue;
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + cb);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
        TIO* o = reinterpret_cast<TIO*>(p.out);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (cb + j < p.Cout)
            o[((((size_t)b * p.Cout + (cb + j)) * p.To + to) * p.Ho + yo) * (size_t)p.Wo + xo] =
                (TIO)(bias_pre ? acc[NI * MREP + r][g * 4 + j] : acc[NI * MREP + r][g * 4 + j] * p.alpha + bb[j]);
      }
    }
    CVVAE_PROBE_MARK();
    return;
  }
  // NDHWC / time-shuffle: for each PAIR of quads (2pr, 2pr+1) the lower half-wave holds channels 16*pr..+7 and the upper
  // one 16*pr+8..+15 of its pixel (sigma row order of the packed weights) -- 8 consecutive channels per lane -> ONE 16-byte
  // store (and one 16-byte residual load) per pair.  (An earlier revision kept natural row order and exchanged quads
  // between the half-waves with v_permlane32_swap in the tail: 64 swaps + hazard nops + copies per lane, ~1.3 k cycles per
  // tile and 3 % of a per-frame conv.)
  // fused GroupNorm statistics of what is stored (the ROUNDED values, as the reference's GroupNorm sees them): per lane
  // 4 slots (pr, q) of 4 consecutive channels each: sum, sum of squares; valid-pixel count per pr
  // SHIFTED sums (numerics: sum x^2 - (sum x)^2 / n cancels as (mean/sigma)^2 * eps): every slot accumulates x - K and (x - K)^2
  // with K = the slot's first stored value of lane 0 of the half-wave (one v_readlane per slot and tile, one v_sub per value),
  // so the cancellation is governed by (mean - K) / sigma = O(1) instead of mean / sigma.
  float gs[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, gq[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, gc[2] = {0.f, 0.f};
  float gk[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  bool gk_set = false;  // fast tail (wave-uniform)
  unsigned gkm = 0;     // general tail (per lane): bit pr = the shifts of channel pair pr are set
  // per channel pair pr (tile independent): my 8-channel run, its bias, where it lands
  int c8v[2], ccv[2], nshv[2];
  float bia[2][8];
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    const int c8 = nb * 32 + pr * 16 + (lane_e >> 5) * 8;  // my 8 consecutive output channels (bias is padded to 32)
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + c8);
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + c8 + 4);
    bia[pr][0] = b0.x; bia[pr][1] = b0.y; bia[pr][2] = b0.z; bia[pr][3] = b0.w;
    bia[pr][4] = b1.x; bia[pr][5] = b1.y; bia[pr][6] = b1.z; bia[pr][7] = b1.w;
    const int n = (p.out_mode == 2 && c8 >= C2) ? 1 : 0;  // channel -> time shuffle (C2 % 8 == 0: a run never straddles)
    c8v[pr] = c8;
    nshv[pr] = n;
    ccv[pr] = c8 - n * C2;
  }
  // Fragments go through the epilogue in batches of RB: first every output offset of the batch is computed and ALL its
  // residual loads are issued (one dependent load -> add -> store chain per fragment cost 25 k cycles per tile on the
  // ResnetBlock conv2 layers), then the math and the stores follow.
  // Output pixel of a fragment for this lane, computed ONCE per fragment (it cost ~20 VALU ops per call when it was redone
  // for every channel pair and pass): the pixel of the n = 0 channel half, or NOPIX when the lane is outside the tensor.
  // 32-bit: the host checks the pixel count.  Time-shuffle stores: the n = 1 half lands one frame later (+ frame_px), and the
  // n = 0 half of conv frame 0 is the dropped frame -1.
  constexpr int NOPIX = -2147483647 - 1;
  const int frame_px = (UPS == 2 ? 4 : 1) * p.Ho * p.Wo;
  auto locate = [&](int r, bool& first_frame) -> int {
    const int m = (wave_m * MREP + r) * 32 + (lane_e & 31);
    const int tx = m % TW, ty = (m / TW) % TH, tt = m / (TW * TH);
    const int to = t0 + tt, yo = y0 + ty, xo = x0 + tx;
    first_frame = to == 0;
    const int tq = p.out_mode == 2 ? 2 * to - 1 : to, Tq = p.out_mode == 2 ? 2 * p.To - 1 : p.To;
    const int pix = UPS == 2 ? (((b * Tq + tq) * (2 * p.Ho) + (2 * yo + py)) * (2 * p.Wo) + (2 * xo + px))
                             : (((b * Tq + tq) * p.Ho + yo) * p.Wo + xo);
    return (to < p.To && yo < p.Ho && xo < p.Wo) ? pix : NOPIX;
  };
  // pixel of channel pair pr given the fragment's pixel, or -1 when there is nothing to store
  auto pix_of = [&](int pixr, bool first_frame, int pr) -> int {
    const bool valid = pixr != NOPIX && c8v[pr] < p.Cout && !(p.out_mode == 2 && first_frame && nshv[pr] == 0);
    return valid ? pixr + nshv[pr] * frame_px : -1;
  };
  constexpr int RB = MREP >= 4 ? 4 : MREP;  // (a single batch of 8 measured the same and needs 16 more VGPRs)
  // ---- fast store tail (every lane of the tile stores full 8-channel runs: all interior tiles of the layers that matter).
  // The general tail below is per-lane code: validity masks, exec-mask branches and 64-bit per-lane address products
  // (quarter-rate integer multiplies) -- measured 9.5 k cycles per 128 accumulators with or WITHOUT the stores themselves,
  // i.e. bound by its own VALU/SALU stream.  Here a lane's address is  uniform(fragment, channel pair) + lane part, the lane
  // part (a 32-bit element offset) is computed once, the uniform part lives in SGPRs, and nothing diverges.
  constexpr int TWm = TW < 32 ? TW : 32;
  static_assert(TW % 32 == 0 || (32 % TW == 0 && TH % (32 / TW) == 0), "fragment rows must tile the workgroup tile");
  const bool tile_full = (t0 + TT <= p.To) && (y0 + TH <= p.Ho) && (x0 + TW <= p.Wo) && ((nb + 1) * 32 <= p.Cout) &&
                         (p.out_mode != 2 || (C2 & 31) == 0);  // (geometry only: the element type is chosen below)
  // The fast tail exists for the storage dtype T and -- split-precision instances and the 1x1x1 family, whose attention score
  // product stores fp32 -- for float (32-byte runs per lane: two 16-byte accesses); `zero` selects the element type.
  auto fast_tail = [&](auto zero) {
    using TO = decltype(zero);
    constexpr bool F32 = std::is_same<TO, float>::value;
    constexpr int RBF = F32 ? (RB > 2 ? 2 : RB) : RB;  // fragments per batch (a float residual run is 8 registers)
    const int l31 = lane_e & 31;
    const int Wst = (UPS == 2 ? 2 : 1) * p.Wo, Hst = (UPS == 2 ? 2 : 1) * p.Ho;  // stored frame size
    const unsigned voff = (unsigned)((UPS == 2 ? 2 : 1) * ((l31 / TWm) * Wst + (l31 % TWm)) * p.out_ps + (lane_e >> 5) * 8);
    const int n_sh = (p.out_mode == 2 && nb * 32 >= C2) ? 1 : 0;  // time shuffle: my 32 channels land one frame later
    const int Tq = p.out_mode == 2 ? 2 * p.To - 1 : p.To;
    const int chan0 = nb * 32 - n_sh * C2;
    TO* __restrict__ outp = reinterpret_cast<TO*>(p.out);
    const TO* __restrict__ resp = reinterpret_cast<const TO*>(p.res);
    // element offset of fragment r (its lane 0, my channel block), wave-uniform; drop = the discarded frame -1
    auto row_of = [&](int r, bool& drop) -> long long {
      const int f = wave_m * MREP + r;
      const int utx = (f * 32) % TW, uty = ((f * 32) / TW) % TH, utt = (f * 32) / (TW * TH);
      const int to = t0 + utt;
      const int tq = (p.out_mode == 2 ? 2 * to - 1 : to) + n_sh;
      drop = tq < 0;
      const long long fr = (long long)b * Tq + (drop ? 0 : tq);
      const long long S = UPS == 2 ? ((fr * Hst + 2 * (y0 + uty) + py) * Wst + 2 * (x0 + utx) + px)
                                   : ((fr * Hst + (y0 + uty)) * Wst + (x0 + utx));
      return S * (long long)p.out_ps + chan0;
    };
#pragma unroll
    for (int r0 = 0; r0 < MREP; r0 += RBF) {
      Raw8<TO> rres[RBF][2];
      long long rowe[RBF];
      bool drop[RBF];
#pragma unroll
      for (int ri = 0; ri < RBF; ++ri) rowe[ri] = row_of(r0 + ri, drop[ri]);
      if (p.res && !res_pre) {
#pragma unroll
        for (int ri = 0; ri < RBF; ++ri) {
          if (KG == 2 && (((r0 + ri) < MREP / 2) != (kgrp == 0))) continue;
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) rres[ri][pr] = ldraw8<TO>(resp + rowe[ri] + pr * 16 + voff);
        }
      }
#pragma unroll
      for (int ri = 0; ri < RBF; ++ri) {
        const int r = r0 + ri;
        if (KG == 2 && ((r < MREP / 2) != (kgrp == 0))) continue;  // each K-group stores the half it reduced
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {  // quads 2pr, 2pr+1 = my 8 consecutive channels
            float lo = acc[NI * MREP + r][(2 * pr) * 4 + j], hi = acc[NI * MREP + r][(2 * pr + 1) * 4 + j];
            asm volatile("" : "+v"(lo), "+v"(hi));  // (pins the reads to this point: see the general tail)
            v[j] = lo;
            v[4 + j] = hi;
          }
          if (drop[ri]) continue;  // wave-uniform
          if (!bias_pre) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * p.alpha + bia[pr][j];
          }
          if (p.res && !res_pre) {
            float rf[8];
            unraw8<TO>(rres[ri][pr], rf);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rf[j];
          }
          float rv[8];  // the values as stored (rounded to the storage dtype; float: as they are): what the statistics see
          if constexpr (F32) {
#ifdef CVVAE_CONV_PROBE
            if (!p.probe_nostore)
#endif
            st8<float>(outp + rowe[ri] + pr * 16 + voff, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) rv[j] = v[j];
          } else {
            const uint4 pk = pack8<T>(v);
#ifdef CVVAE_CONV_PROBE
            if (!p.probe_nostore)
#endif
            *reinterpret_cast<uint4*>(outp + rowe[ri] + pr * 16 + voff) = pk;
            if (p.gnp) {
              uint4 pq = pk;  // opaque copy: the ROUNDED values are unpacked from the packed registers (one shift / and
              asm volatile("" : "+v"(pq.x), "+v"(pq.y), "+v"(pq.z), "+v"(pq.w));  // per value; hipcc otherwise re-converts each float)
              unpack8<T>(pq, rv);
            }
          }
          if (p.gnp) {
            if (!gk_set) {  // first stored fragment of the tile: every lane shifts by ITS OWN first value of the slot (made common
              // to the half-wave just before the reduction -- no cross-lane traffic at the start of the tail)
              gk[pr][0] = p.stats_noshift == 1 ? 0.f : rv[0];
              gk[pr][1] = p.stats_noshift == 1 ? 0.f : rv[4];
              gk_set = pr == 1;
            }
            gc[pr] += 1.f;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float dv = rv[q * 4 + j] - gk[pr][q];
                gs[pr][q] += dv;
                gq[pr][q] += dv * dv;
              }
          }
        }
      }
    }
  };
  constexpr bool F32TAIL = XP != 0 || (KH * KW == 1);  // instances that can store float from the fast tail
  const bool f32out = XP != 0 || p.out_f32;
  if (tile_full && !f32out) {
    fast_tail((T)0.f);
  } else if (F32TAIL && tile_full && f32out) {
    if constexpr (F32TAIL) fast_tail(0.0f);
  } else {
  #pragma unroll
    for (int r0 = 0; r0 < MREP; r0 += RB) {
      Raw8<TIO> rres[RB][2];
      int pixr[RB];
      bool ff[RB];
  #pragma unroll
      for (int ri = 0; ri < RB; ++ri) pixr[ri] = locate(r0 + ri, ff[ri]);
      if (p.res && !res_pre) {
  #pragma unroll
        for (int ri = 0; ri < RB; ++ri) {
          const int r = r0 + ri;
          if (KG == 2 && ((r < MREP / 2) != (kgrp == 0))) continue;  // each K-group stores the half it reduced
  #pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const int pix = pix_of(pixr[ri], ff[ri], pr);
            // unconditional 16-byte load (pixel 0 for lanes with nothing to store) keeps the loads branch-free
            rres[ri][pr] = ldraw8<TIO>(reinterpret_cast<const TIO*>(p.res) + (long long)(pix < 0 ? 0 : pix) * (long long)p.out_ps +
                                       ccv[pr]);
          }
        }
      }
  #pragma unroll
      for (int ri = 0; ri < RB; ++ri) {
        const int r = r0 + ri;
        if (KG == 2 && ((r < MREP / 2) != (kgrp == 0))) continue;
  #pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          float v[8];
  #pragma unroll
          for (int j = 0; j < 4; ++j) {
            // (the empty asm pins the accumulator reads to this point of the tail: with plain reads hipcc keeps 128 more
            // values alive across the tails and spills 500+ VGPRs on most instances; an input-only constraint is not enough)
            float lo = acc[NI * MREP + r][(2 * pr) * 4 + j], hi = acc[NI * MREP + r][(2 * pr + 1) * 4 + j];
            asm volatile("" : "+v"(lo), "+v"(hi));
            v[j] = lo;
            v[4 + j] = hi;
          }
          const int pix = pix_of(pixr[ri], ff[ri], pr);
          if (pix < 0) continue;
          const long long off = (long long)pix * (long long)p.out_ps + ccv[pr];
          if (!bias_pre) {  // alpha != 1 (the attention score product): scale and add the bias here
  #pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * p.alpha + bia[pr][j];
          }
          if (p.res && !res_pre) {
            float rf[8];
            unraw8<TIO>(rres[ri][pr], rf);
  #pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rf[j];
          }
          const int c8 = c8v[pr];
          const bool full = (c8 + 7 < p.Cout);
          if (XP != 0 || p.out_f32) {
            float* o = reinterpret_cast<float*>(p.out) + off;
            if (full) {
              *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
              *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
              if (XP != 0 && p.gnp) {  // fused GroupNorm statistics of the fp32 values stored (shifted sums, as below)
                if (!((gkm >> pr) & 1)) {
                  gk[pr][0] = v[0];  // per-lane shift (see the fast tail)
                  gk[pr][1] = v[4];
                  gkm |= 1u << pr;
                }
                gc[pr] += 1.f;
  #pragma unroll
                for (int q = 0; q < 2; ++q)
  #pragma unroll
                  for (int j = 0; j < 4; ++j) {
                    const float dv = v[q * 4 + j] - gk[pr][q];
                    gs[pr][q] += dv;
                    gq[pr][q] += dv * dv;
                  }
              }
            } else {
  #pragma unroll
              for (int j = 0; j < 8; ++j)
                if (c8 + j < p.Cout) o[j] = v[j];
            }
          } else {
            T* o = reinterpret_cast<T*>(p.out) + off;
            if (full) {
              const uint4 pk = pack8<T>(v);
  #ifdef CVVAE_CONV_PROBE
              if (!p.probe_nostore)  // probe: PROBE_NOSTORE runs the store tail without its stores
  #endif
              *reinterpret_cast<uint4*>(o) = pk;
              if (p.gnp) {
                float rv[8];
                uint4 pq = pk;  // opaque copy: the ROUNDED values are unpacked from the packed registers (one shift / and
                asm volatile("" : "+v"(pq.x), "+v"(pq.y), "+v"(pq.z), "+v"(pq.w));  // per value; hipcc otherwise re-converts each float)
                unpack8<T>(pq, rv);
                if (!((gkm >> pr) & 1)) {  // shifts: the values of the first storing lane of my half-wave (per-lane code here)
                  gk[pr][0] = rv[0];  // per-lane shift (see the fast tail)
                  gk[pr][1] = rv[4];
                  gkm |= 1u << pr;
                }
                gc[pr] += 1.f;
  #pragma unroll
                for (int q = 0; q < 2; ++q)
  #pragma unroll
                  for (int j = 0; j < 4; ++j) {
                    const float dv = rv[q * 4 + j] - gk[pr][q];
                    gs[pr][q] += dv;
                    gq[pr][q] += dv * dv;
                  }
              }
            } else {
  #pragma unroll
              for (int j = 0; j < 8; ++j)
                if (c8 + j < p.Cout) o[j] = (T)v[j];
            }
          }
        }
      }
    }
  }
  if (p.gnp) {
    // wave reduction over the 32 pixels of each half-wave (the two halves hold different channels), then lanes 31 and 63
    // write one (n, mean, M2) record per 4-channel slot: record index inside its group = the slot's position in the group
    // every lane accumulated around its own shift K_l; re-express its sums around the COMMON shift K0 of its half-wave (the
    // K_l of the half's first lane that stored something):  S0 = S + n (K_l - K0),  Q0 = Q + 2 (K_l - K0) S + n (K_l - K0)^2
    // (exact algebra; K_l - K0 is of the order of sigma, so nothing cancels)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const unsigned long long mk = tile_full ? ~0ull : __builtin_amdgcn_ballot_w64(((gkm >> pr) & 1) != 0);
      const unsigned mlo = (unsigned)mk, mhi = (unsigned)(mk >> 32);
      const int slo = mlo ? __builtin_ctz(mlo) : 0, shi = 32 + (mhi ? __builtin_ctz(mhi) : 0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float kl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gk[pr][q]), slo));
        const float kh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gk[pr][q]), shi));
        const float k0 = (lane_e & 32) ? kh : kl;
        const float dk = gk[pr][q] - k0, nl = gc[pr] * 4.f;
        gq[pr][q] += dk * (2.f * gs[pr][q] + nl * dk);
        gs[pr][q] += nl * dk;
        gk[pr][q] = k0;
      }
    }
    // (DPP row shifts + row broadcast inside the VALU -- no LDS-pipe shuffles: lanes 31 / 63 end with the sums of their half)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      gc[pr] = half_wave_sum(gc[pr]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        gs[pr][q] = half_wave_sum(gs[pr][q]);
        gq[pr][q] = half_wave_sum(gq[pr][q]);
      }
    }
    float gdev[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (p.stats_noshift == 3) {  // debug aid: do all lanes of a half hold the same shift at the END of the tail?
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float kl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gk[pr][q]), 31));
          const float kh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gk[pr][q]), 63));
          const float dv = gk[pr][q] - ((lane_e & 32) ? kh : kl);
          gdev[pr][q] = half_wave_sum(dv < 0.f ? -dv : dv);
        }
    }
    if ((lane_e & 31) == 31) {
      const int E = 1 << (p.gn_sh - 2);  // 4-channel slots per group
      const int part = ((UPS == 2 ? tile_in_b * 4 + phase : tile_in_b) * WM + wave_m) * KG + kgrp;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          int c = nb * 32 + pr * 16 + (lane_e >> 5) * 8 + q * 4;
          if (c >= p.Cout) continue;
          int slab = part;
          if (p.out_mode == 2) {  // both channel halves land in the same stored channels (different frames): own slabs
            const int n = c >= C2 ? 1 : 0;
            c -= n * C2;
            slab = part * 2 + n;
          }
          const int g = c >> p.gn_sh, sub = (c >> 2) & (E - 1);
          const float n = gc[pr] * 4.f;
          const float dmean = n > 0.f ? gs[pr][q] / n : 0.f;  // mean - K
          const float mean = n > 0.f ? gk[pr][q] + dmean : 0.f;
          float m2 = gq[pr][q] - gs[pr][q] * dmean;
          m2 = m2 > 0.f ? m2 : 0.f;
          // [row][group][record]: the records of one (row, group) are contiguous, so the merge pass reads whole lines
          float* o = p.gnp + (((size_t)b * p.gn_G + g) * p.gn_slabs + (size_t)(slab * E + sub)) * 3;
          o[0] = n;
          o[1] = mean;
          o[2] = p.stats_noshift == 2 ? gk[pr][q] : (p.stats_noshift == 3 ? gdev[pr][q] : m2);  // (debug aid: expose the shift)
        }
    }
  }
  };
  epilogue(std::integral_constant<int, 0>{});
  if constexpr (NB == 2) epilogue(std::integral_constant<int, 1>{});
  CVVAE_PROBE_MARK();
}

// host-side launcher, one per instantiation (defined in conv_inst_*.hip)
template <typename T, int KT, int KH, int KW, int ST, int SH, int SW, int TT, int TH, int TW, int WM, int WN, int KG, int KSUB,
          int PRO, int UPS, int XP = 0, int NB = 1, int LD = 0>
int launch_conv(const ConvArgs& a, int grid, hipStream_t s) {
  // (debug aid: CVVAE_NW4_SOLO=1 pads a 4-wave launch's LDS so that only ONE workgroup fits a CU)
  static const bool solo = getenv("CVVAE_NW4_SOLO") && atoi(getenv("CVVAE_NW4_SOLO"));
  hipLaunchKernelGGL((conv_fwd_kernel<T, KT, KH, KW, ST, SH, SW, TT, TH, TW, WM, WN, KG, KSUB, PRO, UPS, XP, NB, LD>), dim3(grid),
                     dim3(WM * WN * KG * 64), (WM * WN * KG == 4 && solo) ? 48 * 1024 : 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace cvvae
