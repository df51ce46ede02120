// conv_probe.hip -- tuning probe: launches ONE conv_fwd_kernel instantiation on synthetic data with s_memtime stamps in
// the pipeline (kernel built with -DCVVAE_CONV_PROBE) and prints the per-wave timeline of one workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCVVAE_CONV_PROBE -DCFG=<n> -Icvvae_amd/csrc tools/probes/conv_probe.hip -o /tmp/conv_probe
// Marks per wave: [0] kernel start, [1] first stage done, then per chunk: loop top, after phase 1 (group 0: stage next,
// group 1: nothing), after MFMAs, after phase 2 (group 1: stage next), [last-1] after the final barrier, [last] end.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "conv_kernel.h"
using namespace cvvae;

#ifndef CFG
#define CFG 0
#endif
//                     KT KH KW ST SH SW TT TH TW WM WN KG KSUB PRO UPS   Cin  Cout T   H    W
#if CFG == 0   // c2d128, 2 pixel slabs x 4 N
#define INST 1,3,3, 1,1,1, 1,8,32, 2,4,1, 2, 1,0
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 0;
#elif CFG == 1 // c2d128 K-group
#define INST 1,3,3, 1,1,1, 1,8,32, 1,4,2, 2, 1,0
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 0;
#elif CFG == 2 // enc256
#define INST 3,3,3, 1,1,1, 1,8,32, 1,8,1, 1, 1,0
static const int CIN = 256, COUT = 256, TT_ = 9, HH = 256, WW = 256, PT = 2;
#elif CFG == 3 // enc128 2-frame tile
#define INST 3,3,3, 1,1,1, 2,8,32, 2,4,1, 1, 1,0
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 2;
#elif CFG == 10 // enc128 2-frame tile WITHOUT prologue: what the staging costs with no GroupNorm + SiLU arithmetic
#define INST 3,3,3, 1,1,1, 2,8,32, 2,4,1, 1, 0,0
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 2;
#elif CFG == 11 // enc128 fast fp32, bf8 corrections (XP = 2)
#define INST 3,3,3, 1,1,1, 2,4,32, 2,4,1, 1, 1,0
#define XPV 2
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 2;
#elif CFG == 12 // enc128 fast fp32, fp6 corrections (XP = 3)
#define INST 3,3,3, 1,1,1, 2,4,32, 2,4,1, 1, 1,0
#define XPV 3
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 2;
#elif CFG == 14 // enc128 fast fp32, bf8 corrections, WITHOUT prologue
#define INST 3,3,3, 1,1,1, 2,4,32, 2,4,1, 1, 0,0
#define XPV 2
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 2;
#elif CFG == 13 // enc128 exact fp32 (XP = 1)
#define INST 3,3,3, 1,1,1, 2,4,32, 2,4,1, 1, 1,0
#define XPV 1
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 2;
#elif CFG == 15 // enc128 fast fp32, fp6 corrections, PLANAR layout + eight fragments per wave (round 6: Geo::PL)
#define INST 3,3,3, 1,1,1, 2,8,32, 2,4,1, 1, 1,0
#define XPV 3
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 2;
#elif CFG == 16 // ... WITHOUT prologue (no GroupNorm + SiLU arithmetic; the hi / lo split and the bf6 conversion remain)
#define INST 3,3,3, 1,1,1, 2,8,32, 2,4,1, 1, 0,0
#define XPV 3
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 2;
#elif CFG == 17 // enc256 fast fp32, fp6 corrections, planar, all waves in N
#define INST 3,3,3, 1,1,1, 1,8,32, 1,8,1, 1, 1,0
#define XPV 3
static const int CIN = 256, COUT = 256, TT_ = 9, HH = 256, WW = 256, PT = 2;
#elif CFG == 4 // c2d512
#define INST 1,3,3, 1,1,1, 1,8,32, 1,8,1, 2, 1,0
static const int CIN = 512, COUT = 512, TT_ = 9, HH = 128, WW = 128, PT = 0;
#elif CFG == 7 // c2d256: ResnetBlock conv2 at 9x256^2
#define INST 1,3,3, 1,1,1, 1,8,32, 1,8,1, 2, 1,0
static const int CIN = 256, COUT = 256, TT_ = 9, HH = 256, WW = 256, PT = 0;
#elif CFG == 8 // c2d128, the 16-row tile the library picks at 17x512^2
#define INST 1,3,3, 1,1,1, 1,16,32, 2,4,1, 2, 1,0
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 0;
#elif CFG == 9 // c2d128, four-wave instance (two workgroups per CU)
#define INST 1,3,3, 1,1,1, 1,8,32, 1,4,1, 2, 1,0
static const int CIN = 128, COUT = 128, TT_ = 17, HH = 512, WW = 512, PT = 0;
#elif CFG == 18 // the small-frame 512-channel per-frame conv of cfg 1 (1x32x32: 16 workgroups), 2 pixel slabs x 4 N
#define INST 1,3,3, 1,1,1, 1,8,32, 2,4,1, 2, 1,0
static const int CIN = 512, COUT = 512, TT_ = 1, HH = 32, WW = 32, PT = 0;
#elif CFG == 19 // ... K-group instance
#define INST 1,3,3, 1,1,1, 1,8,32, 1,4,2, 2, 1,0
static const int CIN = 512, COUT = 512, TT_ = 1, HH = 32, WW = 32, PT = 0;
#elif CFG == 20 // the small-frame 512-channel 3x3x3 conv of cfg 2 (5x32x32: 80 workgroups of 128 pixels x 256 channels)
#define INST 3,3,3, 1,1,1, 1,4,32, 1,8,1, 1, 1,0
static const int CIN = 512, COUT = 512, TT_ = 5, HH = 32, WW = 32, PT = 1;
#elif CFG == 21 // ... with 32-channel chunks (54 steps per chunk: -DCVVAE_PF_OVERRIDE=18 doubles the weight ring)
#define INST 3,3,3, 1,1,1, 1,4,32, 1,8,1, 2, 1,0
static const int CIN = 512, COUT = 512, TT_ = 5, HH = 32, WW = 32, PT = 1;
#elif CFG == 5 // enc256 without prologue (pro0) for comparison
#define INST 3,3,3, 1,1,1, 1,8,32, 1,8,1, 1, 0,0
static const int CIN = 256, COUT = 256, TT_ = 9, HH = 256, WW = 256, PT = 2;
#elif CFG == 6 // enc256 with GN only (PRO=2: fma, no SiLU)
#define INST 3,3,3, 1,1,1, 1,8,32, 1,8,1, 1, 2,0
static const int CIN = 256, COUT = 256, TT_ = 9, HH = 256, WW = 256, PT = 2;
#endif

#ifndef XPV
#define XPV 0
#endif
static const int ES = XPV ? 4 : 2, WREC = XPV ? 3 : 1;  // bytes per activation element; weight records per (k16, tap)

template <int KT, int KH, int KW, int ST, int SH, int SW, int TT, int TH, int TW, int WM, int WN, int KG, int KSUB, int PRO, bool UPS>
int run() {
  const int taps = KT * KH * KW;
  const size_t npix = (size_t)TT_ * HH * WW;
  void *in, *out, *w; float *bias, *gsc, *gsh; unsigned long long* dbg;
  hipMalloc(&in, npix * CIN * ES); hipMalloc(&out, npix * COUT * ES);
  const size_t wbytes = (size_t)((COUT + 31) / 32) * 32 * CIN * taps * 2 * WREC + WEIGHT_TAIL_BYTES;
  hipMalloc(&w, wbytes + 65536); hipMalloc(&bias, 4 * ((COUT + 31) / 32) * 32); hipMalloc(&gsc, 4 * CIN); hipMalloc(&gsh, 4 * CIN);
  hipMalloc(&dbg, 8 * 128 * 8);
  std::vector<unsigned short> h(npix * CIN);
  srand(1);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff)) | ((rand() & 1) << 15);  // bf16 in +-[0.0078, 0.0156): random bits
  if (XPV) {
    std::vector<float> hf(npix * CIN);
    for (auto& v : hf) v = (float)((rand() & 0xffff) - 32768) / 32768.0f;
    hipMemcpy(in, hf.data(), npix * CIN * 4, hipMemcpyHostToDevice);
  } else
  hipMemcpy(in, h.data(), npix * CIN * 2, hipMemcpyHostToDevice);
  std::vector<unsigned short> hw(wbytes / 2);
  for (auto& v : hw) v = (unsigned short)(0x3a00 + (rand() & 0x3ff)) | ((rand() & 1) << 15);
  hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice);
  std::vector<float> ones(CIN, 1.0f);
  hipMemcpy(gsc, ones.data(), 4 * CIN, hipMemcpyHostToDevice);
  hipMemset(gsh, 0, 4 * CIN); hipMemset(bias, 0, 4 * ((COUT + 31) / 32) * 32);
  hipMemset(dbg, 0, 8 * 128 * 8);
  ConvArgs a; memset(&a, 0, sizeof(a));
  a.in = in; a.w = w; a.bias = bias; a.gsc = gsc; a.gsh = gsh; a.out = out;
  a.B = 1; a.Ti = TT_; a.Hi = HH; a.Wi = WW; a.Tl = TT_; a.Hl = HH; a.Wl = WW; a.Cin = CIN; a.in_ps = CIN;
  a.To = TT_; a.Ho = HH; a.Wo = WW; a.Cout = COUT; a.out_ps = COUT;
  a.pt = PT; a.ph = 1; a.pw = 1; a.mode_t = 1; a.mode_hw = KT == 3 ? 1 : 0;
  a.tiles_t = (TT_ + TT - 1) / TT; a.tiles_h = HH / TH; a.tiles_w = WW / TW; a.ntiles_n = (COUT + 32 * WN - 1) / (32 * WN);
  a.nchunks = CIN / (16 * KSUB); a.nblk32 = (COUT + 31) / 32; a.order = 1; a.gn_rpb = 1; a.alpha = 1.f;
  a.w_taps = taps * WREC;  // plain packed layout (no time-fold slots)
  a.q6_scale = 0.5f; a.q6_eb = 128;
  const int grid = a.tiles_t * a.tiles_h * a.tiles_w * a.ntiles_n;
  if (getenv("PROBE_RES") || getenv("PROBE_STATS")) {  // PROBE_RES: residual add + fused GroupNorm statistics in the epilogue
    if (getenv("PROBE_RES")) {                          // (what a ResnetBlock conv2 does); PROBE_STATS: statistics only (conv1)
      void* res; hipMalloc(&res, npix * COUT * 2); hipMemcpy(res, in, (npix * COUT * 2 < npix * CIN * 2 ? npix * COUT * 2 : npix * CIN * 2), hipMemcpyDeviceToDevice);
      a.res = res;
      a.res_pre = getenv("PROBE_RES_TAIL") ? 0 : 1;  // default: residual pre-accumulated in the K loop (as the library does)
    }
    const int cpg = COUT / 32; int sh = 0; while ((1 << sh) < cpg) ++sh;
    a.gn_G = 32; a.gn_sh = sh; a.gn_slabs = a.tiles_t * a.tiles_h * a.tiles_w * WM * KG * (1 << (sh - 2));
    float* gnp; hipMalloc(&gnp, (size_t)a.gn_slabs * 32 * 3 * 4); a.gnp = gnp;
  }
  if (getenv("PROBE_NOSTORE")) a.probe_nostore = 1;  // the store tail without its stores: what is left is VALU + addressing
  a.dbg = dbg; a.dbg_block = getenv("PROBE_BLOCK") ? atoi(getenv("PROBE_BLOCK")) : grid / 2 + 3;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0, 0);
    launch_conv<std::conditional_t<XPV != 0, _Float16, __bf16>, KT, KH, KW, ST, SH, SW, TT, TH, TW, WM, WN, KG, KSUB, PRO, UPS, XPV>(a, grid, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * npix * COUT * CIN * taps;
    printf("cfg %d iter %d: %.3f ms  %.1f TFLOP/s  grid %d\n", CFG, it, ms, fl / ms / 1e9, grid);
  }
  std::vector<unsigned long long> t(8 * 128);
  hipMemcpy(t.data(), dbg, 8 * 128 * 8, hipMemcpyDeviceToHost);
  // mark layout per wave: [0] start, [1,2] stage0 load-issue / loads-returned, [3] stage0 end, then per chunk 4 marks
  // (loop top, after phase 1, after MFMAs, after phase 2) + 2 stage sub-marks inside whichever phase stages, then
  // [n-2] after the final barrier, [n-1] end of kernel.
  unsigned long long t0 = ~0ull;
  for (int wv = 0; wv < 8; ++wv) if (t[wv * 128] && t[wv * 128] < t0) t0 = t[wv * 128];
  printf("block %d, %d chunks (shader cycles):\n", a.dbg_block, a.nchunks);
  for (int wv = 0; wv < 8; wv += 4) {
    const unsigned long long* m = &t[wv * 128];
    printf(" wave %d (grp %d): start %llu, stage0 [issue %llu wait %llu valu+write %llu]\n", wv, wv >> 2, m[0] - t0, m[1] - m[0],
           m[2] - m[1], m[3] - m[2]);
    int i = 4;
    for (int c = 0; c < a.nchunks && c < 6; ++c) {
      const bool more = c + 1 < a.nchunks;
      if (wv < 4) {  // group 0: stage in phase 1
        if (more) {
          printf("   c%d: stage[issue %llu wait %llu valu+write %llu] mfma %llu ph2 %llu bar %llu\n", c, m[i + 1] - m[i], m[i + 2] - m[i + 1],
                 m[i + 3] - m[i + 2], m[i + 4] - m[i + 3], m[i + 5] - m[i + 4], m[i + 6] - m[i + 5]);
          i += 6;
        } else {
          printf("   c%d: ph1 %llu mfma %llu ph2 %llu bar %llu\n", c, m[i + 1] - m[i], m[i + 2] - m[i + 1], m[i + 3] - m[i + 2], m[i + 4] - m[i + 3]);
          i += 4;
        }
      } else {
        if (more) {
          printf("   c%d: ph1 %llu mfma %llu stage[issue %llu wait %llu valu+write %llu] bar %llu\n", c, m[i + 1] - m[i], m[i + 2] - m[i + 1],
                 m[i + 3] - m[i + 2], m[i + 4] - m[i + 3], m[i + 5] - m[i + 4], m[i + 6] - m[i + 5]);
          i += 6;
        } else {
          printf("   c%d: ph1 %llu mfma %llu ph2 %llu bar %llu\n", c, m[i + 1] - m[i], m[i + 2] - m[i + 1], m[i + 3] - m[i + 2], m[i + 4] - m[i + 3]);
          i += 4;
        }
      }
    }
    int n = 4 + 6 * (a.nchunks - 1) + 4 + 2;
    printf("   epilogue %llu, total %llu\n", m[n - 1] - m[n - 2], m[n - 1] - m[0]);
  }
  return 0;
}
int main() { return run<INST>(); }
