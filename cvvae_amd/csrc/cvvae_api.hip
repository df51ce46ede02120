// cvvae_api.hip -- extern "C" entry for the convolution: argument checking, tile/instance selection, launch.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/cvvae.h"
#include "conv_kernel.h"
#include "conv_table.h"

namespace cvvae {

// the instantiations live in conv_inst_*.hip
#define CVVAE_EXTERN(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  extern template int launch_conv<__bf16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS>(const ConvArgs&, int, hipStream_t); \
  extern template int launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS>(const ConvArgs&, int, hipStream_t);
CVVAE_CONV_ALL(CVVAE_EXTERN)
#define CVVAE_EXTERN_XP(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  extern template int launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,1>(const ConvArgs&, int, hipStream_t);
CVVAE_CONV_XP(CVVAE_EXTERN_XP)
#define CVVAE_EXTERN_XQ(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  extern template int launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,2>(const ConvArgs&, int, hipStream_t);
CVVAE_CONV_XQ(CVVAE_EXTERN_XQ)
#define CVVAE_EXTERN_XQ6(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  extern template int launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,3>(const ConvArgs&, int, hipStream_t);
CVVAE_CONV_XQ6(CVVAE_EXTERN_XQ6)
#define CVVAE_EXTERN_XQ6_NB2(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  extern template int launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,3,2>(const ConvArgs&, int, hipStream_t);
CVVAE_CONV_XQ6_NB2(CVVAE_EXTERN_XQ6_NB2)
#define CVVAE_EXTERN_NB2(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  extern template int launch_conv<__bf16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,0,2>(const ConvArgs&, int, hipStream_t); \
  extern template int launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,0,2>(const ConvArgs&, int, hipStream_t);
CVVAE_CONV_NB2(CVVAE_EXTERN_NB2)

#define CVVAE_EXTERN_LD(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  extern template int launch_conv<__bf16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,0,1,1>(const ConvArgs&, int, hipStream_t); \
  extern template int launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,0,1,1>(const ConvArgs&, int, hipStream_t);
CVVAE_CONV_LD(CVVAE_EXTERN_LD)

// Per-frame GroupNorm tables are merged from the statistics records of a kT == 1 conv (cvvae_gn_finalize_frames, ops.GNPartials.frames):
// that assumes ONE-FRAME tiles written in frame-major record order, i.e. TT == 1 for every kT == 1 instance of every list
#define CVVAE_CHECK_FRAME_TILES(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  static_assert(KT != 1 || TT == 1, "kT == 1 instances must have one-frame tiles: per-frame statistics are merged from their records");
CVVAE_CONV_ALL(CVVAE_CHECK_FRAME_TILES)
CVVAE_CONV_NB2(CVVAE_CHECK_FRAME_TILES)
CVVAE_CONV_LD(CVVAE_CHECK_FRAME_TILES)
CVVAE_CONV_XP(CVVAE_CHECK_FRAME_TILES)
CVVAE_CONV_XQ(CVVAE_CHECK_FRAME_TILES)
CVVAE_CONV_XQ6(CVVAE_CHECK_FRAME_TILES)
CVVAE_CONV_XQ6_NB2(CVVAE_CHECK_FRAME_TILES)

typedef int (*launch_fn)(const ConvArgs&, int, hipStream_t);

struct Instance {
  int kt, kh, kw, st, sh, sw, tt, th, tw, wm, wn, kg, ksub, pro, ups;
  int nbw;  // 32-channel N-blocks per wave (1; 2 = the register-blocked instances: a wave's tile is 64 channels wide)
  int ld;   // 1: DMA-staged (conv_kernel.h LD): 16-bit, no prologue
  launch_fn fn[5];  // [CVVAE_F16], [CVVAE_BF16], [CVVAE_F32] (split-precision instances: only this one), [CVVAE_F32Q], [CVVAE_F32Q6] (fast fp32)
  char name[96];
};

#define CVVAE_ROW(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  {KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS, 1, 0, \
   {&launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS>, \
    &launch_conv<__bf16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS>, nullptr, nullptr, nullptr}, ""},
#define CVVAE_ROW_NB2(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  {KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS, 2, 0, \
   {&launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,0,2>, \
    &launch_conv<__bf16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,0,2>, nullptr, nullptr, nullptr}, ""},
#define CVVAE_ROW_XP(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  {KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS, 1, 0, \
   {nullptr, nullptr, &launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,1>, nullptr, nullptr}, ""},
#define CVVAE_ROW_XQ(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  {KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS, 1, 0, \
   {nullptr, nullptr, nullptr, &launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,2>, nullptr}, ""},
#define CVVAE_ROW_XQ6(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  {KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS, 1, 0, \
   {nullptr, nullptr, nullptr, nullptr, &launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,3>}, ""},

#define CVVAE_ROW_XQ6_NB2(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  {KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS, 2, 0, \
   {nullptr, nullptr, nullptr, nullptr, &launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,3,2>}, ""},

#define CVVAE_ROW_LD(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  {KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS, 1, 1, \
   {&launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,0,1,1>, \
    &launch_conv<__bf16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,0,1,1>, nullptr, nullptr, nullptr}, ""},

static Instance g_table[] = {CVVAE_CONV_LD(CVVAE_ROW_LD) CVVAE_CONV_ALL(CVVAE_ROW) CVVAE_CONV_NB2(CVVAE_ROW_NB2) CVVAE_CONV_XP(CVVAE_ROW_XP) CVVAE_CONV_XQ(CVVAE_ROW_XQ) CVVAE_CONV_XQ6(CVVAE_ROW_XQ6) CVVAE_CONV_XQ6_NB2(CVVAE_ROW_XQ6_NB2)};
static const int g_ntable = (int)(sizeof(g_table) / sizeof(g_table[0]));

static inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

// compute units of the current device (256 on MI355X); used by the instance cost model only
static int cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;  // no device visible (CPU-side symbol / argument tests)
  }
  return cus;
}

// pick the instance with the least padded work (tile overhang x inactive N waves); ties -> table order.
// Every term is evaluated PER BATCH ITEM: the tiling decides at which K chunk a residual is pre-accumulated and whether a K split
// is used, i.e. the summation order, hence bits -- a choice that depended on how many clips are coded together would break
// "a batch of B clips == B single-clip calls, bit for bit" (tests/test_gpu_model.py::test_batch_of_clips_matches_single_clips).
static double instance_cost(const cvvae_conv_desc* d, const Instance& e) {
  const int fold = d->upsample2x == 2;
  const long long bm = (long long)e.tt * e.th * e.tw, bn = 32LL * e.wn * e.nbw;
  // per-phase output grid for the folded upsample (4 phases of Ho/2 x Wo/2), the output grid otherwise
  const long long tiles = fold ? cdiv(d->To, e.tt) * cdiv(d->Ho / 2, e.th) * cdiv(d->Wo / 2, e.tw) * d->B * 4
                               : cdiv(d->To, e.tt) * cdiv(d->Ho, e.th) * cdiv(d->Wo, e.tw) * d->B;
  const long long ntn = cdiv(d->Cout, bn);
  // cost ~ MFMA work issued (padded) + staging work (halo per N tile)
  double cost = (double)tiles * (double)bm * (double)ntn * (double)bn;
  // staging share: halo pixels staged per output pixel, once per N tile (grows as BN shrinks)
  const double halo = (double)((e.tt - 1) * e.st + e.kt) * ((e.th - 1) * e.sh + e.kh) * ((e.tw - 1) * e.sw + e.kw) / (double)bm;
  // (fp32 models: staging an element costs about three times as much -- 32-byte loads, the GroupNorm + SiLU in fp32, the hi / lo
  //  split, the code conversion, three LDS writes.  Measured in round 6: with the planar XQ6 tiles in the list, the 0.04 of the 16-bit
  //  kernels sent the 256- / 512-channel per-frame convs to the 128-channel 16-row tile -- every halo element staged once per N
  //  tile, eight fragments per wave -- and they ran 20 % slower than on the all-waves-in-N tile with four)
  const bool fp32inst = e.fn[CVVAE_F32] || e.fn[CVVAE_F32Q] || e.fn[CVVAE_F32Q6];
  cost *= 1.0 + (fp32inst ? 0.12 : 0.04) * halo * 256.0 / (double)bn;
  // weight traffic share grows as a weight record feeds fewer MFMAs (pixels per wave = BM / WM)
  cost *= 1.0 + 0.05 * 256.0 / ((double)bm / (double)e.wm);
  if (e.kg == 2) cost *= 1.08;  // accumulator reduction through LDS (three barriers and 64 KiB of LDS traffic per tile)
  // DMA-staged twins: same tile, no staging registers or VALU, every wave multiplies all the time (conv_kernel.h LD).
  // CVVAE_CONV_DMA=0 takes them out (A/B aid), =<factor> scales their cost
  if (e.ld) cost *= 0.9;
  // two N-blocks per wave: half the LDS operand reads per MFMA (the weight-traffic term above already charges its doubled weight
  // stream); CVVAE_CONV_NB2=<factor> scales its cost (tuning aid)
  if (e.nbw == 2) {
    static const double nb2 = getenv("CVVAE_CONV_NB2") ? atof(getenv("CVVAE_CONV_NB2")) : 1.0;
    cost *= nb2;
    // fast-fp32 (fp6) instances: the 2 x 4 register block is the better form of the same tile -- half the LDS operand bytes per MFMA,
    // where the fp6 K loop reads 3.5 KiB per fragment and pair of taps: measured +4 % on the 128-channel 3x3x3 layers, +1.5 % on the
    // 256 / 512-channel ones against the eight-fragment planar tiles (profiles/r6_ab_planar_nb2.log); the weight-traffic term above
    // would otherwise rank it behind them
    if (e.fn[CVVAE_F32Q6]) cost *= 0.93;
  }
  // four-wave instances (two workgroups per CU): candidates only when the DESCRIPTOR says so (cvvae_conv_desc.four_wave, ABI 13: a
  // per-launch field, no library state) -- +6 % on the per-frame 128-channel conv with residual + statistics, 1.71 -> 1.62 ms at
  // 17x512^2 (profiles/r2_ab_4wave_*.log, r5_ab_four_wave.log).  CVVAE_CONV_NW4=<factor> overrides the cost factor (tuning aid).
  if (e.wm * e.wn * e.kg == 4) {
    static const double nw4_env = getenv("CVVAE_CONV_NW4") ? atof(getenv("CVVAE_CONV_NW4")) : -1.0;
    const double nw4 = nw4_env >= 0.0 ? nw4_env : (d->four_wave ? 0.95 : 0.0);
    cost *= nw4 > 0.0 ? nw4 : 100.0;
  }
  // strided convs do 4-8x fewer MFMAs per staged byte and their halos (430 KiB per workgroup at 128 channels) do not survive in
  // L2 between K-chunk passes: every pass re-fetches whole 128-byte lines for 32 bytes of them.  32-channel chunks halve
  // the passes (measured: 128 ch s222 0.89 -> 0.81 ms, 512 ch 0.305 -> 0.260 ms, s122 unchanged)
  if (e.st * e.sh * e.sw > 1 && e.ksub == 1) cost *= 1.15;
  // round quantisation: a grid of W workgroups runs in ceil(W / #CUs) rounds of one workgroup per CU
  // (half weight: measured, a partly filled last round costs less than its share -- the busy CUs clock higher)
  {
    const double wgs = (double)tiles / (double)d->B * (double)ntn, cus = (double)cu_count();
    static const int quant = getenv("CVVAE_CONV_QUANT") ? atoi(getenv("CVVAE_CONV_QUANT")) : 1;  // tuning aid
    // (not for the 1x1 family: its batch items are the FRAMES of the attention blocks -- 32 workgroups each, several per launch --
    //  and the per-item term pushed those products to the instance with more, smaller workgroups: 0.69 -> 0.41 ms per cfg 3 step
    //  with the 256-channel tile)
    // weight of the term: 0.5 until round 4 ("a partly filled last round costs half of its share"), measured when the 128-pixel
    // tiles were as fast per MFMA as the 256-pixel ones.  Since the MREP >= 8 instances keep their B fragments single-buffered and
    // batch their staging loads (round 2), a 128-pixel tile costs ~1.25x per MFMA, and on vae3d's small frames (cfg 2: 9x128x128
    // at 256 channels, 9x64x64 at 512) the 256-pixel tile is 15-23 % faster even at 1.1-2.25 rounds
    // (profiles/r4_tune_instances_cfg2.log): 0.1 keeps the term as the tie-breaker against starved grids (5x32x32: 40 workgroups)
    static const double quant_w = getenv("CVVAE_CONV_QUANT_W") ? atof(getenv("CVVAE_CONV_QUANT_W")) : 0.1;
    // (round 5: charging a STARVED grid -- fewer workgroups than CUs -- its empty CUs in full was tried: right for vae3d's 512-channel
    //  3x3x3 convs at 5x32x32 (80 workgroups of the BN = 256 tile 0.181 ms, 320 of the BN = 32 tile 0.160 ms) and wrong for cfg 3's at
    //  5x64x64 (160 workgroups of the 256-pixel tile 0.29 ms, 320 of the 128-pixel tile 0.30 ms: +1.3 ms per step).  Not kept;
    //  profiles/r5_ab_four_wave.log, profiles/r5_tune_instances_cfg2.log)
    if (quant && !(e.kt == 1 && e.kh == 1 && e.kw == 1)) cost *= 1.0 + quant_w * (ceil(wgs / cus) / (wgs / cus) - 1.0);
    // SMALL LAYERS (round 6): where the 256-pixel x 128-channel tiling of the layer is no more than one workgroup per CU, the launch
    // takes ONE workgroup's time whatever their number (profiles/r6_small_layers.log: a 512-channel per-frame conv takes 61 us at
    // 1x32x32 = 16 workgroups, 65 us at 1x64x64 = 64, 66 us at 5x32x32 = 80), and that time is far from proportional to the tile's
    // area (r6_small_layers_v2.log, 512 -> 512 1x3x3 behind GroupNorm + SiLU: 256 x 128 pixels x channels 61-66 us, 128 x 128 44-45,
    // 64 x 256 40-42, 64 x 128 38-40 -- and 57 us when its 320 workgroups need a second round).  Such layers are ranked by a time
    // model of their own:
    //   t(workgroup) = 0.47 + 0.53 x area x shape / (256 x 128)     a K chunk has a latency floor: load -> GroupNorm + SiLU -> LDS ->
    //                                                               barrier -> MFMAs, ~1.9 us of the 256 x 128 tile's 4.0
    //   shape = (BN + s x halo) / BN, normalised to the 256 x 128 tile: per chunk a wave pair stages, then multiplies (conv_kernel.h, X / Y
    //           groups), and the staging of that tile takes 4.4k cycles against 3.2k of MFMAs (r6_probe_small_layers.log) -- a staged
    //           pixel costs as much as s ~ 130 output channels of MFMAs (~60 without prologue, ~30 as a DMA wave-load)
    //   rounds  = 0.92 + 0.08 r for r = workgroups / CUs <= 1, else r + 0.3 (ceil(r) - r)
    // for layers whose weights stay in L2: the 3x3x3 layers at 512 channels stream 14 MB per workgroup column and get SLOWER with
    // more, smaller tiles (round 5, above).  Per batch item, like every other term.  Measured: cfg 1 (image mode, 256^2) 3.77 ->
    // 2.9 ms with the small tiles of conv_table.h G14; cfg 3 has no such layer but the 1024-token 1x1 convs.
    // CVVAE_CONV_SMALL=0 switches the model off (tuning aid, read once).
    static const bool small_on = !(getenv("CVVAE_CONV_SMALL") && atoi(getenv("CVVAE_CONV_SMALL")) == 0);
    const double wbytes = (double)d->Cout * (double)d->Cin * (double)(d->kT * d->kH * d->kW) * 2.0;
    const double opix = fold ? (double)d->To * (d->Ho / 2) * (d->Wo / 2) * 4.0 : (double)d->To * d->Ho * d->Wo;
    const double wgs_ref = ceil(opix / 256.0) * ceil((double)d->Cout / 128.0);
    if (small_on && wgs_ref <= cus && !d->w_batch_stride && wbytes <= 8.0e6) {
      const double sw = e.ld ? 30.0 : (e.pro ? 130.0 : 60.0);
      const double shape = ((double)bn + sw * halo) / (double)bn * 128.0 / (128.0 + sw * 1.33);
      const double t_wg = 0.47 + 0.53 * (double)bm * (double)bn * shape / 32768.0;
      const double r = wgs / cus;
      const double rounds = r <= 1.0 ? 0.92 + 0.08 * r : r + 0.3 * (ceil(r) - r);
      double c = t_wg * rounds * (double)d->B;
      // K-group split: three barriers and the accumulator exchange through LDS (+8 %) against HALF the weight stream per wave --
      // which is the floor of a small tile when K is long: 512 -> 512 1x3x3 at 1x32x32 0.035 -> 0.026 ms, at 1x64x64 0.038 -> 0.029
      // on the 64 x 128 tile, nothing at 256 channels (0.029 vs 0.030) or 128 (profiles/r6_small_layers_kg2.log)
      if (e.kg == 2) {
        const double K = (double)d->Cin * (double)(e.kt * e.kh * e.kw);
        // (measured behind the GroupNorm + SiLU prologue only; the prologue-free lists keep their DMA-staged twins)
        const double x = !e.pro || K <= 2304.0 ? 0.0 : (K >= 4608.0 ? 1.0 : (K - 2304.0) / 2304.0);
        static const bool kg_off = getenv("CVVAE_CONV_SMALL_KG") && atoi(getenv("CVVAE_CONV_SMALL_KG")) == 0;  // (A/B aid, read once)
        c *= kg_off ? 1.08 : 1.08 - 0.30 * x;
      }
      if (e.ld) c *= 0.9;
      if (e.wm * e.wn * e.kg == 4) c *= 100.0;  // (four-wave instances: measured on full grids only)
      if (e.st * e.sh * e.sw > 1 && e.ksub == 1) c *= 1.15;
      return c;
    }
  }
  return cost;
}

static const Instance* select_instance(const cvvae_conv_desc* d) {
  // tuning aid: CVVAE_CONV_FORCE="TTxTHxTW:WMxWNxKG:KSUB" restricts the choice (ignored when nothing matches)
  // (optional ":NB" suffix: 32-channel N-blocks per wave)
  int ft = 0, fh = 0, fw = 0, fm = 0, fn = 0, fg = 0, fk = 0, fnb = 0;
  if (const char* f = getenv("CVVAE_CONV_FORCE")) sscanf(f, "%dx%dx%d:%dx%dx%d:%d:%d", &ft, &fh, &fw, &fm, &fn, &fg, &fk, &fnb);
  // The DMA-staged instances are taken where they exist (CVVAE_CONV_DMA=0 takes them out; read per call: the GPU tests flip it inside
  // one process).  Measured (profiles/r5_ab_dma_staging.log, three interleaved rounds on one box): bit-identical results; cfg 3
  // 68.97 -> 68.50 ms per step with the layers that have no prologue (folded upsample convs, strided downsamplers, 1x1 layers) on
  // them; the 3x3x3 layers behind the GroupNorm + SiLU pass run at the SAME time in either form -- matrix-pipe busy x clock is 0.44-0.50
  // of nominal both ways (profiles/r5_pmc_sq_dma{0,1}.txt): those kernels sit on the board's power frontier, not on the issue rate of
  // a lone multiplying wave
  const char* dma_env = getenv("CVVAE_CONV_DMA");
  const bool dma_off = dma_env && atoi(dma_env) == 0;
  auto eligible = [&](const Instance& e, bool forced) {
    if (!e.fn[d->dtype]) return false;  // fp32 models run the split-precision instances, fp16 / bf16 models the others
    if (forced && ft && (e.tt != ft || e.th != fh || e.tw != fw || e.wm != fm || e.wn != fn || e.kg != fg || e.ksub != fk)) return false;
    if (forced && ft && fnb && e.nbw != fnb) return false;
    if (e.nbw == 2 && (d->Cout % 64)) return false;  // both N-blocks of every wave must be real
    // four-wave instances: measured where they pay -- ONE N tile (Cout <= 128: with more, the BN = 256 eight-wave tiles win: cfg 3's
    // 256/512-channel per-frame convs 7.66 -> 7.7 ms, cfg 1's image-mode convs 2.76 -> 3.37 ms) on frames large enough to keep
    // every CU double-occupied (17x256^2: +9-13 %, 4x512^2: +2-6 %; profiles/r5_tune_instances_cfg2.log)
    if (e.wm * e.wn * e.kg == 4 && (d->Cout > 128 || (long long)d->B * d->To * d->Ho * d->Wo < (1LL << 19))) return false;
    if (e.ld) {
      if (dma_off || d->in_overlap) return false;
      // the wave-loads address a pixel by a 32-bit byte offset from the tensor's base
      if ((long long)d->B * d->Ti * d->Hi * d->Wi * d->in_pix_stride * 2 >= (1LL << 32)) return false;
      if (d->sc_Cin && (long long)d->B * d->Ti * d->Hi * d->Wi * d->sc_in_pix_stride * 2 >= (1LL << 32)) return false;
    }
    // the folded upsample (upsample2x == 2) runs 3x2x2 phase kernels; everything else matches the descriptor's taps
    const int fold = d->upsample2x == 2;
    if (e.kt != d->kT || e.kh != (fold ? 2 : d->kH) || e.kw != (fold ? 2 : d->kW)) return false;
    if (e.st != d->sT || e.sh != d->sH || e.sw != d->sW) return false;
    if (e.pro != d->prologue || e.ups != d->upsample2x) return false;
    if (d->Cin % (16 * e.ksub)) return false;  // the instance's K-chunk must divide the consumed channels
    if (d->sc_Cin && (e.kg != 1 || e.ups != 0 || d->sc_Cin % (16 * e.ksub))) return false;  // fused shortcut: KG = 1 instances
    if (d->w_time_folds && e.kg != 1) return false;  // the time-fold record layout is walked by the KG = 1 kernels only
    return true;
  };
  for (int pass = 0; pass < 2; ++pass) {
    const bool forced = pass == 0;
    const Instance* best = nullptr;
    double best_cost = 0;
    for (int i = 0; i < g_ntable; ++i) {
      const Instance& e = g_table[i];
      if (!eligible(e, forced)) continue;
      const double cost = instance_cost(d, e);
      if (!best || cost < best_cost) {
        best = &e;
        best_cost = cost;
      }
    }
    if (!best) continue;
    return best;
  }
  return nullptr;
}

// An odd frame count under a two-frame tile wastes half of the last tile's MFMAs (T = 17: 5.9 %).  When the same
// configuration exists with a one-frame tile, the launch is split: [0, To-1) with e, the last frame with the sibling.
static const Instance* odd_frame_sibling(const cvvae_conv_desc* d, const Instance* e) {
  if (e->tt != 2 || !(d->To & 1) || d->To < 3 || d->upsample2x == 2 || d->sT != 1) return nullptr;
  for (int i = 0; i < g_ntable; ++i) {
    const Instance& s = g_table[i];
    if (!s.fn[d->dtype]) continue;
    if (s.tt == 1 && s.th == e->th && s.tw == e->tw && s.wm == e->wm && s.wn == e->wn && s.nbw == e->nbw && s.ld == e->ld && s.kg == e->kg && s.ksub == e->ksub &&
        s.pro == e->pro && s.ups == e->ups && s.kt == e->kt && s.kh == e->kh && s.kw == e->kw && s.st == e->st && s.sh == e->sh &&
        s.sw == e->sw)
      return &s;
  }
  return nullptr;
}

static const char* instance_name(Instance* e, int dtype) {
  if (!e->name[0])
    snprintf(e->name, sizeof(e->name), "conv_k%d%d%d_s%d%d%d_t%dx%dx%d_w%dx%dx%d_c%d_pro%d_ups%d%s", e->kt, e->kh, e->kw, e->st,
             e->sh, e->sw, e->tt, e->th, e->tw, e->wm, e->wn, e->kg, 16 * e->ksub, e->pro, e->ups,
             e->fn[2] ? "_xp" : (e->fn[3] ? "_xq" : (e->fn[4] ? (e->nbw == 2 ? "_xq6nb2" : "_xq6") : (e->nbw == 2 ? "_nb2" : (e->ld ? "_dma" : "")))));
  (void)dtype;
  return e->name;
}

static int check_desc(const cvvae_conv_desc* d) {
  if (!d) return CVVAE_EINVAL;
  if (d->dtype < CVVAE_F16 || d->dtype > CVVAE_F32Q6) return CVVAE_EINVAL;
  // fast fp32: multi-tap convolutions without a fused shortcut or per-item weights (those run as CVVAE_F32)
  const bool fastq = d->dtype == CVVAE_F32Q || d->dtype == CVVAE_F32Q6;
  if (fastq && (d->kH * d->kW == 1 || d->sc_Cin || d->w_batch_stride)) return CVVAE_EUNSUPPORTED;
  // fp6 corrections: the activations' scale comes from the caller's bound (finite, > 0)
  // ... or points at one on the device (act_bound_dev), never both
  if (d->dtype == CVVAE_F32Q6) {
    const bool host_bound = d->act_bound > 0.0f && d->act_bound < 3.0e38f;
    if (d->act_bound_dev ? d->act_bound != 0.0f : !host_bound) return CVVAE_EINVAL;
  } else if (d->act_bound != 0.0f || d->act_bound_dev) return CVVAE_EINVAL;
  if (d->B <= 0 || d->Ti <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->To <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0)
    return CVVAE_EINVAL;
  const int ck = cvvae_conv_kchunk(d->kT, d->kH, d->kW);
  if (!ck) return CVVAE_EUNSUPPORTED;
  if (d->Cin <= 0 || d->Cin % ck) return CVVAE_EINVAL;
  if (d->in_overlap) {  // row-packed first layer: 16 virtual channels = 4 stored pixels of 4 channels, kW folded into them
    if (d->in_overlap != 1 || d->kT != 3 || d->kH != 3 || d->kW != 1 || d->Cin != 16 || d->in_pix_stride != 4 || d->pad_w != 0 ||
        d->sW != 1 || d->prologue != CVVAE_PRO_NONE || d->upsample2x || d->sc_Cin || d->Wo > d->Wi - 3 ||
        (d->dtype != CVVAE_F16 && d->dtype != CVVAE_BF16))
      return CVVAE_EINVAL;
  } else if (d->kW == 1 && d->kH == 3) {
    return CVVAE_EINVAL;  // (3,3,1) exists as the row-packed form only
  } else if (d->in_pix_stride < d->Cin || d->in_pix_stride % 8) return CVVAE_EINVAL;
  if (d->out_mode < 0 || d->out_mode > 2) return CVVAE_EINVAL;
  if (d->out_mode != CVVAE_OUT_NCDHW && (d->out_pix_stride < (d->out_mode == 2 ? d->Cout / 2 : d->Cout))) return CVVAE_EINVAL;
  if (d->out_mode != CVVAE_OUT_NCDHW && (d->out_pix_stride % 8)) return CVVAE_EINVAL;  // 16-byte stores
  if (d->out_f32 && d->out_mode != CVVAE_OUT_NDHWC) return CVVAE_EINVAL;
  if (d->out_mode == CVVAE_OUT_TIME_SHUFFLE && (d->Cout % 16)) return CVVAE_EINVAL;
  if (d->prologue < 0 || d->prologue > 2) return CVVAE_EINVAL;
  if (d->upsample2x < 0 || d->upsample2x > 2) return CVVAE_EINVAL;
  if (d->w_batch_stride < 0 || d->w_batch_stride % 16) return CVVAE_EINVAL;
  if (d->w_time_folds != 0 && (d->w_time_folds != 1 || d->kT != 3 || d->w_batch_stride != 0)) return CVVAE_EINVAL;
  if (d->sc_Cin < 0 || (d->sc_Cin && (d->sc_in_pix_stride < d->sc_Cin || d->sc_in_pix_stride % 8))) return CVVAE_EINVAL;
  if (d->sc_Cin && (d->kT != 1 || d->kH != 3 || d->kW != 3 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->upsample2x))
    return CVVAE_EUNSUPPORTED;
  if (d->upsample2x == 2 && ((d->kT != 3 && d->kT != 1) || d->kH != 3 || d->kW != 3 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pad_h != 1 ||
                             d->pad_w != 1 || d->Ho != 2 * d->Hi || d->Wo != 2 * d->Wi || d->out_mode == CVVAE_OUT_NCDHW))
    return CVVAE_EUNSUPPORTED;
  if (d->gn_rows_per_batch < 1) return CVVAE_EINVAL;
  if (d->gn_rows_per_batch > 1 && (d->kT != 1 || d->gn_rows_per_batch != d->Ti)) return CVVAE_EINVAL;
  if (d->four_wave != 0 && d->four_wave != 1) return CVVAE_EINVAL;
  if ((long long)d->B * d->Ti * d->Hi * d->Wi >= (1LL << 31)) return CVVAE_EUNSUPPORTED;
  if ((long long)d->B * (2LL * d->To) * d->Ho * d->Wo >= (1LL << 31)) return CVVAE_EUNSUPPORTED;  // 32-bit pixel indices
  return CVVAE_OK;
}

}  // namespace cvvae

using namespace cvvae;

extern "C" {

const char* cvvae_conv_kernel_name(const cvvae_conv_desc* d) {
  if (check_desc(d) != CVVAE_OK) return nullptr;
  const Instance* e = select_instance(d);
  return e ? instance_name(const_cast<Instance*>(e), d->dtype) : nullptr;
}

// channels per group of the tensor the conv stores, as a shift; -1 when (d, groups) cannot carry fused statistics
static int gn_shift_of(const cvvae_conv_desc* d, int groups) {
  if (groups <= 0 || d->out_mode == CVVAE_OUT_NCDHW || d->out_f32 || (d->Cout % 8)) return -1;
  const int cst = d->out_mode == CVVAE_OUT_TIME_SHUFFLE ? d->Cout / 2 : d->Cout;
  if (cst % groups) return -1;
  const int cpg = cst / groups;
  if (cpg < 4 || (cpg & (cpg - 1))) return -1;
  int sh = 0;
  while ((1 << sh) < cpg) ++sh;
  return sh;
}

int64_t cvvae_conv_gn_slabs(const cvvae_conv_desc* d, int32_t groups) {
  if (check_desc(d) != CVVAE_OK) return CVVAE_EINVAL;
  const int sh = gn_shift_of(d, groups);
  const Instance* e = select_instance(d);
  if (sh < 0 || !e) return CVVAE_EUNSUPPORTED;
  const int fold = d->upsample2x == 2;
  const long long tt_tiles = odd_frame_sibling(d, e) ? (d->To - 1) / 2 + 1 : cdiv(d->To, e->tt);
  const long long tiles = fold ? tt_tiles * cdiv(d->Ho / 2, e->th) * cdiv(d->Wo / 2, e->tw) * 4
                               : tt_tiles * cdiv(d->Ho, e->th) * cdiv(d->Wo, e->tw);
  return tiles * e->wm * e->kg * (d->out_mode == CVVAE_OUT_TIME_SHUFFLE ? 2 : 1) * (1LL << (sh - 2));
}

int cvvae_conv_fwd(const cvvae_conv_desc* d, const void* in, const void* w_packed, const float* bias, const void* residual,
                   const float* gn_scale, const float* gn_shift, void* out, void* stream) {
  return cvvae_conv_fwd_gn(d, in, w_packed, bias, residual, gn_scale, gn_shift, out, 0, nullptr, stream);
}

static int conv_impl(const cvvae_conv_desc* d, const void* in, const void* w_packed, const float* bias, const void* residual,
                     const float* gn_scale, const float* gn_shift, const void* sc_in, const void* sc_w, void* out,
                     int32_t out_groups, float* out_partials, void* stream);

int cvvae_conv_fwd_gn(const cvvae_conv_desc* d, const void* in, const void* w_packed, const float* bias, const void* residual,
                      const float* gn_scale, const float* gn_shift, void* out, int32_t out_groups, float* out_partials,
                      void* stream) {
  if (d && d->sc_Cin) return CVVAE_EINVAL;  // a descriptor with a shortcut goes through cvvae_conv_fwd_gn_sc
  return conv_impl(d, in, w_packed, bias, residual, gn_scale, gn_shift, nullptr, nullptr, out, out_groups, out_partials, stream);
}

int cvvae_conv_fwd_gn_sc(const cvvae_conv_desc* d, const void* in, const void* w_packed, const float* bias,
                         const float* gn_scale, const float* gn_shift, const void* sc_in, const void* sc_w_packed, void* out,
                         int32_t out_groups, float* out_partials, void* stream) {
  if (!d || !d->sc_Cin || !sc_in || !sc_w_packed) return CVVAE_EINVAL;
  return conv_impl(d, in, w_packed, bias, nullptr, gn_scale, gn_shift, sc_in, sc_w_packed, out, out_groups, out_partials, stream);
}

// Time tiles whose frames take a time fold (conv_kernel.h: fewer time groups) are SHORT.  Mixed tile durations leave the CUs
// finishing at different moments, and the launch ends ~0.7 tile times after its work is done; scheduling the short tiles LAST
// (longest-processing-time-first) bounds that tail by a short tile.  The kernel's block -> tile map places the leading
// t_short_lo and trailing t_short_hi time tiles of every spatial tile after all the long ones (per XCD); this mirrors the
// kernel's per-frame plan (tf_variant).
static void short_time_tiles(const cvvae_conv_desc* d, int TT, int KG, cvvae::ConvArgs& a, int phases) {
  a.t_short_lo = a.t_short_hi = 0;
  static const bool off = getenv("CVVAE_CONV_LPT") && atoi(getenv("CVVAE_CONV_LPT")) == 0;  // tuning aid
  if (off || d->kT != 3 || KG != 1 || a.order != 1 || a.tiles_t < 2) return;
  if (d->pad_mode_t == CVVAE_PAD_REPLICATE && !d->w_time_folds) return;
  auto folded = [&](int to) {  // the frame's plan has fewer than three time groups (tile_map.h)
    return cvvae::time_fold_plan(to, d->sT, d->pad_t, d->Ti, d->pad_mode_t == CVVAE_PAD_REPLICATE, d->w_time_folds != 0).ng != 3;
  };
  auto tile_short = [&](int i) {
    for (int tt = 0; tt < TT; ++tt) {
      const int to = a.t_begin + i * TT + tt;
      if (to < d->To && folded(to)) return true;
    }
    return false;
  };
  int lo = 0, hi = 0;
  while (lo < a.tiles_t && tile_short(lo)) ++lo;
  while (hi < a.tiles_t - lo && tile_short(a.tiles_t - 1 - hi)) ++hi;
  const long long inner = (long long)a.ntiles_n * phases;
  const long long size2 = (long long)d->B * a.tiles_h * a.tiles_w * (lo + hi) * inner;
  if (lo + hi == 0 || lo + hi >= a.tiles_t || size2 < 8) return;
  a.t_short_lo = lo;
  a.t_short_hi = hi;
}

static int conv_impl(const cvvae_conv_desc* d, const void* in, const void* w_packed, const float* bias, const void* residual,
                     const float* gn_scale, const float* gn_shift, const void* sc_in, const void* sc_w, void* out,
                     int32_t out_groups, float* out_partials, void* stream) {
  int rc = check_desc(d);
  if (rc != CVVAE_OK) return rc;
  if ((out_groups != 0) != (out_partials != nullptr)) return CVVAE_EINVAL;
  if (!in || !w_packed || !bias || !out) return CVVAE_EINVAL;
  if (d->prologue != CVVAE_PRO_NONE && (!gn_scale || !gn_shift)) return CVVAE_EINVAL;
  if (residual && d->out_mode != CVVAE_OUT_NDHWC) return CVVAE_EINVAL;
  const Instance* e = select_instance(d);
  if (!e) return CVVAE_EUNSUPPORTED;

  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in;
  a.w = w_packed;
  a.bias = bias;
  a.res = residual;
  a.gsc = gn_scale;
  a.gsh = gn_shift;
  a.out = out;
  a.B = d->B; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi;
  const int fold = d->upsample2x == 2;
  a.Tl = d->Ti; a.Hl = d->upsample2x == 1 ? 2 * d->Hi : d->Hi; a.Wl = d->upsample2x == 1 ? 2 * d->Wi : d->Wi;
  a.Cin = d->Cin;
  a.in_ps = d->in_pix_stride;
  a.To = d->To; a.Ho = fold ? d->Ho / 2 : d->Ho; a.Wo = fold ? d->Wo / 2 : d->Wo; a.Cout = d->Cout;  // per-phase grid when folded
  a.w_bstride = d->w_batch_stride / 2;
  if (sc_in) {
    a.in2 = sc_in;
    a.w2 = sc_w;
    a.in2_ps = d->sc_in_pix_stride;
    a.nchunks2 = d->sc_Cin / (16 * e->ksub);
  }
  const int xpm = d->dtype >= CVVAE_F32 ? 3 : 1;  // packed records per (k16, tap): the fp32 layouts carry three
  a.w_phase_stride = fold ? (long long)(cvvae_packed_weight_bytes(d->Cout, d->Cin, 4 * d->kT * (d->w_time_folds ? 2 : 1) * xpm) / 2) : 0;
  a.out_ps = d->out_pix_stride;
  a.pt = d->pad_t; a.ph = d->pad_h; a.pw = d->pad_w;
  a.mode_t = d->pad_mode_t; a.mode_hw = d->pad_mode_hw;
  const Instance* e2 = odd_frame_sibling(d, e);
  a.tiles_t = e2 ? (d->To - 1) / 2 : (int)cdiv(d->To, e->tt);
  a.tiles_h = (int)cdiv(a.Ho, e->th);
  a.tiles_w = (int)cdiv(a.Wo, e->tw);
  a.ntiles_n = (int)cdiv(d->Cout, 32LL * e->wn * e->nbw);
  a.nchunks = d->Cin / (16 * e->ksub);
  a.nblk32 = (d->Cout + 31) / 32;
  a.out_mode = d->out_mode;
  a.out_f32 = d->out_f32;
  a.gn_rpb = d->gn_rows_per_batch;
  // tuning / debug aids, read ONCE per process (a launch never calls getenv)
  static const int env_order = getenv("CVVAE_CONV_ORDER") ? (atoi(getenv("CVVAE_CONV_ORDER")) ? 1 : 0) : 1;
  static const int env_noshift = getenv("CVVAE_STATS_NOSHIFT") ? atoi(getenv("CVVAE_STATS_NOSHIFT")) : 0;
  a.order = env_order;
  a.alpha = d->alpha;
  a.stats_noshift = env_noshift;
  // (two knobs stay per launch because the GPU tests flip them inside one process: CVVAE_CONV_FORCE in select_instance and this one)
  if (const char* f = getenv("CVVAE_CONV_PHASE_SYNC")) a.phase_sync = atoi(f) ? 1 : 0;  // conv_kernel.h phase_sync
  a.w_taps = (d->upsample2x == 2 ? d->kT * 4 : d->kT * d->kH * d->kW) * (d->w_time_folds ? 2 : 1) * xpm;
  a.q6_scale = 1.0f;
  a.q6_eb = 127;
  a.q6_bound = d->dtype == CVVAE_F32Q6 ? d->act_bound_dev : nullptr;  // (a device-side bound: the kernel derives scale and byte)
  if (d->dtype == CVVAE_F32Q6 && !d->act_bound_dev) {  // activations: codes = value * 2^s with 2^s * act_bound <= 28 (e3m2's largest value)
    int ex = 0;
    const float fr = frexpf(28.0f / d->act_bound, &ex);  // 28 / bound = fr * 2^ex, fr in [0.5, 1)  ->  floor(log2) = ex - 1
    (void)fr;
    int sft = ex - 1;
    sft = sft < -100 ? -100 : (sft > 100 ? 100 : sft);
    a.q6_scale = ldexpf(1.0f, -sft);
    a.q6_eb = 127 - sft;
  }
  static const bool res_pre_off = getenv("CVVAE_RES_PRELOAD") && atoi(getenv("CVVAE_RES_PRELOAD")) == 0;  // tuning aid
  a.res_pre = (residual && d->alpha == 1.0f && !d->out_f32 && !res_pre_off) ? 1 : 0;
  // tuning aid (tools/tune_instances.py re-launches recorded calls under CVVAE_CONV_FORCE: the record table was sized for
  // the default instance, another one would overrun it)
  static const bool nostats = getenv("CVVAE_CONV_TUNE_NOSTATS") != nullptr;
  if (nostats) out_partials = nullptr;
  if (out_partials) {
    const int sh = gn_shift_of(d, out_groups);
    if (sh < 0) return CVVAE_EUNSUPPORTED;
    a.gnp = out_partials;
    a.gn_G = out_groups;
    a.gn_sh = sh;
    a.gn_slabs = (int)((long long)(a.tiles_t + (e2 ? 1 : 0)) * a.tiles_h * a.tiles_w * (fold ? 4 : 1) * e->wm * e->kg *
                       (d->out_mode == CVVAE_OUT_TIME_SHUFFLE ? 2 : 1) * (1LL << (sh - 2)));
  }
  const long long grid = (long long)d->B * a.tiles_t * a.tiles_h * a.tiles_w * a.ntiles_n * (fold ? 4 : 1);
  if (grid <= 0 || grid >= (1LL << 31)) return CVVAE_EUNSUPPORTED;
  short_time_tiles(d, e->tt, e->kg, a, (fold ? 4 : 1));
  if (fold) {  // weight-stationary windows of the folded upsample convs (tile_map.h); CVVAE_UPS_WINDOW: tuning aid
    static const int env_win = getenv("CVVAE_UPS_WINDOW") ? atoi(getenv("CVVAE_UPS_WINDOW")) : 16;  // measured: 0 -> 8 -> 16: 6.20 / 6.03 / 5.99 ms (256 -> 512 @9x256^2)
    a.ws_window = (a.ntiles_n * 4 >= 8) ? env_win : 0;
  }
  rc = e->fn[d->dtype](a, (int)grid, (hipStream_t)stream);
  if (rc != 0 || !e2) return rc;
  // the last (odd) frame, one-frame tiles
  a.t_begin = d->To - 1;
  a.tile_base = a.tiles_t * a.tiles_h * a.tiles_w;
  a.tiles_t = 1;
  short_time_tiles(d, 1, e2->kg, a, (fold ? 4 : 1));
  return e2->fn[d->dtype](a, (int)((long long)d->B * a.tiles_h * a.tiles_w * a.ntiles_n), (hipStream_t)stream);
}

}  // extern "C"
