/*
 * cvvae.h -- C ABI of libcvvae_hip.so: the MI355X (gfx950) kernels under the CV-VAE encode/decode path.
 *
 * The reference (AILab-CVC/CV-VAE) is pure PyTorch: it has no FFI / operator registry.  Its hot path
 * dispatches to ATen ops from Python.  Each entry point below replaces one ATen op family
 * at the call sites cited next to it; the Python host (cvvae_amd/) binds them with ctypes and mirrors the
 * reference's module API (models/modeling_vae.py) above them.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success, <0 for a bad argument / unsupported shape (CVVAE_E*),
 *     >0 for a hipError_t raised by the launch.  Nothing throws, allocates or synchronises.
 *   - pointers are raw DEVICE pointers owned by the caller; `stream` is a hipStream_t (NULL = default).
 *   - activations are NDHWC ("channels last 3d"): element (b,t,y,x,c) at ((b*T+t)*H+y)*W+x)*pix_stride + c.
 *   - dtype is the storage/MFMA operand type; accumulation and GroupNorm statistics are always fp32.
 */
#ifndef CVVAE_H_
#define CVVAE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVVAE_ABI_VERSION 13

/* cvvae dtype.  CVVAE_F32 = the reference's fp32 model path (from_pretrained without torch_dtype, models/modeling_vae.py:41-42
 * force_upcast): activations, residuals, outputs and the source weights are float; the kernels split every fp32 operand into
 * fp16 hi + lo and run each product as three fp16 MFMAs into fp32 accumulators ("split precision": ~1e-6 relative error at 3x
 * the MFMA work) -- the parity mode that meets north_star's |delta| <= 1e-3 on latents as a maximum.
 * CVVAE_F32Q = the same float tensors with the two correction terms of every product on the fp8 matrix pipe
 * (Whi.hi on the fp16 MFMA + bf8(Whi).bf8(lo) + bf8(Wlo).bf8(hi) on v_mfma_f32_32x32x64_f8f6f4, one per pair of taps): ~2^-14
 * relative error per product at 2x the MFMA time of a 16-bit model -- the CHEAPEST mode that meets the 1e-3 bound.  It exists
 * for convolutions with more than one tap (cvvae_pack_weights* and cvvae_conv_fwd* with kH*kW > 1, no fused shortcut); 1x1x1
 * layers of such a model run as CVVAE_F32.  Every other entry point takes CVVAE_F32 for float tensors.
 * CVVAE_F32Q6 = CVVAE_F32Q with the correction terms in the 6-bit format "bf6" (OCP MX e3m2) at FOUR times the fp16 rate per K
 * element (1.5x the MFMA time of a 16-bit model; same ~2^-14 relative error per product: the corrections only need 2-3 mantissa
 * bits).  e3m2 spans 9 binades, so the operands are scaled: the weights per output channel and pair of taps by the packer (the MFMA's
 * block scales), the activations by ONE power of two per launch derived from cvvae_conv_desc.act_bound, an upper bound of the
 * magnitude of the conv's operand (after the prologue) that the caller supplies -- e.g. 8 max|gamma| + max|beta| behind a
 * GroupNorm.  Values beyond the bound saturate in the CORRECTION terms only (those products fall back to ~2^-11 relative error).
 * Exists for the stride-1 3x3x3 and 1x3x3 convolutions with the GroupNorm + SiLU prologue; weights packed with the SAME dtype code. */
enum { CVVAE_F16 = 0, CVVAE_BF16 = 1, CVVAE_F32 = 2, CVVAE_F32Q = 3, CVVAE_F32Q6 = 4 };
enum { CVVAE_PAD_ZERO = 0, CVVAE_PAD_REPLICATE = 1 };         /* out-of-range taps */
enum { CVVAE_PRO_NONE = 0, CVVAE_PRO_GN_SILU = 1, CVVAE_PRO_GN = 2 };   /* fused prologue on the input */
enum { CVVAE_OUT_NDHWC = 0, CVVAE_OUT_NCDHW = 1, CVVAE_OUT_TIME_SHUFFLE = 2 };

enum {
  CVVAE_OK = 0,
  CVVAE_EINVAL = -1,      /* NULL / negative / inconsistent argument */
  CVVAE_EUNSUPPORTED = -2 /* shape or option combination with no kernel instance */
};

/* Kernel families of cvvae_conv_fwd: the granularity the consumed channel count (Cin, and Cin_pad of the packed weights)
 * must be a multiple of -- 16 for 3x3x3, 32 for 1x3x3, 128 for 1x1x1.  The packed-weight LAYOUT is the same for every
 * family: [Cout/32][Cin_pad/16][tap][64 lanes][8]. */
static inline int cvvae_conv_kchunk(int kT, int kH, int kW) {
  return (kT == 3 && kH == 3 && kW == 3) ? 16 : (kT == 1 && kH == 3 && kW == 3) ? 32 : (kT == 1 && kH == 1 && kW == 1) ? 128 :
         (kT == 3 && kH == 3 && kW == 1) ? 16 /* the row-packed first layer, see in_overlap */ :
         (kT == 3 && kH == 1 && kW == 1) ? 32 /* the taps-in-N last layer, see cvvae_conv_out_gather */ : 0;
}

/*
 * One convolution = one implicit GEMM on MFMA (v_mfma_f32_32x32x16_{bf16,f16}).
 * Replaces, in the reference: aten::convolution + the F.pad / F.interpolate / GroupNorm / SiLU / add that
 * surround it --
 *   CausalConv3d.forward   models/vae_blocks3d_sd3.py:81-104, models/vae_models.py:298-328
 *   Conv3d (replicate)     models/vae_blocks3d_sd3.py:16-46;  nn.Conv3d(padding=1) models/vae_models.py:361-362
 *   Conv2dWithExtraDim     models/vae_blocks3d_sd3.py:107-116, models/vae_models.py:331-340   (kT = 1)
 *   Downsample3D           models/vae_blocks3d_sd3.py:224-239, models/vae_models.py:251-263   (stride 2 / (1,2,2))
 *   Upsample3D             models/vae_blocks3d_sd3.py:314-364, models/vae_models.py:214-235   (upsample2x + TIME_SHUFFLE)
 *   GroupNorm+SiLU feeding a conv: ResnetBlock3D.forward models/vae_blocks3d_sd3.py:523-524,547-559 (prologue)
 *   residual add           models/vae_blocks3d_sd3.py:567, models/vae_models.py:410            (epilogue)
 *   nn.Linear / 1x1 Conv2d of the attention blocks (kT=kH=kW=1)
 */
typedef struct cvvae_conv_desc {
  int32_t dtype;              /* CVVAE_F16 | CVVAE_BF16: input, weights, residual, (non-f32) output; CVVAE_F32 / CVVAE_F32Q / CVVAE_F32Q6: float
                               * input / residual / shortcut input / output, weights packed from float with the SAME dtype code */
  /* input, NDHWC, as stored */
  int32_t B, Ti, Hi, Wi;
  int32_t Cin;                /* channels consumed; multiple of cvvae_conv_kchunk(); extra channels must have zero weights */
  int64_t in_pix_stride;      /* elements between consecutive pixels (>= Cin, multiple of 8) */
  int32_t upsample2x;         /* 1: the conv sees nearest x(1,2,2) of the stored input (never materialised; 27 taps gathered
                                 on the upsampled grid).  2: the same result from four 3x2x2 phase convolutions over the stored
                                 input with FOLDED weights (cvvae_pack_weights_upfold): 12 taps instead of 27; the only numeric
                                 difference is one rounding of each folded weight.  kT,kH,kW stay (3,3,3) (the reference op) */
  /* kernel */
  int32_t kT, kH, kW;         /* (3,3,3) | (1,3,3) | (1,1,1) */
  int32_t sT, sH, sW;         /* 1 or 2 each; stride > 1 only with (3,3,3) */
  int32_t pad_t, pad_h, pad_w;            /* FRONT padding per axis (back padding is implied by To/Ho/Wo) */
  int32_t pad_mode_t, pad_mode_hw;        /* CVVAE_PAD_* for taps that fall outside the (upsampled) input */
  /* prologue: y = x*scale[row,c] + shift[row,c] (then SiLU); zero-padded taps stay 0 */
  int32_t prologue;           /* CVVAE_PRO_* */
  int32_t gn_rows_per_batch;  /* 1: row = b (5-D GroupNorm); Ti: row = b*Ti + t (per-frame GroupNorm, kT must be 1) */
  /* output */
  int32_t To, Ho, Wo, Cout;   /* Cout = real output channels (weights are packed to a multiple of 32) */
  int32_t out_mode;           /* CVVAE_OUT_*; TIME_SHUFFLE: channel n*C+c of frame t -> frame 2t+n-1, channel c
                                 (C = Cout/2, frame -1 dropped): 'b (n c) t h w -> b c (t n) h w' then [:, :, 1:] */
  int32_t out_f32;            /* 1: store fp32 (NDHWC only), else dtype */
  int64_t out_pix_stride;     /* NDHWC modes: elements between pixels of `out` and of `residual` (multiple of 8) */
  float alpha;                /* out = alpha*acc + bias (+ residual) */
  int64_t w_batch_stride;     /* 0: one packed weight set for every batch item; else batch item b uses
                                 w_packed + b*w_batch_stride BYTES (a multiple of 16): per-frame K / V^T of the attention
                                 blocks, so QK^T and PV of all frames are one launch each */
  /* fused 1x1 shortcut (cvvae_conv_fwd_gn_sc; 1x3x3 stride-1 convolutions): channels and pixel stride of the second input */
  int32_t sc_Cin;             /* multiple of the instance's K-chunk (32) */
  int32_t w_time_folds;       /* 1: w_packed carries the time-fold slots (cvvae_pack_weights_tfolds / _upfold_tfolds; kT == 3 only) */
  int64_t sc_in_pix_stride;
  /* Row-packed input (the networks' first layer, conv_in: 3 -> 128 channels, models/vae_models3d_sd3.py:97-104, vae_models.py:706-716).
   * With in_overlap = 1 the "channel" vectors of consecutive pixels OVERLAP in memory: in_pix_stride (>= 4, multiple of 4) is
   * smaller than Cin, so pixel x's Cin = 16 channels are the 4 stored pixels x .. x+3 of 4 channels each.  A 3x3x3 convolution
   * over [B,T,H,W,3] then runs as a (kT,kH,kW) = (3,3,1) convolution over a W-padded copy [B,T,H,W+3,4] (cvvae_ncdhw_to_rowpack;
   * padding along W is in the copy, Wi = W + 3 is the STORED row length, Wo = W, pad_w = 0) whose 16 virtual channels are
   * (dx, c) = the three kW taps x 4 channel slots: K = 9 x 16 = 144 per output instead of 27 x 16 = 432 with the channels
   * padded 3 -> 16, and a 36 MB instead of a 142 MB input at 17 x 512^2.  The buffer must stay readable 32 bytes past its end. */
  int32_t in_overlap;
  float act_bound;            /* CVVAE_F32Q6 only: upper bound (> 0) of |operand| after the prologue; 0 otherwise */
  /* (ABI 13) 1: the four-wave conv instances (two workgroups resident per CU; csrc/conv_table.h G11) are candidates of the instance
   * choice for this launch, 0: they are not.  A per-launch field -- the library keeps NO state between calls (until ABI 12 this was a
   * process-wide switch, cvvae_conv_set_four_wave).  cvvae_conv_gn_slabs, cvvae_conv_kernel_name and the launch must see the same
   * value: the instance fixes the layout of the fused GroupNorm records. */
  int32_t four_wave;
  /* (ABI 13) CVVAE_F32Q6 with an operand that has NO GroupNorm in front (the folded upsample convs of a fast-fp32 model: their input
   * is the residual stream): device pointer to ONE float, an upper bound (> 0) of |operand| that the producer's stream computes (a
   * max-abs reduction) -- read by the kernel at launch time, so no host synchronisation is needed; act_bound must then be 0.
   * Elements beyond the bound saturate in the fp6 CORRECTION terms only (they fall back to the fp16 model's error). */
  const float* act_bound_dev;
} cvvae_conv_desc;

/* bytes of the packed weight buffer for (Cout, Cin, taps); includes the read-ahead tail the kernel needs */
size_t cvvae_packed_weight_bytes(int32_t Cout, int32_t Cin, int32_t taps);

/*
 * Pack weights into MFMA-fragment order: [Cout/32][Cin_pad/16][tap][64 lanes][8].  (Lane l of a record holds the 8 input
 * channels (l>>5)*8.. of output channel block*32 + sigma(l&31), sigma = l&31 with bits 2 and 3 swapped; the format is
 * private to the library: only cvvae_pack_weights* write it and only cvvae_conv_fwd* read it.)
 * src element (co, ci, tap) is read at src[co*s_co + ci*s_ci + tap*s_tap] (dtype elements), so the same entry
 * packs torch conv weights [Cout][Cin][kT*kH*kW] (s_co=Cin_src*taps, s_ci=taps, s_tap=1), nn.Linear weights, and
 * per-frame attention K / V^T matrices produced on the device.  co >= Cout_src or ci >= Cin_src pack as 0.
 */
int cvvae_pack_weights(int32_t dtype, const void* src, int32_t Cout_src, int32_t Cin_src, int32_t taps,
                       int64_t s_co, int64_t s_ci, int64_t s_tap, int32_t Cin_pad, int32_t kchunk, void* dst,
                       void* stream);

/* Fold + pack for upsample2x == 2: src = torch conv weight [Cout][Cin][3][3][3] (contiguous, dtype); dst = 4 phase buffers
 * of cvvae_packed_weight_bytes(Cout, Cin_pad, 12) bytes each, back to back (phase = 2*py + px). */
int cvvae_pack_weights_upfold(int32_t dtype, const void* src, int32_t Cout, int32_t Cin, int32_t Cin_pad, int32_t tfold,
                              void* dst, void* stream);

/*
 * Single-frame ("image mode", T = 1: the T2I pipeline's decode(latents, num_frames=1), BASELINE config 1) inputs: every
 * 3x3x3 conv of the path pads time to 3 frames, so its three time taps read the SAME frame (replicate padding:
 * CausalConv3d / Conv3d / Upsample3D / Downsample3D) or two of them read zeros (nn.Conv3d(padding=1) of the vae3d decoder).
 * The taps that coincide are summed into the weights (fp32 sum, one rounding) and the layer runs as a 1x3x3 (or 1x2x2
 * phase) convolution: 3x fewer MFMAs, same result up to that rounding.
 *   cvvae_pack_weights_fold = cvvae_pack_weights whose source element is the sum of fold_n elements s_fold apart
 *     (fold_n = 3, s_fold = kH*kW on a [Cout][Cin][3][kH][kW] weight: replicate time padding; fold_n = 1 with src advanced to
 *     the centre tap: zero time padding);
 *   tfold of cvvae_pack_weights_upfold: 0 = keep the 3 time taps (12 taps per phase), 1 = sum them, 2 = centre tap only
 *     (4 taps per phase; used with kT = 1, upsample2x = 2).
 */
/* batch of `batch` packings in one launch: item i reads src + i*s_batch elements and writes dst + i*dst_batch_stride bytes
 * (dst_batch_stride >= cvvae_packed_weight_bytes(Cout_src, Cin_pad, taps), multiple of 16) */
/* Time-fold slots (kT == 3, replicate time padding).  src taps are ordered tap = kt*nsp + sp (nsp = kH*kW spatial taps).  The
 * packed buffer holds 6*nsp taps per record group: the 3 original time slots, then W0+W1, W1+W2, W0+W1+W2 (fp32 sums rounded
 * once): at a clip boundary, where two or three time taps of an output frame read the same stored frame, cvvae_conv_fwd
 * (desc.w_time_folds = 1) multiplies that frame once with the summed slot.  Size: cvvae_packed_weight_bytes(Cout, Cin_pad,
 * 6*nsp).  _upfold_tfolds: the same for the four folded-upsample phase kernels (24 taps per phase). */
int cvvae_pack_weights_tfolds(int32_t dtype, const void* src, int32_t Cout_src, int32_t Cin_src, int32_t nsp, int64_t s_co,
                              int64_t s_ci, int64_t s_tap, int32_t Cin_pad, int32_t kchunk, void* dst, void* stream);
int cvvae_pack_weights_upfold_tfolds(int32_t dtype, const void* src, int32_t Cout, int32_t Cin, int32_t Cin_pad, void* dst,
                                     void* stream);
int cvvae_pack_weights_batched(int32_t dtype, const void* src, int32_t batch, int64_t s_batch, int32_t Cout_src,
                               int32_t Cin_src, int32_t taps, int64_t s_co, int64_t s_ci, int64_t s_tap, int32_t Cin_pad,
                               int32_t kchunk, void* dst, int64_t dst_batch_stride, void* stream);
int cvvae_pack_weights_fold(int32_t dtype, const void* src, int32_t Cout_src, int32_t Cin_src, int32_t taps, int64_t s_co,
                            int64_t s_ci, int64_t s_tap, int32_t fold_n, int64_t s_fold, int32_t Cin_pad, int32_t kchunk,
                            void* dst, void* stream);

int cvvae_conv_fwd(const cvvae_conv_desc* d, const void* in, const void* w_packed, const float* bias,
                   const void* residual, const float* gn_scale, const float* gn_shift, void* out, void* stream);

/*
 * The same convolution that ALSO emits GroupNorm statistics of the tensor it stores (NDHWC / TIME_SHUFFLE outputs of
 * dtype, Cout % 8 == 0), so that the GroupNorm of the NEXT layer never re-reads the activation: norm2 after conv1, norm1
 * of the next ResnetBlock3D after conv2 (+residual), conv_norm_out -- models/vae_blocks3d_sd3.py:523,547,
 * models/vae_models3d_sd3.py:204,382, models/vae_models.py:395,402,820,999 (5-D GroupNorm: statistics per sample).
 * out_partials: [B][out_groups][slabs][3] fp32 records (n, mean, M2) of the ROUNDED stored values, one per pixel tile /
 * wave slab / 4-channel slot, slabs = cvvae_conv_gn_slabs(d, out_groups); every record is written exactly once (no
 * atomics: results are bit-reproducible).  cvvae_gn_finalize merges them (Chan, fixed order) into the affine table.
 * With out_mode = TIME_SHUFFLE the caller ZERO-FILLS out_partials first: workgroups whose whole tile is the dropped frame
 * (channels [0, Cout/2) of conv frame 0) exit early and leave their records untouched (an all-zero record is empty).
 * out_groups = 0 and out_partials = NULL: identical to cvvae_conv_fwd.
 */
int64_t cvvae_conv_gn_slabs(const cvvae_conv_desc* d, int32_t out_groups);
int cvvae_conv_fwd_gn(const cvvae_conv_desc* d, const void* in, const void* w_packed, const float* bias,
                      const void* residual, const float* gn_scale, const float* gn_shift, void* out, int32_t out_groups,
                      float* out_partials, void* stream);
/*
 * ResnetBlock3D tail in ONE launch: conv2 (per-frame 3x3 over GroupNorm+SiLU(h)) + conv_shortcut / nin_shortcut (1x1 over the
 * block input x) + the add (models/vae_blocks3d_sd3.py:559-567, models/vae_models.py:404-410): out = conv2(h) + W_sc x + bias,
 * where `bias` is the caller's b_conv2 + b_shortcut.  sc_in is NDHWC [B,Ti,Hi,Wi] with d->sc_Cin channels (pixel stride
 * d->sc_in_pix_stride), sc_w_packed = cvvae_pack_weights of the [Cout][sc_Cin] shortcut weight (taps = 1).  The shortcut
 * tensor is never written to or read back from HBM.  Requires kT,kH,kW = (1,3,3), stride 1; no residual pointer.
 */
int cvvae_conv_fwd_gn_sc(const cvvae_conv_desc* d, const void* in, const void* w_packed, const float* bias,
                         const float* gn_scale, const float* gn_shift, const void* sc_in, const void* sc_w_packed, void* out,
                         int32_t out_groups, float* out_partials, void* stream);
/* [B,C,T,H,W] (C <= 4, any cvvae dtype) -> the row-packed first-layer input [B,T,H,W+3,4] of dtype dst_dtype: stored pixel xp of
 * a row holds input pixel xp - 1; xp = 0 and xp = W + 1 hold the W padding (pad_mode_w: replicate = the edge pixel, zero), xp = W + 2
 * and channel slots >= C are zero.  `out` must be (B*T*H*(W+3)*4 + 16) elements long (read-ahead of the last pixels). */
int cvvae_ncdhw_to_rowpack(int32_t src_dtype, int32_t dst_dtype, const void* in, int32_t B, int32_t C, int32_t T, int32_t H,
                           int32_t W, int32_t pad_mode_w, void* out, void* stream);
/*
 * The networks' LAST layer (conv_out: 128 -> 3 channels, 3x3x3; models/vae_models3d_sd3.py:319-321,384-386, vae_models.py:952-955,
 * 1000) with its nine SPATIAL taps moved into the GEMM's N axis.  A 32-column MFMA block would carry 3 useful output channels;
 * instead a (3,1,1) convolution (cvvae_conv_fwd, weights re-ordered by the caller, fp32 NDHWC output, out_f32 = 1) computes, for every
 * INPUT pixel, V[pixel][(dy*3+dx)*Cout + co] = sum over (dt, ci) of w[co][ci][dt][dy][dx] * act[t+dt-pad_t][pixel][ci] -- 9*Cout <= 32
 * columns, no spatial halo, every activation staged 1.5x (time halo) instead of 2.7x, a ninth of the MFMAs -- and this pass
 * finishes the convolution:   out[co][t][y][x] = bias[co] + sum over (dy, dx) of V[t][y+dy-1][x+dx-1][(dy*3+dx)*Cout + co]
 * with the spatial padding of the layer (replicate: clamped neighbour coordinates; zero: neighbours outside the frame add nothing),
 * summed in fp32 in the fixed order dy, dx.  V: fp32 [B,T,H,W,ldv] (ldv >= 9*Cout rounded up to 4, multiple of 4).  Exactly one of out_ncdhw ([B,Cout,T,H,W] of
 * `dtype`) and out_u8 (B = 1, Cout = 3: uint8 frames [T,H,W,3] = the scripts' (clamp(x,-1,1)+1)*127.5 -> uint8 on the value
 * rounded to `dtype`, cvvae_inference_video.py:47-50) is non-NULL.
 */
int cvvae_conv_out_gather(int32_t dtype, const float* V, int32_t B, int32_t T, int32_t H, int32_t W, int32_t Cout, int64_t ldv,
                          const float* bias, int32_t pad_mode_hw, void* out_ncdhw, uint8_t* out_u8, void* stream);
/*
 * Fused single-head attention core, head dimension 512 (csrc/attention_kernel.hip):  o = softmax(q k^T * scale) v  per batch item
 * (= frame).  Replaces F.scaled_dot_product_attention of diffusers' AttnProcessor2_0 under AttentionWithExtraDim
 * (models/vae_blocks3d_sd3.py:119-147) and xformers.ops.memory_efficient_attention (models/vae_models.py:518-520, 581-583).
 * q, k, o: [batch][N][512] of `dtype` (fp16 / bf16); vt: the TRANSPOSED values [batch][512][ldvt] (cvvae_transpose), ldvt >= N rounded
 * up to 32 and a multiple of 8, columns >= N zero.  fp32 accumulation, online softmax; probabilities rounded to `dtype` before the
 * second product (as the two-launch form did).
 */
int cvvae_attention_d512(int32_t dtype, const void* q, const void* k, const void* vt, void* o, int32_t batch, int32_t N,
                         int64_t ldvt, float scale, void* stream);
/*
 * One axis of the scripts' frame resize (`transforms.Resize(size=(height, width))` on the uint8 clip, cvvae_inference_video.py:14-16,28
 * = torch's antialiased bilinear interpolation of uint8 tensors): fixed-point triangle filter
 *   out[o][i][r] = clip_u8( ( (1 << (precision-1)) + sum_{j < xsize[i]} w[i*ksize + j] * in[o][xmin[i] + j][r] ) >> precision )
 * over a tensor viewed as [outer][in_size][inner] -> [outer][out_size][inner] (frames 't h w c': width pass inner = C, height pass
 * inner = W*C).  xmin / xsize / w (int32, device) and `precision` are the tables of that interpolation (cvvae_amd/ops.py
 * resize_tables: the filter support, the int16 weight scale and the rounding follow ATen's uint8 kernel, horizontal pass first).
 */
int cvvae_resize_u8_axis(const uint8_t* in, uint8_t* out, int64_t outer, int32_t in_size, int32_t out_size, int64_t inner,
                         const int32_t* xmin, const int32_t* xsize, const int32_t* w, int32_t ksize, int32_t precision, void* stream);
/* the same from an NDHWC tensor [B,T,H,W] of `dtype` with pixel stride pix_stride (>= C) elements: its first C (<= 4) channels */
int cvvae_ndhwc_to_rowpack(int32_t dtype, const void* in, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W, int64_t pix_stride,
                           int32_t pad_mode_w, void* out, void* stream);
int cvvae_gn_finalize(const float* partials, int32_t rows, int64_t slabs, int32_t C, int32_t groups, float eps,
                      const float* gamma, const float* beta, float* scale, float* shift, void* stream);
/* Per-FRAME statistics from the same records (the per-frame GroupNorm of the attention blocks, models/vae_blocks3d_sd3.py:119-147,
 * vae_models.py:500-537): valid when the records come from a per-frame convolution (kT = 1: one-frame tiles, frame-major record
 * order) over `frames` frames; slabs % frames == 0.  Tables [rows * frames][C]: row = sample * frames + frame. */
int cvvae_gn_finalize_frames(const float* partials, int32_t rows, int32_t frames, int64_t slabs, int32_t C, int32_t groups, float eps,
                             const float* gamma, const float* beta, float* scale, float* shift, void* stream);

/*
 * GroupNorm statistics -> per-(row, channel) affine table consumed by the conv prologue.
 * Replaces aten::native_group_norm: torch.nn.GroupNorm at models/vae_blocks3d_sd3.py:449,472 (5-D, row = b),
 * models/vae_models3d_sd3.py:150,315, models/vae_models.py:192-195, and the per-frame group_norm of the attention
 * blocks (rows = B*T, S = H*W).  x is [rows][S][C] (pix_stride elements per pixel); biased variance; fp32 Chan merge.
 *   scale[row,c] = gamma[c]*rstd(row,g(c)),  shift[row,c] = beta[c] - mean(row,g(c))*scale[row,c]
 * workspace: cvvae_gn_workspace_bytes(rows, groups, S) bytes.
 */
size_t cvvae_gn_workspace_bytes(int32_t rows, int32_t groups, int64_t S);
int cvvae_gn_stats(int32_t dtype, const void* x, int32_t rows, int64_t S, int32_t C, int64_t pix_stride, int32_t groups,
                   float eps, const float* gamma, const float* beta, float* scale, float* shift, void* workspace,
                   void* stream);

/*
 * GroupNorm-apply (+ SiLU) as its own pass: out[row][s][c] = act(x[row][s][c] * scale[row][c] + shift[row][c]), act = SiLU
 * (silu != 0) or identity, fp32 arithmetic, one rounding to the storage dtype -- the very values the conv prologue
 * (CVVAE_PRO_GN_SILU / CVVAE_PRO_GN) stages, so a conv over `out` without prologue is bit-identical to the fused form.
 * The fused prologue re-evaluates the activation for every halo copy and every N-tile of a pixel (4-8x on the 256/512-channel
 * 3x3x3 layers) and VALU work does not overlap the MFMA stream on gfx950; where that costs more than one read + write of the
 * activation, the host applies GroupNorm+SiLU once with this entry (engine.py prepass policy).
 * Replaces aten::native_group_norm's normalisation + aten::silu (models/vae_blocks3d_sd3.py:523-524,547-548).
 */
int cvvae_gn_silu_apply(int32_t dtype, const void* x, int32_t rows, int64_t S, int32_t C, int64_t pix_stride,
                        const float* scale, const float* shift, int32_t silu, void* out, void* stream);

/*
 * Input gradients through a FROZEN module (training-side codec use): the reference back-propagates the
 * latent-compatibility loss through the frozen 2-D constraint decoder into the latents (lvdm/models/autoencoder.py:1057-1069:
 * `self.constraint_decoder.requires_grad_(False)`, `xrec_2d = self.constraint_decoder(z)`), i.e. autograd's input-gradient
 * formulas of the decoder's ops with no weight gradients.  The convolutions' input gradients run on cvvae_conv_fwd* with
 * transposed, tap-flipped weights (cvvae_amd/grad.py); the three entries below are the remaining pieces.
 *
 * cvvae_gn_bwd_input: gradient w.r.t. x of  y = act(GroupNorm(x))  (act = SiLU when silu != 0), replacing
 * aten::native_group_norm_backward's input gradient (+ aten::silu_backward).  x, gy, gx (and the optional `add`, summed into the
 * result: the skip branch of a residual block) are [rows][S][C] tensors of `dtype`; rstd / nmean are the fp32 tables [rows][C]
 * cvvae_gn_finalize / cvvae_gn_stats produce with gamma = 1, beta = 0 (scale = rstd, shift = -mean * rstd); gamma / beta [C] the
 * module's affine parameters.  Statistics are taken per row over its S pixels and C / groups channels.  Deterministic (two passes,
 * partial sums merged in index order).  workspace: cvvae_gn_bwd_workspace_bytes(rows, groups, S) bytes.
 * Constraints: C % 8 == 0, (C / groups) % 4 == 0, 256 % (C / 8) == 0, groups <= 64 (else CVVAE_EUNSUPPORTED).
 */
int64_t cvvae_gn_bwd_workspace_bytes(int32_t rows, int32_t groups, int64_t S);
int cvvae_gn_bwd_input(int32_t dtype, const void* x, const void* gy, const void* add, int32_t rows, int64_t S, int32_t C,
                       int32_t groups, const float* rstd, const float* nmean, const float* gamma, const float* beta, int32_t silu,
                       void* gx, void* workspace, void* stream);

/*
 * cvvae_gn_bwd_input_params: cvvae_gn_bwd_input for a TRAINABLE norm -- the same input gradient, plus the affine gradients
 * dgamma[c] = sum gy act'(a) xh, dbeta[c] = sum gy act'(a) over rows x S (fp32 [C]); the sums ride on the reduction pass of the input
 * gradient (it forms gy act'(a) anyway: no extra pass over x and gy; cvvae_channel_sums computes the same from scratch).
 * workspace: cvvae_gn_bwd_params_workspace_bytes(rows, groups, S, C) bytes.
 */
int64_t cvvae_gn_bwd_params_workspace_bytes(int32_t rows, int32_t groups, int64_t S, int32_t C);
int cvvae_gn_bwd_input_params(int32_t dtype, const void* x, const void* gy, const void* add, int32_t rows, int64_t S, int32_t C,
                              int32_t groups, const float* rstd, const float* nmean, const float* gamma, const float* beta,
                              int32_t silu, void* gx, float* dgamma, float* dbeta, void* workspace, void* stream);

/*
 * Gradient of the row softmax of the attention blocks (aten::_softmax_backward_data + the score scale):
 * gs[r][j] = alpha * p[r][j] * (gp[r][j] - sum_k p[r][k] * gp[r][k]) for j < n_valid, 0 for n_valid <= j < ld_o.
 * p: probabilities [rows][ld_p] of `dtype` (cvvae_softmax_rows' output), gp: fp32 [rows][ld_g], gs: [rows][ld_o] of `dtype`.
 */
int cvvae_softmax_bwd_rows(int32_t dtype, const void* p, int64_t ld_p, const float* gp, int64_t ld_g, int64_t rows, int32_t n_valid,
                           float alpha, void* gs, int64_t ld_o, void* stream);

/*
 * Gradient of the nearest-neighbour x2 upsample in H and W (aten::upsample_nearest2d_backward, Upsample2D of
 * lvdm/modules/diffusionmodules/vae_blocks_sd3.py:178-230): out[n][y][x][c] = sum of g[n][2y+{0,1}][2x+{0,1}][c]; g is
 * [N][2H][2W][C], out [N][H][W][C], C % 8 == 0.
 */
int cvvae_upsample2x_sum(int32_t dtype, const void* g, int64_t N, int32_t H, int32_t W, int32_t C, void* out, void* stream);

/*
 * Training the 3-D networks themselves (second half): the reference's training step runs `z, xrec = self(x)`
 * through the TRAINABLE Encoder3D / Decoder3D (lvdm/models/autoencoder.py:1057-1090; models/vae_models3d_sd3.py:162-208), so
 * autograd also needs every layer's PARAMETER gradients.  Input gradients run on cvvae_conv_fwd* with transposed, tap-flipped
 * weights as above (replicate padding: the full correlation over the padded extent, then cvvae_pad_fold); the entries below are
 * the parameter-gradient pieces (cvvae_amd/grad3d.py).
 *
 * cvvae_conv_wgrad: dW[co][ci][tap] = sum over output pixels of gy[pixel][co] * a[source(pixel, tap)][ci] -- aten::convolution_backward's
 * weight gradient.  `d` describes the FORWARD convolution (B, Ti, Hi, Wi, Cin, in_pix_stride of `a`; kT/kH/kW in {1,3} with
 * kH == kW; strides 1 or 2 with sH == sW; FRONT pads and pad modes; To, Ho, Wo, Cout); prologue / output fields are ignored.
 * a: the operand the forward multiplied, NDHWC -- i.e. AFTER its GroupNorm + SiLU (cvvae_gn_silu_apply) but BEFORE padding: out-of-range
 * taps are mapped exactly as the forward maps them (replicate = clamp, zero = no contribution).  gy: NDHWC [B][To][Ho][Wo] with
 * gy_pix_stride elements per pixel.  dw: fp32 [Cout][Cin][kT*kH*kW] (PyTorch's layout; Cin = d->Cin, i.e. including any channel
 * padding of `a`).  dtype CVVAE_F16 / CVVAE_BF16: 16-bit operands on the matching MFMA; CVVAE_F32 (and the fast codes): float
 * tensors, every product as three bf16 MFMAs (hi/lo split of both operands, ~2^-16 relative).  Accumulation is fp32 and
 * deterministic (per-slab partial tiles summed in index order).  workspace: cvvae_conv_wgrad_workspace_bytes(d) bytes.
 * Cin % 8 == 0, Cout % 8 == 0.
 * cvvae_conv_wgrad_bias (ABI 12): the same launch also writes dbias[Cout] (fp32) = sum of gy over the pixels -- the bias half of
 * aten::convolution_backward -- where cvvae_conv_wgrad_fuses_bias(d) returns 1 (16-bit operands, 3x3 spatial taps: the sums are
 * one more MFMA per 16 pixels on the gy tile the kernel holds anyway); elsewhere it returns CVVAE_EUNSUPPORTED for dbias != NULL
 * and the caller sums gy with cvvae_channel_sums.  dbias = NULL is cvvae_conv_wgrad.
 */
int64_t cvvae_conv_wgrad_workspace_bytes(const cvvae_conv_desc* d);
int cvvae_conv_wgrad_fuses_bias(const cvvae_conv_desc* d);
int cvvae_conv_wgrad_bias(const cvvae_conv_desc* d, const void* a, const void* gy, int64_t gy_pix_stride, float* dw, float* dbias,
                          void* workspace, void* stream);
int cvvae_conv_wgrad(const cvvae_conv_desc* d, const void* a, const void* gy, int64_t gy_pix_stride, float* dw, void* workspace,
                     void* stream);

/*
 * Per-channel sums over rows x S pixels (aten::convolution_backward's bias gradient; aten::native_group_norm_backward's affine
 * gradients):  x == NULL:  sum1[c] = sum g[r][s][c]  (g with g_pix_stride elements per pixel);
 * x != NULL:  d beta = sum1[c] = sum g * act'(a),  d gamma = sum2[c] = sum g * act'(a) * xh  with xh = x * rstd[r][c] + nmean[r][c],
 * a = xh * gamma[c] + beta[c], act = SiLU when silu != 0 (tables as in cvvae_gn_bwd_input; x and g are [rows][S][C] of `dtype`).
 * Deterministic (two passes).  workspace: cvvae_channel_sums_workspace_bytes(rows, S, C) bytes.  C % 8 == 0, C <= 2048.
 */
int64_t cvvae_channel_sums_workspace_bytes(int32_t rows, int64_t S, int32_t C);
int cvvae_channel_sums(int32_t dtype, const void* x, const void* g, int64_t g_pix_stride, int32_t rows, int64_t S, int32_t C,
                       const float* rstd, const float* nmean, const float* gamma, const float* beta, int32_t silu, float* sum1,
                       float* sum2, void* workspace, void* stream);

/*
 * Backward of cvvae_temporal_attention (MemoryEfficientAttnVideoBlock.attention_t, models/vae_models.py:573-587; the vae3d decoder's
 * mid block): given go = dL/d(out), the gradients of q, k, v (all NDHWC [B][T][S pixels][C] of `dtype`), per pixel over its T <= 8
 * frames: dS = scale * P o (dP - rowsum(P o dP)) with P = softmax(q k^T * scale), dP = go v^T, scale = C^-1/2;
 * gq = dS k, gk = dS^T q, gv = P^T go.  Replaces aten::_scaled_dot_product_attention's backward.  T > 8: CVVAE_EUNSUPPORTED.
 */
int cvvae_temporal_attention_bwd(int32_t dtype, const void* q, const void* k, const void* v, const void* go, int32_t B, int32_t T,
                                 int64_t S, int32_t C, void* gq, void* gk, void* gv, void* stream);

/*
 * Adjoint of the padding in front of a convolution (aten::replication_pad3d_backward / constant_pad_nd's slice): gp is the gradient
 * w.r.t. the PADDED input, NDHWC [B][T + pad_t_front + pad_t_back][H + 2 pad_h][W + 2 pad_w][C]; out [B][T][H][W][C] receives, per
 * element, the sum of gp over every padded position the forward's coordinate map sends there (replicate: border elements collect
 * their pad region -- CausalConv3d's two leading copies of frame 0, models/vae_blocks3d_sd3.py:81-104; zero: the interior copy
 * only).  `add` (optional, same shape as out) is summed into the result.  C % 8 == 0.
 */
int cvvae_pad_fold(int32_t dtype, const void* gp, int32_t B, int32_t T, int32_t H, int32_t W, int32_t C, int32_t pad_t_front,
                   int32_t pad_t_back, int32_t pad_h, int32_t pad_w, int32_t pad_mode_t, int32_t pad_mode_hw, const void* add, void* out,
                   void* stream);

/* LayerNorm over C for every pixel (vae3d temporal attention, models/vae_models.py:571,575). in/out [P][C]. */
int cvvae_layernorm(int32_t dtype, const void* x, int64_t P, int32_t C, float eps, const float* gamma, const float* beta,
                    void* out, void* stream);

/* Row softmax of fp32 scores -> dtype probabilities (F.scaled_dot_product_attention's softmax).
 * s: [rows][ld_s] fp32, first n_valid columns are real; p: [rows][ld_p], columns >= n_valid written as 0. */
int cvvae_softmax_rows(int32_t dtype, const float* s, int64_t rows, int32_t n_valid, int64_t ld_s, void* p, int64_t ld_p,
                       void* stream);

/* 2-D transpose of dtype elements: out[c][r] = in[r][c], in [R][C] (ld_in), out [C][R] (ld_out). batched. */
int cvvae_transpose(int32_t dtype, const void* in, int32_t batch, int32_t R, int32_t C, int64_t ld_in, int64_t batch_stride_in,
                    void* out, int64_t ld_out, int64_t batch_stride_out, void* stream);

/* Temporal attention over the T frames of every pixel (vae3d MemoryEfficientAttnVideoBlock.attention_t,
 * models/vae_models.py:581-583), directly on NDHWC: q,k,v,out are [B][T][S][C] (S = H*W pixels, T <= 8),
 * softmax(q k^T * C^-0.5) v over t for each (b, s).  The reference's '(b h w) t c' rearrange is never materialised. */
int cvvae_temporal_attention(int32_t dtype, const void* q, const void* k, const void* v, int32_t B, int32_t T, int64_t S,
                             int32_t C, void* out, void* stream);

/* Layout at the drop-in boundary (the reference's tensors are NCDHW, models/modeling_vae.py):
 * in [B][C][T][H][W] (src_dtype: CVVAE_F16/BF16, or 2 = fp32) -> out NDHWC with Cpad >= C channels, pad = 0. */
int cvvae_ncdhw_to_ndhwc(int32_t src_dtype, int32_t dst_dtype, const void* in, int32_t B, int32_t C, int32_t T, int32_t H,
                         int32_t W, int32_t Cpad, void* out, void* stream);
int cvvae_ndhwc_to_ncdhw(int32_t dtype, const void* in, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                         int64_t pix_stride, void* out, void* stream);

/* Pixel pre/post-processing of the inference scripts on the device (cvvae_inference_video.py:24-38 and 47-50; SURVEY 8f
 * row 1), evaluated op by op in dtype exactly as the scripts' half tensors are:
 *   frames uint8 [T][H][W][3] (decord layout) -> NDHWC [1][T][H][W][Cpad] dtype = ((dtype)u8 / 127.5) - 1.0, pad channels 0
 *     (npix = T*H*W) -- the tensor conv_in consumes, so the NCDHW float clip is never built;
 *   decoder output NCDHW [1][3][T][H][W] dtype -> frames uint8 [T][H][W][3] = u8((clamp(x,-1,1) + 1.0) * 127.5)  (thw = T*H*W). */
int cvvae_frames_u8_to_ndhwc(int32_t dtype, const uint8_t* frames, int64_t npix, int32_t Cpad, void* out, void* stream);
int cvvae_ncdhw_to_frames_u8(int32_t dtype, const void* in, int64_t thw, uint8_t* frames, void* stream);

/* Tile blending, in place on b (blend_h / blend_v, models/modeling_vae.py:321-341,647-667): NCDHW tensors,
 * b[..., :o] = (1-w)*a[..., -o:] + w*b[..., :o], w = arange(o)/o in fp32.  axis: 0 = H (blend_v), 1 = W (blend_h).
 * rows = B*C*T.  a is [rows][Ha][Wa], b is [rows][Hb][Wb]. */
int cvvae_blend(int32_t dtype, const void* a, int32_t Ha, int32_t Wa, void* b, int32_t Hb, int32_t Wb, int64_t rows,
                int32_t overlap, int32_t axis, void* stream);

int cvvae_abi_version(void);
/* name of the kernel instance cvvae_conv_fwd would launch for d (for profiling reports); NULL if unsupported */
const char* cvvae_conv_kernel_name(const cvvae_conv_desc* d);

#ifdef __cplusplus
}
#endif
#endif /* CVVAE_H_ */
