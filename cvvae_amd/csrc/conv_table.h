// conv_table.h -- the list of conv_fwd_kernel instantiations (X-macro), grouped so that each group compiles
// in its own translation unit (conv_inst_N.hip) and the groups build in parallel.
//   X(KT,KH,KW, ST,SH,SW, TT,TH,TW, WM,WN,KG, KSUB, PRO, UPS)
// UPS: 0 = none, 1 = nearest-2x gather fused into the staging (3x3x3 taps on the upsampled grid), 2 = nearest-2x folded
// into four 3x2x2 phase convolutions over the stored input (12 taps instead of 27: 2.25x fewer MFMAs, same result up to
// the rounding of the folded weights).
// WM x WN x KG = 8 waves: WM pixel slabs x WN 32-channel blocks x KG K-groups.  K-chunk = 16*KSUB channels.
// Order inside a family = preference when the cost model ties (first wins).
#pragma once

// 3x3x3 stride 1, BN = 256 (Cout >= 256): all 8 waves side by side in N; the 128-pixel tile is for small frames, where
// 256-pixel tiles leave the last round of workgroups mostly empty (e.g. 288 workgroups on 256 CUs) -- with 32-channel chunks first
// (round 6; a cost tie goes to the first: 16 instead of 32 rounds of stage -> barrier -> 108 MFMAs per wave: 512 -> 512 at 5x32x32
// 0.200 -> 0.163 ms, profiles/r6_probe_weight_ring.log -- where a twice as deep weight ring, the first suspect, made it SLOWER;
// the 256-pixel tile with 32-channel chunks is 163,200 of the 163,840 bytes of LDS and -0 / -1.2 / -2.0 / -2.7 % at 256 ch 9x256^2 /
// 512 ch 9x128^2 / 5x64^2 / 9x64^2, profiles/r6_chunk32.log)
#define CVVAE_CONV_G1(X) \
  X(3,3,3, 1,1,1, 1,8,32, 1,8,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 1,8,32, 1,8,1, 2, 1,0) \
  X(3,3,3, 1,1,1, 1,8,32, 1,8,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 1,4,32, 1,8,1, 2, 1,0) \
  X(3,3,3, 1,1,1, 1,4,32, 1,8,1, 1, 1,0)
// 3x3x3 stride 1, BN = 128 (Cout = 128): 4 N-blocks x 2 K-groups over a 32-channel chunk, reduced through LDS
#define CVVAE_CONV_G2(X) \
  X(3,3,3, 1,1,1, 1,8,32, 1,4,2, 2, 0,0) \
  X(3,3,3, 1,1,1, 1,8,32, 1,4,2, 2, 1,0)
// 3x3x3 stride 1, BN = 128 as 2 pixel slabs x 4 N-blocks: 256-pixel tile, and a 2-frame 512-pixel tile whose halo is
// 2.66 staged pixels per output pixel instead of 3.98 (every weight record feeds 8 MFMAs, no K-group reduction)
#define CVVAE_CONV_G3(X) \
  X(3,3,3, 1,1,1, 1,8,32, 2,4,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 1,8,32, 2,4,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 2,8,32, 2,4,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 2,8,32, 2,4,1, 1, 0,0)
// the row-packed first layer (cvvae_conv_desc.in_overlap: kW taps live in the 16 virtual channels), Cout = 128
#define CVVAE_CONV_G12(X) \
  X(3,3,1, 1,1,1, 2,8,32, 2,4,1, 1, 0,0) \
  X(3,3,1, 1,1,1, 1,8,32, 2,4,1, 1, 0,0)
// the taps-in-N last layer (include/cvvae.h cvvae_conv_out_gather): (3,1,1) over four-frame tiles, 27 of 32 columns useful
// (the 256-pixel tile -- 61 KB of LDS -- leaves room for TWO workgroups per CU: one's staging / store runs under the other's K
//  loop.  Measured at 128 -> 27 columns @17x512^2: 0.71 ms; the 512-pixel tile 0.89 ms, 64-channel chunks 1.09 ms)
#define CVVAE_CONV_G13(X) \
  X(3,1,1, 1,1,1, 4,4,16, 8,1,1, 2, 1,0)
// BN = 32 (conv_out: Cout = 3 / 8 / 32) and the fused nearest-2x upsample conv
#define CVVAE_CONV_G4(X) \
  X(3,3,3, 1,1,1, 1,8,32, 8,1,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 1,8,32, 8,1,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 2,8,32, 8,1,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 1,8,32, 1,8,1, 1, 0,1)
// strided 3x3x3 (encoder downsamplers): 64-pixel tile, BN = 256
// folded upsample convs (KH = KW = 2 taps per phase), BN = 256.  32-channel chunks first (the cost model ties; the first wins): a
// 16-channel chunk is only 96 MFMAs per wave between two barriers.  Round 1 measured the 16-channel form 5 % faster; with the
// weight-stationary tile windows of round 3 the 32-channel form wins: 256 -> 512 @9x256^2 6.18 -> 5.74 ms, 512 -> 512 2.89 -> 2.83 ms
// (profiles/r3_ab_ups_chunk.log)
#define CVVAE_CONV_G9(X) \
  X(3,2,2, 1,1,1, 1,8,32, 1,8,1, 2, 0,2) \
  X(3,2,2, 1,1,1, 1,8,32, 1,8,1, 1, 0,2)
// image mode (T = 1, time taps folded into the weights): strided per-frame conv and the 1x2x2 upsample phases
#define CVVAE_CONV_G10(X) \
  X(1,3,3, 1,2,2, 1,4,16, 1,8,1, 2, 0,0) \
  X(1,2,2, 1,1,1, 1,8,32, 1,8,1, 2, 0,2)
// (the 64-pixel tile also with 32-channel K-chunks: the 128-pixel one would need 2 x 134 KiB of LDS)
// (Cout = 128 -- the first downsampler -- leaves half of the BN = 256 instances' waves without channels; 128-pixel tiles as 2 pixel
//  slabs x 4 N-blocks, two-frame or 8-row, with every wave busy were measured in round 3 and are SLOWER: 0.90-0.92 vs 0.85 ms)
#define CVVAE_CONV_G5(X) \
  X(3,3,3, 2,2,2, 1,4,16, 1,8,1, 1, 0,0) \
  X(3,3,3, 1,2,2, 1,4,16, 1,8,1, 1, 0,0) \
  X(3,3,3, 2,2,2, 1,8,16, 1,8,1, 1, 0,0) \
  X(3,3,3, 1,2,2, 1,8,16, 1,8,1, 1, 0,0) \
  X(3,3,3, 2,2,2, 1,4,16, 1,8,1, 2, 0,0) \
  X(3,3,3, 1,2,2, 1,4,16, 1,8,1, 2, 0,0)
// 1x3x3 per-frame conv (ResnetBlock conv2), K-chunk 32 channels -- and 64 for the prologue form when Cin allows it (first = preferred
// on a cost tie: 288 instead of 144 MFMAs per wave between two barriers; round 3: 256 ch 0.701 -> 0.682 ms, 512 ch 0.637 -> 0.632 ms)
#define CVVAE_CONV_G6(X) \
  X(1,3,3, 1,1,1, 1,8,32, 1,8,1, 2, 0,0) \
  X(1,3,3, 1,1,1, 1,8,32, 1,8,1, 4, 1,0) \
  X(1,3,3, 1,1,1, 1,8,32, 1,8,1, 2, 1,0) \
  X(1,3,3, 1,1,1, 1,8,32, 1,4,2, 2, 0,0) \
  X(1,3,3, 1,1,1, 1,8,32, 1,4,2, 2, 1,0)
// 1x3x3 for 128-channel layers: 2 pixel slabs x 4 N-blocks; the 16-row (512-pixel) tile halves the tiles and their
// prologue / epilogue share (measured +9 % at 17x512^2), the 8-row one covers small frames
#define CVVAE_CONV_G7(X) \
  X(1,3,3, 1,1,1, 1,8,32, 2,4,1, 2, 1,0) \
  X(1,3,3, 1,1,1, 1,16,32, 2,4,1, 2, 1,0)
// small frames (round 6: image mode at 256^2, the 32x32 / 64x64 levels of a 17x256^2 clip): with 256-pixel tiles a 512-channel
// per-frame conv at 32x32 is 16 workgroups on 256 CUs and takes the same 61-66 us as the 80-workgroup layer at 5x32x32 -- the time of
// ONE workgroup's 16 K chunks (profiles/r6_probe_small_layers.log: staging 4.5k + MFMAs 3.2k cycles per chunk and wave).  128- and
// 64-pixel tiles put the same work on 4x as many CUs.  (The 3x3x3 layers of those levels stream 14 MB of weights per workgroup
// column and want FEWER, larger tiles: not here.)  64-channel chunks first (a cost tie goes to the first): 8 instead of 16 rounds of
// load -> GroupNorm + SiLU -> LDS -> barrier, -8 % at 512 channels; 128-channel chunks (74 KB of LDS) another -2 %: not kept
// (profiles/r6_small_layers_v6.log).  K-group forms of the 64- and 128-pixel tiles: half the weight stream per wave, which is the floor
// of a small tile at 512 input channels (-25 %, profiles/r6_small_layers_kg2.log).
#define CVVAE_CONV_G14(X) \
  X(1,3,3, 1,1,1, 1,2,32, 2,4,1, 4, 1,0) \
  X(1,3,3, 1,1,1, 1,2,32, 1,8,1, 4, 1,0) \
  X(1,3,3, 1,1,1, 1,2,32, 1,4,2, 4, 1,0) \
  X(1,3,3, 1,1,1, 1,4,32, 1,4,2, 4, 1,0) \
  X(1,3,3, 1,1,1, 1,4,32, 2,4,1, 2, 1,0) \
  X(1,3,3, 1,1,1, 1,2,32, 2,4,1, 2, 1,0) \
  X(1,3,3, 1,1,1, 1,2,32, 1,8,1, 2, 1,0) \
  X(1,2,2, 1,1,1, 1,4,32, 1,8,1, 2, 0,2) \
  X(1,1,1, 1,1,1, 1,1,64, 2,4,1, 8, 0,0) \
  X(1,1,1, 1,1,1, 1,1,64, 2,4,1, 8, 2,0)
// 1x1x1 (shortcuts, attention projections and the QK^T / PV products), K-chunk 128 channels, 1-D pixel tile
#define CVVAE_CONV_G8(X) \
  X(1,1,1, 1,1,1, 1,1,256, 1,8,1, 8, 0,0) \
  X(1,1,1, 1,1,1, 1,1,256, 1,8,1, 8, 2,0) \
  X(1,1,1, 1,1,1, 1,1,256, 2,4,1, 8, 0,0) \
  X(1,1,1, 1,1,1, 1,1,256, 2,4,1, 8, 2,0)

// FOUR-wave instances (WM x WN x KG = 4: 256 threads, <= 80 KiB of LDS, two workgroups resident per CU) for the 128-channel
// layers at full resolution, whose staging VALU work (GroupNorm + SiLU of every halo element for only 128 output channels) and store
// tail are too large a share of a tile to hide inside one workgroup: see conv_kernel.h (NWV) and DESIGN.md section 3.1.
// The per-frame instance (+6 % on the ResnetBlock conv2 of the 128-channel level) is built, and a candidate of the instance choice when
// the launch's descriptor says so (cvvae_conv_desc.four_wave; the Python layer sets it unless CVVAE_FOUR_WAVE=0).  History: round 2
// saw about one fused GroupNorm record in 10^4 come out wrong with two workgroups co-resident on one box of the pool; rounds 3-6
// could not reproduce it on any other box with either tree (150-repetition stresses, tools/probes/nw4_stress.py; the per-device
// self-check of round 5 never failed) -- the bit-for-bit repetition check now runs as a GPU TEST
// (tests/test_gpu_round6.py::test_four_wave_records_are_reproducible_under_co_residency), not at model load.  The two 3x3x3 forms
// were slower than the 8-wave two-frame tile (-4 %) and stay behind make NW4=1.
#ifdef CVVAE_BUILD_NW4
#define CVVAE_CONV_G11(X) \
  X(1,3,3, 1,1,1, 1,8,32, 1,4,1, 2, 1,0) \
  X(3,3,3, 1,1,1, 2,8,16, 1,4,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 1,8,16, 1,4,1, 1, 1,0)
#else
#define CVVAE_CONV_G11(X) \
  X(1,3,3, 1,1,1, 1,8,32, 1,4,1, 2, 1,0)
#endif

// Split-precision (XP) instances for fp32 models: fp32 activations in HBM, every product as three fp16 MFMAs (conv_kernel.h).
// One instance per kernel family (a pixel occupies twice the LDS, hence the smaller tiles); T = _Float16 only.
// (2 x 4 x 32 two-frame tiles as 2 pixel slabs x 4 N-blocks for the 128-channel layers -- with the BN = 256 instances half of
//  the waves of such a layer had no output channels -- and as 8 pixel slabs x 1 N-block for conv_out, Cout <= 32)
#define CVVAE_CONV_XP(X) \
  X(3,3,3, 1,1,1, 1,4,32, 1,8,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 1,4,32, 1,8,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 2,4,32, 2,4,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 2,4,32, 2,4,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 2,4,32, 8,1,1, 1, 1,0) \
  X(1,3,3, 1,1,1, 1,8,32, 2,4,1, 2, 1,0) \
  X(3,3,3, 2,2,2, 1,4,16, 1,8,1, 1, 0,0) \
  X(3,3,3, 1,2,2, 1,4,16, 1,8,1, 1, 0,0) \
  X(1,3,3, 1,1,1, 1,4,32, 1,8,1, 2, 0,0) \
  X(1,3,3, 1,1,1, 1,4,32, 1,8,1, 2, 1,0) \
  X(1,1,1, 1,1,1, 1,1,128, 1,8,1, 4, 0,0) \
  X(1,1,1, 1,1,1, 1,1,128, 1,8,1, 4, 2,0) \
  X(3,2,2, 1,1,1, 1,4,32, 1,8,1, 2, 0,2) \
  X(3,2,2, 1,1,1, 1,4,32, 1,8,1, 1, 0,2) \
  X(1,3,3, 1,2,2, 1,4,16, 1,8,1, 2, 0,0) \
  X(1,2,2, 1,1,1, 1,4,32, 1,8,1, 2, 0,2) \
  X(3,1,1, 1,1,1, 4,4,16, 8,1,1, 2, 1,0)
// (the last row: the taps-in-N last layer, G13, for fp32 models -- exact AND fast: a (3,1,1) kernel has single-tap runs, which the
//  fast form cannot pair, so both modes run it as three fp16 MFMAs per product: 9 MFMA units per k16 step instead of the 27-tap
//  layer's 55 (fast) / 81 (exact) for 3 of 32 useful output columns)
// (split in two translation units for the parallel build)
#define CVVAE_CONV_XP_A(X) \
  X(3,3,3, 1,1,1, 1,4,32, 1,8,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 1,4,32, 1,8,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 2,4,32, 2,4,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 2,4,32, 2,4,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 2,4,32, 8,1,1, 1, 1,0) \
  X(3,3,3, 2,2,2, 1,4,16, 1,8,1, 1, 0,0) \
  X(3,3,3, 1,2,2, 1,4,16, 1,8,1, 1, 0,0) \
  X(3,2,2, 1,1,1, 1,4,32, 1,8,1, 2, 0,2) \
  X(3,2,2, 1,1,1, 1,4,32, 1,8,1, 1, 0,2)
#define CVVAE_CONV_XP_B(X) \
  X(3,1,1, 1,1,1, 4,4,16, 8,1,1, 2, 1,0) \
  X(1,3,3, 1,1,1, 1,8,32, 2,4,1, 2, 1,0) \
  X(1,3,3, 1,1,1, 1,4,32, 1,8,1, 2, 0,0) \
  X(1,3,3, 1,1,1, 1,4,32, 1,8,1, 2, 1,0) \
  X(1,1,1, 1,1,1, 1,1,128, 1,8,1, 4, 0,0) \
  X(1,1,1, 1,1,1, 1,1,128, 1,8,1, 4, 2,0) \
  X(1,3,3, 1,2,2, 1,4,16, 1,8,1, 2, 0,0) \
  X(1,2,2, 1,1,1, 1,4,32, 1,8,1, 2, 0,2)

// TWO N-blocks per wave (NB = 2, conv_kernel.h): every activation fragment read from LDS feeds two MFMAs.  WM x WN waves cover
// BM pixels x 64*WN channels.  16-bit models only.
//   X(KT,KH,KW, ST,SH,SW, TT,TH,TW, WM,WN,KG, KSUB, PRO, UPS)
// NOT BUILT BY DEFAULT (make NB2=1 / -DCVVAE_BUILD_NB2).  Measured in round 3 (profiles/r3_ab_nb2.log, interleaved A/B on one box):
// bit-identical results, and NO gain -- 128 -> 128 @17x512^2 3.49 vs 3.47 ms, 256 -> 256 @9x256^2 1.64 vs 1.62 ms, folded upsample
// +-0 %, per-frame convs 4-50 % slower.  The K loop is not bound by LDS operand bandwidth.
#ifdef CVVAE_BUILD_NB2
#define CVVAE_CONV_NB2_A(X) \
  X(3,3,3, 1,1,1, 2,8,32, 4,2,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 2,8,32, 4,2,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 1,8,32, 4,2,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 1,8,32, 2,4,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 1,8,32, 2,4,1, 1, 1,0)
#define CVVAE_CONV_NB2_B(X) \
  X(3,2,2, 1,1,1, 1,8,32, 2,4,1, 1, 0,2) \
  X(1,3,3, 1,1,1, 1,8,32, 2,4,1, 2, 0,0) \
  X(1,3,3, 1,1,1, 1,8,32, 2,4,1, 2, 1,0) \
  X(1,3,3, 1,1,1, 1,16,32, 4,2,1, 2, 1,0)
#else
#define CVVAE_CONV_NB2_A(X)
#define CVVAE_CONV_NB2_B(X)
#endif
#define CVVAE_CONV_NB2(X) CVVAE_CONV_NB2_A(X) CVVAE_CONV_NB2_B(X)

// Fast-fp32 (XP = 2) instances: the multi-tap families of the list above (1x1x1 layers of such a model stay on the XP list)
#define CVVAE_CONV_XQ_A(X) \
  X(3,3,3, 1,1,1, 1,4,32, 1,8,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 1,4,32, 1,8,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 2,4,32, 2,4,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 2,4,32, 2,4,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 2,4,32, 8,1,1, 1, 1,0) \
  X(3,3,3, 2,2,2, 1,4,16, 1,8,1, 1, 0,0) \
  X(3,3,3, 1,2,2, 1,4,16, 1,8,1, 1, 0,0) \
  X(3,2,2, 1,1,1, 1,4,32, 1,8,1, 2, 0,2) \
  X(3,2,2, 1,1,1, 1,4,32, 1,8,1, 1, 0,2)
#define CVVAE_CONV_XQ_B(X) \
  X(1,3,3, 1,1,1, 1,8,32, 2,4,1, 2, 1,0) \
  X(1,3,3, 1,1,1, 1,4,32, 1,8,1, 2, 0,0) \
  X(1,3,3, 1,1,1, 1,4,32, 1,8,1, 2, 1,0) \
  X(1,3,3, 1,2,2, 1,4,16, 1,8,1, 2, 0,0) \
  X(1,2,2, 1,1,1, 1,4,32, 1,8,1, 2, 0,2)
#define CVVAE_CONV_XQ(X) CVVAE_CONV_XQ_A(X) CVVAE_CONV_XQ_B(X)
// Fast-fp32 with fp6 corrections (XP = 3, dtype CVVAE_F32Q6): the GroupNorm + SiLU prologue instances of the list above -- the layers
// whose operand has a bound the host can derive from the GroupNorm affine (cvvae_conv_desc.act_bound)
// The first list: the 256- / 128-pixel tiles of rounds 3-5 (register-staged 80-byte pixel, four fragments per wave), which also
// serve small frames and the odd-frame sibling launch.  The second list (round 6): the PLANAR layout (conv_kernel.h Geo::PL -- hi
// planes and code planes in separate LDS regions, 56 bytes per pixel and 16 channels) with EIGHT fragments per wave: the two-frame
// 512-pixel tile of the 128-channel 3x3x3 layers (+ its one-frame sibling on the old layout), the all-waves-in-N 256-pixel tiles
// of the 256- / 512-channel 3x3x3 layers.
#define CVVAE_CONV_XQ6_A(X) \
  X(3,3,3, 1,1,1, 1,4,32, 1,8,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 2,4,32, 2,4,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 2,4,32, 8,1,1, 1, 1,0) \
  X(1,3,3, 1,1,1, 1,8,32, 2,4,1, 2, 1,0) \
  X(1,3,3, 1,1,1, 1,4,32, 1,8,1, 2, 1,0) \
  X(3,2,2, 1,1,1, 1,4,32, 1,8,1, 2, 0,2) \
  X(3,2,2, 1,1,1, 1,4,32, 1,8,1, 1, 0,2)
// (the last two, round 6: the folded upsample convs -- no GroupNorm in front, the bound comes from a device-side max-abs reduction,
//  cvvae_conv_desc.act_bound_dev)
#define CVVAE_CONV_XQ6_B(X) \
  X(3,3,3, 1,1,1, 2,8,32, 2,4,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 1,8,32, 2,4,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 1,8,32, 1,8,1, 1, 1,0)
// (per-frame planar tiles were built and measured too: the 16-row tile of the 128-channel conv2 layers, X(1,3,3, 1,1,1, 1,16,32,
//  2,4,1, 2, 1,0): 3.40 vs 3.14 ms for the 8-row four-fragment tile at 17x512^2 with residual + statistics -- its K loop is short
//  (10 pairs of taps per chunk) and the longer store tail of a 512-pixel tile is exposed; the all-waves-in-N 256-pixel tile,
//  X(1,3,3, 1,1,1, 1,8,32, 1,8,1, 2, 1,0), compiles with an accumulator spilled inside its K loop.  Neither is in the list;
//  profiles/r6_ab_planar_fast_fp32_v2.log)
#define CVVAE_CONV_XQ6(X) CVVAE_CONV_XQ6_A(X) CVVAE_CONV_XQ6_B(X)
// ... and as a 2 x 4 register block (NB = 2: every LDS operand read feeds two MFMAs; conv_kernel.h Geo::PL): the 512-pixel two-frame
// tile as 4 pixel slabs x 2 waves x 2 blocks (+ its one-frame sibling), the 256-pixel tile as 2 slabs x 4 waves x 2 blocks, the 16-row
// per-frame tile of the 128-channel conv2 layers, the folded upsample phases
#define CVVAE_CONV_XQ6_NB2(X) \
  X(3,3,3, 1,1,1, 2,8,32, 4,2,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 1,8,32, 4,2,1, 1, 1,0) \
  X(3,3,3, 1,1,1, 1,8,32, 2,4,1, 1, 1,0) \
  X(1,3,3, 1,1,1, 1,16,32, 4,2,1, 2, 1,0) \
  X(3,2,2, 1,1,1, 1,8,32, 2,4,1, 1, 0,2)
// (measured against the tiles of rounds 3-5, profiles/r6_ab_planar_nb2{,_more}.log: 128-channel 3x3x3 +4 %, 256 / 512-channel 3x3x3
//  +1.5 %, the 128-channel per-frame conv on the 16-row tile +5.7 %, the folded upsample convs +5 %; the 256-pixel per-frame tile,
//  X(1,3,3, 1,1,1, 1,8,32, 2,4,1, 2, 1,0), +2.6 % at 256 channels and -2 % at 512: not in the list)

// DMA-staged instances (LD = 1, conv_kernel.h): 16-bit models, PRO = 0 -- the folded upsample convs, the strided downsamplers, the
// 1x1 layers and the decoder's conv_in as they are, and every other conv when its GroupNorm + SiLU is applied by the pass
// (cvvae_gn_silu_apply; engine.prepass) instead of in the staging.  Same tiles as the register-staged instances they shadow; the
// selection prefers them unless CVVAE_CONV_DMA=0 (A/B aid).  Split in three translation units for the parallel build.
//   X(KT,KH,KW, ST,SH,SW, TT,TH,TW, WM,WN,KG, KSUB, PRO, UPS)
#define CVVAE_CONV_LD_A(X) \
  X(3,3,3, 1,1,1, 2,8,32, 2,4,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 1,8,32, 2,4,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 1,8,32, 1,8,1, 1, 0,0) \
  X(3,3,3, 1,1,1, 1,4,32, 1,8,1, 1, 0,0)
// (the strided downsamplers were built and measured too: 1.28-1.31 ms per cfg 3 step DMA-staged against 1.22-1.26 register-staged --
//  their 64-pixel tiles spend the chunk waiting for the wave-loads either way, and the deeper hand-kept weight ring costs them
//  occupancy; not in the list.  profiles/r5_ab_dma_staging.log)
#define CVVAE_CONV_LD_B(X) \
  X(3,2,2, 1,1,1, 1,8,32, 1,8,1, 2, 0,2) \
  X(3,2,2, 1,1,1, 1,8,32, 1,8,1, 1, 0,2)
// (1x1x1 was in the list too: 0.073 vs 0.071 ms per cfg 3 step, and 0.026 vs 0.019 ms on the 1024-token layers of cfg 1 / 2: taken out)
#define CVVAE_CONV_LD_C(X) \
  X(1,3,3, 1,1,1, 1,16,32, 2,4,1, 2, 0,0) \
  X(1,3,3, 1,1,1, 1,8,32, 2,4,1, 2, 0,0) \
  X(1,3,3, 1,1,1, 1,8,32, 1,8,1, 2, 0,0) \
  X(1,3,3, 1,1,1, 1,8,32, 1,8,1, 4, 0,0)
#define CVVAE_CONV_LD(X) CVVAE_CONV_LD_A(X) CVVAE_CONV_LD_B(X) CVVAE_CONV_LD_C(X)

#define CVVAE_CONV_ALL(X) \
  CVVAE_CONV_G1(X) CVVAE_CONV_G2(X) CVVAE_CONV_G3(X) CVVAE_CONV_G4(X) CVVAE_CONV_G5(X) CVVAE_CONV_G6(X) \
  CVVAE_CONV_G7(X) CVVAE_CONV_G8(X) CVVAE_CONV_G9(X) CVVAE_CONV_G10(X) CVVAE_CONV_G11(X) CVVAE_CONV_G12(X) CVVAE_CONV_G13(X) \
  CVVAE_CONV_G14(X)
