// conv_table.h -- the list of conv_fwd_kernel instantiations (X-macro), grouped so that each group compiles
// in its own translation unit (conv_inst_N.hip) and the groups build in parallel.
//   X(KT,KH,KW, ST,SH,SW, TT,TH,TW, WM,WN, KSUB, PRO, UPS)
#pragma once

// 3x3x3 stride 1, K-chunk 16 channels.   A: BM=256 x BN=256   B: 256x128   C: 512x128 (2 frames)   D: 256x32
#define CVVAE_CONV_G1(X) \
  X(3,3,3, 1,1,1, 1,8,32, 1,8, 1, 0,false) \
  X(3,3,3, 1,1,1, 1,8,32, 1,8, 1, 1,false)
#define CVVAE_CONV_G2(X) \
  X(3,3,3, 1,1,1, 1,8,32, 2,4, 1, 0,false) \
  X(3,3,3, 1,1,1, 1,8,32, 2,4, 1, 1,false)
// (BM=512 tiles -- 2x8x32 for 3x3x3, 1x16x32 for 1x3x3 -- were measured: they need >256 VGPRs per wave in an
//  8-wave workgroup, hipcc spills ~200 registers and they run 30 % SLOWER than the BM=256 tiles; not built)
#define CVVAE_CONV_G3(X)
#define CVVAE_CONV_G4(X) \
  X(3,3,3, 1,1,1, 1,8,32, 8,1, 1, 0,false) \
  X(3,3,3, 1,1,1, 1,8,32, 8,1, 1, 1,false) \
  X(3,3,3, 1,1,1, 1,8,32, 1,8, 1, 0,true)
// strided 3x3x3 (encoder downsamplers): 64-pixel tile, BN = 256
#define CVVAE_CONV_G5(X) \
  X(3,3,3, 2,2,2, 1,4,16, 1,8, 1, 0,false) \
  X(3,3,3, 1,2,2, 1,4,16, 1,8, 1, 0,false)
// 1x3x3 per-frame conv (ResnetBlock conv2), K-chunk 32 channels
#define CVVAE_CONV_G6(X) \
  X(1,3,3, 1,1,1, 1,8,32, 1,8, 2, 0,false) \
  X(1,3,3, 1,1,1, 1,8,32, 1,8, 2, 1,false) \
  X(1,3,3, 1,1,1, 1,8,32, 2,4, 2, 0,false) \
  X(1,3,3, 1,1,1, 1,8,32, 2,4, 2, 1,false)
#define CVVAE_CONV_G7(X)
// 1x1x1 (shortcuts, attention projections and the QK^T / PV products), K-chunk 128 channels, 1-D pixel tile
#define CVVAE_CONV_G8(X) \
  X(1,1,1, 1,1,1, 1,1,256, 1,8, 8, 0,false) \
  X(1,1,1, 1,1,1, 1,1,256, 1,8, 8, 2,false) \
  X(1,1,1, 1,1,1, 1,1,256, 2,4, 8, 0,false) \
  X(1,1,1, 1,1,1, 1,1,256, 2,4, 8, 2,false)

#define CVVAE_CONV_ALL(X) \
  CVVAE_CONV_G1(X) CVVAE_CONV_G2(X) CVVAE_CONV_G3(X) CVVAE_CONV_G4(X) CVVAE_CONV_G5(X) CVVAE_CONV_G6(X) \
  CVVAE_CONV_G7(X) CVVAE_CONV_G8(X)
