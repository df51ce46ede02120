// explicit instantiations of conv_fwd_kernel, fast-fp32 instances with fp6 corrections (fp16 MFMA + bf6 K = 64 MFMA; conv_table.h XQ6_C: planar layout, eight fragments per wave: per-frame 3x3)
#include "conv_kernel.h"
#include "conv_table.h"
namespace cvvae {
#define CVVAE_INST(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  template int launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,3>(const ConvArgs&, int, hipStream_t);
CVVAE_CONV_XQ6_C(CVVAE_INST)
}  // namespace cvvae
