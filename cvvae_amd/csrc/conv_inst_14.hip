// explicit instantiations of conv_fwd_kernel, fast-fp32 (fp16 MFMA + bf8 correction MFMA) instances XQ_A (see conv_table.h)
#include "conv_kernel.h"
#include "conv_table.h"
namespace cvvae {
#define CVVAE_INST(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  template int launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,2>(const ConvArgs&, int, hipStream_t);
CVVAE_CONV_XQ_A(CVVAE_INST)
}  // namespace cvvae
