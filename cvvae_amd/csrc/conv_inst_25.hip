// explicit instantiations of conv_fwd_kernel, group 14 (see conv_table.h)
#include "conv_kernel.h"
#include "conv_table.h"
namespace cvvae {
#define CVVAE_INST(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  template int launch_conv<__bf16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS>(const ConvArgs&, int, hipStream_t); \
  template int launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS>(const ConvArgs&, int, hipStream_t);
CVVAE_CONV_G14(CVVAE_INST)
}  // namespace cvvae
