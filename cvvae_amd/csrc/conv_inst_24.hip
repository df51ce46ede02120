// explicit instantiations of conv_fwd_kernel, fast-fp32 fp6 instances as 2 x 4 register blocks (NB = 2, planar layout; conv_table.h XQ6_NB2)
#include "conv_kernel.h"
#include "conv_table.h"
namespace cvvae {
#define CVVAE_INST(KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS) \
  template int launch_conv<_Float16,KT,KH,KW,ST,SH,SW,TT,TH,TW,WM,WN,KG,KSUB,PRO,UPS,3,2>(const ConvArgs&, int, hipStream_t);
CVVAE_CONV_XQ6_NB2(CVVAE_INST)
}  // namespace cvvae
