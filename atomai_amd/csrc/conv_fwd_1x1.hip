// conv_fwd_1x1.hip — 1x1 instantiations of the MFMA convolution kernel (conv_kernel.h): the UpsampleBlock
// convolutions evaluated at low resolution (atomai/nets/blocks.py:122-132) and their data gradients.
#include "conv_kernel.h"

int amx_conv_launch_1x1(ConvFwdArgs& a, int nt, bool tail, hipStream_t s) {
#define GO(N_) return tail ? launch_conv_fwd<1, N_, 0, true, 4, 0, true>(a, s) \
                           : launch_conv_fwd<1, N_, 0, true, 4, 0, false>(a, s)
    if (nt == 1) GO(1);
    if (nt == 2) GO(2);
    GO(4);
#undef GO
}
