// conv_fwd_dil.hip — dilated 3x3 instantiations of the MFMA convolution kernel (conv_kernel.h): DilatedBlock
// (atomai/nets/blocks.py:300-318).  The halo bound sizes the prefetch registers and the LDS image, so one
// instantiation per dilation class (2, 3-4, 5-6) keeps the light dilations at a higher occupancy.
#include "conv_kernel.h"

int amx_conv_launch_dil(ConvFwdArgs& a, int nt, bool tail, hipStream_t s) {
#define GO(N_, H_) return tail ? launch_conv_fwd<9, N_, H_, false, 4, false, true>(a, s) \
                               : launch_conv_fwd<9, N_, H_, false, 4, false, false>(a, s)
    if (a.dil <= 2) { if (nt == 1) GO(1, 2); if (nt == 2) GO(2, 2); GO(4, 2); }
    if (a.dil <= 4) { if (nt == 1) GO(1, 4); if (nt == 2) GO(2, 4); GO(4, 4); }
    if (nt == 1) GO(1, 6);
    if (nt == 2) GO(2, 6);
    GO(4, 6);
#undef GO
}
