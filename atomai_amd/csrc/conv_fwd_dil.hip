// conv_fwd_dil.hip — dilated 3x3 instantiations of the MFMA convolution kernel (conv_kernel.h): DilatedBlock
// (atomai/nets/blocks.py:300-318).  The dilations the reference's nets use (2, 4, 6: fcnn.py:186-200) get a
// compile-time tile geometry (EXACT); dilations 3 and 5 run generic instantiations whose prefetch registers and LDS
// image are sized by the next larger halo class.
#include "conv_kernel.h"

int amx_conv_launch_dil(ConvFwdArgs& a, int nt, bool tail, hipStream_t s) {
#define GO(N_, H_, E_, M_) return tail ? launch_conv_fwd<9, N_, H_, E_, M_, 0, true>(a, s) \
                                       : launch_conv_fwd<9, N_, H_, E_, M_, 0, false>(a, s)
#define CLASS(H_, E_)                                                                     \
    do {                                                                                  \
        if (a.th == 8) GO(4, H_, E_, 2);   /* experiment AMX_CONV_DIL_TH=8 (64-cout variant only) */ \
        if (nt == 1) GO(1, H_, E_, 4);                                                    \
        if (nt == 2) GO(2, H_, E_, 4);                                                    \
        GO(4, H_, E_, 4);                                                                 \
    } while (0)
    if (a.dil == 2) CLASS(2, true);
    if (a.dil == 4) CLASS(4, true);
    if (a.dil == 6) CLASS(6, true);
    if (a.dil == 3) CLASS(4, false);
    CLASS(6, false);                       // dilation 5
#undef CLASS
#undef GO
}
