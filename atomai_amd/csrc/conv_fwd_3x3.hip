// conv_fwd_3x3.hip — plain 3x3 (dilation 1) instantiations of the MFMA convolution kernel (conv_kernel.h):
// ConvBlock / ResBlock convolutions and their data gradients (atomai/nets/blocks.py:61-76, 199-214).
// th = 8 or 16 rows per workgroup tile.
#include "conv_kernel.h"

int amx_conv_launch_3x3(ConvFwdArgs& a, int nt, int th, bool tail, hipStream_t s) {
#define GO(N_, M_) return tail ? launch_conv_fwd<9, N_, 1, true, M_, 0, true>(a, s) \
                               : launch_conv_fwd<9, N_, 1, true, M_, 0, false>(a, s)
    if (th == 8) {
        if (nt == 1) GO(1, 2);
        if (nt == 2) GO(2, 2);
        GO(4, 2);
    }
    if (nt == 1) GO(1, 4);
    if (nt == 2) GO(2, 4);
    GO(4, 4);
#undef GO
}

// HEAD instantiations (classification head fused into the epilogue, eval mode): the two thin single-block classes
int amx_conv_launch_3x3_head(ConvFwdArgs& a, int nt, bool tail, hipStream_t s) {
#define GO(N_, M_) return tail ? launch_conv_fwd<9, N_, 1, true, M_, 1, true>(a, s) \
                               : launch_conv_fwd<9, N_, 1, true, M_, 1, false>(a, s)
    if (nt == 1) GO(1, 4);
    GO(2, 2);
#undef GO
}

