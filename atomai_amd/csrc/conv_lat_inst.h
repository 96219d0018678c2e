// conv_lat_inst.h — body of the lattice-mode instantiation units conv_fwd_lat{2,4,6}.hip (one per dilation so that
// hipcc compiles them in parallel): dilated 3x3 convolutions of DilatedBlock (atomai/nets/blocks.py:300-318) and their
// data gradients, run as d*d plain 3x3 convolutions on the residue-class sub-images (conv_kernel.h, LAT).
#include "conv_kernel.h"

#define AMX_LAT_UNIT(D_)                                                                                     \
    int amx_conv_launch_lat##D_(ConvFwdArgs& a, int nt, int th, bool tail, hipStream_t s) {                  \
        if (th == 8) {                                                                                       \
            if (nt == 1) AMX_LAT_GO(1, 2, D_);                                                               \
            if (nt == 2) AMX_LAT_GO(2, 2, D_);                                                               \
            AMX_LAT_GO(4, 2, D_);                                                                            \
        }                                                                                                    \
        if (nt == 1) AMX_LAT_GO(1, 4, D_);                                                                   \
        if (nt == 2) AMX_LAT_GO(2, 4, D_);                                                                   \
        AMX_LAT_GO(4, 4, D_);                                                                                \
    }
#define AMX_LAT_UNIT_DSUM(D_)                                                                                \
    int amx_conv_launch_lat##D_##_dsum(ConvFwdArgs& a, bool tail, hipStream_t s) {                           \
        return tail ? launch_conv_fwd<9, 2, 1, true, 2, 2, true, D_>(a, s)                                   \
                    : launch_conv_fwd<9, 2, 1, true, 2, 2, false, D_>(a, s);                                 \
    }
#define AMX_LAT_GO(N_, M_, D_) return tail ? launch_conv_fwd<9, N_, 1, true, M_, 0, true, D_>(a, s)      \
                                           : launch_conv_fwd<9, N_, 1, true, M_, 0, false, D_>(a, s)
