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
// remainder-column classes: (nt, rem) = (1, 3) | (3, 1); the fused DilatedBlock sum (eval) exists for the 52-column class
#define AMX_LAT_UNIT_REM(D_)                                                                                 \
    int amx_conv_launch_lat##D_##_rem(ConvFwdArgs& a, int nt, int rem, bool tail, bool dsum, hipStream_t s) { \
        if (nt == 3 && rem == 1 && dsum)                                                                     \
            return tail ? launch_conv_fwd<9, 3, 1, true, 2, 2, true, D_, 1>(a, s)                            \
                        : launch_conv_fwd<9, 3, 1, true, 2, 2, false, D_, 1>(a, s);                          \
        if (nt == 1 && rem == 3 && dsum)                                                                     \
            return tail ? launch_conv_fwd<9, 1, 1, true, 2, 2, true, D_, 3>(a, s)                            \
                        : launch_conv_fwd<9, 1, 1, true, 2, 2, false, D_, 3>(a, s);                          \
        if (dsum) AMX_BADARG(16);                                                                            \
        if (nt == 3 && rem == 1)                                                                             \
            return tail ? launch_conv_fwd<9, 3, 1, true, 2, 0, true, D_, 1>(a, s)                            \
                        : launch_conv_fwd<9, 3, 1, true, 2, 0, false, D_, 1>(a, s);                          \
        if (nt == 1 && rem == 3)                                                                             \
            return tail ? launch_conv_fwd<9, 1, 1, true, 2, 0, true, D_, 3>(a, s)                            \
                        : launch_conv_fwd<9, 1, 1, true, 2, 0, false, D_, 3>(a, s);                          \
        AMX_BADARG(16);                                                                                      \
    }
