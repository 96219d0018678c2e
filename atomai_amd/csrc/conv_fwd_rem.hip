// conv_fwd_rem.hip — plain 3x3 instantiations of the remainder-column classes of conv_kernel.h (REM): layers whose width
// is not a multiple of 16 (dilnet's 25 / 50 filters, atomai/nets/fcnn.py:186-226) in ONE cout block of exactly the stored
// channels: 16 + 3 x 4 columns (28) and 3 x 16 + 4 columns (52), 8-row tiles; and the dispatcher of the lattice units.
#include "conv_kernel.h"

static std::atomic<long> rem_launches{0};
extern "C" long amx_conv2d_rem_launches(void) { return rem_launches.load(std::memory_order_relaxed); }

int amx_conv_launch_3x3_rem(ConvFwdArgs& a, int nt, int rem, bool tail, hipStream_t s) {
    rem_launches.fetch_add(1, std::memory_order_relaxed);
    if (nt == 1 && rem == 3) return tail ? launch_conv_fwd<9, 1, 1, true, 2, 0, true, 0, 3>(a, s)
                                         : launch_conv_fwd<9, 1, 1, true, 2, 0, false, 0, 3>(a, s);
    if (nt == 3 && rem == 1) return tail ? launch_conv_fwd<9, 3, 1, true, 2, 0, true, 0, 1>(a, s)
                                         : launch_conv_fwd<9, 3, 1, true, 2, 0, false, 0, 1>(a, s);
    AMX_BADARG(16);
}

int amx_conv_launch_lat_rem(ConvFwdArgs& a, int dil, int nt, int rem, bool tail, bool dsum, hipStream_t s) {
    rem_launches.fetch_add(1, std::memory_order_relaxed);
    if (dil == 2) return amx_conv_launch_lat2_rem(a, nt, rem, tail, dsum, s);
    if (dil == 4) return amx_conv_launch_lat4_rem(a, nt, rem, tail, dsum, s);
    return amx_conv_launch_lat6_rem(a, nt, rem, tail, dsum, s);
}
