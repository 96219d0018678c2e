// conv_fwd_lat4_rem.hip — lattice-mode (dilation 4) instantiations of the remainder-column classes (conv_kernel.h, REM;
// conv_fwd_rem.hip).  One unit per dilation so that hipcc compiles them in parallel.
#include "conv_lat_inst.h"
AMX_LAT_UNIT_REM(4)
