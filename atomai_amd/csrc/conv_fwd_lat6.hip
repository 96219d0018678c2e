// conv_fwd_lat6.hip — lattice-mode instantiations for dilation 6 (see conv_lat_inst.h).
#include "conv_lat_inst.h"
AMX_LAT_UNIT(6)
AMX_LAT_UNIT_DSUM(6)
