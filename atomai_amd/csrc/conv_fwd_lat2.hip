// conv_fwd_lat2.hip — lattice-mode instantiations for dilation 2 (see conv_lat_inst.h).
#include "conv_lat_inst.h"
AMX_LAT_UNIT(2)
AMX_LAT_UNIT_DSUM(2)
