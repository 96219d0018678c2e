// conv_fwd_lat4.hip — lattice-mode instantiations for dilation 4 (see conv_lat_inst.h).
#include "conv_lat_inst.h"
AMX_LAT_UNIT(4)
AMX_LAT_UNIT_DSUM(4)
