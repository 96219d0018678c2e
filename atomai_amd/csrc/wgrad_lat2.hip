// wgrad_lat2.hip — lattice-mode weight-gradient instantiations for dilation 2 (see wgrad_kernel.h).
#include "wgrad_kernel.h"
AMX_WGRAD_LAT_UNIT(2)
