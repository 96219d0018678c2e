// wgrad_lat4.hip — lattice-mode weight-gradient instantiations for dilation 4 (see wgrad_kernel.h).
#include "wgrad_kernel.h"
AMX_WGRAD_LAT_UNIT(4)
