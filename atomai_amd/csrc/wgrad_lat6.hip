// wgrad_lat6.hip — lattice-mode weight-gradient instantiations for dilation 6 (see wgrad_kernel.h).
#include "wgrad_kernel.h"
AMX_WGRAD_LAT_UNIT(6)
