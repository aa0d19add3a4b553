/* GSL-API shim (oracle test infrastructure): see gsl_shim_core.h */
#include "gsl_shim_core.h"
