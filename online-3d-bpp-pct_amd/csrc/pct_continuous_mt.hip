// Strict NumPy-stream variants of the continuous-env kernels (include/pct_env.h pct_set_numpy_rng): the same source,
// compiled with the MT19937 paths switched on.
#define PCT_CONT_MT 1
#include "pct_continuous.hip"
