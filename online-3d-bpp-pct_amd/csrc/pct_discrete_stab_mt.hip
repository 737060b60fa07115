// pct_discrete_stab_mt.hip -- the strict NumPy-stream (per-env MT19937, pct_set_numpy_rng) stability-check (settings 1 / 3) kernels of the discrete
// env for 32-bit keys, every leaf expansion; see pct_discrete_impl.cuh.
#include "pct_discrete_impl.cuh"

namespace pct {
hipError_t launch_discrete_u32_stab_mt(const DiscreteParams& p, int act, const void* actions, int row_len, int n_steps,
                                   const int32_t* env_ids, int n_ids, hipStream_t stream) {
  return launch_typed<uint32_t, 5, true, true>(p, act, actions, row_len, n_steps, env_ids, n_ids, stream);
}
}  // namespace pct
