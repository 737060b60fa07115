// pct_discrete_u64.hip -- the discrete-env kernels instantiated for 64-bit keys (six 10-bit
// coordinates, bins up to 1023 per axis), setting 2; see pct_discrete_impl.cuh.
#include "pct_discrete_impl.cuh"

namespace pct {
hipError_t launch_discrete_u64(const DiscreteParams& p, int act, const void* actions, int row_len, int n_steps,
                               const int32_t* env_ids, int n_ids, hipStream_t stream) {
  return launch_typed<uint64_t, 10, false>(p, act, actions, row_len, n_steps, env_ids, n_ids, stream);
}
}  // namespace pct
