// pct_discrete.hip -- instantiates the setting-2 discrete-env kernels for 32-bit keys (bins <= 31 per
// axis: six 5-bit coordinates) and hosts the launch dispatcher; the kernels themselves are in
// pct_discrete_impl.cuh, the other instantiations in pct_discrete_stab.hip, pct_discrete_u64.hip, pct_discrete_u64_stab.hip and
// (strict NumPy-stream mode) pct_discrete_{,stab_,u64_,u64_stab_}mt.hip.
#include "pct_discrete_impl.cuh"

namespace pct {

size_t discrete_lds_bytes(const DiscreteParams& p) { return discrete_lds_bytes_impl(p); }

// the other translation units of the discrete env (keys x plain / stability kernels)
hipError_t launch_discrete_u64(const DiscreteParams& p, int act, const void* actions, int row_len, int n_steps,
                               const int32_t* env_ids, int n_ids, hipStream_t stream);
hipError_t launch_discrete_u64_stab(const DiscreteParams& p, int act, const void* actions, int row_len, int n_steps,
                                    const int32_t* env_ids, int n_ids, hipStream_t stream);
hipError_t launch_discrete_u32_stab(const DiscreteParams& p, int act, const void* actions, int row_len, int n_steps,
                                    const int32_t* env_ids, int n_ids, hipStream_t stream);

// Stand-in policy kernel: one wave per env reads the leaf-mask column (col 8 of rows
// I..I+L-1, tools.py:103) of the observation, k = number of valid leaves, picks
// pct_mix32(g, t) % k and copies that row out (train_tools.py:66 gather).
__global__ void __launch_bounds__(64) pct_policy_hash_rows_kernel(DiscreteParams p, float* __restrict__ rows_out) {
  const int lane = threadIdx.x;
  const int e = blockIdx.x;
  const float* obs = p.obs + (size_t)e * p.row_len;
  int k = 0;
  for (int base = 0; base < p.L; base += 64) {
    int j = base + lane;
    bool v = j < p.L && obs[(p.I + j) * 9 + 8] != 0.f;
    k += __popcll(__ballot(v));
  }
  uint32_t t = (uint32_t)p.scalars[(size_t)e * PCT_SCALARS + 6];
  int li = k > 0 ? (int)(pct_mix32((uint32_t)(p.env_id_base + e), t) % (uint32_t)k) : 0;
  if (lane < 9) rows_out[(size_t)e * 9 + lane] = obs[(p.I + li) * 9 + lane];
}


// ... and their strict NumPy-stream twins (pct_discrete_*_mt.hip)
#define PCT_DECL_MT(name)                                                                                           \
  hipError_t name(const DiscreteParams& p, int act, const void* actions, int row_len, int n_steps, const int32_t* env_ids, \
                  int n_ids, hipStream_t stream)
PCT_DECL_MT(launch_discrete_u32_mt);
PCT_DECL_MT(launch_discrete_u32_stab_mt);
PCT_DECL_MT(launch_discrete_u64_mt);
PCT_DECL_MT(launch_discrete_u64_stab_mt);
#undef PCT_DECL_MT

hipError_t launch_policy_hash_rows(const DiscreteParams& p, float* rows_out, hipStream_t stream) {
  hipLaunchKernelGGL(pct_policy_hash_rows_kernel, dim3(p.N), dim3(64), 0, stream, p, rows_out);
  return hipGetLastError();
}

hipError_t launch_discrete(const DiscreteParams& p, int act, const void* actions, int row_len, int n_steps,
                           const int32_t* env_ids, int n_ids, hipStream_t stream) {
  const bool stab = p.setting != 2;
  if (p.rng_numpy) {
    if (p.key_bytes == 4)
      return stab ? launch_discrete_u32_stab_mt(p, act, actions, row_len, n_steps, env_ids, n_ids, stream)
                  : launch_discrete_u32_mt(p, act, actions, row_len, n_steps, env_ids, n_ids, stream);
    return stab ? launch_discrete_u64_stab_mt(p, act, actions, row_len, n_steps, env_ids, n_ids, stream)
                : launch_discrete_u64_mt(p, act, actions, row_len, n_steps, env_ids, n_ids, stream);
  }
  if (p.key_bytes == 4)
    return stab ? launch_discrete_u32_stab(p, act, actions, row_len, n_steps, env_ids, n_ids, stream)
                : launch_typed<uint32_t, 5, false>(p, act, actions, row_len, n_steps, env_ids, n_ids, stream);
  return stab ? launch_discrete_u64_stab(p, act, actions, row_len, n_steps, env_ids, n_ids, stream)
              : launch_discrete_u64(p, act, actions, row_len, n_steps, env_ids, n_ids, stream);
}

}  // namespace pct
