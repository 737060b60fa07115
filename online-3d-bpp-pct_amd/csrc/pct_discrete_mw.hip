// Multi-wave variant of the discrete step kernels (setting 2, EMS expansion, 32-bit keys, counter-keyed streams):
// a 256-thread workgroup per env.  For most envs three of the four waves leave at once and wave 0 runs the
// one-wave-per-env step unchanged; an env with at least p.heavy_t EMS -- one whose candidate set alone would keep a
// single wave busy long after the rest of the launch has drained -- keeps all four and builds the set with them
// (pct_discrete_mw.cuh).  Same results bit for bit; the launch gets shorter because its slowest envs do.
#include "pct_discrete_impl.cuh"

namespace pct {

template <int ACT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
pct_discrete_kernel_mw(DiscreteParams p, const void* __restrict__ actions, int row_len, int n_steps) {
  typedef uint32_t K;
  constexpr int BITS = 5;
  extern __shared__ __align__(16) unsigned char smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int e = blockIdx.x;
  const int n_ems = __builtin_amdgcn_readfirstlane(p.scalars[(size_t)e * PCT_SCALARS + 0]);
  if (n_ems < p.heavy_t) {
    if (wv) return;
    discrete_env_steps<K, BITS, ACT, false, false, 0, 0, false>(p, actions, row_len, n_steps, e, smem);
    return;
  }
  Lds<K, BITS> l = carve_lds<K, BITS>(p, smem);
  if (wv == 0) {
    discrete_env_steps<K, BITS, ACT, false, false, 0, 0, true>(p, actions, row_len, n_steps, e, smem);
    if (lane == 0) l.ctl[0] = MW_CMD_EXIT;
    __syncthreads();
    return;
  }
  // helper waves: one cooperative build per observation wave 0 produces, until it says it is done
  MwCtl c{l.ctl, 0u};
  while (true) {
    __syncthreads();
    if (c.w[0] == (uint32_t)MW_CMD_EXIT) break;
    SetState<K> st;
    mw_build_set<K, BITS>(p, l, c, wv, lane, st);
  }
}

hipError_t launch_discrete_mw(const DiscreteParams& p, int act, const void* actions, int row_len, int n_steps,
                              hipStream_t stream) {
  const size_t lds = discrete_lds_bytes_impl(p);
  void (*kern)(DiscreteParams, const void*, int, int);
  switch (act) {
    case ACT_ROWS: kern = pct_discrete_kernel_mw<ACT_ROWS>; break;
    case ACT_INDEX: kern = pct_discrete_kernel_mw<ACT_INDEX>; break;
    default: kern = pct_discrete_kernel_mw<ACT_HASH>; break;
  }
  hipLaunchKernelGGL(kern, dim3(p.N), dim3(256), lds, stream, p, actions, row_len, n_steps);
  return hipGetLastError();
}

}  // namespace pct
