// The two-wave candidate pipeline of the continuous env (setting 2, table in LDS: C5 -- see "the two-wave candidate pipeline" in
// pct_continuous.hip): the same source, compiled with the 128-thread pipeline kernels switched on.  In this translation unit
// `__syncthreads()` is a WAVE-level LDS fence: the source uses it as the hand-over between the lanes of the ONE wave that owns an env
// (a 64-thread workgroup's barrier is exactly that), and here two waves of a workgroup run different code and are never at the
// same barrier.  The only workgroup barrier of these kernels is the explicit s_barrier at their start.
#include <hip/hip_runtime.h>
__device__ __forceinline__ void pct_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
#define __syncthreads() pct_wave_lds_sync()
#define PCT_CONT_PIPE 1
#include "pct_continuous.hip"
