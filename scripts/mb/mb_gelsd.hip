// GPU microbenchmark + device check of the strict least-squares split (csrc/pct_gelsd.cuh): a wave solves `per_wave` systems side
// by side, one per group of G lanes, on slots laid out as pct_stab.cuh's stab_gelsd_slots lays them out; the fractions come back
// for a bit-for-bit comparison with the oracle's restatement (scripts/mb_gelsd.py), the shader-clock cycles of the solve per wave.
// variant 0: one lane per system; 1: G lanes per system (what the kernels run).  (The frozen round-4 / register-resident copies this
// file used to compare against are in git history: scripts/mb/pct_gelsd_r04.cuh, pct_gelsd_fixed.cuh at 1446cf8.)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC scripts/mb/mb_gelsd.hip -o scripts/mb/libmb_gelsd.so
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
// phase cycles of the r05 routine, summed over the waves of a run (lane 0 of each): 0 build, 1 dgeqr2 stage, 2 dgebd2 stage, 3 scale +
// dbdsqr, 4 sort / cut / dgemm / dorml2, 5 dnrm2 (inside 1 and 2)
__device__ unsigned long long g_prof[8];
#define PCT_GPROF_T0(var) const unsigned long long var = __builtin_readcyclecounter();
#define PCT_GPROF_ADD(slot, var) if (threadIdx.x == 0) atomicAdd(&g_prof[slot], __builtin_readcyclecounter() - var);
#include "../../online-3d-bpp-pct_amd/csrc/pct_gelsd.cuh"

struct Dot2 {
  bool plain;
  __device__ __forceinline__ double operator()(double x0, double x1, double y0, double y1) const { return plain ? x0 * y0 + x1 * y1 : fma(x1, y1, x0 * y0); }
};
__device__ __host__ inline int slot_doubles(int n) {
  const int M = n * (n - 1) / 2 + 1;
  return 4 + 3 * n + M * n + M + n * n + 8 * n;
}

template <int VAR>
__global__ void __launch_bounds__(64) mb_kernel(const double* in, double* xout, int* illout, unsigned long long* cyc, int nsys, int per_wave, int G,
                                                int n, int avx2) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int sd = slot_doubles(n);
  const int slot = lane / G, gl = lane & (G - 1);
  const int sys = blockIdx.x * per_wave + slot;
  const bool on = slot < per_wave && sys < nsys;
  double* w = lds + (size_t)slot * sd;
  if (on)
    for (int i = gl; i < 36; i += G) {
      if (i < 4) w[i] = in[(size_t)sys * 36 + i];
      else if (i - 4 < 2 * n) w[i] = in[(size_t)sys * 36 + i];
    }
  __syncthreads();
  bool ill = false;
  const unsigned long long c0 = __builtin_readcyclecounter();
  if (on) {
    const int k = (int)w[0];
    if (VAR == 0) {  // one lane per system (G = 1 whatever the launch says): the serial flavour of the same routine
      if (gl == 0) { const pct::gelsd::Grp one = {0, 1}; pct::gelsd::split_t(one, w + 4 + 3 * n, k, w + 4, w[1], w[2], Dot2{avx2 != 0}, w + 4 + 2 * n, ill, avx2 != 0); }
    } else {
      const pct::gelsd::Grp g = {gl, G};
      pct::gelsd::split_t(g, w + 4 + 3 * n, k, w + 4, w[1], w[2], Dot2{avx2 != 0}, w + 4 + 2 * n, ill, avx2 != 0);
    }
  }
  __syncthreads();
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x] = c1 - c0;
  if (on) {
    const int k = (int)w[0];
    for (int i = gl; i < 16; i += G) xout[(size_t)sys * 16 + i] = i < k ? w[4 + 2 * n + i] : 0.0;
    if (gl == 0) illout[sys] = ill ? 1 : 0;
  }
}

// in: [nsys][36] = k, s0, s1, 0, centres (2 per supporter); all systems must have k <= n.  Returns 0, or a HIP error code.
extern "C" int mb_gelsd_run(const double* in, double* xout, int* illout, unsigned long long* cyc, int nsys, int per_wave, int G, int n, int avx2, int variant) {
  double *din, *dx;
  int* dill;
  unsigned long long* dc;
  const int nblk = (nsys + per_wave - 1) / per_wave;
  if (hipMalloc(&din, (size_t)nsys * 36 * 8) || hipMalloc(&dx, (size_t)nsys * 16 * 8) || hipMalloc(&dill, (size_t)nsys * 4) || hipMalloc(&dc, (size_t)nblk * 8)) return 1;
  hipMemcpy(din, in, (size_t)nsys * 36 * 8, hipMemcpyHostToDevice);
  const size_t lds = (size_t)per_wave * slot_doubles(n) * 8;
  hipError_t e;
  if (variant == 0) {
    hipFuncSetAttribute((const void*)mb_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mb_kernel<0>, dim3(nblk), dim3(64), lds, 0, din, dx, dill, dc, nsys, per_wave, G, n, avx2);
  } else {
    hipFuncSetAttribute((const void*)mb_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mb_kernel<1>, dim3(nblk), dim3(64), lds, 0, din, dx, dill, dc, nsys, per_wave, G, n, avx2);
  }
  e = hipDeviceSynchronize();
  if (e != hipSuccess) { fprintf(stderr, "mb_gelsd: %s\n", hipGetErrorString(e)); return (int)e; }
  hipMemcpy(xout, dx, (size_t)nsys * 16 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(illout, dill, (size_t)nsys * 4, hipMemcpyDeviceToHost);
  hipMemcpy(cyc, dc, (size_t)nblk * 8, hipMemcpyDeviceToHost);
  {
    unsigned long long pr[8], zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_prof), sizeof pr);
    hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof zero);
    if (variant == 1 && getenv("MB_PROF"))
      fprintf(stderr, "   phases per wave: build %llu  qr %llu  bidiag %llu  bdsqr %llu  final %llu  (dnrm2 %llu)\n", pr[0] / nblk, pr[1] / nblk, pr[2] / nblk,
              pr[3] / nblk, pr[4] / nblk, pr[5] / nblk);
  }
  hipFree(din); hipFree(dx); hipFree(dill); hipFree(dc);
  return 0;
}
