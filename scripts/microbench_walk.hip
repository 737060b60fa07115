// GPU microbenchmark: what one step of the candidate set's matching walk costs (pct_set.cuh pyset_match_v) and why.
// A wave runs a dependent chain of LDS operations on pseudo-random slots of a 2048-word table, with 1 / 2 / 4 waves per
// SIMD resident (one-wave workgroups whose dynamic LDS size sets the occupancy, as in the transition kernel), in variants:
//   0  ds_read_b32 chain (address of the next read depends on the value read)
//   1  ds_min_rtn_u32 chain (returning LDS atomic, next address depends on the value returned)
//   2  atomic + the probe-sequence arithmetic of walk_advance (perturb shift, 5 i + 1, linear-probe select)
//   3  variant 2 + the wave-uniform loop control of the real walk (ballot of `walking`, per-lane stop)
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench_walk.hip -o scripts/bin/microbench_walk && scripts/bin/microbench_walk
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int VAR>
__global__ void __launch_bounds__(64) walk(uint64_t* out, int iters) {
  extern __shared__ uint32_t tab[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) tab[i] = 0x80000000u | (uint32_t)(i * 2654435761u >> 8);
  __syncthreads();
  uint32_t i = (uint32_t)(lane * 37 + blockIdx.x * 11) & 2047u;
  int j = 0;
  uint64_t perturb = 0x9E3779B97F4A7C15ull * (uint64_t)(lane + 1 + blockIdx.x * 64);
  const uint32_t mask = 2047u, mytag = 0x80000000u | 0x7FFFFF00u | (uint32_t)lane;
  bool walking = true;
  int left = iters + (lane & 7);  // lanes stop at slightly different times, as in a real walk
  const uint64_t c0 = __builtin_readcyclecounter();
  if (VAR == 0) {
    uint32_t a = i;
    for (int s = 0; s < iters; s++) a = (tab[a & mask] >> 3) + a;
    i = a;
  } else if (VAR == 1) {
    uint32_t a = i;
    for (int s = 0; s < iters; s++) a = (atomicMin(&tab[a & mask], 0xFFFFFFFFu) >> 3) + a;
    i = a;
  } else {
    while (VAR == 3 ? (__ballot(walking) != 0) : (left > 0)) {
      const uint32_t cur = (i + (uint32_t)j) & mask;
      const uint32_t old = atomicMin(&tab[cur], walking ? mytag : 0xFFFFFFFFu);
      const bool won = walking & (old > mytag);
      const bool lin = j < ((i + 9u <= mask) ? 9 : 0);
      const uint64_t pn = perturb >> 5;
      const uint32_t in = (i * 5u + 1u + (uint32_t)pn) & mask;
      const int nj = lin ? j + 1 : 0;
      const uint32_t ni = lin ? i : in;
      const uint64_t np = lin ? perturb : pn;
      i = walking ? ni : i;
      j = walking ? nj : j;
      perturb = walking ? (np | ((uint64_t)old << 40)) : perturb;  // keeps the chain dependent on the returned value
      left--;
      if (VAR == 3) walking = walking & !won & (left > 0);
    }
  }
  const uint64_t c1 = __builtin_readcyclecounter();
  if (lane == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = i + j + (uint32_t)perturb; }
}

int main() {
  uint64_t* d;
  const int grid = 4096, iters = 2000;
  hipMalloc(&d, grid * 2 * 8);
  std::vector<uint64_t> h(grid * 2);
  void (*k[4])(uint64_t*, int) = {walk<0>, walk<1>, walk<2>, walk<3>};
  const char* name[4] = {"ds_read_b32 chain", "ds_min_rtn_u32 chain", "atomic + probe arithmetic", "atomic + arithmetic + ballot loop"};
  for (int var = 0; var < 4; var++)
    for (int lds : {40 * 1024, 20 * 1024, 10 * 1024}) {  // 4 / 8 / 16 one-wave workgroups per CU = 1 / 2 / 4 waves per SIMD
      hipFuncSetAttribute(reinterpret_cast<const void*>(k[var]), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      const int g = 256 * (160 * 1024 / lds);  // exactly one resident round
      hipLaunchKernelGGL(k[var], dim3(g), dim3(64), lds, 0, d, iters);
      hipLaunchKernelGGL(k[var], dim3(g), dim3(64), lds, 0, d, iters);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d, g * 16, hipMemcpyDeviceToHost);
      double s = 0, mx = 0;
      for (int b = 0; b < g; b++) { s += (double)h[b * 2]; if ((double)h[b * 2] > mx) mx = (double)h[b * 2]; }
      printf("%-36s %2d waves/SIMD: %7.1f cycles per step (mean over %d waves), slowest wave %7.1f\n", name[var],
             160 * 1024 / lds / 4, s / g / iters, g, mx / iters);
    }
  return 0;
}
