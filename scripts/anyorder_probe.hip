// anyorder_probe.hip -- does this runtime / GPU honour hipExtAnyOrderLaunch on one stream?  (hip_ext.h says "not supported
// on GFX9xx" for the module-launch flavour.)  Kernel A fills the chip for ~60 us; kernel B, one workgroup, is enqueued right
// behind it (a) plainly, (b) with hipExtAnyOrderLaunch.  Every kernel stamps wall_clock64() (100 MHz) at its first and last
// instruction: with the flag honoured B starts while A runs.  Also prints the dependent-dispatch gap (B start - A end) and
// what a pair of hipEventRecord calls around A costs.
//   hipcc --offload-arch=gfx950 -O2 scripts/anyorder_probe.hip -o scripts/bin/anyorder_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void busy(long long* stamps, int us, int slot) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (long long)us * 100) __builtin_amdgcn_s_sleep(8);
  if (blockIdx.x == 0 && threadIdx.x == 0) stamps[2 * slot] = t0;
  if (threadIdx.x == 0) atomicMax((unsigned long long*)&stamps[2 * slot + 1], (unsigned long long)wall_clock64());
}
__global__ void probe(long long* stamps, int slot) {
  if (threadIdx.x == 0) {
    stamps[2 * slot] = wall_clock64();
    stamps[2 * slot + 1] = wall_clock64();
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  long long* d;
  CK(hipMalloc(&d, 64 * sizeof(long long)));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  long long h[64];
  for (int mode = 0; mode < 3; mode++) {
    double gap = 0, over = 0;
    int n_over = 0;
    const int reps = 20;
    for (int r = 0; r < reps; r++) {
      CK(hipMemsetAsync(d, 0, 64 * sizeof(long long), s));
      CK(hipStreamSynchronize(s));
      hipLaunchKernelGGL(busy, dim3(4096), dim3(64), 0, s, d, 60, 0);
      if (mode == 0) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, s, d, 1);
      else hipExtLaunchKernelGGL(probe, dim3(1), dim3(64), 0, s, nullptr, nullptr, mode == 1 ? hipExtAnyOrderLaunch : 0, d, 1);
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, s, d, 2);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
      const double a_end = h[1], b_start = h[2], c_start = h[4], b_end = h[3];
      if (b_start < a_end) { n_over++; over += (a_end - b_start) / 100.0; }
      else gap += (b_start - a_end) / 100.0;
      if (r == 0) printf("  mode %d rep0: A %.1f us long, B starts %+.2f us after A's end, C starts %+.2f us after B's end\n", mode,
                         (h[1] - h[0]) / 100.0, (b_start - a_end) / 100.0, (c_start - b_end) / 100.0);
    }
    printf("mode %d (%s): B overlapped A in %d / %d runs (mean %.1f us early); otherwise mean gap %.2f us\n", mode,
           mode == 0 ? "hipLaunchKernelGGL" : (mode == 1 ? "hipExtLaunchKernelGGL + hipExtAnyOrderLaunch" : "hipExtLaunchKernelGGL, flags 0"),
           n_over, reps, n_over ? over / n_over : 0.0, (reps - n_over) ? gap / (reps - n_over) : 0.0);
  }
  // steady-state cost of a dependent dispatch: 200 x (busy 60 us; probe) with and without the probe, with and without event pairs
  hipEvent_t e0, e1, ea, eb;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  for (int cfg = 0; cfg < 5; cfg++) {
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 200; i++) {
      if (cfg == 3) CK(hipEventRecord(ea, s));
      if (cfg == 4) hipExtLaunchKernelGGL(busy, dim3(4096), dim3(64), 0, s, ea, eb, 0, d, 60, 0);
      else hipLaunchKernelGGL(busy, dim3(4096), dim3(64), 0, s, d, 60, 0);
      if (cfg == 3) CK(hipEventRecord(eb, s));
      if (cfg == 1) hipLaunchKernelGGL(probe, dim3(16), dim3(64), 0, s, d, 1);
      if (cfg == 2) hipExtLaunchKernelGGL(probe, dim3(16), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 1);
    }
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const char* what[] = {"busy only", "busy + dependent 16-block kernel", "busy + any-order 16-block kernel",
                          "busy bracketed by hipEventRecord pairs", "busy with hipExtLaunchKernel start/stop events"};
    float kms = 0;
    if (cfg >= 3) CK(hipEventElapsedTime(&kms, ea, eb));
    printf("steady state, %-48s: %.2f us per iteration%s", what[cfg], ms * 1000.0 / 200, cfg >= 3 ? "" : "\n");
    if (cfg >= 3) printf(" (last pair: %.2f us)\n", kms * 1000.0);
  }
  return 0;
}
