// GPU microbenchmark: (1) what s_memtime / s_memrealtime tick at, (2) how many single-wave
// workgroups are resident per CU for a given dynamic-LDS size (census by HW_ID + time stamps).
//   hipcc --offload-arch=gfx950 -O3 microbench_census.hip -o microbench_census && ./microbench_census
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

__global__ void clock_probe(uint64_t* out, int iters) {
  uint32_t a = threadIdx.x, b = 3;
  uint64_t c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < iters; i++) {  // dependent chain of 8 VALU adds
    a += b; a ^= b; a += b; a ^= b; a += b; a ^= b; a += b; a ^= b;
    asm volatile("" : "+v"(a));
  }
  uint64_t c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = a; }
}

__global__ void census(uint64_t* out, uint64_t spin_ticks) {
  extern __shared__ unsigned char smem[];
  uint64_t t0 = wall_clock64();
  if (threadIdx.x == 0) smem[0] = 1;
  uint32_t hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
  uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(2);
  uint64_t t1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = t0; out[blockIdx.x * 4 + 1] = t1;
    out[blockIdx.x * 4 + 2] = hwid; out[blockIdx.x * 4 + 3] = xcc;
  }
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 8192 * 4 * 8);
  std::vector<uint64_t> h(8192 * 4);
  for (int iters : {100000, 400000}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    clock_probe<<<1, 64>>>(d, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    clock_probe<<<1, 64>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d, 24, hipMemcpyDeviceToHost);
    printf("clock_probe iters=%d: s_memtime ticks %llu, s_memrealtime ticks %llu, event %.3f ms -> memtime %.1f MHz, "
           "realtime %.1f MHz, %.2f memtime ticks per dependent VALU op\n", iters, (unsigned long long)h[0],
           (unsigned long long)h[1], ms, h[0] / (ms * 1e3), h[1] / (ms * 1e3), (double)h[0] / (8.0 * iters));
  }
  const int grid = 8192;
  for (int lds : {0, 2560, 5120, 6144, 8192, 10224, 10240, 10752, 12288, 16384, 20480, 27800, 32768, 65536}) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(census), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    census<<<grid, 64, lds>>>(d, 2000);  // 20 us at 100 MHz
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d, grid * 32, hipMemcpyDeviceToHost);
    // per CU (xcc, se, sh, cu bits of HW_ID above the wave/simd/pipe fields): maximum overlap
    std::map<uint64_t, std::vector<std::pair<uint64_t, int>>> ev;
    for (int b = 0; b < grid; b++) {
      uint64_t key = (h[b * 4 + 3] << 32) | (h[b * 4 + 2] & 0xFFFFFF00u & ~0x3F0000u);  // drop wave/simd/pipe and queue/state bits
      key = (h[b * 4 + 3] << 32) | ((h[b * 4 + 2] >> 8) & 0xFF);                        // cu_id[11:8] sh_id[12] se_id[15:13]
      ev[key].push_back({h[b * 4 + 0], +1});
      ev[key].push_back({h[b * 4 + 1], -1});
    }
    int mn = 1 << 30, mx = 0;
    for (auto& kv : ev) {
      std::sort(kv.second.begin(), kv.second.end());
      int cur = 0, best = 0;
      for (auto& e : kv.second) { cur += e.second; best = std::max(best, cur); }
      mn = std::min(mn, best); mx = std::max(mx, best);
    }
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, census, 64, lds);
    printf("census lds=%6d B: %zu distinct CUs, resident single-wave workgroups per CU min %d max %d (occupancy API %d), "
           "kernel %.1f us (%.1f rounds of 20 us)\n", lds, ev.size(), mn, mx, occ, ms * 1e3, ms * 1e3 / 20.0);
  }
  return 0;
}
