// Round-3 experiment: where does the dispatcher put the 4096 one-wave workgroups of the transition launch?
// Every workgroup records HW_ID (XCC / SE / CU / SIMD / wave slot) and its start time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void __launch_bounds__(64) probe(uint32_t* out, unsigned long long* t0, int spin) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) {
    uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID, 32 bits
    uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // XCC_ID (gfx940+), 4 bits
    out[blockIdx.x * 2] = hw;
    out[blockIdx.x * 2 + 1] = xcc;
    t0[blockIdx.x] = __builtin_readcyclecounter();
    smem[0] = 1;
  }
  // keep the wave alive for a while so that the whole grid is resident at once
  unsigned long long s = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - s < (unsigned long long)spin) {}
}
int main() {
  const int N = 4096;
  uint32_t* d; unsigned long long* t;
  hipMalloc(&d, N * 8); hipMalloc(&t, N * 8);
  hipLaunchKernelGGL(probe, dim3(N), dim3(64), 9968, 0, d, t, 200000);
  hipDeviceSynchronize();
  static uint32_t h[N * 2]; static unsigned long long ht[N];
  hipMemcpy(h, d, N * 8, hipMemcpyDeviceToHost); hipMemcpy(ht, t, N * 8, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull; for (int i = 0; i < N; i++) if (ht[i] < tmin) tmin = ht[i];
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  printf("block xcc se sh cu simd wave start_cycles\n");
  for (int i = 0; i < N; i += (i < 96 ? 1 : 97)) {
    uint32_t w = h[2 * i];
    printf("%5d %3u %2u %2u %3u %3u %3u %8llu\n", i, h[2 * i + 1] & 15, (w >> 13) & 7, (w >> 12) & 1, (w >> 8) & 15, (w >> 4) & 3, w & 15, ht[i] - tmin);
  }
  // how many distinct (xcc,se,sh,cu,simd) and the histogram of waves per SIMD
  static int cnt[16][8][2][16][4];
  for (int i = 0; i < N; i++) { uint32_t w = h[2 * i]; cnt[h[2 * i + 1] & 15][(w >> 13) & 7][(w >> 12) & 1][(w >> 8) & 15][(w >> 4) & 3]++; }
  int hist[64] = {0}, simds = 0;
  for (int a = 0; a < 16; a++) for (int b = 0; b < 8; b++) for (int c = 0; c < 2; c++) for (int e = 0; e < 16; e++) for (int f = 0; f < 4; f++)
    if (cnt[a][b][c][e][f]) { simds++; hist[cnt[a][b][c][e][f] < 63 ? cnt[a][b][c][e][f] : 63]++; }
  printf("distinct SIMDs %d; waves-per-SIMD histogram:", simds);
  for (int k = 0; k < 64; k++) if (hist[k]) printf(" %d:%d", k, hist[k]);
  printf("\n");
  // do consecutive blocks share a SIMD / CU?
  int same_simd = 0, same_cu = 0;
  for (int i = 1; i < N; i++) {
    uint32_t a = h[2 * i], b = h[2 * i - 2];
    bool cu = (h[2 * i + 1] == h[2 * i - 1]) && ((a >> 8) & 0xFF) == ((b >> 8) & 0xFF);
    same_cu += cu; same_simd += cu && (((a >> 4) & 3) == ((b >> 4) & 3));
  }
  printf("consecutive blocks on the same CU: %d of %d, on the same SIMD: %d\n", same_cu, N - 1, same_simd);
  for (int stride = 128; stride <= 2048; stride *= 2) {
    int same = 0;
    for (int i = 0; i + stride < N; i++) {
      uint32_t a = h[2 * i], b = h[2 * (i + stride)];
      same += (h[2 * i + 1] == h[2 * (i + stride) + 1]) && (((a >> 4) & 0xFFF) == ((b >> 4) & 0xFFF));
    }
    printf("blocks b and b + %d on the same SIMD: %d of %d\n", stride, same, N - stride);
  }
  return 0;
}
