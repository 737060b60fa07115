// GPU microbenchmark: issue cost of 32-bit/64-bit integer multiplies vs adds on gfx950, and of one
// CPython tuple hash of six small ints.  hipcc --offload-arch=gfx950 -O3 microbench_mul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../online-3d-bpp-pct_amd/csrc/pct_set.cuh"

template <int MODE>
__global__ void k(uint64_t* out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 77;
  uint64_t A = a, B = b;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) { a += b; b += c; c += d; d += a; a ^= b; b ^= c; c ^= d; d ^= a; }
    if (MODE == 1) { a *= b; b *= c; c *= d; d *= a; a *= c; b *= d; c *= a; d *= b; }
    if (MODE == 2) { A = A * 0x9E3779B185EBCA87ULL + B; B = B * 0xC2B2AE3D27D4EB4FULL + A; A = A * 0x9E3779B185EBCA87ULL + B; B = B * 0xC2B2AE3D27D4EB4FULL + A;}
    if (MODE == 3) {
      uint64_t acc = pct::tuplehash_begin();
      acc = pct::tuplehash_lane(acc, a & 15); acc = pct::tuplehash_lane(acc, b & 15); acc = pct::tuplehash_lane(acc, c & 15);
      acc = pct::tuplehash_lane(acc, d & 15); acc = pct::tuplehash_lane(acc, (a >> 4) & 15); acc = pct::tuplehash_lane(acc, (b >> 4) & 15);
      acc = pct::tuplehash_end6(acc);
      a += (uint32_t)acc; b += (uint32_t)(acc >> 32); c ^= a; d ^= b;
    }
    if (MODE == 4) { a = __umul24(a, b) + c; b = __umul24(b, c) + d; c = __umul24(c, d) + a; d = __umul24(d, a) + b;
                     a = __umul24(a, c) + c; b = __umul24(b, d) + d; c = __umul24(c, a) + a; d = __umul24(d, b) + b; }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + A + B;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int ops_per_iter) {
  uint64_t* out;
  int blocks = 256 * 4 * waves_per_simd;
  hipMalloc(&out, (size_t)blocks * 64 * 8);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 64>>>(out, 100, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 64>>>(out, iters, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: waves_per_simd waves x iters x ops
  double ns_per_op = (double)ms * 1e6 / ((double)waves_per_simd * iters * ops_per_iter);
  printf("%-28s waves/simd=%d  %.3f ms  %.2f ns per wave-op per SIMD\n", name, waves_per_simd, ms, ns_per_op);
  hipFree(out);
}

int main() {
  for (int w : {1, 4}) {
    run<0>("add/xor x8", w, 8);
    run<1>("mul_lo_u32 x8", w, 8);
    run<2>("mul64+add64 x4", w, 4);
    run<3>("tuplehash6 x1", w, 1);
    run<4>("mul_u24+add x8", w, 8);
  }
  return 0;
}
