// pct_pow.cuh -- `tri_base_len ** 2` as the reference computes it.
//
// Reference: pct_envs/PctContinuous0/space.py:108,206 (PctDiscrete0/space.py:112,210): `tri_base_line /= tri_base_len ** 2`
// with tri_base_len an np.float64 -- NumPy's scalar power, which is libm's pow(len, 2.0).  glibc's pow (>= 2.28: the
// algorithm of ARM's optimized-routines, sysdeps/ieee754/dbl-64/e_pow.c) is NOT correctly rounded: it returns
// exp(y * log(x)) with log(x) held as a double-double of ~2^-68 relative error, 0.52 ULP overall, so for 0.08 % of the
// lengths the continuous env produces its last bit differs from the correctly rounded square len * len.  This file
// restates the path pow() takes for a positive normal x and a y with 2^-65 <= |y * log x| < 2^9 (every length of the
// lever rule: 1e-6 .. 2e2 squared), operation for operation AS COMPILED into the FMA variant the dynamic linker selects
// on every x86-64 host with FMA3 + AVX2 (libm.so.6 of glibc 2.35, `__pow_fma`; sysdeps/x86_64/fpu/multiarch/e_pow.c):
// gcc contracts t1 = kd*Ln2hi + logc, lo1 = kd*Ln2lo + logctail, the polynomial's Horner steps, z = InvLn2N*x + Shift,
// r = x + kd*NegLn2hiN + kd*NegLn2loN and scale + scale*tmp into fused multiply-adds, and the order below is the
// binary's.  The tables (pct_pow_tables.h) are that library's data.  Pinned against the live libm in
// tests/test_stab_host.py (host build of this source) and on the device in tests/test_zz_gpu_gelsd.py.
#ifndef PCT_POW_CUH
#define PCT_POW_CUH
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "pct_pow_tables.h"

namespace pct {

#if defined(__HIPCC__)
#define PCT_POW_FN __device__ __forceinline__
#define PCT_POW_TAB static __device__ const
#else
#define PCT_POW_FN static inline
#define PCT_POW_TAB static const
#endif

PCT_POW_TAB uint64_t pow_logtab[128 * 3] = PCT_POW_LOGTAB_INIT;
PCT_POW_TAB uint64_t pow_exptab[128 * 2] = PCT_POW_EXPTAB_INIT;

PCT_POW_FN double pow_d(uint64_t u) {
#if defined(__HIPCC__)
  return __longlong_as_double((long long)u);
#else
  double d; memcpy(&d, &u, 8); return d;
#endif
}
PCT_POW_FN uint64_t pow_u(double d) {
#if defined(__HIPCC__)
  return (uint64_t)__double_as_longlong(d);
#else
  uint64_t u; memcpy(&u, &d, 8); return u;
#endif
}

// pow(x, y) for a positive, normal, finite x and a finite y of ordinary size; anything the main path of glibc's pow
// would hand to a special case returns NaN here (no caller produces one: asserted by the tests over the lever rule's domain).
PCT_POW_FN double pow_glibc_fma(double x, double y) {
  const uint64_t ix = pow_u(x), iy = pow_u(y);
  const uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52);
  if (topx - 1u > 0x7fdu || (topy & 0x7ffu) - 0x3beu > 0x7fu) return pow_d(0x7ff8000000000000ull);
  // log_inline: x = 2^k z, z in [OFF, 2 OFF), c = the centre of z's sub-interval; log x = k ln2 + log c + log1p(z/c - 1)
  const uint64_t tmp = ix - 0x3fe6955500000000ull;
  const int i = (int)((tmp >> 45) & 127u);
  const double kd = (double)(int)((int64_t)tmp >> 52);
  const double z = pow_d(ix - (tmp & 0xfff0000000000000ull));
  const double invc = pow_d(pow_logtab[3 * i]), logc = pow_d(pow_logtab[3 * i + 1]), logctail = pow_d(pow_logtab[3 * i + 2]);
  const double A0 = pow_d(PCT_POW_A0), A1 = pow_d(PCT_POW_A1), A2 = pow_d(PCT_POW_A2), A3 = pow_d(PCT_POW_A3), A4 = pow_d(PCT_POW_A4),
               A5 = pow_d(PCT_POW_A5), A6 = pow_d(PCT_POW_A6);
  const double t1 = fma(kd, pow_d(PCT_POW_LN2HI), logc);
  const double r = fma(z, invc, -1.0);
  const double ar = r * A0;
  const double lo1 = fma(kd, pow_d(PCT_POW_LN2LO), logctail);
  const double p12 = fma(r, A2, A1);
  const double p34 = fma(r, A4, A3);
  const double t2 = r + t1;
  const double ar2 = r * ar;
  const double d12 = t1 - t2;
  const double ar3 = r * ar2;
  const double lo3 = fma(ar, r, -ar2);
  const double lo2 = d12 + r;
  const double p56 = fma(r, A6, A5);
  const double hi = t2 + ar2;
  const double dh = t2 - hi;
  const double q1 = fma(p56, ar2, p34);
  const double lo4 = dh + ar2;
  const double q = fma(ar2, q1, p12);
  double lo = lo1 + lo2;
  lo = lo + lo3;
  lo = lo + lo4;
  lo = fma(ar3, q, lo);
  const double lhi = hi + lo;
  const double llo = (hi - lhi) + lo;
  // pow: ehi + elo = y * (lhi + llo)
  const double ehi = y * lhi;
  const double elo = fma(y, llo, fma(lhi, y, -ehi));
  // exp_inline(ehi, elo, 0)
  const uint32_t abstop = (uint32_t)(pow_u(ehi) >> 52) & 0x7ffu;
  if (abstop - 0x3c9u > 0x3eu) {
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0;  // |y log x| < 2^-54: 1.0 + x rounds to 1.0
    return pow_d(0x7ff8000000000000ull);             // overflow / underflow territory: not this restatement's
  }
  const double shift = pow_d(PCT_POW_SHIFT);
  double kd2 = fma(ehi, pow_d(PCT_POW_INVLN2N), shift);
  const uint64_t ki = pow_u(kd2);
  kd2 = kd2 - shift;
  double rr = fma(kd2, pow_d(PCT_POW_NEGLN2HIN), ehi);
  rr = fma(kd2, pow_d(PCT_POW_NEGLN2LON), rr);
  rr = elo + rr;
  const int idx = 2 * (int)(ki & 127u);
  const uint64_t sbits = pow_exptab[idx + 1] + (ki << 45);
  const double c23 = fma(rr, pow_d(PCT_POW_C3), pow_d(PCT_POW_C2));
  const double tr = rr + pow_d(pow_exptab[idx]);
  const double r2 = rr * rr;
  const double c45 = fma(rr, pow_d(PCT_POW_C5), pow_d(PCT_POW_C4));
  const double s1 = fma(c23, r2, tr);
  const double r4 = r2 * r2;
  const double tm = fma(c45, r4, s1);
  if (abstop == 0) return pow_d(0x7ff8000000000000ull);
  const double scale = pow_d(sbits);
  return fma(tm, scale, scale);
}

}  // namespace pct
#endif
