// pct_gelsd.cuh -- np.linalg.lstsq of the stability check, as the reference's NumPy executes it (PCT_LSTSQ_GELSD).
//
// The reference splits a stack over >= 3 supporters (none of them "direct") with np.linalg.lstsq(coefficient, value, rcond=None):
// D/space.py:134-163, :236-259; C/space.py:130-159, :232-255.  That is LAPACK dgelsd inside the OpenBLAS of the NumPy wheel; the
// default solver of the kernels (a one-sided Jacobi SVD, pct_stab.cuh) returns the same minimum-norm solution up to the last
// bits, and a last bit decides an exactly degenerate test downstream in about one on-domain env-run out of 55
// (profiles/r04_lstsq_ondomain.txt).  This file is the strict alternative: the path dgelsd takes for these systems
// (M = k (k - 1) / 2 + 1 rows, k <= 16 columns, one right-hand side), operation for operation --
//   dgeqr2, dorm2r (Q^T b), dgebd2, dorm2r, dlalsd (scale, dlasdq = dbdsqr with vectors + sort, rank cut, V (.)), dorml2 --
// with the arithmetic of the BLAS kernels OpenBLAS 0.3.29 dispatches to on AVX-512 hosts (NumPy 2.2.6's bundled library in
// the build container; PCT_LSTSQ_GELSD) or on AVX2 hosts, AMD Zen included (its "Haswell" kernel set; PCT_LSTSQ_GELSD_AVX2 --
// dgemv 'N', daxpy and the dgemm kernel differ): which products are fused, how many lanes a sum is split over, and dnrm2's
// 80-bit x87 accumulation, emulated here with 64-bit integer mantissas.  Plain IEEE double operations otherwise (the library is built with
// -ffp-contract=off; fma() is v_fma_f64 / the C fma, division and sqrt are correctly rounded on both sides).
// The routines work on a workspace in LDS (no private arrays, no calls) and are written for a GROUP of G lanes of one wavefront
// per system (round 5; G = 1: one lane, which is also how tests/host compiles them for the CPU, where tests/test_stab_host.py
// checks them against the recorded NumPy vectors and the reference fixtures).  What a bit of the result depends on is the
// arithmetic of every ELEMENT -- which products are fused, in which order a sum runs -- not on who computes it:
//   * the scalar chain of the algorithm (dnrm2, dlapy2, tau, the rotations of dbdsqr, every convergence test) is computed by
//     EVERY lane of the group, redundantly: the same loads (LDS broadcasts), the same operations, the same stores of the same
//     values -- so no lane ever waits for another one's scalar, and control flow is uniform within a group;
//   * the loops over independent elements are split over the lanes, owner computes: a column of the trailing matrix (its
//     dgemv 'T' sum -- by the kernel variant its POSITION selects -- and its dger update, fused), a row of it for a reflector
//     from the right (dgemv 'N' row + dger row), a column of V^T / the right-hand side for a sweep's rotations, an element of
//     dscal / dlascl / the sort's row swaps, a row of the closing dgemm.
// Between a region of the second kind and its neighbours stands PCT_GSYNC: the lanes of a group run in lockstep (one
// wavefront, group-uniform branches) and the LDS executes a wavefront's instructions in order, so what is needed is that the
// COMPILER keeps the accesses on their side of the line -- a workgroup-scope fence (no s_barrier: groups of one wavefront that
// solve different systems diverge from each other, a barrier inside such a branch would be wrong).
#ifndef PCT_GELSD_CUH
#define PCT_GELSD_CUH

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PCT_GD __device__ __forceinline__
#define PCT_GNOUNROLL _Pragma("nounroll")
// (the wave barrier is a scheduling barrier only -- no instruction: it keeps the compiler from moving ANY access across the line, also
// the ones a fence alone leaves free)
#define PCT_GSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define PCT_GD static inline
#define PCT_GNOUNROLL
#define PCT_GSYNC() ((void)0)
#endif
// scripts/mb/mb_gelsd.hip defines these to book shader-clock cycles per phase; nothing anywhere else
#if !defined(PCT_GPROF_T0)
#define PCT_GPROF_T0(var)
#define PCT_GPROF_ADD(slot, var)
#endif
// Every routine is inlined into the kernels (no calls inside a kernel, build.py), so each CALL SITE costs the routine's whole body:
// the drivers below are written as loops over "which vector / which target" with ONE site per routine where that is possible
// (PCT_GNOUNROLL keeps the compiler from undoing it).

namespace pct {
namespace gelsd {

// the lanes that share one system: lane gl of G (G = 1: one lane does everything)
struct Grp { int gl, G; };

// ---- 80-bit extended precision (x87, round to nearest even, 64-bit mantissa), non-negative values only -----------------------
// value = m * 2^e with 2^63 <= m < 2^64, or m == 0
struct Ext { uint64_t m; int e; };

PCT_GD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIPCC__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
PCT_GD int clz64(uint64_t v) {
#if defined(__HIPCC__)
  return __clzll((long long)v);
#else
  return __builtin_clzll(v);
#endif
}
// (hi, lo) with hi's top bit set, `sticky`: non-zero bits below lo -> m rounded to nearest even (carry into the exponent)
PCT_GD Ext ext_round(uint64_t hi, uint64_t lo, bool sticky, int e) {
  const bool rnd = (lo >> 63) != 0;
  const bool rest = (lo << 1) != 0 || sticky;
  Ext r;
  r.m = hi;
  r.e = e;
  if (rnd && (rest || (hi & 1ull))) {
    r.m = hi + 1;
    if (r.m == 0) { r.m = 1ull << 63; r.e = e + 1; }
  }
  return r;
}
// fld + fmul st(0): the square of a double, rounded to 64 bits
PCT_GD Ext ext_square(double v) {
  Ext z;
  z.m = 0; z.e = 0;
  uint64_t bits;
  {
    union { double d; uint64_t u; } cv;
    cv.d = v;
    bits = cv.u;
  }
  const int ex = (int)((bits >> 52) & 0x7FF);
  uint64_t f = bits & 0xFFFFFFFFFFFFFull;
  int q;
  if (ex == 0) {
    if (f == 0) return z;
    q = -1074;
  } else {
    f |= 1ull << 52;
    q = ex - 1075;
  }
  const int s = clz64(f);
  const uint64_t g = f << s;  // v = g * 2^(q - s), 2^63 <= g
  uint64_t hi = mulhi64(g, g), lo = g * g;
  int e = 2 * (q - s) + 64;
  if (!(hi >> 63)) {  // the product lies in [2^126, 2^127)
    hi = (hi << 1) | (lo >> 63);
    lo <<= 1;
    e -= 1;
  }
  return ext_round(hi, lo, false, e);
}
// faddp of two non-negative values
PCT_GD Ext ext_add(Ext x, Ext y) {
  if (y.m == 0) return x;
  if (x.m == 0) return y;
  if (x.e < y.e || (x.e == y.e && x.m < y.m)) { const Ext t = x; x = y; y = t; }
  const int d = x.e - y.e;
  if (d > 65) return x;  // y is below a quarter of x's last place
  // X = x.m : 0, Y = (y.m : 0) >> d
  uint64_t yh, yl;
  bool sticky = false;
  if (d == 0) { yh = y.m; yl = 0; }
  else if (d < 64) { yh = y.m >> d; yl = y.m << (64 - d); }
  else if (d == 64) { yh = 0; yl = y.m; }
  else { yh = 0; yl = y.m >> 1; sticky = (y.m & 1ull) != 0; }
  uint64_t hi = x.m + yh, lo = yl;
  int e = x.e;
  if (hi < x.m) {  // carry out of bit 127: one place down
    sticky = sticky || (lo & 1ull);
    lo = (lo >> 1) | (hi << 63);
    hi = (hi >> 1) | (1ull << 63);
    e += 1;
  }
  return ext_round(hi, lo, sticky, e);
}
// fsqrt, then fstp qword: the square root rounded to 64 bits, then to a double
PCT_GD double ext_sqrt_to_double(Ext x) {
  if (x.m == 0) return 0.0;
  // N = x.m << 64 (exponent e - 64) or x.m << 63 (exponent e - 63), whichever exponent is even; root = isqrt(N), 2^63 <= root
  uint64_t nh, nl;
  int e2;
  if (((x.e - 64) & 1) == 0) { nh = x.m; nl = 0; e2 = x.e - 64; }
  else { nh = x.m >> 1; nl = x.m << 63; e2 = x.e - 63; }
  uint64_t root, rh, rl;  // root = floor(sqrt(N)); remainder N - root^2 = (rh : rl), at most 65 bits
#if defined(PCT_GELSD_BITWISE_SQRT)
  root = 0; rh = 0; rl = 0;  // (the round-4 digit-by-digit root, 64 dependent iterations: kept as the checker of the fast one, tests/host)
  for (int i = 0; i < 64; i++) {
    // rem = (rem << 2) | top two bits of N;  N <<= 2
    rh = (rh << 2) | (rl >> 62);
    rl = (rl << 2) | (nh >> 62);
    nh = (nh << 2) | (nl >> 62);
    nl <<= 2;
    // trial = (root << 2) | 1  (root holds the bits found so far)
    const uint64_t th = root >> 62, tl = (root << 2) | 1ull;
    root <<= 1;
    if (rh > th || (rh == th && rl >= tl)) {
      const uint64_t b = rl < tl ? 1ull : 0ull;
      rl -= tl;
      rh = rh - th - b;
      root |= 1ull;
    }
  }
#else
  {
    // A double-precision seed (within ~2^12 of the root), one Newton step with the exact 128-bit residual, then the exact
    // correction by at most a few units: the result is floor(sqrt(N)) whatever the seed's last bits are (the device's sqrt
    // and the host's may differ there), and the remainder comes out of the same integer arithmetic.
    const double s = sqrt((double)nh) * 4294967296.0;
    uint64_t r = s >= 18446744073709551616.0 ? ~0ull : (uint64_t)s;
    if (r < (1ull << 63)) r = 1ull << 63;
    {
      const uint64_t ph = mulhi64(r, r), pl = r * r;
      const uint64_t dl = nl - pl;
      const int64_t dh = (int64_t)(nh - ph - (nl < pl ? 1ull : 0ull));
      const double dd = (double)dh * 18446744073709551616.0 + (double)dl;
      const double q = floor(dd / (2.0 * (double)r));
      if (q >= 0.0) {
        const uint64_t step = (uint64_t)q;
        r = r + step < r ? ~0ull : r + step;
      } else {
        r -= (uint64_t)(-q);
      }
    }
    // rem = N - r^2 as a two's-complement 128-bit number; while rem < 0: r--, rem += 2 r + 1; while rem > 2 r: rem -= 2 r + 1, r++
    const uint64_t ph = mulhi64(r, r), pl = r * r;
    rl = nl - pl;
    rh = nh - ph - (nl < pl ? 1ull : 0ull);
    PCT_GNOUNROLL
    while ((int64_t)rh < 0) {
      r -= 1;
      const uint64_t th = r >> 63, tl = (r << 1) | 1ull;
      const uint64_t nl2 = rl + tl;
      rh = rh + th + (nl2 < rl ? 1ull : 0ull);
      rl = nl2;
    }
    PCT_GNOUNROLL
    while (true) {
      const uint64_t th = r >> 63, tl2 = r << 1;  // 2 r = (th : tl2)
      if (!(rh > th || (rh == th && rl > tl2))) break;
      const uint64_t tl = tl2 | 1ull;
      const uint64_t b = rl < tl ? 1ull : 0ull;
      rl -= tl;
      rh = rh - th - b;
      r += 1;
    }
    root = r;
  }
#endif
  // round to nearest: up iff N - root^2 > root (a tie cannot occur)
  int e = e2 / 2;
  uint64_t m = root;
  if (rh != 0 || rl > root) {
    m = root + 1;
    if (m == 0) { m = 1ull << 63; e += 1; }
  }
  // to double: 64 -> 53 bits, nearest even
  uint64_t f = m >> 11;
  const uint64_t low = m & 0x7FFull;
  if (low > 0x400ull || (low == 0x400ull && (f & 1ull))) f += 1;
  int ex = e + 63 + 1023;
  if (f >> 53) { f >>= 1; ex += 1; }
  if (ex <= 0) return 0.0;  // (far below anything a stability system holds)
  if (ex >= 2047) return INFINITY;
  union { double d; uint64_t u; } cv;
  cv.u = ((uint64_t)ex << 52) | (f & 0xFFFFFFFFFFFFFull);
  return cv.d;
}

// ---- BLAS, as OpenBLAS' SkylakeX kernel set computes it (kernel/x86_64: nrm2.S, dgemv_t_4.c + Haswell microkernels,
// dgemv_n_4.c + SkylakeX microkernel, ger.c -> daxpy, drot + SkylakeX microkernel) ------------------------------------------
// dnrm2: accumulators A..D over the leading blocks of eight (element i into accumulator i mod 4), the tail into A,
// ((C + A) + B) + D, fsqrt, one rounding to double.  (Every lane of a group runs it: the loads are LDS broadcasts, eight in
// flight at a time; the accumulation is the serial chain it is on the FPU.)
// The same value WITHOUT the integer emulation, for up to eight elements, when a certificate holds (round 6).  dnrm2 returns
// round53(round64(sqrt(S))) with S the x87 sum: every square and every partial sum rounded to 64 bits, so S = T (1 + e),
// |e| <= (n + 4) 2^-64 <= 12 * 2^-64 for the exact sum of squares T (all terms non-negative), and round64(sqrt(S)) lies within
// 2^-61 sqrt(T) of sqrt(T).  Here T is accumulated as a double-double (squares exact by FMA, two-sum additions: relative error
// < 2^-100), its root as s + c with s = sqrt(T_hi) and c = (T_hi - s^2 + T_lo) / 2s (the residual of a square root is exact in one
// FMA; error of s + c below 2^-100 sqrt(T)), and R = fl(s + c) is the correct rounding of that pair.  R is ALSO the x87 value
// whenever s + c keeps a distance of more than 2^-60 R from the nearest rounding boundary (the midpoint of two neighbouring
// doubles: half an ulp from R, a quarter below a power of two) -- tested with a margin of 2 % of half an ulp = 2^-59.6 R.  Otherwise
// (2 % of the calls), or when a square could leave the double range, the caller runs the emulation.  Checked against it on
// 2 * 10^7 vectors (tests/test_stab_host.py: 0 differences).  Three supporters: eight to ten dnrm2 calls of 2 - 4 elements a solve.
PCT_GD bool dnrm2_certified(int n, const double* x, int incx, double& out) {
  double th = 0.0, tl = 0.0;
  PCT_GNOUNROLL
  for (int i = 0; i < n; i++) {
    const double v = x[i * incx];
    const double a = fabs(v);
    if (a != 0.0 && !(a > 1e-100 && a < 1e100)) return false;
    const double p = v * v, e = fma(v, v, -p);
    const double s = th + p;
    const double bb = s - th;
    const double err = (th - (s - bb)) + (p - bb);
    const double lo = (tl + e) + err;
    th = s + lo;
    tl = lo - (th - s);
  }
  if (th == 0.0) { out = 0.0; return true; }
  const double s = sqrt(th);
  const double c = (fma(-s, s, th) + tl) / (2.0 * s);
  const double r = s + c;
  const double err = (s - r) + c;
  union { double d; uint64_t u; } rb, hb;
  rb.d = r;
  hb.u = (rb.u & 0x7ff0000000000000ull) - (53ull << 52);  // half an ulp of r's binade
  if (fabs(err) > hb.d * 0.98) return false;
  if ((rb.u & 0xfffffffffffffull) == 0 && -err > hb.d * 0.49) return false;  // (below a power of two the spacing halves)
  out = r;
  return true;
}
PCT_GD double dnrm2_ext(int n, const double* x, int incx);
PCT_GD double dnrm2(int n, const double* x, int incx) {
  if (n <= 0) return 0.0;
#if !defined(PCT_GELSD_NO_NRM2_SHORTCUT)
  // One element: fsqrt(round64(x^2)) = |x| (1 + d) with |d| < 2^-64, |x| a 53-bit number: the 64-bit rounding of the root and the
  // store to double both return |x| (checked against the long chain: tests/test_stab_host.py).  Three of the six non-trivial
  // reflectors of a three-supporter system have such a one-element tail.
  if (n == 1) return fabs(x[0]);
#endif
#if !defined(PCT_GELSD_NO_NRM2_CERTIFIED)
  if (n <= 8) {
    double r;
    if (dnrm2_certified(n, x, incx, r)) return r;
  }
#endif
  return dnrm2_ext(n, x, incx);
}
// the emulation itself (n >= 1)
PCT_GD double dnrm2_ext(int n, const double* x, int incx) {
  Ext a, b, c, d;
  a.m = b.m = c.m = d.m = 0;
  a.e = b.e = c.e = d.e = 0;
  const int n8 = n & ~7;
  PCT_GNOUNROLL
  for (int i0 = 0; i0 < n; i0 += 8) {
    double v0, v1, v2, v3, v4, v5, v6, v7;
    v0 = x[i0 * incx];
    v1 = i0 + 1 < n ? x[(i0 + 1) * incx] : 0.0;
    v2 = i0 + 2 < n ? x[(i0 + 2) * incx] : 0.0;
    v3 = i0 + 3 < n ? x[(i0 + 3) * incx] : 0.0;
    v4 = i0 + 4 < n ? x[(i0 + 4) * incx] : 0.0;
    v5 = i0 + 5 < n ? x[(i0 + 5) * incx] : 0.0;
    v6 = i0 + 6 < n ? x[(i0 + 6) * incx] : 0.0;
    v7 = i0 + 7 < n ? x[(i0 + 7) * incx] : 0.0;
    const int cnt = n - i0 < 8 ? n - i0 : 8;
    PCT_GNOUNROLL
    for (int u = 0; u < cnt; u++) {
      const double lo4 = (u & 2) ? ((u & 1) ? v3 : v2) : ((u & 1) ? v1 : v0);
      const double hi4 = (u & 2) ? ((u & 1) ? v7 : v6) : ((u & 1) ? v5 : v4);
      const Ext sq = ext_square((u & 4) ? hi4 : lo4);
      const int w = i0 < n8 ? (u & 3) : 0;
      Ext cur = w == 0 ? a : (w == 1 ? b : (w == 2 ? c : d));
      cur = ext_add(cur, sq);
      if (w == 0) a = cur;
      else if (w == 1) b = cur;
      else if (w == 2) c = cur;
      else d = cur;
    }
  }
  Ext t = c;
  PCT_GNOUNROLL
  for (int k = 0; k < 3; k++) t = ext_add(t, k == 0 ? a : (k == 1 ? b : d));
  return ext_sqrt_to_double(t);
}
// dnrm2 by the group, for vectors long enough to pay two hand-overs through LDS: the squares an element per lane, then the four
// accumulator chains (A: elements 0, 4, 8 ... of the leading blocks of eight and then the tail, in order; B, C, D: elements 1, 2, 3
// mod 4 of the blocks) a chain per lane -- every chain the serial x87 sum it is on the FPU -- and ((C + A) + B) + D, the root and the
// rounding in every lane.  The value is dnrm2's bit for bit: the same additions in the same order within each accumulator.
// scratch: 2 n + 8 doubles nobody else uses meanwhile (split_t: the V^T region, idle until dbdsqr).
#ifndef PCT_GELSD_COOP_NRM2_MIN
#define PCT_GELSD_COOP_NRM2_MIN 5
#endif
PCT_GD double dnrm2_g(Grp g, int n, const double* x, int incx, double* scratch) {
#if !defined(PCT_GELSD_NO_NRM2_CERTIFIED)
  if (n > 1 && n <= 8) {  // (every lane of the group the same arithmetic on the same elements: the outcome is group-uniform)
    double r;
    if (dnrm2_certified(n, x, incx, r)) return r;
  }
#endif
#if !defined(PCT_GELSD_COOP_NRM2_ALWAYS)
  if (n <= 1) return dnrm2(n, x, incx);
  if (g.G < 4 || n < PCT_GELSD_COOP_NRM2_MIN) return dnrm2_ext(n, x, incx);
#else
  if (n <= 1) return dnrm2(n, x, incx);
#endif
  uint64_t* sm = reinterpret_cast<uint64_t*>(scratch);  // [2 i] mantissa, [2 i + 1] exponent of element i's square; then the four partial sums
  const int n8 = n & ~7;
  PCT_GSYNC();
  for (int i = g.gl; i < n; i += g.G) {
    const Ext sq = ext_square(x[i * incx]);
    sm[2 * i] = sq.m;
    sm[2 * i + 1] = (uint64_t)(int64_t)sq.e;
  }
  PCT_GSYNC();
  for (int w = g.gl; w < 4; w += g.G) {
    Ext acc;
    acc.m = 0; acc.e = 0;
    PCT_GNOUNROLL
    for (int i = w; i < n8; i += 4) {
      Ext sq;
      sq.m = sm[2 * i];
      sq.e = (int)(int64_t)sm[2 * i + 1];
      acc = ext_add(acc, sq);
    }
    if (w == 0) {
      PCT_GNOUNROLL
      for (int i = n8; i < n; i++) {
        Ext sq;
        sq.m = sm[2 * i];
        sq.e = (int)(int64_t)sm[2 * i + 1];
        acc = ext_add(acc, sq);
      }
    }
    sm[2 * n + 2 * w] = acc.m;
    sm[2 * n + 2 * w + 1] = (uint64_t)(int64_t)acc.e;
  }
  PCT_GSYNC();
  Ext p[4];
  for (int w = 0; w < 4; w++) {
    p[w].m = sm[2 * n + 2 * w];
    p[w].e = (int)(int64_t)sm[2 * n + 2 * w + 1];
  }
  PCT_GSYNC();
  Ext t = p[2];
  t = ext_add(t, p[0]);
  t = ext_add(t, p[1]);
  t = ext_add(t, p[3]);
  return ext_sqrt_to_double(t);
}
// x := alpha x, an element per lane
PCT_GD void dscal(Grp g, int n, double alpha, double* x, int incx) {
  for (int i = g.gl; i < n; i += g.G) x[i * incx] = alpha * x[i * incx];
}
// One column of y = A^T x (alpha = 1, beta = 0), column j of n (c points at it): rows in groups of four -- by the 4x4 kernel
// (AVX2 FMA, four lanes, (l0 + l2) + (l1 + l3)) for the columns of the leading groups of four, then two columns by the 4x2
// kernel (SSE2, two lanes, products and sums rounded separately, l0 + l1), then one by the 4x1 kernel (SSE2, (r0 + r2) +
// (r1 + r3)) -- and the m mod 4 tail rows: t = a1 x1; fma(a0, x0, t); fma(a2, x2, t); y + t.  Which kernel a column meets
// depends on its POSITION j among the n columns of the call.
PCT_GD double dgemv_t_col(int m, int n, int j, const double* c, const double* x, int incx) {
  const int m2 = m & ~3, m3 = m & 3, n4 = n & ~3;
  double yj = 0.0;
  if (m2 > 0) {
    if (j < n4) {
      double l0 = 0, l1 = 0, l2 = 0, l3 = 0;
      for (int i = 0; i < m2; i += 4) {
        l0 = fma(c[i], x[i * incx], l0);
        l1 = fma(c[i + 1], x[(i + 1) * incx], l1);
        l2 = fma(c[i + 2], x[(i + 2) * incx], l2);
        l3 = fma(c[i + 3], x[(i + 3) * incx], l3);
      }
      const double t = (l0 + l2) + (l1 + l3);
      yj = yj + t * 1.0;
    } else if ((n & 2) && j < n4 + 2) {
      double l0 = 0, l1 = 0;
      for (int i = 0; i < m2; i += 2) {
        l0 = l0 + c[i] * x[i * incx];
        l1 = l1 + c[i + 1] * x[(i + 1) * incx];
      }
      yj = fma(1.0, l0 + l1, yj);
    } else {
      double l0 = 0, l1 = 0, l2 = 0, l3 = 0;
      for (int i = 0; i < m2; i += 4) {
        l0 = l0 + c[i] * x[i * incx];
        l1 = l1 + c[i + 1] * x[(i + 1) * incx];
        l2 = l2 + c[i + 2] * x[(i + 2) * incx];
        l3 = l3 + c[i + 3] * x[(i + 3) * incx];
      }
      yj = fma(1.0, (l0 + l2) + (l1 + l3), yj);
    }
  }
  if (m3 == 1) yj = fma(c[m2], x[m2 * incx], yj);
  else if (m3 >= 2) {
    double t = c[m2 + 1] * x[(m2 + 1) * incx];
    t = fma(c[m2], x[m2 * incx], t);
    if (m3 == 3) t = fma(c[m2 + 2], x[(m2 + 2) * incx], t);
    yj = yj + t;
  }
  return yj;
}
// One row of y = A x (alpha = 1, beta = 0), row i of m, with a STRIDED x (dlarf from the right in dgebd2: x is a row of A):
// rows in groups of four -- columns in groups of four (t = a1 x1; fma a0 x0; fma a2 x2; fma a3 x3; y = fma(1, t, y)), the
// rest one by one, y + a x rounded separately -- and the tail rows: t = fma(a, x, t) over all columns.  avx2: the Haswell
// kernel set instead
PCT_GD double dgemv_n_row(int m, int n, int i, const double* a, int lda, const double* x, int incx, bool avx2) {
  const int m2 = m & ~3, n4 = n & ~3;
  if (i < m2) {
    double yi = 0.0;
    for (int j = 0; j < n4; j += 4) {
      double t;
      if (avx2) {  // Haswell microkernel: two FMA chains, (a0 x0 + a2 x2) + (a1 x1 + a3 x3)
        double t4 = a[i + j * lda] * x[j * incx], t5 = a[i + (j + 1) * lda] * x[(j + 1) * incx];
        t4 = fma(a[i + (j + 2) * lda], x[(j + 2) * incx], t4);
        t5 = fma(a[i + (j + 3) * lda], x[(j + 3) * incx], t5);
        t = t4 + t5;
      } else {     // SkylakeX microkernel: one chain starting from a1 x1
        t = a[i + (j + 1) * lda] * x[(j + 1) * incx];
        t = fma(a[i + j * lda], x[j * incx], t);
        t = fma(a[i + (j + 2) * lda], x[(j + 2) * incx], t);
        t = fma(a[i + (j + 3) * lda], x[(j + 3) * incx], t);
      }
      yi = fma(1.0, t, yi);
    }
    for (int j = n4; j < n; j++) yi = yi + a[i + j * lda] * (x[j * incx] * 1.0);
    return yi;
  }
  // the C tail: one FMA chain (SkylakeX build) / products and sums rounded separately (Haswell build)
  double t = 0.0;
  for (int j = 0; j < n; j++) t = avx2 ? t + a[i + j * lda] * x[j * incx] : fma(a[i + j * lda], x[j * incx], t);
  return avx2 ? 0.0 + 1.0 * t : fma(1.0, t, 0.0);
}
// dger's element: A += alpha x y^T is, per column, t = alpha y_j and a_ij = fma(t, x_i, a_ij) (daxpy; Haswell daxpy: blocks of
// sixteen through the FMA microkernel, the rest multiply and add rounded separately -- `fused` says which)
PCT_GD double dger_elem(double t, double xi, double aij, bool fused) { return fused ? fma(t, xi, aij) : aij + t * xi; }

// ---- LAPACK 3.11 (compiled Fortran: plain double arithmetic in program order) ----------------------------------------------
constexpr double EPS = 1.1102230246251565e-16;      // dlamch('E')
constexpr double SAFMIN = 2.2250738585072014e-308;  // dlamch('S')
PCT_GD double sgn(double a, double b) { return copysign(fabs(a), b); }

PCT_GD double dlapy2(double x, double y) {
  const double xa = fabs(x), ya = fabs(y);
  const double w = xa > ya ? xa : ya, z = xa < ya ? xa : ya;
  if (z == 0.0 || w > 1.7976931348623157e308) return w;
  const double q = z / w;
  return w * sqrt(1.0 + q * q);
}
// dlarfg(n, alpha, x, incx): returns tau; x is scaled in place (an element per lane); `beta` is what dlarfg leaves in alpha
// (alpha itself when tau = 0) -- the caller stores it when it has applied the reflector, whose leading 1 it writes there first
PCT_GD double dlarfg(Grp g, int n, const double* alpha, double* x, int incx, double& beta, double* nrm2_scratch) {
  double a0 = *alpha;
  beta = a0;
  if (n <= 1) return 0.0;
  const double safmin = SAFMIN / EPS, rsafmn = 1.0 / safmin;
  int knt = 0;
  PCT_GNOUNROLL
  for (int pass = 0; pass < 2; pass++) {  // (the second pass: after a rescaling of a tiny vector, which these systems never need)
    PCT_GPROF_T0(tn)
    const double xnorm = dnrm2_g(g, n - 1, x, incx, nrm2_scratch);
    PCT_GPROF_ADD(5, tn)
    if (pass == 0 && xnorm == 0.0) return 0.0;
    beta = -sgn(dlapy2(a0, xnorm), a0);
    if (pass == 1 || !(fabs(beta) < safmin)) break;
    do {
      knt++;
      PCT_GSYNC();
      dscal(g, n - 1, rsafmn, x, incx);
      PCT_GSYNC();
      beta *= rsafmn;
      a0 *= rsafmn;
    } while (fabs(beta) < safmin && knt < 20);
  }
  const double tau = (beta - a0) / beta;
  PCT_GSYNC();
  dscal(g, n - 1, 1.0 / (a0 - beta), x, incx);
  PCT_GSYNC();
  for (int j = 0; j < knt; j++) beta *= safmin;
  return tau;
}
// dlarf from the left, H = I - tau v v^T (v[0] = 1 stored): onto the nc columns at c (the call on A's trailing columns) and --
// a call of its own in LAPACK (dorm2r / dorml2: one column) -- onto the column bcol (or null).  A column per lane: its dgemv 'T'
// sum and its dger update (work(j) never leaves the lane).  Trailing zeros of v and trailing zero columns of C are skipped (iladlr / iladlc).
PCT_GD void dlarf_left(Grp g, int m, int nc, const double* v, int incv, double tau, double* c, int ldc, double* bcol, bool avx2) {
  if (tau == 0.0) return;
  int lastv = m;
  while (lastv > 0 && v[(lastv - 1) * incv] == 0.0) lastv--;
  if (lastv <= 0) return;
  int lastc = nc;
  for (; lastc > 0; lastc--) {
    bool nz = false;
    for (int i = 0; i < lastv; i++)
      if (c[i + (lastc - 1) * ldc] != 0.0) { nz = true; break; }
    if (nz) break;
  }
  const int mf = avx2 ? (lastv & ~15) : lastv;
  PCT_GSYNC();
  PCT_GNOUNROLL
  for (int t = g.gl; t < lastc + (bcol ? 1 : 0); t += g.G) {
    const bool isb = t >= lastc;
    double* col = isb ? bcol : c + t * ldc;
    if (isb) {  // (the call on one column: it is skipped when that column is zero)
      bool nz = false;
      for (int i = 0; i < lastv; i++)
        if (col[i] != 0.0) { nz = true; break; }
      if (!nz) continue;
    }
    const double w = dgemv_t_col(lastv, isb ? 1 : lastc, isb ? 0 : t, col, v, incv);
    const double tt = -tau * w;
    for (int i = 0; i < lastv; i++) col[i] = dger_elem(tt, v[i * incv], col[i], i < mf);
  }
  PCT_GSYNC();
}
// dlarf from the right: C (m x n) := C (I - tau v v^T), v a strided row.  A row per lane: its dgemv 'N' sum, its dger update
PCT_GD void dlarf_right(Grp g, int m, int n, const double* v, int incv, double tau, double* c, int ldc, bool avx2) {
  if (tau == 0.0) return;
  int lastv = n;
  while (lastv > 0 && v[(lastv - 1) * incv] == 0.0) lastv--;
  if (lastv <= 0) return;
  int lastc = m;
  for (; lastc > 0; lastc--) {
    bool nz = false;
    for (int j = 0; j < lastv; j++)
      if (c[(lastc - 1) + j * ldc] != 0.0) { nz = true; break; }
    if (nz) break;
  }
  const int mf = avx2 ? (lastc & ~15) : lastc;
  PCT_GSYNC();
  PCT_GNOUNROLL
  for (int i = g.gl; i < lastc; i += g.G) {
    const double w = dgemv_n_row(lastc, lastv, i, c, ldc, v, incv, avx2);
    for (int j = 0; j < lastv; j++) {
      const double tt = -tau * v[j * incv];
      c[i + j * ldc] = dger_elem(tt, w, c[i + j * ldc], i < mf);
    }
  }
  PCT_GSYNC();
}
PCT_GD void dlartg(double f, double g, double& c, double& s, double& r) {
  const double safmax = 1.0 / SAFMIN;
  const double rtmin = 1.4916681462400413e-154;  // sqrt(safmin)
  const double rtmax = 4.7403759540545887e+153;  // sqrt(safmax / 2)
  const double f1 = fabs(f), g1 = fabs(g);
  if (g == 0.0) { c = 1.0; s = 0.0; r = f; }
  else if (f == 0.0) { c = 0.0; s = sgn(1.0, g); r = g1; }
  else if (f1 > rtmin && f1 < rtmax && g1 > rtmin && g1 < rtmax) {
    const double d = sqrt(f * f + g * g);
    c = f1 / d;
    r = sgn(d, f);
    s = g / r;
  } else {
    double u = f1 > g1 ? f1 : g1;
    if (u < SAFMIN) u = SAFMIN;
    if (u > safmax) u = safmax;
    const double fs = f / u, gs = g / u;
    const double d = sqrt(fs * fs + gs * gs);
    c = fabs(fs) / d;
    r = sgn(d, f);
    s = gs / r;
    r = r * u;
  }
}
PCT_GD void dlas2(double f, double g, double h, double& ssmin, double& ssmax) {
  const double fa = fabs(f), ga = fabs(g), ha = fabs(h);
  const double fhmn = fa < ha ? fa : ha, fhmx = fa > ha ? fa : ha;
  if (fhmn == 0.0) {
    ssmin = 0.0;
    if (fhmx == 0.0) ssmax = ga;
    else {
      const double mx = fhmx > ga ? fhmx : ga, mn = fhmx < ga ? fhmx : ga;
      const double q = mn / mx;
      ssmax = mx * sqrt(1.0 + q * q);
    }
  } else if (ga < fhmx) {
    const double as = 1.0 + fhmn / fhmx, at = (fhmx - fhmn) / fhmx;
    const double q = ga / fhmx, au = q * q;
    const double c = 2.0 / (sqrt(as * as + au) + sqrt(at * at + au));
    ssmin = fhmn * c;
    ssmax = fhmx / c;
  } else {
    const double au = fhmx / ga;
    if (au == 0.0) { ssmin = (fhmn * fhmx) / ga; ssmax = ga; }
    else {
      const double as = 1.0 + fhmn / fhmx, at = (fhmx - fhmn) / fhmx;
      const double p = as * au, q = at * au;
      const double c = 1.0 / (sqrt(1.0 + p * p) + sqrt(1.0 + q * q));
      ssmin = (fhmn * c) * au;
      ssmin = ssmin + ssmin;
      ssmax = ga / (c + c);
    }
  }
}
PCT_GD void dlasv2(double f, double g, double h, double& ssmin, double& ssmax, double& snr, double& csr, double& snl, double& csl) {
  double ft = f, fa = fabs(ft), ht = h, ha = fabs(h);
  int pmax = 1;
  const bool swap = ha > fa;
  if (swap) { pmax = 3; double t = ft; ft = ht; ht = t; t = fa; fa = ha; ha = t; }
  const double gt = g, ga = fabs(gt);
  double clt = 1.0, crt = 1.0, slt = 0.0, srt = 0.0;
  if (ga == 0.0) { ssmin = ha; ssmax = fa; }
  else {
    bool gasmal = true;
    if (ga > fa) {
      pmax = 2;
      if (fa / ga < EPS) {
        gasmal = false;
        ssmax = ga;
        if (ha > 1.0) ssmin = fa / (ga / ha); else ssmin = (fa / ga) * ha;
        clt = 1.0; slt = ht / gt; srt = 1.0; crt = ft / gt;
      }
    }
    if (gasmal) {
      const double d = fa - ha;
      double l = d == fa ? 1.0 : d / fa;
      const double m = gt / ft;
      double t = 2.0 - l;
      const double mm = m * m, tt = t * t;
      const double s = sqrt(tt + mm);
      const double r = l == 0.0 ? fabs(m) : sqrt(l * l + mm);
      const double a = 0.5 * (s + r);
      ssmin = ha / a;
      ssmax = fa * a;
      if (mm == 0.0) {
        if (l == 0.0) t = sgn(2.0, ft) * sgn(1.0, gt);
        else t = gt / sgn(d, ft) + m / t;
      } else t = (m / (s + t) + m / (r + l)) * (1.0 + a);
      l = sqrt(t * t + 4.0);
      crt = 2.0 / l;
      srt = t / l;
      clt = (crt + srt * m) / a;
      slt = (ht / ft) * srt / a;
    }
  }
  if (swap) { csl = srt; snl = crt; csr = slt; snr = clt; }
  else { csl = clt; snl = slt; csr = crt; snr = srt; }
  double tsign;
  if (pmax == 1) tsign = sgn(1.0, csr) * sgn(1.0, csl) * sgn(1.0, f);
  else if (pmax == 2) tsign = sgn(1.0, snr) * sgn(1.0, csl) * sgn(1.0, g);
  else tsign = sgn(1.0, snr) * sgn(1.0, snl) * sgn(1.0, h);
  ssmax = sgn(ssmax, tsign);
  ssmin = sgn(ssmin, tsign * sgn(1.0, f) * sgn(1.0, h));
}
// dlascl('G', 0, 0, cfrom, cto, ...) on a vector (the multipliers: every lane; the elements: a lane each)
PCT_GD void dlascl_vec(Grp g, double cfrom, double cto, int n, double* x) {
  const double smlnum = SAFMIN, bignum = 1.0 / smlnum;
  double cfromc = cfrom, ctoc = cto, mul;
  bool done;
  PCT_GSYNC();
  do {
    const double cfrom1 = cfromc * smlnum;
    if (cfrom1 == cfromc) { mul = ctoc / cfromc; done = true; }
    else {
      const double cto1 = ctoc / bignum;
      if (cto1 == ctoc) { mul = ctoc; done = true; cfromc = 1.0; }
      else if (fabs(cfrom1) > fabs(ctoc) && ctoc != 0.0) { mul = smlnum; done = false; cfromc = cfrom1; }
      else if (fabs(cto1) > fabs(cfromc)) { mul = bignum; done = false; ctoc = cto1; }
      else { mul = ctoc / cfromc; done = true; if (mul == 1.0) break; }
    }
    for (int i = g.gl; i < n; i += g.G) x[i] = x[i] * mul;
  } while (!done);
  PCT_GSYNC();
}

// ---- dbdsqr for n = 3 (three supporters: 94 % of the on-domain solves), round 6 -----------------------------------------------------
// The generic routine below keeps d, e and the sweep's rotations (work) in LDS: every d(i) / e(i) of the split search, the
// convergence test, the shift and the chase is an LDS round trip in a dependent chain -- ~30 of them per iteration, 34 k of the 73 k
// cycles of a three-supporter solve (profiles/r06_microbench_gelsd.txt).  For n = 3 every index is known: a block is either the
// whole matrix (ll = 1, m = 3: a sweep over x0 x1 x2 / y0 y1 in chase order, idir 1 = top down, 2 = bottom up) or a 2 x 2 corner
// (dlasv2).  So d0 d1 d2 e0 e1 are five REGISTERS for the whole routine, the rotations of a sweep stay in the registers of the lane
// that computed them (every lane of the group runs the scalar chain anyway), and a lane touches only ITS columns of VT / cc (three
// loads in flight, the sweep's rotations, three stores): no hand-over inside the routine: 27 k cycles.  (A first version for n <= 4
// with run-time positions through select chains was SLOWER than the LDS routine -- 45 k against 34 k cycles -- as round 5 had found
// for the whole solve; a version for n = 4 .. 6 that keeps d, e in LDS but reads them in two batches of independent loads per
// iteration gained 5 - 10 % (71 -> 67 k, 102 -> 92 k): what is left is the division / square-root chain of the rotations themselves,
// ~22 of them per iteration at ~150 cycles.  Neither is in the library; profiles/r06_experiments.txt.)  Operation for operation the generic routine (same tests, same order, same expressions):
// tests/test_stab_host.py compares the two on 2 * 10^6 random bidiagonals, and both with the oracle's dbdsqr.
PCT_GD bool dbdsqr3(Grp g, double* d, double* e, double* vt, double* cc) {
  const int n = 3, ldvt = 3;
  const double hndrth = 0.01;
  const int maxitr = 6;
  double d0 = d[0], d1 = d[1], d2 = d[2], e0 = e[0], e1 = e[1];
  const double tol = 0x1.8ace5422aa0dbp+6 * EPS;
  double smax = 0.0;
  smax = fabs(d0) > smax ? fabs(d0) : smax;
  smax = fabs(d1) > smax ? fabs(d1) : smax;
  smax = fabs(d2) > smax ? fabs(d2) : smax;
  smax = fabs(e0) > smax ? fabs(e0) : smax;
  smax = fabs(e1) > smax ? fabs(e1) : smax;
  double smin = 0.0;
  double sminoa = fabs(d0);
  if (sminoa != 0.0) {
    double mu = sminoa;
    mu = fabs(d1) * (mu / (mu + fabs(e0)));
    sminoa = mu < sminoa ? mu : sminoa;
    if (sminoa != 0.0) {
      mu = fabs(d2) * (mu / (mu + fabs(e1)));
      sminoa = mu < sminoa ? mu : sminoa;
    }
  }
  sminoa = sminoa / sqrt((double)n);
  double thresh = tol * sminoa;
  {
    const double t2 = maxitr * (n * (n * SAFMIN));
    thresh = t2 > thresh ? t2 : thresh;
  }
  const int maxitdivn = maxitr * n;
  int iterdivn = 0, iter = -1, idir = 0;
  bool swept = false;  // (oldll = 1, oldm = 3 once the whole block has been swept: idir is chosen at its first visit only)
  int m = 3;
  // rotation of rows (r, r + 1) of the lane's own columns (the 2 x 2 corner's drot: x' = fma(c, x, s y), y' = fma(c, y, -(s x)))
  auto rot2 = [&](int r, double cosr, double sinr, double cosl, double sinl) __attribute__((always_inline)) {
    for (int i = g.gl; i < n + 1; i += g.G) {
      double* px = i < n ? &vt[r + i * ldvt] : &cc[r];
      const double cr = i < n ? cosr : cosl, sr = i < n ? sinr : sinl;
      const double xv = px[0], yv = px[1];
      px[0] = fma(cr, xv, sr * yv);
      px[1] = fma(cr, yv, -(sr * xv));
    }
  };
  PCT_GNOUNROLL
  while (m > 1) {
    if (iter >= n) {
      iter -= n;
      iterdivn++;
      if (iterdivn >= maxitdivn) return false;
    }
    if (m == 2) {
      // rows 0, 1: either e0 is negligible (the block splits into two 1 x 1) or the 2 x 2 corner is diagonalised
      if (fabs(e0) <= thresh) { e0 = 0.0; m = 1; continue; }
      double sigmn, sigmx, sinr, cosr, sinl, cosl;
      dlasv2(d0, e0, d1, sigmn, sigmx, sinr, cosr, sinl, cosl);
      d0 = sigmx; e0 = 0.0; d1 = sigmn;
      rot2(0, cosr, sinr, cosl, sinl);
      m = 0;
      continue;
    }
    // m == 3: look for a negligible off-diagonal entry from the bottom
    smax = fabs(d2);
    if (fabs(e1) <= thresh) { e1 = 0.0; m = 2; continue; }
    smax = fabs(d1) > smax ? fabs(d1) : smax;
    smax = fabs(e1) > smax ? fabs(e1) : smax;
    if (fabs(e0) <= thresh) {
      // split above row 1: the 2 x 2 corner of rows 1, 2
      e0 = 0.0;
      double sigmn, sigmx, sinr, cosr, sinl, cosl;
      dlasv2(d1, e1, d2, sigmn, sigmx, sinr, cosr, sinl, cosl);
      d1 = sigmx; e1 = 0.0; d2 = sigmn;
      rot2(1, cosr, sinr, cosl, sinl);
      m = 1;
      continue;
    }
    smax = fabs(d0) > smax ? fabs(d0) : smax;
    smax = fabs(e0) > smax ? fabs(e0) : smax;
    // the whole matrix is one block (ll = 1, m = 3)
    if (!swept) idir = fabs(d0) >= fabs(d2) ? 1 : 2;
    bool conv = false;
    if (idir == 1) {
      if (fabs(e1) <= fabs(tol) * fabs(d2)) { e1 = 0.0; continue; }
      double mu = fabs(d0);
      smin = mu;
      if (fabs(e0) <= tol * mu) { e0 = 0.0; conv = true; }
      else {
        mu = fabs(d1) * (mu / (mu + fabs(e0)));
        smin = mu < smin ? mu : smin;
        if (fabs(e1) <= tol * mu) { e1 = 0.0; conv = true; }
        else {
          mu = fabs(d2) * (mu / (mu + fabs(e1)));
          smin = mu < smin ? mu : smin;
        }
      }
    } else {
      if (fabs(e0) <= fabs(tol) * fabs(d0)) { e0 = 0.0; continue; }
      double mu = fabs(d2);
      smin = mu;
      if (fabs(e1) <= tol * mu) { e1 = 0.0; conv = true; }
      else {
        mu = fabs(d1) * (mu / (mu + fabs(e1)));
        smin = mu < smin ? mu : smin;
        if (fabs(e0) <= tol * mu) { e0 = 0.0; conv = true; }
        else {
          mu = fabs(d0) * (mu / (mu + fabs(e0)));
          smin = mu < smin ? mu : smin;
        }
      }
    }
    if (conv) continue;
    swept = true;
    double shift = 0.0, r = 0.0;
    {
      const double bound = EPS > hndrth * tol ? EPS : hndrth * tol;
      if (!(n * tol * (smin / smax) <= bound)) {
        double sll;
        if (idir == 1) { sll = fabs(d0); dlas2(d1, e1, d2, shift, r); }
        else { sll = fabs(d2); dlas2(d0, e0, d1, shift, r); }
        if (sll > 0.0) {
          const double q = shift / sll;
          if (q * q < EPS) shift = 0.0;
        }
      }
    }
    iter = iter + 2;  // m - ll
    // the block in chase order: downwards (idir 1) x = d0 d1 d2, y = e0 e1; upwards (idir 2) x = d2 d1 d0, y = e1 e0
    const bool up = idir != 1;
    double x0 = up ? d2 : d0, x1 = d1, x2 = up ? d0 : d2, y0 = up ? e1 : e0, y1 = up ? e0 : e1;
    const double sg = up ? -1.0 : 1.0;  // (the upward chase stores its sines negated: dbdsqr's WORK(.) = -SN)
    double c00, s00, c10, s10, c01, s01, c11, s11;  // the sweep's rotations: (w0, w1)[p] = (c0p, s0p), (w2, w3)[p] = (c1p, s1p)
    if (shift == 0.0) {
      double cs = 1.0, oldcs = 1.0, sn = 0.0, oldsn = 0.0, dn;
      dlartg(x0 * cs, y0, cs, sn, r);
      dlartg(oldcs * r, x1 * sn, oldcs, oldsn, dn);
      x0 = dn;
      c00 = cs; s00 = sg * sn; c10 = oldcs; s10 = sg * oldsn;
      dlartg(x1 * cs, y1, cs, sn, r);
      y0 = oldsn * r;
      dlartg(oldcs * r, x2 * sn, oldcs, oldsn, dn);
      x1 = dn;
      c01 = cs; s01 = sg * sn; c11 = oldcs; s11 = sg * oldsn;
      const double h = x2 * cs;
      x2 = h * oldcs;
      y1 = h * oldsn;
    } else {
      double f = (fabs(x0) - shift) * (sgn(1.0, x0) + shift / x0);
      double gg = y0;
      double cosr, sinr, cosl, sinl;
      dlartg(f, gg, cosr, sinr, r);
      f = cosr * x0 + sinr * y0;
      y0 = cosr * y0 - sinr * x0;
      gg = sinr * x1;
      x1 = cosr * x1;
      dlartg(f, gg, cosl, sinl, r);
      x0 = r;
      f = cosl * y0 + sinl * x1;
      x1 = cosl * x1 - sinl * y0;
      gg = sinl * y1;
      y1 = cosl * y1;
      c00 = cosr; s00 = sg * sinr; c10 = cosl; s10 = sg * sinl;
      dlartg(f, gg, cosr, sinr, r);
      y0 = r;
      f = cosr * x1 + sinr * y1;
      y1 = cosr * y1 - sinr * x1;
      gg = sinr * x2;
      x2 = cosr * x2;
      dlartg(f, gg, cosl, sinl, r);
      x1 = r;
      f = cosl * y1 + sinl * x2;
      x2 = cosl * x2 - sinl * y1;
      c01 = cosr; s01 = sg * sinr; c11 = cosl; s11 = sg * sinl;
      y1 = f;
    }
    if (fabs(y1) <= thresh) y1 = 0.0;  // (the entry the chase ends on: e(m - 1) downwards, e(ll) upwards)
    d0 = up ? x2 : x0; d1 = x1; d2 = up ? x0 : x2; e0 = up ? y1 : y0; e1 = up ? y0 : y1;
    // dlasr('L', 'V', 'F' / 'B') on the lane's own columns: the right rotations (the first of a step's pair) go into VT when chasing
    // downwards, into cc when chasing upwards, the left ones into the other; rows in chase order, rotated, stored back
    PCT_GNOUNROLL
    for (int c = g.gl; c < n + 1; c += g.G) {
      const bool first = (c < n) == (idir == 1);
      double* col = c < n ? vt + c * ldvt : cc;
      const double a0 = col[0], a1 = col[1], a2 = col[2];
      double v0 = up ? a2 : a0, v1 = a1, v2 = up ? a0 : a2;
      {
        const double ct = first ? c00 : c10, st = first ? s00 : s10;
        if (ct != 1.0 || st != 0.0) {
          // rows (dp, dq) = chase positions (0, 1); downwards: high = dq, low = dp; upwards: high = dp, low = dq
          const double hi = up ? v0 : v1, lo = up ? v1 : v0;
          const double nh = ct * hi - st * lo, nl = st * hi + ct * lo;
          v1 = up ? nl : nh;
          v0 = up ? nh : nl;
        }
      }
      {
        const double ct = first ? c01 : c11, st = first ? s01 : s11;
        if (ct != 1.0 || st != 0.0) {
          const double hi = up ? v1 : v2, lo = up ? v2 : v1;
          const double nh = ct * hi - st * lo, nl = st * hi + ct * lo;
          v2 = up ? nl : nh;
          v1 = up ? nh : nl;
        }
      }
      col[0] = up ? v2 : v0; col[1] = v1; col[2] = up ? v0 : v2;
    }
  }
  // signs, then decreasing order (one transposition per singular value), as the generic routine
  if (d0 == 0.0) d0 = 0.0;
  if (d0 < 0.0) { d0 = -d0; dscal(g, n, -1.0, &vt[0], ldvt); }
  if (d1 == 0.0) d1 = 0.0;
  if (d1 < 0.0) { d1 = -d1; dscal(g, n, -1.0, &vt[1], ldvt); }
  if (d2 == 0.0) d2 = 0.0;
  if (d2 < 0.0) { d2 = -d2; dscal(g, n, -1.0, &vt[2], ldvt); }
  auto swap_rows = [&](int ra, int rb) __attribute__((always_inline)) {
    for (int c = g.gl; c < n + 1; c += g.G) {
      double* col = c < n ? vt + c * ldvt : cc;
      const double t = col[ra];
      col[ra] = col[rb];
      col[rb] = t;
    }
  };
  {
    // i = 1: the smallest of d0 d1 d2 (the LAST one among equals) goes to position 2
    int isub = 0;
    double smn = d0;
    if (d1 <= smn) { isub = 1; smn = d1; }
    if (d2 <= smn) { isub = 2; smn = d2; }
    if (isub != 2) {
      if (isub == 0) d0 = d2; else d1 = d2;
      d2 = smn;
      swap_rows(isub, 2);
    }
    // i = 2: the smaller of d0 d1 goes to position 1
    if (!(d1 <= d0)) {
      const double t = d0;
      d0 = d1;
      d1 = t;
      swap_rows(0, 1);
    }
  }
  PCT_GSYNC();
  d[0] = d0; d[1] = d1; d[2] = d2; e[0] = e0; e[1] = e1;  // (every lane of the group the same values)
  PCT_GSYNC();
  return true;
}

PCT_GD bool dbdsqr_generic(Grp g, int n, double* d, double* e, double* vt, double* cc, double* work);
// dbdsqr('U', n, ncvt = n, 0, ncc = 1): SVD of the upper bidiagonal (d, e); right rotations into VT (n x n, ldvt = n), left
// ones into the column cc.  work: 4 (n - 1) doubles.  Returns false when the iteration limit is reached.
// The chase itself (d, e, the rotations' cosines and sines into work) is the scalar chain every lane runs; a sweep's rotations
// are then applied a COLUMN of VT (and the column cc) per lane.
PCT_GD bool dbdsqr(Grp g, int n, double* d, double* e, double* vt, double* cc, double* work) {
#if !defined(PCT_GELSD_NO_BDSQR3)
  if (n == 3) return dbdsqr3(g, d, e, vt, cc);
#endif
  return dbdsqr_generic(g, n, d, e, vt, cc, work);
}
PCT_GD bool dbdsqr_generic(Grp g, int n, double* d, double* e, double* vt, double* cc, double* work) {
  const double hndrth = 0.01;
  const int maxitr = 6;
  const int ldvt = n;
  const int nm1 = n - 1, nm12 = nm1 + nm1, nm13 = nm12 + nm1;
  int idir = 0;
  if (n > 1) {
    const double tol = 0x1.8ace5422aa0dbp+6 * EPS;  // max(10, min(100, eps^(-1/8))) * eps
    double smax = 0.0;
    for (int i = 0; i < n; i++) smax = fabs(d[i]) > smax ? fabs(d[i]) : smax;
    for (int i = 0; i < n - 1; i++) smax = fabs(e[i]) > smax ? fabs(e[i]) : smax;
    double smin = 0.0;
    double sminoa = fabs(d[0]);
    if (sminoa != 0.0) {
      double mu = sminoa;
      for (int i = 1; i < n; i++) {
        mu = fabs(d[i]) * (mu / (mu + fabs(e[i - 1])));
        sminoa = mu < sminoa ? mu : sminoa;
        if (sminoa == 0.0) break;
      }
    }
    sminoa = sminoa / sqrt((double)n);
    double thresh = tol * sminoa;
    {
      const double t2 = maxitr * (n * (n * SAFMIN));
      thresh = t2 > thresh ? t2 : thresh;
    }
    const int maxitdivn = maxitr * n;
    int iterdivn = 0, iter = -1, oldll = -1, oldm = -1;
    int m = n;
    while (m > 1) {
      if (iter >= n) {
        iter -= n;
        iterdivn++;
        if (iterdivn >= maxitdivn) return false;
      }
      smax = fabs(d[m - 1]);
      int ll = 0;
      bool split = false;
      for (int lll = 1; lll <= m - 1; lll++) {
        ll = m - lll;
        const double abss = fabs(d[ll - 1]), abse = fabs(e[ll - 1]);
        if (abse <= thresh) { split = true; break; }
        smax = abss > smax ? abss : smax;
        smax = abse > smax ? abse : smax;
      }
      if (split) {
        e[ll - 1] = 0.0;
        if (ll == m - 1) { m = m - 1; continue; }
      } else ll = 0;
      ll = ll + 1;
      if (ll == m - 1) {
        double sigmn, sigmx, sinr, cosr, sinl, cosl;
        dlasv2(d[m - 2], e[m - 2], d[m - 1], sigmn, sigmx, sinr, cosr, sinl, cosl);
        d[m - 2] = sigmx; e[m - 2] = 0.0; d[m - 1] = sigmn;
        // drot on rows m - 2, m - 1 of VT (an element pair per lane) and on the pair of cc (x' = fma(c, x, s y), y' = fma(c, y, -(s x)))
        PCT_GSYNC();
        for (int i = g.gl; i < n + 1; i += g.G) {
          double* px = i < n ? &vt[(m - 2) + i * ldvt] : &cc[m - 2];
          const double cr = i < n ? cosr : cosl, sr = i < n ? sinr : sinl;
          const double xv = px[0], yv = px[1];
          px[0] = fma(cr, xv, sr * yv);
          px[1] = fma(cr, yv, -(sr * xv));
        }
        PCT_GSYNC();
        m = m - 2;
        continue;
      }
      if (ll > oldm || m < oldll) idir = fabs(d[ll - 1]) >= fabs(d[m - 1]) ? 1 : 2;
      bool conv = false;
      if (idir == 1) {
        if (fabs(e[m - 2]) <= fabs(tol) * fabs(d[m - 1])) { e[m - 2] = 0.0; continue; }
        double mu = fabs(d[ll - 1]);
        smin = mu;
        for (int lll = ll; lll <= m - 1; lll++) {
          if (fabs(e[lll - 1]) <= tol * mu) { e[lll - 1] = 0.0; conv = true; break; }
          mu = fabs(d[lll]) * (mu / (mu + fabs(e[lll - 1])));
          smin = mu < smin ? mu : smin;
        }
      } else {
        if (fabs(e[ll - 1]) <= fabs(tol) * fabs(d[ll - 1])) { e[ll - 1] = 0.0; continue; }
        double mu = fabs(d[m - 1]);
        smin = mu;
        for (int lll = m - 1; lll >= ll; lll--) {
          if (fabs(e[lll - 1]) <= tol * mu) { e[lll - 1] = 0.0; conv = true; break; }
          mu = fabs(d[lll - 1]) * (mu / (mu + fabs(e[lll - 1])));
          smin = mu < smin ? mu : smin;
        }
      }
      if (conv) continue;
      oldll = ll; oldm = m;
      double shift = 0.0, r = 0.0;
      {
        const double bound = EPS > hndrth * tol ? EPS : hndrth * tol;
        if (!(n * tol * (smin / smax) <= bound)) {
          double sll;
          if (idir == 1) { sll = fabs(d[ll - 1]); dlas2(d[m - 2], e[m - 2], d[m - 1], shift, r); }
          else { sll = fabs(d[m - 1]); dlas2(d[ll - 1], e[ll - 1], d[ll], shift, r); }
          if (sll > 0.0) {
            const double q = shift / sll;
            if (q * q < EPS) shift = 0.0;
          }
        }
      }
      iter = iter + m - ll;
      // One sweep, written once for both chase directions: position p = 0 .. cnt - 1 runs from the larger end of the block to the
      // smaller (idir 1: rows ll .. m downwards, idir 2: rows m .. ll upwards); di(p) / ei(p): the 0-based indices of the p-th
      // diagonal entry and of the off-diagonal entry between positions p and p + 1.
      double* w0 = work;
      double* w1 = work + nm1;
      double* w2 = work + nm12;
      double* w3 = work + nm13;
      const int cnt = m - ll + 1;
      const int dbase = idir == 1 ? ll - 1 : m - 1, ebase = idir == 1 ? ll - 1 : m - 2, step = idir == 1 ? 1 : -1;
      const double sg = idir == 1 ? 1.0 : -1.0;  // (the upward chase stores its sines negated: dbdsqr's WORK(.) = -SN)
      if (shift == 0.0) {
        double cs = 1.0, oldcs = 1.0, sn = 0.0, oldsn = 0.0;
        PCT_GNOUNROLL
        for (int p = 0; p < cnt - 1; p++) {
          const int dp = dbase + step * p, dq = dp + step, ep = ebase + step * p;
          dlartg(d[dp] * cs, e[ep], cs, sn, r);
          if (p > 0) e[ep - step] = oldsn * r;
          double dn;
          dlartg(oldcs * r, d[dq] * sn, oldcs, oldsn, dn);
          d[dp] = dn;
          w0[p] = cs; w1[p] = sg * sn; w2[p] = oldcs; w3[p] = sg * oldsn;
        }
        const int dl = dbase + step * (cnt - 1);
        const double h = d[dl] * cs;
        d[dl] = h * oldcs;
        e[ebase + step * (cnt - 2)] = h * oldsn;
      } else {
        double f = (fabs(d[dbase]) - shift) * (sgn(1.0, d[dbase]) + shift / d[dbase]);
        double gg = e[ebase];
        double cosr, sinr, cosl, sinl;
        PCT_GNOUNROLL
        for (int p = 0; p < cnt - 1; p++) {
          const int dp = dbase + step * p, dq = dp + step, ep = ebase + step * p;
          dlartg(f, gg, cosr, sinr, r);
          if (p > 0) e[ep - step] = r;
          f = cosr * d[dp] + sinr * e[ep];
          e[ep] = cosr * e[ep] - sinr * d[dp];
          gg = sinr * d[dq];
          d[dq] = cosr * d[dq];
          dlartg(f, gg, cosl, sinl, r);
          d[dp] = r;
          f = cosl * e[ep] + sinl * d[dq];
          d[dq] = cosl * d[dq] - sinl * e[ep];
          if (p < cnt - 2) {
            gg = sinl * e[ep + step];
            e[ep + step] = cosl * e[ep + step];
          }
          w0[p] = cosr; w1[p] = sg * sinr; w2[p] = cosl; w3[p] = sg * sinl;
        }
        e[ebase + step * (cnt - 2)] = f;
      }
      // dlasr('L', 'V', 'F' / 'B'): the right rotations (the first pair of a step) go into VT when chasing downwards, into the
      // column cc when chasing upwards, the left rotations into the other one -- in the order of the sweep; a column per lane
      PCT_GSYNC();
      PCT_GNOUNROLL
      for (int c = g.gl; c < n + 1; c += g.G) {
        const bool first = (c < n) == (idir == 1);
        const double* cw = first ? w0 : w2;
        const double* sw = first ? w1 : w3;
        double* col = c < n ? vt + c * ldvt : cc;
        for (int p = 0; p < cnt - 1; p++) {
          const double ct = cw[p], st = sw[p];
          if (ct != 1.0 || st != 0.0) {
            const int dp = dbase + step * p, dq = dp + step;
            const int rhi = idir == 1 ? dq : dp, rlo = idir == 1 ? dp : dq;
            const double temp = col[rhi];
            col[rhi] = ct * temp - st * col[rlo];
            col[rlo] = st * temp + ct * col[rlo];
          }
        }
      }
      PCT_GSYNC();
      {
        const int el = ebase + step * (cnt - 2);
        if (fabs(e[el]) <= thresh) e[el] = 0.0;
      }
    }
  }
  for (int i = 0; i < n; i++) {
    if (d[i] == 0.0) d[i] = 0.0;
    if (d[i] < 0.0) {
      d[i] = -d[i];
      PCT_GSYNC();
      dscal(g, n, -1.0, &vt[i], ldvt);
      PCT_GSYNC();
    }
  }
  // decreasing order: one transposition per singular value (the rows of VT: a column per lane)
  for (int i = 1; i <= n - 1; i++) {
    int isub = 1;
    double smn = d[0];
    for (int j = 2; j <= n + 1 - i; j++)
      if (d[j - 1] <= smn) { isub = j; smn = d[j - 1]; }
    if (isub != n + 1 - i) {
      d[isub - 1] = d[n - i];
      d[n - i] = smn;
      PCT_GSYNC();
      for (int c = g.gl; c < n + 1; c += g.G) {
        double* col = c < n ? vt + c * ldvt : cc;
        const double t = col[isub - 1];
        col[isub - 1] = col[n - i];
        col[n - i] = t;
      }
      PCT_GSYNC();
    }
  }
  return true;
}

// doubles of workspace the solve of an M x n system takes: A (M n), b (M), VT (n n), tau / tauq, taup, d, e (4 n), work (4 n)
PCT_GD int solve_doubles(int n) {
  const int M = n * (n - 1) / 2 + 1;
  return M * n + M + n * n + 8 * n;
}
constexpr double ILL_BAND = 1e3;  // the notice: a singular value within this factor of the rank cut (as STAB_ILL_BAND)

// The split of a stack at (s0, s1) over k >= 3 supporters with contact centres in[2 i], in[2 i + 1] (D/space.py:134-163): the
// system (one row per supporter pair i < j, a closing row of ones, right-hand side e_M) is built in ws and solved as dgelsd
// solves it, by the lanes of g (all of them call, with the same arguments).  x[0..k): the fractions.  `ill`: a singular value
// within 1e3 of the rank cut.  Returns false when dbdsqr does not converge (NumPy raises LinAlgError there; x is zero then).
// (dot2: pct_stab.cuh's stab_dot2, np.dot of two 2-vectors on an FMA host, passed in to keep it the single definition)
template <typename Dot2>
PCT_GD bool split_t(Grp g, double* ws, int k, const double* in, double s0, double s1, Dot2 dot2, double* x, bool& ill, bool avx2 = false) {
  const int n = k, M = k * (k - 1) / 2 + 1, lda = M;
  double* a = ws;
  double* b = a + M * n;
  double* vt = b + M;
  double* tau = vt + n * n;   // (spare: the column reflectors are applied where they are made)
  double* taup = tau + n;
  double* d = taup + n;
  double* e = d + n;
  double* work = e + n;       // 4 n
  PCT_GPROF_T0(tp0)
  PCT_GSYNC();
  for (int i = g.gl; i < M * n + M; i += g.G) a[i] = 0.0;  // (A and b are adjacent)
  PCT_GSYNC();
  {
    // a pair row per lane: row (i, j), i < j, in the order of the double loop
    for (int row = g.gl; row < M - 1; row += g.G) {
      int i = 0, base = 0;
      while (row >= base + (k - 1 - i)) { base += k - 1 - i; i++; }
      const int j = i + 1 + (row - base);
      const double ei0 = in[2 * i], ei1 = in[2 * i + 1], ej0 = in[2 * j], ej1 = in[2 * j + 1];
      const double t0 = ei0 - ej0, t1 = ei1 - ej1;
      const double mol = dot2(s0 - ei0, s1 - ei1, t0, t1);
      if (mol != 0) {
        const double rr = fabs(dot2(s0 - ej0, s1 - ej1, t0, t1)) / mol;
        a[row + i * lda] = 1.0;
        a[row + j * lda] = -rr;
      }
    }
    for (int j = g.gl; j < k; j += g.G) { a[(M - 1) + j * lda] = 1.0; x[j] = 0.0; }
    if (g.gl == 0) b[M - 1] = 1.0;
  }
  PCT_GSYNC();
  ill = false;
  PCT_GPROF_ADD(0, tp0)
  // (dgelsd scales A and b only when an entry leaves [1e-292, 1e292]; the closing row of ones keeps max|A| >= 1)
  // Stage 0, dgeqr2 + dorm2r: A = Q R by column reflectors, each applied to the trailing columns and -- dorm2r('L', 'T') applies
  // the same reflectors to b in the same order, and b meets nothing else in between -- to b at once.  Stage 1, dgebd2 + dormbr('Q'):
  // R (the top n x n) to bidiagonal form by a column reflector (again applied to b at once) and a row reflector per step; the row
  // reflectors (taup) meet b only after the bidiagonal solve.
  PCT_GNOUNROLL
  for (int stage = 0; stage < 2; stage++) {
    PCT_GPROF_T0(tst)
    const int rows = stage == 0 ? M : n;
    PCT_GNOUNROLL
    for (int i = 0; i < n; i++) {
      const int nparts = (stage == 1 && i < n - 1) ? 2 : 1;
      PCT_GNOUNROLL
      for (int part = 0; part < nparts; part++) {
        const int len = part == 0 ? rows - i : n - i - 1;
        const int inc = part == 0 ? 1 : lda;
        double* alpha = part == 0 ? &a[i + i * lda] : &a[i + (i + 1) * lda];
        double* xv = part == 0 ? &a[(i + 1 < rows ? i + 1 : rows - 1) + i * lda] : &a[i + (i + 2 < n ? i + 2 : n - 1) * lda];
        double saved;
        const double tauv = dlarfg(g, len, alpha, xv, inc, saved, vt);  // (dnrm2's hand-over words: V^T and tau, n (n + 1) doubles, idle until dlalsd)
        if (stage == 1) {
          if (part == 0) d[i] = saved;
          else { e[i] = saved; taup[i] = tauv; }
        }
        if (tauv != 0.0) {
          *alpha = 1.0;
          if (part == 0) dlarf_left(g, len, n - i - 1, alpha, 1, tauv, &a[i + (i + 1) * lda], lda, &b[i], avx2);
          else dlarf_right(g, n - i - 1, n - i - 1, alpha, lda, tauv, &a[(i + 1) + (i + 1) * lda], lda, avx2);
        }
        *alpha = saved;
      }
    }
    if (stage == 0) {
      PCT_GSYNC();
      for (int q = g.gl; q < n * n; q += g.G) {
        const int j = q / n, i = q - j * n;
        if (i > j) a[i + j * lda] = 0.0;
      }
      PCT_GSYNC();
    }
    PCT_GPROF_ADD(1 + stage, tst)
  }
  taup[n - 1] = 0.0;
  PCT_GPROF_T0(tbd)
  // dlalsd('U', 25, n, 1, d, e, b, ...)
  double orgnrm = 0.0;
  for (int i = 0; i < n; i++) orgnrm = fabs(d[i]) > orgnrm ? fabs(d[i]) : orgnrm;
  for (int i = 0; i < n - 1; i++) orgnrm = fabs(e[i]) > orgnrm ? fabs(e[i]) : orgnrm;
  if (orgnrm == 0.0) return true;
  dlascl_vec(g, orgnrm, 1.0, n, d);
  dlascl_vec(g, orgnrm, 1.0, n - 1, e);
  for (int q = g.gl; q < n * n; q += g.G) vt[q] = (q / n == q % n) ? 1.0 : 0.0;
  PCT_GSYNC();
  if (!dbdsqr(g, n, d, e, vt, b, work)) { ill = true; return false; }
  PCT_GPROF_ADD(3, tbd)
  PCT_GPROF_T0(tfin)
  // dlasdq: into increasing order
  for (int i = 1; i <= n; i++) {
    int isub = i;
    double smn = d[i - 1];
    for (int j = i + 1; j <= n; j++)
      if (d[j - 1] < smn) { isub = j; smn = d[j - 1]; }
    if (isub != i) {
      d[isub - 1] = d[i - 1];
      d[i - 1] = smn;
      PCT_GSYNC();
      for (int c = g.gl; c < n + 1; c += g.G) {
        double* col = c < n ? vt + c * n : b;
        const double t = col[isub - 1];
        col[isub - 1] = col[i - 1];
        col[i - 1] = t;
      }
      PCT_GSYNC();
    }
  }
  int imax = 0;
  for (int i = 1; i < n; i++)
    if (fabs(d[i]) > fabs(d[imax])) imax = i;
  const double rcond = 2.220446049250313e-16 * (double)(M > n ? M : n);
  const double tol = rcond * fabs(d[imax]);
  for (int i = 0; i < n; i++)
    if (d[i] > 0.0 && d[i] > tol / ILL_BAND && d[i] < tol * ILL_BAND) ill = true;
  // b(i) = 0 below the cut, else dlascl(d(i) -> 1) of the one element; then dgemm('T', 'N', n, 1, n): the packed kernel, acc =
  // fma(vt(k, i), b(k), acc) in k order (Haswell kernel: rows in groups of four with four accumulators over the leading blocks
  // of eight k, the tail into the first, (q0 + q1) + (q2 + q3); its n mod 4 last rows, like every row of the SkylakeX kernel, one
  // chain); a row per lane
  {
    const Grp one = {0, 1};
    for (int i = g.gl; i < n; i += g.G) {
      if (d[i] <= tol) b[i] = 0.0;
      else dlascl_vec(one, d[i], 1.0, 1, &b[i]);
    }
  }
  PCT_GSYNC();
  for (int i = g.gl; i < n; i += g.G) {
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    const bool four = avx2 && i < (n & ~3);
    const int kb = four ? (n & ~7) : 0;
    for (int kk = 0; kk < kb; kk += 4) {
      q0 = fma(vt[kk + i * n], b[kk], q0);
      q1 = fma(vt[kk + 1 + i * n], b[kk + 1], q1);
      q2 = fma(vt[kk + 2 + i * n], b[kk + 2], q2);
      q3 = fma(vt[kk + 3 + i * n], b[kk + 3], q3);
    }
    for (int kk = kb; kk < n; kk++) q0 = fma(vt[kk + i * n], b[kk], q0);
    work[i] = four ? (q0 + q1) + (q2 + q3) : q0;
  }
  PCT_GSYNC();
  for (int i = g.gl; i < n; i += g.G) b[i] = work[i];
  dlascl_vec(g, orgnrm, 1.0, n, b);
  // dormbr('P', 'L', 'N') = dorml2('L', 'T'): the row reflectors, last first
  for (int i = n - 2; i >= 0; i--) {
    const double aii = a[i + (i + 1) * lda];
    const double tp = taup[i];
    if (tp != 0.0) {
      a[i + (i + 1) * lda] = 1.0;
      dlarf_left(g, n - 1 - i, 0, &a[i + (i + 1) * lda], lda, tp, &b[i + 1], M, &b[i + 1], avx2);
      a[i + (i + 1) * lda] = aii;
    }
  }
  PCT_GSYNC();
  for (int j = g.gl; j < n; j += g.G) x[j] = b[j];
  PCT_GSYNC();
  PCT_GPROF_ADD(4, tfin)
  return true;
}
// one lane (the round-4 signature: tests/host and the host build of pct_stab.cuh)
template <typename Dot2>
PCT_GD bool split_t(double* ws, int k, const double* in, double s0, double s1, Dot2 dot2, double* x, bool& ill, bool avx2 = false) {
  const Grp one = {0, 1};
  return split_t(one, ws, k, in, s0, s1, dot2, x, ill, avx2);
}

}  // namespace gelsd
}  // namespace pct
#endif
