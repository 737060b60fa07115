/*
 * pct_oracle_gelsd.c -- np.linalg.lstsq as the reference's NumPy executes it.  TEST INFRASTRUCTURE (see pct_oracle.h).
 *
 * The reference splits a stack over >= 3 supporters with np.linalg.lstsq(coefficient, value, rcond=None)
 * (D/space.py:134-163, :236-259; C/space.py:130-159, :232-255).  NumPy hands that to LAPACK dgelsd; the solver is NOT in
 * /root/reference: it is the OpenBLAS 0.3.29 (reference LAPACK 3.11 inside) that the NumPy 2.2.6 wheel bundles
 * (numpy.libs/libscipy_openblas64_*.so).  This file restates, operation for operation, the path dgelsd takes for the
 * systems the stability check builds (M = k (k - 1) / 2 + 1 rows, N = k >= 3 columns, one right-hand side, so always
 * M >= 1.6 N and N <= 25):
 *
 *   dgelsd:  dgeqr2 (QR, unblocked since N < 32) -> dorm2r (Q^T b) -> dgebd2 (bidiagonalise R) -> dorm2r (Q1^T b)
 *            -> dlalsd: scale by 1 / max|d,e|, dlasdq = dbdsqr with vectors (implicit zero-shift / shifted QR sweeps,
 *               dlasv2 / dlas2 / dlartg / dlasr) + selection sort, rank cut at rcond * max(s), divide, V * (.), unscale
 *            -> dorml2 (P b).
 *
 * The LAPACK layer is compiled Fortran without FMA contraction (plain IEEE double operations in program order).  The
 * BLAS layer is OpenBLAS' hand-written kernels, whose summation orders and fused multiply-adds differ per CPU family
 * (the reference's stability verdicts on tie cases differ between machines for that reason: profiles/r04_lstsq_ondomain.txt).
 * Restated here is the kernel set OpenBLAS dispatches to on AVX-512 hosts ("SkylakeX" -- the build container; it reuses
 * the Haswell dgemv_t / daxpy / dger kernels), read from the disassembly of the bundled library:
 *   dnrm2   x87: squares and sums in 80-bit extended precision, four accumulators over blocks of eight
 *           (A: i = 0 mod 4 ... D: i = 3 mod 4), the n mod 8 tail into A, ((C + A) + B) + D, fsqrt, ONE rounding to double
 *   dgemv T rows in groups of four: columns in groups of four by the AVX2 kernel (per column four lanes of FMA chains,
 *           (l0 + l2) + (l1 + l3)), then two columns by an SSE2 kernel (two lanes, multiply and add rounded separately,
 *           l0 + l1), then one column by another SSE2 kernel (lanes (r0, r1) and (r2, r3) of each group of four,
 *           (r0 + r2) + (r1 + r3)); y = fma(alpha, t, y) (4x4 columns with unit increment: y + t * alpha).  The m mod 4
 *           tail rows: t = a1 x1; t = fma(a0, x0, t); t = fma(a2, x2, t); y = y + t   (x pre-multiplied by alpha)
 *   dgemv N rows in groups of four: columns in groups of four (t = a1 x1; fma a0 x0; fma a2 x2; fma a3 x3;
 *           y = fma(alpha, t, y)), then (unit-stride x only) a pair (t = a1 x1; fma a0 x0; y = fma(alpha, t, y)), single
 *           columns y = y + a (x alpha) rounded separately; tail rows: t = fma(a, x, t) over all columns, y = fma(alpha, t, y)
 *   dger    per column: t = alpha y_j;  a_ij = fma(t, x_i, a_ij)
 *   drot    x' = fma(c, x, s y);  y' = fma(c, y, -(s x))
 *   dscal   x' = alpha x           dgemm 'T','N' (n x n by n x 1): the packed kernel, acc = fma(a_k, b_k, acc) in k order
 * Every routine below was checked bit for bit against the library's own routine (tests/golden/check_gelsd_port.py calls
 * them through ctypes on random and on recorded systems); the whole solve is checked against np.linalg.lstsq there and
 * -- build-container independent -- against the committed vectors tests/golden/lstsq_systems.npz (tests/test_gelsd_port.py).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "pct_oracle_internal.h"

#define A_(i, j) a[(size_t)(i) + (size_t)(j) * (size_t)lda]

/* Which of OpenBLAS' kernel sets is restated: 0 = "SkylakeX" (AVX-512 hosts; the build container, every fixture), 1 = "Haswell" (what
 * OpenBLAS runs on AVX2 hosts, AMD Zen included: OPENBLAS_CORETYPE=HASWELL / ZEN).  The two differ in dgemv 'N' and in daxpy (dger);
 * with them np.linalg.lstsq returns other last bits on 98 % of the stability systems (tests/golden/check_gelsd_port.py), and a run of
 * the reference takes another path at a tie about once in 10^5 steps (profiles/r04_lstsq_ondomain.txt). */
static int g_kernel_set = 0;
void gelsd_set_kernel_set(int s) { g_kernel_set = s; }
int gelsd_get_kernel_set(void) { return g_kernel_set; }

/* ---- BLAS, as the SkylakeX kernel set computes it --------------------------------------------------------- */
double gelsd_dnrm2(int n, const double* x, int incx) {
  if (n <= 0) return 0.0;
  long double acc[4] = {0.0L, 0.0L, 0.0L, 0.0L};
  const int n8 = n & ~7;
  int i = 0;
  for (; i < n8; i++) {
    const long double v = (long double)x[(size_t)i * incx];
    acc[i & 3] += v * v;
  }
  for (; i < n; i++) {
    const long double v = (long double)x[(size_t)i * incx];
    acc[0] += v * v;
  }
  long double s = ((acc[2] + acc[0]) + acc[1]) + acc[3];
  return (double)sqrtl(s);
}
static void skx_dscal(int n, double alpha, double* x, int incx) {
  for (int i = 0; i < n; i++) x[(size_t)i * incx] = alpha * x[(size_t)i * incx];
}
/* y[0..n) = alpha * A^T x  (y overwritten: the callers pass beta = 0), A m x n, x with increment incx, y unit */
void gelsd_dgemv_t(int m, int n, double alpha, const double* a, int lda, const double* x, int incx, double* y) {
  for (int j = 0; j < n; j++) y[j] = 0.0;
  if (m <= 0 || n <= 0 || alpha == 0.0) return;
  const int m2 = m & ~3, m3 = m & 3, n4 = n & ~3;
  if (m2 > 0) {
    for (int j = 0; j < n; j++) {
      const double* c = &A_(0, j);
      double t;
      if (j < n4) { /* dgemv_kernel_4x4: AVX2 FMA, one lane per row residue */
        double l[4] = {0, 0, 0, 0};
        for (int i = 0; i < m2; i++) l[i & 3] = fma(c[i], x[(size_t)i * incx], l[i & 3]);
        t = (l[0] + l[2]) + (l[1] + l[3]);
        y[j] = y[j] + t * alpha; /* add_y, unit increment: mulpd, addpd */
      } else if ((n & 2) && j < n4 + 2) { /* dgemv_kernel_4x2: SSE2, two lanes, no FMA */
        double l[2] = {0, 0};
        for (int i = 0; i < m2; i++) l[i & 1] = l[i & 1] + c[i] * x[(size_t)i * incx];
        t = l[0] + l[1];
        y[j] = fma(alpha, t, y[j]);
      } else { /* dgemv_kernel_4x1: SSE2, lanes (r0, r1) | (r2, r3) */
        double l[4] = {0, 0, 0, 0};
        for (int i = 0; i < m2; i++) l[i & 3] = l[i & 3] + c[i] * x[(size_t)i * incx];
        t = (l[0] + l[2]) + (l[1] + l[3]);
        y[j] = fma(alpha, t, y[j]);
      }
    }
  }
  if (m3) {
    const double x0 = x[(size_t)m2 * incx] * alpha;
    const double x1 = m3 > 1 ? x[(size_t)(m2 + 1) * incx] * alpha : 0.0;
    const double x2 = m3 > 2 ? x[(size_t)(m2 + 2) * incx] * alpha : 0.0;
    for (int j = 0; j < n; j++) {
      const double* c = &A_(m2, j);
      if (m3 == 1) y[j] = fma(c[0], x0, y[j]);
      else {
        double t = c[1] * x1;
        t = fma(c[0], x0, t);
        if (m3 == 3) t = fma(c[2], x2, t);
        y[j] = y[j] + t;
      }
    }
  }
}
/* y[0..m) = alpha * A x  (y overwritten), A m x n, x with increment incx, y unit */
void gelsd_dgemv_n(int m, int n, double alpha, const double* a, int lda, const double* x, int incx, double* y) {
  for (int i = 0; i < m; i++) y[i] = 0.0;
  if (m <= 0 || n <= 0 || alpha == 0.0) return;
  const int m2 = m & ~3, n4 = n & ~3;
  if (g_kernel_set == 1) {
    /* Haswell: the 4x4 microkernel sums two FMA chains, (a0 x0 + a2 x2) + (a1 x1 + a3 x3); the pair kernel (unit stride) a0 x0 + a1 x1;
     * y = fma(t, alpha, y); single columns as below; the C tails are compiled WITHOUT FMA */
    for (int i = 0; i < m2; i++) {
      double yi = 0.0;
      for (int j = 0; j < n4; j += 4) {
        double t4 = A_(i, j) * x[(size_t)j * incx], t5 = A_(i, j + 1) * x[(size_t)(j + 1) * incx];
        t4 = fma(A_(i, j + 2), x[(size_t)(j + 2) * incx], t4);
        t5 = fma(A_(i, j + 3), x[(size_t)(j + 3) * incx], t5);
        yi = fma(t4 + t5, alpha, yi);
      }
      int j = n4;
      if (incx == 1 && (n & 2)) {
        const double t4 = A_(i, j) * x[j], t5 = A_(i, j + 1) * x[j + 1];
        yi = fma(t4 + t5, alpha, yi);
        j += 2;
      }
      for (; j < n; j++) yi = yi + A_(i, j) * (x[(size_t)j * incx] * alpha);
      y[i] = yi;
    }
    for (int i = m2; i < m; i++) {
      double t = 0.0;
      for (int j = 0; j < n; j++) t = t + A_(i, j) * x[(size_t)j * incx];
      y[i] = 0.0 + alpha * t;
    }
    return;
  }
  for (int i = 0; i < m2; i++) {
    double yi = 0.0;
    for (int j = 0; j < n4; j += 4) {
      double t = A_(i, j + 1) * x[(size_t)(j + 1) * incx];
      t = fma(A_(i, j), x[(size_t)j * incx], t);
      t = fma(A_(i, j + 2), x[(size_t)(j + 2) * incx], t);
      t = fma(A_(i, j + 3), x[(size_t)(j + 3) * incx], t);
      yi = fma(alpha, t, yi);
    }
    int j = n4;
    if (incx == 1 && (n & 2)) {
      double t = A_(i, j + 1) * x[j + 1];
      t = fma(A_(i, j), x[j], t);
      yi = fma(alpha, t, yi);
      j += 2;
    }
    for (; j < n; j++) yi = yi + A_(i, j) * (x[(size_t)j * incx] * alpha);
    y[i] = yi;
  }
  for (int i = m2; i < m; i++) {
    double t = 0.0;
    for (int j = 0; j < n; j++) t = fma(A_(i, j), x[(size_t)j * incx], t);
    y[i] = fma(alpha, t, 0.0);
  }
}
/* A += alpha x y^T */
void gelsd_dger(int m, int n, double alpha, const double* x, int incx, const double* y, int incy, double* a, int lda) {
  if (m <= 0 || n <= 0 || alpha == 0.0) return;
  for (int j = 0; j < n; j++) {
    const double t = alpha * y[(size_t)j * incy];
    /* Haswell daxpy: blocks of sixteen through the FMA microkernel, the rest multiply and add rounded separately */
    const int mf = g_kernel_set == 1 ? (m & ~15) : m;
    for (int i = 0; i < mf; i++) A_(i, j) = fma(t, x[(size_t)i * incx], A_(i, j));
    for (int i = mf; i < m; i++) A_(i, j) = A_(i, j) + t * x[(size_t)i * incx];
  }
}
void gelsd_drot(int n, double* x, int incx, double* y, int incy, double c, double s) {
  for (int i = 0; i < n; i++) {
    const double xv = x[(size_t)i * incx], yv = y[(size_t)i * incy];
    x[(size_t)i * incx] = fma(c, xv, s * yv);
    y[(size_t)i * incy] = fma(c, yv, -(s * xv));
  }
}

/* ---- LAPACK 3.11 auxiliaries (plain double arithmetic, program order) ---------------------------------------- */
#define EPS_ 1.1102230246251565e-16   /* dlamch('E') = 2^-53 */
#define SAFMIN_ 2.2250738585072014e-308 /* dlamch('S') */
static double sign_(double a, double b) { return signbit(b) ? -fabs(a) : fabs(a); }

double gelsd_dlapy2(double x, double y) {
  const double xa = fabs(x), ya = fabs(y);
  const double w = xa > ya ? xa : ya, z = xa < ya ? xa : ya;
  if (z == 0.0 || w > DBL_MAX) return w;
  const double q = z / w;
  return w * sqrt(1.0 + q * q);
}
/* dlarfg: elementary reflector H = I - tau (1; v)(1; v)^T with H (alpha; x) = (beta; 0) */
void gelsd_dlarfg(int n, double* alpha, double* x, int incx, double* tau) {
  if (n <= 1) { *tau = 0.0; return; }
  double xnorm = gelsd_dnrm2(n - 1, x, incx);
  if (xnorm == 0.0) { *tau = 0.0; return; }
  double beta = -sign_(gelsd_dlapy2(*alpha, xnorm), *alpha);
  const double safmin = SAFMIN_ / EPS_, rsafmn = 1.0 / safmin;
  int knt = 0;
  if (fabs(beta) < safmin) {
    do {
      knt++;
      skx_dscal(n - 1, rsafmn, x, incx);
      beta *= rsafmn;
      *alpha *= rsafmn;
    } while (fabs(beta) < safmin && knt < 20);
    xnorm = gelsd_dnrm2(n - 1, x, incx);
    beta = -sign_(gelsd_dlapy2(*alpha, xnorm), *alpha);
  }
  *tau = (beta - *alpha) / beta;
  skx_dscal(n - 1, 1.0 / (*alpha - beta), x, incx);
  for (int j = 0; j < knt; j++) beta *= safmin;
  *alpha = beta;
}
/* dlarf: C := H C (left) or C H (right), H = I - tau v v^T; trailing zeros of v and zero columns / rows of C skipped */
void gelsd_dlarf(int left, int m, int n, const double* v, int incv, double tau, double* a, int lda, double* work) {
  int lastv = 0, lastc = 0;
  if (tau != 0.0) {
    lastv = left ? m : n;
    while (lastv > 0 && v[(size_t)(lastv - 1) * incv] == 0.0) lastv--;
    if (left) { /* iladlc: last non-zero column of C(0:lastv, :) */
      for (lastc = n; lastc > 0; lastc--) {
        int nz = 0;
        for (int i = 0; i < lastv; i++) if (A_(i, lastc - 1) != 0.0) { nz = 1; break; }
        if (nz) break;
      }
    } else { /* iladlr: last non-zero row of C(:, 0:lastv) */
      for (lastc = m; lastc > 0; lastc--) {
        int nz = 0;
        for (int j = 0; j < lastv; j++) if (A_(lastc - 1, j) != 0.0) { nz = 1; break; }
        if (nz) break;
      }
    }
  }
  if (lastv <= 0) return;
  if (left) {
    gelsd_dgemv_t(lastv, lastc, 1.0, a, lda, v, incv, work);
    gelsd_dger(lastv, lastc, -tau, v, incv, work, 1, a, lda);
  } else {
    gelsd_dgemv_n(lastc, lastv, 1.0, a, lda, v, incv, work);
    gelsd_dger(lastc, lastv, -tau, work, 1, v, incv, a, lda);
  }
}
/* dlartg (LAPACK 3.10.1+, la_xisnan-free Fortran 90 version) */
void gelsd_dlartg(double f, double g, double* c, double* s, double* r) {
  const double safmin = SAFMIN_, safmax = 1.0 / safmin;
  const double rtmin = sqrt(safmin), rtmax = sqrt(safmax / 2);
  const double f1 = fabs(f), g1 = fabs(g);
  if (g == 0.0) { *c = 1.0; *s = 0.0; *r = f; }
  else if (f == 0.0) { *c = 0.0; *s = sign_(1.0, g); *r = g1; }
  else if (f1 > rtmin && f1 < rtmax && g1 > rtmin && g1 < rtmax) {
    const double d = sqrt(f * f + g * g);
    *c = f1 / d;
    *r = sign_(d, f);
    *s = g / *r;
  } else {
    double u = f1 > g1 ? f1 : g1;
    if (u < safmin) u = safmin;
    if (u > safmax) u = safmax;
    const double fs = f / u, gs = g / u;
    const double d = sqrt(fs * fs + gs * gs);
    *c = fabs(fs) / d;
    *r = sign_(d, f);
    *s = gs / *r;
    *r = *r * u;
  }
}
/* dlas2: singular values of [f g; 0 h] */
void gelsd_dlas2(double f, double g, double h, double* ssmin, double* ssmax) {
  const double fa = fabs(f), ga = fabs(g), ha = fabs(h);
  const double fhmn = fa < ha ? fa : ha, fhmx = fa > ha ? fa : ha;
  if (fhmn == 0.0) {
    *ssmin = 0.0;
    if (fhmx == 0.0) *ssmax = ga;
    else {
      const double mx = fhmx > ga ? fhmx : ga, mn = fhmx < ga ? fhmx : ga;
      const double q = mn / mx;
      *ssmax = mx * sqrt(1.0 + q * q);
    }
  } else if (ga < fhmx) {
    const double as = 1.0 + fhmn / fhmx, at = (fhmx - fhmn) / fhmx;
    const double q = ga / fhmx, au = q * q;
    const double c = 2.0 / (sqrt(as * as + au) + sqrt(at * at + au));
    *ssmin = fhmn * c;
    *ssmax = fhmx / c;
  } else {
    const double au = fhmx / ga;
    if (au == 0.0) { *ssmin = (fhmn * fhmx) / ga; *ssmax = ga; }
    else {
      const double as = 1.0 + fhmn / fhmx, at = (fhmx - fhmn) / fhmx;
      const double p = as * au, q = at * au;
      const double c = 1.0 / (sqrt(1.0 + p * p) + sqrt(1.0 + q * q));
      *ssmin = (fhmn * c) * au;
      *ssmin = *ssmin + *ssmin;
      *ssmax = ga / (c + c);
    }
  }
}
/* dlasv2: SVD of [f g; 0 h] */
void gelsd_dlasv2(double f, double g, double h, double* ssmin, double* ssmax, double* snr, double* csr, double* snl,
                  double* csl) {
  double ft = f, fa = fabs(ft), ht = h, ha = fabs(h);
  int pmax = 1;
  const int swap = ha > fa;
  if (swap) { pmax = 3; double t = ft; ft = ht; ht = t; t = fa; fa = ha; ha = t; }
  const double gt = g, ga = fabs(gt);
  double clt, crt, slt, srt;
  if (ga == 0.0) { *ssmin = ha; *ssmax = fa; clt = 1.0; crt = 1.0; slt = 0.0; srt = 0.0; }
  else {
    int gasmal = 1;
    if (ga > fa) {
      pmax = 2;
      if (fa / ga < EPS_) {
        gasmal = 0;
        *ssmax = ga;
        if (ha > 1.0) *ssmin = fa / (ga / ha); else *ssmin = (fa / ga) * ha;
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
      *ssmin = ha / a;
      *ssmax = fa * a;
      if (mm == 0.0) {
        if (l == 0.0) t = sign_(2.0, ft) * sign_(1.0, gt);
        else t = gt / sign_(d, ft) + m / t;
      } else t = (m / (s + t) + m / (r + l)) * (1.0 + a);
      l = sqrt(t * t + 4.0);
      crt = 2.0 / l;
      srt = t / l;
      clt = (crt + srt * m) / a;
      slt = (ht / ft) * srt / a;
    }
  }
  if (swap) { *csl = srt; *snl = crt; *csr = slt; *snr = clt; }
  else { *csl = clt; *snl = slt; *csr = crt; *snr = srt; }
  double tsign;
  if (pmax == 1) tsign = sign_(1.0, *csr) * sign_(1.0, *csl) * sign_(1.0, f);
  else if (pmax == 2) tsign = sign_(1.0, *snr) * sign_(1.0, *csl) * sign_(1.0, g);
  else tsign = sign_(1.0, *snr) * sign_(1.0, *snl) * sign_(1.0, h);
  *ssmax = sign_(*ssmax, tsign);
  *ssmin = sign_(*ssmin, tsign * sign_(1.0, f) * sign_(1.0, h));
}
/* dlasr('L', 'V', forward ? 'F' : 'B'): plane rotations (c[j], s[j]) in planes (j, j + 1) applied from the left to the
 * m x n matrix A */
static void dlasr_lv(int forward, int m, int n, const double* c, const double* s, double* a, int lda) {
  if (m <= 0 || n <= 0) return;
  for (int jj = 0; jj < m - 1; jj++) {
    const int j = forward ? jj : m - 2 - jj;
    const double ct = c[j], st = s[j];
    if (ct != 1.0 || st != 0.0)
      for (int i = 0; i < n; i++) {
        const double temp = A_(j + 1, i);
        A_(j + 1, i) = ct * temp - st * A_(j, i);
        A_(j, i) = st * temp + ct * A_(j, i);
      }
  }
}
/* dlascl('G', ., ., cfrom, cto, ...) on a vector: the multiplier sequence */
static void dlascl_vec(double cfrom, double cto, int n, double* x, int incx) {
  const double smlnum = SAFMIN_, bignum = 1.0 / smlnum;
  double cfromc = cfrom, ctoc = cto, mul;
  int done;
  do {
    const double cfrom1 = cfromc * smlnum;
    if (cfrom1 == cfromc) { mul = ctoc / cfromc; done = 1; }
    else {
      const double cto1 = ctoc / bignum;
      if (cto1 == ctoc) { mul = ctoc; done = 1; cfromc = 1.0; }
      else if (fabs(cfrom1) > fabs(ctoc) && ctoc != 0.0) { mul = smlnum; done = 0; cfromc = cfrom1; }
      else if (fabs(cto1) > fabs(cfromc)) { mul = bignum; done = 0; ctoc = cto1; }
      else { mul = ctoc / cfromc; done = 1; if (mul == 1.0) return; }
    }
    for (int i = 0; i < n; i++) x[(size_t)i * incx] = x[(size_t)i * incx] * mul;
  } while (!done);
}

/* dbdsqr('U', n, ncvt, 0, ncc = 1, d, e, VT, ldvt, -, -, C, ldc, work): SVD of the upper bidiagonal (d, e); the right
 * rotations accumulate into VT (n x ncvt), the left ones into C (n x 1).  Returns info. */
int gelsd_dbdsqr(int n, int ncvt, double* d, double* e, double* vt, int ldvt, double* cc, double* work) {
  const double meigth = -0.125, hndrth = 0.01;
  const int maxitr = 6;
  if (n == 0) return 0;
  const int nm1 = n - 1, nm12 = nm1 + nm1, nm13 = nm12 + nm1;
  int idir = 0;
  const double eps = EPS_, unfl = SAFMIN_;
  if (n > 1) {
    double tolmul = pow(eps, meigth);
    if (tolmul > 100.0) tolmul = 100.0;
    if (tolmul < 10.0) tolmul = 10.0;
    const double tol = tolmul * eps;
    double smax = 0.0;
    for (int i = 0; i < n; i++) if (fabs(d[i]) > smax) smax = fabs(d[i]);
    for (int i = 0; i < n - 1; i++) if (fabs(e[i]) > smax) smax = fabs(e[i]);
    double smin = 0.0;
    double sminoa = fabs(d[0]);
    if (sminoa != 0.0) {
      double mu = sminoa;
      for (int i = 1; i < n; i++) {
        mu = fabs(d[i]) * (mu / (mu + fabs(e[i - 1])));
        if (mu < sminoa) sminoa = mu;
        if (sminoa == 0.0) break;
      }
    }
    sminoa = sminoa / sqrt((double)n);
    double thresh = tol * sminoa;
    { const double t2 = maxitr * (n * (n * unfl)); if (t2 > thresh) thresh = t2; }
    const int maxitdivn = maxitr * n;
    int iterdivn = 0, iter = -1, oldll = -1, oldm = -1;
    int m = n; /* 1-based index of the bottom of the active block */
    for (;;) {
      if (m <= 1) break;
      if (iter >= n) { iter -= n; iterdivn++; if (iterdivn >= maxitdivn) return 1; }
      /* find the diagonal block to work on */
      smax = fabs(d[m - 1]);
      int ll = 0, split = 0;
      for (int lll = 1; lll <= m - 1; lll++) {
        ll = m - lll;
        const double abss = fabs(d[ll - 1]), abse = fabs(e[ll - 1]);
        if (abse <= thresh) { split = 1; break; }
        if (abss > smax) smax = abss;
        if (abse > smax) smax = abse;
      }
      if (split) {
        e[ll - 1] = 0.0;
        if (ll == m - 1) { m = m - 1; continue; }
      } else ll = 0;
      ll = ll + 1;
      /* e(ll) .. e(m-1) are non-zero */
      if (ll == m - 1) { /* 2 x 2 block */
        double sigmn, sigmx, sinr, cosr, sinl, cosl;
        gelsd_dlasv2(d[m - 2], e[m - 2], d[m - 1], &sigmn, &sigmx, &sinr, &cosr, &sinl, &cosl);
        d[m - 2] = sigmx; e[m - 2] = 0.0; d[m - 1] = sigmn;
        if (ncvt > 0) gelsd_drot(ncvt, &vt[m - 2], ldvt, &vt[m - 1], ldvt, cosr, sinr);
        gelsd_drot(1, &cc[m - 2], 1, &cc[m - 1], 1, cosl, sinl);
        m = m - 2;
        continue;
      }
      if (ll > oldm || m < oldll) idir = fabs(d[ll - 1]) >= fabs(d[m - 1]) ? 1 : 2;
      /* convergence tests */
      int conv = 0;
      if (idir == 1) {
        if (fabs(e[m - 2]) <= fabs(tol) * fabs(d[m - 1])) { e[m - 2] = 0.0; continue; }
        double mu = fabs(d[ll - 1]);
        smin = mu;
        for (int lll = ll; lll <= m - 1; lll++) {
          if (fabs(e[lll - 1]) <= tol * mu) { e[lll - 1] = 0.0; conv = 1; break; }
          mu = fabs(d[lll]) * (mu / (mu + fabs(e[lll - 1])));
          if (mu < smin) smin = mu;
        }
      } else {
        if (fabs(e[ll - 1]) <= fabs(tol) * fabs(d[ll - 1])) { e[ll - 1] = 0.0; continue; }
        double mu = fabs(d[m - 1]);
        smin = mu;
        for (int lll = m - 1; lll >= ll; lll--) {
          if (fabs(e[lll - 1]) <= tol * mu) { e[lll - 1] = 0.0; conv = 1; break; }
          mu = fabs(d[lll - 1]) * (mu / (mu + fabs(e[lll - 1])));
          if (mu < smin) smin = mu;
        }
      }
      if (conv) continue;
      oldll = ll; oldm = m;
      /* shift */
      double shift, r;
      {
        const double bound = eps > hndrth * tol ? eps : hndrth * tol;
        if (n * tol * (smin / smax) <= bound) shift = 0.0;
        else {
          double sll;
          if (idir == 1) { sll = fabs(d[ll - 1]); gelsd_dlas2(d[m - 2], e[m - 2], d[m - 1], &shift, &r); }
          else { sll = fabs(d[m - 1]); gelsd_dlas2(d[ll - 1], e[ll - 1], d[ll], &shift, &r); }
          if (sll > 0.0) { const double q = shift / sll; if (q * q < eps) shift = 0.0; }
        }
      }
      iter = iter + m - ll;
      double* w0 = work;            /* work(1 ..)      */
      double* w1 = work + nm1;      /* work(nm1 + 1 ..) */
      double* w2 = work + nm12;
      double* w3 = work + nm13;
      const int cnt = m - ll + 1;
      if (shift == 0.0) {
        if (idir == 1) {
          double cs = 1.0, oldcs = 1.0, sn = 0.0, oldsn = 0.0;
          for (int i = ll; i <= m - 1; i++) {
            gelsd_dlartg(d[i - 1] * cs, e[i - 1], &cs, &sn, &r);
            if (i > ll) e[i - 2] = oldsn * r;
            gelsd_dlartg(oldcs * r, d[i] * sn, &oldcs, &oldsn, &d[i - 1]);
            w0[i - ll] = cs; w1[i - ll] = sn; w2[i - ll] = oldcs; w3[i - ll] = oldsn;
          }
          const double h = d[m - 1] * cs;
          d[m - 1] = h * oldcs;
          e[m - 2] = h * oldsn;
          if (ncvt > 0) dlasr_lv(1, cnt, ncvt, w0, w1, &vt[ll - 1], ldvt);
          dlasr_lv(1, cnt, 1, w2, w3, &cc[ll - 1], n);
          if (fabs(e[m - 2]) <= thresh) e[m - 2] = 0.0;
        } else {
          double cs = 1.0, oldcs = 1.0, sn = 0.0, oldsn = 0.0;
          for (int i = m; i >= ll + 1; i--) {
            gelsd_dlartg(d[i - 1] * cs, e[i - 2], &cs, &sn, &r);
            if (i < m) e[i - 1] = oldsn * r;
            gelsd_dlartg(oldcs * r, d[i - 2] * sn, &oldcs, &oldsn, &d[i - 1]);
            w0[i - ll - 1] = cs; w1[i - ll - 1] = -sn; w2[i - ll - 1] = oldcs; w3[i - ll - 1] = -oldsn;
          }
          const double h = d[ll - 1] * cs;
          d[ll - 1] = h * oldcs;
          e[ll - 1] = h * oldsn;
          if (ncvt > 0) dlasr_lv(0, cnt, ncvt, w2, w3, &vt[ll - 1], ldvt);
          dlasr_lv(0, cnt, 1, w0, w1, &cc[ll - 1], n);
          if (fabs(e[ll - 1]) <= thresh) e[ll - 1] = 0.0;
        }
      } else {
        if (idir == 1) {
          double f = (fabs(d[ll - 1]) - shift) * (sign_(1.0, d[ll - 1]) + shift / d[ll - 1]);
          double g = e[ll - 1];
          double cosr, sinr, cosl, sinl;
          for (int i = ll; i <= m - 1; i++) {
            gelsd_dlartg(f, g, &cosr, &sinr, &r);
            if (i > ll) e[i - 2] = r;
            f = cosr * d[i - 1] + sinr * e[i - 1];
            e[i - 1] = cosr * e[i - 1] - sinr * d[i - 1];
            g = sinr * d[i];
            d[i] = cosr * d[i];
            gelsd_dlartg(f, g, &cosl, &sinl, &r);
            d[i - 1] = r;
            f = cosl * e[i - 1] + sinl * d[i];
            d[i] = cosl * d[i] - sinl * e[i - 1];
            if (i < m - 1) { g = sinl * e[i]; e[i] = cosl * e[i]; }
            w0[i - ll] = cosr; w1[i - ll] = sinr; w2[i - ll] = cosl; w3[i - ll] = sinl;
          }
          e[m - 2] = f;
          if (ncvt > 0) dlasr_lv(1, cnt, ncvt, w0, w1, &vt[ll - 1], ldvt);
          dlasr_lv(1, cnt, 1, w2, w3, &cc[ll - 1], n);
          if (fabs(e[m - 2]) <= thresh) e[m - 2] = 0.0;
        } else {
          double f = (fabs(d[m - 1]) - shift) * (sign_(1.0, d[m - 1]) + shift / d[m - 1]);
          double g = e[m - 2];
          double cosr, sinr, cosl, sinl;
          for (int i = m; i >= ll + 1; i--) {
            gelsd_dlartg(f, g, &cosr, &sinr, &r);
            if (i < m) e[i - 1] = r;
            f = cosr * d[i - 1] + sinr * e[i - 2];
            e[i - 2] = cosr * e[i - 2] - sinr * d[i - 1];
            g = sinr * d[i - 2];
            d[i - 2] = cosr * d[i - 2];
            gelsd_dlartg(f, g, &cosl, &sinl, &r);
            d[i - 1] = r;
            f = cosl * e[i - 2] + sinl * d[i - 2];
            d[i - 2] = cosl * d[i - 2] - sinl * e[i - 2];
            if (i > ll + 1) { g = sinl * e[i - 3]; e[i - 3] = cosl * e[i - 3]; }
            w0[i - ll - 1] = cosr; w1[i - ll - 1] = -sinr; w2[i - ll - 1] = cosl; w3[i - ll - 1] = -sinl;
          }
          e[ll - 1] = f;
          if (fabs(e[ll - 1]) <= thresh) e[ll - 1] = 0.0;
          if (ncvt > 0) dlasr_lv(0, cnt, ncvt, w2, w3, &vt[ll - 1], ldvt);
          dlasr_lv(0, cnt, 1, w0, w1, &cc[ll - 1], n);
        }
      }
    }
  }
  /* all singular values converged: make them positive */
  for (int i = 0; i < n; i++) {
    if (d[i] == 0.0) d[i] = 0.0; /* no -0 */
    if (d[i] < 0.0) {
      d[i] = -d[i];
      if (ncvt > 0) skx_dscal(ncvt, -1.0, &vt[i], ldvt);
    }
  }
  /* sort into decreasing order: one transposition per singular value */
  for (int i = 1; i <= n - 1; i++) {
    int isub = 1;
    double smn = d[0];
    for (int j = 2; j <= n + 1 - i; j++)
      if (d[j - 1] <= smn) { isub = j; smn = d[j - 1]; }
    if (isub != n + 1 - i) {
      d[isub - 1] = d[n - i];
      d[n - i] = smn;
      for (int c = 0; c < ncvt; c++) {
        const double t = vt[(isub - 1) + (size_t)c * ldvt];
        vt[(isub - 1) + (size_t)c * ldvt] = vt[(n - i) + (size_t)c * ldvt];
        vt[(n - i) + (size_t)c * ldvt] = t;
      }
      { const double t = cc[isub - 1]; cc[isub - 1] = cc[n - i]; cc[n - i] = t; }
    }
  }
  return 0;
}

/* dlasdq('U', sqre = 0, n, ncvt = n, 0, 1, ...): dbdsqr, then a selection sort into decreasing order */
static int dlasdq_u(int n, double* d, double* e, double* vt, int ldvt, double* cc, double* work) {
  const int info = gelsd_dbdsqr(n, n, d, e, vt, ldvt, cc, work);
  if (info) return info;
  for (int i = 1; i <= n; i++) {
    int isub = i;
    double smn = d[i - 1];
    for (int j = i + 1; j <= n; j++)
      if (d[j - 1] < smn) { isub = j; smn = d[j - 1]; }
    if (isub != i) {
      d[isub - 1] = d[i - 1];
      d[i - 1] = smn;
      for (int c = 0; c < n; c++) {
        const double t = vt[(isub - 1) + (size_t)c * ldvt];
        vt[(isub - 1) + (size_t)c * ldvt] = vt[(i - 1) + (size_t)c * ldvt];
        vt[(i - 1) + (size_t)c * ldvt] = t;
      }
      { const double t = cc[isub - 1]; cc[isub - 1] = cc[i - 1]; cc[i - 1] = t; }
    }
  }
  return 0;
}

/* np.linalg.lstsq(A, b, rcond=None)[0] for a row-major M x N system with M >= 1.6 N, 2 <= N <= 25.
 * sv_out[N]: the singular values (decreasing), *rank_out: the effective rank, *near_cut: a singular value lies within a
 * factor 1e3 of the rank cut.  Returns 0, or non-zero when dbdsqr did not converge (NumPy raises LinAlgError there). */
int gelsd_lstsq(const double* Arow, const double* brow, int M, int N, double* x, int* rank_out, double* sv_out, int* near_cut) {
  const int lda = M;
  double* a = (double*)malloc(sizeof(double) * (size_t)M * N);
  double* b = (double*)malloc(sizeof(double) * (size_t)(M > N ? M : N));
  double* tau = (double*)calloc((size_t)4 * N + 8, sizeof(double));
  double* work = (double*)calloc((size_t)N * N + 8 * (size_t)N + (size_t)M + 8, sizeof(double));
  for (int i = 0; i < M; i++) {
    for (int j = 0; j < N; j++) A_(i, j) = Arow[(size_t)i * N + j];
    b[i] = brow[i];
  }
  double* tauq = tau + N; double* taup = tau + 2 * N; double* e = tau + 3 * N;
  double* d = sv_out;
  int info = 0, rank = 0;
  if (near_cut) *near_cut = 0;
  const double rcond = DBL_EPSILON * (M > N ? M : N);
  /* dgelsd: anrm / bnrm scaling is a no-op unless an entry leaves [1e-292, 1e292]; a zero matrix gives x = 0 */
  double anrm = 0.0;
  for (int i = 0; i < M * N; i++) if (fabs(a[i]) > anrm) anrm = fabs(a[i]);
  if (anrm == 0.0) { for (int j = 0; j < N; j++) { x[j] = 0.0; d[j] = 0.0; } goto done; }
  /* dgeqr2 */
  for (int i = 0; i < N; i++) {
    gelsd_dlarfg(M - i, &A_(i, i), &A_(i + 1 < M ? i + 1 : M - 1, i), 1, &tau[i]);
    if (i < N - 1) {
      const double aii = A_(i, i);
      A_(i, i) = 1.0;
      gelsd_dlarf(1, M - i, N - i - 1, &A_(i, i), 1, tau[i], &A_(i, i + 1), lda, work);
      A_(i, i) = aii;
    }
  }
  /* dorm2r('L', 'T'): b := Q^T b */
  for (int i = 0; i < N; i++) {
    const double aii = A_(i, i);
    A_(i, i) = 1.0;
    gelsd_dlarf(1, M - i, 1, &A_(i, i), 1, tau[i], &b[i], M, work);
    A_(i, i) = aii;
  }
  /* zero below R */
  for (int j = 0; j < N - 1; j++) for (int i = j + 1; i < N; i++) A_(i, j) = 0.0;
  /* dgebd2 on the N x N R */
  for (int i = 0; i < N; i++) {
    gelsd_dlarfg(N - i, &A_(i, i), &A_(i + 1 < N ? i + 1 : N - 1, i), 1, &tauq[i]);
    d[i] = A_(i, i);
    A_(i, i) = 1.0;
    if (i < N - 1) gelsd_dlarf(1, N - i, N - i - 1, &A_(i, i), 1, tauq[i], &A_(i, i + 1), lda, work);
    A_(i, i) = d[i];
    if (i < N - 1) {
      gelsd_dlarfg(N - i - 1, &A_(i, i + 1), &A_(i, i + 2 < N ? i + 2 : N - 1), lda, &taup[i]);
      e[i] = A_(i, i + 1);
      A_(i, i + 1) = 1.0;
      gelsd_dlarf(0, N - i - 1, N - i - 1, &A_(i, i + 1), lda, taup[i], &A_(i + 1, i + 1), lda, work);
      A_(i, i + 1) = e[i];
    } else taup[i] = 0.0;
  }
  /* dormbr('Q', 'L', 'T') = dorm2r over the N reflectors of the bidiagonalisation */
  for (int i = 0; i < N; i++) {
    const double aii = A_(i, i);
    A_(i, i) = 1.0;
    gelsd_dlarf(1, N - i, 1, &A_(i, i), 1, tauq[i], &b[i], M, work);
    A_(i, i) = aii;
  }
  /* dlalsd('U', smlsiz = 25, N, 1, d, e, b, ...) */
  {
    double orgnrm = 0.0;
    for (int i = 0; i < N; i++) if (fabs(d[i]) > orgnrm) orgnrm = fabs(d[i]);
    for (int i = 0; i < N - 1; i++) if (fabs(e[i]) > orgnrm) orgnrm = fabs(e[i]);
    if (orgnrm == 0.0) { for (int j = 0; j < N; j++) x[j] = 0.0; goto done; }
    dlascl_vec(orgnrm, 1.0, N, d, 1);
    dlascl_vec(orgnrm, 1.0, N - 1, e, 1);
    double* vt = work;              /* N x N, identity */
    double* wk = work + (size_t)N * N;
    for (int i = 0; i < N * N; i++) vt[i] = 0.0;
    for (int i = 0; i < N; i++) vt[i + (size_t)i * N] = 1.0;
    info = dlasdq_u(N, d, e, vt, N, b, wk + N);
    if (info) goto done;
    int imax = 0;
    for (int i = 1; i < N; i++) if (fabs(d[i]) > fabs(d[imax])) imax = i;
    const double tol = rcond * fabs(d[imax]);
    /* the notice of pct_oracle_stab.c: a singular value within a factor 1e3 of the rank cut */
    if (near_cut)
      for (int i = 0; i < N; i++) if (d[i] > 0.0 && d[i] > tol / 1e3 && d[i] < tol * 1e3) *near_cut = 1;
    for (int i = 0; i < N; i++) {
      if (d[i] <= tol) b[i] = 0.0;
      else { dlascl_vec(d[i], 1.0, 1, &b[i], 1); rank++; }
    }
    /* dgemm('T', 'N', N, 1, N, 1, VT, N, b, ., 0, wk, N) */
    for (int i = 0; i < N; i++) {
      double acc = 0.0;
      if (gelsd_get_kernel_set() == 1 && i < (N & ~3)) {
        /* Haswell dgemm kernel, rows in groups of four: four accumulators over the leading blocks of eight k (k mod 4), the tail
         * into the first, (a0 + a1) + (a2 + a3); the n mod 4 last rows one FMA chain */
        double q[4] = {0, 0, 0, 0};
        const int kb = N & ~7;
        for (int k = 0; k < kb; k++) q[k & 3] = fma(vt[k + (size_t)i * N], b[k], q[k & 3]);
        for (int k = kb; k < N; k++) q[0] = fma(vt[k + (size_t)i * N], b[k], q[0]);
        acc = (q[0] + q[1]) + (q[2] + q[3]);
      } else {
        for (int k = 0; k < N; k++) acc = fma(vt[k + (size_t)i * N], b[k], acc);
      }
      wk[i] = acc;
    }
    for (int i = 0; i < N; i++) b[i] = wk[i];
    dlascl_vec(1.0, orgnrm, N, d, 1);
    /* dlasrt('D'): d is used only for the singular values handed back */
    for (int i = 1; i < N; i++) { const double v = d[i]; int j = i - 1; while (j >= 0 && d[j] < v) { d[j + 1] = d[j]; j--; } d[j + 1] = v; }
    dlascl_vec(orgnrm, 1.0, N, b, 1);
  }
  /* dormbr('P', 'L', 'N') = dorml2('L', 'T') over taup(0 .. N-2), reflectors in the rows of A(0:, 1:), last first */
  for (int i = N - 2; i >= 0; i--) {
    const double aii = A_(i, i + 1);
    A_(i, i + 1) = 1.0;
    gelsd_dlarf(1, N - 1 - i, 1, &A_(i, i + 1), lda, taup[i], &b[i + 1], M, work);
    A_(i, i + 1) = aii;
  }
  for (int j = 0; j < N; j++) x[j] = b[j];
done:
  if (rank_out) *rank_out = rank;
  free(a); free(b); free(tau); free(work);
  return info;
}
