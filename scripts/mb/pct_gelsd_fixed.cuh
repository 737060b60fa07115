// pct_gelsd_fixed.cuh -- the strict least-squares split (pct_gelsd.cuh) for N = 3 and N = 4 supporters with the WHOLE system in
// registers: one lane per system, every array index a compile-time constant.
//
// 94 % of the least-squares splits of the reference's domains are over three supporters, 6 % over four (pct_stab.cuh).  The
// lane-group routine of pct_gelsd.cuh keeps the system in LDS and walks it with run-time indices: a 4 x 3 solve is ~85 k cycles
// of dependent LDS round trips, loop control and exec-mask bookkeeping around ~5 k instructions of arithmetic, and the wave waits
// for it once per virtual-check round.  Here N (and with it M = N (N - 1) / 2 + 1, every vector length of dgeqr2 / dgebd2 and
// every loop bound) is a template constant: the loops unroll, A / b / V^T / d / e are scalars the register allocator sees, and
// what is data dependent -- the trailing zeros dlarf skips (iladlr / iladlc: they decide WHICH dgemv kernel a column meets), the
// block [ll, m] and the direction of a dbdsqr sweep, the sort's transpositions -- goes through predicates and select chains
// (fx_get / fx_set over at most seven elements).  The arithmetic of every element is the one of pct_gelsd.cuh, operation for
// operation (that file is the description; the scalar routines dlapy2 / dlartg / dlas2 / dlasv2 and the x87 emulation are its own);
// for these sizes some of OpenBLAS' kernels cannot occur at all and are not written out:
//   * dgemv 'T': at most three trailing columns, so never a column of a leading group of four (the 4x4 AVX2 kernel); at most
//     seven rows, so the row block is rows 0..3 or empty;
//   * dgemv 'N' (reflector from the right): at most three rows, so every row is a "tail" row: one FMA chain over the columns
//     (AVX-512 kernel set) / products and sums rounded separately (AVX2 set);
//   * dnrm2: fewer than eight elements, so everything goes through accumulator A in order;
//   * daxpy inside dger: fewer than sixteen rows, so the AVX2 set never reaches its fused blocks.
// EXPERIMENT, NOT PART OF THE LIBRARY (round 5, profiles/r05_experiments.txt item 2): bit for bit the lane-group routine (12 000 systems
// on the host, both kernel sets; 128 on the GPU) -- and only 13 % faster for three supporters (77 k against 89 k cycles; 7.1 k VALU
// instructions against 6.0 k + 444 LDS: the select chains cost what the LDS round trips did), 7 % for four, at +100 KB of code per
// kernel.  A lone wave issues one dependent instruction per ~9 cycles whatever the operands' home; the solve is its ~9 k instructions.
// scripts/mb_gelsd.py runs it as variant 2.
#ifndef PCT_GELSD_FIXED_CUH
#define PCT_GELSD_FIXED_CUH

#include "../../online-3d-bpp-pct_amd/csrc/pct_gelsd.cuh"

#if defined(__HIPCC__)
#define PCT_GUNROLL _Pragma("unroll")
#else
#define PCT_GUNROLL
#endif

namespace pct {
namespace gelsd {

// Element i (run-time) of a register array: a select chain over VALUES.  (Left to itself LLVM folds the chain back into one load
// through a computed address -- a dynamically indexed private array, i.e. scratch memory: PCT_GOPAQUE pins each element as a value.)
#if defined(__HIPCC__)
#define PCT_GOPAQUE(x) asm volatile("" : "+v"(x))
#else
#define PCT_GOPAQUE(x) ((void)0)
#endif
template <int L>
PCT_GD double fx_get(const double (&v)[L], int i) {
  double r = v[0];
  PCT_GOPAQUE(r);
  PCT_GUNROLL
  for (int t = 1; t < L; t++) {
    double vt_ = v[t];
    PCT_GOPAQUE(vt_);
    r = i == t ? vt_ : r;
  }
  return r;
}
template <int L>
PCT_GD void fx_set(double (&v)[L], int i, double x) {
  PCT_GUNROLL
  for (int t = 0; t < L; t++) {
    double vt_ = v[t];
    PCT_GOPAQUE(vt_);
    v[t] = i == t ? x : vt_;
  }
}

// dnrm2 of x[0..n), n < 8: every element through accumulator A, in order (((C + A) + B) + D = A with B = C = D = 0)
template <int L>
PCT_GD double fx_dnrm2(int n, const double (&x)[L]) {
  static_assert(L < 8, "one accumulator only");
  if (n <= 0) return 0.0;
#if !defined(PCT_GELSD_NO_NRM2_SHORTCUT)
  if (n == 1) return fabs(x[0]);
#endif
  Ext a;
  a.m = 0; a.e = 0;
  PCT_GNOUNROLL
  for (int u = 0; u < n; u++) a = ext_add(a, ext_square(fx_get(x, u)));
  return ext_sqrt_to_double(a);
}
// dlarfg(n, alpha, x): alpha in, beta out; x[0..n-1) scaled; returns tau
template <int L>
PCT_GD double fx_dlarfg(int n, double& alpha, double (&x)[L]) {
  if (n <= 1) return 0.0;
  const double safmin = SAFMIN / EPS, rsafmn = 1.0 / safmin;
  int knt = 0;
  double a0 = alpha, beta = a0;
  PCT_GNOUNROLL
  for (int pass = 0; pass < 2; pass++) {
    const double xnorm = fx_dnrm2(n - 1, x);
    if (pass == 0 && xnorm == 0.0) return 0.0;
    beta = -sgn(dlapy2(a0, xnorm), a0);
    if (pass == 1 || !(fabs(beta) < safmin)) break;
    do {
      knt++;
      PCT_GUNROLL
      for (int i = 0; i < L; i++) x[i] = i < n - 1 ? rsafmn * x[i] : x[i];
      beta *= rsafmn;
      a0 *= rsafmn;
    } while (fabs(beta) < safmin && knt < 20);
  }
  const double tau = (beta - a0) / beta;
  const double sc = 1.0 / (a0 - beta);
  PCT_GUNROLL
  for (int i = 0; i < L; i++) x[i] = i < n - 1 ? sc * x[i] : x[i];
  for (int j = 0; j < knt; j++) beta *= safmin;
  alpha = beta;
  return tau;
}
// one column of y = C^T v over the first m (<= L <= 7) rows, column j of n (<= 3) columns: the 4x2 kernel (two columns when n & 2,
// the first two), else the 4x1 kernel, over rows 0..3 when m >= 4; then the m mod 4 tail rows
template <int L>
PCT_GD double fx_dgemv_t_col(int m, int n, int j, const double (&c)[L], const double (&x)[L]) {
  static_assert(L <= 7, "row block is rows 0..3 or empty");
  const int m2 = m & ~3, m3 = m & 3;
  double yj = 0.0;
  if (L >= 4 && m2 > 0) {
    constexpr int i1 = L > 1 ? 1 : 0, i2 = L > 2 ? 2 : 0, i3 = L > 3 ? 3 : 0;
    if ((n & 2) && j < 2) {
      double l0 = 0, l1 = 0;
      l0 = l0 + c[0] * x[0];
      l1 = l1 + c[i1] * x[i1];
      l0 = l0 + c[i2] * x[i2];
      l1 = l1 + c[i3] * x[i3];
      yj = fma(1.0, l0 + l1, yj);
    } else {
      double l0 = 0, l1 = 0, l2 = 0, l3 = 0;
      l0 = l0 + c[0] * x[0];
      l1 = l1 + c[i1] * x[i1];
      l2 = l2 + c[i2] * x[i2];
      l3 = l3 + c[i3] * x[i3];
      yj = fma(1.0, (l0 + l2) + (l1 + l3), yj);
    }
  }
  if (m3 == 1) yj = fma(fx_get(c, m2), fx_get(x, m2), yj);
  else if (m3 >= 2) {
    double t = fx_get(c, m2 + 1) * fx_get(x, m2 + 1);
    t = fma(fx_get(c, m2), fx_get(x, m2), t);
    if (m3 == 3) t = fma(fx_get(c, m2 + 2), fx_get(x, m2 + 2), t);
    yj = yj + t;
  }
  return yj;
}

// The split of a stack at (s0, s1) over N supporters with contact centres in[2 i], in[2 i + 1]: x[0..N) the fractions, `ill` the
// notice; false when dbdsqr does not converge (x is zero then).  One lane, no memory besides `in`.
template <int N, typename Dot2>
PCT_GD bool split_fixed(const double* in, double s0, double s1, Dot2 dot2, double (&x)[N], bool& ill, bool avx2) {
  static_assert(N == 3 || N == 4, "register-resident sizes");
  constexpr int M = N * (N - 1) / 2 + 1;
  double a[N][M];  // a[j][i] = A(i, j): a column is a register array
  double b[M];
  PCT_GUNROLL
  for (int j = 0; j < N; j++)
    PCT_GUNROLL
    for (int i = 0; i < M; i++) a[j][i] = 0.0;
  PCT_GUNROLL
  for (int i = 0; i < M; i++) b[i] = 0.0;
  {
    double cen[N][2];
    PCT_GUNROLL
    for (int i = 0; i < N; i++) { cen[i][0] = in[2 * i]; cen[i][1] = in[2 * i + 1]; }
    int row = 0;
    PCT_GUNROLL
    for (int i = 0; i < N - 1; i++)
      PCT_GUNROLL
      for (int j = i + 1; j < N; j++) {
        const double ei0 = cen[i][0], ei1 = cen[i][1], ej0 = cen[j][0], ej1 = cen[j][1];
        const double t0 = ei0 - ej0, t1 = ei1 - ej1;
        const double mol = dot2(s0 - ei0, s1 - ei1, t0, t1);
        if (mol != 0) {
          const double rr = fabs(dot2(s0 - ej0, s1 - ej1, t0, t1)) / mol;
          a[i][row] = 1.0;
          a[j][row] = -rr;
        }
        row++;
      }
    PCT_GUNROLL
    for (int j = 0; j < N; j++) a[j][M - 1] = 1.0;
    b[M - 1] = 1.0;
  }
  PCT_GUNROLL
  for (int j = 0; j < N; j++) x[j] = 0.0;
  ill = false;
  double d[N], e[N], taup[N];
  PCT_GUNROLL
  for (int j = 0; j < N; j++) { d[j] = 0.0; e[j] = 0.0; taup[j] = 0.0; }
  // ---- stage 0: dgeqr2 + dorm2r('L', 'T') on b; stage 1: dgebd2 + the column reflectors on b -------------------------------
  PCT_GUNROLL
  for (int stage = 0; stage < 2; stage++) {
    const int rows = stage == 0 ? M : N;  // (a constant once unrolled)
    PCT_GUNROLL
    for (int i = 0; i < N; i++) {
      // -- the column reflector of column i over rows i .. rows - 1
      {
        const int len = rows - i;
        double v[M], alpha = a[i][i];
        PCT_GUNROLL
        for (int t = 0; t < M; t++) v[t] = 0.0;
        // x = rows i + 1 .. rows - 1 of column i, as v[1..]; v[0] will be the reflector's leading 1
        double xs[M - 1 > 0 ? M - 1 : 1];
        PCT_GUNROLL
        for (int t = 0; t < M - 1; t++) xs[t] = (t < len - 1 && i + 1 + t < M) ? a[i][(i + 1 + t < M) ? i + 1 + t : 0] : 0.0;
        const double tauv = fx_dlarfg(len, alpha, xs);
        PCT_GUNROLL
        for (int t = 0; t < M - 1; t++)
          if (t < len - 1 && i + 1 + t < M) a[i][(i + 1 + t < M) ? i + 1 + t : 0] = xs[t];
        if (stage == 1) d[i] = alpha;
        if (tauv != 0.0) {
          v[0] = 1.0;
          PCT_GUNROLL
          for (int t = 1; t < M; t++) v[t] = t < len ? xs[t - 1] : 0.0;
          int lastv = len;
          PCT_GUNROLL
          for (int t = M - 1; t >= 0; t--)
            if (t < len && lastv == t + 1 && v[t] == 0.0) lastv = t;
          if (lastv > 0) {
            // the trailing columns i + 1 .. N - 1 (rows i ..): iladlc, then dgemv 'T' + dger per column
            const int nc = N - i - 1;
            int lastc = nc;
            PCT_GUNROLL
            for (int c = N - 1; c > i; c--) {
              if (lastc == c - i) {
                bool nz = false;
                PCT_GUNROLL
                for (int t = 0; t < M; t++)
                  if (t < lastv && i + t < M && a[c][(i + t < M) ? i + t : 0] != 0.0) nz = true;
                if (!nz) lastc = c - i - 1;
              }
            }
            PCT_GUNROLL
            for (int tgt = i + 1; tgt <= N; tgt++) {  // columns i + 1 .. N - 1 of A, then (tgt == N) the right-hand side
              const bool isb = tgt == N;
              double col[M];
              PCT_GUNROLL
              for (int t = 0; t < M; t++) col[t] = (i + t < M) ? (isb ? b[(i + t < M) ? i + t : 0] : a[isb ? 0 : tgt][(i + t < M) ? i + t : 0]) : 0.0;
              bool on;
              if (isb) {
                on = false;
                PCT_GUNROLL
                for (int t = 0; t < M; t++)
                  if (t < lastv && col[t] != 0.0) on = true;
              } else {
                on = tgt - i - 1 < lastc;
              }
              if (on) {
                const double w = fx_dgemv_t_col(lastv, isb ? 1 : lastc, isb ? 0 : tgt - i - 1, col, v);
                const double tt = -tauv * w;
                PCT_GUNROLL
                for (int t = 0; t < M; t++)
                  if (t < lastv && i + t < M) {
                    const double nv = avx2 ? col[t] + tt * v[t] : fma(tt, v[t], col[t]);  // (fewer than sixteen rows: the AVX2 daxpy's unfused tail)
                    if (isb) b[(i + t < M) ? i + t : 0] = nv;
                    else a[isb ? 0 : tgt][(i + t < M) ? i + t : 0] = nv;
                  }
              }
            }
          }
        }
        a[i][i] = alpha;
      }
      // -- dgebd2's row reflector of row i over columns i + 1 .. N - 1, applied from the right to rows i + 1 .. N - 1
      if (stage == 1 && i < N - 1) {
        const int len = N - i - 1;
        double alpha = a[i + 1 < N ? i + 1 : 0][i];
        double xs[N];
        PCT_GUNROLL
        for (int t = 0; t < N; t++) xs[t] = (t < len - 1 && i + 2 + t < N) ? a[(i + 2 + t < N) ? i + 2 + t : 0][i] : 0.0;
        const double tauv = fx_dlarfg(len, alpha, xs);
        PCT_GUNROLL
        for (int t = 0; t < N; t++)
          if (t < len - 1 && i + 2 + t < N) a[(i + 2 + t < N) ? i + 2 + t : 0][i] = xs[t];
        e[i] = alpha;
        taup[i] = tauv;
        if (tauv != 0.0) {
          double v[N];
          v[0] = 1.0;
          PCT_GUNROLL
          for (int t = 1; t < N; t++) v[t] = t < len ? xs[t - 1] : 0.0;
          int lastv = len;
          PCT_GUNROLL
          for (int t = N - 1; t >= 0; t--)
            if (t < len && lastv == t + 1 && v[t] == 0.0) lastv = t;
          if (lastv > 0) {
            // C = rows i + 1 .. N - 1, columns i + 1 ..: iladlr (trailing zero rows), then per row the dgemv 'N' tail-row chain + dger
            int lastc = len;
            PCT_GUNROLL
            for (int r = N - 1; r > i; r--) {
              if (lastc == r - i) {
                bool nz = false;
                PCT_GUNROLL
                for (int t = 0; t < N; t++)
                  if (t < lastv && i + 1 + t < N && a[(i + 1 + t < N) ? i + 1 + t : 0][r] != 0.0) nz = true;
                if (!nz) lastc = r - i - 1;
              }
            }
            PCT_GUNROLL
            for (int r = i + 1; r < N; r++) {
              if (r - i - 1 < lastc) {
                double t = 0.0;
                PCT_GUNROLL
                for (int q = 0; q < N; q++)
                  if (q < lastv && i + 1 + q < N) {
                    const double av = a[(i + 1 + q < N) ? i + 1 + q : 0][r];
                    t = avx2 ? t + av * v[q] : fma(av, v[q], t);
                  }
                const double w = avx2 ? 0.0 + 1.0 * t : fma(1.0, t, 0.0);
                PCT_GUNROLL
                for (int q = 0; q < N; q++)
                  if (q < lastv && i + 1 + q < N) {
                    const double tt = -tauv * v[q];
                    double& ref = a[(i + 1 + q < N) ? i + 1 + q : 0][r];
                    ref = avx2 ? ref + tt * w : fma(tt, w, ref);
                  }
              }
            }
          }
        }
        a[i + 1 < N ? i + 1 : 0][i] = alpha;
      }
    }
    if (stage == 0) {
      PCT_GUNROLL
      for (int j = 0; j < N; j++)
        PCT_GUNROLL
        for (int i = 0; i < N; i++)
          if (i > j) a[j][i] = 0.0;
    }
  }
  taup[N - 1] = 0.0;
  // ---- dlalsd: scale, dbdsqr with vectors, sort, rank cut ------------------------------------------------------------------
  double orgnrm = 0.0;
  PCT_GUNROLL
  for (int i = 0; i < N; i++) orgnrm = fabs(d[i]) > orgnrm ? fabs(d[i]) : orgnrm;
  PCT_GUNROLL
  for (int i = 0; i < N - 1; i++) orgnrm = fabs(e[i]) > orgnrm ? fabs(e[i]) : orgnrm;
  if (orgnrm == 0.0) return true;
  auto lascl_mul = [](double cfrom, double cto, double (&vec)[N], int cnt) __attribute__((always_inline)) {
    const double smlnum = SAFMIN, bignum = 1.0 / smlnum;
    double cfromc = cfrom, ctoc = cto, mul;
    bool done;
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
      PCT_GUNROLL
      for (int i = 0; i < N; i++) vec[i] = i < cnt ? vec[i] * mul : vec[i];
    } while (!done);
  };
  lascl_mul(orgnrm, 1.0, d, N);
  lascl_mul(orgnrm, 1.0, e, N - 1);
  double vt[N][N];  // vt[c][r] = VT(r, c): a column is a register array
  PCT_GUNROLL
  for (int c = 0; c < N; c++)
    PCT_GUNROLL
    for (int r = 0; r < N; r++) vt[c][r] = r == c ? 1.0 : 0.0;
  double cc[N];  // the right-hand side's first N entries (b's rows N.. take no further part)
  PCT_GUNROLL
  for (int i = 0; i < N; i++) cc[i] = b[i];
  {
    // dbdsqr('U', N, ncvt = N, 0, ncc = 1) on (d, e), VT, cc
    const double hndrth = 0.01;
    const int maxitr = 6;
    int idir = 0;
    const int n = N;
    const double tol = 0x1.8ace5422aa0dbp+6 * EPS;
    double smax = 0.0;
    PCT_GUNROLL
    for (int i = 0; i < N; i++) smax = fabs(d[i]) > smax ? fabs(d[i]) : smax;
    PCT_GUNROLL
    for (int i = 0; i < N - 1; i++) smax = fabs(e[i]) > smax ? fabs(e[i]) : smax;
    double smin = 0.0;
    double sminoa = fabs(d[0]);
    if (sminoa != 0.0) {
      double mu = sminoa;
      bool stop = false;
      PCT_GUNROLL
      for (int i = 1; i < N; i++) {
        if (!stop) {
          mu = fabs(d[i]) * (mu / (mu + fabs(e[i - 1])));
          sminoa = mu < sminoa ? mu : sminoa;
          if (sminoa == 0.0) stop = true;
        }
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
    bool failed = false;
    PCT_GNOUNROLL
    while (m > 1) {
      if (iter >= n) {
        iter -= n;
        iterdivn++;
        if (iterdivn >= maxitdivn) { failed = true; break; }
      }
      smax = fabs(fx_get(d, m - 1));
      int ll = 0;
      bool split = false;
      PCT_GUNROLL
      for (int lll = 1; lll <= N - 1; lll++) {
        if (!split && lll <= m - 1) {
          ll = m - lll;
          const double abss = fabs(fx_get(d, ll - 1)), abse = fabs(fx_get(e, ll - 1));
          if (abse <= thresh) split = true;
          else {
            smax = abss > smax ? abss : smax;
            smax = abse > smax ? abse : smax;
          }
        }
      }
      if (split) {
        fx_set(e, ll - 1, 0.0);
        if (ll == m - 1) { m = m - 1; continue; }
      } else ll = 0;
      ll = ll + 1;
      if (ll == m - 1) {
        double sigmn, sigmx, sinr, cosr, sinl, cosl;
        dlasv2(fx_get(d, m - 2), fx_get(e, m - 2), fx_get(d, m - 1), sigmn, sigmx, sinr, cosr, sinl, cosl);
        fx_set(d, m - 2, sigmx); fx_set(e, m - 2, 0.0); fx_set(d, m - 1, sigmn);
        PCT_GUNROLL
        for (int c = 0; c < N; c++) {
          const double xv = fx_get(vt[c], m - 2), yv = fx_get(vt[c], m - 1);
          fx_set(vt[c], m - 2, fma(cosr, xv, sinr * yv));
          fx_set(vt[c], m - 1, fma(cosr, yv, -(sinr * xv)));
        }
        {
          const double xv = fx_get(cc, m - 2), yv = fx_get(cc, m - 1);
          fx_set(cc, m - 2, fma(cosl, xv, sinl * yv));
          fx_set(cc, m - 1, fma(cosl, yv, -(sinl * xv)));
        }
        m = m - 2;
        continue;
      }
      if (ll > oldm || m < oldll) idir = fabs(fx_get(d, ll - 1)) >= fabs(fx_get(d, m - 1)) ? 1 : 2;
      bool conv = false;
      if (idir == 1) {
        if (fabs(fx_get(e, m - 2)) <= fabs(tol) * fabs(fx_get(d, m - 1))) { fx_set(e, m - 2, 0.0); continue; }
        double mu = fabs(fx_get(d, ll - 1));
        smin = mu;
        PCT_GUNROLL
        for (int t = 0; t < N - 1; t++) {
          const int lll = ll + t;
          if (!conv && lll <= m - 1) {
            const double el = fabs(fx_get(e, lll - 1));
            if (el <= tol * mu) { fx_set(e, lll - 1, 0.0); conv = true; }
            else {
              mu = fabs(fx_get(d, lll)) * (mu / (mu + el));
              smin = mu < smin ? mu : smin;
            }
          }
        }
      } else {
        if (fabs(fx_get(e, ll - 1)) <= fabs(tol) * fabs(fx_get(d, ll - 1))) { fx_set(e, ll - 1, 0.0); continue; }
        double mu = fabs(fx_get(d, m - 1));
        smin = mu;
        PCT_GUNROLL
        for (int t = 0; t < N - 1; t++) {
          const int lll = m - 1 - t;
          if (!conv && lll >= ll) {
            const double el = fabs(fx_get(e, lll - 1));
            if (el <= tol * mu) { fx_set(e, lll - 1, 0.0); conv = true; }
            else {
              mu = fabs(fx_get(d, lll - 1)) * (mu / (mu + el));
              smin = mu < smin ? mu : smin;
            }
          }
        }
      }
      if (conv) continue;
      oldll = ll; oldm = m;
      double shift = 0.0, r = 0.0;
      {
        const double bound = EPS > hndrth * tol ? EPS : hndrth * tol;
        if (!(n * tol * (smin / smax) <= bound)) {
          double sll;
          if (idir == 1) { sll = fabs(fx_get(d, ll - 1)); dlas2(fx_get(d, m - 2), fx_get(e, m - 2), fx_get(d, m - 1), shift, r); }
          else { sll = fabs(fx_get(d, m - 1)); dlas2(fx_get(d, ll - 1), fx_get(e, ll - 1), fx_get(d, ll), shift, r); }
          if (sll > 0.0) {
            const double q = shift / sll;
            if (q * q < EPS) shift = 0.0;
          }
        }
      }
      iter = iter + m - ll;
      const int cnt = m - ll + 1;
      const int dbase = idir == 1 ? ll - 1 : m - 1, ebase = idir == 1 ? ll - 1 : m - 2, step = idir == 1 ? 1 : -1;
      const double sg = idir == 1 ? 1.0 : -1.0;
      double w0[N - 1], w1[N - 1], w2[N - 1], w3[N - 1];
      PCT_GUNROLL
      for (int p = 0; p < N - 1; p++) { w0[p] = 1.0; w1[p] = 0.0; w2[p] = 1.0; w3[p] = 0.0; }
      if (shift == 0.0) {
        double cs = 1.0, oldcs = 1.0, sn = 0.0, oldsn = 0.0;
        PCT_GUNROLL
        for (int p = 0; p < N - 1; p++) {
          if (p < cnt - 1) {
            const int dp = dbase + step * p, dq = dp + step, ep = ebase + step * p;
            dlartg(fx_get(d, dp) * cs, fx_get(e, ep), cs, sn, r);
            if (p > 0) fx_set(e, ep - step, oldsn * r);
            double dn;
            dlartg(oldcs * r, fx_get(d, dq) * sn, oldcs, oldsn, dn);
            fx_set(d, dp, dn);
            w0[p] = cs; w1[p] = sg * sn; w2[p] = oldcs; w3[p] = sg * oldsn;
          }
        }
        const int dl = dbase + step * (cnt - 1);
        const double h = fx_get(d, dl) * cs;
        fx_set(d, dl, h * oldcs);
        fx_set(e, ebase + step * (cnt - 2), h * oldsn);
      } else {
        const double d0 = fx_get(d, dbase);
        double f = (fabs(d0) - shift) * (sgn(1.0, d0) + shift / d0);
        double gg = fx_get(e, ebase);
        double cosr, sinr, cosl, sinl;
        PCT_GUNROLL
        for (int p = 0; p < N - 1; p++) {
          if (p < cnt - 1) {
            const int dp = dbase + step * p, dq = dp + step, ep = ebase + step * p;
            dlartg(f, gg, cosr, sinr, r);
            if (p > 0) fx_set(e, ep - step, r);
            double ddp = fx_get(d, dp), eep = fx_get(e, ep), ddq = fx_get(d, dq);
            f = cosr * ddp + sinr * eep;
            eep = cosr * eep - sinr * ddp;
            gg = sinr * ddq;
            ddq = cosr * ddq;
            dlartg(f, gg, cosl, sinl, r);
            ddp = r;
            f = cosl * eep + sinl * ddq;
            ddq = cosl * ddq - sinl * eep;
            fx_set(d, dp, ddp); fx_set(e, ep, eep); fx_set(d, dq, ddq);
            if (p < cnt - 2) {
              const double en = fx_get(e, ep + step);
              gg = sinl * en;
              fx_set(e, ep + step, cosl * en);
            }
            w0[p] = cosr; w1[p] = sg * sinr; w2[p] = cosl; w3[p] = sg * sinl;
          }
        }
        fx_set(e, ebase + step * (cnt - 2), f);
      }
      // dlasr: the first pair of a step into VT when chasing downwards (idir 1), into cc when chasing upwards; the second pair into the other
      PCT_GUNROLL
      for (int c = 0; c <= N; c++) {
        const bool isvt = c < N;
        const bool first = isvt == (idir == 1);
        PCT_GUNROLL
        for (int p = 0; p < N - 1; p++) {
          if (p < cnt - 1) {
            const double ct = first ? w0[p] : w2[p], st = first ? w1[p] : w3[p];
            if (ct != 1.0 || st != 0.0) {
              const int dp = dbase + step * p, dq = dp + step;
              const int rhi = idir == 1 ? dq : dp, rlo = idir == 1 ? dp : dq;
              if (isvt) {
                const double temp = fx_get(vt[isvt ? c : 0], rhi), lo = fx_get(vt[isvt ? c : 0], rlo);
                fx_set(vt[isvt ? c : 0], rhi, ct * temp - st * lo);
                fx_set(vt[isvt ? c : 0], rlo, st * temp + ct * lo);
              } else {
                const double temp = fx_get(cc, rhi), lo = fx_get(cc, rlo);
                fx_set(cc, rhi, ct * temp - st * lo);
                fx_set(cc, rlo, st * temp + ct * lo);
              }
            }
          }
        }
      }
      {
        const int el = ebase + step * (cnt - 2);
        if (fabs(fx_get(e, el)) <= thresh) fx_set(e, el, 0.0);
      }
    }
    if (failed) { ill = true; return false; }
    PCT_GUNROLL
    for (int i = 0; i < N; i++) {
      if (d[i] == 0.0) d[i] = 0.0;
      if (d[i] < 0.0) {
        d[i] = -d[i];
        PCT_GUNROLL
        for (int c = 0; c < N; c++) vt[c][i] = -1.0 * vt[c][i];
      }
    }
    // decreasing order: one transposition per singular value
    PCT_GUNROLL
    for (int i = 1; i <= N - 1; i++) {
      int isub = 1;
      double smn = d[0];
      PCT_GUNROLL
      for (int j = 2; j <= N; j++)
        if (j <= n + 1 - i && d[j - 1] <= smn) { isub = j; smn = d[j - 1]; }
      if (isub != n + 1 - i) {
        fx_set(d, isub - 1, d[N - i]);
        d[N - i] = smn;
        PCT_GUNROLL
        for (int c = 0; c < N; c++) {
          const double t = fx_get(vt[c], isub - 1);
          fx_set(vt[c], isub - 1, vt[c][N - i]);
          vt[c][N - i] = t;
        }
        const double t = fx_get(cc, isub - 1);
        fx_set(cc, isub - 1, cc[N - i]);
        cc[N - i] = t;
      }
    }
  }
  // dlasdq: into increasing order
  PCT_GUNROLL
  for (int i = 1; i <= N; i++) {
    int isub = i;
    double smn = d[i - 1];
    PCT_GUNROLL
    for (int j = 2; j <= N; j++)
      if (j >= i + 1 && d[j - 1] < smn) { isub = j; smn = d[j - 1]; }
    if (isub != i) {
      fx_set(d, isub - 1, d[i - 1]);
      d[i - 1] = smn;
      PCT_GUNROLL
      for (int c = 0; c < N; c++) {
        const double t = fx_get(vt[c], isub - 1);
        fx_set(vt[c], isub - 1, vt[c][i - 1]);
        vt[c][i - 1] = t;
      }
      const double t = fx_get(cc, isub - 1);
      fx_set(cc, isub - 1, cc[i - 1]);
      cc[i - 1] = t;
    }
  }
  int imax = 0;
  PCT_GUNROLL
  for (int i = 1; i < N; i++)
    if (fabs(d[i]) > fabs(fx_get(d, imax))) imax = i;
  const double rcond = 2.220446049250313e-16 * (double)(M > N ? M : N);
  const double tol = rcond * fabs(fx_get(d, imax));
  PCT_GUNROLL
  for (int i = 0; i < N; i++)
    if (d[i] > 0.0 && d[i] > tol / ILL_BAND && d[i] < tol * ILL_BAND) ill = true;
  PCT_GUNROLL
  for (int i = 0; i < N; i++) {
    if (d[i] <= tol) cc[i] = 0.0;
    else {
      // dlascl(d(i) -> 1) of the one element
      const double smlnum = SAFMIN, bignum = 1.0 / smlnum;
      double cfromc = d[i], ctoc = 1.0, mul;
      bool done;
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
        cc[i] = cc[i] * mul;
      } while (!done);
    }
  }
  // dgemm('T', 'N', N, 1, N): acc = fma(vt(k, i), b(k), acc) in k order (the AVX2 kernel's four-accumulator blocks need eight k: never here;
  // its rows in groups of four take (q0 + q1) + (q2 + q3) with q1 = q2 = q3 = 0)
  double wk[N];
  PCT_GUNROLL
  for (int i = 0; i < N; i++) {
    double q0 = 0.0;
    PCT_GUNROLL
    for (int kk = 0; kk < N; kk++) q0 = fma(vt[i][kk], cc[kk], q0);
    const bool four = avx2 && i < (N & ~3);
    wk[i] = four ? (q0 + 0.0) + (0.0 + 0.0) : q0;
  }
  PCT_GUNROLL
  for (int i = 0; i < N; i++) cc[i] = wk[i];
  lascl_mul(orgnrm, 1.0, cc, N);
  // dorml2('L', 'T'): the row reflectors, last first, onto b(i + 1 ..): v = (1, A(i, i + 2 ..)), strided row of A
  PCT_GUNROLL
  for (int i = N - 2; i >= 0; i--) {
    const double tp = taup[i];
    if (tp != 0.0) {
      const int len = N - 1 - i;
      double v[N], col[N];
      PCT_GUNROLL
      for (int t = 0; t < N; t++) {
        v[t] = t == 0 ? 1.0 : ((t < len && i + 1 + t < N) ? a[(i + 1 + t < N) ? i + 1 + t : 0][i] : 0.0);
        col[t] = (t < len && i + 1 + t < N) ? cc[(i + 1 + t < N) ? i + 1 + t : 0] : 0.0;
      }
      int lastv = len;
      PCT_GUNROLL
      for (int t = N - 1; t >= 0; t--)
        if (t < len && lastv == t + 1 && v[t] == 0.0) lastv = t;
      bool on = false;
      PCT_GUNROLL
      for (int t = 0; t < N; t++)
        if (t < lastv && col[t] != 0.0) on = true;
      if (lastv > 0 && on) {
        const double w = fx_dgemv_t_col(lastv, 1, 0, col, v);
        const double tt = -tp * w;
        PCT_GUNROLL
        for (int t = 0; t < N; t++)
          if (t < lastv && i + 1 + t < N) {
            double& ref = cc[(i + 1 + t < N) ? i + 1 + t : 0];
            ref = avx2 ? ref + tt * v[t] : fma(tt, v[t], ref);
          }
      }
    }
  }
  PCT_GUNROLL
  for (int j = 0; j < N; j++) x[j] = cc[j];
  return true;
}

}  // namespace gelsd
}  // namespace pct
#endif
