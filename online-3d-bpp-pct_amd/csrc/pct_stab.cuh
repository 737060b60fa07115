// pct_stab.cuh -- the reference's stability check (settings 1 / 3) restructured for one GPU
// lane per candidate: no recursion, no per-box dictionaries, no hidden state.
//
// Reference (D/ = pct_envs/PctDiscrete0/): Box.calculate_new_com D/space.py:51-71,
// calculated_impact :73-164, calculated_impact_virtual :166-267, scale_down :341-345, supporter
// search :358-379 / :405-426, check_box :447-454; ConvexHull / Line2D.orientation /
// point_in_polygen D/convex_hull.py:4-112.
//
// What is stored per placed box b (HBM, per env): its committed stack `stack[b]` (centre xyz,
// mass), its supporters `sup[b][i]` in bottom_edges order, the share it hands each of them
// `share[b][i]` and its scaled support polygon.  Everything else is recomputed:
//   * S.up_edges (a dict keyed by boxes, iterated in insertion order) == the boxes that list S
//     as a supporter, in ascending id (a key is first inserted when that box is committed), so
//     calculate_new_com(S) is a scan over the boxes above S reading share[B][idx(S)];
//   * up_edges entries shared BY REFERENCE: with one supporter, or for the "direct" supporter, the reference
//     stores the box's own thisStack OBJECT in the supporter's up_edges (D/space.py:80,96) and
//     calculate_new_com later mutates it in place (:67-71) -- such an entry always reads as the box's
//     CURRENT committed stack.  `alias[b]` = index of the supporter that holds b's stack by reference (or -1);
//     stab_com reads stack[b] instead of share[b][alias[b]].  This only matters inside a commit walk (a box
//     with >= 2 supporters recomputes all of them before it visits the first), where it changes verdicts;
//   * of S.up_virtual_edges only the entry of the currently `involved` parent is ever read and
//     it is written just before; `involved` == "on the active path";
//   * a supporter's virtual stack is recomputed when it is visited instead of when its parent
//     distributes: nothing it depends on can change in between (siblings' subtrees lie strictly
//     below the parent and never contain a sibling).
// The control flow (first False anywhere aborts everything) makes the recursion an iterative
// depth-first walk with a small explicit stack.
//
// The same source compiles for the host (tests/host/stab_host.cpp) so that the restructured
// algorithm is checked against the oracle and the reference fixtures on the CPU as well.
#ifndef PCT_STAB_CUH
#define PCT_STAB_CUH
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PCT_SD __device__ inline
#else
#define PCT_SD static inline
#endif

// 1: the least-squares split of up to five supporters runs fully unrolled in registers (stab_lstsq_fixed).  The strict
// NumPy-stream build of the continuous kernels (pct_continuous_mt.hip) sets 0 and keeps the generic solve: with the
// unrolled one its setting-1 kernel died with a GPU fault that the other builds of the same code do not show (round 2,
// profiles/r02_stability_experiments.txt); same results either way.
#ifndef PCT_STAB_FIXED_SOLVE
#define PCT_STAB_FIXED_SOLVE 1
#endif

namespace pct {

constexpr int STAB_SMAX = 16;   // supporters per box kept (more -> PCT_FLAG_STABILITY_OVERFLOW)
constexpr int STAB_LSQ = 8;     // supporters the least-squares split handles
constexpr int STAB_PMAX = 24;   // hull vertices kept
constexpr int STAB_DEPTH = 24;  // explicit stack depth

// per-env view of the stability state; geometry via `geo(i, g)` -> lx,ly,lz,xe,ye,ze,sx,sy,sz
// (the sizes are carried explicitly: in float64 (lx + x) - lx need not equal x)
struct StabState {
  int I;           // internal_node_holder (row stride)
  double* stack;   // [I][4]
  int* nsup;       // [I]
  int* sup;        // [I][STAB_SMAX]
  double* share;   // [I][STAB_SMAX][4]
  int* npoly;      // [I]
  double* poly;    // [I][STAB_PMAX][2]
  double* den;     // [I] density of each placed box (setting 3; 1.0 otherwise), D/space.py:38
  int* alias;      // [I] supporter index whose up_edges entry is this box's thisStack object itself, or -1
};

struct StabBox {  // a box being examined (candidate or placed), with its supporters
  double g[9];    // lx,ly,lz,xe,ye,ze,sx,sy,sz
  int nsup;
  int sup[STAB_SMAX];
  // the contact rectangle with supporter k is recomputed where it is needed (stab_area: two box rows from LDS and a
  // dozen flops) rather than carried here: per-lane arrays that are indexed at run time live in scratch memory, and
  // every access to them is a memory round trip
};

// ---- D/convex_hull.py ---------------------------------------------------------------------
PCT_SD double stab_slope(const double* p1, const double* p2) {
  if (p2[0] != p1[0]) return (p2[1] - p1[1]) / (p2[0] - p1[0]);
  return (p2[1] - p1[1]) * INFINITY;
}
PCT_SD int stab_orientation(double s1, double s2) {
  if (fabs(s1) == INFINITY && fabs(s2) == INFINITY) return 0;
  double diff = s2 - s1;
  if (diff > 0) return -1;
  else if (diff == 0) return 0;
  return 1;
}
// one chain of ConvexHull (:50-63 / :66-83), stale line slopes and collapse-break included
PCT_SD int stab_chain(const double (*sorted)[2], int n, bool reverse, double (*hull)[2]) {
  int len = 0;
  double s1 = 0, s2 = 0;
  for (int q = 0; q < n; q++) {
    const double* point = sorted[reverse ? n - 1 - q : q];
    if (len >= 2) {
      s1 = stab_slope(hull[len - 2], hull[len - 1]);
      s2 = stab_slope(hull[len - 1], point);
    }
    while (len >= 2 && stab_orientation(s1, s2) != -1) {
      len--;
      if (hull[0][0] == hull[len - 1][0] && hull[0][1] == hull[len - 1][1]) break;
      s1 = stab_slope(hull[len - 2], hull[len - 1]);
      s2 = stab_slope(hull[len - 1], point);
    }
    hull[len][0] = point[0];
    hull[len][1] = point[1];
    len++;
  }
  return len;
}
PCT_SD double stab_around6(double x) { return rint(x * 1e6) / 1e6; }
// contact rectangle of box `b` with placed box geometry `t`, or false if `t` does not support it.
// Discrete (D/space.py:358-376): same top, non-degenerate overlap.  Continuous
// (C/space.py:305-314,351-357): overlap decided on the np.around(.,6) of the negated-min
// intersection, top within 1e-6, rectangle = the rounded intersection.
template <bool CONT>
PCT_SD bool stab_contact(const double* bg, const double* t, double area[4]) {
  if (CONT) {
    double i0 = stab_around6(fmin(-bg[0], -t[0])), i1 = stab_around6(fmin(-bg[1], -t[1]));
    double i2 = stab_around6(fmin(bg[3], t[3])), i3 = stab_around6(fmin(bg[4], t[4]));
    if (!((i0 + i2 > 0) && (i1 + i3 > 0))) return false;
    if (!(fabs(t[5] - bg[2]) < 1e-6)) return false;
    area[0] = -i0; area[1] = -i1; area[2] = i2; area[3] = i3;
    return true;
  }
  if (t[5] != bg[2]) return false;
  double x1 = fmax(bg[0], t[0]), y1 = fmax(bg[1], t[1]);
  double x2 = fmin(bg[3], t[3]), y2 = fmin(bg[4], t[4]);
  if (x1 >= x2 || y1 >= y2) return false;
  area[0] = x1; area[1] = y1; area[2] = x2; area[3] = y2;
  return true;
}

// contact rectangle of `b` with its k-th supporter
template <bool CONT, typename Geo>
PCT_SD void stab_area(const Geo& geo, const StabBox& b, int k, double area[4]) {
  double t[9];
  geo(b.sup[k], t);
  stab_contact<CONT>(b.g, t, area);
}

// ConvexHull + scale_down of the 4*nsup contact corners of `b`; returns the vertex count
// (<= STAB_PMAX, else -1)
template <bool CONT, typename Geo>
PCT_SD int stab_polygon(const Geo& geo, const StabBox& b, double (*out)[2]) {
  double pts[4 * STAB_SMAX][2];
  int n = 0;
  for (int i = 0; i < b.nsup; i++) {
    double a[4];
    stab_area<CONT>(geo, b, i, a);
    pts[n][0] = a[0]; pts[n][1] = a[1]; n++;
    pts[n][0] = a[0]; pts[n][1] = a[3]; n++;
    pts[n][0] = a[2]; pts[n][1] = a[1]; n++;
    pts[n][0] = a[2]; pts[n][1] = a[3]; n++;
  }
  for (int i = 0; i < n; i++) pts[i][0] += pts[i][1] * 1e-6;
  for (int i = 1; i < n; i++) {  // stable sort by x
    double v0 = pts[i][0], v1 = pts[i][1];
    int j = i;
    while (j > 0 && pts[j - 1][0] > v0) { pts[j][0] = pts[j - 1][0]; pts[j][1] = pts[j - 1][1]; j--; }
    pts[j][0] = v0; pts[j][1] = v1;
  }
  double lo[4 * STAB_SMAX + 1][2], up[4 * STAB_SMAX + 1][2];
  int nl = stab_chain(pts, n, false, lo) - 1;
  int nu = stab_chain(pts, n, true, up) - 1;
  if (nl + nu > STAB_PMAX) return -1;
  int m = 0;
  for (int i = 0; i < nl; i++) { out[m][0] = lo[i][0]; out[m][1] = lo[i][1]; m++; }
  for (int i = 0; i < nu; i++) { out[m][0] = up[i][0]; out[m][1] = up[i][1]; m++; }
  double cx = 0, cy = 0;
  for (int i = 0; i < m; i++) { cx += out[i][0]; cy += out[i][1]; }
  cx /= (double)m; cy /= (double)m;
  for (int i = 0; i < m; i++) {
    out[i][0] -= (out[i][0] - cx) * 0.1;
    out[i][1] -= (out[i][1] - cy) * 0.1;
  }
  return m;
}
// point_in_polygen :97-112
PCT_SD bool stab_pip(const double* pt, const double (*co)[2], int n) {
  double lat = pt[0], lon = pt[1];
  int j = n - 1;
  bool odd = false;
  for (int i = 0; i < n; i++) {
    double a0 = co[i][0] - pt[0], a1 = co[i][1] - pt[1];
    double b0 = pt[0] - co[j][0], b1 = pt[1] - co[j][1];
    double cp = a0 * b1;
    cp -= a1 * b0;
    if (cp == 0) return false;
    if ((co[i][1] < lon && co[j][1] >= lon) || (co[j][1] < lon && co[i][1] >= lon)) {
      if ((co[i][0] + (lon - co[i][1]) / (co[j][1] - co[i][1]) * (co[j][0] - co[i][0])) < lat) odd = !odd;
    }
    j = i;
  }
  return odd;
}

// minimum-norm least squares for the >= 3 supporter case (stands in for np.linalg.lstsq / LAPACK dgelsd):
// one-sided Jacobi (Hestenes) SVD of A itself, x = sum_j V_j (U_j . b) / sigma_j^2 over sigma_j > eps * max(M,N) *
// sigma_max -- the same method, operation for operation, as the oracle's lstsq_min_norm (see there for why not A^T A)
PCT_SD void stab_lstsq(const double* A, const double* b, int M, int N, double* x) {
  double U[(STAB_LSQ * (STAB_LSQ - 1) / 2 + 1) * STAB_LSQ], V[STAB_LSQ * STAB_LSQ];
  for (int i = 0; i < M * N; i++) U[i] = A[i];
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) V[i * N + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < N; p++)
      for (int q = p + 1; q < N; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < M; r++) {
          alpha += U[r * N + p] * U[r * N + p];
          beta += U[r * N + q] * U[r * N + q];
          gamma += U[r * N + p] * U[r * N + q];
        }
        if (gamma == 0 || fabs(gamma) <= 2.220446049250313e-16 * sqrt(alpha * beta)) continue;
        rotated = true;
        double zeta = (beta - alpha) / (2 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
        double c = 1 / sqrt(1 + t * t), sn = c * t;
        for (int r = 0; r < M; r++) {
          double up = U[r * N + p], uq = U[r * N + q];
          U[r * N + p] = c * up - sn * uq;
          U[r * N + q] = sn * up + c * uq;
        }
        for (int r = 0; r < N; r++) {
          double vp = V[r * N + p], vq = V[r * N + q];
          V[r * N + p] = c * vp - sn * vq;
          V[r * N + q] = sn * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double s2[STAB_LSQ], smax2 = 0;
  for (int j = 0; j < N; j++) {
    double a2 = 0;
    for (int r = 0; r < M; r++) a2 += U[r * N + j] * U[r * N + j];
    s2[j] = a2;
    if (a2 > smax2) smax2 = a2;
  }
  const double rc = 2.220446049250313e-16 * (M > N ? M : N);
  for (int i = 0; i < N; i++) x[i] = 0;
  for (int j = 0; j < N; j++) {
    if (s2[j] <= 0 || sqrt(s2[j]) <= rc * sqrt(smax2)) continue;
    double proj = 0;
    for (int r = 0; r < M; r++) proj += U[r * N + j] * b[r];
    proj /= s2[j];
    for (int i = 0; i < N; i++) x[i] += V[i * N + j] * proj;
  }
}

// np.dot of two 2-vectors as NumPy's BLAS computes it (OpenBLAS ddot on x86 cores with FMA): acc = x0*y0;
// acc = fma(x1, y1, acc) -- see oracle/pct_oracle_stab.c dot2 (v_fma_f64 on the GPU: the same IEEE operation)
PCT_SD double stab_dot2(double x0, double x1, double y0, double y1) { return fma(x1, y1, x0 * y0); }

// The same solve with every extent a compile-time constant (N supporters, M = N(N-1)/2 + 1 rows) and every loop over
// rows, columns and column pairs unrolled: U and V are then registers, not dynamically indexed private arrays in
// scratch memory (where one rotation costs dozens of dependent HBM-latency round trips: a lane in the generic solve
// held its whole wave, and with it the launch, for milliseconds).  Operation for operation the generic routine.
template <int N>
PCT_SD void stab_lstsq_fixed(const double (&A)[(N * (N - 1) / 2 + 1) * N], double (&x)[N]) {
  constexpr int M = N * (N - 1) / 2 + 1;
  double U[M * N], V[N * N];
#pragma unroll
  for (int i = 0; i < M * N; i++) U[i] = A[i];
#pragma unroll
  for (int i = 0; i < N; i++)
#pragma unroll
    for (int j = 0; j < N; j++) V[i * N + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < N; p++)
#pragma unroll
      for (int q = p + 1; q < N; q++) {
        double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
        for (int r = 0; r < M; r++) {
          alpha += U[r * N + p] * U[r * N + p];
          beta += U[r * N + q] * U[r * N + q];
          gamma += U[r * N + p] * U[r * N + q];
        }
        if (gamma == 0 || fabs(gamma) <= 2.220446049250313e-16 * sqrt(alpha * beta)) continue;
        rotated = true;
        double zeta = (beta - alpha) / (2 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
        double c = 1 / sqrt(1 + t * t), sn = c * t;
#pragma unroll
        for (int r = 0; r < M; r++) {
          double up = U[r * N + p], uq = U[r * N + q];
          U[r * N + p] = c * up - sn * uq;
          U[r * N + q] = sn * up + c * uq;
        }
#pragma unroll
        for (int r = 0; r < N; r++) {
          double vp = V[r * N + p], vq = V[r * N + q];
          V[r * N + p] = c * vp - sn * vq;
          V[r * N + q] = sn * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double s2[N], smax2 = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    double a2 = 0;
#pragma unroll
    for (int r = 0; r < M; r++) a2 += U[r * N + j] * U[r * N + j];
    s2[j] = a2;
    if (a2 > smax2) smax2 = a2;
  }
  const double rc = 2.220446049250313e-16 * (M > N ? M : N);
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    if (s2[j] <= 0 || sqrt(s2[j]) <= rc * sqrt(smax2)) continue;
    double proj = 0;
#pragma unroll
    for (int r = 0; r < M; r++) proj += U[r * N + j] * (r == M - 1 ? 1.0 : 0.0);  // rhs = e_{M-1}
    proj /= s2[j];
#pragma unroll
    for (int i = 0; i < N; i++) x[i] += V[i * N + j] * proj;
  }
}
// the >= 3 supporter system of stab_shares for N supporters with contact centres c2 (D/space.py:118-150): one row
// per supporter pair, a closing row of ones; solved into the shares xr
template <int N>
PCT_SD void stab_split_fixed(const double (*c2)[2], const double stk[4], double (&xr)[N]) {
  constexpr int M = N * (N - 1) / 2 + 1;
  double A[M * N];
#pragma unroll
  for (int i = 0; i < M * N; i++) A[i] = 0;
  int row = 0;
#pragma unroll
  for (int i = 0; i < N - 1; i++)
#pragma unroll
    for (int j = i + 1; j < N; j++) {
      const double ei0 = c2[i][0], ei1 = c2[i][1], ej0 = c2[j][0], ej1 = c2[j][1];
      double t0 = ei0 - ej0, t1 = ei1 - ej1;
      double mol = stab_dot2(stk[0] - ei0, stk[1] - ei1, t0, t1);
      if (mol != 0) {
        double rr = fabs(stab_dot2(stk[0] - ej0, stk[1] - ej1, t0, t1)) / mol;
        A[row * N + i] = 1;
        A[row * N + j] = -rr;
      }
      row++;
    }
#pragma unroll
  for (int j = 0; j < N; j++) A[(M - 1) * N + j] = 1;
  stab_lstsq_fixed<N>(A, xr);
}


// supporters of `b` among the first n placed boxes, in box order.  False if > STAB_SMAX.
template <bool CONT, typename Geo>
PCT_SD bool stab_find_supporters(const Geo& geo, int n, StabBox& b) {
  b.nsup = 0;
  for (int i = 0; i < n; i++) {
    double t[9], area[4];
    geo(i, t);
    if (!stab_contact<CONT>(b.g, t, area)) continue;
    if (b.nsup == STAB_SMAX) return false;
    b.sup[b.nsup++] = i;
  }
  return true;
}
// a placed box again as a StabBox, from its stored supporter ids
template <bool CONT, typename Geo>
PCT_SD void stab_load_box(const Geo& geo, const StabState& st, int id, StabBox& b) {
  geo(id, b.g);
  b.nsup = st.nsup[id];
  for (int k = 0; k < b.nsup; k++) b.sup[k] = st.sup[id * STAB_SMAX + k];
}

// How `b` with stack (c, m) splits over its supporters (D/space.py:88-160 / :182-256): the share handed to supporter
// `want` (0 <= want < nsup) -- what one step of the virtual walk needs -- or, with want < 0, every share into
// out_all (the commit).  `own_centre` is the box's own centre (the virtual flavour's zero-mass shares use it).
// Cases: one supporter (everything), a supporter whose contact rectangle holds the centre of mass ("direct":
// everything, the others zero), two supporters (lever rule on the line through the contact centres), three and
// more (least squares over all pairs).
template <bool CONT, typename Geo>
PCT_SD bool stab_shares(const Geo& geo, const StabBox& b, const double stk[4], const double own_centre[3], bool virtual_,
                        int want, double out_one[4], double (*out_all)[4], int* alias = nullptr) {
  const int k = b.nsup;
  if (alias) *alias = -1;
#define put(i_, c0_, c1_, c2_, m_)                                                                        \
  do {                                                                                                    \
    if (want < 0) { out_all[i_][0] = (c0_); out_all[i_][1] = (c1_); out_all[i_][2] = (c2_); out_all[i_][3] = (m_); } \
    else if ((i_) == want) { out_one[0] = (c0_); out_one[1] = (c1_); out_one[2] = (c2_); out_one[3] = (m_); }        \
  } while (0)
  if (k == 1) {
    put(0, stk[0], stk[1], stk[2], stk[3]);
    if (alias) *alias = 0;  // up_edges[self] = self.thisStack: the object itself
    return true;
  }
  int direct = -1;
  for (int i = 0; i < k; i++) {
    double a[4];
    stab_area<CONT>(geo, b, i, a);
    bool inside = CONT ? (stk[0] - a[0] > 1e-6 && a[2] - stk[0] > 1e-6 && stk[1] - a[1] > 1e-6 && a[3] - stk[1] > 1e-6)
                       : (stk[0] > a[0] && stk[0] < a[2] && stk[1] > a[1] && stk[1] < a[3]);
    if (inside) { direct = i; break; }
  }
  if (direct >= 0) {
    if (alias) *alias = direct;
    const double* cc = virtual_ ? own_centre : stk;
    for (int i = 0; i < k; i++) {
      if (i == direct) put(i, stk[0], stk[1], stk[2], stk[3]);
      else put(i, cc[0], cc[1], cc[2], 0);
    }
    return true;
  }
  if (k > STAB_LSQ) return false;
  // contact centres: static indices (registers) up to five supporters, a run-time loop beyond
  double c2[STAB_LSQ][2];
  if (k <= 5) {
#pragma unroll
    for (int i = 0; i < 5; i++) {
      c2[i][0] = 0; c2[i][1] = 0;
      if (i < k) {
        double a[4];
        stab_area<CONT>(geo, b, i, a);
        c2[i][0] = (a[0] + a[2]) / 2;
        c2[i][1] = (a[1] + a[3]) / 2;
      }
    }
  } else {
    for (int i = 0; i < k; i++) {
      double a[4];
      stab_area<CONT>(geo, b, i, a);
      c2[i][0] = (a[0] + a[2]) / 2;
      c2[i][1] = (a[1] + a[3]) / 2;
    }
  }
  if (k == 2) {
    const double e00 = c2[0][0], e01 = c2[0][1], e10 = c2[1][0], e11 = c2[1][1];
    double t0 = e00 - e10, t1 = e01 - e11;
    double len = sqrt(stab_dot2(t0, t1, t0, t1));
    // tri_base_len ** 2: NumPy calls libm pow(len, 2.0); a correctly rounded square is len*len
    // (glibc's pow agrees except for rare near-midpoint roundings; the device pow does not)
    double l2 = len * len;
    t0 /= l2; t1 /= l2;
    double r0 = fabs(stab_dot2(stk[0] - e10, stk[1] - e11, t0, t1));
    double r1 = fabs(stab_dot2(stk[0] - e00, stk[1] - e01, t0, t1));
    put(0, e00, e01, stk[2], stk[3] * r0);
    put(1, e10, e11, stk[2], stk[3] * r1);
    return true;
  }
  if (PCT_STAB_FIXED_SOLVE && k <= 5) {  // 99.99 % of the solves (k = 3: 94 %, 4: 6 %, 5: 0.1 %): register-resident
    double x5[5] = {0, 0, 0, 0, 0};
    if (k == 3) { double x3[3]; stab_split_fixed<3>(c2, stk, x3); x5[0] = x3[0]; x5[1] = x3[1]; x5[2] = x3[2]; }
    else if (k == 4) { double x4[4]; stab_split_fixed<4>(c2, stk, x4); x5[0] = x4[0]; x5[1] = x4[1]; x5[2] = x4[2]; x5[3] = x4[3]; }
    else stab_split_fixed<5>(c2, stk, x5);
#pragma unroll
    for (int i = 0; i < 5; i++)
      if (i < k) put(i, c2[i][0], c2[i][1], stk[2], stk[3] * x5[i]);
    return true;
  }
  const int M = k * (k - 1) / 2 + 1;
  double A[(STAB_LSQ * (STAB_LSQ - 1) / 2 + 1) * STAB_LSQ], rhs[STAB_LSQ * (STAB_LSQ - 1) / 2 + 1], xr[STAB_LSQ];
  for (int i = 0; i < M * k; i++) A[i] = 0;
  for (int i = 0; i < M; i++) rhs[i] = 0;
  int row = 0;
  for (int i = 0; i < k - 1; i++)
    for (int j = i + 1; j < k; j++) {
      const double* ei = c2[i];
      const double* ej = c2[j];
      double t0 = ei[0] - ej[0], t1 = ei[1] - ej[1];
      double mol = stab_dot2(stk[0] - ei[0], stk[1] - ei[1], t0, t1);
      if (mol != 0) {
        double rr = fabs(stab_dot2(stk[0] - ej[0], stk[1] - ej[1], t0, t1)) / mol;
        A[row * k + i] = 1;
        A[row * k + j] = -rr;
      }
      row++;
    }
  for (int j = 0; j < k; j++) A[(M - 1) * k + j] = 1;
  rhs[M - 1] = 1;
  stab_lstsq(A, rhs, M, k, xr);
  for (int i = 0; i < k; i++) put(i, c2[i][0], c2[i][1], stk[2], stk[3] * xr[i]);
  return true;
#undef put
}

// calculate_new_com (D/space.py:51-71) of placed box S: own + the committed shares of the
// boxes resting on it that are not on the active path (ascending id) + `extra` (the involved
// parent's virtual share, or null).  n = number of placed boxes to consider.
template <typename Geo>
PCT_SD void stab_com(const Geo& geo, const StabState& st, int n, int S, const int* path, int npath,
                     const double* extra, double out[4]) {
  double g[9];
  geo(S, g);
  double sx = g[6], sy = g[7], sz = g[8];
  double mass = sx * sy * sz * st.den[S];
  double c0 = (g[0] + sx / 2) * mass, c1 = (g[1] + sy / 2) * mass, c2 = (g[2] + sz / 2) * mass, m = mass;
  for (int B = S + 1; B < n; B++) {
    bool inv = false;
    for (int q = 0; q < npath; q++) inv = inv || (path[q] == B);
    if (inv) continue;
    const int ns = st.nsup[B];
    for (int k = 0; k < ns; k++)
      if (st.sup[B * STAB_SMAX + k] == S) {
        // an entry held by reference reads as B's current committed stack (see the header)
        const double* e = (st.alias[B] == k) ? st.stack + (size_t)B * 4 : st.share + ((size_t)B * STAB_SMAX + k) * 4;
        c0 += e[0] * e[3]; c1 += e[1] * e[3]; c2 += e[2] * e[3];
        m += e[3];
      }
  }
  if (extra) {
    c0 += extra[0] * extra[3]; c1 += extra[1] * extra[3]; c2 += extra[2] * extra[3];
    m += extra[3];
  }
  out[0] = c0 / m; out[1] = c1 / m; out[2] = c2 / m; out[3] = m;
}

// calculated_impact_virtual(first=True) for a candidate (D/space.py:166-267): is it stable?
// err is set if a capacity (supporters, hull vertices, depth) was exceeded.
template <bool CONT, typename Geo>
PCT_SD bool stab_virtual(const Geo& geo, const StabState& st, int n, const double cand[9], double density, bool& err) {
  err = false;
  StabBox b;
  for (int i = 0; i < 9; i++) b.g[i] = cand[i];
  if (!stab_find_supporters<CONT>(geo, n, b)) { err = true; return false; }
  if (b.nsup == 0) return true;
  // explicit depth-first walk: frame = (box id or -1, its virtual stack, next supporter)
  int fid[STAB_DEPTH], fnext[STAB_DEPTH];
  double fstk[STAB_DEPTH][4];
  int path[STAB_DEPTH];  // ids of the placed boxes on the active path (frames 1..depth-1)
  int depth = 0;
  {
    double sx = cand[6], sy = cand[7], sz = cand[8];
    fid[0] = -1; fnext[0] = 0;
    fstk[0][0] = cand[0] + sx / 2; fstk[0][1] = cand[1] + sy / 2; fstk[0][2] = cand[2] + sz / 2;
    fstk[0][3] = sx * sy * sz * density * 1.0;
    depth = 1;
  }
  while (depth > 0) {
    const int d = depth - 1;
    if (fid[d] >= 0) stab_load_box<CONT>(geo, st, fid[d], b);
    else { for (int i = 0; i < 9; i++) b.g[i] = cand[i]; stab_find_supporters<CONT>(geo, n, b); }
    if (fnext[d] == 0) {
      if (b.nsup == 0) { depth--; continue; }
      if (fid[d] >= 0) {  // a placed box: its stored polygon, read where it lies
        const double (*gp)[2] = reinterpret_cast<const double (*)[2]>(st.poly + (size_t)fid[d] * STAB_PMAX * 2);
        if (!stab_pip(fstk[d], gp, st.npoly[fid[d]])) return false;
      } else {
        double poly[STAB_PMAX][2];
        const int np = stab_polygon<CONT>(geo, b, poly);
        if (np < 0) { err = true; return false; }
        if (!stab_pip(fstk[d], poly, np)) return false;
      }
    }
    if (fnext[d] >= b.nsup) { depth--; continue; }
    const int i = fnext[d]++;
    double own[3] = {b.g[0] + b.g[6] / 2, b.g[1] + b.g[7] / 2, b.g[2] + b.g[8] / 2};
    double share[4];
    if (!stab_shares<CONT>(geo, b, fstk[d], own, true, i, share, nullptr)) { err = true; return false; }
    if (depth >= STAB_DEPTH) { err = true; return false; }
    const int S = b.sup[i];
    // path = placed boxes currently involved: frames 1..d (the candidate has no id)
    int np2 = 0;
    for (int q = 1; q <= d; q++) path[np2++] = fid[q];
    stab_com(geo, st, n, S, path, np2, share, fstk[depth]);
    fid[depth] = S;
    fnext[depth] = 0;
    depth++;
  }
  return true;
}

// calculated_impact() of the box just placed as id `n` (geometry already visible through
// geo(n, .)): records its supporters / polygon / stack, propagates the shares downward and
// re-checks every box on the way (D/space.py:73-164).  Returns the stability verdict.
template <bool CONT, typename Geo>
PCT_SD bool stab_commit(const Geo& geo, StabState& st, int n, double density, bool& err) {
  err = false;
  StabBox b;
  geo(n, b.g);
  if (!stab_find_supporters<CONT>(geo, n, b)) { err = true; return false; }
  st.den[n] = density;
  {
    double sx = b.g[6], sy = b.g[7], sz = b.g[8];
    double* s = st.stack + (size_t)n * 4;
    s[0] = b.g[0] + sx / 2; s[1] = b.g[1] + sy / 2; s[2] = b.g[2] + sz / 2; s[3] = sx * sy * sz * density;
  }
  st.nsup[n] = b.nsup;
  st.alias[n] = -1;
  for (int k = 0; k < b.nsup; k++) st.sup[n * STAB_SMAX + k] = b.sup[k];
  st.npoly[n] = 0;
  if (b.nsup > 0) {
    double poly[STAB_PMAX][2];
    int np = stab_polygon<CONT>(geo, b, poly);
    if (np < 0) { err = true; return false; }
    st.npoly[n] = np;
    for (int i = 0; i < np; i++) {
      st.poly[((size_t)n * STAB_PMAX + i) * 2 + 0] = poly[i][0];
      st.poly[((size_t)n * STAB_PMAX + i) * 2 + 1] = poly[i][1];
    }
  }
  if (CONT ? (fabs(b.g[2]) < 1e-6) : (b.g[2] == 0)) return true;  // max_h == 0: check_box returns first (:448-449)
  int fid[STAB_DEPTH], fnext[STAB_DEPTH];
  int depth = 1;
  fid[0] = n; fnext[0] = 0;
  double shares[STAB_SMAX][4];
  while (depth > 0) {
    const int d = depth - 1;
    const int id = fid[d];
    stab_load_box<CONT>(geo, st, id, b);
    if (fnext[d] == 0) {
      if (b.nsup == 0) { depth--; continue; }
      const double (*gp)[2] = reinterpret_cast<const double (*)[2]>(st.poly + (size_t)id * STAB_PMAX * 2);
      const double* stk = st.stack + (size_t)id * 4;
      if (!stab_pip(stk, gp, st.npoly[id])) return false;
      // distribute to every supporter first (up_edges[self] = share; calculate_new_com())
      double own[3] = {stk[0], stk[1], stk[2]};
      int alias_k;
      if (!stab_shares<CONT>(geo, b, stk, own, false, -1, nullptr, shares, &alias_k)) { err = true; return false; }
      st.alias[id] = alias_k;
      for (int k = 0; k < b.nsup; k++) {
        double* e = st.share + ((size_t)id * STAB_SMAX + k) * 4;
        e[0] = shares[k][0]; e[1] = shares[k][1]; e[2] = shares[k][2]; e[3] = shares[k][3];
        stab_com(geo, st, n + 1, b.sup[k], (const int*)0, 0, (const double*)0, st.stack + (size_t)b.sup[k] * 4);
      }
    }
    if (fnext[d] >= b.nsup) { depth--; continue; }
    const int i = fnext[d]++;
    if (depth >= STAB_DEPTH) { err = true; return false; }
    fid[depth] = b.sup[i];
    fnext[depth] = 0;
    depth++;
  }
  return true;
}

}  // namespace pct
#endif
