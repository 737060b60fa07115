// pct_stab.cuh -- the reference's stability check (settings 1 / 3) restructured for a 64-lane wavefront that owns one
// environment: no recursion, no per-box dictionaries, no per-lane walk stacks.
//
// Reference (D/ = pct_envs/PctDiscrete0/): Box.calculate_new_com D/space.py:51-71, calculated_impact :73-164,
// calculated_impact_virtual :166-267, scale_down :341-345, supporter search :358-379 / :405-426, check_box :447-454;
// ConvexHull / Line2D.orientation / point_in_polygen D/convex_hull.py:4-112.  (C/ = PctContinuous0: the same code with
// 1e-6 margins and rounded contact rectangles.)
//
// Round 3 layout.  The per-env stability state is COMPACT and LDS-RESIDENT for the whole transition (loaded from /
// stored to its HBM mirror by the kernels): per placed box b its committed stack `stk[b]` (centre xyz, mass), its
// density, a supporter list and a scaled support polygon of exactly the lengths it needs in two append-only POOLS, and
// -- what replaces the reference's `up_edges` dictionaries -- for every (box B, supporter S) pair one pool ENTRY that
// carries the share B hands S and is chained into S's UP-LIST in the order the entries are created.  A key is first
// inserted into S.up_edges when box B is committed, so that order is ascending B: iterating S's up-list IS iterating
// the dict, and calculate_new_com(S) costs as many steps as S carries boxes, not a scan over every box above it.
//   * up_edges entries shared BY REFERENCE: with one supporter, or for the "direct" supporter, the reference stores the
//     box's own thisStack OBJECT in the supporter's up_edges (D/space.py:80,96) and calculate_new_com later mutates it
//     in place (:67-71) -- such an entry always reads as the box's CURRENT committed stack.  `alias(b)` = index of the
//     supporter that holds b's stack by reference (or none); stab_com reads stk[b] instead of the entry's share.
//   * of S.up_virtual_edges only the entry of the currently `involved` parent is ever read and it is written just
//     before.  Among the boxes on the active path only the PARENT can appear in S's up-list (an ancestor further up
//     rests strictly higher than S's top), so "skip the involved boxes" is "skip the parent's entry".
//   * the virtual check of a candidate (calculated_impact_virtual(first=True)) has no side effect and its verdict is the
//     AND of one point-in-polygon test per visited (path, box): the visits are independent TASKS -- (candidate, box S,
//     parent P, the virtual stack S receives) -- that a wave pops 64 at a time from an LDS queue; every task pushes one
//     child per supporter of S.  The level-0 task of a candidate finds its supporters, builds its hull in a small
//     per-lane LDS workspace and pushes its children.  Nothing of a walk lives in scratch memory.
//   * the commit of the placed box (calculated_impact) is order dependent and stays sequential (one lane), on the same
//     LDS state, its depth-first stack and hull workspace in LDS as well.
// Capacities (pool entries, polygon vertices, workspace, queue) are run-time parameters of a launch; exceeding one
// makes the caller requeue the env -- state untouched, because nothing is stored before the transition is complete --
// for the large-capacity retry pass.
//
// The same source compiles for the host (tests/host/stab_host.cpp) so that the restructured algorithm is checked
// against the tests' CPU restatement and the reference fixtures on the CPU as well.
#ifndef PCT_STAB_CUH
#define PCT_STAB_CUH
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PCT_SD __device__ __forceinline__
#define PCT_HD __host__ __device__ inline
#define PCT_SM __device__ __forceinline__
#else
#define PCT_SD static inline
#define PCT_HD static inline
#define PCT_SM inline
#endif

#include "pct_gelsd.cuh"
#include "pct_pow.cuh"

namespace pct {

constexpr int STAB_LSQ = 25;         // most supporters a least-squares split can be given (a launch's caps.lsq_n may be lower:
                                     // 8 in the normal pass, 25 in the retry pass).  25 is LAPACK's own limit for the path of dgelsd
                                     // that pct_gelsd.cuh restates: dlalsd solves an n <= SMLSIZ = 25 bidiagonal system by dlasdq and
                                     // switches to the divide-and-conquer routines (dlasda / dlalsa) beyond it.  A 5 x 5 footprint on
                                     // unit tiles -- the widest stack of the 10^3 / items 1..5 domain -- has 25 supporters (round 5: 16)
constexpr int STAB_NSUP_MAX = 255;   // supporters per box (8-bit count)
constexpr uint32_t STAB_END = 0xFFFu;  // end of an up-list / "no parent"
constexpr uint32_t STAB_NOBOX = 0x3FFu;
// why a capacity error was raised (kept in the env's flag word next to PCT_FLAG_STABILITY_OVERFLOW, include/pct_env.h)
constexpr uint32_t STAB_WHY_QUEUE = 0x100u;     // one task's children do not fit the walk queue
constexpr uint32_t STAB_WHY_WS = 0x200u;        // a candidate's hull does not fit the workspace / > 255 supporters
constexpr uint32_t STAB_WHY_HULL = 0x400u;      // a hull of more than 255 vertices
constexpr uint32_t STAB_WHY_SPLIT = 0x800u;     // more supporters than this launch's least-squares workspace takes (caps.lsq_n; 25 in the retry pass), none of them direct
constexpr uint32_t STAB_WHY_COMMIT = 0x1000u;   // the commit: pools, workspace or depth-first stack
constexpr uint32_t STAB_WHY_LOAD = 0x2000u;     // the stored state does not fit this launch's pools
constexpr uint32_t STAB_NOTE_ILL = 0x4000u;     // (not a capacity: carries the ill-conditioning notice out of a heuristic's probes)

// capacities of one launch (LDS sizes follow from them: stab_state_bytes / stab_ws_bytes)
struct StabCaps {
  int SP;       // share / supporter pool entries (sum of supporter counts over the placed boxes), <= 4094
  int PP;       // polygon pool vertices
  int ws_bytes;  // hull workspace (level-0 candidates of a round / the commit's hull and depth-first stack)
  int queue;     // walk tasks the LDS queue holds
  int lsq_n;     // supporters the wave-cooperative least-squares split takes, 6..STAB_LSQ
  int lsq_bytes; // its LDS workspace: at least one system of lsq_n supporters (stab_lsq_bytes); what is beyond that holds
                 // further systems of the smaller size classes side by side
  int gelsd;     // PCT_LSTSQ_GELSD (1) / PCT_LSTSQ_GELSD_AVX2 (2): every least-squares split (three supporters and more) is solved as
                 // LAPACK dgelsd solves it (pct_gelsd.cuh; with the arithmetic of OpenBLAS' AVX-512 / AVX2 kernel set), one lane per
                 // system on a slot of the same workspace, instead of the Jacobi solves
};
#if !defined(__HIPCC__)
static int g_stab_host_gelsd = 1;  // host test build (tests/host): the mode the handle carries in StabCaps on the device
#endif

// per-env stability state: pointers into LDS (device) or host memory
struct StabState {
  double* stk;     // [I][4] committed stack of every placed box
  double* den;     // [I] density (setting 3; 1.0 otherwise), D/space.py:38
  uint32_t* meta;  // [I][2]: [0] = nsup | (alias + 1) << 8 | npoly << 16, [1] = soff | poff << 16
  uint32_t* up;    // [I] up-list head | tail << 16 (STAB_END: empty)
  uint32_t* ent;   // [SP] supporter S | owner B << 10 | next << 20
  double* share;   // [SP][4] the share entry j hands its supporter
  double* poly;    // [PP][2] scaled support polygons
  int SP, PP;
  int n_ent, n_poly;  // used pool entries / vertices
};
PCT_SD int stab_nsup(const StabState& st, int b) { return (int)(st.meta[2 * b] & 0xFFu); }
PCT_SD int stab_alias(const StabState& st, int b) { return (int)((st.meta[2 * b] >> 8) & 0xFFu) - 1; }
PCT_SD int stab_npoly(const StabState& st, int b) { return (int)((st.meta[2 * b] >> 16) & 0xFFu); }
PCT_SD int stab_soff(const StabState& st, int b) { return (int)(st.meta[2 * b + 1] & 0xFFFFu); }
PCT_SD int stab_poff(const StabState& st, int b) { return (int)(st.meta[2 * b + 1] >> 16); }

PCT_HD size_t stab_state_bytes(int I, const StabCaps& c) {
  return (size_t)I * (4 + 1) * sizeof(double) + (size_t)I * 3 * sizeof(uint32_t) + (size_t)c.SP * (sizeof(uint32_t) + 4 * sizeof(double)) +
         (size_t)c.PP * 2 * sizeof(double);
}
// carve the state out of a 16-byte aligned region (doubles first)
PCT_SD StabState stab_carve(unsigned char* base, int I, const StabCaps& c) {
  StabState st;
  double* d = reinterpret_cast<double*>(base);
  st.stk = d; d += (size_t)I * 4;
  st.den = d; d += I;
  st.share = d; d += (size_t)c.SP * 4;
  st.poly = d; d += (size_t)c.PP * 2;
  uint32_t* w = reinterpret_cast<uint32_t*>(d);
  st.meta = w; w += (size_t)I * 2;
  st.up = w; w += I;
  st.ent = w;
  st.SP = c.SP; st.PP = c.PP;
  st.n_ent = 0; st.n_poly = 0;
  return st;
}

// supporter ids of a box: the low 10 bits of consecutive 32-bit words -- the pool entries of a placed box, or the id
// list a box under examination keeps in its workspace
struct StabSup {
  const uint32_t* w;
  PCT_SM int operator()(int i) const { return (int)(w[i] & 0x3FFu); }
};

// ---- D/convex_hull.py ---------------------------------------------------------------------
PCT_SD double stab_slope(double p1x, double p1y, double p2x, double p2y) {
  if (p2x != p1x) return (p2y - p1y) / (p2x - p1x);
  return (p2y - p1y) * INFINITY;
}
PCT_SD int stab_orientation(double s1, double s2) {
  if (fabs(s1) == INFINITY && fabs(s2) == INFINITY) return 0;
  double diff = s2 - s1;
  if (diff > 0) return -1;
  else if (diff == 0) return 0;
  return 1;
}
// one chain of ConvexHull (:50-63 / :66-83), stale line slopes and collapse-break included.  `hull` may be `sorted`
// itself for the forward chain (the chain never holds more points than it has consumed).
PCT_SD int stab_chain(const double (*sorted)[2], int n, bool reverse, double (*hull)[2]) {
  int len = 0;
  double s1 = 0, s2 = 0;
  for (int q = 0; q < n; q++) {
    const int src = reverse ? n - 1 - q : q;
    const double px = sorted[src][0], py = sorted[src][1];
    if (len >= 2) {
      s1 = stab_slope(hull[len - 2][0], hull[len - 2][1], hull[len - 1][0], hull[len - 1][1]);
      s2 = stab_slope(hull[len - 1][0], hull[len - 1][1], px, py);
    }
    while (len >= 2 && stab_orientation(s1, s2) != -1) {
      len--;
      if (hull[0][0] == hull[len - 1][0] && hull[0][1] == hull[len - 1][1]) break;
      s1 = stab_slope(hull[len - 2][0], hull[len - 2][1], hull[len - 1][0], hull[len - 1][1]);
      s2 = stab_slope(hull[len - 1][0], hull[len - 1][1], px, py);
    }
    hull[len][0] = px;
    hull[len][1] = py;
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
// contact rectangle of the box with geometry `bg` with placed box `s` (a supporter of it): recomputed where it is
// needed -- two box rows from LDS and a dozen flops -- rather than carried per box
template <bool CONT, typename Geo>
PCT_SD void stab_area(const Geo& geo, const double bg[9], int s, double area[4]) {
  double t[9];
  geo(s, t);
  stab_contact<CONT>(bg, t, area);
}

// ConvexHull of the 4 * k contact corners of the box `bg` with its supporters, in the workspace: the sorted points (and,
// in place, the lower chain) in pts[4k], the upper chain in up[4k + 1].  The polygon is lower[0..nl) then upper[0..nu).
template <bool CONT, typename Geo, typename Sup>
PCT_SD void stab_hull(const Geo& geo, const double bg[9], int k, const Sup& sup, double (*pts)[2], double (*up)[2], int& nl,
                      int& nu) {
  int n = 0;
  for (int i = 0; i < k; i++) {
    double a[4];
    stab_area<CONT>(geo, bg, sup(i), a);
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
  nu = stab_chain(pts, n, true, up) - 1;   // the upper chain first: it reads the sorted points ...
  nl = stab_chain(pts, n, false, pts) - 1;  // ... which the lower chain then overwrites in place
}
// centroid of the hull vertices (scale_down D/space.py:341-345 shrinks every vertex towards it by a tenth)
PCT_SD void stab_hull_centroid(const double (*lo)[2], int nl, const double (*up)[2], int nu, double& cx, double& cy) {
  cx = 0; cy = 0;
  for (int i = 0; i < nl; i++) { cx += lo[i][0]; cy += lo[i][1]; }
  for (int i = 0; i < nu; i++) { cx += up[i][0]; cy += up[i][1]; }
  const double m = (double)(nl + nu);
  cx /= m; cy /= m;
}
PCT_SD void stab_hull_vertex(const double (*lo)[2], int nl, const double (*up)[2], int i, double cx, double cy, double& vx,
                             double& vy) {
  const double* v = i < nl ? lo[i] : up[i - nl];
  vx = v[0]; vy = v[1];
  vx -= (vx - cx) * 0.1;
  vy -= (vy - cy) * 0.1;
}
// one edge of point_in_polygen :97-112: vertex i (cx, cy) against the previous vertex j
PCT_SD bool stab_pip_edge(double lat, double lon, double ix, double iy, double jx, double jy, bool& odd) {
  double a0 = ix - lat, a1 = iy - lon;
  double b0 = lat - jx, b1 = lon - jy;
  double cp = a0 * b1;
  cp -= a1 * b0;
  if (cp == 0) return false;
  if ((iy < lon && jy >= lon) || (jy < lon && iy >= lon)) {
    if ((ix + (lon - iy) / (jy - iy) * (jx - ix)) < lat) odd = !odd;
  }
  return true;
}
// point_in_polygen on stored (scaled) vertices
PCT_SD bool stab_pip(const double* pt, const double (*co)[2], int n) {
  bool odd = false;
  int j = n - 1;
  for (int i = 0; i < n; i++) {
    if (!stab_pip_edge(pt[0], pt[1], co[i][0], co[i][1], co[j][0], co[j][1], odd)) return false;
    j = i;
  }
  return odd;
}
// ... and on a hull still in the workspace (vertices scaled on the fly)
PCT_SD bool stab_pip_hull(const double* pt, const double (*lo)[2], int nl, const double (*up)[2], int nu) {
  const int n = nl + nu;
  double cx, cy;
  stab_hull_centroid(lo, nl, up, nu, cx, cy);
  bool odd = false;
  double jx = 0, jy = 0;
  if (n > 0) stab_hull_vertex(lo, nl, up, n - 1, cx, cy, jx, jy);
  for (int i = 0; i < n; i++) {
    double ix, iy;
    stab_hull_vertex(lo, nl, up, i, cx, cy, ix, iy);
    if (!stab_pip_edge(pt[0], pt[1], ix, iy, jx, jy, odd)) return false;
    jx = ix; jy = iy;
  }
  return odd;
}

// minimum-norm least squares for the >= 3 supporter case (stands in for np.linalg.lstsq / LAPACK dgelsd):
// one-sided Jacobi (Hestenes) SVD of A itself, x = sum_j V_j (U_j . b) / sigma_j^2 over sigma_j > eps * max(M,N) *
// sigma_max -- the same method, operation for operation, as the CPU restatement the tests check against (lstsq_min_norm; A^T A would square the condition number).
// `ill` reports a rank decision taken within a factor STAB_ILL_BAND of the cut: there the reference's own verdict
// depends on the rounding noise of its LAPACK build (profiles/r02_lstsq_limit.txt, r03_lstsq_limit.txt).
constexpr double STAB_ILL_BAND = 1e3;
#if !defined(__HIPCC__)
PCT_SD void stab_lstsq(const double* A, const double* b, int M, int N, double* x, bool& ill) {
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
    const double sj = sqrt(s2[j]), cut = rc * sqrt(smax2);
    if (s2[j] > 0 && sj > cut / STAB_ILL_BAND && sj < cut * STAB_ILL_BAND) ill = true;
    if (s2[j] <= 0 || sj <= cut) continue;
    double proj = 0;
    for (int r = 0; r < M; r++) proj += U[r * N + j] * b[r];
    proj /= s2[j];
    for (int i = 0; i < N; i++) x[i] += V[i * N + j] * proj;
  }
}
#endif

// np.dot of two 2-vectors as NumPy's BLAS computes it (OpenBLAS' ddot kernel for AVX-512 hosts): acc = x0*y0;
// acc = fma(x1, y1, acc) (NumPy's 2-vector dot on an FMA host; v_fma_f64 on the GPU is the same IEEE operation)
PCT_SD double stab_dot2(double x0, double x1, double y0, double y1) { return fma(x1, y1, x0 * y0); }
// ... and on a host whose OpenBLAS runs the "Haswell" kernel set (AVX2 hosts, AMD Zen included): that ddot's tail loop is compiled
// without FMA, both products and the sum are rounded (PCT_LSTSQ_GELSD_AVX2; measured against np.dot under OPENBLAS_CORETYPE=HASWELL)
struct StabDot2 {
  bool plain;
  PCT_SM double operator()(double x0, double x1, double y0, double y1) const { return plain ? x0 * y0 + x1 * y1 : fma(x1, y1, x0 * y0); }
};

// The same solve with every extent a compile-time constant (N supporters, M = N(N-1)/2 + 1 rows) and every loop over
// rows, columns and column pairs unrolled: U and V are then registers, not dynamically indexed private arrays in
// scratch memory.  Operation for operation the generic routine.
template <int N>
PCT_SD void stab_lstsq_fixed(const double (&A)[(N * (N - 1) / 2 + 1) * N], double (&x)[N], bool& ill) {
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
    const double sj = sqrt(s2[j]), cut = rc * sqrt(smax2);
    if (s2[j] > 0 && sj > cut / STAB_ILL_BAND && sj < cut * STAB_ILL_BAND) ill = true;
    if (s2[j] <= 0 || sj <= cut) continue;
    double proj = 0;
#pragma unroll
    for (int r = 0; r < M; r++) proj += U[r * N + j] * (r == M - 1 ? 1.0 : 0.0);  // rhs = e_{M-1}
    proj /= s2[j];
#pragma unroll
    for (int i = 0; i < N; i++) x[i] += V[i * N + j] * proj;
  }
}
// the >= 3 supporter system for N supporters with contact centres c2 (D/space.py:118-150): one row per supporter pair,
// a closing row of ones; solved into the shares xr
template <int N>
PCT_SD void stab_split_fixed(const double (*c2)[2], const double stk[4], double (&xr)[N], bool& ill) {
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
  stab_lstsq_fixed<N>(A, xr, ill);
}

// optional counters of one env-step (timed build only; a null pointer compiles to nothing): where a long step spends its time
struct StabStats {
  int commit_visits;                  // boxes the commit's depth-first walk expanded (re-visits through other paths included)
  int v_passes, v_tasks, v_narrow;    // virtual checks: queue passes, tasks popped, passes that popped a single task
  int v_level0;                       // candidates that went through a level-0 task
  int lsq3, lsq4, lsq5, lsqx;         // least-squares splits by supporter count (x: the generic 6..STAB_LSQ solve)
  int lsq_rounds;                     // calls of the wave's solve (stab_lsq_wave / stab_gelsd_slots): each is one solve's latency for the wave
  int lsq_rounds_l0;                  // ... of them in the first pass of a round (the candidates' own splits)
  int v_rounds, v_calls;              // level-0 rounds (class x workspace capacity), calls of stab_virtual_wave (batches of 64 candidates)
};

// How a box with stack `stk` splits over its k supporters (D/space.py:88-160 / :182-256).
//   mode 0: one supporter (everything; the entry IS the box's stack object)
//   mode 1: a supporter whose contact rectangle holds the centre of mass ("direct": everything, the others a zero-mass
//           share; that entry is the stack object too)
//   mode 2: two supporters (lever rule on the line through the contact centres)
//   mode 3: three and more (least squares over all pairs)
// f[i]: the fraction of the mass supporter i receives (modes 2, 3; i < 5 -- more than five supporters keep theirs in fx)
//   mode 4 (device only): three and more, beyond five -- the system is left to the wave-cooperative solve (stab_lsq_wave), which
//           turns it into mode 3 with the fractions in StabSplitX
struct StabSplit {
  int mode, direct;
  double f[5];
  bool ill;
};
// the fractions of a split over more than five supporters (0.01 % of them) live in a run-time indexed private array --
// scratch memory -- kept OUT of StabSplit so that the struct itself stays in registers
struct StabSplitX { double fx[STAB_LSQ]; };
template <bool CONT, typename Geo, typename Sup>
PCT_SD bool stab_split(const Geo& geo, const double bg[9], int k, const Sup& sup, const double stk[4], StabSplit& sp, StabSplitX& sx,
                       StabStats* ss = nullptr, int lsq_n = STAB_LSQ, int lsq_mode = 0) {
#if !defined(__HIPCC__)
  if (g_stab_host_gelsd) lsq_mode = g_stab_host_gelsd;
#endif
  const bool gelsd = lsq_mode != 0, gelsd_avx2 = lsq_mode == 2;
  (void)gelsd_avx2;
  sp.mode = 0; sp.direct = -1; sp.ill = false;
#pragma unroll
  for (int i = 0; i < 5; i++) sp.f[i] = 0;
  if (k == 1) return true;
  for (int i = 0; i < k; i++) {
    double a[4];
    stab_area<CONT>(geo, bg, sup(i), a);
    bool inside = CONT ? (stk[0] - a[0] > 1e-6 && a[2] - stk[0] > 1e-6 && stk[1] - a[1] > 1e-6 && a[3] - stk[1] > 1e-6)
                       : (stk[0] > a[0] && stk[0] < a[2] && stk[1] > a[1] && stk[1] < a[3]);
    if (inside) { sp.direct = i; break; }
  }
  if (sp.direct >= 0) { sp.mode = 1; return true; }
  if (k > lsq_n || k > STAB_LSQ) return false;
  if (gelsd && k >= 3) {
    // strict mode: the split is np.linalg.lstsq as the reference's NumPy executes it (pct_gelsd.cuh)
    if (ss) { ss->lsq3 += k == 3; ss->lsq4 += k == 4; ss->lsq5 += k == 5; ss->lsqx += k > 5; }
#if defined(__HIPCC__)
    sp.mode = 4;  // solved by one lane on a slot of the wave's least-squares workspace (stab_gelsd_slots)
    (void)sx;
    return true;
#else
    sp.mode = 3;
    double cen[2 * STAB_LSQ], wsg[(STAB_LSQ * (STAB_LSQ - 1) / 2 + 1) * (STAB_LSQ + 1) + STAB_LSQ * STAB_LSQ + 8 * STAB_LSQ];
    for (int i = 0; i < k; i++) {
      double a[4];
      stab_area<CONT>(geo, bg, sup(i), a);
      cen[2 * i] = (a[0] + a[2]) / 2;
      cen[2 * i + 1] = (a[1] + a[3]) / 2;
    }
    double xg[STAB_LSQ];
    gelsd::split_t(wsg, k, cen, stk[0], stk[1], StabDot2{gelsd_avx2}, xg, sp.ill, gelsd_avx2);
    for (int i = 0; i < k; i++) {
      if (k <= 5) sp.f[i] = xg[i];
      else sx.fx[i] = xg[i];
    }
    return true;
#endif
  }
  if (k <= 5) {
    // contact centres with static indices (registers).  Two to five supporters are 99.99 % of the splits (of the
    // least-squares ones k = 3: 94 %, 4: 6 %, 5: 0.1 %).  The register-resident solve for five costs 160 VGPRs -- and the
    // generic one, on private arrays in scratch memory, costs the WAVE that holds such a lane hundreds of thousands of
    // cycles: with 4096 envs per launch there is one in every other launch, and a launch lasts as long as its slowest
    // env (measured, c1: 9.2 M env-steps/s with it, 7.8 M without; profiles/r03_stability_tuning.txt)
    double c2[5][2];
#pragma unroll
    for (int i = 0; i < 5; i++) {
      c2[i][0] = 0; c2[i][1] = 0;
      if (i < k) {
        double a[4];
        stab_area<CONT>(geo, bg, sup(i), a);
        c2[i][0] = (a[0] + a[2]) / 2;
        c2[i][1] = (a[1] + a[3]) / 2;
      }
    }
    if (k == 2) {
      const double e00 = c2[0][0], e01 = c2[0][1], e10 = c2[1][0], e11 = c2[1][1];
      double t0 = e00 - e10, t1 = e01 - e11;
      const StabDot2 dot2{gelsd_avx2};  // (the lever rule's np.dot / np.linalg.norm follow the host flavour too)
      double len = sqrt(dot2(t0, t1, t0, t1));
      // tri_base_len ** 2: NumPy calls libm pow(len, 2.0), which is not correctly rounded -- pct_pow.cuh restates glibc's
      // pow as its FMA build executes it.  Geo::kSquareIsPow: every length this env can produce -- sqrt of a sum of two squares
      // of half-integers below 32 (the 5-bit-coordinate discrete kernels) -- has pow(len, 2.0) == len * len (exhaustive up
      // to 43 per axis, tests/test_stab_host.py; the first length that differs is that of (39.5, 43.5))
      double l2 = Geo::kSquareIsPow ? len * len : pow_glibc_fma(len, 2.0);
      t0 /= l2; t1 /= l2;
      sp.f[0] = fabs(dot2(stk[0] - e10, stk[1] - e11, t0, t1));
      sp.f[1] = fabs(dot2(stk[0] - e00, stk[1] - e01, t0, t1));
      sp.mode = 2;
      return true;
    }
    sp.mode = 3;
    if (ss) { ss->lsq3 += k == 3; ss->lsq4 += k == 4; ss->lsq5 += k == 5; }
#if defined(PCT_STAB_COOP_ALL) && defined(__HIPCC__)
    // every least-squares split goes to the wave (stab_lsq_wave): the register-resident solves below are not compiled
    sp.mode = 4;
    return true;
#else
    if (k == 3) { double x3[3]; stab_split_fixed<3>(c2, stk, x3, sp.ill); sp.f[0] = x3[0]; sp.f[1] = x3[1]; sp.f[2] = x3[2]; }
    else if (k == 4) { double x4[4]; stab_split_fixed<4>(c2, stk, x4, sp.ill); sp.f[0] = x4[0]; sp.f[1] = x4[1]; sp.f[2] = x4[2]; sp.f[3] = x4[3]; }
    else stab_split_fixed<5>(c2, stk, sp.f, sp.ill);
    return true;
#endif
  }
  if (ss) ss->lsqx++;
#if defined(__HIPCC__)
  // six and more supporters (0.01 % of the splits): a per-lane solve on private arrays lived in scratch memory and cost the wave
  // that held such a lane 1.7 M cycles -- THE long launches of the stability settings (profiles/r04_stability_cliff.txt).  The
  // system goes to the wave instead: stab_lsq_wave, rows across the lanes, the matrix in LDS.
  sp.mode = 4;
  (void)sx;
  return true;
#else
  sp.mode = 3;
  double c2[STAB_LSQ][2];
  for (int i = 0; i < k; i++) {
    double a[4];
    stab_area<CONT>(geo, bg, sup(i), a);
    c2[i][0] = (a[0] + a[2]) / 2;
    c2[i][1] = (a[1] + a[3]) / 2;
  }
  const int M = k * (k - 1) / 2 + 1;
  double A[(STAB_LSQ * (STAB_LSQ - 1) / 2 + 1) * STAB_LSQ], rhs[STAB_LSQ * (STAB_LSQ - 1) / 2 + 1];
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
  stab_lstsq(A, rhs, M, k, sx.fx, sp.ill);
  return true;
#endif
}
// the share supporter i receives.  `own_centre`: the box's own centre, which the VIRTUAL flavour puts into the zero-mass
// shares of a direct split (the commit uses the stack's centre there)
template <bool CONT, typename Geo, typename Sup>
PCT_SD void stab_share_of(const Geo& geo, const double bg[9], int k, const Sup& sup, const double stk[4],
                          const double own_centre[3], bool virtual_, const StabSplit& sp, const StabSplitX& sx, int i, double out[4]) {
  if (sp.mode == 0 || (sp.mode == 1 && i == sp.direct)) {
    out[0] = stk[0]; out[1] = stk[1]; out[2] = stk[2]; out[3] = stk[3];
    return;
  }
  if (sp.mode == 1) {
    const double* cc = virtual_ ? own_centre : stk;
    out[0] = cc[0]; out[1] = cc[1]; out[2] = cc[2]; out[3] = 0;
    return;
  }
  double a[4];
  stab_area<CONT>(geo, bg, sup(i), a);
  double f = sp.f[0];
  f = i == 1 ? sp.f[1] : f;
  f = i == 2 ? sp.f[2] : f;
  f = i == 3 ? sp.f[3] : f;
  f = i == 4 ? sp.f[4] : f;
  if (k > 5) f = sx.fx[i];
  out[0] = (a[0] + a[2]) / 2;
  out[1] = (a[1] + a[3]) / 2;
  out[2] = stk[2];
  out[3] = stk[3] * f;
}

// calculate_new_com (D/space.py:51-71) of placed box S: own + the committed shares of the boxes resting on it (its
// up-list, in insertion order) except the one of the involved parent `skip` (or none: STAB_NOBOX) + `extra` (the
// parent's virtual share, or null).
template <typename Geo>
PCT_SD void stab_com(const Geo& geo, const StabState& st, int S, int skip, const double* extra, double out[4]) {
  double g[9];
  geo(S, g);
  double sx = g[6], sy = g[7], sz = g[8];
  double mass = sx * sy * sz * st.den[S];
  double c0 = (g[0] + sx / 2) * mass, c1 = (g[1] + sy / 2) * mass, c2 = (g[2] + sz / 2) * mass, m = mass;
  for (uint32_t j = st.up[S] & 0xFFFFu; j != STAB_END; ) {
    const uint32_t w = st.ent[j];
    const int B = (int)((w >> 10) & 0x3FFu);
    if (B != skip) {
      // an entry held by reference reads as B's current committed stack (see the header)
      const int kk = (int)j - stab_soff(st, B);
      const double* e = (stab_alias(st, B) == kk) ? st.stk + (size_t)B * 4 : st.share + (size_t)j * 4;
      c0 += e[0] * e[3]; c1 += e[1] * e[3]; c2 += e[2] * e[3];
      m += e[3];
    }
    j = w >> 20;
  }
  if (extra) {
    c0 += extra[0] * extra[3]; c1 += extra[1] * extra[3]; c2 += extra[2] * extra[3];
    m += extra[3];
  }
  out[0] = c0 / m; out[1] = c1 / m; out[2] = c2 / m; out[3] = m;
}

// ---- the virtual check as tasks ---------------------------------------------------------------------------------
// The children of a box under examination (geometry bg, k supporters `sup`, stack `stk`): its stack is split over the
// supporters and every supporter Si gets the virtual stack it would then carry -- calculate_new_com of Si with the
// parent's committed entry (`skip`: the parent's id, STAB_NOBOX for a candidate) replaced by the virtual share.
// `emit(Si, child_stack)` is called once per supporter, in order.  False: a capacity was exceeded.
template <bool CONT, typename Geo, typename Emit>
PCT_SD void stab_children_emit(const Geo& geo, const StabState& st, const double bg[9], int k, const StabSup& sup, const double stk[4],
                               int skip, const StabSplit& sp, const StabSplitX& sx, Emit emit) {
  const double own[3] = {bg[0] + bg[6] / 2, bg[1] + bg[7] / 2, bg[2] + bg[8] / 2};
  for (int i = 0; i < k; i++) {
    double share[4], child[4];
    stab_share_of<CONT>(geo, bg, k, sup, stk, own, true, sp, sx, i, share);
    const int Si = sup(i);
    stab_com(geo, st, Si, skip, share, child);
    emit(Si, child);
  }
}
#if !defined(__HIPCC__)
template <bool CONT, typename Geo, typename Emit>
PCT_SD bool stab_children(const Geo& geo, const StabState& st, const double bg[9], int k, const StabSup& sup, const double stk[4],
                          int skip, bool& ill, Emit emit, StabStats* ss = nullptr) {
  StabSplit sp;
  StabSplitX sx;
  if (!stab_split<CONT>(geo, bg, k, sup, stk, sp, sx, ss)) return false;
  ill = ill || sp.ill;
  const double own[3] = {bg[0] + bg[6] / 2, bg[1] + bg[7] / 2, bg[2] + bg[8] / 2};
  for (int i = 0; i < k; i++) {
    double share[4], child[4];
    stab_share_of<CONT>(geo, bg, k, sup, stk, own, true, sp, sx, i, share);
    const int Si = sup(i);
    stab_com(geo, st, Si, skip, share, child);
    emit(Si, child);
  }
  return true;
}
// One walk task: placed box S receives the virtual stack `vstk` from its parent on some candidate's path.
// Returns 0: the candidate is unstable (S's polygon does not hold the stack), 1: fine, -1: a capacity was exceeded.
template <bool CONT, typename Geo, typename Emit>
PCT_SD int stab_visit(const Geo& geo, const StabState& st, int S, const double vstk[4], bool& ill, Emit emit) {
  const int k = stab_nsup(st, S);
  if (k == 0) return 1;
  const double (*gp)[2] = reinterpret_cast<const double (*)[2]>(st.poly + (size_t)stab_poff(st, S) * 2);
  if (!stab_pip(vstk, gp, stab_npoly(st, S))) return 0;
  double bg[9];
  geo(S, bg);
  StabSup sup{st.ent + stab_soff(st, S)};
  return stab_children<CONT>(geo, st, bg, k, sup, vstk, S, ill, emit) ? 1 : -1;
}
#endif  // host flavour: split and emit in one go
// workspace bytes the hull of a box with up to k supporters needs (points, upper chain, supporter ids)
PCT_HD int stab_ws_need(int k) { return (4 * k) * 16 + (4 * k + 1) * 16 + ((4 * k + 7) & ~7); }
struct StabWsView {
  double (*pts)[2];
  double (*up)[2];
  uint32_t* ids;
};
PCT_SD StabWsView stab_ws_view(unsigned char* base, int kcap) {
  StabWsView v;
  v.pts = reinterpret_cast<double (*)[2]>(base);
  v.up = reinterpret_cast<double (*)[2]>(base + (size_t)(4 * kcap) * 16);
  v.ids = reinterpret_cast<uint32_t*>(base + (size_t)(4 * kcap) * 16 + (size_t)(4 * kcap + 1) * 16);
  return v;
}
// supporters of the box `bg` among the first n placed boxes, in box order: the count, and the first `cap` ids
template <bool CONT, typename Geo>
PCT_SD int stab_find_supporters(const Geo& geo, int n, const double bg[9], uint32_t* ids, int cap) {
  int k = 0;
  for (int i = 0; i < n; i++) {
    double t[9], area[4];
    geo(i, t);
    if (!stab_contact<CONT>(bg, t, area)) continue;
    if (k < cap) ids[k] = (uint32_t)i;
    k++;
  }
  return k;
}
// the stack a candidate starts with: its own centre and mass (D/space.py:38-45; the virtual flavour's `* 1.0`)
PCT_SD void stab_cand_stack(const double cand[9], double density, double cstk[4]) {
  const double sx = cand[6], sy = cand[7], sz = cand[8];
  cstk[0] = cand[0] + sx / 2; cstk[1] = cand[1] + sy / 2; cstk[2] = cand[2] + sz / 2;
  cstk[3] = sx * sy * sz * density * 1.0;
}
// The first half of a candidate's level-0 task (calculated_impact_virtual(first=True), D/space.py:166-267): its hull
// in the workspace (supporters already listed in w.ids[0..k)) and the point-in-polygon test of its own stack.
// 0: unstable, 1: go on to the children, -1: capacity.
template <bool CONT, typename Geo>
PCT_SD int stab_level0_pip(const Geo& geo, const double cand[9], const double cstk[4], int k, const StabWsView& w) {
  StabSup sup{w.ids};
  int nl, nu;
  stab_hull<CONT>(geo, cand, k, sup, w.pts, w.up, nl, nu);
  if (nl + nu > 255) return -1;
  return stab_pip_hull(cstk, w.pts, nl, w.up, nu) ? 1 : 0;
}
#if !defined(__HIPCC__)
// the whole level-0 task.  Same return values and `emit` as stab_visit.
template <bool CONT, typename Geo, typename Emit>
PCT_SD int stab_level0(const Geo& geo, const StabState& st, const double cand[9], double density, int k, const StabWsView& w,
                       bool& ill, Emit emit) {
  if (k == 0) return 1;
  double cstk[4];
  stab_cand_stack(cand, density, cstk);
  const int rc = stab_level0_pip<CONT>(geo, cand, cstk, k, w);
  if (rc != 1) return rc;
  StabSup sup{w.ids};
  return stab_children<CONT>(geo, st, cand, k, sup, cstk, (int)STAB_NOBOX, ill, emit) ? 1 : -1;
}
#endif

// HBM mirror of the per-env stability state (struct-of-arrays over envs; the pool rows are as long as the largest
// capacities of any pass, the LDS copies as long as THIS launch's)
struct StabHbm {
  StabCaps caps;             // this launch's LDS capacities
  int sp_stride, pp_stride;  // HBM row lengths of the share / entry and the polygon pools
  double* stk;     // [N][I][4]
  double* den;     // [N][I]
  double* share;   // [N][sp_stride][4]
  double* poly;    // [N][pp_stride][2]
  uint32_t* meta;  // [N][I][2]
  uint32_t* up;    // [N][I]
  uint32_t* ent;   // [N][sp_stride]
};

// LDS doubles of ONE system of the wave-cooperative least-squares split for up to n supporters: a header (k, the stack's x / y, n
// contact centres), the n fractions, three product columns, U (M x n, column-major), V (n x n, column-major); M = n (n - 1) / 2 + 1
PCT_HD int stab_lsq_rows(int n) { return n * (n - 1) / 2 + 1; }
PCT_HD int stab_lsq_rows8(int n) { return (stab_lsq_rows(n) + 7) & ~7; }  // the product columns are padded to a multiple of 8 rows
PCT_HD size_t stab_lsq_slot_doubles(int n, bool gelsd = false) {
  const size_t M = (size_t)stab_lsq_rows(n), M8 = (size_t)stab_lsq_rows8(n);
  // gelsd mode: the same header and fractions, then the dgelsd workspace (A, b, VT, four n-vectors, 4 n of work); never more than
  // the Jacobi slot from 8 supporters on, i.e. for the classes a launch's workspace is sized by (lsq_n = 8 / 16)
  if (gelsd) return 4 + 2 * (size_t)n + (size_t)n + M * (size_t)n + M + (size_t)n * (size_t)n + 8 * (size_t)n;
  return 4 + 2 * (size_t)n + (size_t)n + 3 * M8 + M * (size_t)n + (size_t)n * (size_t)n;
}
// lanes that share one system: 16 up to 6 supporters (16 rows), 32 up to 8 (29 rows), else the whole wave.  gelsd mode: the classes
// are up to 4 / up to 8 / up to lsq_n supporters, solved by groups of 4 / 8 / 16 lanes (a trailing column, the right-hand side, a
// row of V^T per lane: pct_gelsd.cuh); the widths also tell the classes apart (lanes rk * G ... solve slot rk)
PCT_HD int stab_lsq_group(int k, bool gelsd = false) {
  if (gelsd) return k <= 4 ? 4 : (k <= 8 ? 8 : 16);
  return k <= 6 ? 16 : (k <= 8 ? 32 : 64);
}
PCT_HD int stab_lsq_class_n(int k, int lsq_n, bool gelsd = false) {
  // (gelsd: a class of its own for five and six supporters -- a slot of 218 doubles instead of 417, so eight such systems instead of
  // five share a round's workspace: the c3s1 env whose candidates' walks pass through a box on six supporters 33 times in one step)
  if (gelsd) return k <= 4 ? 4 : (k <= 6 ? 6 : (k <= 8 ? 8 : lsq_n));
  return k <= 6 ? 6 : (k <= 8 ? 8 : lsq_n);
}
// workspace bytes: one system of up to n supporters (narrow: the normal pass, where LDS is what bounds the resident envs --
// that still takes two systems of the 6-supporter class), or (wide: the retry pass) as many systems as a class's lane groups
// allow -- four of the 6-supporter class, two of the 8-supporter class
PCT_HD size_t stab_lsq_bytes(int n, bool wide) {
  if (n <= 0) return 0;
  size_t d = stab_lsq_slot_doubles(n);
  if (wide) {
    if (4 * stab_lsq_slot_doubles(6) > d) d = 4 * stab_lsq_slot_doubles(6);
    if (2 * stab_lsq_slot_doubles(8) > d) d = 2 * stab_lsq_slot_doubles(8);
  }
  return sizeof(double) * d;
}

#if defined(__HIPCC__)
// ---- the wave-cooperative least-squares split -------------------------------------------------------------------------
// np.linalg.lstsq of the >= 3-supporter system (D/space.py:134-152) for k supporters, by the wave: the one-sided Jacobi SVD of
// stab_lstsq / lstsq_min_norm, operation for operation -- the column sums run over the rows in order (every lane of a system
// adds the same M products, read back from LDS), rotations and the final back-substitution use the same expressions -- with
// the rows of U spread over the lanes and the matrix in LDS instead of one lane's private arrays in scratch memory.  Several
// INDEPENDENT systems of one size class are solved side by side, one per group of G lanes (64 / G of them): the same
// instruction stream, every group on its own slot of the workspace, a group whose system has converged sitting out.
// Slot s (at ws + s * slot doubles, laid out for `n` supporters): [0] = k (0: idle slot), [1], [2] = the stack's x, y,
// [4 + 2 i], [5 + 2 i] = contact centre of supporter i.  Out: [4 + 2 n + i] = the fraction of supporter i.  All 64 lanes call
// (wave-uniform G, n, nslot = systems the workspace holds, <= 64 / G).  Returns, per lane, the notice of its group's system.
__device__ __forceinline__ bool stab_lsq_wave(double* ws, int G, int n, int nslot, int lane) {
  const int gl = lane & (G - 1);
  const int grp = lane / G < nslot ? lane / G : 0;  // (a lane group beyond the workspace's slots sits out)
  double* w = ws + (size_t)grp * stab_lsq_slot_doubles(n);
  const int k = lane / G < nslot ? (int)w[0] : 0;
  const int M = stab_lsq_rows(k), M8 = stab_lsq_rows8(k);
  const double* in = w + 4;
  double* x = w + 4 + 2 * n;
  double* S = x + n;                                 // [3][M8] products of a column pair (zero beyond row M), later proj / keep
  double* U = S + 3 * (size_t)stab_lsq_rows8(n);     // [k][M]
  double* V = U + (size_t)stab_lsq_rows(n) * n;      // [k][k]
  const double s0 = w[1], s1 = w[2];
  int kmax = k;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int o = __shfl_xor(kmax, off, 64);
    kmax = o > kmax ? o : kmax;
  }
  // the system: one row per supporter pair (i < j, in that order), a closing row of ones
  if (k > 0)
    for (int r = gl; r < M; r += G) {
      int i = 0, base = 0;
      while (r < M - 1 && r >= base + (k - 1 - i)) { base += k - 1 - i; i++; }
      const int j = i + 1 + (r - base);
      double rr = 0;
      bool row_on = false;
      if (r < M - 1) {
        const double ei0 = in[2 * i], ei1 = in[2 * i + 1], ej0 = in[2 * j], ej1 = in[2 * j + 1];
        const double t0 = ei0 - ej0, t1 = ei1 - ej1;
        const double mol = stab_dot2(s0 - ei0, s1 - ei1, t0, t1);
        if (mol != 0) { rr = fabs(stab_dot2(s0 - ej0, s1 - ej1, t0, t1)) / mol; row_on = true; }
      }
      for (int c = 0; c < k; c++) {
        double v = 0;
        if (r == M - 1) v = 1;
        else if (row_on) v = c == i ? 1.0 : (c == j ? -rr : 0.0);
        U[(size_t)c * M + r] = v;
      }
    }
  for (int q = gl; q < k * k; q += G) V[q] = (q / k == q % k) ? 1.0 : 0.0;
  __syncthreads();
  bool done = k == 0;  // this group's system has converged (the same in every lane of a group)
  for (int sweep = 0; sweep < 60 && __ballot(!done); sweep++) {
    bool rotated = false;
    for (int p = 0; p < kmax; p++)
      for (int q = p + 1; q < kmax; q++) {
        const bool on = !done && q < k;
        if (on)
          for (int r = gl; r < M8; r += G) {
            const double up = r < M ? U[(size_t)p * M + r] : 0.0, uq = r < M ? U[(size_t)q * M + r] : 0.0;
            S[r] = up * up; S[M8 + r] = uq * uq; S[2 * M8 + r] = up * uq;
          }
        __syncthreads();
        // the three column sums, over the rows IN ORDER (every lane of the group the same additions); eight rows' products are
        // fetched at a time -- the rows beyond M hold +0.0, which leaves a sum that started at +0.0 unchanged
        double alpha = 0, beta = 0, gamma = 0;
        if (on)
          for (int r = 0; r < M8; r += 8) {
            double a[8], b[8], g[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { a[u] = S[r + u]; b[u] = S[M8 + r + u]; g[u] = S[2 * M8 + r + u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) { alpha += a[u]; beta += b[u]; gamma += g[u]; }
          }
        __syncthreads();
        const bool rot = on && !(gamma == 0 || fabs(gamma) <= 2.220446049250313e-16 * sqrt(alpha * beta));
        if (__ballot(rot)) {
          rotated = rotated || rot;
          const double zeta = (beta - alpha) / (2 * gamma);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
          const double c = 1 / sqrt(1 + t * t), sn = c * t;
          if (rot) {
            for (int r = gl; r < M; r += G) {
              const double up = U[(size_t)p * M + r], uq = U[(size_t)q * M + r];
              U[(size_t)p * M + r] = c * up - sn * uq;
              U[(size_t)q * M + r] = sn * up + c * uq;
            }
            if (gl < k) {
              const double vp = V[p * k + gl], vq = V[q * k + gl];
              V[p * k + gl] = c * vp - sn * vq;
              V[q * k + gl] = sn * vp + c * vq;
            }
          }
          __syncthreads();
        }
      }
    if (!rotated) done = true;
  }
  // singular values: lane j of a group sums column j over the rows, in order
  double s2 = 0;
  if (gl < k)
    for (int r = 0; r < M; r++) { const double u = U[(size_t)gl * M + r]; s2 += u * u; }
  double smax2 = gl < k ? s2 : 0.0;
  for (int off = G >> 1; off >= 1; off >>= 1) {  // (within the group: the partner of lane l is l ^ off)
    const double o = __shfl_xor(smax2, off, 64);
    smax2 = o > smax2 ? o : smax2;
  }
  const double rc = 2.220446049250313e-16 * (double)(M > k ? M : k);
  const double sj = sqrt(s2), cut = rc * sqrt(smax2);
  const bool ill = gl < k && s2 > 0 && sj > cut / STAB_ILL_BAND && sj < cut * STAB_ILL_BAND;
  const bool keep = gl < k && !(s2 <= 0 || sj <= cut);
  if (gl < k) {
    // proj = sum_r U[r][j] b[r] with b = e_{M-1}: zeros, then the last row's entry
    double proj = 0.0;
    proj += U[(size_t)gl * M + (M - 1)] * 1.0;
    proj /= s2;
    S[gl] = keep ? proj : 0.0;
    S[M8 + gl] = keep ? 1.0 : 0.0;
  }
  __syncthreads();
  if (gl < k) {
    double xi = 0;
    for (int j = 0; j < k; j++)
      if (S[M8 + j] != 0.0) xi += V[j * k + gl] * S[j];
    x[gl] = xi;
  }
  __syncthreads();
  // the notice of a group: any of its lanes
  const uint64_t im = __ballot(ill);
  const uint64_t gm = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << (lane & ~(G - 1));
  return (im & gm) != 0;
}
// one system's slot as stab_lsq_wave reads it: k, the stack's x / y, the contact centres of the box's supporters (one lane writes)
template <bool CONT, typename Geo, typename Sup>
__device__ __forceinline__ void stab_lsq_inputs(const Geo& geo, const double bg[9], int k, const Sup& sup, const double stk[4], double* slot) {
  slot[0] = (double)k;
  slot[1] = stk[0];
  slot[2] = stk[1];
  for (int i = 0; i < k; i++) {
    double a[4];
    stab_area<CONT>(geo, bg, sup(i), a);
    slot[4 + 2 * i] = (a[0] + a[2]) / 2;
    slot[5 + 2 * i] = (a[1] + a[3]) / 2;
  }
}

// gelsd mode: the systems of slots 0 .. nslot - 1 (laid out for class n: stab_lsq_slot_doubles(n, true) doubles each, the header
// as stab_lsq_wave reads it), each solved by a GROUP of G lanes -- lanes s * G .. s * G + G - 1 take slot s -- with pct_gelsd.cuh on
// the slot's own workspace (round 4: one lane per system).  All 64 lanes call; returns the system's notice in its group's lanes.
// (gslot: the slot of this lane's group -- contiguous slots of the workspace, or the hull-workspace slices of the level-0 candidates
// themselves, PCT_STAB_OWN_SLICE; gact: the group has a system)
__device__ __forceinline__ bool stab_gelsd_slots(double* gslot, bool gact, int G, int n, int lane, bool avx2) {
  bool ill = false;
  if (gact) {
    double* w = gslot;
    const int k = (int)w[0];
    const gelsd::Grp g = {lane & (G - 1), G};
    if (k >= 3) gelsd::split_t(g, w + 4 + 3 * n, k, w + 4, w[1], w[2], StabDot2{avx2}, w + 4 + 2 * n, ill, avx2);
  }
  __syncthreads();
  return ill;
}

// ---- the wave-cooperative driver ----------------------------------------------------------------------------------
// LDS workspace of a wave: the hull workspace (shared by the level-0 tasks of a round) and the task queue.
struct StabWave {
  double* lsq;   // workspace of the cooperative least-squares split (caps.lsq_bytes)
  int lsq_n, lsq_doubles;
  bool gelsd, gelsd_avx2;  // caps.gelsd != 0 / == 2
  int lsq_mode;            // caps.gelsd
  unsigned char* hull;
  int hull_bytes;
  uint32_t* ctl;    // [6] queue count, failed candidates (lanes 0..31, 32..63), global capacity error (STAB_WHY_*),
                    //     candidates with a capacity error of their own (lanes 0..31, 32..63)
  uint32_t* qmeta;  // [qcap] candidate lane | S << 6
  double* qstk;     // [qcap][4]
  int qcap;
};
PCT_HD size_t stab_wave_bytes(const StabCaps& c) {
  return (size_t)c.lsq_bytes + (size_t)((c.ws_bytes + 7) & ~7) + (size_t)c.queue * (4 * sizeof(double) + sizeof(uint32_t)) + 32;
}
// carve: the least-squares workspace and the queue's doubles first (8-byte aligned base), then the hull workspace, then the words
PCT_SD StabWave stab_wave_carve(unsigned char* base, const StabCaps& c) {
  StabWave w;
  w.lsq = reinterpret_cast<double*>(base);
  w.lsq_n = c.lsq_n;
  w.lsq_doubles = c.lsq_bytes / (int)sizeof(double);
  w.gelsd = c.gelsd != 0;
  w.gelsd_avx2 = c.gelsd == 2;
  w.lsq_mode = c.gelsd;
  base += (size_t)c.lsq_bytes;
  // the hull workspace follows the least-squares workspace directly: in the passes of a round that work on popped tasks (whose
  // supporter lists are pool entries) it is idle, and further systems of the small size classes are solved in it
  w.hull = base;
  w.hull_bytes = c.ws_bytes;
  base += (size_t)((c.ws_bytes + 7) & ~7);
  w.qstk = reinterpret_cast<double*>(base);
  w.qmeta = reinterpret_cast<uint32_t*>(base + (size_t)c.queue * 4 * sizeof(double));
  w.ctl = w.qmeta + c.queue;
  w.qcap = c.queue;
  return w;
}
// (a box on more than STAB_NSUP_MAX supporters gets a class no workspace holds: a capacity error)
__device__ __forceinline__ int stab_wave_incl_sum(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}
// slots of the walk queue kept free for the depth-first descent of a single candidate (see stab_virtual_wave)
PCT_HD int stab_queue_reserve(int qcap) { return qcap / 2 < 96 ? qcap / 2 : 96; }
PCT_SD int stab_class(int k) { return k <= 2 ? 2 : (k <= 8 ? 8 : (k <= 32 ? 32 : (k <= STAB_NSUP_MAX ? STAB_NSUP_MAX : (1 << 20)))); }

// calculated_impact_virtual(first=True) of up to 64 candidates at once: lane = candidate (`need`: this lane carries
// one, geometry `cand`, resting on boxes -- the caller has dealt with the floor case).  Returns the lane's verdict.
// All 64 lanes must call.  Rounds: the pending candidates of one supporter-count class share the hull workspace for
// their level-0 tasks (as many as fit), then the queue is drained -- 64 tasks per pass, last in first out -- before
// the next round starts.  cap_err: STAB_WHY_* bits of a capacity of the WAVE (the queue) that was exceeded -- every
// verdict is then void.  lane_err: a capacity of THIS lane's candidate (hull workspace, hull vertices, more than
// STAB_LSQ supporters in a least-squares split) was exceeded and the candidate was not found unstable elsewhere: its
// verdict is unknown (the reference, which has no such limits, evaluates the visits of a walk one after the other and
// stops at the first failure; here they are evaluated side by side, so a limit hit on a branch of a candidate that
// fails anyway must not count).
template <bool CONT, typename Geo>
__device__ __forceinline__ bool stab_virtual_wave(const Geo& geo, const StabState& st, int n, bool need, const double cand[9],
                                                  double density, const StabWave& w, int lane, uint32_t& cap_err, bool& lane_err, bool& ill,
                                                  StabStats* ss = nullptr) {
  // supporter count of every candidate; the first two ids stay in registers (the common class needs no second scan)
  int k = 0;
  uint32_t id0 = 0, id1 = 0;
  if (need) {
    for (int i = 0; i < n; i++) {
      double t[9], area[4];
      geo(i, t);
      if (!stab_contact<CONT>(cand, t, area)) continue;
      id0 = k == 0 ? (uint32_t)i : id0;
      id1 = k == 1 ? (uint32_t)i : id1;
      k++;
    }
  }
  if (!__ballot(need && k > 0)) return true;
  if (ss && lane == 0) ss->v_calls++;
  double cstk[4];
  stab_cand_stack(cand, density, cstk);
  lane_err = false;
  if (lane < 6) w.ctl[lane] = 0;
  __syncthreads();
  bool pending = need && k > 0;
  const int mycls = stab_class(k);
#ifndef PCT_STAB_INTERLEAVE
#define PCT_STAB_INTERLEAVE 1 /* round 5, discrete env: the level-0 rounds of a call run back to back as far as the queue has room, and
                                 their subtrees are drained TOGETHER -- walk tasks of different rounds share passes (each pass is a latency
                                 chain, each solve round a solve's latency): c1 +2.8 %.  0 (and always in the continuous env: c3s1 -15 %
                                 with it, also when capped to a half / a third of the queue's room -- profiles/r05_experiments.txt
                                 item 9): every round drains its own subtree before the next one starts */
#endif
  constexpr bool kInterleave = PCT_STAB_INTERLEAVE && !CONT;
  while (true) {
    const uint64_t pm = __ballot(pending);
    // tasks earlier rounds have left in the queue (wave-uniform; 0 when every round drains its own)
    const int qn0 = kInterleave ? (int)(w.ctl[0] < (uint32_t)w.qcap ? w.ctl[0] : (uint32_t)w.qcap) : 0;
    if (!pm && qn0 == 0) break;
    int kc = 2, per = 1, rk = 0;
    bool act = false;
    if (pm) {
      const int first = __ffsll((unsigned long long)pm) - 1;
      kc = __builtin_amdgcn_readlane(mycls, first);
      per = kc <= STAB_NSUP_MAX ? stab_ws_need(kc) : 0x7FFFFFFF;
      const int fit = w.hull_bytes / per;
      const bool member = pending && mycls == kc;
      const uint64_t mm = __ballot(member);
      rk = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
      act = member && rk < fit;
      if (fit == 0) {  // this class does not fit the workspace at all (wave-uniform)
        if (member) atomicOr(&w.ctl[4 + (lane >> 5)], 1u << (lane & 31));
        pending = pending && !member;
        continue;
      }
      // ... and only as many candidates as the queue takes children of: the first ones whose supporter counts add up to the room it
      // has left (all of it up to the reserve when it is empty)
      const int cum = stab_wave_incl_sum(act ? k : 0, lane);
      act = act && (qn0 + cum <= w.qcap - stab_queue_reserve(w.qcap) || (qn0 == 0 && lane == first && k <= w.qcap));
      if (!__ballot(act)) {
        if (qn0 == 0) {  // a single candidate's children do not fit an empty queue
          if (lane == first) atomicOr(&w.ctl[3], STAB_WHY_QUEUE);
          pending = pending && !member;
          continue;
        }
        // (no room beside what the earlier rounds have queued: this pass only pops; the round starts once the queue has drained)
      }
      pending = pending && !act;
    }
    // what the lane examines in this pass: a candidate (level 0) or, further down, a popped task
    bool have = false;
    double bg[9], stk[4] = {0, 0, 0, 0};
    int kk = 0, skip = (int)STAB_NOBOX, cl = lane;
    const uint32_t* supw = w.qmeta;
#ifndef PCT_STAB_OWN_SLICE
#define PCT_STAB_OWN_SLICE 1
#endif
    // the candidate's own slice of the hull workspace: once its hull has been tested only the supporter ids at the slice's end are
    // still read, and the front takes the candidate's least-squares system (gelsd mode, up to four supporters: below)
    double* const lvl0_slice = reinterpret_cast<double*>(w.hull + (size_t)(act ? rk : 0) * (act ? per : 0));
    const bool slice_takes_system = PCT_STAB_OWN_SLICE && (size_t)kc * 128 + 16 >= sizeof(double) * stab_lsq_slot_doubles(4, true);  // (ids begin at 128 kc + 16)
    if (act) {
      StabWsView v = stab_ws_view(w.hull + (size_t)rk * per, kc);
      if (kc == 2) { v.ids[0] = id0; v.ids[1] = id1; }
      else stab_find_supporters<CONT>(geo, n, cand, v.ids, kc);
      const int rc = stab_level0_pip<CONT>(geo, cand, cstk, k, v);
      if (rc == 0) atomicOr(&w.ctl[1 + (lane >> 5)], 1u << (lane & 31));
      if (rc < 0) atomicOr(&w.ctl[4 + (lane >> 5)], 1u << (lane & 31));
      have = rc == 1;
      if (ss) ss->v_level0++;
      kk = k;
      supw = v.ids;
#pragma unroll
      for (int c = 0; c < 9; c++) bg[c] = cand[c];
      stk[0] = cstk[0]; stk[1] = cstk[1]; stk[2] = cstk[2]; stk[3] = cstk[3];
    }
    if (ss && lane == 0) ss->v_rounds++;
    // (the first pass of a round examines the candidates themselves: their supporter ids are in the hull workspace; a pass that only
    // pops -- the last candidates have had their round, or the queue has no room for the next one yet -- leaves it idle)
    bool hull_idle = !__ballot(act);
    while (true) {
      {
        // how the box under examination splits its stack over its supporters; a split over six and more goes to the wave
        StabSplit sp;
        StabSplitX sx;
        sp.mode = 0;
        StabSup sup{supw};
        if (have && !stab_split<CONT>(geo, bg, kk, sup, stk, sp, sx, ss, w.lsq_n, w.lsq_mode)) {
          atomicOr(&w.ctl[4 + (cl >> 5)], 1u << (cl & 31));
          have = false;
        }
        // the splits for the wave, one size class at a time, as many systems side by side as the class's lane groups allow
        for (uint64_t cm = __ballot(have && sp.mode == 4); cm;) {
          const int first = __ffsll((unsigned long long)cm) - 1;
          const int kf = __builtin_amdgcn_readlane(kk, first);
          const int G = stab_lsq_group(kf, w.gelsd), cn = stab_lsq_class_n(kf, w.lsq_n, w.gelsd);
          const size_t sd = stab_lsq_slot_doubles(cn, w.gelsd);
          const bool mine = ((cm >> lane) & 1ull) && stab_lsq_class_n(kk, w.lsq_n, w.gelsd) == cn;
          const uint64_t mm = __ballot(mine);
          const int rk = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
          // gelsd mode, the first pass of a round (the candidates themselves; the hull workspace is not idle, so the shared workspace
          // holds 4 systems of the <= 4-supporter class): every such candidate solves in its own hull slice -- 16 systems per solve
          // round instead of 4.  The slot addresses travel through the first words of the shared workspace.
          const bool own = w.gelsd && !hull_idle && cn == 4 && slice_takes_system;  // (wave-uniform)
          // systems of this class the workspace holds (with the hull workspace behind it when that is idle), at most one per lane group
          int nslot = (w.lsq_doubles + (hull_idle ? w.hull_bytes / (int)sizeof(double) : 0)) / (int)sd;
          if (own) nslot = 64;
          nslot = nslot < 64 / G ? nslot : 64 / G;
          const bool sel = mine && rk < nslot;
          double* const myslot = own ? lvl0_slice : w.lsq + (size_t)rk * sd;
          if (!own && lane < nslot) w.lsq[(size_t)lane * sd] = 0.0;  // idle slots
          __syncthreads();
          if (sel) stab_lsq_inputs<CONT>(geo, bg, kk, sup, stk, myslot);
          if (own && sel) reinterpret_cast<uint32_t*>(w.lsq)[rk] = (uint32_t)(reinterpret_cast<unsigned char*>(myslot) - reinterpret_cast<unsigned char*>(w.lsq));
          __syncthreads();
          if (ss && lane == 0) { ss->lsq_rounds++; ss->lsq_rounds_l0 += hull_idle ? 0 : 1; }
          const int grp = lane / G;
          bool gact = grp < nslot;
          double* gslot = w.lsq + (size_t)(gact ? grp : 0) * sd;
          if (own) {
            gact = grp < __popcll(__ballot(sel));
            gslot = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(w.lsq) + (gact ? reinterpret_cast<const uint32_t*>(w.lsq)[grp] : 0u));
          }
          const bool gill = w.gelsd ? stab_gelsd_slots(gslot, gact, G, cn, lane, w.gelsd_avx2) : stab_lsq_wave(w.lsq, G, cn, nslot, lane);
          // a system's notice comes back in its group's lanes: fetch the one of this lane's slot
          const uint64_t illm = __ballot(gill);
          if (sel) {
            const double* slot = myslot;
            if (kk <= 5) {  // (gelsd mode only: up to five fractions live in registers, as after the per-lane solves)
              sp.f[0] = slot[4 + 2 * cn]; sp.f[1] = slot[5 + 2 * cn]; sp.f[2] = slot[6 + 2 * cn];
              sp.f[3] = kk > 3 ? slot[7 + 2 * cn] : 0.0;
              sp.f[4] = kk > 4 ? slot[8 + 2 * cn] : 0.0;
            } else {
              for (int i = 0; i < kk; i++) sx.fx[i] = slot[4 + 2 * cn + i];
            }
            sp.mode = 3;
            sp.ill = (illm >> (rk * G)) & 1ull;
          }
          __syncthreads();
          cm &= ~__ballot(sel);
        }
        if (have) {
          ill = ill || sp.ill;
          auto emit = [&](int Si, const double child[4]) __attribute__((always_inline)) {
            const uint32_t pos = atomicAdd(&w.ctl[0], 1u);
            if (pos < (uint32_t)w.qcap) {
              w.qmeta[pos] = (uint32_t)cl | ((uint32_t)Si << 6);
              w.qstk[(size_t)pos * 4 + 0] = child[0]; w.qstk[(size_t)pos * 4 + 1] = child[1];
              w.qstk[(size_t)pos * 4 + 2] = child[2]; w.qstk[(size_t)pos * 4 + 3] = child[3];
            } else {
              atomicOr(&w.ctl[3], STAB_WHY_QUEUE);
            }
          };
          stab_children_emit<CONT>(geo, st, bg, kk, sup, stk, skip, sp, sx, emit);
        }
      }
      hull_idle = true;
      if (ss && lane == 0) ss->v_passes++;
      __syncthreads();
      const uint32_t qraw = w.ctl[0];
      const int qn = (int)(qraw < (uint32_t)w.qcap ? qraw : (uint32_t)w.qcap);
      if (qn == 0 || w.ctl[3]) break;
      if (kInterleave) {
        // candidates that have not had their level-0 pass yet come first, as long as the first of them finds room for its children
        // (the same test the admission above applies): their tasks then share the passes that follow with the ones queued already
        const uint64_t pm2 = __ballot(pending);
        if (pm2) {
          const int k1 = __builtin_amdgcn_readlane(k, __ffsll((unsigned long long)pm2) - 1);
          if (qn + k1 <= w.qcap - stab_queue_reserve(w.qcap)) break;
        }
      }
      const uint64_t failed = (((uint64_t)w.ctl[2] << 32) | w.ctl[1]) | (((uint64_t)w.ctl[5] << 32) | w.ctl[4]);
      // pop from the end: lane j looks at task qn - 1 - j.  As many tasks are taken as the queue can then hold the
      // children of (every task of box S pushes nsup(S)): the longest prefix of lanes with
      // (tasks left) + (children so far) <= capacity -- so the queue cannot overflow unless one task alone does
      const int ti = qn - 1 - lane;
      uint32_t meta = 0;
      int kS = 0;
      if (ti >= 0) {
        meta = w.qmeta[ti];
        kS = stab_nsup(st, (int)(meta >> 6));
        if ((failed >> (meta & 63u)) & 1ull) kS = 0;  // tasks of a candidate that has already failed push nothing
      }
      const int cum = stab_wave_incl_sum(kS, lane);
      // WIDE pops fill the queue only up to (capacity - reserve).  When not even the last task can be popped that way the
      // wave goes NARROW: one task per pass, the last one -- a plain depth-first descent into that candidate's subtree,
      // which needs (depth x fan-out) slots at most and finds them in the reserve; the wide passes resume once it is done.
      const bool fits = ti >= 0 && (qn - (lane + 1)) + cum <= w.qcap - stab_queue_reserve(w.qcap);
      const uint64_t fm = __ballot(fits);
      int take = (~fm) ? __ffsll((unsigned long long)~fm) - 1 : 64;
      if (take == 0 && (qn - 1) + __builtin_amdgcn_readfirstlane(kS) <= w.qcap) take = 1;
      if (take == 0) {  // the last task alone does not fit
        __syncthreads();
        if (lane == 0) atomicOr(&w.ctl[3], STAB_WHY_QUEUE);
        __syncthreads();
        break;
      }
      const bool has = lane < take;
      if (ss && lane == 0) { ss->v_tasks += take; ss->v_narrow += take == 1; }
      if (has) {
        stk[0] = w.qstk[(size_t)ti * 4 + 0]; stk[1] = w.qstk[(size_t)ti * 4 + 1];
        stk[2] = w.qstk[(size_t)ti * 4 + 2]; stk[3] = w.qstk[(size_t)ti * 4 + 3];
      }
      __syncthreads();
      if (lane == 0) w.ctl[0] = (uint32_t)(qn - take);
      __syncthreads();
      cl = (int)(meta & 63u);
      const int S = (int)(meta >> 6);
      have = false;
      if (has && !((failed >> cl) & 1ull)) {
        kk = stab_nsup(st, S);
        if (kk > 0) {
          const double (*gp)[2] = reinterpret_cast<const double (*)[2]>(st.poly + (size_t)stab_poff(st, S) * 2);
          if (!stab_pip(stk, gp, stab_npoly(st, S))) atomicOr(&w.ctl[1 + (cl >> 5)], 1u << (cl & 31));
          else {
            have = true;
            geo(S, bg);
            supw = st.ent + stab_soff(st, S);
            skip = S;
          }
        }
      }
    }
    if (w.ctl[3]) break;
  }
  __syncthreads();
  const uint64_t failed = ((uint64_t)w.ctl[2] << 32) | w.ctl[1];
  const uint64_t unknown = (((uint64_t)w.ctl[5] << 32) | w.ctl[4]) & ~failed;
  cap_err |= w.ctl[3];
  lane_err = (unknown >> lane) & 1ull;
  __syncthreads();
  return !((failed >> lane) & 1ull) && !lane_err;
}
// HBM -> LDS: the used part of env e's state (n placed boxes, n_ent pool entries, n_poly vertices).  False: it does not
// fit this launch's pools (the env belongs to the retry pass).
__device__ __forceinline__ bool stab_load(const StabHbm& hb, int I, int e, int n, int n_ent, int n_poly, StabState& st, int lane) {
  st.n_ent = n_ent;
  st.n_poly = n_poly;
  if (n_ent > st.SP || n_poly > st.PP) return false;
  const double* gs = hb.stk + (size_t)e * I * 4;
  for (int i = lane; i < n * 4; i += 64) st.stk[i] = gs[i];
  const double* gd = hb.den + (size_t)e * I;
  for (int i = lane; i < n; i += 64) st.den[i] = gd[i];
  const uint32_t* gm = hb.meta + (size_t)e * I * 2;
  for (int i = lane; i < n * 2; i += 64) st.meta[i] = gm[i];
  const uint32_t* gu = hb.up + (size_t)e * I;
  for (int i = lane; i < n; i += 64) st.up[i] = gu[i];
  const uint32_t* ge = hb.ent + (size_t)e * hb.sp_stride;
  for (int i = lane; i < n_ent; i += 64) st.ent[i] = ge[i];
  const double* gh = hb.share + (size_t)e * hb.sp_stride * 4;
  for (int i = lane; i < n_ent * 4; i += 64) st.share[i] = gh[i];
  const double* gp = hb.poly + (size_t)e * hb.pp_stride * 2;
  for (int i = lane; i < n_poly * 2; i += 64) st.poly[i] = gp[i];
  return true;
}
// LDS -> HBM.  `from_box`: boxes below it were already in HBM when the launch began -- of those only what a commit
// walk may have rewritten goes back (stacks, meta words, up-list words, entry words and shares: everything but the
// polygons, which never change once stored).
__device__ __forceinline__ void stab_store(const StabHbm& hb, int I, int e, int n, const StabState& st, int poly_from, int lane) {
  double* gs = hb.stk + (size_t)e * I * 4;
  for (int i = lane; i < n * 4; i += 64) gs[i] = st.stk[i];
  double* gd = hb.den + (size_t)e * I;
  for (int i = lane; i < n; i += 64) gd[i] = st.den[i];
  uint32_t* gm = hb.meta + (size_t)e * I * 2;
  for (int i = lane; i < n * 2; i += 64) gm[i] = st.meta[i];
  uint32_t* gu = hb.up + (size_t)e * I;
  for (int i = lane; i < n; i += 64) gu[i] = st.up[i];
  uint32_t* ge = hb.ent + (size_t)e * hb.sp_stride;
  for (int i = lane; i < st.n_ent; i += 64) ge[i] = st.ent[i];
  double* gh = hb.share + (size_t)e * hb.sp_stride * 4;
  for (int i = lane; i < st.n_ent * 4; i += 64) gh[i] = st.share[i];
  double* gp = hb.poly + (size_t)e * hb.pp_stride * 2;
  for (int i = poly_from * 2 + lane; i < st.n_poly * 2; i += 64) gp[i] = st.poly[i];
}
#endif  // __HIPCC__

// ---- the commit ---------------------------------------------------------------------------------------------------
// calculated_impact() of the box just placed as id `n` (geometry already visible through geo(n, .)): records its
// supporters / polygon / stack, propagates the shares downward and re-checks every box on the way (D/space.py:73-164).
// The walk is order dependent and sequential.  `ws` / `ws_bytes`: workspace for the hull, then for the depth-first stack.
// Returns 1: stable, 0: unstable, -1: a capacity (pools, workspace, supporters) was exceeded.
//
// First half: the box's own record.  1: it stands on the floor (accepted without a walk), 2: the walk follows, -1: capacity.
template <bool CONT, typename Geo>
PCT_SD int stab_commit_record(const Geo& geo, StabState& st, int n, double density, unsigned char* ws, int ws_bytes) {
  double bg[9];
  geo(n, bg);
  // supporters: as many ids as the workspace can take a hull for
  int kcap = 1;
  while (kcap < STAB_NSUP_MAX && stab_ws_need(kcap + 1) <= ws_bytes) kcap++;
  if (stab_ws_need(kcap) > ws_bytes) return -1;
  StabWsView w = stab_ws_view(ws, kcap);
  const int k = stab_find_supporters<CONT>(geo, n, bg, w.ids, kcap);
  if (k > kcap || k > STAB_NSUP_MAX) return -1;
  if (st.n_ent + k > st.SP || st.n_ent + k >= (int)STAB_END) return -1;
  st.den[n] = density;
  {
    double sx = bg[6], sy = bg[7], sz = bg[8];
    double* s = st.stk + (size_t)n * 4;
    s[0] = bg[0] + sx / 2; s[1] = bg[1] + sy / 2; s[2] = bg[2] + sz / 2; s[3] = sx * sy * sz * density;
  }
  const int soff = st.n_ent, poff = st.n_poly;
  st.up[n] = STAB_END | (STAB_END << 16);
  for (int i = 0; i < k; i++) {  // one pool entry per supporter, chained at the tail of that supporter's up-list
    const int S = (int)w.ids[i];
    const uint32_t j = (uint32_t)(soff + i);
    st.ent[j] = (uint32_t)S | ((uint32_t)n << 10) | (STAB_END << 20);
    const uint32_t ht = st.up[S];
    const uint32_t head = ht & 0xFFFFu, tail = ht >> 16;
    if (head == STAB_END) st.up[S] = j | (j << 16);
    else {
      st.ent[tail] = (st.ent[tail] & 0xFFFFFu) | (j << 20);
      st.up[S] = head | (j << 16);
    }
  }
  st.n_ent += k;
  int np = 0;
  if (k > 0) {
    StabSup sup{w.ids};
    int nl, nu;
    stab_hull<CONT>(geo, bg, k, sup, w.pts, w.up, nl, nu);
    np = nl + nu;
    if (np > 255 || st.n_poly + np > st.PP || st.n_poly + np > 0xFFFF) return -1;
    double cx, cy;
    stab_hull_centroid(w.pts, nl, w.up, nu, cx, cy);
    for (int i = 0; i < np; i++) {
      double vx, vy;
      stab_hull_vertex(w.pts, nl, w.up, i, cx, cy, vx, vy);
      st.poly[(size_t)(poff + i) * 2 + 0] = vx;
      st.poly[(size_t)(poff + i) * 2 + 1] = vy;
    }
    st.n_poly += np;
  }
  st.meta[2 * n] = (uint32_t)k | (0u << 8) | ((uint32_t)np << 16);
  st.meta[2 * n + 1] = (uint32_t)soff | ((uint32_t)poff << 16);
  if (CONT ? (fabs(bg[2]) < 1e-6) : (bg[2] == 0)) return 1;  // max_h == 0: check_box returns first (:448-449)
  return 2;
}
// A box of the walk hands its stack down, once its split is known: up_edges[self] = share; calculate_new_com() for every supporter
template <bool CONT, typename Geo>
PCT_SD void stab_commit_expand(const Geo& geo, StabState& st, int id, int kk, const StabSup& sup, const double g[9], const double stk[4],
                               const StabSplit& sp, const StabSplitX& sx) {
  const int alias_k = sp.mode == 0 ? 0 : (sp.mode == 1 ? sp.direct : -1);
  st.meta[2 * id] = (st.meta[2 * id] & 0xFFFF00FFu) | ((uint32_t)(alias_k + 1) << 8);
  const double own[3] = {stk[0], stk[1], stk[2]};
  for (int i = 0; i < kk; i++) {
    double sh[4];
    stab_share_of<CONT>(geo, g, kk, sup, stk, own, false, sp, sx, i, sh);
    double* e = st.share + (size_t)(stab_soff(st, id) + i) * 4;
    e[0] = sh[0]; e[1] = sh[1]; e[2] = sh[2]; e[3] = sh[3];
    const int Si = sup(i);
    double com[4];
    stab_com(geo, st, Si, (int)STAB_NOBOX, (const double*)0, com);
    double* t = st.stk + (size_t)Si * 4;
    t[0] = com[0]; t[1] = com[1]; t[2] = com[2]; t[3] = com[3];
  }
}
#if !defined(__HIPCC__)
// host flavour (tests/host/stab_host.cpp): one thread, every split solved in line
template <bool CONT, typename Geo>
PCT_SD int stab_commit(const Geo& geo, StabState& st, int n, double density, unsigned char* ws, int ws_bytes, bool& ill,
                       StabStats* ss = nullptr) {
  const int rec = stab_commit_record<CONT>(geo, st, n, density, ws, ws_bytes);
  if (rec != 2) return rec;
  // explicit depth-first walk; frame = box id | next supporter << 16
  uint32_t* frames = reinterpret_cast<uint32_t*>(ws);
  const int fcap = ws_bytes / 4;
  int depth = 1;
  frames[0] = (uint32_t)n;
  while (depth > 0) {
    const int d = depth - 1;
    const int id = (int)(frames[d] & 0xFFFFu);
    const int next = (int)(frames[d] >> 16);
    const int kk = stab_nsup(st, id);
    if (kk == 0) { depth--; continue; }
    StabSup sup{st.ent + stab_soff(st, id)};
    if (next == 0) {
      const double (*gp)[2] = reinterpret_cast<const double (*)[2]>(st.poly + (size_t)stab_poff(st, id) * 2);
      const double stk[4] = {st.stk[(size_t)id * 4 + 0], st.stk[(size_t)id * 4 + 1], st.stk[(size_t)id * 4 + 2],
                             st.stk[(size_t)id * 4 + 3]};
      if (!stab_pip(stk, gp, stab_npoly(st, id))) return 0;
      if (ss) ss->commit_visits++;
      double g[9];
      geo(id, g);
      StabSplit sp;
      StabSplitX sx;
      if (!stab_split<CONT>(geo, g, kk, sup, stk, sp, sx, ss)) return -1;
      ill = ill || sp.ill;
      stab_commit_expand<CONT>(geo, st, id, kk, sup, g, stk, sp, sx);
    }
    if (next >= kk) { depth--; continue; }
    frames[d] = (uint32_t)id | ((uint32_t)(next + 1) << 16);
    if (depth >= fcap) return -1;
    frames[depth] = (uint32_t)sup(next);
    depth++;
  }
  return 1;
}
#else
// device flavour: lane 0 walks; when a box splits over six and more supporters the whole wave solves that system
// (stab_lsq_wave) and lane 0 takes the walk up again where it left.  All 64 lanes call; the return value is wave-uniform;
// only lane 0's copy of `st` (n_ent, n_poly) and of `ill` / `ss` is updated -- the caller broadcasts what it needs.
template <bool CONT, typename Geo>
__device__ __forceinline__ int stab_commit_wave(const Geo& geo, StabState& st, int n, double density, const StabWave& w, int lane, bool& ill,
                                                StabStats* ss = nullptr) {
  unsigned char* ws = w.hull;
  const int ws_bytes = w.hull_bytes;
  int rec = 0;
  if (lane == 0) rec = stab_commit_record<CONT>(geo, st, n, density, ws, ws_bytes);
  rec = __builtin_amdgcn_readfirstlane(rec);
  if (rec != 2) return rec;
  uint32_t* frames = reinterpret_cast<uint32_t*>(ws);
  const int fcap = ws_bytes / 4;
  int depth = 1;
  if (lane == 0) frames[0] = (uint32_t)n;
  StabSplit sp;
  StabSplitX sx;
  sp.mode = 0; sp.direct = -1; sp.ill = false;
  bool resume = false;
  int pend_k = 0;
  int rc = 1;
  while (true) {
    int code = 0;  // 0: the walk is through, 1: unstable, 2: capacity, 3: a split for the wave
    if (lane == 0) {
      while (depth > 0) {
        const int d = depth - 1;
        const int id = (int)(frames[d] & 0xFFFFu);
        const int next = (int)(frames[d] >> 16);
        const int kk = stab_nsup(st, id);
        if (kk == 0) { depth--; continue; }
        StabSup sup{st.ent + stab_soff(st, id)};
        if (next == 0) {
          const double stk[4] = {st.stk[(size_t)id * 4 + 0], st.stk[(size_t)id * 4 + 1], st.stk[(size_t)id * 4 + 2],
                                 st.stk[(size_t)id * 4 + 3]};
          double g[9];
          geo(id, g);
          if (!resume) {
            const double (*gp)[2] = reinterpret_cast<const double (*)[2]>(st.poly + (size_t)stab_poff(st, id) * 2);
            if (!stab_pip(stk, gp, stab_npoly(st, id))) { code = 1; break; }
            if (ss) ss->commit_visits++;
            if (!stab_split<CONT>(geo, g, kk, sup, stk, sp, sx, ss, w.lsq_n, w.lsq_mode)) { code = 2; break; }
            if (sp.mode == 4) {
              stab_lsq_inputs<CONT>(geo, g, kk, sup, stk, w.lsq);  // (slot 0)
              pend_k = kk;
              code = 3;
              break;
            }
          }
          resume = false;
          ill = ill || sp.ill;
          stab_commit_expand<CONT>(geo, st, id, kk, sup, g, stk, sp, sx);
        }
        if (next >= kk) { depth--; continue; }
        frames[d] = (uint32_t)id | ((uint32_t)(next + 1) << 16);
        if (depth >= fcap) { code = 2; break; }
        frames[depth] = (uint32_t)sup(next);
        depth++;
      }
    }
    code = __builtin_amdgcn_readfirstlane(code);
    if (code != 3) {
      rc = code == 0 ? 1 : (code == 1 ? 0 : -1);
      break;
    }
    const int ks = __builtin_amdgcn_readfirstlane(pend_k);
    const int G = stab_lsq_group(ks, w.gelsd), cn = stab_lsq_class_n(ks, w.lsq_n, w.gelsd);
    __syncthreads();
    if (ss && lane == 0) ss->lsq_rounds++;
    const bool sill = w.gelsd ? stab_gelsd_slots(w.lsq, lane < G, G, cn, lane, w.gelsd_avx2) : stab_lsq_wave(w.lsq, G, cn, 1, lane);  // one system, in slot 0
    if (lane == 0) {
      if (ks <= 5) {  // (gelsd mode only)
        sp.f[0] = w.lsq[4 + 2 * cn]; sp.f[1] = w.lsq[5 + 2 * cn]; sp.f[2] = w.lsq[6 + 2 * cn];
        sp.f[3] = ks > 3 ? w.lsq[7 + 2 * cn] : 0.0;
        sp.f[4] = ks > 4 ? w.lsq[8 + 2 * cn] : 0.0;
      } else {
        for (int i = 0; i < ks; i++) sx.fx[i] = w.lsq[4 + 2 * cn + i];
      }
      sp.mode = 3;
      sp.ill = sill;
      resume = true;
    }
    __syncthreads();
  }
  return rc;
}
#endif

}  // namespace pct
#endif
