/*
 * pct_oracle_stab.c -- CPU restatement of the reference's stability check (settings 1 / 3):
 * D/space.py:26-267 (Box, calculate_new_com, calculated_impact, calculated_impact_virtual),
 * :341-345 scale_down, :358-379 / :405-426 supporter search, D/convex_hull.py (ConvexHull,
 * Line2D.orientation, point_in_polygen).  TEST INFRASTRUCTURE (see pct_oracle.h).
 *
 * The reference algorithm is sequential, order dependent and written in float64 NumPy; this
 * file follows its operation order.  Known sources of last-bit ambiguity that no restatement
 * can pin down: np.dot / np.linalg.norm go through the BLAS ddot of whatever OpenBLAS kernel
 * the host selects (FMA or not), and np.linalg.lstsq is LAPACK gelsd (>= 3 supporters without
 * a "direct" one; 0.13 % of visits).  Here dot products take the FMA form of
 * OpenBLAS' ddot as built for AVX-512 hosts (dot2 below; its build for AVX2 hosts does NOT fuse: lstsq mode 2) and the
 * least-squares problem is solved by a one-sided Jacobi SVD (minimum-norm) or, in lstsq modes 1 / 2, as LAPACK dgelsd solves it
 * (pct_oracle_gelsd.c).  Parity status: pinned on the setting-1 fixtures of
 * tests/golden/gen_golden.py (observations, rewards, dones identical over the recorded
 * episodes); not a proof of bit-identity on every input.
 *
 * Python object semantics that matter and how they are kept:
 *   - up_edges / up_virtual_edges are dicts keyed by Box objects, iterated in insertion order;
 *     a re-assignment keeps the original position.  up_edges is kept as an ordered array.
 *   - of up_virtual_edges only entries whose key is `involved` are ever read, and at most one
 *     key of a given box can be involved at a time (the active recursion path holds one box per
 *     height level), always written just before it is read: one slot per box suffices.
 *   - Stack objects shared BY REFERENCE: with one supporter, or for the "direct" supporter, the reference
 *     stores the box's own thisStack object in the supporter's up_edges (`up_edges[self] = self.thisStack`,
 *     D/space.py:80,96), and calculate_new_com later mutates that object in place (:67-71).  A supporter
 *     reading such an entry therefore always sees the box's CURRENT committed stack, not the value at the
 *     time of the assignment.  The two differ inside a commit walk: a box with >= 2 supporters first
 *     recomputes ALL of them and only then visits them one by one, so while the first supporter's subtree
 *     is walked, a box further down that also carries the second supporter through an aliased entry
 *     already sees that supporter's new stack.  Kept as an `alias` flag per up_edges entry (pinned by the
 *     *_flat_lstsq fixtures, whose reference runs fail a commit that a by-value copy lets pass).  The
 *     virtual flavour's aliases (up_virtual_edges[self] = self.thisVirtualStack) are read only while `self`
 *     is involved, right after they are written: value copies are equivalent there.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "pct_oracle_internal.h"

/* np.dot of two 2-vectors as NumPy's BLAS computes it: OpenBLAS' ddot kernel for AVX-512 hosts ("SkylakeX": the build this
 * oracle was pinned against dispatches to it) accumulates
 * acc = x0*y0; acc = fma(x1, y1, acc) -- ONE rounding for the second product and the sum, not two.  Measured
 * against np.dot on 200 000 random pairs (0 disagreements; the two-rounding form disagrees on 40 %).  It
 * decides the exactly-degenerate tests downstream (a stack centre on the line through a polygon edge). */
/* (...on hosts where OpenBLAS runs its SkylakeX kernel set -- AVX-512, the build container.  Its "Haswell" kernel set -- AVX2 hosts, AMD
 * Zen included -- compiles that ddot's tail loop WITHOUT FMA: x0*y0 + x1*y1, three roundings; measured under OPENBLAS_CORETYPE=HASWELL,
 * 20 000 of 20 000.  stab_set_lstsq_mode(2) selects that flavour together with the Haswell BLAS arithmetic of pct_oracle_gelsd.c.) */
static int g_dot_plain = 0;
static double dot2(double x0, double x1, double y0, double y1) { return g_dot_plain ? x0 * y0 + x1 * y1 : fma(x1, y1, x0 * y0); }

typedef struct { double c[3]; double m; } sstack;
typedef struct { int box; double area[4]; double c2[2]; } sdown; /* DownEdge */
typedef struct { int key; int alias; sstack st; } sedge; /* alias: the entry IS boxes[key].thisStack */

typedef struct sbox {
  double x, y, z, lx, ly, lz;
  double centre[3], mass;
  sdown* bottom; int nbottom;
  double (*poly)[2]; int npoly;
  sedge* up; int nup, capup;
  sstack vshare; /* the up_virtual_edges entry of the currently involved parent */
  sstack thisStack, thisVirtual;
  int involved;
} sbox;

struct stab {
  sbox* boxes; /* placed boxes, placement order */
  int n, cap;
  double eps; /* 0 for the discrete env, 1e-6 margins for the continuous env */
  int ill;    /* sticky: a least-squares split took its rank decision within a factor ILL_BAND of the cut, or a test on a
                 stack that may carry a least-squares share was decided by less than a relative 1e-9 (the notice the
                 product raises as PCT_FLAG_ILL_CONDITIONED: there the reference's verdict depends on its LAPACK build) */
  int lsq_seen; /* this episode has committed a least-squares split (reset with the episode) */
};
#define ILL_BAND 1e3
#define ILL_NEAR 1e-9
static _Thread_local int g_ill; /* set by lstsq_min_norm / the degeneracy tests below, collected by stab_check */
/* "tainted": the stacks being tested may carry a least-squares share -- the episode has committed such a split, or the
 * walk under way has made one.  A point-in-polygon / direct-supporter test that is then decided by less than a relative
 * 1e-9 is one whose outcome in the reference depends on the last bits LAPACK returned (profiles/r03_lstsq_limit.txt:
 * 15 of the 17 diverging env-runs part ways like this, with solves that agree to 1e-9 and better). */
static _Thread_local int g_tainted;
/* The tie notice is an ANALYSIS mode (tests/golden/check_ill_notice.py), off by default and not part of the product's
 * flag: it announces all 17 diverging runs of the adversarial seeds but also every run that never diverges, and 14 % of
 * the episodes of the reference's own item domain (profiles/r03_lstsq_limit.txt). */
static int g_ill_near = 0;
void stab_set_ill_near(int on) { g_ill_near = on; }
static int near_rel(double a, double b) { return fabs(a - b) <= ILL_NEAR * fmax(fmax(fabs(a), fabs(b)), 1e-300); }
int stab_ill_conditioned(const struct stab* s) { return s->ill & 1; }
/* ... and whether a solve of a COMMIT raised it (calculated_impact, D/space.py:73-164: the state-changing walk, whose solves every
 * evaluation order makes) rather than only a virtual check's (a candidate's walk stops at its first unstable supporter: which of a
 * doomed candidate's solves are made at all depends on the order its supporters are examined in) */
int stab_ill_commit(const struct stab* s) { return (s->ill >> 1) & 1; }
static int stab_check_(struct stab* s, double x, double y, double z, double lx, double ly, double max_h, double density,
                       int virtual_);

struct stab* stab_create(int cap, double eps) {
  struct stab* s = (struct stab*)calloc(1, sizeof *s);
  s->cap = cap + 2;
  s->boxes = (sbox*)calloc((size_t)s->cap, sizeof(sbox));
  s->eps = eps;
  return s;
}
static void sbox_clear(sbox* b) {
  free(b->bottom); free(b->poly); free(b->up);
  memset(b, 0, sizeof *b);
}
void stab_reset(struct stab* s) {
  for (int i = 0; i < s->n; i++) sbox_clear(&s->boxes[i]);
  s->n = 0;
  s->lsq_seen = 0;
}
void stab_clear_ill(struct stab* s) { s->ill = 0; }
void stab_free(struct stab* s) {
  if (!s) return;
  stab_reset(s);
  free(s->boxes);
  free(s);
}
int stab_count(const struct stab* s) { return s->n; }

/* ---- D/convex_hull.py ------------------------------------------------------------------- */
static double slope_of(const double* p1, const double* p2) { /* Line2D.__init__ :6-14 */
  if (p2[0] != p1[0]) return (p2[1] - p1[1]) / (p2[0] - p1[0]);
  return (p2[1] - p1[1]) * INFINITY;
}
static int orientation(double slope1, double slope2) { /* :16-32 */
  if (fabs(slope1) == INFINITY && fabs(slope2) == INFINITY) return 0;
  double diff = slope2 - slope1;
  if (diff > 0) return -1;
  else if (diff == 0) return 0;
  else return 1;
}
/* one monotone chain exactly as :50-63 / :66-83 write it (including the stale line1/line2 and
 * the break when the chain collapses onto its first point) */
static int chain(double (*sorted)[2], int n, int reverse, double (*hull)[2]) {
  int len = 0;
  double s1 = 0, s2 = 0;
  for (int q = 0; q < n; q++) {
    const double* point = sorted[reverse ? n - 1 - q : q];
    if (len >= 2) {
      s1 = slope_of(hull[len - 2], hull[len - 1]);
      s2 = slope_of(hull[len - 1], point);
    }
    while (len >= 2 && orientation(s1, s2) != -1) {
      len--; /* pop */
      if (hull[0][0] == hull[len - 1][0] && hull[0][1] == hull[len - 1][1]) break;
      s1 = slope_of(hull[len - 2], hull[len - 1]);
      s2 = slope_of(hull[len - 1], point);
    }
    hull[len][0] = point[0];
    hull[len][1] = point[1];
    len++;
  }
  return len;
}
/* ConvexHull :39-95 followed by Space.scale_down (D/space.py:341-345); returns vertex count */
static int hull_scaled(double (*pts)[2], int n, double (*out)[2]) {
  for (int i = 0; i < n; i++) pts[i][0] += pts[i][1] * 1e-6;
  /* sorted(point_list, key=lambda x: x[0]): stable */
  for (int i = 1; i < n; i++) {
    double v0 = pts[i][0], v1 = pts[i][1];
    int j = i;
    while (j > 0 && pts[j - 1][0] > v0) { pts[j][0] = pts[j - 1][0]; pts[j][1] = pts[j - 1][1]; j--; }
    pts[j][0] = v0; pts[j][1] = v1;
  }
  double (*lo)[2] = malloc(sizeof(double[2]) * (size_t)(n + 1));
  double (*up)[2] = malloc(sizeof(double[2]) * (size_t)(n + 1));
  int nl = chain(pts, n, 0, lo);
  int nu = chain(pts, n, 1, up);
  nu--; /* removed = upperHull.pop() */
  nl--;
  int m = 0;
  for (int i = 0; i < nl; i++) { out[m][0] = lo[i][0]; out[m][1] = lo[i][1]; m++; }
  for (int i = 0; i < nu; i++) { out[m][0] = up[i][0]; out[m][1] = up[i][1]; m++; }
  free(lo); free(up);
  /* scale_down: centre = np.mean(axis=0); hull -= (hull - centre) * 0.1 */
  double cx = 0, cy = 0;
  for (int i = 0; i < m; i++) { cx += out[i][0]; cy += out[i][1]; }
  cx /= (double)m; cy /= (double)m;
  for (int i = 0; i < m; i++) {
    out[i][0] -= (out[i][0] - cx) * 0.1;
    out[i][1] -= (out[i][1] - cy) * 0.1;
  }
  return m;
}
/* point_in_polygen :97-112 */
static int point_in_polygon(const double pt[2], double (*co)[2], int n) {
  double lat = pt[0], lon = pt[1];
  int j = n - 1, odd = 0;
  for (int i = 0; i < n; i++) {
    double a0 = co[i][0] - pt[0], a1 = co[i][1] - pt[1];
    double b0 = pt[0] - co[j][0], b1 = pt[1] - co[j][1];
    double cp = a0 * b1;
    if (g_tainted && near_rel(a0 * b1, a1 * b0)) g_ill = 1;
    cp -= a1 * b0;
    if (cp == 0) return 0;
    if (g_tainted && (near_rel(co[i][1], lon) || near_rel(co[j][1], lon))) g_ill = 1;
    if ((co[i][1] < lon && co[j][1] >= lon) || (co[j][1] < lon && co[i][1] >= lon)) {
      const double xc = co[i][0] + (lon - co[i][1]) / (co[j][1] - co[i][1]) * (co[j][0] - co[i][0]);
      if (g_tainted && near_rel(xc, lat)) g_ill = 1;
      if (xc < lat) odd = !odd;
    }
    j = i;
  }
  return odd;
}

/* ---- minimum-norm least squares (stands in for np.linalg.lstsq = LAPACK dgelsd, see the header) ---------
 * One-sided Jacobi (Hestenes) SVD of A itself: the columns of U = A V are rotated pairwise until they are
 * orthogonal; then sigma_j = |U_j| and x = sum_j V_j (U_j . b) / sigma_j^2 over sigma_j > rcond * sigma_max with
 * NumPy's default rcond = eps * max(M, N).  Working on A (not on A^T A) keeps the small singular values accurate
 * to eps * sigma_max, as dgelsd does -- the normal-equations form loses everything below sqrt(eps) * sigma_max
 * and then disagrees with LAPACK by up to 5e-3 on the nearly rank-deficient systems this check produces. */
static int g_lstsq_mode = 1; /* 1 (default since round 5, as the kernels): LAPACK dgelsd as the reference's NumPy executes it
                                (pct_oracle_gelsd.c); 0: the Jacobi stand-in below (PCT_LSTSQ_JACOBI) */
void stab_set_lstsq_mode(int mode) { /* 0: Jacobi; 1: dgelsd, AVX-512 kernel set; 2: dgelsd, AVX2 (Haswell / Zen) kernel set */
  g_lstsq_mode = mode != 0;
  gelsd_set_kernel_set(mode == 2);
  g_dot_plain = mode == 2;
}
int stab_get_lstsq_mode(void) { return g_lstsq_mode ? 1 + gelsd_get_kernel_set() : 0; }
static long g_beyond_smlsiz = 0; /* dgelsd-mode splits over more than 25 supporters handed to the stand-in (unpinned territory) */
long stab_beyond_smlsiz(void) { return g_beyond_smlsiz; }
static void lstsq_min_norm(const double* A, const double* b, int M, int N, double* x) {
  if (g_lstsq_mode == 1) {
    double sv[64];
    int near_cut = 0;
    if (N <= 25) { /* dlalsd: n <= SMLSIZ = 25 -> dlasdq, the path pct_oracle_gelsd.c restates (pinned on 20 - 24 unknowns by
                      tests/golden/plate_discrete_s1.npz); beyond it LAPACK runs dlasda / dlalsa (divide and conquer) */
      if (gelsd_lstsq(A, b, M, N, x, NULL, sv, &near_cut) != 0) { /* dbdsqr did not converge: NumPy raises LinAlgError there */
        for (int i = 0; i < N; i++) x[i] = 0;
        near_cut = 1;
      }
      if (near_cut) g_ill = 1;
      return;
    }
    g_ill = 1; /* more than 25 supporters: NOT the reference's arithmetic any more -- the stand-in takes over, counted below
                  (the product raises PCT_FLAG_STABILITY_OVERFLOW | STAB_WHY_SPLIT there and ends the episode) */
    g_beyond_smlsiz++;
  }
  double* U = (double*)malloc(sizeof(double) * (size_t)M * N);
  double* V = (double*)malloc(sizeof(double) * (size_t)N * N);
  double* s2 = (double*)malloc(sizeof(double) * (size_t)N);
  for (int i = 0; i < M * N; i++) U[i] = A[i];
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) V[i * N + j] = (i == j);
  for (int sweep = 0; sweep < 60; sweep++) {
    int rotated = 0;
    for (int p = 0; p < N; p++)
      for (int q = p + 1; q < N; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < M; r++) {
          alpha += U[r * N + p] * U[r * N + p];
          beta += U[r * N + q] * U[r * N + q];
          gamma += U[r * N + p] * U[r * N + q];
        }
        if (gamma == 0 || fabs(gamma) <= 2.220446049250313e-16 * sqrt(alpha * beta)) continue;
        rotated = 1;
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
  double smax2 = 0;
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
    if (s2[j] > 0 && sj > cut / ILL_BAND && sj < cut * ILL_BAND) g_ill = 1;
    if (s2[j] <= 0 || sj <= cut) continue;
    double proj = 0;
    for (int r = 0; r < M; r++) proj += U[r * N + j] * b[r];
    proj /= s2[j];
    for (int i = 0; i < N; i++) x[i] += V[i * N + j] * proj;
  }
  free(U); free(V); free(s2);
}

/* ---- Box ------------------------------------------------------------------------------- */
/* D/space.py:51-71 calculate_new_com */
static void calc_com(struct stab* s, sbox* b, int virtual_, const sbox* cand) {
  (void)cand;
  double c0 = b->centre[0] * b->mass, c1 = b->centre[1] * b->mass, c2 = b->centre[2] * b->mass, m = b->mass;
  for (int i = 0; i < b->nup; i++) {
    int key = b->up[i].key;
    if (!s->boxes[key].involved) {
      const sstack* e = b->up[i].alias ? &s->boxes[key].thisStack : &b->up[i].st;
      c0 += e->c[0] * e->m; c1 += e->c[1] * e->m; c2 += e->c[2] * e->m;
      m += e->m;
    }
  }
  if (virtual_) { /* the single involved up_virtual_edges entry, written just before */
    const sstack* e = &b->vshare;
    c0 += e->c[0] * e->m; c1 += e->c[1] * e->m; c2 += e->c[2] * e->m;
    m += e->m;
  }
  c0 /= m; c1 /= m; c2 /= m;
  sstack* t = virtual_ ? &b->thisVirtual : &b->thisStack;
  t->c[0] = c0; t->c[1] = c1; t->c[2] = c2; t->m = m;
}
static void set_up_edge(sbox* sup, int key, const sstack* st, int alias) {
  for (int i = 0; i < sup->nup; i++)
    if (sup->up[i].key == key) { sup->up[i].st = *st; sup->up[i].alias = alias; return; } /* dict re-assignment keeps position */
  if (sup->nup == sup->capup) {
    sup->capup = sup->capup ? sup->capup * 2 : 4;
    sup->up = (sedge*)realloc(sup->up, sizeof(sedge) * (size_t)sup->capup);
  }
  sup->up[sup->nup].key = key;
  sup->up[sup->nup].alias = alias;
  sup->up[sup->nup].st = *st;
  sup->nup++;
}

/* D/space.py:73-164 calculated_impact (virtual_ == 0) and :166-267 calculated_impact_virtual.
 * `b` is either a placed box or the candidate; `key` is its id as a dict key (the candidate of
 * a commit gets the id it will have once appended; a virtual candidate never needs one). */
static int impact_(struct stab* s, sbox* b, int key, int virtual_);
static int impact(struct stab* s, sbox* b, int key, int virtual_) {
  const int taint_in = g_tainted;
  const int rc = impact_(s, b, key, virtual_);
  if (virtual_) g_tainted = taint_in; /* the taint follows the walk tree: siblings of a least-squares node are not its heirs */
  return rc;
}
static int impact_(struct stab* s, sbox* b, int key, int virtual_) {
  const double eps = s->eps;
  if (virtual_) b->involved = 1;
  if (b->nbottom == 0) { if (virtual_) b->involved = 0; return 1; }
  sstack* st = virtual_ ? &b->thisVirtual : &b->thisStack;
  if (!point_in_polygon(st->c, b->poly, b->npoly)) { if (virtual_) b->involved = 0; return 0; }
  int k = b->nbottom, ok = 1;
#define GIVE(i, stk, alias_)                                            \
  do {                                                                 \
    sbox* sup_ = &s->boxes[b->bottom[i].box];                          \
    if (virtual_) sup_->vshare = (stk); else set_up_edge(sup_, key, &(stk), (alias_)); \
    calc_com(s, sup_, virtual_, b);                                    \
  } while (0)
  if (k == 1) {
    sstack sh = *st;
    GIVE(0, sh, 1); /* up_edges[self] = self.thisStack: the object itself */
    if (!impact(s, &s->boxes[b->bottom[0].box], b->bottom[0].box, virtual_)) ok = 0;
  } else {
    int direct = -1;
    for (int i = 0; i < k; i++) {
      const double* a = b->bottom[i].area;
      int inside = eps > 0 /* C/space.py:85-86,182-183 vs D/space.py:89-90,186-187 */
                       ? (st->c[0] - a[0] > 1e-6 && a[2] - st->c[0] > 1e-6 && st->c[1] - a[1] > 1e-6 && a[3] - st->c[1] > 1e-6)
                       : (st->c[0] > a[0] && st->c[0] < a[2] && st->c[1] > a[1] && st->c[1] < a[3]);
      if (g_tainted && (near_rel(st->c[0], a[0]) || near_rel(st->c[0], a[2]) || near_rel(st->c[1], a[1]) || near_rel(st->c[1], a[3]))) g_ill = 1;
      if (inside) { direct = i; break; }
    }
    if (direct >= 0) {
      for (int i = 0; i < k; i++) {
        sstack sh;
        if (i == direct) sh = *st;
        else { /* Stack(centre, 0): commit uses thisStack.centre, virtual uses self.centre */
          const double* cc = virtual_ ? b->centre : st->c;
          sh.c[0] = cc[0]; sh.c[1] = cc[1]; sh.c[2] = cc[2]; sh.m = 0;
        }
        GIVE(i, sh, i == direct);
      }
      for (int i = 0; i < k && ok; i++)
        if (!impact(s, &s->boxes[b->bottom[i].box], b->bottom[i].box, virtual_)) ok = 0;
    } else if (k == 2) {
      const double* e0 = b->bottom[0].c2;
      const double* e1 = b->bottom[1].c2;
      double t0 = e0[0] - e1[0], t1 = e0[1] - e1[1];
      double len = sqrt(dot2(t0, t1, t0, t1)); /* np.linalg.norm = sqrt(x.dot(x)) */
      double l2 = pow(len, 2.0);               /* tri_base_len ** 2 */
      t0 /= l2; t1 /= l2;
      double r0 = fabs(dot2(st->c[0] - e1[0], st->c[1] - e1[1], t0, t1));
      double r1 = fabs(dot2(st->c[0] - e0[0], st->c[1] - e0[1], t0, t1));
      sstack s0 = {{e0[0], e0[1], st->c[2]}, st->m * r0};
      sstack s1 = {{e1[0], e1[1], st->c[2]}, st->m * r1};
      GIVE(0, s0, 0);
      GIVE(1, s1, 0);
      if (!impact(s, &s->boxes[b->bottom[0].box], b->bottom[0].box, virtual_)) ok = 0;
      else if (!impact(s, &s->boxes[b->bottom[1].box], b->bottom[1].box, virtual_)) ok = 0;
    } else {
      int M = k * (k - 1) / 2 + 1;
      double* A = (double*)calloc((size_t)M * k, sizeof(double));
      double* rhs = (double*)calloc((size_t)M, sizeof(double));
      double* xr = (double*)calloc((size_t)k, sizeof(double));
      int row = 0;
      for (int i = 0; i < k - 1; i++)
        for (int j = i + 1; j < k; j++) {
          const double* ei = b->bottom[i].c2;
          const double* ej = b->bottom[j].c2;
          double t0 = ei[0] - ej[0], t1 = ei[1] - ej[1];
          double mol = dot2(st->c[0] - ei[0], st->c[1] - ei[1], t0, t1);
          if (mol != 0) {
            double rr = fabs(dot2(st->c[0] - ej[0], st->c[1] - ej[1], t0, t1)) / mol;
            A[row * k + i] = 1;
            A[row * k + j] = -rr;
          }
          row++;
        }
      for (int j = 0; j < k; j++) A[(M - 1) * k + j] = 1;
      rhs[M - 1] = 1;
      lstsq_min_norm(A, rhs, M, k, xr);
      if (g_ill_near) g_tainted = 1; /* from here on this walk hands least-squares shares down */
      if (!virtual_) s->lsq_seen = 1;
      for (int i = 0; i < k; i++) {
        sstack sh = {{b->bottom[i].c2[0], b->bottom[i].c2[1], st->c[2]}, st->m * xr[i]};
        GIVE(i, sh, 0);
      }
      for (int i = 0; i < k && ok; i++)
        if (!impact(s, &s->boxes[b->bottom[i].box], b->bottom[i].box, virtual_)) ok = 0;
      free(A); free(rhs); free(xr);
    }
  }
#undef GIVE
  if (virtual_) b->involved = 0;
  return ok;
}

/* D/space.py:358-379 / :405-426: supporters, contact rectangles, scaled hull; then the
 * stability branch of check_box (:448-454).  Geometry in doubles (ints for the discrete env).
 * virtual_ != 0: no lasting change.  virtual_ == 0: on success the box is appended. */
int stab_check(struct stab* s, double x, double y, double z, double lx, double ly, double max_h, double density,
               int virtual_) {
  g_ill = 0;
  g_tainted = g_ill_near && s->lsq_seen;
  const int rc_ = stab_check_(s, x, y, z, lx, ly, max_h, density, virtual_);
  if (g_ill) s->ill |= virtual_ ? 1 : 3;
  return rc_;
}
static int stab_check_(struct stab* s, double x, double y, double z, double lx, double ly, double max_h, double density,
                       int virtual_) {
  const double eps = s->eps;
  sbox nb;
  memset(&nb, 0, sizeof nb);
  nb.x = x; nb.y = y; nb.z = z; nb.lx = lx; nb.ly = ly; nb.lz = max_h;
  nb.centre[0] = lx + x / 2; nb.centre[1] = ly + y / 2; nb.centre[2] = max_h + z / 2;
  nb.mass = x * y * z * density;
  if (virtual_) nb.mass *= 1.0;
  nb.thisStack.c[0] = nb.thisVirtual.c[0] = nb.centre[0];
  nb.thisStack.c[1] = nb.thisVirtual.c[1] = nb.centre[1];
  nb.thisStack.c[2] = nb.thisVirtual.c[2] = nb.centre[2];
  nb.thisStack.m = nb.thisVirtual.m = nb.mass;
  nb.bottom = (sdown*)calloc((size_t)s->n + 1, sizeof(sdown));
  double (*pts)[2] = malloc(sizeof(double[2]) * (size_t)(4 * s->n + 4));
  int np_ = 0;
  for (int i = 0; i < s->n; i++) {
    const sbox* t = &s->boxes[i];
    double x1, y1, x2, y2;
    if (eps > 0) {
      /* C/space.py:305-314 interSect2D picks the overlapping boxes on rounded values, then
       * :353 keeps those whose top is max_h within 1e-6; the contact rectangle is the rounded
       * intersection (:355-357) */
      double i0 = rint(fmin(-lx, -t->lx) * 1e6) / 1e6, i1 = rint(fmin(-ly, -t->ly) * 1e6) / 1e6;
      double i2 = rint(fmin(lx + x, t->lx + t->x) * 1e6) / 1e6, i3 = rint(fmin(ly + y, t->ly + t->y) * 1e6) / 1e6;
      if (!((i0 + i2 > 0) && (i1 + i3 > 0))) continue;
      if (!(fabs(t->lz + t->z - max_h) < 1e-6)) continue;
      x1 = -i0; y1 = -i1; x2 = i2; y2 = i3;
    } else {
      if (!(t->lz + t->z == max_h)) continue;
      x1 = fmax(lx, t->lx); y1 = fmax(ly, t->ly);
      x2 = fmin(lx + x, t->lx + t->x); y2 = fmin(ly + y, t->ly + t->y);
      if (x1 >= x2 || y1 >= y2) continue;
    }
    sdown* d = &nb.bottom[nb.nbottom++];
    d->box = i;
    d->area[0] = x1; d->area[1] = y1; d->area[2] = x2; d->area[3] = y2;
    d->c2[0] = (x1 + x2) / 2; d->c2[1] = (y1 + y2) / 2;
    pts[np_][0] = x1; pts[np_][1] = y1; np_++;
    pts[np_][0] = x1; pts[np_][1] = y2; np_++;
    pts[np_][0] = x2; pts[np_][1] = y1; np_++;
    pts[np_][0] = x2; pts[np_][1] = y2; np_++;
  }
  if (np_ > 0) {
    nb.poly = malloc(sizeof(double[2]) * (size_t)(2 * np_ + 2));
    nb.npoly = hull_scaled(pts, np_, nb.poly);
  }
  free(pts);
  int ok;
  if (eps > 0 ? (fabs(max_h) < eps) : (max_h == 0)) ok = 1; /* :448-449 */
  else if (virtual_) ok = impact(s, &nb, -1, 1);
  else {
    /* the candidate must be addressable as boxes[n] while its supporters record it as a key */
    s->boxes[s->n] = nb;
    ok = impact(s, &s->boxes[s->n], s->n, 0);
    nb = s->boxes[s->n];
  }
  if (!virtual_ && ok) {
    s->boxes[s->n] = nb;
    s->n++;
    return 1;
  }
  if (!virtual_) memset(&s->boxes[s->n], 0, sizeof(sbox));
  free(nb.bottom); free(nb.poly); free(nb.up);
  return ok;
}
