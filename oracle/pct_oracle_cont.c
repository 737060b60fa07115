/*
 * pct_oracle_cont.c -- CPU restatement of pct_envs/PctContinuous0 (TEST INFRASTRUCTURE; see
 * pct_oracle.h).  "C/" = pct_envs/PctContinuous0/.
 *
 * The reference computes in float64 with 1e-6 epsilons, np.around(.,6) and Python round();
 * this file performs THE SAME float64 operations in the same order (IEEE add/sub/mul/div,
 * rint) so that observations, set-iteration order (float hashes!) and every decision are
 * bit-identical to the reference for float64 actions.  Container sizes are integers in the
 * reference (np.array([10,10,10]) -> int64), item sizes are 3-decimal floats.
 *
 *   np.around(x, 6)  == rint(x * 1e6) / 1e6           (numpy: multiply, rint, true_divide)
 *   round(x, 6)      == the same for values within ~1e-9 of the 1e-6 lattice (CPython rounds
 *                       the exact decimal expansion; k/1e6 is correctly rounded either way)
 *
 * Parity status: PINNED against the unmodified reference by tests/golden/gen_golden.py
 * (continuous cases) for setting 2, scripted item streams, float64 actions.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "pct_oracle_internal.h"

#define EMS_ROWS 1000 /* C/space.py:276 self.EMS = np.zeros((1000, 6)) */

struct cenv {
  double* upLetter; /* [I*5] */
  double* box_vec;  /* [I*9] */
  double* vol;      /* per placed box x*y*z (get_ratio) */
  int n_boxes;      /* len(self.boxes) */
  int box_idx;
  double* ems; /* [EMS_ROWS*6] */
  int noems;
  double next_box[3];
  double next_den;
  double queue_item[3];
  int queue_len;
  uint64_t cursor;
  uint32_t t;
  int traj;
  struct stab* stab; /* settings 1 / 3 */
  uint64_t oc;       /* observations produced so far (shuffle key) */
  uint32_t mt[624];  /* strict NumPy-stream mode: this env's MT19937 state (pct_oracle_internal.h npmt_*) */
  int mt_pos;
};

static double around6(double x) { return rint(x * 1e6) / 1e6; }

/* ---- CPython float / tuple hashing and set order ---------------------------------------- */
#define PYHASH_BITS 61
#define PYHASH_MOD (((uint64_t)1 << PYHASH_BITS) - 1)
/* Python/pyhash.c _Py_HashDouble (finite values) */
static int64_t py_hash_double(double v) {
  if (v == 0.0) return 0;
  int e;
  double m = frexp(v, &e);
  int sign = 1;
  if (m < 0) { sign = -1; m = -m; }
  uint64_t x = 0;
  while (m != 0.0) {
    x = ((x << 28) & PYHASH_MOD) | x >> (PYHASH_BITS - 28);
    m *= 268435456.0;
    e -= 28;
    uint64_t y = (uint64_t)m;
    m -= (double)y;
    x += y;
    if (x >= PYHASH_MOD) x -= PYHASH_MOD;
  }
  e = e >= 0 ? e % PYHASH_BITS : PYHASH_BITS - 1 - ((-1 - e) % PYHASH_BITS);
  x = ((x << e) & PYHASH_MOD) | x >> (PYHASH_BITS - e);
  int64_t r = (int64_t)x * sign;
  if (r == -1) r = -2;
  return r;
}
#define XXPRIME_1 11400714785074694791ULL
#define XXPRIME_2 14029467366897019727ULL
#define XXPRIME_5 2870177450012600261ULL
static uint64_t py_tuplehash6d(const double* v) {
  uint64_t acc = XXPRIME_5;
  for (int i = 0; i < 6; i++) {
    uint64_t lane = (uint64_t)py_hash_double(v[i]);
    acc += lane * XXPRIME_2;
    acc = (acc << 31) | (acc >> 33);
    acc *= XXPRIME_1;
  }
  acc += 6ULL ^ (XXPRIME_5 ^ 3527539ULL);
  if (acc == (uint64_t)-1) return 1546275796ULL;
  return acc;
}
static int tuple_eq(const double* a, const double* b) {
  for (int i = 0; i < 6; i++)
    if (!(a[i] == b[i])) return 0;
  return 1;
}
/* Objects/setobject.c emulation on double 6-tuples (see pct_oracle.c for the int version) */
typedef struct {
  int32_t* slot;
  uint64_t* hash;
  size_t mask, fill;
} dset;
static void dset_insert_clean(int32_t* slot, uint64_t* hs, size_t mask, int32_t key, uint64_t hash) {
  size_t perturb = hash, i = (size_t)hash & mask;
  while (1) {
    size_t e = i;
    if (slot[e] < 0) { slot[e] = key; hs[e] = hash; return; }
    if (i + 9 <= mask)
      for (int j = 0; j < 9; j++) {
        e++;
        if (slot[e] < 0) { slot[e] = key; hs[e] = hash; return; }
      }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}
static void dset_add(dset* s, const double* keys, int32_t k) {
  const double* key = keys + 6 * (size_t)k;
  uint64_t hash = py_tuplehash6d(key);
  size_t mask = s->mask, i = (size_t)hash & mask, perturb = hash;
  while (1) {
    size_t e = i;
    int probes = (i + 9 <= mask) ? 9 : 0;
    do {
      if (s->slot[e] < 0) {
        s->slot[e] = k;
        s->hash[e] = hash;
        s->fill++;
        if (s->fill * 5 >= mask * 3) {
          size_t minused = s->fill > 50000 ? s->fill * 2 : s->fill * 4, newsize = 8;
          while (newsize <= minused) newsize <<= 1;
          int32_t* ns = (int32_t*)malloc(newsize * sizeof(int32_t));
          uint64_t* nh = (uint64_t*)malloc(newsize * sizeof(uint64_t));
          for (size_t q = 0; q < newsize; q++) ns[q] = -1;
          for (size_t q = 0; q <= s->mask; q++)
            if (s->slot[q] >= 0) dset_insert_clean(ns, nh, newsize - 1, s->slot[q], s->hash[q]);
          free(s->slot); free(s->hash);
          s->slot = ns; s->hash = nh; s->mask = newsize - 1;
        }
        return;
      }
      if (s->hash[e] == hash && tuple_eq(keys + 6 * (size_t)s->slot[e], key)) return;
      e++;
    } while (probes--);
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

/* ---- strict NumPy-stream mode ------------------------------------------------------------- */
/* Python's round(x, 3) of a positive float as a lattice index k (the result is the double nearest to k / 1000):
 * correctly rounded on the EXACT binary value of x, ties to even (float.__round__ -> _Py_dg_dtoa mode 3).
 * x is compared with the midpoints (2k +- 1) / 2000 in integer arithmetic: x = m * 2^e, so x * 2000 = (m * 2000) * 2^e. */
static int cmp_x_mid(double x, int64_t twok1) { /* sign of x * 2000 - twok1 */
  int ex;
  double fr = frexp(x, &ex);                      /* x = fr * 2^ex, fr in [0.5, 1) */
  uint64_t m = (uint64_t)ldexp(fr, 53);           /* 53-bit integer mantissa */
  int e2 = ex - 53;                               /* x = m * 2^e2 */
  unsigned __int128 A = (unsigned __int128)m * 2000u, B = (unsigned __int128)(uint64_t)twok1;
  if (e2 >= 0) A <<= e2; else B <<= -e2;
  return A > B ? 1 : (A < B ? -1 : 0);
}
static int32_t round3_lattice(double x) {
  int64_t k = (int64_t)(x * 1000.0 + 0.5);
  for (int it = 0; it < 3; it++) {
    int up = cmp_x_mid(x, 2 * k + 1), dn = cmp_x_mid(x, 2 * k - 1);
    if (up > 0 || (up == 0 && (k & 1))) { k++; continue; }          /* above the upper midpoint (tie -> even) */
    if (dn < 0 || (dn == 0 && (k & 1))) { k--; continue; }          /* below the lower midpoint (tie -> even) */
    break;
  }
  return (int32_t)k;
}
/* C/bin3D.py:103-113 gen_next_box in sampling mode: round(np.random.uniform(a, b), 3) three times, or twice plus
 * np.random.choice([0.1, ..., 0.5]) (legacy choice -> randint(0, 5)) under settings 1 / 3 */
static void draw_item_numpy(const struct pcto_env* h, struct cenv* s, double out[3]) {
  const double a = (double)h->sample_left / 1000.0, b = (double)h->sample_right / 1000.0;
  int32_t k[3];
  const int nu = h->cfg.setting == 2 ? 3 : 2;
  for (int d = 0; d < nu; d++) k[d] = round3_lattice(a + (b - a) * npmt_double(s->mt, &s->mt_pos));
  if (nu == 2) k[2] = 100 * (1 + (int32_t)npmt_interval(s->mt, &s->mt_pos, 4u));
  for (int d = 0; d < 3; d++) out[d] = (double)k[d] / 1000.0;
}

/* ---- item source ------------------------------------------------------------------------ */
/* lattice integer k (1e-3 units) -> the float the reference holds: round(U(a,b), 3) /
 * round(x, 3) (C/bin3D.py:85,106-108) == k / 1000.0 correctly rounded */
static void draw_item(const struct pcto_env* h, int e, struct cenv* s, double out[3]) {
  uint64_t c = s->cursor++;
  int32_t k[3];
  if (h->source == PCT_ITEMS_DATASET) { /* binCreator.py:64-72; sizes are round(.,3) (C/bin3D.py:85) */
    int t = s->traj < h->ds_ntraj ? s->traj : h->ds_ntraj - 1;
    int len = h->ds_len[t];
    if (c < (uint64_t)len) {
      const int32_t* p = h->stream + ((size_t)t * h->ds_maxlen + (size_t)c) * 3;
      k[0] = p[0]; k[1] = p[1]; k[2] = p[2];
    } else {
      k[0] = k[1] = k[2] = (c == (uint64_t)len) ? 100000 : 10000;
    }
  } else if (h->source == PCT_ITEMS_STREAM) {
    const int32_t* p = h->stream + ((size_t)e * (size_t)h->T + (size_t)(c % (uint64_t)h->T)) * 3;
    k[0] = p[0]; k[1] = p[1]; k[2] = p[2];
  } else {
    uint64_t g = (uint64_t)(h->cfg.env_id_base + e);
    if (h->sample_right <= 0) {
      /* not sample_from_distribution: RandomBoxCreator(item_set) (C/bin3D.py:36-39,113; binCreator.py:37-39) */
      const int32_t* it = h->item_set + 3 * (size_t)pct_pick(h->seed, g, c, (uint32_t)h->n_items);
      k[0] = it[0]; k[1] = it[1]; k[2] = it[2];
    } else {
      uint64_t span = (uint64_t)(h->sample_right - h->sample_left + 1);
      for (int d = 0; d < 3; d++) k[d] = h->sample_left + (int32_t)(pct_pick(h->seed, g, c * 3 + (uint64_t)d, (uint32_t)span));
      /* C/bin3D.py:110-112: settings 1 and 3 take z from np.random.choice([0.1,0.2,0.3,0.4,0.5]) */
      if (h->cfg.setting != 2) k[2] = 100 * (1 + (int32_t)pct_pick(h->seed, g, c * 3 + 2, 5u));
    }
  }
  for (int d = 0; d < 3; d++) out[d] = (double)k[d] / 1000.0;
}

/* ---- C/space.py ------------------------------------------------------------------------- */
/* :281-303 reset */
static void space_reset(const struct pcto_env* h, struct cenv* s) {
  memset(s->upLetter, 0, sizeof(double) * 5 * h->I);
  memset(s->box_vec, 0, sizeof(double) * 9 * h->I);
  s->box_vec[8] = 1.0;
  memset(s->ems, 0, sizeof(double) * 6 * (size_t)(s->noems > 0 ? s->noems : 1));
  s->ems[0] = 0; s->ems[1] = 0; s->ems[2] = 0;
  s->ems[3] = h->cfg.container[0] / 1000; s->ems[4] = h->cfg.container[1] / 1000; s->ems[5] = h->cfg.container[2] / 1000;
  s->noems = 1;
  s->n_boxes = 0;
  s->box_idx = 0;
  if (s->stab) stab_reset(s->stab);
}

/* :305-314 interSect2D: max top of the placed boxes whose footprint overlaps `box`
 * (box = [-lx,-ly,lx+x,ly+y,0]); 0 if none */
static double intersect2d(const struct pcto_env* h, const struct cenv* s, const double box[5]) {
  (void)h;
  if (s->box_idx == 0) return 0.0;
  double max_h = 0.0;
  int any = 0;
  for (int i = 0; i < s->box_idx; i++) {
    const double* u = s->upLetter + 5 * i;
    double i0 = around6(fmin(box[0], u[0])), i1 = around6(fmin(box[1], u[1]));
    double i2 = around6(fmin(box[2], u[2])), i3 = around6(fmin(box[3], u[3]));
    if ((i0 + i2 > 0) && (i1 + i3 > 0)) {
      if (!any || u[4] > max_h) max_h = u[4];
      any = 1;
    }
  }
  return any ? max_h : 0.0;
}

/* :380-425 drop_box_virtual (setting 2: check_box is True, :430-431) */
static int drop_box_virtual(const struct pcto_env* h, const struct cenv* s, double x, double y, double z, double lx,
                            double ly) {
  double W = h->cfg.container[0] / 1000, L = h->cfg.container[1] / 1000, H = h->cfg.container[2] / 1000;
  int ok = 1;
  if (lx + x - 1e-6 > W || ly + y - 1e-6 > L) ok = 0;
  if (lx + 1e-6 < 0 || ly + 1e-6 < 0) ok = 0;
  double box[5] = {-lx, -ly, lx + x, ly + y, 0};
  double max_h = intersect2d(h, s, box);
  if (max_h + z - 1e-6 > H) ok = 0;
  /* :399-425: supporters / hull only if checkResult; check_box :428-439 */
  if (ok && h->cfg.setting != 2) ok = stab_check(s->stab, x, y, z, lx, ly, max_h, s->next_den, 1);
  return ok;
}

/* :329-376 drop_box */
static int drop_box(const struct pcto_env* h, struct cenv* s, const double bs[3], double lx, double ly, int flag,
                    uint32_t* flags) {
  double W = h->cfg.container[0] / 1000, L = h->cfg.container[1] / 1000, H = h->cfg.container[2] / 1000;
  double x, y, z;
  if (!flag) { x = bs[0]; y = bs[1]; z = bs[2]; }
  else       { y = bs[0]; x = bs[1]; z = bs[2]; }
  if (lx + x - 1e-6 > W || ly + y - 1e-6 > L) return 0;
  if (lx + 1e-6 < 0 || ly + 1e-6 < 0) return 0;
  double box[5] = {-lx, -ly, lx + x, ly + y, 0};
  double max_h = intersect2d(h, s, box);
  if (max_h + z - 1e-6 > H) return 0;
  box[4] = max_h + z;
  if (s->box_idx >= h->I) { /* IndexError :371 (raised after a successful check_box) */
    if (h->cfg.setting == 2 || stab_check(s->stab, x, y, z, lx, ly, max_h, s->next_den, 1)) *flags |= PCT_FLAG_INTERNAL_OVERFLOW;
    return 0;
  }
  if (h->cfg.setting != 2 && !stab_check(s->stab, x, y, z, lx, ly, max_h, s->next_den, 0)) return 0; /* :366 check_box */
  s->vol[s->n_boxes++] = x * y * z;
  memcpy(s->upLetter + 5 * s->box_idx, box, sizeof box);
  double* r = s->box_vec + 9 * s->box_idx;
  r[0] = lx; r[1] = ly; r[2] = max_h; r[3] = lx + x; r[4] = ly + y; r[5] = max_h + z; r[6] = 0; r[7] = 0; r[8] = 1;
  s->box_idx++;
  return 1;
}

/* :17-20 IsUsableEMS + :490-506 Difference/AddNewEMS */
static int usable(double lb, double x1, double y1, double z1, double x2, double y2, double z2) {
  return (x2 - x1 + 1e-6 >= lb) && (y2 - y1 + 1e-6 >= lb) && (z2 - z1 + 1e-6 >= lb);
}
static void add_ems(struct cenv* s, uint32_t* flags, double a, double b, double c, double x, double y, double z) {
  if (s->noems >= EMS_ROWS) { *flags |= PCT_FLAG_EMS_OVERFLOW; return; } /* IndexError :505 */
  double* e = s->ems + 6 * s->noems++;
  e[0] = a; e[1] = b; e[2] = c; e[3] = x; e[4] = y; e[5] = z;
}

/* :441-487 interSectEMS3D + GENEMS, :510-528 EliminateInscribedEMS */
static void genems(const struct pcto_env* h, struct cenv* s, const double loc[6], uint32_t* flags) {
  double lb = (double)h->low_bound / 1000.0;
  int origin = s->noems;
  double item[6] = {-loc[0], -loc[1], -loc[2], loc[3], loc[4], loc[5]};
  char* del = (char*)calloc((size_t)origin + 1, 1);
  int ndel = 0;
  for (int i = 0; i < origin; i++) {
    const double* e = s->ems + 6 * i;
    double q0 = around6(fmin(item[0], -e[0])), q1 = around6(fmin(item[1], -e[1])), q2 = around6(fmin(item[2], -e[2]));
    double q3 = around6(fmin(item[3], e[3])), q4 = around6(fmin(item[4], e[4])), q5 = around6(fmin(item[5], e[5]));
    if ((q0 + q3 > 0) && (q1 + q4 > 0) && (q2 + q5 > 0)) {
      del[i] = 1;
      ndel++;
      double x1 = e[0], y1 = e[1], z1 = e[2], x2 = e[3], y2 = e[4], z2 = e[5];
      double x3 = -q0, y3 = -q1, z3 = -q2, x4 = q3, y4 = q4, z4 = q5; /* intersect[:,0:3] *= -1 */
      (void)z3;
      if (usable(lb, x1, y1, z1, x3, y2, z2)) add_ems(s, flags, x1, y1, z1, x3, y2, z2);
      if (usable(lb, x4, y1, z1, x2, y2, z2)) add_ems(s, flags, x4, y1, z1, x2, y2, z2);
      if (usable(lb, x1, y1, z1, x2, y3, z2)) add_ems(s, flags, x1, y1, z1, x2, y3, z2);
      if (usable(lb, x1, y4, z1, x2, y2, z2)) add_ems(s, flags, x1, y4, z1, x2, y2, z2);
      if (usable(lb, x1, y1, z4, x2, y2, z2)) add_ems(s, flags, x1, y1, z4, x2, y2, z2);
    }
  }
  if (ndel) {
    int m = 0, total = s->noems;
    for (int i = 0; i < total; i++) {
      if (i < origin && del[i]) continue;
      if (m != i) memcpy(s->ems + 6 * m, s->ems + 6 * i, 6 * sizeof(double));
      m++;
    }
    memset(s->ems + 6 * m, 0, sizeof(double) * 6 * (size_t)(total - m));
    s->noems = m;
  }
  free(del);
  /* EliminateInscribedEMS */
  int n = s->noems;
  char* d2 = (char*)calloc((size_t)n + 1, 1);
  for (int i = 0; i < n; i++) {
    const double* a = s->ems + 6 * i;
    for (int j = 0; j < n; j++) {
      if (i == j) continue;
      const double* b = s->ems + 6 * j;
      if (a[0] >= b[0] && a[1] >= b[1] && a[2] >= b[2] && a[3] <= b[3] && a[4] <= b[4] && a[5] <= b[5]) {
        d2[i] = 1;
        break;
      }
    }
  }
  int m = 0;
  for (int i = 0; i < n; i++)
    if (!d2[i]) {
      if (m != i) memcpy(s->ems + 6 * m, s->ems + 6 * i, 6 * sizeof(double));
      m++;
    }
  memset(s->ems + 6 * m, 0, sizeof(double) * 6 * (size_t)(n - m));
  s->noems = m;
  free(d2);
}

/* :531-568 EMSPoint (set iteration order) */
static int ems_point(const struct pcto_env* h, const struct cenv* s, double** out) {
  int orientation = h->cfg.setting == 2 ? 6 : 2;
  const double* nb = s->next_box;
  double* keys = (double*)malloc(sizeof(double) * 6 * (size_t)(s->noems * orientation * 4 + 1));
  int nk = 0;
  for (int ei = 0; ei < s->noems; ei++) {
    const double* ems = s->ems + 6 * ei;
    for (int rot = 0; rot < orientation; rot++) {
      double sx, sy, sz;
      switch (rot) {
        case 0: sx = nb[0]; sy = nb[1]; sz = nb[2]; break;
        case 1: sx = nb[1]; sy = nb[0]; sz = nb[2]; if (fabs(sx - sy) < 1e-6) continue; break;
        case 2: sx = nb[0]; sy = nb[2]; sz = nb[1]; if (fabs(sx - sy) < 1e-6 && fabs(sy - sz) < 1e-6) continue; break;
        case 3: sx = nb[1]; sy = nb[2]; sz = nb[0]; if (fabs(sx - sy) < 1e-6 && fabs(sy - sz) < 1e-6) continue; break;
        case 4: sx = nb[2]; sy = nb[0]; sz = nb[1]; if (fabs(sx - sy) < 1e-6) continue; break;
        default: sx = nb[2]; sy = nb[1]; sz = nb[0]; if (fabs(sx - sy) < 1e-6) continue; break;
      }
      if (ems[3] - ems[0] + 1e-6 >= sx && ems[4] - ems[1] + 1e-6 >= sy && ems[5] - ems[2] + 1e-6 >= sz) {
        double c[4][6] = {{ems[0], ems[1], ems[2], ems[0] + sx, ems[1] + sy, ems[2] + sz},
                          {ems[3] - sx, ems[1], ems[2], ems[3], ems[1] + sy, ems[2] + sz},
                          {ems[0], ems[4] - sy, ems[2], ems[0] + sx, ems[4], ems[2] + sz},
                          {ems[3] - sx, ems[4] - sy, ems[2], ems[3], ems[4], ems[2] + sz}};
        memcpy(keys + 6 * (size_t)nk, c, sizeof c);
        nk += 4;
      }
    }
  }
  dset st;
  st.mask = 7; st.fill = 0;
  st.slot = (int32_t*)malloc(8 * sizeof(int32_t));
  st.hash = (uint64_t*)malloc(8 * sizeof(uint64_t));
  for (int i = 0; i < 8; i++) st.slot[i] = -1;
  for (int k = 0; k < nk; k++) dset_add(&st, keys, k);
  double* res = (double*)malloc(sizeof(double) * 6 * (st.fill + 1));
  int cnt = 0;
  for (size_t i = 0; i <= st.mask; i++)
    if (st.slot[i] >= 0) memcpy(res + 6 * (size_t)cnt++, keys + 6 * (size_t)st.slot[i], 6 * sizeof(double));
  free(st.slot); free(st.hash); free(keys);
  *out = res;
  return cnt;
}

/* C/bin3D.py:118-148 get_possible_position */
static void get_possible_position(const struct pcto_env* h, int e, struct cenv* s, double* leaf) {
  memset(leaf, 0, sizeof(double) * 9 * h->L);
  double* pos = NULL;
  int n = ems_point(h, s, &pos), idx = 0;
  if (h->cfg.shuffle && h->rng_numpy) { /* np.random.shuffle(allPostion), legacy Fisher-Yates on random_interval */
    for (int i = n - 1; i >= 1; i--) {
      int j = (int)npmt_interval(s->mt, &s->mt_pos, (uint32_t)i);
      if (j == i) continue;
      for (int c = 0; c < 6; c++) { double t_ = pos[6 * i + c]; pos[6 * i + c] = pos[6 * j + c]; pos[6 * j + c] = t_; }
    }
  } else if (h->cfg.shuffle && n > 1) { /* C/bin3D.py:126-127 -> include/pct_env.h pct_shuffle_priority */
    uint32_t* pr = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    int* ord = (int*)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) { pr[i] = pct_shuffle_priority(h->shuffle_seed, (uint64_t)(h->cfg.env_id_base + e), s->oc, (uint32_t)i); ord[i] = i; }
    for (int i = 1; i < n; i++) { int v = ord[i], j = i; while (j > 0 && pr[ord[j - 1]] > pr[v]) { ord[j] = ord[j - 1]; j--; } ord[j] = v; }
    double* tmp = (double*)malloc(sizeof(double) * 6 * (size_t)n);
    for (int i = 0; i < n; i++) memcpy(tmp + 6 * (size_t)i, pos + 6 * (size_t)ord[i], 6 * sizeof(double));
    memcpy(pos, tmp, sizeof(double) * 6 * (size_t)n);
    free(tmp); free(ord); free(pr);
  }
  s->oc++;
  double H = h->cfg.container[2] / 1000;
  for (int i = 0; i < n; i++) {
    const double* p = pos + 6 * i;
    double x = p[3] - p[0], y = p[4] - p[1], z = p[5] - p[2];
    if (drop_box_virtual(h, s, x, y, z, p[0], p[1])) {
      double* r = leaf + 9 * idx;
      r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = p[3]; r[4] = p[4]; r[5] = H; r[6] = 0; r[7] = 0; r[8] = 1;
      idx++;
    }
    if (idx >= h->L) break;
  }
  free(pos);
}

/* C/bin3D.py:78-100 cur_observation (scripted items: sample_from_distribution=False path) */
static void cur_observation(const struct pcto_env* h, int e, struct cenv* s, double* obs) {
  if (h->rng_numpy) {
    /* sampling mode draws a NEW item inside EVERY cur_observation (C/bin3D.py:81,103-113) -- also inside the
     * discarded observation of a failed step -- and ignores the BoxCreator queue */
    draw_item_numpy(h, s, s->next_box);
    s->next_den = 1.0;
    if (h->cfg.setting == 3) { do { s->next_den = npmt_double(s->mt, &s->mt_pos); } while (s->next_den == 0); }
  } else {
  if (s->queue_len < 1) { draw_item(h, e, s, s->queue_item); s->queue_len = 1; }
  memcpy(s->next_box, s->queue_item, sizeof s->next_box);
  s->next_den = pcto_next_density(h, e, s->oc, s->traj, s->cursor - 1); /* C/bin3D.py:81-90 */
  }
  memcpy(obs, s->box_vec, sizeof(double) * 9 * h->I);
  get_possible_position(h, e, s, obs + 9 * h->I);
  double a = s->next_box[0], b = s->next_box[1], c = s->next_box[2], t;
  if (a > b) { t = a; a = b; b = t; }
  if (b > c) { t = b; b = c; c = t; }
  if (a > b) { t = a; a = b; b = t; }
  double* r = obs + 9 * (h->I + h->L);
  r[0] = s->next_den; r[1] = 0; r[2] = 0; r[3] = a; r[4] = b; r[5] = c; r[6] = 0; r[7] = 0; r[8] = 1;
}

static double get_ratio(const struct pcto_env* h, const struct cenv* s) {
  double vo = 0.0;
  for (int i = 0; i < s->n_boxes; i++) vo = vo + s->vol[i];
  double mx = (double)((int64_t)(h->cfg.container[0] / 1000) * (h->cfg.container[1] / 1000) * (h->cfg.container[2] / 1000));
  return vo / mx;
}

void pctc_reset(struct pcto_env* h, int e, double* obs) {
  struct cenv* s = &h->cenvs[e];
  s->queue_len = 0;
  if (h->source == PCT_ITEMS_DATASET) {
    s->traj++;
    s->cursor = 0;
    if (s->traj >= h->ds_ntraj) h->flags[e] |= PCT_FLAG_DATASET_EXHAUSTED;
  }
  space_reset(h, s);
  if (h->rng_numpy) (void)npmt_interval(s->mt, &s->mt_pos, (uint32_t)h->n_items - 1u); /* box_creator.generate_box_size(): a randint nobody reads */
  else draw_item(h, e, s, s->queue_item);
  s->queue_len = 1;
  cur_observation(h, e, s, obs);
}

/* C/bin3D.py:151-167 LeafNode2Action + :169-207 step */
void pctc_step(struct pcto_env* h, int e, const double* act, int len, double* obs, double* reward, uint8_t* done,
               int32_t* counter, double* ratio, uint32_t* flags) {
  struct cenv* s = &h->cenvs[e];
  s->t++;
  int flag;
  double a1, a2, nb[3];
  if (len != 3) {
    double sum = 0;
    for (int i = 0; i < 6; i++) sum += act[i];
    if (sum == 0) {
      flag = 0; a1 = 0; a2 = 0;
      memcpy(nb, s->next_box, sizeof nb);
    } else {
      double x = around6(act[3] - act[0]);
      double y = around6(act[4] - act[1]);
      int rec[3] = {0, 1, 2}, nr = 3;
      for (int i = 0; i < nr; i++)
        if (fabs(x - s->next_box[rec[i]]) < 1e-6) { for (int j = i; j < nr - 1; j++) rec[j] = rec[j + 1]; nr--; break; }
      for (int i = 0; i < nr; i++)
        if (fabs(y - s->next_box[rec[i]]) < 1e-6) { for (int j = i; j < nr - 1; j++) rec[j] = rec[j + 1]; nr--; break; }
      flag = 0; a1 = act[0]; a2 = act[1];
      nb[0] = x; nb[1] = y; nb[2] = s->next_box[rec[0]];
    }
  } else {
    flag = (int)act[0]; a1 = act[1]; a2 = act[2];
    memcpy(nb, s->next_box, sizeof nb);
  }
  double lx = around6(a1), ly = around6(a2); /* idx = [round(action[1],6), round(action[2],6)] */
  double x = flag ? nb[1] : nb[0], y = flag ? nb[0] : nb[1], z = nb[2];
  if (!drop_box(h, s, nb, lx, ly, flag, flags)) {
    *reward = 0.0; *done = 1; *counter = s->n_boxes; *ratio = get_ratio(h, s);
    cur_observation(h, e, s, obs);
    return;
  }
  double lz = s->box_vec[9 * (s->box_idx - 1) + 2];
  double loc[6] = {lx, ly, lz, around6(lx + x), around6(ly + y), around6(lz + z)};
  genems(h, s, loc, flags);
  double mx = (double)((int64_t)(h->cfg.container[0] / 1000) * (h->cfg.container[1] / 1000) * (h->cfg.container[2] / 1000));
  double box_ratio = (s->next_box[0] * s->next_box[1] * s->next_box[2]) / mx;
  s->queue_len = 0;
  if (h->rng_numpy) (void)npmt_interval(s->mt, &s->mt_pos, (uint32_t)h->n_items - 1u); /* generate_box_size() */
  else draw_item(h, e, s, s->queue_item);
  s->queue_len = 1;
  *reward = box_ratio * 10;
  *done = 0;
  *counter = s->n_boxes;
  *ratio = 0.0;
  cur_observation(h, e, s, obs);
}

/* ---- heuristic.py on PackingContinuous (tools.py:217-218: only LSAH, OnlineBPH and BR are allowed there) -------------
 * The baselines as in-env policies, like pct_oracle.c heur_choose for the discrete env.  Everything is float64 in the
 * reference's own operation order; env.space.EMS is the whole 1000-row array there, of which only the live rows can
 * pass the tests (an all-zero row is skipped by LASH / OnlineBPH and fits no item in BR). */
static void cheur_rot(const double nb[3], int rot, double* x, double* y, double* z) { /* heuristic.py:171-182 */
  switch (rot) {
    case 0: *x = nb[0]; *y = nb[1]; *z = nb[2]; break;
    case 1: *y = nb[0]; *x = nb[1]; *z = nb[2]; break;
    case 2: *z = nb[0]; *x = nb[1]; *y = nb[2]; break;
    case 3: *z = nb[0]; *y = nb[1]; *x = nb[2]; break;
    case 4: *x = nb[0]; *z = nb[1]; *y = nb[2]; break;
    default: *y = nb[0]; *z = nb[1]; *x = nb[2]; break;
  }
}
/* drop_box_virtual(..., returnH=True) (C/space.py:380-425): feasibility and the height the box would rest at */
static int drop_box_virtual_h(const struct pcto_env* h, const struct cenv* s, double x, double y, double z, double lx,
                              double ly, double* height) {
  double W = h->cfg.container[0] / 1000, L = h->cfg.container[1] / 1000, H = h->cfg.container[2] / 1000;
  int ok = 1;
  if (lx + x - 1e-6 > W || ly + y - 1e-6 > L) ok = 0;
  if (lx + 1e-6 < 0 || ly + 1e-6 < 0) ok = 0;
  double box[5] = {-lx, -ly, lx + x, ly + y, 0};
  double max_h = intersect2d(h, s, box);
  if (max_h + z - 1e-6 > H) ok = 0;
  if (ok && h->cfg.setting != 2) ok = stab_check(s->stab, x, y, z, lx, ly, max_h, s->next_den, 1);
  *height = max_h;
  return ok;
}
int pctc_heur_choose(const struct pcto_env* h, int e, int kind, double* olx, double* oly, double* ox, double* oy, double* oz) {
  const struct cenv* s = &h->cenvs[e];
  const int orientation = h->cfg.setting == 2 ? 6 : 2;
  const double* nb = s->next_box;
  int found = 0;
  if (kind == PCT_HEUR_OBPH) {
    /* heuristic.py:364-425: EMS sorted by (z, y, x), stable; the first feasible (EMS, rotation) -- no fit-in-EMS test */
    int n = s->noems;
    int* idx = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
      int j = i;
      const double* ei = s->ems + 6 * i;
      while (j > 0) {
        const double* ep = s->ems + 6 * idx[j - 1];
        int greater = ep[2] > ei[2] || (ep[2] == ei[2] && (ep[1] > ei[1] || (ep[1] == ei[1] && ep[0] > ei[0])));
        if (!greater) break;
        idx[j] = idx[j - 1];
        j--;
      }
      idx[j] = i;
    }
    for (int q = 0; q < n && !found; q++) {
      const double* em = s->ems + 6 * idx[q];
      if (fabs(em[0]) + fabs(em[1]) + fabs(em[2]) + fabs(em[3]) + fabs(em[4]) + fabs(em[5]) == 0) continue;
      for (int rot = 0; rot < orientation; rot++) {
        double x, y, z, hh;
        cheur_rot(nb, rot, &x, &y, &z);
        if (drop_box_virtual_h(h, s, x, y, z, em[0], em[1], &hh)) {
          found = 1; *olx = em[0]; *oly = em[1]; *ox = x; *oy = y; *oz = z;
          break;
        }
      }
    }
    free(idx);
    return found;
  }
  if (kind == PCT_HEUR_LSAH) {
    /* :138-226 LASH: least surface area of the bounding box of everything packed; maxXY / minXY are the running extents
     * of this episode's placements (:204-207): lx + x and lx of every placed box, as the reference accumulates them */
    const double W = h->cfg.container[0] / 1000, L = h->cfg.container[1] / 1000, H = h->cfg.container[2] / 1000;
    double maxX = 0, maxY = 0, minX = W, minY = L;
    for (int i = 0; i < s->box_idx; i++) {
      const double* r = s->box_vec + 9 * i;
      if (r[3] > maxX) maxX = r[3];
      if (r[4] > maxY) maxY = r[4];
      if (r[0] < minX) minX = r[0];
      if (r[1] < minY) minY = r[1];
    }
    double best = W * L + L * H + H * W; /* ints in the reference: exact */
    double bd[3] = {0, 0, 0};
    for (int q = 0; q < s->noems; q++) {
      const double* em = s->ems + 6 * q;
      if (fabs(em[0]) + fabs(em[1]) + fabs(em[2]) + fabs(em[3]) + fabs(em[4]) + fabs(em[5]) == 0) continue;
      double dx = em[3] - em[0], dy = em[4] - em[1], dz = em[5] - em[2];
      for (int rot = 0; rot < orientation; rot++) {
        double x, y, z, height;
        cheur_rot(nb, rot, &x, &y, &z);
        if (!(dx >= x && dy >= y && dz >= z)) continue;
        double lx = em[0], ly = em[1];
        if (!drop_box_virtual_h(h, s, x, y, z, lx, ly, &height)) continue;
        double ex = fmax(lx + x, maxX) - fmin(lx, minX), ey = fmax(ly + y, maxY) - fmin(ly, minY);
        double score = ex * ey + (height + z) * ey + (height + z) * ex;
        int take = 0;
        if (score < best) take = 1;
        else if (score == best && found) {
          double m1 = fmin(fmin(dx - x, dy - y), dz - z), m2 = fmin(fmin(bd[0] - x, bd[1] - y), bd[2] - z);
          if (m1 < m2) take = 1;
        }
        if (take) {
          best = score; found = 1; *olx = lx; *oly = ly; *ox = x; *oy = y; *oz = z;
          bd[0] = dx; bd[1] = dy; bd[2] = dz;
        }
      }
    }
    return found;
  }
  if (kind == PCT_HEUR_BR) {
    /* :500-569 BR: the EMS with the best eval_ems (volume + number of item types that fit unrotated, + 10 if all do);
     * first best in (EMS, rotation) order.  env.item_set: the item set the env was handed (integers) */
    double best = -1e10;
    for (int q = 0; q < s->noems; q++) {
      const double* em = s->ems + 6 * q;
      double dx = em[3] - em[0], dy = em[4] - em[1], dz = em[5] - em[2];
      double sc = 0;
      int have = 0;
      for (int rot = 0; rot < orientation; rot++) {
        double x, y, z, height;
        cheur_rot(nb, rot, &x, &y, &z);
        if (!(dx >= x && dy >= y && dz >= z)) continue;
        if (!drop_box_virtual_h(h, s, x, y, z, em[0], em[1], &height)) continue;
        if (!have) {
          int valid = 0;
          for (int i = 0; i < h->n_items; i++)
            if (dx >= h->item_set[3 * i] / 1000.0 && dy >= h->item_set[3 * i + 1] / 1000.0 && dz >= h->item_set[3 * i + 2] / 1000.0) valid++;
          sc = 0;
          sc += dx * dy * dz;
          sc += valid;
          if (valid == h->n_items) sc += 10;
          have = 1;
        }
        if (sc > best) {
          best = sc; found = 1; *olx = em[0]; *oly = em[1]; *ox = x; *oy = y; *oz = z;
        }
      }
    }
    return found;
  }
  return 0;
}
/* env.next_box = [x, y, z]; env.step([0, lx, ly]) (heuristic.py:215, 416, 560) */
void pctc_step_place(struct pcto_env* h, int e, double lx, double ly, double x, double y, double z, double* obs) {
  struct cenv* s = &h->cenvs[e];
  s->next_box[0] = x; s->next_box[1] = y; s->next_box[2] = z;
  const double act[3] = {0, lx, ly};
  pctc_step(h, e, act, 3, obs, &h->reward[e], &h->done[e], &h->counter[e], &h->ratio[e], &h->flags[e]);
}
/* a heuristic found no placement: the episode is recorded and the env reset, no step() */
void pctc_giveup(struct pcto_env* h, int e) {
  struct cenv* s = &h->cenvs[e];
  s->t++;
  h->reward[e] = 0.0;
  h->done[e] = 1;
  h->counter[e] = s->n_boxes;
  h->ratio[e] = get_ratio(h, s);
}

uint32_t pctc_t(const struct pcto_env* h, int e) { return h->cenvs[e].t; }
const struct stab* pctc_stab(const struct pcto_env* h, int e) { return h->cenvs[e].stab; }

int pctc_alloc(struct pcto_env* h) {
  for (int d = 0; d < 3; d++)
    if (h->cfg.container[d] % 1000 != 0) return 1; /* integer container sizes only */
  h->cenvs = (struct cenv*)calloc((size_t)h->N, sizeof(struct cenv));
  for (int e = 0; e < h->N; e++) {
    struct cenv* s = &h->cenvs[e];
    s->upLetter = (double*)calloc((size_t)h->I * 5, sizeof(double));
    s->box_vec = (double*)calloc((size_t)h->I * 9, sizeof(double));
    s->vol = (double*)calloc((size_t)h->I + 1, sizeof(double));
    s->ems = (double*)calloc((size_t)EMS_ROWS * 6, sizeof(double));
    if (h->cfg.setting != 2) s->stab = stab_create(h->I, 1e-6);
  }
  return 0;
}
void pctc_free(struct pcto_env* h) {
  for (int e = 0; e < h->N; e++) {
    free(h->cenvs[e].upLetter); free(h->cenvs[e].box_vec); free(h->cenvs[e].vol); free(h->cenvs[e].ems);
    stab_free(h->cenvs[e].stab);
  }
  free(h->cenvs);
  h->cenvs = NULL;
}
int pctc_debug_state(struct pcto_env* h, int e, double* ems, int cap_ems, int* n_ems, int* n_boxes, double* next_item,
                     int64_t* cursor) {
  const struct cenv* s = &h->cenvs[e];
  if (ems) memcpy(ems, s->ems, sizeof(double) * 6 * (size_t)(s->noems < cap_ems ? s->noems : cap_ems));
  if (n_ems) *n_ems = s->noems;
  if (n_boxes) *n_boxes = s->n_boxes;
  if (next_item) memcpy(next_item, s->next_box, 3 * sizeof(double));
  if (cursor) *cursor = (int64_t)s->cursor;
  return 0;
}

/* strict NumPy-stream mode for the continuous env (sampling from U(a,b)): `n_item_set` = len(item_set) the
 * reference's RandomBoxCreator draws its (unused) index from (givenData.item_size_set: 125) */
void pctc_set_numpy_rng(struct pcto_env* h, uint32_t seed) {
  for (int e = 0; e < h->N; e++) npmt_seed(h->cenvs[e].mt, &h->cenvs[e].mt_pos, seed + (uint32_t)h->cfg.env_id_base + (uint32_t)e);
}
