/*
 * pct_oracle.c -- CPU restatement of the reference PCT env hot path (TEST INFRASTRUCTURE;
 * see pct_oracle.h for who may use it and for the parity status).
 *
 * Every function cites the reference lines it follows (paths relative to the reference
 * repo; "D/" = pct_envs/PctDiscrete0/).  The code is written for fidelity, not speed.
 */
#include "pct_oracle_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[256];
static int g_threads = 1;

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
const char* pcto_last_error(void) { return g_err; }
int pcto_num_threads(void) { return g_threads; }
void pcto_set_num_threads(int n) { g_threads = n < 1 ? 1 : n; }

/* ===================================================================================== */
/* CPython 3.10 set-of-tuples emulation                                                   */
/* ===================================================================================== */
/* Objects/tupleobject.c tuplehash (xxHash-derived, CPython >= 3.8); hash(int v) == v for
 * 0 <= v < 2^61-1 (Objects/longobject.c long_hash). */
#define XXPRIME_1 11400714785074694791ULL
#define XXPRIME_2 14029467366897019727ULL
#define XXPRIME_5 2870177450012600261ULL

static uint64_t py_tuplehash(const int64_t* v, int len) {
  uint64_t acc = XXPRIME_5;
  for (int i = 0; i < len; i++) {
    /* small ints hash to themselves, except hash(-1) == -2 (long_hash) */
    uint64_t lane = v[i] == -1 ? (uint64_t)(int64_t)-2 : (uint64_t)v[i];
    acc += lane * XXPRIME_2;
    acc = (acc << 31) | (acc >> 33);
    acc *= XXPRIME_1;
  }
  acc += (uint64_t)len ^ (XXPRIME_5 ^ 3527539ULL);
  if (acc == (uint64_t)-1) return 1546275796ULL;
  return acc;
}
static uint64_t py_tuplehash6(const int64_t* v) { return py_tuplehash(v, 6); }

/* Objects/setobject.c: set_add_entry (LINEAR_PROBES 9, PERTURB_SHIFT 5, growth when
 * fill*5 >= mask*3 to the first power of two > used*4), set_table_resize +
 * set_insert_clean (re-insertion in old-table slot order), iteration = slot order. */
typedef struct {
  int32_t* slot;  /* -1 empty, else index into keys */
  uint64_t* hash; /* per slot */
  size_t mask;
  size_t fill;
} pyset;

static void pyset_init(pyset* s) {
  s->mask = 7;
  s->fill = 0;
  s->slot = (int32_t*)malloc(8 * sizeof(int32_t));
  s->hash = (uint64_t*)malloc(8 * sizeof(uint64_t));
  for (int i = 0; i < 8; i++) s->slot[i] = -1;
}
static void pyset_free(pyset* s) {
  free(s->slot);
  free(s->hash);
}
static void pyset_insert_clean(int32_t* slot, uint64_t* hs, size_t mask, int32_t key, uint64_t hash) {
  size_t perturb = hash;
  size_t i = (size_t)hash & mask;
  while (1) {
    size_t e = i;
    if (slot[e] < 0) goto found;
    if (i + 9 <= mask) {
      for (int j = 0; j < 9; j++) {
        e++;
        if (slot[e] < 0) goto found;
      }
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
    continue;
  found:
    slot[e] = key;
    hs[e] = hash;
    return;
  }
}
static void pyset_resize(pyset* s, size_t minused) {
  size_t newsize = 8;
  while (newsize <= minused) newsize <<= 1;
  int32_t* ns = (int32_t*)malloc(newsize * sizeof(int32_t));
  uint64_t* nh = (uint64_t*)malloc(newsize * sizeof(uint64_t));
  for (size_t i = 0; i < newsize; i++) ns[i] = -1;
  for (size_t i = 0; i <= s->mask; i++)
    if (s->slot[i] >= 0) pyset_insert_clean(ns, nh, newsize - 1, s->slot[i], s->hash[i]);
  free(s->slot);
  free(s->hash);
  s->slot = ns;
  s->hash = nh;
  s->mask = newsize - 1;
}
static void pyset_add(pyset* s, const int64_t* keys, int32_t k) {
  const int64_t* key = keys + 6 * (size_t)k;
  uint64_t hash = py_tuplehash6(key);
  size_t mask = s->mask;
  size_t i = (size_t)hash & mask;
  size_t perturb = hash;
  while (1) {
    size_t e = i;
    int probes = (i + 9 <= mask) ? 9 : 0;
    do {
      if (s->slot[e] < 0) {
        s->slot[e] = k;
        s->hash[e] = hash;
        s->fill++;
        if (s->fill * 5 >= mask * 3) pyset_resize(s, s->fill > 50000 ? s->fill * 2 : s->fill * 4);
        return;
      }
      if (s->hash[e] == hash && memcmp(keys + 6 * (size_t)s->slot[e], key, 6 * sizeof(int64_t)) == 0)
        return;
      e++;
    } while (probes--);
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}
int pcto_pyset_order(const int64_t* keys, int32_t n, int32_t* order_out) {
  pyset s;
  pyset_init(&s);
  for (int32_t k = 0; k < n; k++) pyset_add(&s, keys, k);
  int cnt = 0;
  for (size_t i = 0; i <= s.mask; i++)
    if (s.slot[i] >= 0) order_out[cnt++] = s.slot[i];
  pyset_free(&s);
  return cnt;
}

/* ===================================================================================== */
/* per-env state                                                                          */
/* ===================================================================================== */
/* item source shared with the HIP path (include/pct_env.h pct_set_item_stream /
 * pct_set_sampler); stands in for binCreator.py:37-39 generate_box_size */
static void draw_item(const pcto_env* h, int e, oenv* s, int out[3]) {
  uint64_t c = s->cursor++;
  if (h->source == PCT_ITEMS_DATASET) { /* binCreator.py:64-72 LoadBoxCreator.generate_box_size */
    int t = s->traj < h->ds_ntraj ? s->traj : h->ds_ntraj - 1;
    int len = h->ds_len[t];
    if (c < (uint64_t)len) {
      const int32_t* p = h->stream + ((size_t)t * h->ds_maxlen + (size_t)c) * 3;
      out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
    } else {
      out[0] = out[1] = out[2] = (c == (uint64_t)len) ? 100 : 10; /* sentinel :62, then (10,10,10) :69-72 */
    }
  } else if (h->source == PCT_ITEMS_STREAM) {
    const int32_t* p = h->stream + ((size_t)e * (size_t)h->T + (size_t)(c % (uint64_t)h->T)) * 3;
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
  } else {
    uint64_t g = (uint64_t)(h->cfg.env_id_base + e);
    uint64_t idx = h->rng_numpy ? npmt_interval(s->mt, &s->mt_pos, (uint32_t)h->n_items - 1u) /* np.random.randint(0, n), binCreator.py:38 */
                                : pct_pick(h->seed, g, c, (uint32_t)h->n_items);
    out[0] = h->item_set[idx * 3 + 0];
    out[1] = h->item_set[idx * 3 + 1];
    out[2] = h->item_set[idx * 3 + 2];
  }
}

/* D/space.py:290-314 Space.reset (ZMAP/EMS3D bookkeeping is unobservable under
 * LNES='EMS', SURVEY.md a13, and is not restated) */
static void space_reset(const pcto_env* h, oenv* s) {
  memset(s->plain, 0, sizeof(int) * h->A * h->A);
  memset(s->box_vec, 0, sizeof(double) * h->I * 9);
  s->box_vec[8] = 1.0; /* box_vec[0][-1] = 1 */
  s->n_ems = 1;
  s->ems[0] = 0; s->ems[1] = 0; s->ems[2] = 0;
  s->ems[3] = h->cfg.container[0]; s->ems[4] = h->cfg.container[1]; s->ems[5] = h->cfg.container[2];
  s->n_boxes = 0;
  s->box_idx = 0;
  if (s->stab) stab_reset(s->stab);
}

/* np.max(self.plain[lx:lx+x, ly:ly+y]) with NumPy slice clipping (D/space.py:354-355,
 * 400-401).  Returns -1 for an empty slice (NumPy raises ValueError there). */
static int footprint_max(const pcto_env* h, const oenv* s, int lx, int ly, int x, int y) {
  int A = h->A;
  /* Python slice semantics for possibly negative / out-of-range bounds */
  int x0 = lx, x1 = lx + x, y0 = ly, y1 = ly + y;
  if (x0 < 0) { x0 += A; if (x0 < 0) x0 = 0; }
  if (x1 < 0) { x1 += A; if (x1 < 0) x1 = 0; }
  if (y0 < 0) { y0 += A; if (y0 < 0) y0 = 0; }
  if (y1 < 0) { y1 += A; if (y1 < 0) y1 = 0; }
  if (x0 > A) x0 = A;
  if (x1 > A) x1 = A;
  if (y0 > A) y0 = A;
  if (y1 > A) y1 = A;
  if (x1 <= x0 || y1 <= y0) return -1;
  int m = 0;
  for (int i = x0; i < x1; i++)
    for (int j = y0; j < y1; j++)
      if (s->plain[i * A + j] > m) m = s->plain[i * A + j];
  return m;
}

/* D/space.py:436-454 check_box; the stability branch (:447-454) lives in pct_oracle_stab.c */
static int check_box(const pcto_env* h, const oenv* s, int x, int y, int lx, int ly, int z, int max_h, double density,
                     int virtual_) {
  if (lx + x > h->cfg.container[0] || ly + y > h->cfg.container[1]) return 0;
  if (lx < 0 || ly < 0) return 0;
  if (max_h + z > h->cfg.container[2]) return 0; /* self.height stays == H (space.py:383) */
  if (h->cfg.setting == 2) return 1;
  return stab_check(s->stab, x, y, z, lx, ly, max_h, density, virtual_);
}

/* D/space.py:393-433 drop_box_virtual(box_size, idx, False, den, setting) */
static int drop_box_virtual(const pcto_env* h, const oenv* s, int x, int y, int z, int lx, int ly) {
  int max_h = footprint_max(h, s, lx, ly, x, y);
  if (max_h < 0) return 0; /* unreachable for EMS-generated candidates */
  return check_box(h, s, x, y, lx, ly, z, max_h, s->next_den, 1);
}

/* D/space.py:347-389 drop_box.  Returns 1 ok, 0 infeasible; sets *flags on what the
 * reference would raise. */
static int drop_box(const pcto_env* h, oenv* s, const int box[3], int lx, int ly, int flag, double density,
                    uint32_t* flags) {
  int x, y, z;
  if (!flag) { x = box[0]; y = box[1]; z = box[2]; }
  else       { y = box[0]; x = box[1]; z = box[2]; }
  int max_h = footprint_max(h, s, lx, ly, x, y);
  if (max_h < 0) { *flags |= PCT_FLAG_BAD_ACTION; return 0; } /* ValueError in np.max */
  if (s->box_idx >= h->I) { /* IndexError :385 (would be raised after a successful check) */
    if (h->cfg.setting == 2 ? check_box(h, s, x, y, lx, ly, z, max_h, density, 0)
                            : check_box(h, s, x, y, lx, ly, z, max_h, density, 1))
      *flags |= PCT_FLAG_INTERNAL_OVERFLOW;
    return 0;
  }
  if (!check_box(h, s, x, y, lx, ly, z, max_h, density, 0)) return 0;
  obox b = {x, y, z, lx, ly, max_h};
  s->boxes[s->n_boxes++] = b;
  /* update_height_graph :316-326 */
  int A = h->A, top = max_h + z;
  for (int i = lx; i < lx + x; i++)
    for (int j = ly; j < ly + y; j++) s->plain[i * A + j] = top;
  double* r = s->box_vec + 9 * s->box_idx;
  r[0] = lx; r[1] = ly; r[2] = max_h; r[3] = lx + x; r[4] = ly + y; r[5] = max_h + z;
  r[6] = density; r[7] = 0; r[8] = 1;
  s->box_idx++;
  return 1;
}

/* D/space.py:514-515 AddNewEMS */
static int add_ems(const pcto_env* h, oenv* s, int64_t a, int64_t b, int64_t c, int64_t x, int64_t y, int64_t z) {
  (void)h;
  if (s->n_ems >= s->cap_ems) {
    s->cap_ems *= 2;
    s->ems = (int64_t*)realloc(s->ems, sizeof(int64_t) * 6 * s->cap_ems);
  }
  int64_t* e = s->ems + 6 * s->n_ems++;
  e[0] = a; e[1] = b; e[2] = c; e[3] = x; e[4] = y; e[5] = z;
  return 0;
}

/* D/space.py:18-24 IsUsableEMS + :498-512 Difference (low_bound 0 -> 0.1, :501-502) */
static void difference(const pcto_env* h, oenv* s, int emsID, const int64_t inter[6]) {
  int64_t x1 = s->ems[6 * emsID + 0], y1 = s->ems[6 * emsID + 1], z1 = s->ems[6 * emsID + 2];
  int64_t x2 = s->ems[6 * emsID + 3], y2 = s->ems[6 * emsID + 4], z2 = s->ems[6 * emsID + 5];
  int64_t x3 = inter[0], y3 = inter[1], z3 = inter[2], x4 = inter[3], y4 = inter[4], z4 = inter[5];
  (void)z3;
  int64_t lb = h->low_bound <= 0 ? 1 : h->low_bound; /* ints: >= 0.1  <=>  >= 1 */
#define USABLE(ax, ay, az, bx, by, bz) (((bx) - (ax) >= lb) && ((by) - (ay) >= lb) && ((bz) - (az) >= lb))
  if (USABLE(x1, y1, z1, x3, y2, z2)) add_ems(h, s, x1, y1, z1, x3, y2, z2);
  if (USABLE(x4, y1, z1, x2, y2, z2)) add_ems(h, s, x4, y1, z1, x2, y2, z2);
  if (USABLE(x1, y1, z1, x2, y3, z2)) add_ems(h, s, x1, y1, z1, x2, y3, z2);
  if (USABLE(x1, y4, z1, x2, y2, z2)) add_ems(h, s, x1, y4, z1, x2, y2, z2);
  if (USABLE(x1, y1, z4, x2, y2, z2)) add_ems(h, s, x1, y1, z4, x2, y2, z2);
#undef USABLE
}

/* D/space.py:518-531 EliminateInscribedEMS (non-strict containment evaluated on the
 * pre-deletion list: identical EMS delete each other) */
static void eliminate_inscribed(oenv* s) {
  int n = s->n_ems;
  char* del = (char*)calloc((size_t)n + 1, 1);
  for (int i = 0; i < n; i++) {
    const int64_t* a = s->ems + 6 * i;
    for (int j = 0; j < n; j++) {
      if (i == j) continue;
      const int64_t* b = s->ems + 6 * j;
      if (a[0] >= b[0] && a[1] >= b[1] && a[2] >= b[2] && a[3] <= b[3] && a[4] <= b[4] && a[5] <= b[5]) {
        del[i] = 1;
        break;
      }
    }
  }
  int m = 0;
  for (int i = 0; i < n; i++)
    if (!del[i]) {
      if (m != i) memcpy(s->ems + 6 * m, s->ems + 6 * i, 6 * sizeof(int64_t));
      m++;
    }
  s->n_ems = m;
  free(del);
}

/* D/space.py:457-483 GENEMS (event-point upkeep :485-495 omitted, see space_reset) */
static void genems(const pcto_env* h, oenv* s, const int64_t item[6]) {
  int numofemss = s->n_ems;
  char* delflag = (char*)calloc((size_t)numofemss + 1, 1);
  int ndel = 0;
  for (int emsIdx = 0; emsIdx < numofemss; emsIdx++) {
    const int64_t* e = s->ems + 6 * emsIdx;
    int64_t t1 = item[0], u1 = item[1], v1 = item[2], t2 = item[3], u2 = item[4], v2 = item[5];
    if (e[0] > t1) t1 = e[0];
    if (e[1] > u1) u1 = e[1];
    if (e[2] > v1) v1 = e[2];
    if (e[3] < t2) t2 = e[3];
    if (e[4] < u2) u2 = e[4];
    if (e[5] < v2) v2 = e[5];
    if (t1 > t2) t1 = t2;
    if (u1 > u2) u1 = u2;
    if (v1 > v2) v1 = v2;
    if (t1 == t2 || u1 == u2 || v1 == v2) continue;
    int64_t inter[6] = {t1, u1, v1, t2, u2, v2};
    difference(h, s, emsIdx, inter);
    delflag[emsIdx] = 1;
    ndel++;
  }
  if (ndel) {
    int total = s->n_ems, m = 0;
    for (int i = 0; i < total; i++) {
      if (i < numofemss && delflag[i]) continue;
      if (m != i) memcpy(s->ems + 6 * m, s->ems + 6 * i, 6 * sizeof(int64_t));
      m++;
    }
    s->n_ems = m;
  }
  free(delflag);
  eliminate_inscribed(s);
}

/* D/space.py:534-570 EMSPoint: candidate placements in CPython-set iteration order.
 * Returns count; *out (malloc'd) holds [count,6]. */
static int ems_point(const pcto_env* h, const oenv* s, int64_t** out) {
  int orientation = (h->cfg.setting == 2) ? 6 : 2;
  const int* nb = s->next_box;
  int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * 6 * (size_t)(s->n_ems * orientation * 4 + 1));
  int nk = 0;
  for (int ei = 0; ei < s->n_ems; ei++) {
    const int64_t* ems = s->ems + 6 * ei;
    for (int rot = 0; rot < orientation; rot++) {
      int64_t sx, sy, sz;
      switch (rot) {
        case 0: sx = nb[0]; sy = nb[1]; sz = nb[2]; break;
        case 1: sx = nb[1]; sy = nb[0]; sz = nb[2]; if (sx == sy) continue; break;
        case 2: sx = nb[0]; sy = nb[2]; sz = nb[1]; if (sx == sy && sy == sz) continue; break;
        case 3: sx = nb[1]; sy = nb[2]; sz = nb[0]; if (sx == sy && sy == sz) continue; break;
        case 4: sx = nb[2]; sy = nb[0]; sz = nb[1]; if (sx == sy) continue; break;
        default: sx = nb[2]; sy = nb[1]; sz = nb[0]; if (sx == sy) continue; break;
      }
      if (ems[3] - ems[0] >= sx && ems[4] - ems[1] >= sy && ems[5] - ems[2] >= sz) {
        int64_t c[4][6] = {
            {ems[0], ems[1], ems[2], ems[0] + sx, ems[1] + sy, ems[2] + sz},
            {ems[3] - sx, ems[1], ems[2], ems[3], ems[1] + sy, ems[2] + sz},
            {ems[0], ems[4] - sy, ems[2], ems[0] + sx, ems[4], ems[2] + sz},
            {ems[3] - sx, ems[4] - sy, ems[2], ems[3], ems[4], ems[2] + sz}};
        memcpy(keys + 6 * (size_t)nk, c, sizeof c);
        nk += 4;
      }
    }
  }
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nk + 1));
  int cnt = pcto_pyset_order(keys, nk, order);
  int64_t* res = (int64_t*)malloc(sizeof(int64_t) * 6 * (size_t)(cnt + 1));
  for (int i = 0; i < cnt; i++) memcpy(res + 6 * (size_t)i, keys + 6 * (size_t)order[i], 6 * sizeof(int64_t));
  free(order);
  free(keys);
  *out = res;
  return cnt;
}

/* D/PctTools.py:137-158 corners2D on rects (lx,ly,xe,ye); writes (x,y) pairs, returns count */
static int corners2d(const int (*rects)[4], int n, int (*out)[2]) {
  if (n == 0) { out[0][0] = 0; out[0][1] = 0; return 1; }
  /* sorted(..., key=(ye, xe), reverse=True): stable, equal keys keep their original order */
  int* idx = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; i++) {
    int j = i;
    while (j > 0) {
      const int* a = rects[idx[j - 1]];
      const int* b = rects[i];
      int less = (a[3] < b[3]) || (a[3] == b[3] && a[2] < b[2]); /* a's key strictly smaller */
      if (!less) break;
      idx[j] = idx[j - 1];
      j--;
    }
    idx[j] = i;
  }
  int* em = (int*)malloc(sizeof(int) * (size_t)n);
  int m = 0, xRecord = 0;
  for (int i = 0; i < n; i++)
    if (rects[idx[i]][2] > xRecord) { em[m++] = idx[i]; xRecord = rects[idx[i]][2]; }
  int c = 0;
  out[c][0] = 0; out[c][1] = rects[idx[0]][3]; c++;
  for (int q = 1; q < m; q++) { out[c][0] = rects[em[q - 1]][2]; out[c][1] = rects[em[q]][3]; c++; }
  out[c][0] = rects[em[m - 1]][2]; out[c][1] = 0; c++;
  free(idx); free(em);
  return c;
}

/* D/space.py:752-805 CornerPoint.  Returns count; *out holds [count,6] in list order (the
 * empty-bin case returns a plain 2-element list, duplicates and all; otherwise a set). */
static int corner_point(const pcto_env* h, const oenv* s, int64_t** out) {
  int orientation = (h->cfg.setting == 2) ? 6 : 2;
  const int* nb = s->next_box;
  if (s->n_boxes == 0) {
    int64_t* res = (int64_t*)malloc(sizeof(int64_t) * 12);
    int64_t a[12] = {0, 0, 0, nb[0], nb[1], nb[2], 0, 0, 0, nb[1], nb[0], nb[2]};
    memcpy(res, a, sizeof a);
    *out = res;
    return 2;
  }
  int n = s->n_boxes;
  int* T = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  int nT = 0;
  T[nT++] = 0;
  for (int i = 0; i < n; i++) {
    int top = s->boxes[i].z + s->boxes[i].lz, dup = 0;
    for (int j = 0; j < nT; j++) if (T[j] == top) dup = 1;
    if (!dup) T[nT++] = top;
  }
  for (int i = 1; i < nT; i++) { int v = T[i], j = i; while (j > 0 && T[j - 1] > v) { T[j] = T[j - 1]; j--; } T[j] = v; }
  int (*rects)[4] = malloc(sizeof(int[4]) * (size_t)n);
  int (*cik)[2] = malloc(sizeof(int[2]) * (size_t)(n + 2));
  int (*last)[2] = malloc(sizeof(int[2]) * (size_t)(n + 2));
  int nlast = 0;
  int (*CI)[3] = malloc(sizeof(int[3]) * (size_t)(nT * (n + 2)));
  int nCI = 0;
  for (int ti = 0; ti < nT; ti++) {
    int k = T[ti], nr = 0;
    for (int i = 0; i < n; i++)
      if (s->boxes[i].lz + s->boxes[i].z > k) {
        rects[nr][0] = s->boxes[i].lx; rects[nr][1] = s->boxes[i].ly;
        rects[nr][2] = s->boxes[i].lx + s->boxes[i].x; rects[nr][3] = s->boxes[i].ly + s->boxes[i].y;
        nr++;
      }
    int nc = corners2d(rects, nr, cik);
    for (int c = 0; c < nc; c++) {
      int seen = 0;
      for (int q = 0; q < nlast; q++) if (last[q][0] == cik[c][0] && last[q][1] == cik[c][1]) seen = 1;
      if (!seen) { CI[nCI][0] = cik[c][0]; CI[nCI][1] = cik[c][1]; CI[nCI][2] = k; nCI++; }
    }
    memcpy(last, cik, sizeof(int[2]) * (size_t)nc);
    nlast = nc;
  }
  int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * 6 * (size_t)(nCI * orientation + 1));
  int nk = 0;
  for (int c = 0; c < nCI; c++)
    for (int rot = 0; rot < orientation; rot++) {
      int64_t sx, sy, sz;
      switch (rot) {
        case 0: sx = nb[0]; sy = nb[1]; sz = nb[2]; break;
        case 1: sx = nb[1]; sy = nb[0]; sz = nb[2]; if (sx == sy) continue; break;
        case 2: sx = nb[0]; sy = nb[2]; sz = nb[1]; if (sx == sy && sy == sz) continue; break;
        case 3: sx = nb[1]; sy = nb[2]; sz = nb[0]; if (sx == sy && sy == sz) continue; break;
        case 4: sx = nb[2]; sy = nb[0]; sz = nb[1]; if (sx == sy) continue; break;
        default: sx = nb[2]; sy = nb[1]; sz = nb[0]; if (sx == sy) continue; break;
      }
      if (CI[c][0] + sx <= h->cfg.container[0] && CI[c][1] + sy <= h->cfg.container[1] &&
          CI[c][2] + sz <= h->cfg.container[2]) {
        int64_t* kk = keys + 6 * (size_t)nk++;
        kk[0] = CI[c][0]; kk[1] = CI[c][1]; kk[2] = CI[c][2];
        kk[3] = CI[c][0] + sx; kk[4] = CI[c][1] + sy; kk[5] = CI[c][2] + sz;
      }
    }
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nk + 1));
  int cnt = pcto_pyset_order(keys, nk, order);
  int64_t* res = (int64_t*)malloc(sizeof(int64_t) * 6 * (size_t)(cnt + 1));
  for (int i = 0; i < cnt; i++) memcpy(res + 6 * (size_t)i, keys + 6 * (size_t)order[i], 6 * sizeof(int64_t));
  free(order); free(keys); free(T); free(rects); free(cik); free(last); free(CI);
  *out = res;
  return cnt;
}

/* D/PctTools.py:114-136 extreme2D on the rectangles above one level.  in[i] = {lx, ly, lxe, lye};
 * out gets the extreme points in list order, returns their number (<= 2n).  n == 0 is handled
 * by the caller (the reference returns the 3-tuple (0,0,0) there). */
static int extreme2d(int (*in)[4], int n, int (*out)[2]) {
  int* idx = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  for (int i = 0; i < n; i++) { /* sorted(key=(ly, lxe)): stable insertion sort, ascending */
    int j = i;
    while (j > 0) {
      const int* a = in[idx[j - 1]];
      const int* b = in[i];
      int greater = (a[1] > b[1]) || (a[1] == b[1] && a[2] > b[2]);
      if (!greater) break;
      idx[j] = idx[j - 1];
      j--;
    }
    idx[j] = i;
  }
  static const int demo[2][4] = {{-1, 0, 0, 10}, {0, -1, 10, 0}}; /* PctTools.py:118, the literal 10 included */
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    const int* nw = in[idx[i]];
    int max0 = -10, max2 = -10, have0 = 0, have2 = 0, first = -1;
    int e0[2] = {0, 0}, e2[2] = {0, 0};
    for (int q = -2; q < i; q++) {
      const int* bx = q < 0 ? demo[q + 2] : in[idx[q]];
      int px = bx[2], py = bx[3]; /* projectedX / projectedY */
      if (nw[0] >= bx[2] && nw[3] < bx[3] && px > max0) { /* IsProjectionValid2D(newItem, box, 0) */
        e0[0] = px; e0[1] = nw[3]; max0 = px;
        if (!have0) { have0 = 1; if (first < 0) first = 0; }
      }
      if (nw[1] >= bx[3] && nw[2] < bx[2] && py > max2) { /* direction 2 */
        e2[0] = nw[2]; e2[1] = py; max2 = py;
        if (!have2) { have2 = 1; if (first < 0) first = 2; }
      }
    }
    /* deleteEps2D(newItem, alleps) */
    int w = 0;
    for (int c = 0; c < cnt; c++) {
      int inside = out[c][0] >= nw[0] && out[c][0] < nw[2] && out[c][1] >= nw[1] && out[c][1] < nw[3];
      if (!inside) { out[w][0] = out[c][0]; out[w][1] = out[c][1]; w++; }
    }
    cnt = w;
    /* alleps.extend(list(set(newEps.values()))): dict order = first-insertion order of the keys,
     * then the iteration order of a fresh 8-slot set holding one or two 2-tuples */
    int nv = have0 + have2;
    if (nv == 1) {
      const int* v = have0 ? e0 : e2;
      out[cnt][0] = v[0]; out[cnt][1] = v[1]; cnt++;
    } else if (nv == 2) {
      const int* a = first == 0 ? e0 : e2;
      const int* b = first == 0 ? e2 : e0;
      if (a[0] == b[0] && a[1] == b[1]) {
        out[cnt][0] = a[0]; out[cnt][1] = a[1]; cnt++;
      } else {
        int64_t ta[2] = {a[0], a[1]}, tb[2] = {b[0], b[1]};
        uint64_t ha = py_tuplehash(ta, 2), hb = py_tuplehash(tb, 2);
        size_t ia = (size_t)ha & 7, ib = (size_t)hb & 7;
        uint64_t perturb = hb;
        while (ib == ia) { /* mask 7: no linear probes (i + 9 > mask) */
          perturb >>= 5;
          ib = (ib * 5 + 1 + (size_t)perturb) & 7;
        }
        const int* f = ia < ib ? a : b;
        const int* g = ia < ib ? b : a;
        out[cnt][0] = f[0]; out[cnt][1] = f[1]; cnt++;
        out[cnt][0] = g[0]; out[cnt][1] = g[1]; cnt++;
      }
    }
  }
  free(idx);
  return cnt;
}

/* rotation `rot` of the item (D/space.py:626-649 and the same block in every scheme); returns 0
 * for a skipped rotation */
static int rot_size(const int* nb, int rot, int64_t* sx, int64_t* sy, int64_t* sz) {
  switch (rot) {
    case 0: *sx = nb[0]; *sy = nb[1]; *sz = nb[2]; return 1;
    case 1: *sx = nb[1]; *sy = nb[0]; *sz = nb[2]; return *sx != *sy;
    case 2: *sx = nb[0]; *sy = nb[2]; *sz = nb[1]; return !(*sx == *sy && *sy == *sz);
    case 3: *sx = nb[1]; *sy = nb[2]; *sz = nb[0]; return !(*sx == *sy && *sy == *sz);
    case 4: *sx = nb[2]; *sy = nb[0]; *sz = nb[1]; return *sx != *sy;
    default: *sx = nb[2]; *sy = nb[1]; *sz = nb[0]; return *sx != *sy;
  }
}

/* D/space.py:696-750 ExtremePoint2D */
static int extreme_point(const pcto_env* h, const oenv* s, int64_t** out) {
  int orientation = (h->cfg.setting == 2) ? 6 : 2;
  const int* nb = s->next_box;
  if (s->n_boxes == 0) { /* :700-701 a plain two-element list */
    int64_t* res = (int64_t*)malloc(sizeof(int64_t) * 12);
    int64_t a[12] = {0, 0, 0, nb[0], nb[1], nb[2], 0, 0, 0, nb[1], nb[0], nb[2]};
    memcpy(res, a, sizeof a);
    *out = res;
    return 2;
  }
  int n = s->n_boxes;
  int* T = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  int nT = 0;
  T[nT++] = 0;
  for (int i = 0; i < n; i++) {
    int top = s->boxes[i].z + s->boxes[i].lz, dup = 0;
    for (int j = 0; j < nT; j++) if (T[j] == top) dup = 1;
    if (!dup) T[nT++] = top;
  }
  for (int i = 1; i < nT; i++) { int v = T[i], j = i; while (j > 0 && T[j - 1] > v) { T[j] = T[j - 1]; j--; } T[j] = v; }
  int (*rects)[4] = malloc(sizeof(int[4]) * (size_t)n);
  int (*cik)[2] = malloc(sizeof(int[2]) * (size_t)(2 * n + 2));
  int (*last)[2] = malloc(sizeof(int[2]) * (size_t)(2 * n + 2));
  int nlast = 0;
  int (*CI)[3] = malloc(sizeof(int[3]) * (size_t)(nT * (2 * n + 2)));
  int nCI = 0;
  for (int ti = 0; ti < nT; ti++) {
    int k = T[ti], nr = 0;
    for (int i = 0; i < n; i++)
      if (s->boxes[i].lz + s->boxes[i].z > k) {
        rects[nr][0] = s->boxes[i].lx; rects[nr][1] = s->boxes[i].ly;
        rects[nr][2] = s->boxes[i].lx + s->boxes[i].x; rects[nr][3] = s->boxes[i].ly + s->boxes[i].y;
        nr++;
      }
    if (nr == 0) { /* extreme2D([]) == [(0,0,0)]: a 3-tuple, never equal to a 2-tuple of lastCik */
      CI[nCI][0] = 0; CI[nCI][1] = 0; CI[nCI][2] = k; nCI++;
      nlast = 0;
      continue;
    }
    int nc = extreme2d(rects, nr, cik);
    for (int c = 0; c < nc; c++) {
      int seen = 0;
      for (int q = 0; q < nlast; q++) if (last[q][0] == cik[c][0] && last[q][1] == cik[c][1]) seen = 1;
      if (!seen) { CI[nCI][0] = cik[c][0]; CI[nCI][1] = cik[c][1]; CI[nCI][2] = k; nCI++; }
    }
    memcpy(last, cik, sizeof(int[2]) * (size_t)nc);
    nlast = nc;
  }
  int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * 6 * (size_t)(nCI * orientation + 1));
  int nk = 0;
  for (int c = 0; c < nCI; c++)
    for (int rot = 0; rot < orientation; rot++) {
      int64_t sx, sy, sz;
      if (!rot_size(nb, rot, &sx, &sy, &sz)) continue;
      if (CI[c][0] + sx <= h->cfg.container[0] && CI[c][1] + sy <= h->cfg.container[1] &&
          CI[c][2] + sz <= h->cfg.container[2]) {
        int64_t* kk = keys + 6 * (size_t)nk++;
        kk[0] = CI[c][0]; kk[1] = CI[c][1]; kk[2] = CI[c][2];
        kk[3] = CI[c][0] + sx; kk[4] = CI[c][1] + sy; kk[5] = CI[c][2] + sz;
      }
    }
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nk + 1));
  int cnt = pcto_pyset_order(keys, nk, order);
  int64_t* res = (int64_t*)malloc(sizeof(int64_t) * 6 * (size_t)(cnt + 1));
  for (int i = 0; i < cnt; i++) memcpy(res + 6 * (size_t)i, keys + 6 * (size_t)order[i], 6 * sizeof(int64_t));
  free(order); free(keys); free(T); free(rects); free(cik); free(last); free(CI);
  *out = res;
  return cnt;
}

/* D/space.py:613-693 EventPoint.  bin3D.py:171 runs GENEMS only under LNES == 'EMS', so under
 * 'EV' neither the EMS list nor ZMAP ever changes after Space.reset (space.py:290-314): one
 * level k = 0 with x_up = [0], y_left = [0], x_bottom = [W], y_right = [Ly] and the single EMS
 * [0,0,0,W,Ly,H].  posVec = the four bin corners per rotation (a set: may hold negative
 * coordinates), kept if the footprint lies inside the EMS (:677-688). */
static int event_point(const pcto_env* h, const oenv* s, int64_t** out) {
  int orientation = (h->cfg.setting == 2) ? 6 : 2;
  const int* nb = s->next_box;
  int64_t W = h->cfg.container[0], L = h->cfg.container[1];
  int64_t keys[24 * 6];
  int nk = 0;
  for (int rot = 0; rot < orientation; rot++) {
    int64_t sx, sy, sz;
    if (!rot_size(nb, rot, &sx, &sy, &sz)) continue;
    int64_t cand[4][2] = {{0, 0}, {0, L - sy}, {W - sx, 0}, {W - sx, L - sy}}; /* :654-674 add order */
    for (int c = 0; c < 4; c++) {
      int64_t* kk = keys + 6 * nk++;
      kk[0] = cand[c][0]; kk[1] = cand[c][1]; kk[2] = 0;
      kk[3] = cand[c][0] + sx; kk[4] = cand[c][1] + sy; kk[5] = sz;
    }
  }
  int32_t order[25];
  int cnt = pcto_pyset_order(keys, nk, order);
  int64_t* res = (int64_t*)malloc(sizeof(int64_t) * 6 * (size_t)(cnt + 1));
  int m = 0;
  for (int i = 0; i < cnt; i++) {
    const int64_t* kk = keys + 6 * (size_t)order[i];
    if (kk[0] >= 0 && kk[1] >= 0 && kk[3] <= W && kk[4] <= L) memcpy(res + 6 * (size_t)m++, kk, 6 * sizeof(int64_t));
  }
  *out = res;
  return m;
}

/* D/space.py:573-610 FullCoord: every (lx, ly) at its own cell height, every rotation */
static int full_coord(const pcto_env* h, const oenv* s, int64_t** out) {
  int orientation = (h->cfg.setting == 2) ? 6 : 2;
  const int* nb = s->next_box;
  int W = h->cfg.container[0], L = h->cfg.container[1], H = h->cfg.container[2], A = h->A;
  int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * 6 * (size_t)(orientation * W * L + 1));
  int nk = 0;
  for (int rot = 0; rot < orientation; rot++) {
    int64_t sx, sy, sz;
    switch (rot) {
      case 0: sx = nb[0]; sy = nb[1]; sz = nb[2]; break;
      case 1: sx = nb[1]; sy = nb[0]; sz = nb[2]; if (sx == sy) continue; break;
      case 2: sx = nb[0]; sy = nb[2]; sz = nb[1]; if (sx == sy && sy == sz) continue; break;
      case 3: sx = nb[1]; sy = nb[2]; sz = nb[0]; if (sx == sy && sy == sz) continue; break;
      case 4: sx = nb[2]; sy = nb[0]; sz = nb[1]; if (sx == sy) continue; break;
      default: sx = nb[2]; sy = nb[1]; sz = nb[0]; if (sx == sy) continue; break;
    }
    for (int lx = 0; lx < W; lx++)
      for (int ly = 0; ly < L; ly++) {
        int lz = s->plain[lx * A + ly];
        if (lx + sx <= W && ly + sy <= L && lz + sz <= H) {
          int64_t* kk = keys + 6 * (size_t)nk++;
          kk[0] = lx; kk[1] = ly; kk[2] = lz; kk[3] = lx + sx; kk[4] = ly + sy; kk[5] = lz + sz;
        }
      }
  }
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nk + 1));
  int cnt = pcto_pyset_order(keys, nk, order);
  int64_t* res = (int64_t*)malloc(sizeof(int64_t) * 6 * (size_t)(cnt + 1));
  for (int i = 0; i < cnt; i++) memcpy(res + 6 * (size_t)i, keys + 6 * (size_t)order[i], 6 * sizeof(int64_t));
  free(order); free(keys);
  *out = res;
  return cnt;
}

/* D/bin3D.py:100-136 get_possible_position (LNES='EMS' / 'CP' / 'FC', shuffle=False): writes the L
 * leaf rows into leaf[L*9] */
static void shuffle_rows_i64(const pcto_env* h, int e, uint64_t oc, int64_t* pos, int n) {
  /* include/pct_env.h pct_shuffle_priority: ascending (priority, index) */
  if (n < 2) return;
  uint32_t* pr = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
  int* idx = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; i++) { pr[i] = pct_shuffle_priority(h->shuffle_seed, (uint64_t)(h->cfg.env_id_base + e), oc, (uint32_t)i); idx[i] = i; }
  for (int i = 1; i < n; i++) { /* stable insertion sort by priority */
    int v = idx[i], j = i;
    while (j > 0 && pr[idx[j - 1]] > pr[v]) { idx[j] = idx[j - 1]; j--; }
    idx[j] = v;
  }
  int64_t* tmp = (int64_t*)malloc(sizeof(int64_t) * 6 * (size_t)n);
  for (int i = 0; i < n; i++) memcpy(tmp + 6 * (size_t)i, pos + 6 * (size_t)idx[i], 6 * sizeof(int64_t));
  memcpy(pos, tmp, sizeof(int64_t) * 6 * (size_t)n);
  free(tmp); free(idx); free(pr);
}

static void get_possible_position(const pcto_env* h, int e, oenv* s, double* leaf) {
  memset(leaf, 0, sizeof(double) * 9 * h->L);
  int64_t* pos = NULL;
  int n = h->cfg.lnes == PCT_LNES_CP ? corner_point(h, s, &pos)
          : h->cfg.lnes == PCT_LNES_FC ? full_coord(h, s, &pos)
          : h->cfg.lnes == PCT_LNES_EP ? extreme_point(h, s, &pos)
          : h->cfg.lnes == PCT_LNES_EV ? event_point(h, s, &pos) : ems_point(h, s, &pos);
  if (h->cfg.shuffle && h->rng_numpy) {
    /* np.random.shuffle(allPostion) on the [n,6] array (legacy RandomState.shuffle: Fisher-Yates from the back,
     * j = random_interval(i), rows swapped) */
    for (int i = n - 1; i >= 1; i--) {
      int j = (int)npmt_interval(s->mt, &s->mt_pos, (uint32_t)i);
      if (j == i) continue;
      for (int c = 0; c < 6; c++) { int64_t t_ = pos[6 * i + c]; pos[6 * i + c] = pos[6 * j + c]; pos[6 * j + c] = t_; }
    }
  } else if (h->cfg.shuffle) shuffle_rows_i64(h, e, s->oc, pos, n); /* D/bin3D.py:114-115 */
  s->oc++;
  int idx = 0;
  for (int i = 0; i < n; i++) {
    const int64_t* p = pos + 6 * i;
    int x = (int)(p[3] - p[0]), y = (int)(p[4] - p[1]), z = (int)(p[5] - p[2]);
    if (drop_box_virtual(h, s, x, y, z, (int)p[0], (int)p[1])) {
      double* r = leaf + 9 * idx;
      r[0] = (double)p[0]; r[1] = (double)p[1]; r[2] = (double)p[2];
      r[3] = (double)p[3]; r[4] = (double)p[4]; r[5] = (double)h->cfg.container[2];
      r[6] = 0; r[7] = 0; r[8] = 1;
      idx++;
    }
    if (idx >= h->L) break;
  }
  free(pos);
}

/* D/bin3D.py:70-93 cur_observation */
static void cur_observation(const pcto_env* h, int e, oenv* s, double* obs) {
  /* gen_next_box -> box_creator.preview(1)[0] (binCreator.py:15-18) */
  if (s->queue_len < 1) {
    draw_item(h, e, s, s->queue_item);
    s->queue_len = 1;
  }
  s->next_box[0] = s->queue_item[0]; s->next_box[1] = s->queue_item[1]; s->next_box[2] = s->queue_item[2];
  s->next_den = pcto_next_density(h, e, s->oc, s->traj, s->cursor - 1); /* :75-84 */
  if (h->rng_numpy && h->cfg.setting == 3 && h->source != PCT_ITEMS_DATASET) { /* np.random.random(), redrawn while 0 (:82-84) */
    do { s->next_den = npmt_double(s->mt, &s->mt_pos); } while (s->next_den == 0);
  }
  memcpy(obs, s->box_vec, sizeof(double) * 9 * h->I);
  get_possible_position(h, e, s, obs + 9 * h->I);
  int a = s->next_box[0], b = s->next_box[1], c = s->next_box[2], tmp;
  if (a > b) { tmp = a; a = b; b = tmp; }
  if (b > c) { tmp = b; b = c; c = tmp; }
  if (a > b) { tmp = a; a = b; b = tmp; }
  double* r = obs + 9 * (h->I + h->L);
  r[0] = s->next_den; r[1] = 0; r[2] = 0; r[3] = a; r[4] = b; r[5] = c; r[6] = 0; r[7] = 0; r[8] = 1;
}

/* D/bin3D.py:61-67 reset */
static void env_reset(const pcto_env* h, int e, oenv* s, double* obs) {
  s->queue_len = 0;            /* box_creator.reset() */
  if (h->source == PCT_ITEMS_DATASET) { /* LoadBoxCreator.reset: index += 1 (:51-62) */
    s->traj++;
    s->cursor = 0;
    if (s->traj >= h->ds_ntraj) h->flags[e] |= PCT_FLAG_DATASET_EXHAUSTED; /* IndexError :58 */
  }
  space_reset(h, s);           /* space.reset() */
  draw_item(h, e, s, s->queue_item); /* box_creator.generate_box_size() */
  s->queue_len = 1;
  cur_observation(h, e, s, obs);
}

/* D/space.py:334-339 get_ratio */
static double get_ratio(const pcto_env* h, const oenv* s) {
  double vo = 0.0;
  for (int i = 0; i < s->n_boxes; i++) vo = vo + (double)((int64_t)s->boxes[i].x * s->boxes[i].y * s->boxes[i].z);
  double mx = (double)((int64_t)h->cfg.container[0] * h->cfg.container[1] * h->cfg.container[2]);
  return vo / mx;
}

/* D/bin3D.py:139-149 LeafNode2Action + :151-188 step.  `act` has `len` entries. */
static void env_step(const pcto_env* h, int e, oenv* s, const double* act, int len, double* obs, double* reward,
                     uint8_t* done, int32_t* counter, double* ratio, uint32_t* flags) {
  int flag, lx, ly, nb[3];
  s->t++;
  if (len != 3) {
    double sum = 0;
    for (int i = 0; i < 6; i++) sum += act[i];
    if (sum == 0) {
      flag = 0; lx = 0; ly = 0;
      nb[0] = s->next_box[0]; nb[1] = s->next_box[1]; nb[2] = s->next_box[2];
    } else {
      int x = (int)(act[3] - act[0]);
      int y = (int)(act[4] - act[1]);
      int z[3] = {s->next_box[0], s->next_box[1], s->next_box[2]};
      int nz = 3, found = 0;
      for (int i = 0; i < nz; i++)
        if (z[i] == x) { for (int j = i; j < nz - 1; j++) z[j] = z[j + 1]; nz--; found = 1; break; }
      if (found) {
        found = 0;
        for (int i = 0; i < nz; i++)
          if (z[i] == y) { for (int j = i; j < nz - 1; j++) z[j] = z[j + 1]; nz--; found = 1; break; }
      }
      if (!found) { /* ValueError: list.remove(x): x not in list */
        *flags |= PCT_FLAG_BAD_ACTION;
        *reward = 0.0; *done = 1; *counter = s->n_boxes; *ratio = get_ratio(h, s);
        cur_observation(h, e, s, obs);
        return;
      }
      flag = 0; lx = (int)act[0]; ly = (int)act[1];
      nb[0] = x; nb[1] = y; nb[2] = z[0];
    }
  } else {
    flag = (int)act[0]; lx = (int)act[1]; ly = (int)act[2];
    nb[0] = s->next_box[0]; nb[1] = s->next_box[1]; nb[2] = s->next_box[2];
  }
  int ok = drop_box(h, s, nb, lx, ly, flag, s->next_den, flags);
  if (!ok) {
    *reward = 0.0;
    *done = 1;
    *counter = s->n_boxes;
    *ratio = get_ratio(h, s);
    cur_observation(h, e, s, obs);
    return;
  }
  const obox* pb = &s->boxes[s->n_boxes - 1];
  int64_t loc[6] = {pb->lx, pb->ly, pb->lz, pb->lx + pb->x, pb->ly + pb->y, pb->lz + pb->z};
  if (h->cfg.lnes == PCT_LNES_EMS) genems(h, s, loc); /* D/bin3D.py:172-175 */
  /* get_box_ratio :57-59 on self.next_box (the unrotated item) */
  double box_ratio = (double)((int64_t)s->next_box[0] * s->next_box[1] * s->next_box[2]) /
                     (double)((int64_t)h->cfg.container[0] * h->cfg.container[1] * h->cfg.container[2]);
  s->queue_len = 0;                  /* box_creator.drop_box() */
  draw_item(h, e, s, s->queue_item); /* generate_box_size() */
  s->queue_len = 1;
  *reward = box_ratio * 10;
  *done = 0;
  *counter = s->n_boxes;
  *ratio = 0.0; /* not part of a non-terminal info dict (:186-187) */
  cur_observation(h, e, s, obs);
}

/* ===================================================================================== */
/* batched handle                                                                         */
/* ===================================================================================== */
int pcto_create(const pct_config* cfg, pcto_env** out) {
  if (!cfg || !out) return fail(PCT_ERR_INVALID_ARG, "null argument");
  if (cfg->struct_size != (int32_t)sizeof(pct_config)) return fail(PCT_ERR_INVALID_ARG, "pct_config size mismatch");
  if (cfg->env_kind != PCT_ENV_DISCRETE && cfg->env_kind != PCT_ENV_CONTINUOUS)
    return fail(PCT_ERR_UNSUPPORTED, "oracle: unknown env kind");
  if (cfg->setting < 1 || cfg->setting > 3) return fail(PCT_ERR_UNSUPPORTED, "oracle: setting must be 1, 2 or 3");
  if (cfg->lnes < PCT_LNES_EMS || cfg->lnes > PCT_LNES_FC) return fail(PCT_ERR_INVALID_ARG, "oracle: unknown LNES");
  if (cfg->lnes != PCT_LNES_EMS && cfg->env_kind != PCT_ENV_DISCRETE)
    return fail(PCT_ERR_UNSUPPORTED, "EV / EP / CP / FC are reachable only in the discrete env (C/bin3D.py:53)");
  if (cfg->num_envs < 1 || cfg->internal_node_holder < 1 || cfg->leaf_node_holder < 1)
    return fail(PCT_ERR_INVALID_ARG, "bad sizes");
  pcto_env* h = (pcto_env*)calloc(1, sizeof *h);
  h->cfg = *cfg;
  h->N = cfg->num_envs;
  h->A = cfg->container[0] > cfg->container[1] ? cfg->container[0] : cfg->container[1];
  h->I = cfg->internal_node_holder;
  h->L = cfg->leaf_node_holder;
  h->row_len = (h->I + h->L + 1) * 9;
  h->envs = (oenv*)calloc((size_t)h->N, sizeof(oenv));
  if (cfg->env_kind == PCT_ENV_CONTINUOUS) {
    if (pctc_alloc(h)) return fail(PCT_ERR_INVALID_ARG, "continuous alloc failed");
  }
  for (int e = 0; e < h->N && cfg->env_kind == PCT_ENV_DISCRETE; e++) {
    oenv* s = &h->envs[e];
    s->plain = (int*)calloc((size_t)h->A * h->A, sizeof(int));
    s->box_vec = (double*)calloc((size_t)h->I * 9, sizeof(double));
    s->boxes = (obox*)calloc((size_t)h->I + 1, sizeof(obox));
    s->cap_ems = 64;
    s->ems = (int64_t*)calloc((size_t)s->cap_ems * 6, sizeof(int64_t));
    if (cfg->setting != 2) s->stab = stab_create(h->I, 0.0);
  }
  h->obs = (double*)calloc((size_t)h->N * h->row_len, sizeof(double));
  h->reward = (double*)calloc((size_t)h->N, sizeof(double));
  h->done = (uint8_t*)calloc((size_t)h->N, 1);
  h->counter = (int32_t*)calloc((size_t)h->N, sizeof(int32_t));
  h->ratio = (double*)calloc((size_t)h->N, sizeof(double));
  h->flags = (uint32_t*)calloc((size_t)h->N, sizeof(uint32_t));
  *out = h;
  return PCT_OK;
}

int pcto_destroy(pcto_env* h) {
  if (!h) return PCT_OK;
  for (int e = 0; e < h->N; e++) {
    free(h->envs[e].plain); free(h->envs[e].box_vec); free(h->envs[e].boxes); free(h->envs[e].ems);
    stab_free(h->envs[e].stab);
  }
  if (h->cenvs) pctc_free(h);
  free(h->envs); free(h->obs); free(h->reward); free(h->done); free(h->counter); free(h->ratio);
  free(h->flags); free(h->item_set); free(h->stream); free(h->ds_len); free(h->den_stream); free(h->ds_den);
  free(h);
  return PCT_OK;
}

int pcto_set_item_set(pcto_env* h, const int32_t* item_set, int32_t n) {
  if (!h || !item_set || n < 1) return fail(PCT_ERR_INVALID_ARG, "bad item set");
  free(h->item_set);
  h->item_set = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)n);
  memcpy(h->item_set, item_set, sizeof(int32_t) * 3 * (size_t)n);
  h->n_items = n;
  int mn = item_set[0];
  for (int i = 0; i < 3 * n; i++) if (item_set[i] < mn) mn = item_set[i];
  h->low_bound = mn; /* bin3D.py:23 size_minimum */
  return PCT_OK;
}
int pcto_set_sample_bounds(pcto_env* h, int32_t left, int32_t right) {
  if (!h || left < 1 || right < left) return fail(PCT_ERR_INVALID_ARG, "bad bounds");
  h->low_bound = left; /* C/bin3D.py:25-27 size_minimum = sample_left_bound */
  h->sample_left = left;
  h->sample_right = right;
  if (!h->item_set) { /* placeholder so that ready() passes; the sampler draws from the bounds */
    h->item_set = (int32_t*)calloc(3, sizeof(int32_t));
    h->n_items = 1;
  }
  return PCT_OK;
}
int pcto_set_item_stream(pcto_env* h, const int32_t* items, int64_t T) {
  if (!h || !items || T < 1) return fail(PCT_ERR_INVALID_ARG, "bad stream");
  free(h->stream);
  size_t n = (size_t)h->N * (size_t)T * 3;
  h->stream = (int32_t*)malloc(sizeof(int32_t) * n);
  memcpy(h->stream, items, sizeof(int32_t) * n);
  h->T = T;
  h->source = PCT_ITEMS_STREAM;
  return PCT_OK;
}
int pcto_set_item_dataset(pcto_env* h, const int32_t* items, const int32_t* lengths, int32_t n_traj, int32_t max_len) {
  if (!h || !items || !lengths || n_traj < 2 || max_len < 1) return fail(PCT_ERR_INVALID_ARG, "bad dataset");
  free(h->stream);
  free(h->ds_len);
  size_t n = (size_t)n_traj * (size_t)max_len * 3;
  h->stream = (int32_t*)malloc(sizeof(int32_t) * n);
  memcpy(h->stream, items, sizeof(int32_t) * n);
  h->ds_len = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_traj);
  memcpy(h->ds_len, lengths, sizeof(int32_t) * (size_t)n_traj);
  h->ds_ntraj = n_traj;
  h->ds_maxlen = max_len;
  h->source = PCT_ITEMS_DATASET;
  return PCT_OK;
}
int pcto_set_density_stream(pcto_env* h, const double* den, int64_t T) {
  if (!h || !den || T < 1) return fail(PCT_ERR_INVALID_ARG, "bad density stream");
  free(h->den_stream);
  size_t n = (size_t)h->N * (size_t)T;
  h->den_stream = (double*)malloc(sizeof(double) * n);
  memcpy(h->den_stream, den, sizeof(double) * n);
  h->den_T = T;
  return PCT_OK;
}
int pcto_set_dataset_density(pcto_env* h, const double* den) {
  if (!h || !den || h->source != PCT_ITEMS_DATASET) return fail(PCT_ERR_STATE, "set the dataset items first");
  free(h->ds_den);
  size_t n = (size_t)h->ds_ntraj * (size_t)h->ds_maxlen;
  h->ds_den = (double*)malloc(sizeof(double) * n);
  memcpy(h->ds_den, den, sizeof(double) * n);
  return PCT_OK;
}
int pcto_set_shuffle_seed(pcto_env* h, uint64_t seed) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  h->shuffle_seed = seed;
  return PCT_OK;
}
/* strict NumPy-stream mode: env e consumes the MT19937 stream np.random.seed(seed + env_id_base + e) starts
 * (every worker of ShmemVecEnv(fork) seeds its own process-global RandomState: envs.py:49, bin3D.py:47-54);
 * items come from the item set through np.random.randint.  Discrete env. */
void pctc_set_numpy_rng(struct pcto_env* h, uint32_t seed);
int pcto_set_numpy_rng(pcto_env* h, uint32_t seed) {
  if (!h || !h->item_set) return fail(PCT_ERR_STATE, "set the item set / sample bounds first");
  h->rng_numpy = 1;
  h->source = PCT_ITEMS_SAMPLER;
  if (h->cfg.env_kind != PCT_ENV_DISCRETE) {
    if (h->sample_right <= 0) return fail(PCT_ERR_UNSUPPORTED, "NumPy-stream mode, continuous env: sampling from U(a,b) only");
    pctc_set_numpy_rng(h, seed);
    return PCT_OK;
  }
  for (int e = 0; e < h->N; e++) npmt_seed(h->envs[e].mt, &h->envs[e].mt_pos, seed + (uint32_t)h->cfg.env_id_base + (uint32_t)e);
  return PCT_OK;
}
/* continuous NumPy-stream mode: len(item_set) of the RandomBoxCreator whose randint draws are consumed but unused */
int pcto_set_numpy_item_count(pcto_env* h, int32_t n) {
  if (!h || n < 1) return fail(PCT_ERR_INVALID_ARG, "bad count");
  if (h->cfg.env_kind == PCT_ENV_DISCRETE) return fail(PCT_ERR_UNSUPPORTED, "continuous env only");
  h->n_items = n;
  return PCT_OK;
}

int pcto_set_sampler(pcto_env* h, uint64_t seed) {
  if (!h || !h->item_set) return fail(PCT_ERR_STATE, "set the item set / sample bounds first");
  h->seed = seed;
  h->source = PCT_ITEMS_SAMPLER;
  return PCT_OK;
}

double* pcto_obs(pcto_env* h) { return h->obs; }
double* pcto_reward(pcto_env* h) { return h->reward; }
uint8_t* pcto_done(pcto_env* h) { return h->done; }
int32_t* pcto_info_counter(pcto_env* h) { return h->counter; }
double* pcto_info_ratio(pcto_env* h) { return h->ratio; }
uint32_t* pcto_error_flags(pcto_env* h) { return h->flags; }
void pcto_set_ill_near(int on) { stab_set_ill_near(on); } /* analysis mode of the notice, see pct_oracle_stab.c */
/* process-wide: which solver stands behind np.linalg.lstsq in the stability check (pct_oracle_stab.c / pct_oracle_gelsd.c) */
void pcto_set_lstsq_mode(int mode) { stab_set_lstsq_mode(mode); }
int pcto_get_lstsq_mode(void) { return stab_get_lstsq_mode(); }
/* out[e] = 1 iff env e has taken an ill-conditioned least-squares split so far (the product's PCT_FLAG_ILL_CONDITIONED) */
int pcto_ill_conditioned(pcto_env* h, uint8_t* out) {
  if (!h || !out) return fail(PCT_ERR_INVALID_ARG, "null argument");
  for (int e = 0; e < h->N; e++) {
    const struct stab* st = h->cfg.env_kind == PCT_ENV_CONTINUOUS ? pctc_stab(h, e) : h->envs[e].stab;
    out[e] = (st && stab_ill_conditioned(st)) ? 1 : 0;
  }
  return PCT_OK;
}

/* out[e] = 1 iff a solve of env e's COMMIT walks raised the notice (the product's PCT_FLAG_ILL_COMMIT) */
int pcto_ill_commit(pcto_env* h, uint8_t* out) {
  if (!h || !out) return fail(PCT_ERR_INVALID_ARG, "null argument");
  for (int e = 0; e < h->N; e++) {
    const struct stab* st = h->cfg.env_kind == PCT_ENV_CONTINUOUS ? pctc_stab(h, e) : h->envs[e].stab;
    out[e] = (st && stab_ill_commit(st)) ? 1 : 0;
  }
  return PCT_OK;
}

static int ready(const pcto_env* h) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  if (!h->item_set) return fail(PCT_ERR_STATE, "item set not configured");
  if (h->source == PCT_ITEMS_NONE) return fail(PCT_ERR_STATE, "item source not configured");
  return PCT_OK;
}

static void any_reset(pcto_env* h, int e, double* obs) {
  if (h->cfg.env_kind == PCT_ENV_CONTINUOUS) pctc_reset(h, e, obs);
  else env_reset(h, e, &h->envs[e], obs);
}
static void any_step(pcto_env* h, int e, const double* act, int len, double* obs) {
  if (h->cfg.env_kind == PCT_ENV_CONTINUOUS)
    pctc_step(h, e, act, len, obs, &h->reward[e], &h->done[e], &h->counter[e], &h->ratio[e], &h->flags[e]);
  else
    env_step(h, e, &h->envs[e], act, len, obs, &h->reward[e], &h->done[e], &h->counter[e], &h->ratio[e], &h->flags[e]);
}

int pcto_reset(pcto_env* h, const int32_t* env_ids, int32_t n) {
  int rc = ready(h);
  if (rc) return rc;
  int cnt = env_ids ? n : h->N;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int k = 0; k < cnt; k++) {
    int e = env_ids ? env_ids[k] : k;
    if (e < 0 || e >= h->N) continue;
    any_reset(h, e, h->obs + (size_t)e * h->row_len);
  }
  return PCT_OK;
}

int pcto_step_rows(pcto_env* h, const double* rows, int32_t row_len, int32_t auto_reset) {
  int rc = ready(h);
  if (rc) return rc;
  if (row_len != 9 && row_len != 6 && row_len != 3) return fail(PCT_ERR_INVALID_ARG, "row_len must be 9, 6 or 3");
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 8)
  for (int e = 0; e < h->N; e++) {
    double* obs = h->obs + (size_t)e * h->row_len;
    any_step(h, e, rows + (size_t)e * row_len, row_len, obs);
    if (h->done[e] && auto_reset) any_reset(h, e, obs);
  }
  return PCT_OK;
}

int pcto_step_index(pcto_env* h, const int64_t* leaf_index, int32_t auto_reset) {
  int rc = ready(h);
  if (rc) return rc;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 8)
  for (int e = 0; e < h->N; e++) {
    double* obs = h->obs + (size_t)e * h->row_len;
    double row[9];
    int64_t li = leaf_index[e];
    if (li < 0 || li >= h->L) li = 0;
    memcpy(row, obs + 9 * ((size_t)h->I + (size_t)li), sizeof row);
    any_step(h, e, row, 9, obs);
    if (h->done[e] && auto_reset) any_reset(h, e, obs);
  }
  return PCT_OK;
}

int pcto_step_hash_policy(pcto_env* h, int32_t n_steps) {
  int rc = ready(h);
  if (rc) return rc;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 8)
  for (int e = 0; e < h->N; e++) {
    oenv* s = &h->envs[e];
    double* obs = h->obs + (size_t)e * h->row_len;
    for (int it = 0; it < n_steps; it++) {
      const double* leaf = obs + 9 * (size_t)h->I;
      int k = 0;
      for (int i = 0; i < h->L; i++) k += leaf[9 * i + 8] != 0;
      uint32_t tt = h->cfg.env_kind == PCT_ENV_CONTINUOUS ? pctc_t(h, e) : s->t;
      int li = k > 0 ? (int)(pct_mix32((uint32_t)(h->cfg.env_id_base + e), tt) % (uint32_t)k) : 0;
      double row[9];
      memcpy(row, leaf + 9 * li, sizeof row);
      any_step(h, e, row, 9, obs);
      if (h->done[e]) any_reset(h, e, obs);
    }
  }
  return PCT_OK;
}

/* ---- heuristic.py: the placement rules of the heuristic baselines, as in-env policies ---------
 * rotation convention of heuristic.py:260-271 (and every heuristic there): not the one of EMSPoint */
static void heur_rot(const int nb[3], int rot, int* x, int* y, int* z) {
  switch (rot) {
    case 0: *x = nb[0]; *y = nb[1]; *z = nb[2]; break;
    case 1: *y = nb[0]; *x = nb[1]; *z = nb[2]; break;
    case 2: *z = nb[0]; *x = nb[1]; *y = nb[2]; break;
    case 3: *z = nb[0]; *y = nb[1]; *x = nb[2]; break;
    case 4: *x = nb[0]; *z = nb[1]; *y = nb[2]; break;
    default: *y = nb[0]; *z = nb[1]; *x = nb[2]; break;
  }
}

/* Picks the placement heuristic `kind` would step with.  Returns 0 if it finds none (the
 * reference loop then records the episode and resets the env WITHOUT stepping). */
static int heur_choose(const pcto_env* h, int e, const oenv* s, int kind, int* olx, int* oly, int* ox, int* oy, int* oz) {
  const int W = h->cfg.container[0], L = h->cfg.container[1], H = h->cfg.container[2], A = h->A;
  const int orientation = h->cfg.setting == 2 ? 6 : 2;
  const int* nb = s->next_box;
  int found = 0;
  if (kind == PCT_HEUR_DBL || kind == PCT_HEUR_HM) {
    /* heuristic.py:431-498 DBL / :232-298 heightmap_min: every (lx, ly) the UNROTATED item fits at,
     * every rotation; score lx + ly + 100*height resp. lx + ly + 100*sum(new heightmap); first best */
    int64_t best = 0, total = 0;
    if (kind == PCT_HEUR_HM)
      for (int i = 0; i < W; i++) for (int j = 0; j < L; j++) total += s->plain[i * A + j];
    for (int lx = 0; lx < W - nb[0] + 1; lx++)
      for (int ly = 0; ly < L - nb[1] + 1; ly++)
        for (int rot = 0; rot < orientation; rot++) {
          int x, y, z;
          heur_rot(nb, rot, &x, &y, &z);
          int max_h = footprint_max(h, s, lx, ly, x, y);
          if (max_h < 0 || !check_box(h, s, x, y, lx, ly, z, max_h, s->next_den, 1)) continue;
          int64_t score;
          if (kind == PCT_HEUR_DBL) score = lx + ly + 100 * (int64_t)max_h;
          else { /* update_height_graph :316-326: the footprint becomes max_h + z */
            int64_t under = 0;
            for (int i = lx; i < lx + x; i++) for (int j = ly; j < ly + y; j++) under += s->plain[i * A + j];
            score = lx + ly + 100 * (total - under + (int64_t)(max_h + z) * x * y);
          }
          if (!found || score < best) { best = score; found = 1; *olx = lx; *oly = ly; *ox = x; *oy = y; *oz = z; }
        }
    return found;
  }
  if (kind == PCT_HEUR_RANDOM) {
    /* :300-362 random: uniform over the feasible (lx, ly, rotation) of the DBL / HM enumeration; the
     * reference's np.random.randint(0, n) is the counter-keyed pct_mix32(global env id, t) % n */
    int n = 0;
    for (int pass = 0; pass < 2; pass++) {
      int pick = pass ? (int)(pct_mix32((uint32_t)(h->cfg.env_id_base + e), s->t) % (uint32_t)n) : -1, c = 0;
      for (int lx = 0; lx < W - nb[0] + 1; lx++)
        for (int ly = 0; ly < L - nb[1] + 1; ly++)
          for (int rot = 0; rot < orientation; rot++) {
            int x, y, z;
            heur_rot(nb, rot, &x, &y, &z);
            int max_h = footprint_max(h, s, lx, ly, x, y);
            if (max_h < 0 || !check_box(h, s, x, y, lx, ly, z, max_h, s->next_den, 1)) continue;
            if (pass && c == pick) { *olx = lx; *oly = ly; *ox = x; *oy = y; *oz = z; return 1; }
            c++;
          }
      if (!pass) { n = c; if (n == 0) return 0; }
    }
    return 0;
  }
  if (kind == PCT_HEUR_OBPH) {
    /* :364-425 OnlineBPH: EMS sorted by (z, y, x), stable; first feasible (EMS corner, rotation) */
    int n = s->n_ems;
    int* idx = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; i++) {
      int j = i;
      while (j > 0) {
        const int64_t* a = s->ems + 6 * idx[j - 1];
        const int64_t* b = s->ems + 6 * i;
        int greater = a[2] > b[2] || (a[2] == b[2] && (a[1] > b[1] || (a[1] == b[1] && a[0] > b[0])));
        if (!greater) break;
        idx[j] = idx[j - 1];
        j--;
      }
      idx[j] = i;
    }
    for (int q = 0; q < n && !found; q++) {
      const int64_t* e = s->ems + 6 * idx[q];
      for (int rot = 0; rot < orientation; rot++) {
        int x, y, z;
        heur_rot(nb, rot, &x, &y, &z);
        if (drop_box_virtual(h, s, x, y, z, (int)e[0], (int)e[1])) {
          found = 1; *olx = (int)e[0]; *oly = (int)e[1]; *ox = x; *oy = y; *oz = z;
          break;
        }
      }
    }
    free(idx);
    return found;
  }
  if (kind == PCT_HEUR_LSAH) {
    /* :138-226 LASH: least surface area of the bounding box of everything packed; maxXY / minXY are
     * the running extents of the placements of this episode (:204-207) */
    int maxX = 0, maxY = 0, minX = W, minY = L;
    for (int i = 0; i < s->n_boxes; i++) {
      const obox* b = &s->boxes[i];
      if (b->lx + b->x > maxX) maxX = b->lx + b->x;
      if (b->ly + b->y > maxY) maxY = b->ly + b->y;
      if (b->lx < minX) minX = b->lx;
      if (b->ly < minY) minY = b->ly;
    }
    int64_t best = (int64_t)W * L + (int64_t)L * H + (int64_t)H * W;
    int bd[3] = {0, 0, 0};
    for (int q = 0; q < s->n_ems; q++) {
      const int64_t* e = s->ems + 6 * q;
      int dx = (int)(e[3] - e[0]), dy = (int)(e[4] - e[1]), dz = (int)(e[5] - e[2]);
      for (int rot = 0; rot < orientation; rot++) {
        int x, y, z;
        heur_rot(nb, rot, &x, &y, &z);
        if (!(dx >= x && dy >= y && dz >= z)) continue;
        int lx = (int)e[0], ly = (int)e[1];
        int height = footprint_max(h, s, lx, ly, x, y);
        if (height < 0 || !check_box(h, s, x, y, lx, ly, z, height, s->next_den, 1)) continue;
        int ex = (lx + x > maxX ? lx + x : maxX) - (lx < minX ? lx : minX);
        int ey = (ly + y > maxY ? ly + y : maxY) - (ly < minY ? ly : minY);
        int64_t score = (int64_t)ex * ey + (int64_t)(height + z) * ey + (int64_t)(height + z) * ex;
        int take = 0;
        if (score < best) take = 1;
        else if (score == best && found) {
          int m1 = dx - x < dy - y ? dx - x : dy - y; if (dz - z < m1) m1 = dz - z;
          int m2 = bd[0] - x < bd[1] - y ? bd[0] - x : bd[1] - y; if (bd[2] - z < m2) m2 = bd[2] - z;
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
    /* :500-569 BR: the EMS with the best eval_ems (volume + number of item types that fit unrotated,
     * + 10 if all do); first best in (EMS, rotation) order */
    int64_t best = 0;
    for (int q = 0; q < s->n_ems; q++) {
      const int64_t* e = s->ems + 6 * q;
      int dx = (int)(e[3] - e[0]), dy = (int)(e[4] - e[1]), dz = (int)(e[5] - e[2]);
      int64_t sc = -1;
      for (int rot = 0; rot < orientation; rot++) {
        int x, y, z;
        heur_rot(nb, rot, &x, &y, &z);
        if (!(dx >= x && dy >= y && dz >= z)) continue;
        int lx = (int)e[0], ly = (int)e[1];
        if (!drop_box_virtual(h, s, x, y, z, lx, ly)) continue;
        if (sc < 0) {
          int fits = 0;
          for (int i = 0; i < h->n_items; i++)
            fits += dx >= h->item_set[3 * i] && dy >= h->item_set[3 * i + 1] && dz >= h->item_set[3 * i + 2];
          sc = (int64_t)dx * dy * dz + fits + (fits == h->n_items ? 10 : 0);
        }
        if (!found || sc > best) { best = sc; found = 1; *olx = lx; *oly = ly; *ox = x; *oy = y; *oz = z; }
      }
    }
    return found;
  }
  if (kind == PCT_HEUR_MACS) {
    /* :11-136 MACS: the placement (EMS, rotation, corner) maximising the sum, over the levels below the
     * item's base, of the largest empty rectangle of the level AFTER the placement.  The heuristic's own
     * voxel container (:47-52) is nonzero exactly below the heightmap (a placement fills its box and
     * marks the empty cells under it), so cell (i, j) is empty at level h iff plain[i][j] <= h; for
     * h < base the footprint of the candidate is not. */
    int64_t best = 0;
    for (int q = 0; q < s->n_ems; q++) {
      const int64_t* e = s->ems + 6 * q;
      int dx = (int)(e[3] - e[0]), dy = (int)(e[4] - e[1]), dz = (int)(e[5] - e[2]);
      for (int rot = 0; rot < orientation; rot++) {
        int x, y, z;
        heur_rot(nb, rot, &x, &y, &z);
        if (!(dx >= x && dy >= y && dz >= z)) continue;
        for (int corner = 0; corner < 4; corner++) {
          int lx = (corner & 1) ? (int)e[3] - x : (int)e[0];
          int ly = (corner & 2) ? (int)e[4] - y : (int)e[1];
          int height = footprint_max(h, s, lx, ly, x, y);
          if (height < 0 || !check_box(h, s, x, y, lx, ly, z, height, s->next_den, 1)) continue;
          int64_t score = 0;
          for (int lv = 0; lv < height; lv++) {
            int level_max = 0;
            for (int i1 = 0; i1 < W; i1++)
              for (int j1 = 0; j1 < L; j1++)
                for (int i2 = i1; i2 < W; i2++)
                  for (int j2 = j1; j2 < L; j2++) {
                    int area = (i2 - i1 + 1) * (j2 - j1 + 1);
                    if (area <= level_max) continue;
                    int ok = 1;
                    for (int i = i1; i <= i2 && ok; i++)
                      for (int j = j1; j <= j2; j++) {
                        int inside = i >= lx && i < lx + x && j >= ly && j < ly + y;
                        if (inside || s->plain[i * A + j] > lv) { ok = 0; break; }
                      }
                    if (ok) level_max = area;
                  }
            score += level_max;
          }
          if (!found || score > best) { best = score; found = 1; *olx = lx; *oly = ly; *ox = x; *oy = y; *oz = z; }
        }
      }
    }
    return found;
  }
  return 0;
}

int pcto_step_heuristic(pcto_env* h, int32_t kind, int32_t n_steps) {
  int rc = ready(h);
  if (rc) return rc;
  if (h->cfg.lnes != PCT_LNES_EMS) return fail(PCT_ERR_UNSUPPORTED, "the heuristics read the EMS list (LNES = EMS)");
  if (kind < PCT_HEUR_LSAH || kind > PCT_HEUR_RANDOM) return fail(PCT_ERR_INVALID_ARG, "unknown heuristic");
  if (h->cfg.env_kind == PCT_ENV_CONTINUOUS) {
    /* tools.py:217-218: only LSAH, OnlineBPH and BR run on PackingContinuous */
    if (kind != PCT_HEUR_LSAH && kind != PCT_HEUR_OBPH && kind != PCT_HEUR_BR)
      return fail(PCT_ERR_UNSUPPORTED, "only LSAH, OnlineBPH and BR are allowed for the continuous environment (tools.py:217-218)");
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 8)
    for (int e = 0; e < h->N; e++) {
      double* obs = h->obs + (size_t)e * h->row_len;
      for (int it = 0; it < n_steps; it++) {
        double lx = 0, ly = 0, x = 0, y = 0, z = 0;
        if (pctc_heur_choose(h, e, kind, &lx, &ly, &x, &y, &z)) pctc_step_place(h, e, lx, ly, x, y, z, obs);
        else pctc_giveup(h, e);
        if (h->done[e]) any_reset(h, e, obs);
      }
    }
    return PCT_OK;
  }
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 8)
  for (int e = 0; e < h->N; e++) {
    oenv* s = &h->envs[e];
    double* obs = h->obs + (size_t)e * h->row_len;
    for (int it = 0; it < n_steps; it++) {
      int lx = 0, ly = 0, x = 0, y = 0, z = 0;
      if (heur_choose(h, e, s, kind, &lx, &ly, &x, &y, &z)) {
        /* env.next_box = [x, y, z]; env.step([0, lx, ly]) == the 6-vector leaf form of the same placement */
        double row[6] = {(double)lx, (double)ly, 0.0, (double)(lx + x), (double)(ly + y), (double)z};
        any_step(h, e, row, 6, obs);
      } else { /* no feasible placement: the episode is recorded and the env reset, no step() */
        s->t++;
        h->reward[e] = 0.0;
        h->done[e] = 1;
        h->counter[e] = s->n_boxes;
        h->ratio[e] = get_ratio(h, s);
      }
      if (h->done[e]) any_reset(h, e, obs);
    }
  }
  return PCT_OK;
}

int pcto_debug_state(pcto_env* h, int32_t e, int32_t* heightmap, int32_t* ems, int32_t cap_ems, int32_t* n_ems,
                     int32_t* n_boxes, int32_t* next_item, int64_t* draw_cursor) {
  if (!h || e < 0 || e >= h->N) return fail(PCT_ERR_INVALID_ARG, "bad env id");
  const oenv* s = &h->envs[e];
  if (heightmap) for (int i = 0; i < h->A * h->A; i++) heightmap[i] = s->plain[i];
  if (ems) for (int i = 0; i < s->n_ems && i < cap_ems; i++) for (int c = 0; c < 6; c++) ems[6 * i + c] = (int32_t)s->ems[6 * i + c];
  if (n_ems) *n_ems = s->n_ems;
  if (n_boxes) *n_boxes = s->n_boxes;
  if (next_item) { next_item[0] = s->next_box[0]; next_item[1] = s->next_box[1]; next_item[2] = s->next_box[2]; }
  if (draw_cursor) *draw_cursor = (int64_t)s->cursor;
  return PCT_OK;
}
