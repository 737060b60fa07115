// stab_host.cpp -- TEST HARNESS: compiles the product's restructured stability check
// (online-3d-bpp-pct_amd/csrc/pct_stab.cuh, the code the GPU lanes run) for the host and
// exposes it under the oracle's stab_* interface, so that `make -C tests/host` yields an
// oracle variant (libpct_oracle_prodstab.so) whose stability decisions come from the product
// source.  tests/test_stab_host.py runs the setting-1 reference fixtures through it: the
// restructuring (pooled LDS-shaped state, up-lists instead of dictionaries, the virtual walk as
// independent tasks) is thereby checked on the CPU against the reference before it is trusted on
// the GPU.  The tasks a wave pops 64 at a time are popped one at a time here, LAST IN FIRST OUT
// like the device queue (their order cannot matter: see the header of pct_stab.cuh).
// Nothing in the product uses this.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../online-3d-bpp-pct_amd/csrc/pct_stab.cuh"

struct stab {
  int cap;
  int n;
  std::vector<double> geo;  // [cap][9] lx,ly,lz,xe,ye,ze,sx,sy,sz
  bool cont;
  pct::StabCaps caps;
  std::vector<unsigned char> mem, ws;
  pct::StabState st;
  int overflow, ill;
  int max_k0, max_q, max_ent, max_poly, max_split_k;  // statistics (PCT_STAB_HOST_STATS=1 prints them at stab_free)
};

struct GeoFn {
  const double* g;
  static constexpr bool kSquareIsPow = false;  // the host flavour always runs the pow restatement
  void operator()(int i, double out[9]) const { memcpy(out, g + 9 * (size_t)i, 9 * sizeof(double)); }
};
struct Task { int S; double stk[4]; };

extern "C" {
struct stab* stab_create(int cap, double eps) {
  stab* s = new stab();
  s->cont = eps > 0;
  s->cap = cap + 2;
  s->n = 0;
  s->overflow = 0;
  s->ill = 0;
  s->geo.assign((size_t)s->cap * 9, 0.0);
  // deliberately small pools by default (PCT_STAB_HOST_SP / _PP / _WS override): the capacity paths are exercised too
  const char* e;
  s->caps.SP = (e = getenv("PCT_STAB_HOST_SP")) ? atoi(e) : 4000;
  s->caps.PP = (e = getenv("PCT_STAB_HOST_PP")) ? atoi(e) : 60000;
  s->caps.ws_bytes = (e = getenv("PCT_STAB_HOST_WS")) ? atoi(e) : pct::stab_ws_need(255);
  s->caps.queue = 0;
  s->caps.lsq_n = pct::STAB_LSQ;
  s->caps.lsq_bytes = 0;
  s->mem.assign(pct::stab_state_bytes(s->cap, s->caps) + 16, 0);
  s->ws.assign((size_t)s->caps.ws_bytes + 16, 0);
  s->st = pct::stab_carve(s->mem.data(), s->cap, s->caps);
  return s;
}
void stab_reset(struct stab* s) { s->n = 0; s->st.n_ent = 0; s->st.n_poly = 0; }
void stab_free(struct stab* s) {
  if (getenv("PCT_STAB_HOST_STATS"))
    fprintf(stderr, "stab stats: max supporters of a candidate %d, max queue %d, max pool entries %d, max polygon vertices %d\n",
            s->max_k0, s->max_q, s->max_ent, s->max_poly);
  delete s;
}
int stab_overflowed(struct stab* s) { return s->overflow; }
int stab_ill_conditioned(struct stab* s) { return s->ill & 1; }
int stab_ill_commit(struct stab* s) { return (s->ill >> 1) & 1; }
void stab_set_ill_near(int) {}  // (the tie notice is an analysis mode of the oracle only)
// the product's PCT_LSTSQ_GELSD (csrc/pct_gelsd.cuh compiled for the host) under the oracle's switch
void stab_set_lstsq_mode(int mode) { pct::g_stab_host_gelsd = mode; }
int stab_get_lstsq_mode(void) { return pct::g_stab_host_gelsd; }
// pct_oracle.h exports the oracle's own restatement under this name; here it is the PRODUCT's solve (geometry-free entry for tests):
// row-major M x N system with the right-hand side e_M is not what the product takes, so tests/test_stab_host.py uses gelsd_host_split
int gelsd_lstsq(const double*, const double*, int, int, double*, int*, double*, int*) { return -1; }
int gelsd_host_split(int k, const double* centres, double s0, double s1, double* x, int* ill) {
  std::vector<double> ws((size_t)pct::gelsd::solve_doubles(k) + 2, 0.0);
  bool i2 = false;
  const bool ok = pct::gelsd::split_t(ws.data(), k, centres, s0, s1, pct::StabDot2{pct::g_stab_host_gelsd == 2}, x, i2, pct::g_stab_host_gelsd == 2);
  *ill = i2;
  return ok;
}
double gelsd_host_dnrm2(int n, const double* x, int incx) { return pct::gelsd::dnrm2(n, x, incx); }
// dbdsqr3 (n = 3: d, e, the sweep's rotations in registers, every index static) against the generic routine: `count` random
// bidiagonals of the kinds tests/test_stab_host.py::test_product_dbdsqr_equals_oracle_dbdsqr draws.
// out[0] = systems, out[1] = differences (any bit of d, VT, cc or the return value).
void gelsd_host_dbdsqr3_sweep(long count, unsigned long long seed, long* out) {
  unsigned long long s = seed;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
  auto gauss = [&]() { return sqrt(-2.0 * log(rnd() + 1e-300)) * cos(6.283185307179586 * rnd()); };
  const pct::gelsd::Grp one = {0, 1};
  out[0] = out[1] = 0;
  for (long t = 0; t < count; t++) {
    const int n = 3;
    double d[4], e[4] = {0, 0, 0, 0}, vt[16], c[4], d2[4], e2[4], vt2[16], c2[4], work[32];
    for (int i = 0; i < n; i++) { d[i] = gauss(); c[i] = gauss(); }
    for (int i = 0; i < n - 1; i++) e[i] = gauss();
    if (t % 3 == 0) for (int i = 0; i < n; i++) d[i] *= pow(10.0, -(double)(int)(rnd() * 10));   // graded: zero-shift sweeps
    if (t % 4 == 0) for (int i = 0; i < n / 2; i++) { const double x = d[i]; d[i] = d[n - 1 - i]; d[n - 1 - i] = x; }
    if (t % 5 == 0) e[(int)(rnd() * (n - 1))] = 0.0;
    if (t % 7 == 0) d[(int)(rnd() * n)] = 0.0;
    if (t % 11 == 0) for (int i = 0; i < n; i++) d[i] = rint(d[i] * 4) / 4;  // ties in the ordering
    if (t % 13 == 0) for (int i = 0; i < n - 1; i++) e[i] *= 1e-9;             // nearly diagonal: the convergence tests fire
    for (int i = 0; i < n * n; i++) vt[i] = (i / n == i % n) ? 1.0 : 0.0;
    memcpy(d2, d, sizeof d); memcpy(e2, e, sizeof e); memcpy(vt2, vt, sizeof vt); memcpy(c2, c, sizeof c);
    const bool ra = pct::gelsd::dbdsqr3(one, d, e, vt, c);
    const bool rb = pct::gelsd::dbdsqr_generic(one, n, d2, e2, vt2, c2, work);
    out[0]++;
    if (ra != rb || memcmp(d, d2, n * 8) || memcmp(vt, vt2, n * n * 8) || memcmp(c, c2, n * 8)) out[1]++;
  }
}
// Bulk checks that would be slow through ctypes, one call per element.  `count` pseudo-random vectors of 2..8 elements drawn
// from the value classes the stability systems hold (uniform, small integers and halves, ratios, 1e-6-lattice values, wide
// exponent ranges, zeros): the certified double-double dnrm2 (pct_gelsd.cuh dnrm2_certified) against the x87 emulation.
// out[0] = vectors, out[1] = certified (not handed to the emulation), out[2] = differences.
void gelsd_host_dnrm2_certified_sweep(long count, unsigned long long seed, long* out) {
  unsigned long long s = seed;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
  out[0] = out[1] = out[2] = 0;
  for (long it = 0; it < count; it++) {
    const int n = 2 + (int)(rnd() * 7);
    const int mode = (int)(rnd() * 7);
    double x[8];
    for (int i = 0; i < n; i++) {
      double v;
      switch (mode) {
        case 0: v = rnd() * 2 - 1; break;
        case 1: v = (double)((int)(rnd() * 21) - 10); break;
        case 2: v = ((int)(rnd() * 21) - 10) / 2.0; break;
        case 3: v = exp(rnd() * 40 - 20) * (rnd() < 0.5 ? -1 : 1); break;
        case 4: v = rnd() < 0.4 ? 0.0 : (rnd() < 0.5 ? 1.0 : -(double)((int)(rnd() * 1000)) / (1 + (int)(rnd() * 1000))); break;
        case 5: v = exp(rnd() * 600 - 300); break;  // (beyond +-1e100 the certificate must decline)
        default: v = rint(rnd() * 1e6) / 1e6; break;
      }
      x[i] = v;
    }
    const double ref = pct::gelsd::dnrm2_ext(n, x, 1);
    double f;
    out[0]++;
    if (!pct::gelsd::dnrm2_certified(n, x, 1, f)) continue;
    out[1]++;
    if (memcmp(&ref, &f, 8)) out[2]++;
  }
}
// pct_pow.cuh (glibc's pow as its FMA build executes it) against the live libm, on lengths of the lever rule: `count` values from
// the continuous env's domain (norms of differences of 1e-6-lattice contact centres halved, i.e. multiples of 5e-7 up to `span`).
// out[0] = lengths, out[1] = differences to libm's pow(len, 2.0), out[2] = lengths with pow(len, 2.0) != len * len.
void pow_host_sweep_continuous(long count, unsigned long long seed, double span, long* out) {
  unsigned long long s = seed;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
  out[0] = out[1] = out[2] = 0;
  volatile double two = 2.0;  // (keeps the compiler from folding pow(x, 2) into x * x)
  for (long it = 0; it < count; it++) {
    // contact centres are (a + b) / 2 of np.around(., 6) values; the base line is a difference of two of them
    const double c0x = (rint(rnd() * span * 1e6) / 1e6 + rint(rnd() * span * 1e6) / 1e6) / 2, c0y = (rint(rnd() * span * 1e6) / 1e6 + rint(rnd() * span * 1e6) / 1e6) / 2;
    const double c1x = (rint(rnd() * span * 1e6) / 1e6 + rint(rnd() * span * 1e6) / 1e6) / 2, c1y = (rint(rnd() * span * 1e6) / 1e6 + rint(rnd() * span * 1e6) / 1e6) / 2;
    const double t0 = c0x - c1x, t1 = c0y - c1y;
    const double len = sqrt(fma(t1, t1, t0 * t0));
    if (!(len > 0)) continue;
    const double a = pow(len, two), b = pct::pow_glibc_fma(len, 2.0);
    out[0]++;
    if (memcmp(&a, &b, 8)) out[1]++;
    if (a != len * len) out[2]++;
  }
}
// ... and the DISCRETE env's lengths, exhaustively: sqrt(dx^2 + dy^2) for dx, dy multiples of 1/2 up to `side` (contact centres of
// integer rectangles), both np.dot flavours.  out[0] = lengths, out[1] = pow_glibc_fma != libm, out[2] = libm pow(len, 2.0) != len * len.
void pow_host_sweep_discrete(int side, long* out) {
  out[0] = out[1] = out[2] = 0;
  volatile double two = 2.0;
  for (int i = 0; i <= 2 * side; i++)
    for (int j = 0; j <= 2 * side; j++) {
      if (!i && !j) continue;
      const double t0 = i / 2.0, t1 = j / 2.0;
      for (int avx2 = 0; avx2 < 2; avx2++) {
        const double len = sqrt(avx2 ? t0 * t0 + t1 * t1 : fma(t1, t1, t0 * t0));
        const double a = pow(len, two), b = pct::pow_glibc_fma(len, 2.0);
        out[0]++;
        if (memcmp(&a, &b, 8)) out[1]++;
        if (a != len * len) out[2]++;
      }
    }
}
double pow_host(double x, double y) { return pct::pow_glibc_fma(x, y); }
// dbdsqr('U', n, ncvt = n, 0, 1): d[n], e[n - 1], vt[n * n] (column-major), c[n], work[4 n]
int gelsd_host_dbdsqr(int n, double* d, double* e, double* vt, double* c, double* work) { const pct::gelsd::Grp one = {0, 1}; return pct::gelsd::dbdsqr(one, n, d, e, vt, c, work) ? 0 : 1; }

int stab_check(struct stab* s, double x, double y, double z, double lx, double ly, double max_h, double density,
               int virtual_) {
  GeoFn geo{s->geo.data()};
  bool ill = false;
  double cand[9] = {lx, ly, max_h, lx + x, ly + y, max_h + z, x, y, z};
  if (virtual_) {
    if (s->cont ? (fabs(max_h) < 1e-6) : (max_h == 0)) return 1;  // space.py:448-449
    int kcap = 1;
    while (kcap < pct::STAB_NSUP_MAX && pct::stab_ws_need(kcap + 1) <= s->caps.ws_bytes) kcap++;
    pct::StabWsView w = pct::stab_ws_view(s->ws.data(), kcap);
    const int k = s->cont ? pct::stab_find_supporters<true>(geo, s->n, cand, w.ids, kcap)
                          : pct::stab_find_supporters<false>(geo, s->n, cand, w.ids, kcap);
    if (k > kcap) { s->overflow = 1; return 0; }
    if (k > s->max_k0) s->max_k0 = k;
    std::vector<Task> q;
    auto emit = [&](int Si, const double child[4]) { Task t; t.S = Si; memcpy(t.stk, child, sizeof t.stk); q.push_back(t); };
    int rc = s->cont ? pct::stab_level0<true>(geo, s->st, cand, density, k, w, ill, emit)
                     : pct::stab_level0<false>(geo, s->st, cand, density, k, w, ill, emit);
    while (rc == 1 && !q.empty()) {
      if ((int)q.size() > s->max_q) s->max_q = (int)q.size();
      Task t = q.back();
      q.pop_back();
      rc = s->cont ? pct::stab_visit<true>(geo, s->st, t.S, t.stk, ill, emit) : pct::stab_visit<false>(geo, s->st, t.S, t.stk, ill, emit);
    }
    if (ill) s->ill |= 1;
    if (rc < 0) { s->overflow = 1; return 0; }
    return rc;
  }
  memcpy(s->geo.data() + 9 * (size_t)s->n, cand, sizeof cand);
  // a failed commit leaves the pools as they were (the device requeues / resets the env; here the caller goes on)
  const int ne = s->st.n_ent, np = s->st.n_poly;
  std::vector<unsigned char> snap(s->mem);
  int rc = s->cont ? pct::stab_commit<true>(geo, s->st, s->n, density, s->ws.data(), s->caps.ws_bytes, ill)
                   : pct::stab_commit<false>(geo, s->st, s->n, density, s->ws.data(), s->caps.ws_bytes, ill);
  if (ill) s->ill |= 3;
  if (rc < 0) s->overflow = 1;
  if (s->st.n_ent > s->max_ent) s->max_ent = s->st.n_ent;
  if (s->st.n_poly > s->max_poly) s->max_poly = s->st.n_poly;
  if (rc == 1) { s->n++; return 1; }
  s->mem = snap;
  s->st.n_ent = ne; s->st.n_poly = np;
  return 0;
}
}
