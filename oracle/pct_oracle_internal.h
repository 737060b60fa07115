/* pct_oracle_internal.h -- shared between the discrete and continuous restatements
 * (TEST INFRASTRUCTURE, see pct_oracle.h). */
#ifndef PCT_ORACLE_INTERNAL_H
#define PCT_ORACLE_INTERNAL_H
#include <stddef.h>
#include "pct_oracle.h"

struct stab;
typedef struct {
  int x, y, z, lx, ly, lz;
} obox; /* D/space.py:26-33 Box geometry */

typedef struct {
  /* D/space.py:271-314 Space */
  int* plain;      /* [A*A] heightmap, plain[x*A+y] */
  double* box_vec; /* [I*9] */
  obox* boxes;
  int n_boxes; /* len(self.boxes) */
  int box_idx;
  int64_t* ems; /* [n_ems*6] */
  int n_ems, cap_ems;
  /* D/bin3D.py env */
  int next_box[3];
  double next_den;
  int queue_item[3]; /* box_creator.box_list[0] */
  int queue_len;
  uint64_t cursor; /* draws taken from the item source */
  uint32_t t;      /* lifetime step counter (hash policy) */
  struct stab* stab; /* stability state (settings 1 / 3), pct_oracle_stab.c */
  int traj;          /* dataset mode: LoadBoxCreator.index (binCreator.py:46,54-55) */
  uint64_t oc;       /* observations produced so far (shuffle key) */
  /* strict NumPy-stream mode (pcto_set_numpy_rng): this env's MT19937 state, as np.random.seed(seed + rank)
   * leaves it in the env's worker process (bin3D.py:47-54, envs.py:49) */
  uint32_t mt[624];
  int mt_pos;
} oenv;

struct cenv; /* continuous per-env state, pct_oracle_cont.c */

struct pcto_env {
  pct_config cfg;
  struct cenv* cenvs; /* env_kind == PCT_ENV_CONTINUOUS */
  int N, A, I, L, row_len;
  int low_bound; /* lattice units */
  int sample_left, sample_right; /* continuous sampler bounds, lattice units */
  int32_t* item_set;
  int n_items;
  int32_t* stream;
  int64_t T;
  int32_t* ds_len; /* dataset mode: stream is [n_traj,max_len,3] */
  int ds_ntraj, ds_maxlen;
  double* den_stream; /* setting 3: scripted densities [N,den_T], or NULL */
  int64_t den_T;
  double* ds_den;     /* setting 3, dataset mode: [n_traj,max_len] fourth column, or NULL */
  uint64_t seed;
  uint64_t shuffle_seed;
  int rng_numpy; /* 1: item picks, densities and the candidate shuffle consume the env's NumPy MT19937 stream */
  int source;
  oenv* envs;
  double* obs;
  double* reward;
  uint8_t* done;
  int32_t* counter;
  double* ratio;
  uint32_t* flags;
};


/* ---- NumPy's legacy RandomState on MT19937 (numpy/random/src/mt19937/mt19937.c, legacy-distributions.c,
 * distributions.c; the module-level np.random.* functions the reference calls) --------------------------- */
static inline void npmt_seed(uint32_t* mt, int* pos, uint32_t seed) { /* mt19937_seed == init_genrand */
  for (int i = 0; i < 624; i++) {
    mt[i] = seed;
    seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
  }
  *pos = 624;
}
static inline uint32_t npmt_next32(uint32_t* mt, int* pos) { /* mt19937_next: regenerate all 624 words, then temper */
  if (*pos >= 624) {
    int kk = 0;
    for (; kk < 624 - 397; kk++) {
      uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < 623; kk++) {
      uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    *pos = 0;
  }
  uint32_t y = mt[(*pos)++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
/* random_interval(max) / the masked rejection of legacy randint(0, max + 1): smallest bit mask >= max, draw
 * 32-bit words until (word & mask) <= max */
static inline uint32_t npmt_interval(uint32_t* mt, int* pos, uint32_t max) {
  if (max == 0) return 0;
  uint32_t mask = max;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  uint32_t v;
  while ((v = (npmt_next32(mt, pos) & mask)) > max) {}
  return v;
}
static inline double npmt_double(uint32_t* mt, int* pos) { /* random_sample: 53 bits from two words */
  uint32_t a = npmt_next32(mt, pos) >> 5, b = npmt_next32(mt, pos) >> 6;
  return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

/* D/bin3D.py:75-84 next_den of observation number `oc`; `traj`/`item_index` locate the previewed
 * item in dataset mode */
static inline double pcto_next_density(const struct pcto_env* h, int e, uint64_t oc, int traj, uint64_t item_index) {
  if (h->cfg.setting != 3) return 1.0;
  if (h->source == PCT_ITEMS_DATASET) { /* self.next_box[3] */
    int t = traj < h->ds_ntraj ? traj : h->ds_ntraj - 1;
    if (!h->ds_den || t < 0 || item_index >= (uint64_t)h->ds_len[t]) return 1.0;
    return h->ds_den[(size_t)t * h->ds_maxlen + (size_t)item_index];
  }
  if (h->den_stream) return h->den_stream[(size_t)e * (size_t)h->den_T + (size_t)(oc % (uint64_t)h->den_T)];
  return pct_density(h->seed, (uint64_t)(h->cfg.env_id_base + e), oc);
}

/* stability (pct_oracle_stab.c) */
struct stab* stab_create(int cap, double eps);
void stab_reset(struct stab* s);
void stab_free(struct stab* s);
int stab_ill_conditioned(const struct stab* s); /* sticky notice, see pct_oracle_stab.c */
int stab_ill_commit(const struct stab* s);      /* ... raised by a solve of a commit walk */
void stab_set_ill_near(int on);
void stab_set_lstsq_mode(int mode); /* 0: Jacobi stand-in, 1 (default): dgelsd as NumPy's OpenBLAS executes it, 2: ... on AVX2 hosts */
int stab_get_lstsq_mode(void);
/* pct_oracle_gelsd.c */
void gelsd_set_kernel_set(int s); /* 0: OpenBLAS "SkylakeX" kernels (AVX-512 hosts), 1: "Haswell" (AVX2 hosts, AMD Zen) */
int gelsd_get_kernel_set(void);
int gelsd_lstsq(const double* Arow, const double* brow, int M, int N, double* x, int* rank_out, double* sv_out, int* near_cut);
int stab_check(struct stab* s, double x, double y, double z, double lx, double ly, double max_h, double density,
               int virtual_);

/* continuous env (pct_oracle_cont.c) */
int pctc_alloc(struct pcto_env* h);
void pctc_free(struct pcto_env* h);
void pctc_reset(struct pcto_env* h, int e, double* obs);
void pctc_step(struct pcto_env* h, int e, const double* act, int len, double* obs, double* reward, uint8_t* done,
               int32_t* counter, double* ratio, uint32_t* flags);
uint32_t pctc_t(const struct pcto_env* h, int e);
const struct stab* pctc_stab(const struct pcto_env* h, int e);
int pctc_heur_choose(const struct pcto_env* h, int e, int kind, double* olx, double* oly, double* ox, double* oy, double* oz);
void pctc_step_place(struct pcto_env* h, int e, double lx, double ly, double x, double y, double z, double* obs);
void pctc_giveup(struct pcto_env* h, int e);
int pctc_debug_state(struct pcto_env* h, int e, double* ems, int cap_ems, int* n_ems, int* n_boxes, double* next_item,
                     int64_t* cursor);
#endif
