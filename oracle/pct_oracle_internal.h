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
  int source;
  oenv* envs;
  double* obs;
  double* reward;
  uint8_t* done;
  int32_t* counter;
  double* ratio;
  uint32_t* flags;
};


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
int stab_check(struct stab* s, double x, double y, double z, double lx, double ly, double max_h, double density,
               int virtual_);

/* continuous env (pct_oracle_cont.c) */
int pctc_alloc(struct pcto_env* h);
void pctc_free(struct pcto_env* h);
void pctc_reset(struct pcto_env* h, int e, double* obs);
void pctc_step(struct pcto_env* h, int e, const double* act, int len, double* obs, double* reward, uint8_t* done,
               int32_t* counter, double* ratio, uint32_t* flags);
uint32_t pctc_t(const struct pcto_env* h, int e);
int pctc_debug_state(struct pcto_env* h, int e, double* ems, int cap_ems, int* n_ems, int* n_boxes, double* next_item,
                     int64_t* cursor);
#endif
