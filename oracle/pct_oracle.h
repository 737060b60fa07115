/*
 * pct_oracle.h -- CPU restatement (plain C) of the reference env hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it.  The product
 * (online-3d-bpp-pct_amd/) never links, imports or falls back to it.
 *
 * Parity status: PINNED -- every part is checked bit for bit against the unmodified Python reference (imported from /root/reference
 * under tests/golden/ref_shim.py) by tests/golden/gen_golden.py, which writes the committed fixtures (the .npz files of tests/golden/) only
 * when the oracle equals the reference, and it reproduces the survey's known-answer hashes: both envs, settings
 * 1 / 2 / 3, all five leaf-node schemes, the dataset and NumPy-stream item sources, the heuristics.  The one place that needs a
 * qualifier is np.linalg.lstsq in the stability check (settings 1 / 3), which is not reference code but the LAPACK of the NumPy wheel:
 *   - pcto_set_lstsq_mode(0), the default (what the kernels run by default): a Jacobi SVD, equal to the reference up to the last
 *     bits of that solve (one on-domain env-run in 55 parts ways on such a bit, profiles/r04_lstsq_ondomain.txt);
 *   - pcto_set_lstsq_mode(1 / 2): dgelsd restated operation for operation with the arithmetic of OpenBLAS' AVX-512 / AVX2 kernel set
 *     (pct_oracle_gelsd.c), pinned routine by routine to the bundled library (tests/golden/check_gelsd_port.py) and against the
 *     reference on 11.5 M on-domain env-steps and the adversarial streams without a single difference
 *     (profiles/r04_lstsq_ondomain_gelsd.txt, r04_lstsq_ondomain_gelsd_large.txt, r04_lstsq_ondomain_avx2.txt, r04_gelsd_port.txt, r04_gelsd_other_numpy.txt).
 *
 * The batched API mirrors include/pct_env.h one to one (pcto_* instead of pct_*, host
 * pointers instead of device pointers, float64 observations like the gym env returns
 * before envs.py:180 casts them).
 */
#ifndef PCT_ORACLE_H
#define PCT_ORACLE_H

#include "../include/pct_env.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcto_env pcto_env;

int pcto_create(const pct_config* cfg, pcto_env** out);
int pcto_destroy(pcto_env* env);
const char* pcto_last_error(void);

int pcto_set_item_set(pcto_env* env, const int32_t* item_set, int32_t n);
int pcto_set_sample_bounds(pcto_env* env, int32_t left, int32_t right);
int pcto_set_item_stream(pcto_env* env, const int32_t* items, int64_t T);
int pcto_set_item_dataset(pcto_env* env, const int32_t* items, const int32_t* lengths, int32_t n_traj, int32_t max_len);
int pcto_set_sampler(pcto_env* env, uint64_t seed);
/* strict NumPy-stream mode (discrete env): env e consumes the MT19937 stream of np.random.seed(seed + env_id_base + e) */
int pcto_set_numpy_rng(pcto_env* env, uint32_t seed);
int pcto_set_numpy_item_count(pcto_env* env, int32_t n); /* continuous env: len(item_set) behind the unused randint draws */
int pcto_step_heuristic(pcto_env* env, int32_t kind, int32_t n_steps);
int pcto_set_density_stream(pcto_env* env, const double* den, int64_t T);
int pcto_set_dataset_density(pcto_env* env, const double* den);
int pcto_set_shuffle_seed(pcto_env* env, uint64_t seed);

/* outputs (host, owned by the handle) */
double* pcto_obs(pcto_env* env);      /* float64 [N,(I+L+1)*9] */
double* pcto_reward(pcto_env* env);   /* float64 [N] */
uint8_t* pcto_done(pcto_env* env);    /* [N] */
int32_t* pcto_info_counter(pcto_env* env);
double* pcto_info_ratio(pcto_env* env);
uint32_t* pcto_error_flags(pcto_env* env);
/* np.linalg.lstsq of the stability check (settings 1 / 3), process-wide.  0 (default): the one-sided Jacobi SVD that the kernels
 * run by default (pct_set_lstsq_mode(env, PCT_LSTSQ_JACOBI)); 1: LAPACK dgelsd operation for operation as the reference's NumPy
 * (2.2.6, OpenBLAS 0.3.29, AVX-512 kernel set) executes it -- pct_oracle_gelsd.c; the kernels' PCT_LSTSQ_GELSD */
void pcto_set_lstsq_mode(int mode);
int pcto_get_lstsq_mode(void);
/* the solve itself, for tests: row-major M x N system -> x[N], singular values sv[N], rank; returns dbdsqr's info */
int gelsd_lstsq(const double* A, const double* b, int M, int N, double* x, int* rank, double* sv, int* near_cut);
void pcto_set_ill_near(int on); /* analysis only: also note decisions within 1e-9 of a tie on stacks carrying a least-squares share */
int pcto_ill_commit(pcto_env* env, uint8_t* out);      /* [N] ... raised by a solve of a commit walk (PCT_FLAG_ILL_COMMIT) */
int pcto_ill_conditioned(pcto_env* env, uint8_t* out); /* [N] sticky notice of the stability settings (PCT_FLAG_ILL_CONDITIONED) */

int pcto_reset(pcto_env* env, const int32_t* env_ids, int32_t n);
/* auto_reset != 0: VecEnv worker semantics (shmem_vec_env.py:139-143) -- a done env is
 * reset inside the step and its observation is the reset observation.
 * auto_reset == 0: raw gym env semantics (bin3D.py:160-165). */
int pcto_step_rows(pcto_env* env, const double* rows, int32_t row_len, int32_t auto_reset);
int pcto_step_index(pcto_env* env, const int64_t* leaf_index, int32_t auto_reset);
int pcto_step_hash_policy(pcto_env* env, int32_t n_steps);

int pcto_debug_state(pcto_env* env, int32_t local_id, int32_t* heightmap, int32_t* ems,
                     int32_t cap_ems, int32_t* n_ems, int32_t* n_boxes, int32_t* next_item,
                     int64_t* draw_cursor);

/* number of threads used by the batched entry points (OpenMP if built with it, else 1) */
int pcto_num_threads(void);
void pcto_set_num_threads(int n);

/* CPython 3.10 `set` of 6-int tuples: insertion -> iteration order (Objects/setobject.c,
 * Objects/tupleobject.c tuplehash).  keys int64 [n,6]; writes the iteration order as
 * indices into `keys` (first occurrence of each distinct tuple); returns the count. */
int pcto_pyset_order(const int64_t* keys, int32_t n, int32_t* order_out);

#ifdef __cplusplus
}
#endif
#endif
