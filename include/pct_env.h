/*
 * pct_env.h -- C ABI of the MI355X-native batched PCT bin-packing environment.
 *
 * One handle (`pct_env`) owns N independent packing environments on ONE GPU and
 * advances all of them with hand-written HIP kernels (gfx950).  The handle replaces,
 * for the env hot path only, what the reference runs as N forked Python workers:
 *
 *   reference interface (file:line under the reference repo)        -> entry point here
 *   ---------------------------------------------------------------------------------------
 *   envs.py:75-116 make_vec_envs / envs.py:33-49 gym.make kwargs     -> pct_create(pct_config)
 *   givenData.py:4-14 item_size_set, binCreator.py:24-39             -> pct_set_item_set
 *   binCreator.py:41-72 LoadBoxCreator (scripted trajectories)       -> pct_set_item_stream
 *   binCreator.py:41-72 LoadBoxCreator on a dataset (README.md:75-77) -> pct_set_item_dataset
 *   binCreator.py:37-39 RandomBoxCreator (on-the-fly sampling)       -> pct_set_sampler (counter-keyed draws)
 *   envs.py:49 env.seed(seed + rank) -> bin3D.py:47-54 np.random.seed;
 *   binCreator.py:38 randint, bin3D.py:82-84 random, :114-115 shuffle -> pct_set_numpy_rng (the env's own MT19937
 *                                                                      stream, draw for draw)
 *   bin3D.py:75-84 next_den (setting 3)                              -> pct_set_density_stream /
 *                                                                      pct_set_dataset_density
 *   bin3D.py:114-115 np.random.shuffle(allPostion)                   -> pct_config.shuffle
 *   wrapper/vec_env.py:48-58 VecEnv.reset,
 *   wrapper/shmem_vec_env.py:61-68,112-117 reset / reset_specific    -> pct_reset
 *   wrapper/vec_env.py:60-88 step_async/step_wait,
 *   wrapper/shmem_vec_env.py:70-82,139-143 (auto-reset on done),
 *   pct_envs/PctDiscrete0/bin3D.py:151-188 PackingDiscrete.step      -> pct_step_rows
 *   train_tools.py:66-67 leaf_nodes[batchX, idx] gather              -> pct_step_index
 *   (benchmark-only stand-in policy)                 -> pct_step_hash_policy
 *   heuristic.py:11-569 (the seven baselines' placement rules)       -> pct_step_heuristic
 *   envs.py:178-182 VecPyTorch.step_wait outputs                     -> pct_bind_outputs /
 *                                                                      pct_obs, pct_reward, ...
 *   storage.py:33-39 PCTRolloutStorage.insert, train_tools.py:70     -> pct_bind_rollout_slot (the transition kernel
 *                                                                      writes obs[t+1], rewards[t], masks[t+1])
 *   bin3D.py:163-164,186-187 info dict                               -> pct_info_counter/ratio
 *   wrapper/vec_env.py:90-99 close                                   -> pct_destroy
 *
 * Conventions
 *   - every entry point returns an int status (PCT_OK == 0); pct_last_error() gives text;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); entry points
 *     only enqueue work, they never synchronise the device;
 *   - all device pointers returned by getters are owned by the handle (or by the caller
 *     if bound with pct_bind_outputs) and their contents are valid after the enqueued
 *     step completes and until the next step/reset on the same handle;
 *   - a GPU kernel cannot raise: per-env sticky `error_flags` record what the reference
 *     would have raised as a Python exception (see PCT_ERR_*); an env that hits one is
 *     force-terminated (done=1) and auto-reset like any finished episode.
 *
 * Observation layout (bin3D.py:70-93, tools.py:70-105): per env (I+L+1) rows x 9 float32
 *   rows 0..I-1    internal nodes [lx,ly,lz,lx+x,ly+y,lz+z,density,0,1]
 *   rows I..I+L-1  leaf nodes     [xs,ys,zs,xe,ye,H,0,0,1]   (first <=L feasible, list order)
 *   row  I+L       next item      [density,0,0,s0,s1,s2,0,0,1] (sizes sorted ascending)
 */
#ifndef PCT_ENV_H
#define PCT_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCT_ABI_VERSION 1

/* status codes */
#define PCT_OK 0
#define PCT_ERR_INVALID_ARG 1
#define PCT_ERR_UNSUPPORTED 2
#define PCT_ERR_HIP 3
#define PCT_ERR_NO_DEVICE 4
#define PCT_ERR_STATE 5

/* env_kind */
#define PCT_ENV_DISCRETE 0   /* pct_envs/PctDiscrete0 */
#define PCT_ENV_CONTINUOUS 1 /* pct_envs/PctContinuous0 */

/* leaf-node expansion scheme (tools.py:133 --lnes) */
#define PCT_LNES_EMS 0
#define PCT_LNES_EV 1 /* event points (D/space.py:613-693; static under the reference's step) */
#define PCT_LNES_EP 2 /* extreme points (D/space.py:696-750, PctTools.py:114-136) */
#define PCT_LNES_CP 3
#define PCT_LNES_FC 4 /* full coordinate space (D/space.py:573-610) */

/* heuristic baselines (heuristic.py) usable as in-env policies: pct_step_heuristic */
#define PCT_HEUR_LSAH 0 /* heuristic.py:138-226 LASH */
#define PCT_HEUR_HM 1   /* :232-298 heightmap_min */
#define PCT_HEUR_OBPH 2 /* :364-425 OnlineBPH */
#define PCT_HEUR_DBL 3  /* :431-498 DBL */
#define PCT_HEUR_BR 4   /* :500-569 BR */
#define PCT_HEUR_MACS 5 /* :11-136 MACS */
#define PCT_HEUR_RANDOM 6 /* :300-362 random; np.random.randint(0, n) -> pct_mix32(global env id, t) % n */

/* item source */
#define PCT_ITEMS_NONE 0
#define PCT_ITEMS_STREAM 1  /* scripted per-env trajectories (parity runs) */
#define PCT_ITEMS_SAMPLER 2 /* counter-based on-device sampler (training / bench) */
#define PCT_ITEMS_DATASET 3 /* the reference's dataset trajectories (binCreator.py:41-72) */

/* per-env sticky error flags (bitmask, uint32) */
#define PCT_FLAG_INTERNAL_OVERFLOW 0x1u  /* packed boxes >= internal_node_holder: reference
                                            raises IndexError at space.py:385 */
#define PCT_FLAG_EMS_OVERFLOW 0x2u       /* EMS list exceeded ems_capacity */
#define PCT_FLAG_CANDIDATE_OVERFLOW 0x4u /* leaf-candidate set exceeded candidate_capacity */
#define PCT_FLAG_STABILITY_OVERFLOW 0x10u /* stability check (settings 1 / 3): the share / polygon pools, the hull
                                             workspace or the walk queue were exceeded in the retry pass too (the
                                             normal pass requeues such an env, state untouched), or a box rests on
                                             more than 25 supporters none of which holds its centre of mass.  25 is
                                             LAPACK's own limit for the path of dgelsd the library restates (dlalsd:
                                             n <= SMLSIZ = 25 -> dlasdq; beyond it dgelsd runs the divide-and-conquer
                                             dlasda / dlalsa, which is not restated) and covers every stack of the 10^3
                                             / items 1..5 domain (a 5 x 5 footprint on unit tiles).  The normal pass
                                             takes 8, the retry pass 25 (16 on handles whose retry pools -- hundreds of
                                             internal nodes -- leave no 73 KB of LDS for the 301 x 25 system) */
#define PCT_FLAG_ILL_CONDITIONED 0x40u   /* NON-FATAL notice (the env is not terminated, PctVecEnv(strict=True) does not
                                            raise): a >= 3-supporter load split (np.linalg.lstsq, space.py:134-163)
                                            took its rank decision within a factor 1000 of the rcond cut.  In the DEFAULT
                                            mode (PCT_LSTSQ_GELSD since round 5: the split is solved exactly as the
                                            reference's NumPy solves it -- LAPACK dgelsd with OpenBLAS' kernels) the
                                            library follows the reference through such a decision (tests/golden
                                            discrete_s1_flat_diverging; 0 of 6 000 on-domain env-runs part ways) and the
                                            notice is informational: the reference's own verdict at such a step differs
                                            between NumPy builds (AVX-512 / AVX2 kernel sets: PCT_LSTSQ_GELSD_AVX2,
                                            profiles/r04_lstsq_ondomain.txt).  In the opt-in PCT_LSTSQ_JACOBI mode (the
                                            default of rounds 1-4) a run may part ways with the reference there, and
                                            also on the last bit of a well-conditioned solve that decides an exactly
                                            degenerate point-in-polygon test (convex_hull.py:104-105; one step in 10^5
                                            of discrete setting 1), which carries no notice */
#define PCT_FLAG_ILL_COMMIT 0x80u        /* NON-FATAL, provenance of the notice above: it was raised (also) by a solve of a COMMIT walk
                                            (calculated_impact, space.py:73-164 -- the placement's own, state-changing walk, whose
                                            solves every evaluation order makes).  A notice WITHOUT this bit came from a candidate's
                                            virtual check (space.py:166-267) only: the reference walks a candidate's supporters one
                                            after the other and stops at the first unstable one, the kernels examine them side by
                                            side, so which solves of a candidate that fails anyway are made at all differs between
                                            the two -- the commit part of the notice is the part that is comparable */
#define PCT_FLAG_DATASET_EXHAUSTED 0x20u /* LoadBoxCreator ran past its last trajectory: the
                                            reference raises IndexError at binCreator.py:58 */
#define PCT_FLAG_BAD_ACTION 0x8u         /* malformed action: reference raises ValueError at
                                            bin3D.py:144-145 (list.remove) or in np.max of an
                                            empty slice (space.py:354-355) */


/* Lattice: all geometry is integral in "lattice units".  Discrete env: 1 unit = 1.
 * Continuous env: 1 unit = 1e-3 (item sizes are round(U(a,b),3), bin3D.py:106-108). */
typedef struct pct_config {
  int32_t struct_size;          /* = sizeof(pct_config), ABI guard */
  int32_t env_kind;             /* PCT_ENV_* */
  int32_t setting;              /* 1, 2 or 3 (tools.py:132) */
  int32_t num_envs;             /* envs owned by this handle (this GPU's shard) */
  int32_t container[3];         /* W, Ly, H in lattice units (envs.py:35) */
  int32_t internal_node_holder; /* I (tools.py:173) */
  int32_t leaf_node_holder;     /* L (tools.py:174) */
  int32_t lnes;                 /* PCT_LNES_* */
  int32_t env_id_base;          /* global id of local env 0 (multi-GPU sharding: env e of
                                   the job lives on rank e / num_envs) */
  int32_t ems_capacity;         /* EMS kept per env after elimination (the LDS list of the normal pass); 0 =
                                   default (128 for bins up to 12 per axis, else 256 discrete / 768 continuous) */
  int32_t candidate_capacity;   /* hash-table slots for the leaf-candidate set (LDS table of the normal pass);
                                   0 = default (pct_env.hip pct_create).  Bins up to 12 per axis: 2048 (setting 2) /
                                   512 (settings 1 / 3: two orientations).  Larger bins: discrete 8192 (setting 2)
                                   / 2048 (settings 1 / 3); continuous setting 2 8192 -- one 32 KB LDS region that
                                   the 2048-slot table grows into, its old entries parked in an HBM row meanwhile --
                                   and continuous settings 1 / 3 32768 in HBM.  Always 8 * 4^k.  An explicit
                                   capacity above 8192 puts the continuous table in HBM.
                                   An env that outgrows either list is NOT terminated: it is handed, state untouched,
                                   to a large-capacity retry pass enqueued right behind the normal one (setting 2:
                                   discrete 4x the EMS list and, LDS permitting, 4x the table; continuous a 32768-slot
                                   table in HBM -- with the 8192-slot LDS default every env beyond 4915 candidates
                                   goes that way).  Only what outgrows the retry pass too raises
                                   PCT_FLAG_EMS_OVERFLOW / PCT_FLAG_CANDIDATE_OVERFLOW. */
  int32_t shuffle;              /* 1: permute the candidate list before the first-L cut
                                   (bin3D.py:114-115 `--shuffle`); see pct_shuffle_priority */
  int32_t reserved[3];          /* [0]: PCT_OVERFLOW_RETRY_* (discrete env); others 0 */
} pct_config;
#define PCT_OVERFLOW_RETRY_ON 0  /* default: the retry pass is enqueued with every transition (a 16-block kernel that
                                    exits at once when no env overflowed: about 5 us per step on MI355X, profiles/r04_experiments.txt) */
#define PCT_OVERFLOW_RETRY_OFF 1 /* no retry pass: an overflow raises its flag and terminates the env */

typedef struct pct_env pct_env;

/* ---- lifetime ---------------------------------------------------------------------- */
int pct_abi_version(void);
const char* pct_last_error(void);
int pct_create(const pct_config* cfg, int device, pct_env** out);
int pct_destroy(pct_env* env);

/* ---- item sources ------------------------------------------------------------------ */
/* item_set: host int32 [n,3] in lattice units (givenData.py:10-14).  Also fixes
 * low_bound = min over all entries (bin3D.py:23). */
int pct_set_item_set(pct_env* env, const int32_t* item_set, int32_t n);
/* Continuous env (PCT_ENV_CONTINUOUS): pct_config.container is given in lattice units
 * (1e-3; the reference's integer bin sizes are multiples of 1000) and items are 3-decimal
 * sizes (C/bin3D.py:106-108).  Sampler bounds in lattice units (tools.py:178-181);
 * low_bound = left (C/bin3D.py:25-27).  The c-th draw of global env g is
 * left + pct_pick(seed, g, 3c+d, right-left+1) for d = 0,1,2; under settings 1 and 3 the third size is
 * 100 * (1 + pct_pick(seed, g, 3c+2, 5)) instead -- np.random.choice([0.1,...,0.5]), C/bin3D.py:110-112.
 * A float32 action row is
 * matched back to the env's current leaf whose float32 cast it is and decoded from that
 * leaf's float64 values, i.e. exactly like the reference decodes the float64 row
 * (round(.,6), C/bin3D.py:153-173); a row matching no leaf is decoded from the widened
 * floats.  pct_step_index carries the leaf in full precision by construction. */
int pct_set_sample_bounds(pct_env* env, int32_t left, int32_t right);
/* Scripted trajectories: host int32 [num_envs, T, 3]; env e draws items[e][c % T] for its
 * c-th draw (one draw per reset and one per successful placement,
 * bin3D.py:61-67,181-182).  Copied to the device. */
int pct_set_item_stream(pct_env* env, const int32_t* items, int64_t T);
/* Dataset trajectories with LoadBoxCreator semantics (binCreator.py:41-72): host int32
 * [n_traj, max_len, 3] + lengths [n_traj].  Every reset moves the env to the NEXT trajectory --
 * the first episode plays trajectory 1, not 0 (:54-55) -- an exhausted trajectory is followed
 * by the sentinel (100,100,100) (:62, in the units of `items`) and then (10,10,10) for ever
 * (:69-72).  All envs walk the same trajectory sequence, like the reference's workers. */
int pct_set_item_dataset(pct_env* env, const int32_t* items, const int32_t* lengths, int32_t n_traj,
                         int32_t max_len);
/* Setting 3 (random item densities, D/bin3D.py:76,80-84; box mass = volume x density in the
 * stability check, space.py:38).  The reference draws np.random.random() at every
 * cur_observation(); here the density of an env's c-th observation (c counts every observation
 * the env has produced, including the one of a terminal step that the VecEnv worker discards) is
 *   - den[e, c % T] after pct_set_density_stream (host float64 [N,T], scripted), else
 *   - pct_density(seed, global env id, c), the counter-based uniform (0,1) draw below;
 * under pct_set_item_dataset it is the fourth column of the previewed dataset item
 * (bin3D.py:76 `next_box[3]`): pct_set_dataset_density takes it as host float64
 * [n_traj, max_len], same indexing as `items`; the sentinel items carry density 1 (the
 * reference raises IndexError on them).  Ignored unless pct_config.setting == 3. */
int pct_set_density_stream(pct_env* env, const double* den, int64_t T);
int pct_set_dataset_density(pct_env* env, const double* den);
/* Counter-based sampler: the c-th draw of global env g is
 * item_set[pct_pick(seed, g, c, n)] (discrete) -- see pct_pick below. */
int pct_set_sampler(pct_env* env, uint64_t seed);

/* Strict NumPy-stream mode (the discrete env with any leaf expansion -- bin3D.py:114-115 shuffles whatever --lnes
 * produced -- and any bin size, or the continuous env -- see pct_set_numpy_item_count below; before the first reset).
 * Env e then
 * consumes the MT19937 stream that np.random.seed(seed + env_id_base + e) starts, exactly as the env's worker process
 * does under ShmemVecEnv(fork) (envs.py:49 env.seed(seed + rank); bin3D.py:47-54): the item is
 * item_set[np.random.randint(0, n)] (binCreator.py:37-39), the setting-3 density np.random.random() redrawn while 0
 * (bin3D.py:82-84), and -- with pct_config.shuffle -- the candidate list goes through np.random.shuffle
 * (bin3D.py:114-115; Fisher-Yates from the back on random_interval draws), including the extra shuffle and density
 * draw of the discarded observation of a failed step (bin3D.py:165).  Given the same seed and actions the env then
 * reproduces the reference's trajectory bit for bit with the CLI's defaults (tools.py:136 shuffle=True,
 * RandomBoxCreator).  The heuristic policies are not available in this mode. */
int pct_set_numpy_rng(pct_env* env, uint32_t seed);

/* The same for the continuous env in its sampling mode (pct_set_sample_bounds; C/bin3D.py:14-16
 * sample_from_distribution=True, the CLI's --continuous default): every observation -- the discarded one of a failed
 * step included -- draws its item as round(np.random.uniform(a, b), 3) per edge (C/bin3D.py:103-113; settings 1/3:
 * two edges and np.random.choice of five heights), then the setting-3 density, then np.random.shuffle of the candidate
 * positions; reset() and every successful step also spend the RandomBoxCreator's randint over the item set nobody
 * reads (C/bin3D.py:73,202) -- `n` is that set's length (default 125 = len(givenData.item_size_set)). */
int pct_set_numpy_item_count(pct_env* env, int32_t n);

/* Seed of the shuffle permutation (default 0). */
int pct_set_shuffle_seed(pct_env* env, uint64_t seed);

/* Which solver stands behind np.linalg.lstsq in the stability check (settings 1 / 3: a stack split over three and more
 * supporters none of which holds its centre of mass, D/space.py:134-163, :236-259; C/space.py:130-159, :232-255).
 *   PCT_LSTSQ_GELSD (DEFAULT since round 5): LAPACK dgelsd operation for operation AS THE REFERENCE'S NUMPY EXECUTES IT (NumPy 2.2.6 =
 *     OpenBLAS 0.3.29 / LAPACK 3.11 with the kernel set OpenBLAS selects on AVX-512 hosts: dgeqr2, dgebd2, dbdsqr ..., the fused /
 *     split sums of its dgemv / dger / drot kernels and the 80-bit x87 dnrm2; csrc/pct_gelsd.cuh).  Bit-identical solutions: the 13
 *     on-domain and 17 adversarial env-runs that part ways under PCT_LSTSQ_JACOBI follow the reference to the end
 *     (profiles/r04_gelsd_port.txt).  A group of 4 / 8 / 16 lanes solves a system (up to 4 / 8 / 16 supporters).
 *   PCT_LSTSQ_JACOBI (the default of rounds 1-4; ~1.35 x the throughput on the stability settings): a one-sided Jacobi SVD -- the
 *     same minimum-norm solution as the reference's, equal to it up to the last bits; on the reference's own item domain one
 *     env-run in 55 (2000 steps each) parts ways with the reference through such a bit (see PCT_FLAG_ILL_CONDITIONED).
 *   PCT_LSTSQ_GELSD_AVX2: the same with the kernel set OpenBLAS selects on AVX2 hosts without AVX-512 ("Haswell"; also what
 *     OPENBLAS_CORETYPE=ZEN runs) -- its
 *     dgemv 'N', daxpy and dgemm kernels sum differently and its ddot does not fuse (np.dot of the 2-vectors at D/space.py:114-115,
 *     143-145 is x0 y0 + x1 y1 there, also in the two-supporter lever rule); np.linalg.lstsq then returns other last bits on 98 % of these systems,
 *     and a reference run on such a host follows another trajectory at a tie (about one step in 10^5 on the discrete env).  Choose the
 *     flavour of the machine the reference ran on.
 * Callable at any time between steps; applies to the normal and the retry pass.  Replaces nothing in the reference's
 * interface: it names the LAPACK the reference inherits from its NumPy wheel. */
#define PCT_LSTSQ_JACOBI 0
#define PCT_LSTSQ_GELSD 1
#define PCT_LSTSQ_GELSD_AVX2 2
int pct_set_lstsq_mode(pct_env* env, int32_t mode);

/* ---- outputs ------------------------------------------------------------------------ */
/* Bind caller-owned device buffers (e.g. torch tensors).  Any pointer may be NULL to keep
 * the handle-owned buffer.  The observation buffer belongs to the env between steps: a step only
 * rewrites the rows that changed (the placed box's row, the leaf rows, the next-item row; every row
 * when an episode ends), so the caller must treat it as read-only; binding a buffer makes the
 * next launch rewrite it whole.  obs float32 [N,(I+L+1)*9]; reward float32 [N]; done uint8 [N];
 * counter int32 [N]; ratio float64 [N]; error_flags uint32 [N] (must be zero-filled). */
int pct_bind_outputs(pct_env* env, float* obs, float* reward, uint8_t* done,
                     int32_t* counter, double* ratio, uint32_t* error_flags);
/* Device-resident rollout edge (storage.py:33-39 PCTRolloutStorage.insert fused into the transition): the
 * NEXT launch writes its observation -- every row -- into `obs_next` (float32 [N,(I+L+1)*9], e.g.
 * rollout.obs[t+1]), its reward into `reward` (float32 [N], e.g. rollout.rewards[t]; NULL keeps the current
 * buffer) and 1 - done into `mask` (float32 [N], e.g. rollout.masks[t+1]; NULL: not written).  The binding
 * stays until the next pct_bind_rollout_slot / pct_bind_outputs; the caller re-binds once per step. */
int pct_bind_rollout_slot(pct_env* env, float* obs_next, float* reward, float* mask);
float* pct_obs(pct_env* env);
float* pct_reward(pct_env* env);
uint8_t* pct_done(pct_env* env);
int32_t* pct_info_counter(pct_env* env);
double* pct_info_ratio(pct_env* env);
uint32_t* pct_error_flags(pct_env* env);
int32_t pct_obs_row_len(pct_env* env); /* (I+L+1)*9 */

/* ---- transitions -------------------------------------------------------------------- */
/* Reset all envs (env_ids == NULL) or the listed local ids (device int32 [n]). */
int pct_reset(pct_env* env, const int32_t* env_ids, int32_t n, void* stream);
/* One batched step from leaf rows: device float32 [N,row_len], row_len in {9,6,3}
 * (bin3D.py:152-153: 3 = (flag,lx,ly) heuristic form). */
int pct_step_rows(pct_env* env, const float* rows, int32_t row_len, void* stream);
/* One batched step from leaf indices: device int64 [N], index into the env's current
 * leaf rows (the gather of train_tools.py:66 fused into the step). */
int pct_step_index(pct_env* env, const int64_t* leaf_index, void* stream);
/* n_steps batched steps with the stand-in policy leaf = pct_mix32(g, t) % k over the k
 * valid leaves (leaf 0 if k == 0), t = the env's lifetime step counter. */
int pct_step_hash_policy(pct_env* env, int32_t n_steps, void* stream);
/* n_steps batched steps with a heuristic baseline of heuristic.py as the in-env policy (kind =
 * PCT_HEUR_*; LNES = EMS; the discrete env takes all seven, the continuous env LSAH / OnlineBPH / BR -- the three
 * tools.py:217-218 allows on PackingContinuous; not in strict NumPy-stream mode): the placement rule reads the env's
 * heightmap / EMS list / stability state, the chosen placement is stepped exactly as `env.next_box = [x,y,z];
 * env.step([0,lx,ly])`; an env whose heuristic finds no placement ends its episode WITHOUT a
 * step (done = 1, reward = 0, counter / ratio as at a failed step) and is reset, as the reference
 * loops do (e.g. heuristic.py:241-249,291-296). */
int pct_step_heuristic(pct_env* env, int32_t kind, int32_t n_steps, void* stream);

/* The stand-in policy as its own kernel (what a policy network would do between two
 * steps): reads each env's leaf-mask column from the observation, picks
 * leaf = pct_mix32(g, t) % k and writes that leaf row to rows_out (device float32 [N,9]),
 * ready for pct_step_rows. */
int pct_policy_hash_rows(pct_env* env, float* rows_out, void* stream);
/* ... and as a leaf INDEX (device int64 [N]), ready for pct_step_index: the form the reference's policy samples
 * (train_tools.py:63-66: `selected_leaf_node` is the leaf row gathered by this index).  One launch; index_out may be the
 * rollout's actions[t] slot (storage.py:8).  Reads the observation buffer currently bound (pct_bind_outputs /
 * pct_bind_rollout_slot). */
int pct_policy_hash_index(pct_env* env, int64_t* index_out, void* stream);
/* The same stand-in policy as an EPILOGUE of every following launch (reset / step_*): the transition kernel, having
 * written an env's new observation, also writes the row pct_policy_hash_rows would gather from it to rows_out (device
 * float32 [N,9]; the same bytes) -- a benchmark loop `pct_step_rows(rows)` then needs no policy dispatch between two
 * transitions (the rows still travel through HBM, one launch per step).  NULL switches it off. */
int pct_bind_policy_rows(pct_env* env, float* rows_out);

/* ---- kernel timing ------------------------------------------------------------------- */
/* When enabled, the transition kernel of every launch (reset / step_*) carries a pair of hipEvents that bracket
 * exactly that dispatch (hipExtLaunchKernel start / stop events: the kernel's own begin and end timestamps, no marker
 * packets on the stream) -- the step kernel itself, not the small large-capacity retry pass that follows it.
 * pct_profile_read synchronises on the recorded events, returns the number of TIMED launches and their summed duration
 * since the last read, and clears the accumulator.  `on` = K > 1: only every K-th launch carries a pair (a dispatch with
 * events costs ~7 us of its own on ROCm 7.2 / MI355X: time every launch when single launches matter, sample when the
 * average does). */
int pct_profile_enable(pct_env* env, int32_t on);
int pct_profile_read(pct_env* env, int64_t* n_launches, double* total_ms);

/* ---- introspection (tests / debugging) ---------------------------------------------- */
/* Copies env `local_id`'s geometric state to host: heightmap int32 [A*A] (A = max(W,Ly)),
 * EMS int32 [*n_ems,6] (<= cap_ems rows written), counters.  Synchronises the device. */
int pct_debug_state(pct_env* env, int32_t local_id, int32_t* heightmap, int32_t* ems,
                    int32_t cap_ems, int32_t* n_ems, int32_t* n_boxes, int32_t* next_item,
                    int64_t* draw_cursor);

/* ---- shared arithmetic: part of the ABI, used identically on host and device -------- */
/* Continuous env: EMS float64 [*n_ems,6] (row-major copy), next item in bin units. */
int pct_debug_state_f64(pct_env* env, int32_t local_id, double* ems, int32_t cap_ems, int32_t* n_ems,
                        int32_t* n_boxes, double* next_item, int64_t* draw_cursor);

/* The heavy-first dispatch's work keys, host uint32 [N]: shader-clock cycles of every env's last step / 256 << 12 | its live
 * EMS count (0 right after a reset).  Synchronises the device. */
int pct_debug_work_keys(pct_env* env, uint32_t* host_out);
/* The large-capacity retry pass: how many envs the LAST launch queued for it, how many envs it has re-run since
 * pct_create and in how many launches it found work (any pointer may be NULL).  Synchronises the device. */
int pct_debug_retry_count(pct_env* env, int32_t* last, int64_t* envs_total, int64_t* launches_total);

/* Per-phase cycle accounting of the transition kernel (profiling aid).  on != 0: (re)start
 * accumulation; host_out, if non-NULL, first receives the accumulators gathered so far:
 * uint64 [N, pct_debug_timing_slots()] (44) = s_memtime cycles in {load, drop_box, GENEMS, candidate set, feasibility,
 * observation write, state store}, the number of steps, then the candidate-set detail
 * {generation + membership probes, batch de-duplication, matching, rebuilds}, the set's statistics
 * (slots 12..29) and the stability settings' counters (slots 30..43, csrc/pct_set.cuh).
 * Synchronises the device. */
int pct_debug_phase_timing(pct_env* env, int32_t on, uint64_t* host_out);
/* uint64 words per env that pct_debug_phase_timing writes (44 since round 5, 32 before): size host_out by THIS, not by a
 * constant compiled into the caller.  No reference counterpart (profiling aid). */
int32_t pct_debug_timing_slots(void);

#if defined(__HIPCC__)
#define PCT_INLINE static inline __host__ __device__
#else
#define PCT_INLINE static inline
#endif
PCT_INLINE uint64_t pct_mix64(uint64_t seed, uint64_t env_global_id, uint64_t counter) {
  /* splitmix64 finaliser over a 3-word counter; stateless, so any (env, draw) is O(1) */
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (env_global_id + 1) +
               0xD1B54A32D192ED03ull * (counter + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
/* shuffle=True.  The reference permutes the candidate list with the process-global NumPy
 * MT19937 (np.random.shuffle), which is shared with item sampling and cannot be reproduced
 * per env.  Here candidate i of the list (set iteration order) gets this priority and the list
 * is visited in ascending (priority, i) order: a uniform random permutation keyed by (seed,
 * global env id, observation counter).  The oracle implements the same rule, so the HIP path is
 * still checked bit for bit; against the reference the comparison is distributional. */
/* uniform index in [0, n): floor(u * n), u = the top 32 bits of pct_mix64 as a fraction of 2^32
 * (a multiply instead of a 64-bit modulo on the device) */
PCT_INLINE uint32_t pct_pick(uint64_t seed, uint64_t env_global_id, uint64_t counter, uint32_t n) {
  return (uint32_t)(((pct_mix64(seed, env_global_id, counter) >> 32) * (uint64_t)n) >> 32);
}
PCT_INLINE uint32_t pct_shuffle_priority(uint64_t seed, uint64_t env_global_id, uint64_t obs_counter, uint32_t i) {
  return (uint32_t)(pct_mix64(seed ^ 0x5BD1E9955BD1E995ull, env_global_id, (obs_counter << 20) | (uint64_t)i) >> 32);
}
/* setting 3: density of observation `obs_counter` -- uniform on (0,1), k * 2^-53 with k >= 1
 * (np.random.random() redrawn while it is 0, bin3D.py:82-84) */
PCT_INLINE double pct_density(uint64_t seed, uint64_t env_global_id, uint64_t obs_counter) {
  uint64_t k = pct_mix64(seed ^ 0xD6E8FEB86659FD93ull, env_global_id, obs_counter) >> 11;
  if (k == 0) k = 1;
  return (double)k * (1.0 / 9007199254740992.0);
}
PCT_INLINE uint32_t pct_mix32(uint32_t env_global_id, uint32_t t) {
  uint32_t h = env_global_id * 0x9E3779B1u + t * 0x85EBCA77u + 0xC2B2AE3Du;
  h ^= h >> 16; h *= 0x7FEB352Du;
  h ^= h >> 15; h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}

#ifdef __cplusplus
}
#endif
#endif /* PCT_ENV_H */
