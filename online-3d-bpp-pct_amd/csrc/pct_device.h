// pct_device.h -- parameter blocks shared by the kernels and the C-ABI host code.
#ifndef PCT_DEVICE_H
#define PCT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "pct_stab.cuh"

#define PCT_SCALARS 16 /* int32 words of per-env scalar state (word 15: stability pools in use, entries | vertices << 16) */

#ifndef PCT_KERNARG_PTR
#define PCT_KERNARG_PTR 1 /* 1: the kernels read their parameter block from the kernarg segment at the point of use */
#endif

namespace pct {

#if defined(__HIPCC__)
// A kernel's parameter block as it lies in the kernarg segment (constant address space: uniform scalar loads).
template <typename P>
using PctConstParams = const __attribute__((address_space(4))) P*;
// An opaque copy of the pointer: loads through the result cannot be merged with, or hoisted above, the ones before the
// fence -- a cold phase re-reads the handful of words it needs instead of keeping them live (or spilled) across the hot
// loops.  The pointer stays in scalar registers and in the constant address space.
template <typename P>
__device__ __forceinline__ PctConstParams<P> pct_param_fence(PctConstParams<P> q) {
  asm volatile("" : "+s"(q));
  return q;
}
template <typename P>
__device__ __forceinline__ const P& pct_param_refresh(const P& p) {
#if PCT_KERNARG_PTR
  return *(const P*)pct_param_fence((PctConstParams<P>)(&p));
#else
  return p;
#endif
}
#endif

// HBM layout of the discrete env state: struct-of-arrays over envs, every array holds one
// contiguous slice per env (a wave reads its env's slice with consecutive lanes on
// consecutive addresses).
struct DiscreteParams {
  // geometry / sizes
  int N, W, Ly, H, A, AA; /* AA = A*A rounded up to a multiple of 8 */
  int I, L, row_len;
  int setting, low_bound;
  int lnes; /* PCT_LNES_EMS / PCT_LNES_CP / PCT_LNES_FC */
  int shuffle;
  int full_obs; /* 1: the observation buffer was (re)bound since the last launch -- rewrite every row */
  unsigned long long shuffle_seed;
  int ems_cap, cand_cap; /* capacities of THIS launch's LDS lists (the retry pass has larger ones) */
  int ems_stride;        /* row stride of the HBM EMS state = the largest ems_cap of any pass */
  int retry_mode;        /* != 0: this launch is the large-capacity retry pass (value = offset, +1 / -1, of the
                            other counter of the ping-pong pair, which it zeroes for the next step) */
  int* retry_count;      /* this step's counter of envs queued by the normal pass, or null: no retry pass */
  int* retry_ids;        /* [N] */
  /* Round 6: the retry pass INSIDE the launch (pct_discrete_tail_kernel, the plain setting-2 kernels): tail_blocks extra workgroups
   * behind the N of the normal pass wait until those have all left (tail_done: 64 sub-counters, workgroup b adds to b mod 64), then
   * re-run the queued envs with the retry pass's capacities, their lists in a per-workgroup row of HBM (tail_scratch) instead of LDS.
   * One dispatch per step instead of two: the dependent-dispatch gap of the (normally idle) retry kernel was 5 us of C2's 61. */
  int tail_blocks;             /* 0: no in-launch tail (a separate retry dispatch follows) */
  int tail_rm;                 /* +64 / -64: offset of the OTHER step's sub-counters in the ping-pong pair (the tail zeroes them) */
  int* tail_done;              /* [64] this step's sub-counters of finished normal-pass workgroups */
  unsigned char* tail_scratch; /* [tail_blocks][tail_scratch_bytes] */
  int tail_scratch_bytes;
  const DiscreteParams* tail_q; /* HOST pointer (never read on the device): the retry pass's parameter block, handed to the tail kernel by value */
  int key_bytes; /* 4: six 5-bit coords (bins <= 31); 8: six 10-bit coords (<= 1023) */
  // item source
  int source, n_items, env_id_base;
  int prio_t[3];     /* wave_priority thresholds on the EMS count (0: off) */
  /* Heavy-first dispatch (pct_env.hip, pct_order_kernel): when a launch holds more envs than the chip keeps resident,
   * workgroup b steps env order[b] -- the envs sorted by their work key, the shader-clock cycles their previous step
   * took, longest first -- so that no long step starts last (any bijection is a correct placement).  null: b steps
   * env b.  The keys live behind the scalars (work_key_slot): no pointer of their own is kept live in the kernels, and
   * the step's start time waits in the first 16 bytes of the workgroup's LDS (PCT_LDS_STASH). */
  const int32_t* order; /* [N] this launch's workgroup -> env map, or null */
  int rng_numpy;     /* 1: strict NumPy-stream mode -- item picks, setting-3 densities and the candidate shuffle consume
                        the env's own MT19937 stream exactly as the reference's worker process does */
  uint32_t* mt;      /* [N,624] MT19937 state words of every env (position: scalars[7]) */
  double* mt_den;    /* [N] density drawn for the current observation (setting 3) */
  long long T;
  unsigned long long seed;
  const int32_t* item_set; /* [n_items,3] */
  const int32_t* stream;   /* [N,T,3]; dataset mode: [n_traj,max_len,3] */
  const int32_t* ds_len;   /* dataset mode: [n_traj] */
  int ds_ntraj, ds_maxlen;
  // persistent state
  int16_t* hmap;    /* [N,AA] heightmap */
  void* ems;        /* [N,ems_cap] packed EMS (key words) */
  void* boxes;      /* [N,I] packed placed boxes, placement order */
  void* leaves;     /* [N,L] packed current leaf nodes */
  int32_t* scalars; /* [N,PCT_SCALARS]: n_ems,n_boxes,n_leaf,item[3],t,-,cursor lo/hi,vol lo/hi */
  uint32_t* flags;  /* [N] sticky PCT_FLAG_* */
  unsigned long long* timing; /* [N,16] per-phase cycle accumulators, or null */
  uint32_t* set_scratch;      /* [grid, 768] insertion-order lists of the candidate sets that reach the 2048-slot table */
  // stability state (settings 1/3 only): compact pooled layout, LDS-resident during a transition (csrc/pct_stab.cuh)
  StabHbm sb;
  // setting 3 density source (include/pct_env.h pct_set_density_stream / pct_set_dataset_density)
  const double* den_stream; /* [N,den_T] or null */
  long long den_T;
  const double* ds_den;     /* dataset mode: [n_traj,max_len] or null */
  // outputs
  float* obs;       /* [N,row_len] */
  float* reward;    /* [N] */
  uint8_t* done;    /* [N] */
  int32_t* counter; /* [N] */
  double* ratio;    /* [N] */
  float* mask;      /* [N] 1 - done as float32 (storage.py masks), or null */
  float* policy_rows; /* [N,9] stand-in policy epilogue (pct_bind_policy_rows): the leaf row pct_policy_hash_rows would gather
                         from the observation this launch writes, or null */
  int* retry_total; /* [2] envs the retry pass has ever re-run, launches in which it found work (pct_debug_retry_count) */
  /* host-side launch options, never read by a kernel: events that bracket exactly this dispatch (hipExtLaunchKernel) */
  void* launch_ev_start;
  void* launch_ev_stop;
};

// HBM layout of the continuous env state (float64, SoA over envs; within an env every
// coordinate column is contiguous: ems[c][i], boxes[c][i], leaves[c][i]).
#define PCT_PARK_WORDS 1280 /* a 2048-slot table grows at 1229 entries (fill * 5 >= 2047 * 3) */
struct ContinuousParams {
  int N, I, L, row_len, setting;
  int full_obs; /* as in DiscreteParams */
  double W, Ly, H; /* container (integral values, as the reference's int64 plain_size) */
  double low_bound; /* C/bin3D.py:25-29 size_minimum */
  int shuffle;
  unsigned long long shuffle_seed;
  int ems_stride; /* row stride of the HBM EMS state = the retry pass's ems_cap */
  int ems_cap, cand_cap, order_cap, union_words; /* union_words: LDS words shared by the hash table and the GENEMS children */
  int source, env_id_base;
  int sample_left, sample_right; /* lattice 1e-3; right <= 0: items come from item_set instead */
  const int32_t* item_set;       /* [n_items,3] lattice 1e-3 (not sample_from_distribution, C/bin3D.py:36-39) */
  int n_items;
  long long T;
  unsigned long long seed;
  const int32_t* stream; /* [N,T,3] lattice 1e-3; dataset mode: [n_traj,max_len,3] */
  const int32_t* ds_len;
  int ds_ntraj, ds_maxlen;
  int32_t* ems;     /* [N,6,ems_cap] EMS coordinates on the 1e-6 lattice (every one is an np.around(., 6) result) */
  double* boxes;    /* [N,6,I] lx,ly,lz,xe,ye,top; a row is written when its box is placed */
  uint16_t* leafg;  /* [N,L] generator ids (EMS, rotation, corner) of the current leaf nodes */
  double* volsum;   /* [N] running sum of placed volumes (get_ratio) */
  double* bsz;      /* [N,3,I] placed sizes x,y,z */
  StabHbm sb;       /* stability state, as in DiscreteParams (settings 1/3 only) */
  const double* den_stream; /* setting 3 density source, as in DiscreteParams */
  long long den_T;
  const double* ds_den;
  int table_global; /* 1: hash table + order list live in HBM (capacity beyond LDS) */
  uint32_t* gtab;   /* [N, cand_cap*5/4] */
  uint16_t* gorder; /* [N, order_cap] */
  uint32_t* gfpri;  /* [N, order_cap] shuffle priorities (HBM-table variant + shuffle) */
  int gt_by_block;  /* HBM table slices indexed by blockIdx (retry pass) instead of env */
  int pipe;         /* 1: the normal pass runs the two-wave candidate pipeline (pct_continuous_pipe.hip; setting 2, LDS table): a second
                       wave of the env's workgroup generates, hashes and de-duplicates the candidate batches while the first inserts them;
                       2: ... for the envs that enter the step with at least pipe_min_ems EMS only -- the second wave of every other
                       workgroup leaves at once (the small-bin configs: the registers bound the resident envs there, and the launch is
                       as long as its EMS-richest env) */
  int pipe_min_ems;
  uint32_t* gpark;  /* [N, PCT_PARK_WORDS] LDS table of 8192 slots: the 2048-slot table's entries wait here, dense and in slot order, while
                       the ONE LDS region both sizes share is wiped (round 5: the 100^3 env's table left HBM) */
  int prio_t[3];    /* wave_priority thresholds on the EMS count (0: off) */
  const int32_t* order; /* heavy-first dispatch, as in DiscreteParams */
  int rng_numpy;    /* 1: strict NumPy-stream mode (pct_set_numpy_rng), as in DiscreteParams */
  int np_items;     /* len(item_set) behind RandomBoxCreator's unread randint draws (sampling mode) */
  uint32_t* mt;     /* [N,624] MT19937 state words (position: scalars[7]) */
  double* mt_den;   /* [N] density drawn for the current observation (setting 3) */
  int retry_mode;   /* this launch is the large-capacity retry pass */
  int* retry_count; /* [1] envs queued by the normal pass (zeroed before it) */
  int* retry_ids;   /* [N] */
  int32_t* scalars; /* [N,PCT_SCALARS] */
  uint32_t* flags;
  unsigned long long* timing;
  float* obs;
  float* reward;
  uint8_t* done;
  int32_t* counter;
  double* ratio;
  float* mask; /* as in DiscreteParams */
  float* policy_rows; /* as in DiscreteParams */
  int* retry_total;   /* as in DiscreteParams */
  void* launch_ev_start; /* host-side launch options, as in DiscreteParams */
  void* launch_ev_stop;
};

// D/bin3D.py:75-84 next_den of the observation number `oc` (the env's life-long observation
// counter); traj / item_index locate the previewed item in dataset mode.
// A launch lasts as long as its slowest env, and an env's work grows with its EMS count: waves of EMS-rich envs
// take a higher issue priority (s_setprio) so that they run at a lone wave's pace from the start while the light
// waves of the same SIMD fill the gaps, instead of crawling at a quarter of it until the light ones are gone
// (C2: 75.4 -> 71.2 us per launch).  prio_t: ascending EMS-count thresholds of priorities 1..3 (0: off).
// Heavy-first dispatch: the sort key of env e (shader-clock cycles of its last step -- low 32 bits of the counter: a step
// is far shorter than their 1.8 s wrap -- and its live EMS count) sits behind the [N, PCT_SCALARS] scalars; the step's start time is parked in the
// first PCT_LDS_STASH bytes of the workgroup's LDS, in front of everything carve_lds / carve lay out.
#define PCT_LDS_STASH 16
__device__ __forceinline__ uint32_t* work_key_slot(int32_t* scalars, int N, int e) {
  return reinterpret_cast<uint32_t*>(scalars) + (size_t)N * PCT_SCALARS + e;
}
__device__ __forceinline__ void work_key_begin(unsigned char* smem) {
  if (threadIdx.x == 0) *reinterpret_cast<uint32_t*>(smem) = (uint32_t)__builtin_readcyclecounter();
}
// key word: cycles / 256 (20 bits, saturating) << 12 | live EMS count (12 bits, saturating); 0 after a reset
__device__ __forceinline__ void work_key_end(unsigned char* smem, int32_t* scalars, int N, int e, bool reset, int n_ems) {
  if (threadIdx.x == 0) {
    const uint32_t cyc = ((uint32_t)__builtin_readcyclecounter() - *reinterpret_cast<uint32_t*>(smem)) >> 8;
    const uint32_t ne = n_ems < 0 ? 0u : (uint32_t)n_ems;
    *work_key_slot(scalars, N, e) = reset ? 0u : ((cyc > 0xFFFFFu ? 0xFFFFFu : cyc) << 12) | (ne > 0xFFFu ? 0xFFFu : ne);
  }
}
__device__ __forceinline__ void wave_priority(int n_ems, const int prio_t[3]) {
  if (prio_t[0] <= 0) return;
  if (n_ems >= prio_t[2]) __builtin_amdgcn_s_setprio(3);
  else if (n_ems >= prio_t[1]) __builtin_amdgcn_s_setprio(2);
  else if (n_ems >= prio_t[0]) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}

// Observation rows are written once and read by the NEXT kernel (the policy), never again by this one.  Two ways of keeping them
// from waiting in the L2 as dirty lines for the end-of-kernel release were measured (profiles/r04_experiments.txt items 2, 7):
// PCT_OBS_NT = 1, non-temporal stores: no gain; = 2, write-through stores (system scope: sc0 sc1): C2 +1.4 % env-steps/s, but
// every 4-byte store becomes a partial-line write to memory -- HBM traffic 21.6 -> 71.2 MB per launch (1.11 x -> 3.65 x the
// algorithmic bytes).  Default 0: plain stores.
#ifndef PCT_OBS_NT
#define PCT_OBS_NT 0
#endif
__device__ __forceinline__ void obs_st(float* q, float v) {
#if PCT_OBS_NT == 2
  __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (write-through: sc0 sc1)
#elif PCT_OBS_NT
  __builtin_nontemporal_store(v, q);
#else
  *q = v;
#endif
}

// Retry queue of a launch: written by the normal pass, read by the retry pass enqueued behind it.  (Round 4 tried to overlap
// the two -- the retry pass dispatched with hipExtAnyOrderLaunch and waiting in-kernel on a counter of finished workgroups:
// the flag does not lift the barrier on gfx950 and the counter cost the normal pass 1 us, profiles/r04_experiments.txt.)
__device__ __forceinline__ void retry_enqueue(int* count, int* ids, int e) {
  if (threadIdx.x == 0) ids[atomicAdd(count, 1)] = e;
  // (round 6: the retry pass may run as the TAIL of this very launch, on another XCD: the entry is released at agent scope before
  // the workgroup reports itself finished -- pct_discrete_tail_kernel; a rare path, the fence costs an ordinary step nothing)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

template <typename Params>
__device__ __forceinline__ double next_density(const Params& p, int e, uint32_t oc, int traj, unsigned long long item_index) {
  if (p.setting != 3) return 1.0;
  if (p.source == PCT_ITEMS_DATASET) {  // self.next_box[3]
    int t = traj < p.ds_ntraj ? traj : p.ds_ntraj - 1;
    if (!p.ds_den || t < 0 || item_index >= (unsigned long long)p.ds_len[t]) return 1.0;
    return p.ds_den[(size_t)t * p.ds_maxlen + (size_t)item_index];
  }
  if (p.den_stream) return p.den_stream[(size_t)e * (size_t)p.den_T + (size_t)((unsigned long long)oc % (unsigned long long)p.den_T)];
  return pct_density(p.seed, (uint64_t)(p.env_id_base + e), (uint64_t)oc);
}

size_t continuous_lds_bytes(const ContinuousParams& p);
hipError_t launch_cpolicy_hash_rows(const ContinuousParams& p, float* rows_out, hipStream_t stream);
hipError_t launch_continuous_pipe(const ContinuousParams& p, int act, const void* actions, int row_len, int n_steps, const int32_t* env_ids,
                                  int n_ids, hipStream_t stream);
hipError_t launch_continuous_mt(const ContinuousParams& p, int act, const void* actions, int row_len, int n_steps,
                                const int32_t* env_ids, int n_ids, hipStream_t stream);
hipError_t launch_continuous(const ContinuousParams& p, int act, const void* actions, int row_len, int n_steps,
                             const int32_t* env_ids, int n_ids, hipStream_t stream);

size_t discrete_lds_bytes(const DiscreteParams& p);
hipError_t launch_policy_hash_rows(const DiscreteParams& p, float* rows_out, hipStream_t stream);
// which launches run the retry pass as their own tail (pct_discrete_tail_kernel) -- the host (pct_env.hip: no separate retry dispatch
// then) and the launcher must agree: the plain setting-2 transition of a whole batch, LNES = EMS, counter-keyed draws, untimed
inline bool discrete_tail_eligible(const DiscreteParams& p, int act, const int32_t* env_ids) {
  return p.tail_blocks > 0 && p.setting == 2 && p.lnes == PCT_LNES_EMS && !p.shuffle && !p.rng_numpy && p.timing == nullptr && !p.retry_mode &&
         env_ids == nullptr && (act == 0 /* ACT_ROWS */ || act == 1 /* ACT_INDEX */ || act == 2 /* ACT_HASH */);
}
hipError_t launch_discrete(const DiscreteParams& p, int act, const void* actions, int row_len, int n_steps,
                           const int32_t* env_ids, int n_ids, hipStream_t stream);

}  // namespace pct
#endif
