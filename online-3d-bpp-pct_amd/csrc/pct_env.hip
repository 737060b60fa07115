// pct_env.hip -- the C-ABI of include/pct_env.h: handle lifetime, HBM state, launches.
// No torch types, no CPU fallback: every transition is a HIP kernel launch; if there is no
// usable device pct_create fails with PCT_ERR_NO_DEVICE / PCT_ERR_HIP.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <utility>
#include <vector>

#include "../../include/pct_env.h"
#include "pct_device.h"
#include "pct_set.cuh"
#include "pct_stab.cuh"

namespace {
thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) return fail(PCT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));    \
  } while (0)

enum { ACT_ROWS = 0, ACT_INDEX = 1, ACT_HASH = 2, ACT_RESET = 3, ACT_HEUR = 4 };

/* Experiment knobs are read only under PCT_EXPERIMENT=1, and range-checked: a stray variable in a
 * production environment must not resize an LDS pool. */
bool experiments_on() {
  const char* e = getenv("PCT_EXPERIMENT");
  return e && atoi(e) == 1;
}
const char* knob(const char* name) { return experiments_on() ? getenv(name) : nullptr; }
int knob_int(const char* name, int dflt, int lo, int hi) {
  const char* e = knob(name);
  if (!e) return dflt;
  const int v = atoi(e);
  return (v < lo || v > hi) ? dflt : v;
}

}  // namespace

struct pct_env {
  pct_config cfg;
  int device;
  pct::DiscreteParams dp;
  pct::ContinuousParams cp;
  pct::ContinuousParams cp_retry; /* large-capacity HBM-table pass for envs whose candidate set outgrew LDS */
  bool has_retry;
  int cp_retry_blocks;
  bool has_dretry;      /* discrete env: large-capacity retry pass for envs that outgrow the LDS lists */
  int d_retry_ems, d_retry_cand;
  pct::StabCaps d_retry_stab; /* stability pools / workspace / queue of the retry pass (settings 1 / 3) */
  int* d_retry_base;    /* [2] ping-pong queue counters */
  int d_retry_parity;
  int* c_retry_base;    /* continuous env: the same */
  int c_retry_parity;
  int d_retry_blocks;   /* grid of the discrete retry pass */
  int* d_tail_base;     /* [2][64] ping-pong sub-counters of finished normal-pass workgroups (the retry pass as the launch's own tail) */
  size_t d_tail_lds;    /* bytes of one tail workgroup's HBM row = the retry pass's LDS layout */
  int d_tail_blocks;    /* tail workgroups of a launch that carries its retry pass (0: never) */
  bool continuous;
  // owned device memory
  std::vector<void*> owned;
  float* own_obs;
  float* own_reward;
  uint8_t* own_done;
  int32_t* own_counter;
  double* own_ratio;
  uint32_t* own_flags;
  int32_t* d_item_set;
  int32_t* d_stream;
  bool have_items;
  bool was_reset;
  // kernel timing (pct_profile_*)
  bool profiling;
  int prof_every;       /* every prof_every-th launch carries the event pair (pct_profile_enable) */
  long prof_tick;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used;
  int64_t prof_launches;
  double prof_ms;
  unsigned long long* timing_buf;
  // heavy-first dispatch (pct_device.h: work_key / order)
  int order_state;      /* 0: not decided yet (first full launch), 1: on, -1: off */
  int32_t* d_order;     /* [N] */
  int order_mode;       /* what the sort key is made of (pct_order_kernel) */
  /* (Sorting AHEAD on a stream of its own, overlapping the caller's policy, was tried: the two cross-stream event waits per
   * step cost ~20 us, five times what the in-line sort kernel takes -- profiles/r03_heavy_first.txt.) */

};

namespace {
int dev_alloc(pct_env* h, void** p, size_t bytes, bool zero) {
  HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
  h->owned.push_back(*p);
  if (zero) HIP_TRY(hipMemset(*p, 0, bytes ? bytes : 16));
  return PCT_OK;
}
// a source buffer that is being replaced: wait for the kernels that may still read it, then free it
int release_owned(pct_env* h, const void* old) {
  if (!old) return PCT_OK;
  for (size_t i = 0; i < h->owned.size(); i++)
    if (h->owned[i] == old) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(h->owned[i]));
      h->owned.erase(h->owned.begin() + (long)i);
      break;
    }
  return PCT_OK;
}
int use_device(const pct_env* h) {
  HIP_TRY(hipSetDevice(h->device));
  return PCT_OK;
}
// drain the recorded event pairs into the accumulator
int prof_drain(pct_env* h) {
  for (size_t i = 0; i < h->ev_used; i++) {
    HIP_TRY(hipEventSynchronize(h->ev_pool[i].second));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_pool[i].first, h->ev_pool[i].second));
    h->prof_ms += ms;
    h->prof_launches++;
  }
  h->ev_used = 0;
  return PCT_OK;
}
int prof_begin(pct_env* h, hipStream_t s, size_t* slot) {
  if (h->ev_used == h->ev_pool.size()) {
    if (h->ev_pool.size() >= 8192) {
      int rc = prof_drain(h);
      if (rc) return rc;
    } else {
      hipEvent_t a, b;
      HIP_TRY(hipEventCreate(&a));
      HIP_TRY(hipEventCreate(&b));
      h->ev_pool.push_back(std::make_pair(a, b));
    }
  }
  *slot = h->ev_used++;  /* the pair is handed to hipExtLaunchKernel: it brackets exactly the step kernel's dispatch */
  (void)s;
  return PCT_OK;
}
/* Heavy-first dispatch.  A launch that holds more envs than the chip keeps resident is worked off in the order the
 * dispatcher hands out workgroups; an env whose step is five times the mean (a crowded bin, a deep stability walk) that
 * happens to start last sets the launch's length all by itself (c1: 4 envs per SIMD slot in turn, launch = 1.7 x the
 * mean slot).  Every env's wave therefore leaves the cycles its step took (work_key_slot), this kernel turns them into a
 * workgroup -> env map, longest first (one 1024-thread workgroup, a 256-bin counting sort -- a few microseconds), and
 * the step kernel looks its env up in it: longest-processing-time-first list scheduling.  Any bijection is a correct
 * placement, so the results do not depend on it. */
__global__ void __launch_bounds__(1024) pct_order_kernel(const uint32_t* __restrict__ key, int32_t* __restrict__ order, int N, int mode) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t start[256];
  __shared__ uint32_t cmax, emax;
  const int t = threadIdx.x;
  if (t < 256) hist[t] = 0;
  if (t == 0) cmax = emax = 0;
  // key word (pct_device.h work_key_end): cycles / 256 << 12 | live EMS count; the sort value is the sum of the two, each
  // scaled to its maximum over the envs (mode 2, the default; 0: the cycles alone, 1: the EMS count alone -- measured:
  // profiles/r03_heavy_first.txt), cut into 256 bins, bin 0 = the longest.  Up to 16 keys per thread (N <= 16 384) stay
  // in registers over the three passes -- maxima, histogram, scatter; a larger N re-reads the rest from memory.
  constexpr int R = 16;
  uint32_t kr[R];
#pragma unroll
  for (int j = 0; j < R; j++) {
    const int i = t + j * 1024;
    kr[j] = i < N ? key[i] : 0u;
  }
  uint32_t mc = 0, me = 0;
#pragma unroll
  for (int j = 0; j < R; j++) {
    mc = max(mc, kr[j] >> 12);
    me = max(me, kr[j] & 0xFFFu);
  }
  for (int i = t + R * 1024; i < N; i += 1024) {
    const uint32_t k = key[i];
    mc = max(mc, k >> 12);
    me = max(me, k & 0xFFFu);
  }
  for (int o = 32; o; o >>= 1) {
    mc = max(mc, (uint32_t)__shfl_xor((int)mc, o, 64));
    me = max(me, (uint32_t)__shfl_xor((int)me, o, 64));
  }
  __syncthreads();
  if ((t & 63) == 0) { atomicMax(&cmax, mc); atomicMax(&emax, me); }
  __syncthreads();
  const float wc = mode == 1 ? 0.f : (mode == 0 ? 255.9f : 127.95f) / (float)(cmax + 1u);
  const float we = mode == 0 ? 0.f : (mode == 1 ? 255.9f : 127.95f) / (float)(emax + 1u);
  auto bin = [&](uint32_t k) -> uint32_t {
    const uint32_t v = k ? (uint32_t)((float)(k >> 12) * wc + (float)(k & 0xFFFu) * we) : 0u;  // (0: just reset)
    return 255u - (v > 255u ? 255u : v);
  };
#pragma unroll
  for (int j = 0; j < R; j++)
    if (t + j * 1024 < N) atomicAdd(&hist[bin(kr[j])], 1u);
  for (int i = t + R * 1024; i < N; i += 1024) atomicAdd(&hist[bin(key[i])], 1u);
  __syncthreads();
  if (t < 64) {  // exclusive prefix sum over the 256 bins: four per lane of one wave
    const uint32_t a = hist[4 * t], b = hist[4 * t + 1], c = hist[4 * t + 2], d = hist[4 * t + 3];
    uint32_t incl = a + b + c + d;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
      if (t >= o) incl += v;
    }
    const uint32_t excl = incl - (a + b + c + d);
    start[4 * t] = excl;
    start[4 * t + 1] = excl + a;
    start[4 * t + 2] = excl + a + b;
    start[4 * t + 3] = excl + a + b + c;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < R; j++)
    if (t + j * 1024 < N) order[atomicAdd(&start[bin(kr[j])], 1u)] = t + j * 1024;
  for (int i = t + R * 1024; i < N; i += 1024) order[atomicAdd(&start[bin(key[i])], 1u)] = i;
}

int order_setup(pct_env* h) {
  h->order_state = -1;
  const char* ev = knob("PCT_ORDER"); /* kernel experiments: 0 = never, 1 = always */
  if (ev && atoi(ev) <= 0) return PCT_OK;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, h->device));
  const bool stab = h->cfg.setting != 2;
  size_t lds = h->continuous ? pct::continuous_lds_bytes(h->cp) : pct::discrete_lds_bytes(h->dp);
  lds = (lds + 511) & ~(size_t)511;
  const size_t lds_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 160 * 1024;
  long per_cu = lds ? (long)(lds_cu / lds) : 64;
  /* one-wave workgroups: the stability kernels run 1 wave per SIMD, the plain discrete ones 4, the plain continuous ones 3 */
  const long by_regs = stab ? 4 : (h->continuous ? 12 : 16);
  if (per_cu > by_regs) per_cu = by_regs;
  const long resident = per_cu * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
  const long N = h->continuous ? h->cp.N : h->dp.N;
  if (!(ev && atoi(ev) > 0) && N <= resident) return PCT_OK;
  int rc = dev_alloc(h, (void**)&h->d_order, (size_t)N * sizeof(int32_t), true);
  if (rc) return rc;
  h->order_mode = knob_int("PCT_ORDER_MODE", 2, 0, 2); /* kernel experiments only */

  h->order_state = 1;
  return PCT_OK;
}

hipError_t order_launch(pct_env* h, hipStream_t s) {
  const int N = h->continuous ? h->cp.N : h->dp.N;
  const int32_t* scalars = h->continuous ? h->cp.scalars : h->dp.scalars;
  hipLaunchKernelGGL(pct_order_kernel, dim3(1), dim3(1024), 0, s,
                     reinterpret_cast<const uint32_t*>(scalars) + (size_t)N * PCT_SCALARS, h->d_order, N, h->order_mode);
  return hipGetLastError();
}
/* the step kernel was not dispatched: the event pair taken for it will never be signalled -- hand it back, or every later
 * pct_profile_read would fail in hipEventElapsedTime on it */
int launch_failed(pct_env* h, bool took_pair, hipError_t e, const char* what) {
  if (took_pair && h->ev_used > 0) h->ev_used--;
  h->dp.launch_ev_start = h->dp.launch_ev_stop = h->cp.launch_ev_start = h->cp.launch_ev_stop = nullptr;
  return fail(PCT_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}
int launch(pct_env* h, int act, const void* actions, int row_len, int n_steps, const int32_t* ids, int n_ids,
           void* stream) {
  hipStream_t s = (hipStream_t)stream;
  size_t slot = 0;
  if (h->order_state == 0) {
    int rc = order_setup(h);
    if (rc) return rc;
  }
  h->dp.order = h->cp.order = nullptr;
  if (h->order_state == 1 && act != ACT_RESET && !ids) {
    HIP_TRY(order_launch(h, s));
    h->dp.order = h->cp.order = h->d_order;
  }
  /* the profiling pair brackets the step kernel itself (the kernel the roofline is about), not the retry pass */
  h->dp.launch_ev_start = h->dp.launch_ev_stop = h->cp.launch_ev_start = h->cp.launch_ev_stop = nullptr;
  /* (a dispatch that carries events costs ~7 us of its own on this runtime -- C2: 64.6 M env-steps/s with a pair on every launch,
   * 72.6 M with none, profiles/r04_experiments.txt item 9 -- so a caller that only wants the AVERAGE kernel time samples) */
  const bool timed = h->profiling && (h->prof_tick++ % (h->prof_every > 0 ? h->prof_every : 1)) == 0;
  if (timed) {
    int rc = prof_begin(h, s, &slot);
    if (rc) return rc;
    h->dp.launch_ev_start = h->cp.launch_ev_start = (void*)h->ev_pool[slot].first;
    h->dp.launch_ev_stop = h->cp.launch_ev_stop = (void*)h->ev_pool[slot].second;
  }
  const int normal_grid = (act == ACT_RESET && ids) ? n_ids : (h->continuous ? h->cp.N : h->dp.N);
  if (timed && normal_grid <= 0) { /* nothing is dispatched: the pair would never be signalled */
    h->ev_used--;
    h->dp.launch_ev_start = h->dp.launch_ev_stop = h->cp.launch_ev_start = h->cp.launch_ev_stop = nullptr;
  }
  if (h->continuous) {
    h->cp.full_obs = h->dp.full_obs;
    h->cp.policy_rows = h->dp.policy_rows;
    if (h->has_retry) h->cp.retry_count = h->c_retry_base + h->c_retry_parity; /* ping-pong pair of queue counters */
    {
      const hipError_t le = pct::launch_continuous(h->cp, act, actions, row_len, n_steps, ids, n_ids, s);
      if (le != hipSuccess) return launch_failed(h, timed && normal_grid > 0, le, "launch_continuous");
    }
    if (h->has_retry) {
      /* keep the retry pass in step with everything that may have changed on the handle */
      pct::ContinuousParams& q = h->cp_retry;
      const pct::ContinuousParams& c = h->cp;
      q.source = c.source; q.stream = c.stream; q.T = c.T; q.seed = c.seed; q.ds_len = c.ds_len;
      q.den_stream = c.den_stream; q.den_T = c.den_T; q.ds_den = c.ds_den;
      q.ds_ntraj = c.ds_ntraj; q.ds_maxlen = c.ds_maxlen; q.sample_left = c.sample_left; q.sample_right = c.sample_right;
      q.item_set = c.item_set; q.n_items = c.n_items;
      q.rng_numpy = c.rng_numpy; q.np_items = c.np_items; q.mt = c.mt; q.mt_den = c.mt_den;
      q.low_bound = c.low_bound; q.obs = c.obs; q.reward = c.reward; q.done = c.done; q.counter = c.counter;
      q.ratio = c.ratio; q.flags = c.flags; q.timing = nullptr; q.full_obs = c.full_obs; q.mask = c.mask;
      q.retry_count = c.retry_count;
      q.policy_rows = c.policy_rows;
      q.retry_total = c.retry_total;
      q.launch_ev_start = q.launch_ev_stop = nullptr;
      q.order = nullptr;
      q.retry_mode = h->c_retry_parity ? -1 : 1;
      h->c_retry_parity ^= 1;
      HIP_TRY(pct::launch_continuous(q, act, actions, row_len, n_steps, ids, h->cp_retry_blocks, s));
    }
    if (act != ACT_RESET || !ids) h->dp.full_obs = 0;
  } else {
    pct::DiscreteParams q;
    h->dp.tail_q = nullptr;
    if (h->has_dretry) { /* ping-pong pair of queue counters: this step's is h->d_retry_base[parity] */
      h->dp.retry_count = h->d_retry_base + h->d_retry_parity;
      if (h->d_tail_base) {
        h->dp.tail_done = h->d_tail_base + 64 * h->d_retry_parity;
        h->dp.tail_rm = h->d_retry_parity ? -64 : 64;
        /* one dispatch per step where every env is resident at once (C2: +1.3 %, 72.3 against 71.4 M env-steps/s); a launch that
         * holds more envs than the chip (heavy-first dispatch on: C4) keeps the separate retry dispatch -- there the tail's second
         * inlined transition costs the common path more than the dispatch gap it saves (101.5 against 102.9 M): profiles/r06_experiments.txt */
        h->dp.tail_blocks = (h->order_state == 1 && !knob("PCT_TAIL")) ? 0 : h->d_tail_blocks;
      }
      /* the same step again, with larger lists, for the envs the normal pass queued (usually none) */
      q = h->dp;
      q.ems_cap = h->d_retry_ems;
      q.cand_cap = h->d_retry_cand;
      q.sb.caps = h->d_retry_stab;
      q.retry_mode = h->d_retry_parity ? -1 : 1;
      q.timing = nullptr;
      q.launch_ev_start = q.launch_ev_stop = nullptr;
      q.tail_blocks = 0;
      h->dp.tail_q = &q; /* (read by the launcher only if this launch carries the retry pass as its own tail) */
    }
    const bool tail = h->has_dretry && pct::discrete_tail_eligible(h->dp, act, ids);
    {
      const hipError_t le = pct::launch_discrete(h->dp, act, actions, row_len, n_steps, ids, n_ids, s);
      h->dp.tail_q = nullptr;
      if (le != hipSuccess) return launch_failed(h, timed && normal_grid > 0, le, "launch_discrete");
    }
    if (h->has_dretry) {
      /* ... as a small grid-strided dispatch of its own (the grid exits at once when nothing was queued) -- unless the
       * launch above carried it as its tail (pct_discrete_tail_kernel: one dispatch per step) */
      if (!tail) HIP_TRY(pct::launch_discrete(q, act, actions, row_len, n_steps, nullptr, h->d_retry_blocks, s));
      h->d_retry_parity ^= 1;
    }
    if (act != ACT_RESET || !ids) h->dp.full_obs = 0; /* every env has rewritten its rows */
  }
  return PCT_OK;
}
/* Stability capacities (pct_stab.cuh).  Normal pass: one pool entry and two polygon vertices per internal node (measured
 * on 10^3 bins with items 1..5, 48 envs x 600 steps: at most 36 entries and 122 vertices live at a time -- an episode ends
 * at 30..35 boxes, far below the 80 internal nodes), a hull workspace for 16 two-supporter candidates per round and a
 * queue of 96 walk tasks (half of it the reserve of the depth-first mode): LDS is what bounds the resident envs per CU
 * (profiles/r03_stability_tuning.txt), and what outgrows these goes through the retry pass.  Retry pass: eight entries
 * / sixteen vertices per node, a workspace that takes a box on 120 supporters and 512 tasks. */
void stab_default_caps(int I, pct::StabCaps& normal, pct::StabCaps& retry, int retry_lsq_n = pct::STAB_LSQ) {
  normal.SP = I < 64 ? 64 : I;
  if (normal.SP > 4094) normal.SP = 4094; /* STAB_END: 12-bit pool offsets */
  normal.PP = 2 * I < 128 ? 128 : 2 * I;
  normal.ws_bytes = 16 * pct::stab_ws_need(2);  /* stab_fit_wave() widens these two into the LDS the env has left */
  normal.queue = 96;
  /* least-squares splits over up to 8 supporters in the normal pass (3.3 KB of LDS: one system of 7 / 8 supporters or two of 6 at
   * a time), up to 16 in the retry pass (four / two systems of the small classes side by side) */
  normal.gelsd = PCT_LSTSQ_GELSD; /* the default since round 5 (pct_set_lstsq_mode): the reference's own np.linalg.lstsq */
  retry.gelsd = PCT_LSTSQ_GELSD;
  normal.lsq_n = 8;
  normal.lsq_bytes = (int)pct::stab_lsq_bytes(normal.lsq_n, false);
  /* round 6: 25 supporters -- LAPACK's own limit for this path of dgelsd (pct_stab.cuh STAB_LSQ); 73 KB of the retry pass's LDS.  A
   * handle whose other retry capacities leave no room for that (hundreds of internal nodes) falls back to 16 (pct_create) */
  retry.lsq_n = retry_lsq_n;
  retry.lsq_bytes = (int)pct::stab_lsq_bytes(retry.lsq_n, true);
  retry.SP = 8 * I < 4094 ? 8 * I : 4094;
  retry.PP = 16 * I < 65535 ? 16 * I : 65535;
  retry.ws_bytes = 16 * 1024;
  retry.queue = 512;
  /* kernel experiments only: PCT_STAB_SP / _PP / _WS / _Q override the normal pass */
  normal.SP = knob_int("PCT_STAB_SP", normal.SP, 1, 4094);
  normal.PP = knob_int("PCT_STAB_PP", normal.PP, 4, 65535);
  normal.ws_bytes = knob_int("PCT_STAB_WS", normal.ws_bytes, pct::stab_ws_need(2), 64 * 1024);
  normal.queue = knob_int("PCT_STAB_Q", normal.queue, 8, 4096);
  if (retry.SP < normal.SP) retry.SP = normal.SP;
  if (retry.PP < normal.PP) retry.PP = normal.PP;
  if (retry.ws_bytes < normal.ws_bytes) retry.ws_bytes = normal.ws_bytes;
  if (retry.queue < normal.queue) retry.queue = normal.queue;
}
/* The stability kernels run one wave per SIMD (their float64 code needs more than 256 VGPRs), i.e. four envs per CU:
 * each may use 40 KiB of LDS for nothing.  What the env's own lists and the stability pools leave of that goes to the
 * wave's hull workspace (as many two-supporter candidates per round as fit, up to 64) and walk queue (three tasks per
 * lane): the rounds of a check are latency chains, so the more candidates share one the better (measured, c1: 32
 * lanes / 112 tasks 9.2 M env-steps/s, 16 lanes / 96 tasks 7.0 M; profiles/r03_stability_tuning.txt). */
void stab_fit_wave(size_t lds_without_wave, pct::StabCaps& caps) {
  if (knob("PCT_STAB_WS") || knob("PCT_STAB_Q")) return; /* explicit (experiments / tests) */
  const long budget = 40 * 1024 - 64 - (long)lds_without_wave - (long)caps.lsq_bytes;
  const int per = pct::stab_ws_need(2);
  for (int lanes = 64; lanes >= 16; lanes -= 8) {
    const int q = 3 * lanes > 96 ? 3 * lanes : 96;
    if ((long)lanes * per + (long)q * 36 + 32 <= budget) {
      caps.ws_bytes = lanes * per;
      caps.queue = q;
      return;
    }
  }
}
/* The retry pass's stability capacities, fitted into `room` bytes of LDS.  Two kinds of rare overflow compete for that LDS: large
 * pools / queue (hundreds of boxes, deep walks) and the least-squares workspace of a box on up to 25 supporters (73 KB).  First choice:
 * the 25-supporter workspace with pools of at least three times the DEFAULT normal pass's (3 I entries, 6 I vertices, 192 tasks) --
 * the 10^3 / 80-node handles; else round 5's configuration (16 supporters, pools as large as fit).  lds(caps) = the pass's LDS bytes.
 * Returns false when not even the normal pass's own capacities fit (no retry pass then). */
template <typename F>
bool fit_retry_stab(F lds, size_t room, int I, const pct::StabCaps& normal, pct::StabCaps& out) {
  const int Ie = I < 64 ? 64 : I;
  auto mx = [](int a, int b) { return a > b ? a : b; };
  for (int a = 0; a < 2; a++) {
    pct::StabCaps nrm = normal, rt = normal;
    stab_default_caps(I, nrm, rt, a == 0 ? pct::STAB_LSQ : 16);
    rt.gelsd = normal.gelsd;
    rt.SP = mx(rt.SP, normal.SP); rt.PP = mx(rt.PP, normal.PP); rt.ws_bytes = mx(rt.ws_bytes, normal.ws_bytes); rt.queue = mx(rt.queue, normal.queue);
    const int sp_floor = a == 0 ? mx(3 * Ie, normal.SP) : normal.SP;
    const int pp_floor = a == 0 ? mx(6 * Ie, normal.PP) : normal.PP;
    const int q_floor = a == 0 ? mx(192, normal.queue) : normal.queue;
    while (lds(rt) > room && (rt.SP > sp_floor || rt.PP > pp_floor || rt.queue > q_floor)) {
      rt.SP = mx((rt.SP * 3) / 4, sp_floor);
      rt.PP = mx((rt.PP * 3) / 4, pp_floor);
      rt.queue = mx((rt.queue * 3) / 4, q_floor);
    }
    if (lds(rt) <= room) { out = rt; return true; }
  }
  return false;
}
bool is_cand_cap_ok(int c) {
  for (int s = 8; s <= (1 << 20); s <<= 2)
    if (s == c) return true;
  return false;
}
}  // namespace

extern "C" {

int pct_abi_version(void) { return PCT_ABI_VERSION; }
const char* pct_last_error(void) { return g_err; }

/* wave_priority thresholds (pct_device.h) on the env's live EMS count: defaults measured on C2 / C3
 * (profiles/r02_prio_sweep.txt); PCT_WAVE_PRIO="t1,t2,t3" overrides ("0": off) -- a tuning knob, not ABI */
static void prio_thresholds(int out[3], int d1, int d2, int d3) {
  out[0] = d1; out[1] = d2; out[2] = d3;
  const char* s = knob("PCT_WAVE_PRIO");
  if (!s) return;
  int a = 0, b = 0, c = 0;
  int n = sscanf(s, "%d,%d,%d", &a, &b, &c);
  if (n == 1 && a <= 0) { out[0] = out[1] = out[2] = 0; return; }
  if (n == 3 && a > 0 && b >= a && c >= b) { out[0] = a; out[1] = b; out[2] = c; }
}

int pct_create(const pct_config* cfg, int device, pct_env** out) {
  if (!cfg || !out) return fail(PCT_ERR_INVALID_ARG, "null argument");
  if (cfg->struct_size != (int32_t)sizeof(pct_config))
    return fail(PCT_ERR_INVALID_ARG, "pct_config size mismatch (%d vs %d)", cfg->struct_size, (int)sizeof(pct_config));
  if (cfg->env_kind != PCT_ENV_DISCRETE && cfg->env_kind != PCT_ENV_CONTINUOUS)
    return fail(PCT_ERR_INVALID_ARG, "unknown env_kind");
  const bool cont = cfg->env_kind == PCT_ENV_CONTINUOUS;
  if (cfg->setting < 1 || cfg->setting > 3) return fail(PCT_ERR_INVALID_ARG, "setting must be 1, 2 or 3 (tools.py:132)");
  if (cfg->lnes < PCT_LNES_EMS || cfg->lnes > PCT_LNES_FC) return fail(PCT_ERR_INVALID_ARG, "unknown LNES (tools.py:133)");
  if (cfg->lnes != PCT_LNES_EMS && cfg->env_kind != PCT_ENV_DISCRETE)
    return fail(PCT_ERR_UNSUPPORTED, "EV / EP / CP / FC are reachable only in the discrete env (C/bin3D.py:53)");
  if (cfg->num_envs < 1 || cfg->internal_node_holder < 1 || cfg->leaf_node_holder < 1)
    return fail(PCT_ERR_INVALID_ARG, "num_envs / holders must be positive");
  /* the pooled stability state packs box ids into 10 bits (0x3FF = "no box") and pool offsets into 12 / 16 bits (pct_stab.cuh) */
  if (cfg->setting != 2 && cfg->internal_node_holder > 1022)
    return fail(PCT_ERR_UNSUPPORTED, "settings 1 / 3 support at most 1022 internal nodes (10-bit box ids of the stability state)");
  int W = cfg->container[0], Ly = cfg->container[1], H = cfg->container[2];
  if (W < 1 || Ly < 1 || H < 1) return fail(PCT_ERR_INVALID_ARG, "bad container");
  int maxdim = W > Ly ? W : Ly;
  if (H > maxdim) maxdim = H;
  if (!cont && maxdim > 1023) return fail(PCT_ERR_UNSUPPORTED, "discrete bins are limited to 1023 per axis");
  if (cont && (W % 1000 || Ly % 1000 || H % 1000))
    return fail(PCT_ERR_UNSUPPORTED, "continuous container sizes must be whole bin units (multiples of 1000 lattice units)");
  /* discrete bins <= 31 per axis (32-bit keys): 128 EMS (82 is the most the 10^3 probes ever
   * held before elimination) keeps the env at 10 KB of LDS = 16 resident envs per CU */
  /* EMS kept after elimination: 128 covers the 10-unit bins of both envs with room to spare (most
   * ever seen: 59 discrete, 81 continuous), larger bins default to 256 */
  /* continuous bins beyond 12 units (BASELINE configs[4]: 100^3 with U(5,25) items holds up to ~260 live EMS and
   * several thousand distinct candidates): a 512-EMS LDS list (longer lists go through the retry pass, up to 1536; measured at C5: 320 / 384 / 448 / 512 /
   * 640 -> 0.78 / 0.91 / 1.12 / 1.14 / 1.05 M env-steps/s)
   * and a 32768-slot candidate table, which lives in HBM */
  int ems_cap = cfg->ems_capacity > 0 ? cfg->ems_capacity
                                      : (((cont ? maxdim / 1000 : maxdim) <= 12) ? 128 : (cont ? 512 : 256));
  if (!cont && ems_cap < 64) return fail(PCT_ERR_INVALID_ARG, "ems_capacity must be >= 64");
  /* candidate table: 2048 slots (1228 distinct candidates) cover the 10^3-class bins with room to
   * spare; larger discrete bins default to 8192 (4915) */
  /* the stability settings generate two orientations instead of six: a 512-slot table (307 candidates) covers their
   * 10^3-class bins (at most 150 distinct candidates were seen) and frees 6 KB of LDS for the stability state */
  const bool small_bin = (cont ? maxdim / 1000 : maxdim) <= 12;
  int cand_cap = cfg->candidate_capacity > 0 ? cfg->candidate_capacity
                                             : (small_bin ? (cfg->setting != 2 ? 512 : 2048)
                                                          /* larger discrete bins under the stability settings: two orientations x
                                                           * 256 EMS = at most 512 candidates, a 2048-slot table; 8192 slots of
                                                           * 64-bit keys would not leave room for the stability state */
                                                          /* larger continuous bins (round 5): 8192 slots in LDS (4915 distinct candidates;
                                                           * C5 holds ~1800, beyond goes to the retry pass's 32768-slot HBM table).
                                                           * Rounds 1-4: 32768 slots in HBM, every probe an agent-scope atomic */
                                                          : (cont ? (cfg->setting != 2 ? 32768 : 8192) : (cfg->setting != 2 ? 2048 : 8192)));
  if (knob("PCT_CAND_CAP") && cfg->candidate_capacity <= 0) cand_cap = atoi(knob("PCT_CAND_CAP")); /* kernel experiments only; never over an explicit capacity */
  if (!is_cand_cap_ok(cand_cap)) return fail(PCT_ERR_INVALID_ARG, "candidate_capacity must be 8*4^k (8,32,...,2048,8192)");

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1) return fail(PCT_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(PCT_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);

  pct_env* h = new pct_env();
  h->cfg = *cfg;
  h->device = device;
  h->have_items = false;
  h->was_reset = false;
  h->d_item_set = nullptr;
  h->d_stream = nullptr;
  h->profiling = false;
  h->prof_every = 1;
  h->prof_tick = 0;
  h->ev_used = 0;
  h->prof_launches = 0;
  h->prof_ms = 0.0;
  h->timing_buf = nullptr;
  h->order_state = 0;
  h->d_order = nullptr;
  h->continuous = cont;
  h->has_dretry = false;
  h->d_retry_blocks = knob_int("PCT_RETRY_BLOCKS", 16, 1, 1024);
  memset(&h->cp, 0, sizeof h->cp);
  int rc = use_device(h);
  if (rc) { delete h; return rc; }

  if (cont) {
    pct::ContinuousParams& c = h->cp;
    c.N = cfg->num_envs;
    c.I = cfg->internal_node_holder;
    c.L = cfg->leaf_node_holder;
    c.row_len = (c.I + c.L + 1) * 9;
    c.setting = cfg->setting;
    c.shuffle = cfg->shuffle ? 1 : 0;
    c.W = W / 1000; c.Ly = Ly / 1000; c.H = H / 1000;
    c.ems_cap = ems_cap;
    c.cand_cap = cand_cap;
    c.order_cap = (cand_cap * 3) / 5 + 8;
    c.table_global = cand_cap > 8192 ? 1 : 0; /* beyond 8192 slots the table cannot share LDS with the EMS */
    /* the two-wave candidate pipeline (pct_continuous_pipe.hip): on where the LDS -- not the registers -- bounds the resident envs
     * (the 8192-slot LDS table of the large bins: three envs per CU, one wave each, a SIMD idle), off for the small bins, whose
     * one-wave workgroups already fill the SIMDs' wave slots.  PCT_PIPE = 0 / 1 (kernel experiments) overrides for any LDS table. */
    c.pipe = (cfg->setting == 2 && !c.table_global) ? knob_int("PCT_PIPE", cand_cap > 2048 ? 1 : 0, 0, 2) : 0;
    c.pipe_min_ems = knob_int("PCT_PIPE_MIN_EMS", 32, 0, 4096); /* pipe = 2 (experiment): a second wave for the EMS-rich envs only */
    /* the LDS region shared by the hash table (one region for every table size up to 2048 slots; a 8192-slot
     * table sits behind the 2048-slot one it grows from) and the GENEMS children scratch (6 int32 words per
     * child): 2 * ems_cap children when the table lives in LDS (225 pre-elimination entries were seen at C3),
     * ems_cap otherwise */
    /* (round 5: the 8192-slot table shares ONE region with the 2048-slot one it grows from -- the old entries wait in an HBM row,
     * gpark, while the region is wiped) */
    size_t tab_words = c.table_global ? 0 : (size_t)cand_cap;
    size_t child_words = (size_t)6 * ems_cap * (c.table_global ? 1 : 2);
    c.union_words = (int)(tab_words > child_words ? tab_words : child_words);
    if (c.union_words < 192) c.union_words = 192; /* the fast start parks 64 generator ids behind a 128-slot table */
    if (maxdim / 1000 > 2000) { delete h; return fail(PCT_ERR_UNSUPPORTED, "continuous bins are limited to 2000 units per axis (int32 lattice 1e-6)"); }
    c.env_id_base = cfg->env_id_base;
    c.source = PCT_ITEMS_NONE;
    if ((size_t)ems_cap * 24 + 23 > 65535) { delete h; return fail(PCT_ERR_INVALID_ARG, "ems_capacity too large for 16-bit generator ids"); }
    pct::StabCaps retry_stab = c.sb.caps;
    if (cfg->setting != 2) {
      stab_default_caps(c.I, c.sb.caps, retry_stab);
      {
        pct::ContinuousParams t0 = c;
        t0.sb.caps.ws_bytes = 0;
        t0.sb.caps.queue = 0;
        t0.sb.caps.lsq_n = 0;
        t0.sb.caps.lsq_bytes = 0;
        stab_fit_wave(pct::continuous_lds_bytes(t0), c.sb.caps);
      }
      c.sb.sp_stride = retry_stab.SP;
      c.sb.pp_stride = retry_stab.PP;
    }
    size_t clds = pct::continuous_lds_bytes(c);
    if (clds > 160 * 1024) { delete h; return fail(PCT_ERR_INVALID_ARG, "capacities need %zu B of LDS (> 160 KiB)", clds); }
    size_t Nn = (size_t)c.N;
#define CALLOC_(ptr, bytes)                                  \
  do {                                                       \
    rc = dev_alloc(h, (void**)&(ptr), (bytes), true);        \
    if (rc) { pct_destroy(h); return rc; }                   \
  } while (0)
    /* the retry pass keeps four times the EMS list (its LDS permitting, and below the 16-bit generator-id limit of
     * 2729 EMS); the HBM rows are as long as the longest list any pass may leave behind */
    c.ems_stride = ems_cap;
    if (!c.table_global || true) {
      int big = ems_cap * 4;
      if (big > 2560) big = 2560;
      pct::ContinuousParams t = c;
      t.ems_cap = big;
      t.table_global = 1;
      t.pipe = 0;
      t.union_words = 12 * big;
      while (big > ems_cap && pct::continuous_lds_bytes(t) > 150 * 1024) {
        big -= ems_cap;
        t.ems_cap = big;
        t.union_words = 12 * big;
      }
      if (big > ems_cap) c.ems_stride = big;
    }
    CALLOC_(c.ems, Nn * 6 * c.ems_stride * sizeof(int32_t));
    CALLOC_(c.boxes, Nn * 6 * c.I * sizeof(double));
    CALLOC_(c.leafg, Nn * c.L * sizeof(uint16_t));
    CALLOC_(c.volsum, Nn * sizeof(double));
    CALLOC_(c.bsz, Nn * 3 * c.I * sizeof(double));
    if (cfg->setting != 2) {
      CALLOC_(c.sb.stk, Nn * c.I * 4 * sizeof(double));
      CALLOC_(c.sb.den, Nn * c.I * sizeof(double));
      CALLOC_(c.sb.share, Nn * c.sb.sp_stride * 4 * sizeof(double));
      CALLOC_(c.sb.poly, Nn * c.sb.pp_stride * 2 * sizeof(double));
      CALLOC_(c.sb.meta, Nn * c.I * 2 * sizeof(uint32_t));
      CALLOC_(c.sb.up, Nn * c.I * sizeof(uint32_t));
      CALLOC_(c.sb.ent, Nn * c.sb.sp_stride * sizeof(uint32_t));
    }
    if (!c.table_global && cand_cap > 2048) CALLOC_(c.gpark, Nn * (size_t)PCT_PARK_WORDS * sizeof(uint32_t));
    if (c.table_global) {
      CALLOC_(c.gtab, Nn * (size_t)(cand_cap + cand_cap / 4) * sizeof(uint32_t));
      CALLOC_(c.gorder, Nn * (size_t)c.order_cap * sizeof(uint16_t));
      if (c.shuffle) CALLOC_(c.gfpri, Nn * (size_t)c.order_cap * sizeof(uint32_t));
    }
    CALLOC_(c.scalars, Nn * (PCT_SCALARS + 1) * sizeof(int32_t)); /* (+ the [N] work keys of the heavy-first dispatch) */
    CALLOC_(h->own_flags, Nn * sizeof(uint32_t));
    CALLOC_(h->own_obs, Nn * c.row_len * sizeof(float));
    CALLOC_(h->own_reward, Nn * sizeof(float));
    CALLOC_(h->own_done, Nn);
    CALLOC_(h->own_counter, Nn * sizeof(int32_t));
    CALLOC_(h->own_ratio, Nn * sizeof(double));
    h->has_retry = false;
    {
      /* normal pass: table in LDS (or, beyond 8192 slots, already in HBM); envs that need more -- a larger
       * candidate table or a longer EMS list -- are re-run by a small grid-strided pass with 32768-slot tables in
       * HBM (covers ems_capacity * 24 candidates up to 19660) and ems_stride EMS */
      const int RB = 128, big = 32768;
      CALLOC_(h->c_retry_base, 4 * sizeof(int)); /* queue counters [0,1], totals [2,3] */
      c.retry_total = h->c_retry_base + 2;
      c.retry_count = h->c_retry_base;
      h->c_retry_parity = 0;
      CALLOC_(c.retry_ids, Nn * sizeof(int));
      h->has_retry = true;
      h->cp_retry_blocks = RB;
      h->cp_retry = c;
      pct::ContinuousParams& q = h->cp_retry;
      q.retry_mode = 1;
      q.pipe = 0;
      q.table_global = 1;
      q.gt_by_block = 1;
      q.cand_cap = big;
      q.order_cap = (big * 3) / 5 + 8;
      q.ems_cap = c.ems_stride;
      q.union_words = 12 * q.ems_cap > 192 ? 12 * q.ems_cap : 192; /* children scratch only: 2 * ems_cap of them */
      if (cfg->setting != 2) { /* larger stability pools / workspace / queue, as far as the LDS of that pass goes */
        /* (strict NumPy-stream mode, chosen after pct_create, adds the env's 624 MT19937 words: room is kept for them) */
        const size_t lds_room = 160 * 1024 - 624 * sizeof(uint32_t);
        q.sb.caps = retry_stab;
        auto lds_of = [&](const pct::StabCaps& caps) { pct::ContinuousParams t = q; t.sb.caps = caps; return pct::continuous_lds_bytes(t); };
        if (!fit_retry_stab(lds_of, lds_room, c.I, c.sb.caps, q.sb.caps)) q.sb.caps = retry_stab;
        if (pct::continuous_lds_bytes(q) > lds_room) { pct_destroy(h); return fail(PCT_ERR_INVALID_ARG, "the retry pass does not fit the LDS"); }
      }
      CALLOC_(q.gtab, (size_t)RB * (size_t)(big + big / 4) * sizeof(uint32_t));
      CALLOC_(q.gorder, (size_t)RB * (size_t)q.order_cap * sizeof(uint16_t));
      if (c.shuffle) CALLOC_(q.gfpri, (size_t)RB * (size_t)q.order_cap * sizeof(uint32_t));
    }
#undef CALLOC_
    c.obs = h->own_obs; c.reward = h->own_reward; c.done = h->own_done; c.counter = h->own_counter;
    c.ratio = h->own_ratio; c.flags = h->own_flags;
    h->dp.N = c.N; h->dp.row_len = c.row_len; h->dp.I = c.I; h->dp.L = c.L;
    h->dp.obs = c.obs; h->dp.reward = c.reward; h->dp.done = c.done; h->dp.counter = c.counter;
    h->dp.ratio = c.ratio; h->dp.flags = c.flags;
    prio_thresholds(c.prio_t, 24, 36, 48);
    if (h->has_retry) memcpy(h->cp_retry.prio_t, c.prio_t, sizeof c.prio_t);
    *out = h;
    return PCT_OK;
  }
  pct::DiscreteParams& p = h->dp;
  memset(&p, 0, sizeof p);
  p.N = cfg->num_envs;
  p.W = W; p.Ly = Ly; p.H = H;
  p.A = W > Ly ? W : Ly;
  p.AA = (p.A * p.A + 7) & ~7;
  p.I = cfg->internal_node_holder;
  p.L = cfg->leaf_node_holder;
  p.row_len = (p.I + p.L + 1) * 9;
  p.setting = cfg->setting;
  p.lnes = cfg->lnes;
  p.shuffle = cfg->shuffle ? 1 : 0;
  p.ems_cap = ems_cap;
  p.cand_cap = cand_cap;
  p.key_bytes = maxdim <= 31 ? 4 : 8;
  /* retry pass: four times the EMS list (up to 1024) and, while it still fits the 160 KB of LDS, four times
   * the candidate table; the HBM EMS rows are as long as the longest list any pass may leave behind.  The
   * stability settings keep per-box state beyond the lists and run without it (overflow -> flag). */
  /* (strict NumPy-stream mode, chosen after pct_create, adds the env's 624 MT19937 words: room is kept for them) */
  const size_t lds_room = 160 * 1024 - 624 * sizeof(uint32_t);
  h->d_retry_ems = ems_cap * 4 < 1024 ? ems_cap * 4 : (ems_cap > 1024 ? ems_cap : 1024);
  h->d_retry_cand = cand_cap;
  {
    pct::DiscreteParams t = p;
    t.ems_cap = h->d_retry_ems;
    t.cand_cap = cand_cap * 4;
    if (pct::discrete_lds_bytes(t) <= lds_room) h->d_retry_cand = cand_cap * 4;
    t.cand_cap = h->d_retry_cand;
    if (pct::discrete_lds_bytes(t) > lds_room) h->d_retry_ems = ems_cap;
  }
  /* stability settings: pool / workspace / queue capacities of the normal pass and of the retry pass (pct_stab.cuh) */
  h->d_retry_stab = p.sb.caps;
  if (cfg->setting != 2) {
    stab_default_caps(p.I, p.sb.caps, h->d_retry_stab);
    {
      pct::DiscreteParams t0 = p;
      t0.sb.caps.ws_bytes = 0;
      t0.sb.caps.queue = 0;
      t0.sb.caps.lsq_n = 0;
      t0.sb.caps.lsq_bytes = 0;
      stab_fit_wave(pct::discrete_lds_bytes(t0), p.sb.caps);
    }
    pct::DiscreteParams t = p;
    t.ems_cap = h->d_retry_ems;
    t.cand_cap = h->d_retry_cand;
    {
      auto lds_of = [&](const pct::StabCaps& caps) { pct::DiscreteParams u = t; u.sb.caps = caps; return pct::discrete_lds_bytes(u); };
      t.sb.caps = h->d_retry_stab;
      pct::StabCaps fitted;
      if (fit_retry_stab(lds_of, lds_room, p.I, p.sb.caps, fitted)) t.sb.caps = fitted;
    }
    if (pct::discrete_lds_bytes(t) > lds_room) { h->d_retry_ems = ems_cap; h->d_retry_cand = cand_cap; t.sb.caps = p.sb.caps; }
    h->d_retry_stab = t.sb.caps;
    p.sb.sp_stride = h->d_retry_stab.SP;
    p.sb.pp_stride = h->d_retry_stab.PP;
  }
  h->has_dretry = cfg->reserved[0] != PCT_OVERFLOW_RETRY_OFF &&
                  (h->d_retry_ems > ems_cap || h->d_retry_cand > cand_cap ||
                   (cfg->setting != 2 && (h->d_retry_stab.SP > p.sb.caps.SP || h->d_retry_stab.queue > p.sb.caps.queue)));
  p.ems_stride = h->has_dretry ? h->d_retry_ems : ems_cap;
  p.env_id_base = cfg->env_id_base;
  p.source = PCT_ITEMS_NONE;

  size_t lds = pct::discrete_lds_bytes(p);
  if (lds > 160 * 1024) { delete h; return fail(PCT_ERR_INVALID_ARG, "capacities need %zu B of LDS (> 160 KiB)", lds); }

  size_t N = (size_t)p.N;
#define ALLOC(ptr, bytes)                                   \
  do {                                                      \
    rc = dev_alloc(h, (void**)&(ptr), (bytes), true);       \
    if (rc) { pct_destroy(h); return rc; }                  \
  } while (0)
  ALLOC(p.hmap, N * p.AA * sizeof(int16_t));
  ALLOC(p.ems, N * p.ems_stride * p.key_bytes);
  ALLOC(p.boxes, N * p.I * p.key_bytes);
  ALLOC(p.leaves, N * p.L * p.key_bytes);
  ALLOC(p.scalars, N * (PCT_SCALARS + 1) * sizeof(int32_t)); /* (+ the [N] work keys of the heavy-first dispatch) */
  ALLOC(p.set_scratch, (N + 1024) * 768 * sizeof(uint32_t)); /* pct_discrete_impl.cuh WS_KMAX: one row per workgroup of any pass (the tail's included) */
  if (cfg->setting != 2) {
    ALLOC(p.sb.stk, N * p.I * 4 * sizeof(double));
    ALLOC(p.sb.den, N * p.I * sizeof(double));
    ALLOC(p.sb.share, N * p.sb.sp_stride * 4 * sizeof(double));
    ALLOC(p.sb.poly, N * p.sb.pp_stride * 2 * sizeof(double));
    ALLOC(p.sb.meta, N * p.I * 2 * sizeof(uint32_t));
    ALLOC(p.sb.up, N * p.I * sizeof(uint32_t));
    ALLOC(p.sb.ent, N * p.sb.sp_stride * sizeof(uint32_t));
  }
  if (h->has_dretry) {
    ALLOC(h->d_retry_base, 4 * sizeof(int)); /* queue counters [0,1], totals [2,3] */
    p.retry_total = h->d_retry_base + 2;
    p.retry_count = h->d_retry_base;
    h->d_retry_parity = 0;
    /* Round 6: the plain setting-2 steps run the retry pass as the TAIL of their own launch (pct_discrete_tail_kernel): its
     * workgroups keep the retry pass's LDS layout in a row of HBM each.  PCT_TAIL=0 (kernel experiments): two dispatches */
    if (cfg->setting == 2 && knob_int("PCT_TAIL", 1, 0, 1)) {
      pct::DiscreteParams t = p;
      t.ems_cap = h->d_retry_ems;
      t.cand_cap = h->d_retry_cand;
      h->d_tail_lds = (pct::discrete_lds_bytes(t) + 255) & ~(size_t)255;
      h->d_tail_blocks = h->d_retry_blocks;
      p.tail_blocks = h->d_tail_blocks;
      p.tail_scratch_bytes = (int)h->d_tail_lds;
      ALLOC(h->d_tail_base, 2 * 64 * sizeof(int));
      ALLOC(p.tail_scratch, (size_t)p.tail_blocks * h->d_tail_lds);
      p.tail_done = h->d_tail_base;
      p.tail_rm = 64;
    }
    ALLOC(p.retry_ids, N * sizeof(int));
  }
  ALLOC(h->own_flags, N * sizeof(uint32_t));
  ALLOC(h->own_obs, N * p.row_len * sizeof(float));
  ALLOC(h->own_reward, N * sizeof(float));
  ALLOC(h->own_done, N);
  ALLOC(h->own_counter, N * sizeof(int32_t));
  ALLOC(h->own_ratio, N * sizeof(double));
#undef ALLOC
  p.obs = h->own_obs;
  p.reward = h->own_reward;
  p.done = h->own_done;
  p.counter = h->own_counter;
  p.ratio = h->own_ratio;
  p.flags = h->own_flags;
  prio_thresholds(p.prio_t, 16, 22, 28);
  *out = h;
  return PCT_OK;
}

int pct_destroy(pct_env* h) {
  if (!h) return PCT_OK;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* q : h->owned) (void)hipFree(q);
  for (auto& ev : h->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  delete h;
  return PCT_OK;
}

int pct_set_item_set(pct_env* h, const int32_t* item_set, int32_t n) {
  if (!h || !item_set || n < 1) return fail(PCT_ERR_INVALID_ARG, "bad item set");
  int rc = use_device(h);
  if (rc) return rc;
  int mn = item_set[0], mx = item_set[0];
  for (int i = 0; i < 3 * n; i++) {
    if (item_set[i] < mn) mn = item_set[i];
    if (item_set[i] > mx) mx = item_set[i];
  }
  if (mn < 1) return fail(PCT_ERR_INVALID_ARG, "item sizes must be >= 1 lattice unit");
  (void)mx;
  rc = release_owned(h, h->d_item_set);
  if (rc) return rc;
  h->d_item_set = nullptr;
  void* d = nullptr;
  rc = dev_alloc(h, &d, sizeof(int32_t) * 3 * (size_t)n, false);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(d, item_set, sizeof(int32_t) * 3 * (size_t)n, hipMemcpyHostToDevice));
  h->d_item_set = (int32_t*)d;
  h->dp.item_set = h->d_item_set;
  h->dp.n_items = n;
  h->dp.low_bound = mn; /* bin3D.py:23 size_minimum */
  if (h->continuous) { /* C/bin3D.py:29,36-39: items from the set (lattice 1e-3), size_minimum = its smallest entry */
    h->cp.item_set = h->d_item_set;
    h->cp.n_items = n;
    h->cp.sample_left = 0;
    h->cp.sample_right = 0;
    h->cp.low_bound = (double)mn / 1000.0;
    h->cp_retry.item_set = h->d_item_set;
    h->cp_retry.n_items = n;
  }
  h->have_items = true;
  return PCT_OK;
}

int pct_set_sample_bounds(pct_env* h, int32_t left, int32_t right) {
  if (!h || !h->continuous) return fail(PCT_ERR_INVALID_ARG, "sample bounds apply to the continuous env");
  if (left < 1 || right < left) return fail(PCT_ERR_INVALID_ARG, "bad bounds");
  h->cp.sample_left = left;
  h->cp.sample_right = right;
  h->cp.low_bound = (double)left / 1000.0; /* C/bin3D.py:25-27 */
  h->have_items = true;
  return PCT_OK;
}

int pct_set_item_stream(pct_env* h, const int32_t* items, int64_t T) {
  if (!h || !items || T < 1) return fail(PCT_ERR_INVALID_ARG, "bad stream");
  int rc = use_device(h);
  if (rc) return rc;
  size_t n = (size_t)h->dp.N * (size_t)T * 3;
  rc = release_owned(h, h->dp.stream);  /* a replaced stream / dataset (synchronises: kernels may still read it) */
  if (rc) return rc;
  if (h->dp.ds_len) { rc = release_owned(h, h->dp.ds_len); if (rc) return rc; }
  h->dp.stream = nullptr; h->dp.ds_len = nullptr; h->cp.stream = nullptr; h->cp.ds_len = nullptr;
  void* d = nullptr;
  rc = dev_alloc(h, &d, sizeof(int32_t) * n, false);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(d, items, sizeof(int32_t) * n, hipMemcpyHostToDevice));
  h->d_stream = (int32_t*)d;
  h->dp.stream = h->d_stream;
  h->dp.T = T;
  h->dp.source = PCT_ITEMS_STREAM;
  h->cp.stream = h->d_stream;
  h->cp.T = T;
  h->cp.source = PCT_ITEMS_STREAM;
  return PCT_OK;
}

int pct_set_item_dataset(pct_env* h, const int32_t* items, const int32_t* lengths, int32_t n_traj, int32_t max_len) {
  if (!h || !items || !lengths || n_traj < 2 || max_len < 1)
    return fail(PCT_ERR_INVALID_ARG, "bad dataset (at least 2 trajectories: the first episode plays trajectory 1)");
  int rc = use_device(h);
  if (rc) return rc;
  rc = release_owned(h, h->dp.stream);
  if (rc) return rc;
  if (h->dp.ds_len) { rc = release_owned(h, h->dp.ds_len); if (rc) return rc; }
  h->dp.stream = nullptr; h->dp.ds_len = nullptr; h->cp.stream = nullptr; h->cp.ds_len = nullptr;
  void* d = nullptr;
  void* dl = nullptr;
  size_t n = (size_t)n_traj * (size_t)max_len * 3;
  rc = dev_alloc(h, &d, sizeof(int32_t) * n, false);
  if (rc) return rc;
  rc = dev_alloc(h, &dl, sizeof(int32_t) * (size_t)n_traj, false);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(d, items, sizeof(int32_t) * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dl, lengths, sizeof(int32_t) * (size_t)n_traj, hipMemcpyHostToDevice));
  h->dp.stream = (int32_t*)d; h->dp.ds_len = (int32_t*)dl; h->dp.ds_ntraj = n_traj; h->dp.ds_maxlen = max_len;
  h->dp.source = PCT_ITEMS_DATASET;
  h->cp.stream = (int32_t*)d; h->cp.ds_len = (int32_t*)dl; h->cp.ds_ntraj = n_traj; h->cp.ds_maxlen = max_len;
  h->cp.source = PCT_ITEMS_DATASET;
  return PCT_OK;
}

int pct_set_density_stream(pct_env* h, const double* den, int64_t T) {
  if (!h || !den || T < 1) return fail(PCT_ERR_INVALID_ARG, "bad density stream");
  int rc = use_device(h);
  if (rc) return rc;
  rc = release_owned(h, h->dp.den_stream);
  if (rc) return rc;
  h->dp.den_stream = nullptr; h->cp.den_stream = nullptr;
  void* d = nullptr;
  size_t n = (size_t)h->cfg.num_envs * (size_t)T;
  rc = dev_alloc(h, &d, sizeof(double) * n, false);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(d, den, sizeof(double) * n, hipMemcpyHostToDevice));
  h->dp.den_stream = (const double*)d; h->dp.den_T = T;
  h->cp.den_stream = (const double*)d; h->cp.den_T = T;
  return PCT_OK;
}

int pct_set_dataset_density(pct_env* h, const double* den) {
  if (!h || !den) return fail(PCT_ERR_INVALID_ARG, "null argument");
  if (h->dp.source != PCT_ITEMS_DATASET) return fail(PCT_ERR_STATE, "pct_set_item_dataset must come first");
  int rc = use_device(h);
  if (rc) return rc;
  rc = release_owned(h, h->dp.ds_den);
  if (rc) return rc;
  h->dp.ds_den = nullptr; h->cp.ds_den = nullptr;
  void* d = nullptr;
  size_t n = (size_t)h->dp.ds_ntraj * (size_t)h->dp.ds_maxlen;
  rc = dev_alloc(h, &d, sizeof(double) * n, false);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(d, den, sizeof(double) * n, hipMemcpyHostToDevice));
  h->dp.ds_den = (const double*)d;
  h->cp.ds_den = (const double*)d;
  return PCT_OK;
}

int pct_set_shuffle_seed(pct_env* h, uint64_t seed) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  h->dp.shuffle_seed = seed;
  h->cp.shuffle_seed = seed;
  h->cp_retry.shuffle_seed = seed;
  return PCT_OK;
}

int pct_set_lstsq_mode(pct_env* h, int32_t mode) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  if (mode != PCT_LSTSQ_JACOBI && mode != PCT_LSTSQ_GELSD && mode != PCT_LSTSQ_GELSD_AVX2)
    return fail(PCT_ERR_INVALID_ARG, "lstsq mode: PCT_LSTSQ_JACOBI, PCT_LSTSQ_GELSD or PCT_LSTSQ_GELSD_AVX2");
  if (h->cfg.setting == 2) return PCT_OK; /* no stability check: nothing to select */
  /* a launch's workspace is sized for one Jacobi system of its largest class; the dgelsd workspace of that class must fit it */
  const pct::StabCaps* all[] = {&h->dp.sb.caps, &h->cp.sb.caps, &h->cp_retry.sb.caps, &h->d_retry_stab};
  if (mode != PCT_LSTSQ_JACOBI)
    for (const pct::StabCaps* c : all)
      if (c->lsq_n > 0 && sizeof(double) * pct::stab_lsq_slot_doubles(c->lsq_n, true) > (size_t)c->lsq_bytes)
        return fail(PCT_ERR_UNSUPPORTED, "the least-squares workspace of this handle does not hold a dgelsd system");
  h->dp.sb.caps.gelsd = mode;
  h->cp.sb.caps.gelsd = mode;
  h->cp_retry.sb.caps.gelsd = mode;
  h->d_retry_stab.gelsd = mode;
  return PCT_OK;
}

int pct_set_numpy_rng(pct_env* h, uint32_t seed) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  if (h->continuous && h->cp.sample_right <= 0)
    return fail(PCT_ERR_UNSUPPORTED, "NumPy-stream mode, continuous env: items sampled from U(a, b) (pct_set_sample_bounds)");
  if (!h->have_items) return fail(PCT_ERR_STATE, "pct_set_item_set / pct_set_sample_bounds must come first");
  if (h->was_reset) return fail(PCT_ERR_STATE, "pct_set_numpy_rng must precede the first reset");
  int rc = use_device(h);
  if (rc) return rc;
  const size_t N = (size_t)h->dp.N;
  if (!h->dp.mt) {
    rc = dev_alloc(h, (void**)&h->dp.mt, N * 624 * sizeof(uint32_t), false);
    if (rc) return rc;
    rc = dev_alloc(h, (void**)&h->dp.mt_den, N * sizeof(double), true);
    if (rc) return rc;
  }
  h->cp.mt = h->dp.mt;
  h->cp.mt_den = h->dp.mt_den;
  /* np.random.seed(seed + rank) in every worker (envs.py:49, bin3D.py:47-54): mt19937_seed == init_genrand */
  std::vector<uint32_t> st(N * 624);
  for (size_t e = 0; e < N; e++) {
    uint32_t sd = seed + (uint32_t)h->cfg.env_id_base + (uint32_t)e;
    for (int i = 0; i < 624; i++) {
      st[e * 624 + i] = sd;
      sd = 1812433253u * (sd ^ (sd >> 30)) + (uint32_t)i + 1u;
    }
  }
  HIP_TRY(hipMemcpy(h->dp.mt, st.data(), st.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  /* position 624: the first draw regenerates the block (scalars[7] of every env) */
  std::vector<int32_t> pos(N, 624);
  int32_t* scalars = h->continuous ? h->cp.scalars : h->dp.scalars;
  HIP_TRY(hipMemcpy2D(scalars + 7, PCT_SCALARS * sizeof(int32_t), pos.data(), sizeof(int32_t), sizeof(int32_t), N,
                      hipMemcpyHostToDevice));
  if (h->continuous) {
    h->cp.rng_numpy = 1;
    h->cp.source = PCT_ITEMS_SAMPLER;
    if (h->cp.np_items < 1) h->cp.np_items = 125; /* givenData.py:13-18 item_size_set, what main.py hands the env */
    pct::ContinuousParams q = h->cp_retry;
    q.rng_numpy = 1;
    if (pct::continuous_lds_bytes(h->cp) > 160 * 1024 || (h->has_retry && pct::continuous_lds_bytes(q) > 160 * 1024))
      return fail(PCT_ERR_INVALID_ARG, "capacities + MT19937 state exceed the LDS");
    return PCT_OK;
  }
  h->dp.rng_numpy = 1;
  h->dp.source = PCT_ITEMS_SAMPLER;
  if (pct::discrete_lds_bytes(h->dp) > 160 * 1024) return fail(PCT_ERR_INVALID_ARG, "capacities + MT19937 state exceed the LDS");
  if (h->has_dretry) {
    pct::DiscreteParams q = h->dp;
    q.ems_cap = h->d_retry_ems;
    q.cand_cap = h->d_retry_cand;
    q.sb.caps = h->d_retry_stab;
    if (pct::discrete_lds_bytes(q) > 160 * 1024)
      return fail(PCT_ERR_INVALID_ARG, "the retry pass's capacities + MT19937 state exceed the LDS");
  }
  return PCT_OK;
}

int pct_set_numpy_item_count(pct_env* h, int32_t n) {
  if (!h || n < 1) return fail(PCT_ERR_INVALID_ARG, "bad count");
  if (!h->continuous) return fail(PCT_ERR_UNSUPPORTED, "continuous env only (the discrete env draws from its item set)");
  h->cp.np_items = n;
  return PCT_OK;
}

int pct_set_sampler(pct_env* h, uint64_t seed) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  if (!h->have_items) return fail(PCT_ERR_STATE, "pct_set_item_set must come first");
  h->dp.seed = seed;
  h->dp.source = PCT_ITEMS_SAMPLER;
  h->cp.seed = seed;
  h->cp.source = PCT_ITEMS_SAMPLER;
  return PCT_OK;
}

int pct_bind_outputs(pct_env* h, float* obs, float* reward, uint8_t* done, int32_t* counter, double* ratio,
                     uint32_t* error_flags) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  h->dp.full_obs = 1; /* the new buffer holds nobody's previous observation */
  h->dp.obs = obs ? obs : h->own_obs;
  h->dp.reward = reward ? reward : h->own_reward;
  h->dp.done = done ? done : h->own_done;
  h->dp.counter = counter ? counter : h->own_counter;
  h->dp.ratio = ratio ? ratio : h->own_ratio;
  h->dp.flags = error_flags ? error_flags : h->own_flags;
  h->dp.mask = nullptr;
  h->cp.mask = nullptr;
  h->cp.obs = h->dp.obs; h->cp.reward = h->dp.reward; h->cp.done = h->dp.done; h->cp.counter = h->dp.counter;
  h->cp.ratio = h->dp.ratio; h->cp.flags = h->dp.flags;
  return PCT_OK;
}

int pct_bind_rollout_slot(pct_env* h, float* obs_next, float* reward, float* mask) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  if (!obs_next) return fail(PCT_ERR_INVALID_ARG, "null observation slot");
  h->dp.full_obs = 1; /* the slot does not hold this env's previous observation: every row is written */
  h->dp.obs = obs_next;
  if (reward) h->dp.reward = reward;
  h->dp.mask = mask;
  h->cp.obs = h->dp.obs; h->cp.reward = h->dp.reward; h->cp.mask = h->dp.mask;
  return PCT_OK;
}

float* pct_obs(pct_env* h) { return h ? h->dp.obs : nullptr; }
float* pct_reward(pct_env* h) { return h ? h->dp.reward : nullptr; }
uint8_t* pct_done(pct_env* h) { return h ? h->dp.done : nullptr; }
int32_t* pct_info_counter(pct_env* h) { return h ? h->dp.counter : nullptr; }
double* pct_info_ratio(pct_env* h) { return h ? h->dp.ratio : nullptr; }
uint32_t* pct_error_flags(pct_env* h) { return h ? h->dp.flags : nullptr; }
int32_t pct_obs_row_len(pct_env* h) { return h ? h->dp.row_len : 0; }

static int ready(pct_env* h, bool need_reset) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  if (!h->have_items) return fail(PCT_ERR_STATE, "item set not configured");
  if ((h->continuous ? h->cp.source : h->dp.source) == PCT_ITEMS_NONE)
    return fail(PCT_ERR_STATE, "item source not configured");
  if (need_reset && !h->was_reset) return fail(PCT_ERR_STATE, "step before the first reset");
  return use_device(h);
}

int pct_reset(pct_env* h, const int32_t* env_ids, int32_t n, void* stream) {
  int rc = ready(h, false);
  if (rc) return rc;
  if (env_ids && !h->was_reset) return fail(PCT_ERR_STATE, "first reset must cover all envs");
  rc = launch(h, ACT_RESET, nullptr, 0, 1, env_ids, n, stream);
  if (rc) return rc;
  h->was_reset = true;
  return PCT_OK;
}

int pct_step_rows(pct_env* h, const float* rows, int32_t row_len, void* stream) {
  int rc = ready(h, true);
  if (rc) return rc;
  if (!rows) return fail(PCT_ERR_INVALID_ARG, "null actions");
  if (row_len != 9 && row_len != 6 && row_len != 3) return fail(PCT_ERR_INVALID_ARG, "row_len must be 9, 6 or 3");
  return launch(h, ACT_ROWS, rows, row_len, 1, nullptr, 0, stream);
}

int pct_step_index(pct_env* h, const int64_t* leaf_index, void* stream) {
  int rc = ready(h, true);
  if (rc) return rc;
  if (!leaf_index) return fail(PCT_ERR_INVALID_ARG, "null actions");
  return launch(h, ACT_INDEX, leaf_index, 0, 1, nullptr, 0, stream);
}

int pct_step_hash_policy(pct_env* h, int32_t n_steps, void* stream) {
  int rc = ready(h, true);
  if (rc) return rc;
  if (n_steps < 1) return fail(PCT_ERR_INVALID_ARG, "n_steps must be >= 1");
  /* (the stability state is LDS-resident during a launch since round 3: n steps in one launch under every setting) */
  return launch(h, ACT_HASH, nullptr, 0, n_steps, nullptr, 0, stream);
}

int pct_step_heuristic(pct_env* h, int32_t kind, int32_t n_steps, void* stream) {
  int rc = ready(h, true);
  if (rc) return rc;
  if (n_steps < 1) return fail(PCT_ERR_INVALID_ARG, "n_steps must be >= 1");
  if (kind < PCT_HEUR_LSAH || kind > PCT_HEUR_RANDOM) return fail(PCT_ERR_INVALID_ARG, "unknown heuristic");
  /* MACS keeps one empty-cell mask per (level, row) in the idle table region: 32-bit words, two per 64 cells along y */
  if (!h->continuous && kind == PCT_HEUR_MACS &&
      (size_t)h->cfg.container[0] * h->cfg.container[2] * (h->cfg.container[1] > 32 ? 2 * (size_t)((h->cfg.container[1] + 63) / 64) : 1) >
          (size_t)h->dp.cand_cap * (h->dp.key_bytes / 4))
    return fail(PCT_ERR_UNSUPPORTED, "MACS: the level masks (W * H * ceil(Ly / 64) 64-bit words) do not fit the table scratch (raise candidate_capacity)");
  if (!h->continuous && kind == PCT_HEUR_RANDOM && ((size_t)h->cfg.container[0] * h->cfg.container[1] * 6 / 64 + 1) * 2 >
                                     (size_t)h->dp.cand_cap * (h->dp.key_bytes / 4))
    return fail(PCT_ERR_UNSUPPORTED, "RANDOM: the feasibility masks do not fit the table scratch (raise candidate_capacity)");
  if (h->cfg.lnes != PCT_LNES_EMS) return fail(PCT_ERR_UNSUPPORTED, "the heuristics read the EMS list (LNES = EMS)");
  if (h->continuous) {
    /* tools.py:217-218: only LSAH, OnlineBPH and BR run on PackingContinuous */
    if (kind != PCT_HEUR_LSAH && kind != PCT_HEUR_OBPH && kind != PCT_HEUR_BR)
      return fail(PCT_ERR_UNSUPPORTED, "only LSAH, OnlineBPH and BR are allowed for the continuous environment (tools.py:217-218)");
    if (h->cp.rng_numpy)
      return fail(PCT_ERR_UNSUPPORTED, "the heuristic policies are not available in strict NumPy-stream mode (pct_set_numpy_rng)");
    /* per-pair scores (LSAH: one double per (EMS, rotation)) live in the idle table region */
    if ((size_t)h->cp.ems_cap * (h->cfg.setting == 2 ? 6 : 2) * 2 + 64 > (size_t)h->cp.union_words)
      return fail(PCT_ERR_UNSUPPORTED, "the heuristic scores do not fit the table scratch (lower ems_capacity)");
    return launch(h, ACT_HEUR, nullptr, kind, n_steps, nullptr, 0, stream);
  }
  /* strict NumPy-stream mode has no heuristic kernels: the ACT_HEUR templates draw from the counter-keyed sources and
   * the LDS layout of that mode carries the MT19937 state where their shuffle arrays would lie (include/pct_env.h) */
  if (h->dp.rng_numpy)
    return fail(PCT_ERR_UNSUPPORTED, "the heuristic policies are not available in strict NumPy-stream mode (pct_set_numpy_rng)");
  return launch(h, ACT_HEUR, nullptr, kind, n_steps, nullptr, 0, stream);
}

int pct_policy_hash_rows(pct_env* h, float* rows_out, void* stream) {
  int rc = ready(h, true);
  if (rc) return rc;
  if (!rows_out) return fail(PCT_ERR_INVALID_ARG, "null rows_out");
  if (h->continuous) HIP_TRY(pct::launch_cpolicy_hash_rows(h->cp, rows_out, (hipStream_t)stream));
  else HIP_TRY(pct::launch_policy_hash_rows(h->dp, rows_out, (hipStream_t)stream));
  return PCT_OK;
}

/* The stand-in policy as an INDEX: four envs per 256-thread workgroup, a wave per env counts the valid leaves of its
 * observation (mask column 8 of rows I .. I+L-1, tools.py:103) and writes pct_mix32(global id, t) % k -- what
 * pct_policy_hash_rows gathers a row by.  One launch, like a policy's forward pass. */
__global__ void __launch_bounds__(256) pct_policy_hash_index_kernel(const float* __restrict__ obs, const int32_t* __restrict__ scalars,
                                                                     int64_t* __restrict__ out, int N, int row_len, int I, int L, int base) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= N) return;
  const float* o = obs + (size_t)e * row_len;
  int k = 0;
  for (int b = 0; b < L; b += 64) {
    const int j = b + lane;
    const bool v = j < L && o[(I + j) * 9 + 8] != 0.f;
    k += __popcll(__ballot(v));
  }
  const uint32_t t = (uint32_t)scalars[(size_t)e * PCT_SCALARS + 6];
  if (lane == 0) out[e] = k > 0 ? (int64_t)(pct_mix32((uint32_t)(base + e), t) % (uint32_t)k) : 0;
}

int pct_policy_hash_index(pct_env* h, int64_t* index_out, void* stream) {
  int rc = ready(h, true);
  if (rc) return rc;
  if (!index_out) return fail(PCT_ERR_INVALID_ARG, "null index_out");
  const int N = h->dp.N;
  const int32_t* scalars = h->continuous ? h->cp.scalars : h->dp.scalars;
  const int base = h->continuous ? h->cp.env_id_base : h->dp.env_id_base;
  hipLaunchKernelGGL(pct_policy_hash_index_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, h->dp.obs, scalars, index_out, N,
                     h->dp.row_len, h->dp.I, h->dp.L, base);
  HIP_TRY(hipGetLastError());
  return PCT_OK;
}

int pct_bind_policy_rows(pct_env* h, float* rows_out) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  h->dp.policy_rows = rows_out; /* (launch() hands it to the continuous block and to the retry passes) */
  return PCT_OK;
}

int pct_profile_enable(pct_env* h, int32_t on) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  int rc = use_device(h);
  if (rc) return rc;
  if (!on && h->profiling) {
    rc = prof_drain(h);
    if (rc) return rc;
  }
  h->profiling = on != 0;
  h->prof_every = on > 1 ? on : 1;
  h->prof_tick = 0;
  return PCT_OK;
}

int pct_profile_read(pct_env* h, int64_t* n_launches, double* total_ms) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  int rc = use_device(h);
  if (rc) return rc;
  rc = prof_drain(h);
  if (rc) return rc;
  if (n_launches) *n_launches = h->prof_launches;
  if (total_ms) *total_ms = h->prof_ms;
  h->prof_launches = 0;
  h->prof_ms = 0.0;
  return PCT_OK;
}

int pct_debug_work_keys(pct_env* h, uint32_t* host_out) {
  if (!h || !host_out) return fail(PCT_ERR_INVALID_ARG, "null argument");
  int rc = use_device(h);
  if (rc) return rc;
  const size_t N = (size_t)(h->continuous ? h->cp.N : h->dp.N);
  const int32_t* scalars = h->continuous ? h->cp.scalars : h->dp.scalars;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host_out, reinterpret_cast<const uint32_t*>(scalars) + N * PCT_SCALARS, N * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return PCT_OK;
}

int pct_debug_retry_count(pct_env* h, int32_t* last, int64_t* envs_total, int64_t* launches_total) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  int rc = use_device(h);
  if (rc) return rc;
  int w[4] = {0, 0, 0, 0};
  const int* base = h->continuous ? (h->has_retry ? h->c_retry_base : nullptr) : (h->has_dretry ? h->d_retry_base : nullptr);
  const int parity = h->continuous ? h->c_retry_parity : h->d_retry_parity;
  HIP_TRY(hipDeviceSynchronize());
  if (base) HIP_TRY(hipMemcpy(w, base, sizeof w, hipMemcpyDeviceToHost));
  if (last) *last = w[parity ^ 1]; /* the last launch's counter: the parity was flipped after it was enqueued */
  if (envs_total) *envs_total = w[2];
  if (launches_total) *launches_total = w[3];
  return PCT_OK;
}

int32_t pct_debug_timing_slots(void) { return PCT_TIMING_SLOTS; }

int pct_debug_phase_timing(pct_env* h, int32_t on, uint64_t* host_out) {
  if (!h) return fail(PCT_ERR_INVALID_ARG, "null handle");
  int rc = use_device(h);
  if (rc) return rc;
  size_t bytes = (size_t)h->dp.N * PCT_TIMING_SLOTS * sizeof(unsigned long long);
  HIP_TRY(hipDeviceSynchronize());
  if (host_out && h->timing_buf) HIP_TRY(hipMemcpy(host_out, h->timing_buf, bytes, hipMemcpyDeviceToHost));
  if (on) {
    if (!h->timing_buf) {
      rc = dev_alloc(h, (void**)&h->timing_buf, bytes, true);
      if (rc) return rc;
    }
    HIP_TRY(hipMemset(h->timing_buf, 0, bytes));
    h->dp.timing = h->timing_buf;
    h->cp.timing = h->timing_buf;
  } else {
    h->dp.timing = nullptr;
    h->cp.timing = nullptr;
  }
  return PCT_OK;
}

int pct_debug_state(pct_env* h, int32_t e, int32_t* heightmap, int32_t* ems, int32_t cap_ems, int32_t* n_ems,
                    int32_t* n_boxes, int32_t* next_item, int64_t* draw_cursor) {
  if (!h || e < 0 || e >= h->dp.N) return fail(PCT_ERR_INVALID_ARG, "bad env id");
  if (h->continuous) return fail(PCT_ERR_UNSUPPORTED, "use pct_debug_state_f64 for the continuous env");
  int rc = use_device(h);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  const pct::DiscreteParams& p = h->dp;
  int32_t sc[PCT_SCALARS];
  HIP_TRY(hipMemcpy(sc, p.scalars + (size_t)e * PCT_SCALARS, sizeof sc, hipMemcpyDeviceToHost));
  if (heightmap) {
    std::vector<int16_t> hm(p.AA);
    HIP_TRY(hipMemcpy(hm.data(), p.hmap + (size_t)e * p.AA, sizeof(int16_t) * p.AA, hipMemcpyDeviceToHost));
    for (int i = 0; i < p.A * p.A; i++) heightmap[i] = hm[i];
  }
  if (ems) {
    int n = sc[0];
    std::vector<unsigned char> raw((size_t)p.ems_stride * p.key_bytes);
    HIP_TRY(hipMemcpy(raw.data(), (const char*)p.ems + (size_t)e * p.ems_stride * p.key_bytes, raw.size(),
                      hipMemcpyDeviceToHost));
    int bits = p.key_bytes == 4 ? 5 : 10;
    for (int i = 0; i < n && i < cap_ems; i++) {
      uint64_t k = p.key_bytes == 4 ? (uint64_t)((uint32_t*)raw.data())[i] : ((uint64_t*)raw.data())[i];
      for (int c = 0; c < 6; c++) ems[6 * i + c] = (int32_t)((k >> (c * bits)) & ((1u << bits) - 1));
    }
  }
  if (n_ems) *n_ems = sc[0];
  if (n_boxes) *n_boxes = sc[1];
  if (next_item) { next_item[0] = sc[3]; next_item[1] = sc[4]; next_item[2] = sc[5]; }
  if (draw_cursor) *draw_cursor = (int64_t)(((uint64_t)(uint32_t)sc[9] << 32) | (uint32_t)sc[8]);
  return PCT_OK;
}

int pct_debug_state_f64(pct_env* h, int32_t e, double* ems, int32_t cap_ems, int32_t* n_ems, int32_t* n_boxes,
                        double* next_item, int64_t* draw_cursor) {
  if (!h || !h->continuous || e < 0 || e >= h->cp.N) return fail(PCT_ERR_INVALID_ARG, "bad env id / not continuous");
  int rc = use_device(h);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  const pct::ContinuousParams& c = h->cp;
  int32_t sc[PCT_SCALARS];
  HIP_TRY(hipMemcpy(sc, c.scalars + (size_t)e * PCT_SCALARS, sizeof sc, hipMemcpyDeviceToHost));
  if (ems) {
    std::vector<int32_t> raw((size_t)6 * c.ems_stride);
    HIP_TRY(hipMemcpy(raw.data(), c.ems + (size_t)e * 6 * c.ems_stride, raw.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < sc[0] && i < cap_ems; i++)
      for (int k = 0; k < 6; k++) ems[6 * i + k] = (double)raw[(size_t)k * c.ems_stride + i] / 1e6; /* the double the lattice index stands for */
  }
  if (n_ems) *n_ems = sc[0];
  if (n_boxes) *n_boxes = sc[1];
  if (next_item) { next_item[0] = sc[3] / 1000.0; next_item[1] = sc[4] / 1000.0; next_item[2] = sc[5] / 1000.0; }
  if (draw_cursor) *draw_cursor = (int64_t)(((uint64_t)(uint32_t)sc[9] << 32) | (uint32_t)sc[8]);
  return PCT_OK;
}

}  // extern "C"
