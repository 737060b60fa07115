// pct_continuous.hip -- gfx950 kernels for the batched PctContinuous0 environment.
//
// Same execution shape as pct_discrete.hip (one 64-lane wavefront = one workgroup = one env
// per transition, state staged in LDS, SoA over envs in HBM).  The reference computes in
// float64 with 1e-6 epsilons, np.around(.,6) and exact float comparisons, and its candidate
// order is the iteration order of a CPython set of float 6-tuples; to be identical this
// kernel performs THE SAME float64 operations in the same order (IEEE add/sub/mul/div/rint
// are deterministic), hashes doubles exactly like CPython, and emulates the set.  Where the
// reference rounds with np.around before comparing, the comparison is done on the integer
// lattice index rint(v*1e6), which is equivalent (v -> rint(v*1e6)/1e6 is strictly monotone
// in the index and odd) and keeps the per-candidate box scan in int32.
//
// Reference lines restated (C/ = pct_envs/PctContinuous0/):
//   step / LeafNode2Action      C/bin3D.py:151-207, wrapper/shmem_vec_env.py:139-143
//   drop_box / interSect2D      C/space.py:305-314, 329-376
//   drop_box_virtual            C/space.py:380-425
//   GENEMS / interSectEMS3D     C/space.py:441-487, Difference :490-506, IsUsableEMS :17-20
//   EliminateInscribedEMS       C/space.py:510-528
//   EMSPoint                    C/space.py:531-568
//   get_possible_position       C/bin3D.py:118-148, cur_observation :78-100
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/pct_env.h"
#include "pct_device.h"
#include "pct_set.cuh"
#include "pct_stab.cuh"
#include "pct_mt.cuh"

namespace pct {

__device__ __forceinline__ double around6(double x) { return rint(x * 1e6) / 1e6; }  // np.around(x, 6)
__device__ __forceinline__ int klat(double x) { return (int)rint(x * 1e6); }
// An EMS coordinate is always the result of np.around(., 6) (or a bin size, or 0): the double nearest to k / 1e6
// for an integer k.  The lists hold k (int32: half the LDS and HBM of the double); lat2d gives the double back
// EXACTLY -- one Newton step on k * RN(1e-6) with fused residuals equals the correctly rounded quotient for
// every |k| <= 2e8 (checked exhaustively on the host), at three instructions instead of a float64 division.
__device__ __forceinline__ double lat2d(int k) {
  const double a = (double)k, q1 = a * 1e-6;
  return fma(fma(-1e6, q1, a), 1e-6, q1);
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    double o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// Python/pyhash.c _Py_HashDouble for finite v: (M * 2^k) mod (2^61 - 1) with v = M * 2^k,
// i.e. the 53-bit mantissa rotated left by k mod 61 inside 61 bits; sign applied after; -1 -> -2.
__device__ __forceinline__ uint64_t py_hash_double(double v) {
  if (v == 0.0) return 0;
  uint64_t bits = (uint64_t)__double_as_longlong(v);
  int ef = (int)((bits >> 52) & 0x7FF);
  uint64_t M = (bits & 0xFFFFFFFFFFFFFull) | (ef ? 0x10000000000000ull : 0ull);
  int k = (ef ? ef : 1) - 1075;
  int r = k % 61;
  if (r < 0) r += 61;
  const uint64_t P = (1ull << 61) - 1;
  uint64_t x = r ? (((M << r) & P) | (M >> (61 - r))) : M;
  int64_t sx = (bits >> 63) ? -(int64_t)x : (int64_t)x;
  if (sx == -1) sx = -2;
  return (uint64_t)sx;
}
__device__ __forceinline__ uint64_t tuplehash6d(const double t[6]) {
  uint64_t acc = tuplehash_begin();
#pragma unroll
  for (int i = 0; i < 6; i++) acc = tuplehash_lane(acc, py_hash_double(t[i]));
  return tuplehash_end6(acc);
}

struct CRegs {  // wave-uniform per-env scalars
  int n_ems, n_boxes, n_leaf;
  int ik0, ik1, ik2;  // current item, lattice 1e-3
  double b0, b1, b2;  // the same as the floats the reference holds
  uint64_t cursor;
  uint32_t t;
  double volsum;
  uint32_t flags;
  int traj;
  uint32_t oc;  // observations produced so far (shuffle key)
  int mt_pos;      // strict NumPy-stream mode: position in this env's MT19937 block (LDS)
  bool mt_dirty;   // the block was regenerated in this launch
  double den_cur;  // density drawn for the current observation (setting 3, NumPy-stream mode)
  int poly_from;   // stability settings: polygon-pool vertices from this one on are newer than the HBM copy
  uint32_t stab_over;  // STAB_WHY_* bits: a stability capacity (pools, workspace, queue) was exceeded -- the step belongs to the retry pass
};

struct CLds {
  int32_t* emsk;  // [6][ems_cap] SoA current EMS list, lattice 1e-6 (see lat2d)
  int32_t* emsb;  // [6][scap] children of the step's GENEMS (aliases the idle hash table)
  uint32_t* tab;  // hash table region (aliases emsb)
  double* box;    // [6][I] lx,ly,lz,xe,ye,top (stability settings only; setting 2 keeps bk / top)
  double* bsz;    // [3][I] item sizes as placed (x,y,z): (lx + x) - lx need not equal x (stability only)
  double* top;    // [I] tops of the placed boxes (= box + 5 I under the stability settings)
  int32_t* bk;    // [4][I] lattice indices of (-lx,-ly,xe,ye)
  uint16_t* leafg;  // [L] generator ids of the current leaf nodes (the 6-tuples are recomputed from the EMS list)
  uint16_t* pend;   // [128] generator ids waiting for insertion
  uint16_t* vp;     // [64]
  uint32_t* fpri;   // [order_cap] shuffle priorities of the feasible candidates (shuffle only)
  uint32_t* dd;     // [128] bucket words of the batch de-duplication
  uint32_t* mt;     // [624] MT19937 state of this env (strict NumPy-stream mode only)
  StabState st;     // settings 1 / 3: the env's stability state, resident for the whole transition (pct_stab.cuh) ...
  StabWave sw;      // ... and the wave's hull workspace + task queue
  // two-wave candidate pipeline (p.pipe): control words, a ring of batch records, the producer's own de-duplication buckets
  struct PipeCtl* pc;
  uint64_t* ring_hash;  // [PIPE_K][64]
  uint16_t* ring_g;     // [PIPE_K][64]
  uint64_t* ring_mask;  // [PIPE_K]
  uint32_t* pdd;        // [128]
};
// ---- the two-wave candidate pipeline (round 6; pct_continuous_pipe.hip) -----------------------------------------------------
// C5 (100^3, 200 / 200): 51 KB of LDS per env = THREE one-wave workgroups per CU -- one of the four SIMDs idles, the other three
// run one latency-bound wave each -- and 550 k of the 768 k cycles of an env-step are the candidate set: per batch of 64 tuples,
// generation + CPython float hashes (82 k in all), exact in-batch de-duplication (167 k), then the insertion walks and the
// table rebuilds (185 k + 115 k), strictly in batch order.  Only the insertion depends on the table; the front half does not.
// So the env's workgroup gets a SECOND wave (on another SIMD of the same CU): it runs the pair loop, builds, hashes and
// de-duplicates batch after batch and hands {generator ids, hashes, survivor mask} over through a two-slot ring in LDS; the
// first wave takes the records in order and does what flush() does after its de-duplication.  Same batches, same order, same
// table: the result is bit for bit the one-wave kernel's.  Everything else of the step runs on the first wave as before (the
// second waits on an LDS ticket).  Hand-over = LDS words + workgroup-scope fences; inside either wave `__syncthreads()` is a
// WAVE-level LDS fence (pct_continuous_pipe.hip redefines it: the two waves are never at the same barrier).
struct PipeCtl {
  int go;      // ticket of the set phase the first wave has opened (0: none yet, -1: the workgroup is done)
  int prod;    // batch records produced in this phase
  int cons;    // ... consumed (the producer keeps at most PIPE_K ahead)
  int done;    // the producer has finished the phase (prod is final)
  int abort;   // the consumer gave up (table overflow): stop producing
  int E;       // EMS count
  int kind;    // what the ticket asks the second wave for: PIPE_JOB_*
  int a0, a1, a2;  // job arguments (GENEMS: a0 = the number of children)
  double b0, b1, b2;  // the item
  uint64_t mask;      // FEAS: the chunk's feasibility bits, back from the second wave
};
enum { PIPE_JOB_SET = 0 /* produce the candidate batches */, PIPE_JOB_ELIM = 1 /* GENEMS: the odd chunks of the children's elimination */,
       PIPE_JOB_FEAS = 2 /* the feasibility bits of one 64-candidate chunk of list(set) (first index a0, list length a1, a2 boxes) */ };
#ifndef PCT_PIPE_K
#define PCT_PIPE_K 2 /* ring depth: batches the producer may run ahead (3: 2.98 -> see profiles/r06_experiments.txt) */
#endif
constexpr int PIPE_K = PCT_PIPE_K;
constexpr size_t PIPE_CTL_BYTES = 128;
static_assert(sizeof(PipeCtl) <= PIPE_CTL_BYTES, "PipeCtl outgrew its LDS slot");
constexpr size_t PIPE_BYTES = PIPE_CTL_BYTES + (size_t)PIPE_K * (64 * 8 + 64 * 2 + 8) + 128 * 4;

// words of the region shared by the hash table and the GENEMS children scratch
__host__ __device__ __forceinline__ int cunion_words(const ContinuousParams& p) { return p.union_words; }

// LDS bytes of everything but the stability state (which follows, 16-byte aligned)
__host__ __device__ __forceinline__ size_t continuous_lds_base_bytes(const ContinuousParams& p) {
  const bool stab = p.setting != 2;
  size_t dbl = stab ? 9 * (size_t)p.I : (size_t)p.I;
  size_t i32 = 6 * (size_t)p.ems_cap + (size_t)p.union_words + 4 * (size_t)p.I + 128 + (p.rng_numpy ? 624 : 0);
  size_t u16 = 128 + 64 + (((size_t)p.L + 1) & ~(size_t)1);
  if (p.shuffle && !p.table_global) u16 += 2 * (size_t)p.order_cap;
  return ((dbl * 8 + i32 * 4 + u16 * 2 + 16 + 15) & ~(size_t)15) + (p.pipe ? PIPE_BYTES : 0);
}

__device__ __forceinline__ CLds carve(const ContinuousParams& p, unsigned char* base) {
  CLds l;
  double* d = reinterpret_cast<double*>(base);
  const bool stab = p.setting != 2;
  l.box = d; d += stab ? 6 * p.I : 0;
  l.bsz = d; d += stab ? 3 * p.I : 0;
  l.top = stab ? l.box + 5 * p.I : d; d += stab ? 0 : p.I;
  int32_t* q = reinterpret_cast<int32_t*>(d);
  l.emsk = q; q += 6 * p.ems_cap;
  l.emsb = q;
  l.tab = reinterpret_cast<uint32_t*>(q);
  q += p.union_words;
  l.bk = q; q += 4 * p.I;
  l.dd = reinterpret_cast<uint32_t*>(q); q += 128;
  l.mt = reinterpret_cast<uint32_t*>(q); q += p.rng_numpy ? 624 : 0;
  uint16_t* h = reinterpret_cast<uint16_t*>(q);
  l.pend = h; h += 128;
  l.vp = h; h += 64;
  l.leafg = h; h += (p.L + 1) & ~1;
  l.fpri = reinterpret_cast<uint32_t*>(h);
  l.pc = nullptr; l.ring_hash = nullptr; l.ring_g = nullptr; l.ring_mask = nullptr; l.pdd = nullptr;
  if (p.pipe) {  // (the last PIPE_BYTES of the base region)
    unsigned char* pb = base + continuous_lds_base_bytes(p) - PIPE_BYTES;
    l.pc = reinterpret_cast<PipeCtl*>(pb); pb += PIPE_CTL_BYTES;
    l.ring_hash = reinterpret_cast<uint64_t*>(pb); pb += (size_t)PIPE_K * 64 * 8;
    l.ring_mask = reinterpret_cast<uint64_t*>(pb); pb += (size_t)PIPE_K * 8;
    l.ring_g = reinterpret_cast<uint16_t*>(pb); pb += (size_t)PIPE_K * 64 * 2;
    l.pdd = reinterpret_cast<uint32_t*>(pb);
  }
  if (stab) {
    unsigned char* sbase = base + continuous_lds_base_bytes(p);
    l.st = stab_carve(sbase, p.I, p.sb.caps);
    l.sw = stab_wave_carve(sbase + ((stab_state_bytes(p.I, p.sb.caps) + 15) & ~(size_t)15), p.sb.caps);
  }
  return l;
}

#if !defined(PCT_CONT_MT) && !defined(PCT_CONT_PIPE)  // (the translation unit that owns the non-template symbols)
size_t continuous_lds_bytes(const ContinuousParams& p) {
  size_t b = PCT_LDS_STASH + continuous_lds_base_bytes(p);
  if (p.setting != 2) b += ((stab_state_bytes(p.I, p.sb.caps) + 15) & ~(size_t)15) + stab_wave_bytes(p.sb.caps);
  return b;
}
#endif

__device__ __forceinline__ void cdraw_item(const ContinuousParams& p, int e, CRegs& r) {
  uint64_t c = r.cursor++;
  if (p.source == PCT_ITEMS_DATASET) {  // binCreator.py:64-72; sizes round(.,3) (C/bin3D.py:85)
    int t = r.traj < p.ds_ntraj ? r.traj : p.ds_ntraj - 1;
    int len = p.ds_len[t];
    if (c < (uint64_t)len) {
      const int32_t* it = p.stream + ((size_t)t * p.ds_maxlen + (size_t)c) * 3;
      r.ik0 = it[0]; r.ik1 = it[1]; r.ik2 = it[2];
    } else {
      int v = (c == (uint64_t)len) ? 100000 : 10000;
      r.ik0 = v; r.ik1 = v; r.ik2 = v;
    }
  } else if (p.source == PCT_ITEMS_STREAM) {
    const int32_t* it = p.stream + ((size_t)e * (size_t)p.T + (size_t)(c % (uint64_t)p.T)) * 3;
    r.ik0 = it[0]; r.ik1 = it[1]; r.ik2 = it[2];
  } else {
    uint64_t g = (uint64_t)(p.env_id_base + e);
    if (p.sample_right <= 0) {
      // not sample_from_distribution: RandomBoxCreator(item_set) (C/bin3D.py:36-39,113; binCreator.py:37-39)
      const int32_t* it = p.item_set + 3 * (size_t)pct_pick(p.seed, g, c, (uint32_t)p.n_items);
      r.ik0 = it[0]; r.ik1 = it[1]; r.ik2 = it[2];
    } else {
      uint64_t span = (uint64_t)(p.sample_right - p.sample_left + 1);
      r.ik0 = p.sample_left + (int)(pct_pick(p.seed, g, c * 3 + 0, (uint32_t)span));
      r.ik1 = p.sample_left + (int)(pct_pick(p.seed, g, c * 3 + 1, (uint32_t)span));
      r.ik2 = p.sample_left + (int)(pct_pick(p.seed, g, c * 3 + 2, (uint32_t)span));
      // C/bin3D.py:110-112: settings 1 and 3 take z from np.random.choice([0.1, 0.2, 0.3, 0.4, 0.5])
      if (p.setting != 2) r.ik2 = 100 * (1 + (int)pct_pick(p.seed, g, c * 3 + 2, 5u));
    }
  }
  // round(U(a,b), 3) (C/bin3D.py:106-108): the double nearest to k/1000
  r.b0 = (double)r.ik0 / 1000.0; r.b1 = (double)r.ik1 / 1000.0; r.b2 = (double)r.ik2 / 1000.0;
}

// ---- strict NumPy-stream mode (include/pct_env.h pct_set_numpy_rng) --------------------------------------------
// Python's round(x, 3) of a positive double as the lattice index k (the result is the double nearest k/1000):
// correctly rounded on the exact binary value of x, ties to even.  x = m 2^e2 is compared with the midpoints
// (2k +- 1)/2000 in integers: m 2000 < 2^64.
__device__ __forceinline__ int cmp_x_mid(double x, long long twok1) {  // sign of x * 2000 - twok1
  if (twok1 <= 0) return 1;
  const uint64_t bits = (uint64_t)__double_as_longlong(x);
  const int e2 = (int)((bits >> 52) & 0x7FFu) - 1075;
  const uint64_t A = ((bits & 0xFFFFFFFFFFFFFull) | (1ull << 52)) * 2000ull, B = (uint64_t)twok1;
  if (e2 >= 0) return 1;
  const int sft = -e2;
  if (sft >= 64) return -1;
  const uint64_t hi = A >> sft;
  if (hi != B) return hi > B ? 1 : -1;
  return (A & ((1ull << sft) - 1ull)) ? 1 : 0;
}
__device__ __forceinline__ int round3_lattice(double x) {
  long long k = (long long)(x * 1000.0 + 0.5);
  for (int it = 0; it < 3; it++) {
    const int up = cmp_x_mid(x, 2 * k + 1), dn = cmp_x_mid(x, 2 * k - 1);
    if (up > 0 || (up == 0 && (k & 1))) { k++; continue; }
    if (dn < 0 || (dn == 0 && (k & 1))) { k--; continue; }
    break;
  }
  return (int)k;
}
// C/bin3D.py:103-113 gen_next_box, sampling mode: round(np.random.uniform(a, b), 3) (legacy uniform: a + (b - a) *
// random_sample(), no contraction), and np.random.choice of five heights (-> randint(0, 5)) under settings 1 / 3;
// then cur_observation's density draw (:88-90)
__device__ __forceinline__ void cdraw_item_mt(const ContinuousParams& p, CLds& l, CRegs& r, int lane) {
  r.cursor++;
  const double a = (double)p.sample_left / 1000.0, b = (double)p.sample_right / 1000.0;
  const double span = __dsub_rn(b, a);
  int k[3];
  const int nu = p.setting == 2 ? 3 : 2;
  for (int d = 0; d < nu; d++) k[d] = round3_lattice(__dadd_rn(a, __dmul_rn(span, mt_double(l, r, lane))));
  if (nu == 2) k[2] = 100 * (1 + (int)mt_interval(l, r, lane, 4u));
  r.ik0 = k[0]; r.ik1 = k[1]; r.ik2 = k[2];
  r.b0 = (double)r.ik0 / 1000.0; r.b1 = (double)r.ik1 / 1000.0; r.b2 = (double)r.ik2 / 1000.0;
  if (p.setting == 3) r.den_cur = mt_density(l, r, lane);
}
// box_creator.generate_box_size() (C/bin3D.py:73,202; binCreator.py:37-39): a randint over the item set that the
// sampling mode never reads
__device__ __forceinline__ void cskip_creator_mt(const ContinuousParams& p, CLds& l, CRegs& r, int lane) {
  (void)mt_interval(l, r, lane, (uint32_t)p.np_items - 1u);
}

// C/space.py:281-303 reset
__device__ __forceinline__ void cspace_reset(const ContinuousParams& p, CLds& l, CRegs& r, int lane) {
  if (lane == 0) {
    l.emsk[0 * p.ems_cap] = 0; l.emsk[1 * p.ems_cap] = 0; l.emsk[2 * p.ems_cap] = 0;
    l.emsk[3 * p.ems_cap] = klat(p.W); l.emsk[4 * p.ems_cap] = klat(p.Ly); l.emsk[5 * p.ems_cap] = klat(p.H);
  }
  r.n_ems = 1;
  r.n_boxes = 0;
  r.volsum = 0.0;
  l.st.n_ent = 0;
  l.st.n_poly = 0;
  r.poly_from = 0;
  if (p.source == PCT_ITEMS_DATASET) {
    r.traj++;
    r.cursor = 0;
    if (r.traj >= p.ds_ntraj) r.flags |= PCT_FLAG_DATASET_EXHAUSTED;
  }
}

// rotation `rot` of the item (C/space.py:537-557): extents and the skip rule (abs < 1e-6)
__device__ __forceinline__ bool crot_size(const CRegs& r, int rot, double& sx, double& sy, double& sz) {
  switch (rot) {
    case 0: sx = r.b0; sy = r.b1; sz = r.b2; return false;
    case 1: sx = r.b1; sy = r.b0; sz = r.b2; return fabs(sx - sy) < 1e-6;
    case 2: sx = r.b0; sy = r.b2; sz = r.b1; return fabs(sx - sy) < 1e-6 && fabs(sy - sz) < 1e-6;
    case 3: sx = r.b1; sy = r.b2; sz = r.b0; return fabs(sx - sy) < 1e-6 && fabs(sy - sz) < 1e-6;
    case 4: sx = r.b2; sy = r.b0; sz = r.b1; return fabs(sx - sy) < 1e-6;
    default: sx = r.b2; sy = r.b1; sz = r.b0; return fabs(sx - sy) < 1e-6;
  }
}

// The 6-tuple a generator id stands for: g = (ems * orient + rot) * 4 + corner (:560-563)
__device__ __forceinline__ void cand_tuple(const ContinuousParams& p, const CLds& l, const CRegs& r, int orient, uint32_t g,
                                  double t[6]) {
  int corner = (int)(g & 3u);
  int q = (int)(g >> 2);
  int ei = q / orient, rot = q - ei * orient;
  double sx, sy, sz;
  crot_size(r, rot, sx, sy, sz);
  double x0 = lat2d(l.emsk[0 * p.ems_cap + ei]), y0 = lat2d(l.emsk[1 * p.ems_cap + ei]), z0 = lat2d(l.emsk[2 * p.ems_cap + ei]);
  double x1 = lat2d(l.emsk[3 * p.ems_cap + ei]), y1 = lat2d(l.emsk[4 * p.ems_cap + ei]);
  if (corner & 1) { t[0] = x1 - sx; t[3] = x1; } else { t[0] = x0; t[3] = x0 + sx; }
  if (corner & 2) { t[1] = y1 - sy; t[4] = y1; } else { t[1] = y0; t[4] = y0 + sy; }
  t[2] = z0;
  t[5] = z0 + sz;
}
__device__ __forceinline__ bool tuple_eq(const double a[6], const double b[6]) {
  return (a[0] == b[0]) & (a[1] == b[1]) & (a[2] == b[2]) & (a[3] == b[3]) & (a[4] == b[4]) & (a[5] == b[5]);
}
// table word of a key: 15-bit fingerprint of its hash | 16-bit generator id (tag bit clear)
__device__ __forceinline__ uint32_t cword(uint64_t hash, uint32_t g) { return (uint32_t)((hash >> 40) & 0x7FFFu) << 16 | g; }

struct CGeo {  // placed-box geometry for the stability code
  const double* box;
  const double* bsz;
  int I;
  static constexpr bool kSquareIsPow = false;  // (pct_stab.cuh: the lever rule's `tri_base_len ** 2` is glibc's pow)
  __device__ __forceinline__ void operator()(int i, double g[9]) const {
#pragma unroll
    for (int c = 0; c < 6; c++) g[c] = box[c * I + i];
    g[6] = bsz[0 * I + i]; g[7] = bsz[1 * I + i]; g[8] = bsz[2 * I + i];
  }
};
// pipe-side helpers: relaxed LDS words both waves poll (volatile: one ds_read per look), handed over under workgroup-scope fences
__device__ __forceinline__ int pipe_ld(const int* w) { return __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int*>(w)); }
__device__ __forceinline__ void pipe_st(int* w, int v) { *reinterpret_cast<volatile int*>(w) = v; }
// first wave: open a job for the second wave (its arguments are already in the control block) / wait until it has finished it
__device__ __forceinline__ void pipe_post(PipeCtl* pc, int kind, int lane) {
  if (lane == 0) { pipe_st(&pc->kind, kind); pipe_st(&pc->done, 0); }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) pipe_st(&pc->go, pipe_ld(&pc->go) + 1);
}
__device__ __forceinline__ void pipe_join(PipeCtl* pc) {
  while (!pipe_ld(&pc->done)) __builtin_amdgcn_s_sleep(2);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// GENEMS, "which children survive": the chunks first, first + stride, ... of the C children in l.emsb against the survivors (smask) of the
// E EMS in l.emsk and against each other; a chunk's verdicts go to its two kmask words.  Chunks are independent of each other
// (everything they read is final), so the two waves of a pipeline workgroup take every other one.
__device__ __forceinline__ void cgenems_elim(const ContinuousParams& p, CLds& l, int lane, int E, int C, int first, int stride) {
  const int cap = p.ems_cap, scap = p.union_words / 6;
  uint32_t* const smask = l.dd;
  uint32_t* const kmask = l.dd + 64;
  for (int base = 64 * first; base < C; base += 64 * stride) {
    int i = base + lane;
    bool live = i < C;
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
    if (live) {
      a0 = l.emsb[0 * scap + i]; a1 = l.emsb[1 * scap + i]; a2 = l.emsb[2 * scap + i];
      a3 = l.emsb[3 * scap + i]; a4 = l.emsb[4 * scap + i]; a5 = l.emsb[5 * scap + i];
    }
    uint64_t mk;
    if (C + E <= 128) {
      // Short lists: each test is `a inside b` for a wave-uniform b: the 64 b of a chunk sit one per lane in registers and reach
      // the comparison through readlane (scalar operands) -- no LDS round trip inside the pair loop.
      bool del = false;
      for (int jb = 0; jb < E; jb += 64) {
        const int jj = jb + lane;
        const bool jl = jj < E;
        const int b0 = jl ? l.emsk[0 * cap + jj] : 0, b1 = jl ? l.emsk[1 * cap + jj] : 0, b2 = jl ? l.emsk[2 * cap + jj] : 0;
        const int b3 = jl ? l.emsk[3 * cap + jj] : 0, b4 = jl ? l.emsk[4 * cap + jj] : 0, b5 = jl ? l.emsk[5 * cap + jj] : 0;
        uint64_t sm = ((uint64_t)smask[(jb >> 6) * 2 + 1] << 32) | smask[(jb >> 6) * 2];  // survivors only (wave-uniform)
        while (sm) {
          const int j = __ffsll((unsigned long long)sm) - 1;
          sm &= sm - 1;
          const int c0 = __builtin_amdgcn_readlane(b0, j), c1 = __builtin_amdgcn_readlane(b1, j), c2 = __builtin_amdgcn_readlane(b2, j);
          const int c3 = __builtin_amdgcn_readlane(b3, j), c4 = __builtin_amdgcn_readlane(b4, j), c5 = __builtin_amdgcn_readlane(b5, j);
          del |= (a0 >= c0) & (a1 >= c1) & (a2 >= c2) & (a3 <= c3) & (a4 <= c4) & (a5 <= c5);
        }
      }
      for (int jb = 0; jb < C; jb += 64) {
        const int jj = jb + lane;
        const bool jl = jj < C;
        const int b0 = jl ? l.emsb[0 * scap + jj] : 0, b1 = jl ? l.emsb[1 * scap + jj] : 0, b2 = jl ? l.emsb[2 * scap + jj] : 0;
        const int b3 = jl ? l.emsb[3 * scap + jj] : 0, b4 = jl ? l.emsb[4 * scap + jj] : 0, b5 = jl ? l.emsb[5 * scap + jj] : 0;
        const int n = C - jb < 64 ? C - jb : 64;
        for (int j = 0; j < n; j++) {
          const int c0 = __builtin_amdgcn_readlane(b0, j), c1 = __builtin_amdgcn_readlane(b1, j), c2 = __builtin_amdgcn_readlane(b2, j);
          const int c3 = __builtin_amdgcn_readlane(b3, j), c4 = __builtin_amdgcn_readlane(b4, j), c5 = __builtin_amdgcn_readlane(b5, j);
          const bool inside = (a0 >= c0) & (a1 >= c1) & (a2 >= c2) & (a3 <= c3) & (a4 <= c4) & (a5 <= c5);
          del |= inside & (jb + j != i);
        }
      }
      mk = __ballot(live && !del);
    } else {
      // Long lists (C5): `a inside b`, one child a at a time against 64 b held one per lane: a's coordinates come out of the chunk's registers
      // through readlane (scalar operands, no LDS round trip in the pair loop), and a child STOPS at its first container.
      // The other children go first: at C5 two of three children are deleted, 93 % of those by another child (the children
      // of overlapping parents nest) -- half a scan on average instead of the whole list.
      uint64_t pend = __ballot(live);  // children of this chunk not (yet) found inside another box
      auto sweep = [&](bool jl, int jidx, int b0, int b1, int b2, int b3, int b4, int b5) __attribute__((always_inline)) {
        uint64_t todo = pend;
        while (todo) {
          const int ai = __ffsll((unsigned long long)todo) - 1;
          todo &= todo - 1;
          const int c0 = __builtin_amdgcn_readlane(a0, ai), c1 = __builtin_amdgcn_readlane(a1, ai), c2 = __builtin_amdgcn_readlane(a2, ai);
          const int c3 = __builtin_amdgcn_readlane(a3, ai), c4 = __builtin_amdgcn_readlane(a4, ai), c5 = __builtin_amdgcn_readlane(a5, ai);
          const bool cont = jl & (c0 >= b0) & (c1 >= b1) & (c2 >= b2) & (c3 <= b3) & (c4 <= b4) & (c5 <= b5) & (jidx != base + ai);
          if (__ballot(cont)) pend &= ~(1ull << ai);
        }
      };
      for (int jb = 0; jb < C && pend; jb += 64) {
        const int jj = jb + lane;
        const bool jl = jj < C;
        const int b0 = jl ? l.emsb[0 * scap + jj] : 0, b1 = jl ? l.emsb[1 * scap + jj] : 0, b2 = jl ? l.emsb[2 * scap + jj] : 0;
        const int b3 = jl ? l.emsb[3 * scap + jj] : 0, b4 = jl ? l.emsb[4 * scap + jj] : 0, b5 = jl ? l.emsb[5 * scap + jj] : 0;
        sweep(jl, jj, b0, b1, b2, b3, b4, b5);
      }
      for (int jb = 0; jb < E && pend; jb += 64) {
        const int jj = jb + lane;
        const uint64_t sm = ((uint64_t)smask[(jb >> 6) * 2 + 1] << 32) | smask[(jb >> 6) * 2];  // survivors only
        const bool jl = (sm >> lane) & 1ull;
        const int b0 = jl ? l.emsk[0 * cap + jj] : 0, b1 = jl ? l.emsk[1 * cap + jj] : 0, b2 = jl ? l.emsk[2 * cap + jj] : 0;
        const int b3 = jl ? l.emsk[3 * cap + jj] : 0, b4 = jl ? l.emsk[4 * cap + jj] : 0, b5 = jl ? l.emsk[5 * cap + jj] : 0;
        sweep(jl, -1, b0, b1, b2, b3, b4, b5);
      }
      mk = pend;
    }
    if (lane == 0) { kmask[(base >> 6) * 2] = (uint32_t)mk; kmask[(base >> 6) * 2 + 1] = (uint32_t)(mk >> 32); }
  }
}

// C/space.py:441-487 GENEMS + :510-528 EliminateInscribedEMS.  l.emsk -> l.emsk.
__device__ __forceinline__ void cgenems(const ContinuousParams& p, CLds& l, CRegs& r, int lane, const double loc[6], bool helper = false) {
  // Survivors (EMS the box does not intersect) stay where they are in l.emsk until the end; only the
  // children go to the scratch list (l.emsb, [6][scap], aliasing the idle hash table).  The pre-GENEMS
  // list is containment-free and a child lies inside its parent, so a survivor can neither be deleted
  // nor sit inside a child: each child is tested against the survivors and the other children (non-strict,
  // on the pre-deletion list: identical children delete each other).  The reference compares the float64
  // coordinates; they are all of the form lat2d(k), strictly increasing in k, so the lattice integers compare
  // the same way.  The intersection and usability tests keep the reference's float64 arithmetic.
  const int E = r.n_ems, cap = p.ems_cap, scap = p.union_words / 6;
  const double lb = p.low_bound;
  const double n0 = -loc[0], n1 = -loc[1], n2 = -loc[2];
  uint32_t* const smask = l.dd;       // [2 per 64-EMS chunk] survivor bits (the de-duplication buckets are idle)
  uint32_t* const kmask = l.dd + 64;  // [2 per 64-child chunk] children that survive the elimination
  int C = 0;
  for (int base = 0; base < E; base += 64) {
    int i = base + lane;
    bool live = i < E;
    int k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0;
    if (live) {
      k0 = l.emsk[0 * cap + i]; k1 = l.emsk[1 * cap + i]; k2 = l.emsk[2 * cap + i];
      k3 = l.emsk[3 * cap + i]; k4 = l.emsk[4 * cap + i]; k5 = l.emsk[5 * cap + i];
    }
    const double x1 = lat2d(k0), y1 = lat2d(k1), z1 = lat2d(k2), x2 = lat2d(k3), y2 = lat2d(k4), z2 = lat2d(k5);
    // np.around(np.minimum(item, EMS), 6): the rounded values decide (and give the child coordinates)
    double q0 = around6(fmin(n0, -x1)), q1 = around6(fmin(n1, -y1)), q2 = around6(fmin(n2, -z1));
    double q3 = around6(fmin(loc[3], x2)), q4 = around6(fmin(loc[4], y2)), q5 = around6(fmin(loc[5], z2));
    bool inter = live && (q0 + q3 > 0) && (q1 + q4 > 0) && (q2 + q5 > 0);
    uint64_t ms = __ballot(live && !inter);
    if (lane == 0) { smask[(base >> 6) * 2] = (uint32_t)ms; smask[(base >> 6) * 2 + 1] = (uint32_t)(ms >> 32); }
    double x3 = -q0, y3 = -q1, x4 = q3, y4 = q4, z4 = q5;  // intersect[:, 0:3] *= -1
    bool uy = (y2 - y1 + 1e-6 >= lb), uz = (z2 - z1 + 1e-6 >= lb), ux = (x2 - x1 + 1e-6 >= lb);
    bool c0 = inter && (x3 - x1 + 1e-6 >= lb) && uy && uz;  // [x1,y1,z1,x3,y2,z2]
    bool c1 = inter && (x2 - x4 + 1e-6 >= lb) && uy && uz;  // [x4,y1,z1,x2,y2,z2]
    bool c2 = inter && ux && (y3 - y1 + 1e-6 >= lb) && uz;  // [x1,y1,z1,x2,y3,z2]
    bool c3 = inter && ux && (y2 - y4 + 1e-6 >= lb) && uz;  // [x1,y4,z1,x2,y2,z2]
    bool c4 = inter && ux && uy && (z2 - z4 + 1e-6 >= lb);  // [x1,y1,z4,x2,y2,z2]
    uint64_t m0 = __ballot(c0), m1 = __ballot(c1), m2 = __ballot(c2), m3 = __ballot(c3), m4 = __ballot(c4);
    int pos = C + rank_below(m0) + rank_below(m1) + rank_below(m2) + rank_below(m3) + rank_below(m4);
    const int kx3 = klat(x3), ky3 = klat(y3), kx4 = klat(x4), ky4 = klat(y4), kz4 = klat(z4);  // the rounded values' own k
#define PCT_PUT(A, B, Cc, D, Ee, F)                                                                \
  do {                                                                                            \
    if (pos < scap) {                                                                             \
      l.emsb[0 * scap + pos] = (A); l.emsb[1 * scap + pos] = (B); l.emsb[2 * scap + pos] = (Cc);  \
      l.emsb[3 * scap + pos] = (D); l.emsb[4 * scap + pos] = (Ee); l.emsb[5 * scap + pos] = (F);  \
    }                                                                                             \
    pos++;                                                                                        \
  } while (0)
    if (c0) PCT_PUT(k0, k1, k2, kx3, k4, k5);
    if (c1) PCT_PUT(kx4, k1, k2, k3, k4, k5);
    if (c2) PCT_PUT(k0, k1, k2, k3, ky3, k5);
    if (c3) PCT_PUT(k0, ky4, k2, k3, k4, k5);
    if (c4) PCT_PUT(k0, k1, kz4, k3, k4, k5);
#undef PCT_PUT
    C += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3) + __popcll(m4);
  }
  if (C > scap) {
    C = scap;
    r.flags |= PCT_FLAG_EMS_OVERFLOW;
  }
  __syncthreads();
  // which children survive (cgenems_elim); with a second wave at hand it takes the odd chunks
  if (helper && C > 64) {
    if (lane == 0) { l.pc->a0 = C; l.pc->E = E; }
    pipe_post(l.pc, PIPE_JOB_ELIM, lane);
    cgenems_elim(p, l, lane, E, C, 0, 2);
    pipe_join(l.pc);
  } else {
    cgenems_elim(p, l, lane, E, C, 0, 1);
  }
  __syncthreads();
  // survivors close ranks in place (a chunk is read whole before it is written, leftwards) ...
  int out = 0;
  for (int base = 0; base < E; base += 64) {
    int i = base + lane;
    const uint64_t ms = ((uint64_t)smask[(base >> 6) * 2 + 1] << 32) | smask[(base >> 6) * 2];
    bool surv = (ms >> lane) & 1ull;
    int e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0;
    if (surv) {
      e0 = l.emsk[0 * cap + i]; e1 = l.emsk[1 * cap + i]; e2 = l.emsk[2 * cap + i];
      e3 = l.emsk[3 * cap + i]; e4 = l.emsk[4 * cap + i]; e5 = l.emsk[5 * cap + i];
    }
    __syncthreads();
    if (surv) {
      int o = out + rank_below(ms);
      l.emsk[0 * cap + o] = e0; l.emsk[1 * cap + o] = e1; l.emsk[2 * cap + o] = e2;
      l.emsk[3 * cap + o] = e3; l.emsk[4 * cap + o] = e4; l.emsk[5 * cap + o] = e5;
    }
    out += __popcll(ms);
    __syncthreads();
  }
  // ... and the surviving children follow, in order
  bool over = false;
  for (int base = 0; base < C; base += 64) {
    int i = base + lane;
    const uint64_t mk = ((uint64_t)kmask[(base >> 6) * 2 + 1] << 32) | kmask[(base >> 6) * 2];
    bool keep = (mk >> lane) & 1ull;
    if (keep) {
      int o = out + rank_below(mk);
      if (o < cap) {
        for (int c = 0; c < 6; c++) l.emsk[c * cap + o] = l.emsb[c * scap + i];
      } else {
        over = true;
      }
    }
    out += __popcll(mk);
  }
  if (__ballot(over) || out > cap) {  // the list that survives must fit the state array
    out = out > cap ? cap : out;
    r.flags |= PCT_FLAG_EMS_OVERFLOW;
  }
  __syncthreads();
  l.dd[lane] = 0xFFFFFFFFu;       // hand the bucket words back to the de-duplication in their idle state
  l.dd[lane + 64] = 0xFFFFFFFFu;
  r.n_ems = out;
  __syncthreads();
}

// C/space.py:531-568 EMSPoint (CPython set order over float tuples) + C/bin3D.py:118-148
// MT: strict NumPy-stream mode (np.random.shuffle drawn from the env's MT19937).  COUNT_ONLY (MT): the observation
// a failed step builds and discards (C/bin3D.py:183) -- only its draws matter: the set is built for its size, the
// shuffle's draws are consumed, nothing is written.

// drop_box_virtual of one 64-candidate chunk of list(set) (C/space.py:380-425), setting 2: lane = candidate order[base + lane]; returns the
// chunk's feasibility bits.  (cleaf_nodes' `feasible` without the stability part -- the pipeline kernels are setting 2 -- as a function
// of its own, so that both waves of a pipeline workgroup can run it on alternate chunks.)
__device__ __forceinline__ uint64_t cfeas_chunk(const ContinuousParams& p, const CLds& l, const CRegs& r, int lane, const uint16_t* order,
                                                  int base, int norder, int nb, int orient) {
  const int i = base + lane;
  const bool live = i < norder;
  const uint32_t gid = live ? (uint32_t)order[i] : 0u;
  double t[6];
  cand_tuple(p, l, r, orient, gid, t);
  const double lx = t[0], ly = t[1];
  const double x = t[3] - t[0], y = t[4] - t[1], z = t[5] - t[2];
  bool ok = live;
  if (lx + x - 1e-6 > p.W || ly + y - 1e-6 > p.Ly) ok = false;
  if (lx + 1e-6 < 0 || ly + 1e-6 < 0) ok = false;
  const int c0 = klat(-lx), c1 = klat(-ly), c2 = klat(lx + x), c3 = klat(ly + y);
  double max_h = 0.0;
  if (live)
    for (int b2 = 0; b2 < nb; b2++) {
      const int u0 = l.bk[0 * p.I + b2], u1 = l.bk[1 * p.I + b2], u2 = l.bk[2 * p.I + b2], u3 = l.bk[3 * p.I + b2];
      const bool ov = (min(c0, u0) + min(c2, u2) > 0) && (min(c1, u1) + min(c3, u3) > 0);
      const double top = l.top[b2];
      max_h = (ov && top > max_h) ? top : max_h;
    }
  if (max_h + z - 1e-6 > p.H) ok = false;
  return __ballot(ok);
}

// The producer wave of the candidate pipeline: the pair loop, the pending queue and the front half of flush() -- tuples, CPython
// hashes, exact in-batch de-duplication -- of cleaf_nodes below, batch by batch into the ring.
__device__ __forceinline__ void cpipe_produce(const ContinuousParams& p, CLds& l, int lane) {
  PipeCtl* const pc = l.pc;
  CRegs r;
  r.b0 = pc->b0; r.b1 = pc->b1; r.b2 = pc->b2;
  const int E = pipe_ld(&pc->E), cap = p.ems_cap;
  const int orient = (p.setting == 2) ? 6 : 2;
  const int NP = E * orient;
  int npend = 0, nprod = 0;
  bool stop = false;
  l.pdd[lane] = 0xFFFFFFFFu;
  l.pdd[lane + 64] = 0xFFFFFFFFu;
  __syncthreads();
  auto emit = [&](int cnt) __attribute__((always_inline)) {
    bool pending = lane < cnt;
    const uint32_t g = pending ? (uint32_t)l.pend[lane] : 0u;
    const uint16_t mv = (lane + 64 < npend) ? l.pend[lane + 64] : (uint16_t)0;
    __syncthreads();
    if (lane + 64 < npend) l.pend[lane] = mv;
    npend -= cnt;
    double t[6];
    cand_tuple(p, l, r, orient, g, t);
    const uint64_t hash = tuplehash6d(t);
    __syncthreads();
    pending = pending && !batch_find_duplicates_t6<128>(l.pdd, pending, hash, t, lane);
    // a free slot of the ring (the consumer frees one as soon as it has the record in registers)
    while (nprod - pipe_ld(&pc->cons) >= PIPE_K) {
      if (pipe_ld(&pc->abort)) { stop = true; return; }
      __builtin_amdgcn_s_sleep(2);
    }
    const int slot = nprod % PIPE_K;
    l.ring_hash[slot * 64 + lane] = hash;
    l.ring_g[slot * 64 + lane] = (uint16_t)g;
    const uint64_t pm = __ballot(pending);
    if (lane == 0) l.ring_mask[slot] = pm;
    nprod++;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) pipe_st(&pc->prod, nprod);
  };
  for (int pbase = 0; pbase < NP && !stop; pbase += 64) {
    const int q = pbase + lane;
    bool pv = q < NP;
    const int ei = q / orient, rot = q - ei * orient;
    double sx, sy, sz;
    const bool skip = crot_size(r, rot, sx, sy, sz);
    if (pv) {
      const double x0 = lat2d(l.emsk[0 * cap + ei]), y0 = lat2d(l.emsk[1 * cap + ei]), z0 = lat2d(l.emsk[2 * cap + ei]);
      const double x1 = lat2d(l.emsk[3 * cap + ei]), y1 = lat2d(l.emsk[4 * cap + ei]), z1 = lat2d(l.emsk[5 * cap + ei]);
      pv = !skip && (x1 - x0 + 1e-6 >= sx) && (y1 - y0 + 1e-6 >= sy) && (z1 - z0 + 1e-6 >= sz);
    }
    const uint64_t pm = __ballot(pv);
    const int nt = 4 * __popcll(pm);
    if (pv) l.vp[rank_below(pm)] = (uint16_t)q;
    __syncthreads();
    for (int tb = 0; tb < nt && !stop; tb += 64) {
      const int tt = tb + lane;
      const bool valid = tt < nt;
      const uint32_t g = valid ? ((uint32_t)l.vp[tt >> 2] << 2 | (uint32_t)(tt & 3)) : 0u;
      const uint64_t nm = __ballot(valid);
      if (valid) l.pend[npend + rank_below(nm)] = (uint16_t)g;
      npend += __popcll(nm);
      __syncthreads();
      if (npend >= 64) emit(64);
    }
    __syncthreads();
  }
  while (npend > 0 && !stop) emit(npend < 64 ? npend : 64);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) pipe_st(&pc->done, 1);
}

template <bool GT, bool STAB, bool MT, bool COUNT_ONLY, typename TM, bool PIPE = false>
__device__ __forceinline__ bool cleaf_nodes(const ContinuousParams& p, int e, CLds& l, CRegs& r, int lane, TM& tm, bool helper = PIPE) {
  const int E = r.n_ems, cap = p.ems_cap;
  const int orient = (p.setting == 2) ? 6 : 2;
  const int NP = E * orient;
  const uint32_t EMPTY = SlotWord<uint32_t>::EMPTY;
  uint32_t size = 8, fill = 0;
  // table and list(set) order: LDS, or this env's HBM slice when the capacity does not fit
  const size_t gslot = p.gt_by_block ? (size_t)blockIdx.x : (size_t)e;
  uint32_t* const tabs = GT ? p.gtab + gslot * (size_t)(p.cand_cap + p.cand_cap / 4) : l.tab;
  // list(set) as 16-bit generator ids: HBM slice, or -- LDS table -- written over the front of the table
  // region itself once the table is complete (entry k is written after slot k has been read)
  uint16_t* const order = GT ? p.gorder + gslot * (size_t)p.order_cap : reinterpret_cast<uint16_t*>(l.tab);
  // LDS table: every size starts at offset 0 (a rebuild first lifts the old table into registers); HBM table:
  // two regions, ping-pong
  auto region = [&](uint32_t sz) __attribute__((always_inline)) -> uint32_t { return GT ? table_region(p.cand_cap, sz) : 0u; };
  uint32_t toff = region(size);
  if (lane < 8) tab_st<GT, uint32_t>(&tabs[toff + lane], EMPTY);
  l.dd[lane] = 0xFFFFFFFFu;
  l.dd[lane + 64] = 0xFFFFFFFFu;
  __syncthreads();
  bool cand_overflow = false;
  int npend = 0;
  tm.sub_start();

  // the back half of a batch: the de-duplicated tuples (generator id g, hash, coordinates t) enter the set, in lane order
  auto consume = [&](bool pending, uint32_t g, uint64_t hash, const double (&t)[6]) __attribute__((always_inline)) {
    if (fill == 0 && size == 8 && p.cand_cap >= 512) {
      const uint64_t pm0 = __ballot(pending);
      if (__popcll(pm0) >= 19) {
        // Fast start of a fresh set with >= 19 new keys, as in the discrete kernel
        // (pct_discrete_impl.cuh): the 8- and 32-slot tables are replayed on the scalar unit, the
        // 128-slot table receives the 19 keys in 32-table slot order and then the rest of the batch
        // in one pass.  Generator ids and hashes travel through the (now idle) de-duplication arrays.
        uint64_t rem = pm0;
        int t8 = 0xFF, t32 = 0xFF;
        uint32_t occ8 = 0, occ32 = 0;
        auto lane_hash = [&](int src) __attribute__((always_inline)) -> uint64_t {
          uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)hash, src);
          uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(hash >> 32), src);
          return ((uint64_t)hi << 32) | lo;
        };
        for (int o = 0; o < 5; o++) {
          const int src = __ffsll((unsigned long long)rem) - 1;
          rem &= rem - 1;
          const uint64_t h = lane_hash(src);
          uint32_t i = (uint32_t)h & 7u;
          uint64_t perturb = h;
          while ((occ8 >> i) & 1u) {
            perturb >>= 5;
            i = (i * 5u + 1u + (uint32_t)perturb) & 7u;
          }
          occ8 |= 1u << i;
          t8 = lane == (int)i ? src : t8;
        }
        auto insert32 = [&](int src) __attribute__((always_inline)) {
          const uint64_t h = lane_hash(src);
          uint32_t i = (uint32_t)h & 31u;
          uint64_t perturb = h;
          while (true) {
            const uint32_t span = (i + 9u <= 31u) ? 10u : 1u;
            const uint32_t w = (~occ32 >> i) & ((1u << span) - 1u);
            if (w) {
              i += (uint32_t)__ffs((int)w) - 1u;
              break;
            }
            perturb >>= 5;
            i = (i * 5u + 1u + (uint32_t)perturb) & 31u;
          }
          occ32 |= 1u << i;
          t32 = lane == (int)i ? src : t32;
        };
        for (uint32_t m8 = occ8; m8; m8 &= m8 - 1u) insert32(__builtin_amdgcn_readlane(t8, __ffs((int)m8) - 1));
        for (int o = 5; o < 19; o++) {
          const int src = __ffsll((unsigned long long)rem) - 1;
          rem &= rem - 1;
          insert32(src);
        }
        const uint32_t g_s = (uint32_t)__shfl((int)g, t32 & 63, 64);
        const uint64_t h_s = shfl_key<uint64_t>(hash, t32 & 63);
        const bool in32 = lane < 32 && ((occ32 >> lane) & 1u);
        const bool later = (rem >> lane) & 1ull;
        __syncthreads();
        // the reordered batch travels through LDS: hashes in the (idle) bucket words, ids behind the 128 slots
        // of the new table in the table region (LDS table) or in the pair list's upper half... the ids need
        // 64 x 2 bytes: the 32 words after the table's first 128
        uint64_t* const sh = reinterpret_cast<uint64_t*>(l.dd);
        uint16_t* const sg = reinterpret_cast<uint16_t*>(l.tab + 128);
        if (in32) { sg[rank_below((uint64_t)occ32)] = (uint16_t)g_s; sh[rank_below((uint64_t)occ32)] = h_s; }
        if (later) { sg[19 + rank_below(rem)] = (uint16_t)g; sh[19 + rank_below(rem)] = hash; }
        const int total = 19 + __popcll(rem);
        const uint32_t noff = region(128u);
        tab_st<GT, uint32_t>(&tabs[noff + lane], EMPTY);
        tab_st<GT, uint32_t>(&tabs[noff + 64 + lane], EMPTY);
        __syncthreads();
        const uint32_t mg = lane < total ? sg[lane] : 0u;
        const uint64_t mh = lane < total ? sh[lane] : 0ull;
        __syncthreads();
        l.dd[lane] = 0xFFFFFFFFu;  // back to the all-ones state the de-duplication expects
        l.dd[lane + 64] = 0xFFFFFFFFu;
        __syncthreads();
        bool mplaced;
        uint32_t mslot;
        pyset_match<uint32_t, GT>(tabs + noff, 127u, lane < total, mh, lane, false, mplaced, mslot, [&](uint32_t) { return false; });
        if (mplaced) tab_st<GT, uint32_t>(&tabs[noff + mslot], cword(mh, mg));
        toff = noff;
        size = 128;
        fill = (uint32_t)total;
        pending = false;
        __syncthreads();
        tm.sub_tick(PH_SET_MATCH);
      }
    }
    const uint32_t word = cword(hash, g);
    auto same = [&](uint32_t w) __attribute__((always_inline)) -> bool {
      if ((w >> 16) != (word >> 16)) return false;  // different hash
      double o[6];
      cand_tuple(p, l, r, orient, w & 0xFFFFu, o);
      return tuple_eq(o, t);
    };
    while (true) {
      uint64_t pm = __ballot(pending);
      if (!pm) break;
      uint32_t mask = size - 1;
      uint32_t thr = (mask * 3u + 4u) / 5u;
      bool part = pending && (uint32_t)rank_below(pm) < thr - fill;
      bool placed;
      uint32_t slot;
      pyset_match<uint32_t, GT>(tabs + toff, mask, part, hash, lane, true, placed, slot, same);
      if (placed) tab_st<GT, uint32_t>(&tabs[toff + slot], word);
      pending = pending && !part;
      fill += (uint32_t)__popcll(__ballot(placed));
      __syncthreads();
      tm.sub_tick(PH_SET_MATCH);
      if (fill >= thr) {
        uint32_t newsize = 8;
        while (newsize <= fill * 4u) newsize <<= 1;
        if (newsize > (uint32_t)p.cand_cap) {
          cand_overflow = true;
          break;
        }
        const uint32_t noff = region(newsize);
        // Old slots arrive sparse (at most 3/5 of a table are occupied): they are queued, in slot order, in the (idle)
        // de-duplication words and go into the new table 64 at a time -- 60 % of the matching passes of a
        // chunk-by-chunk re-insertion, each of which recomputes and re-hashes its tuples.
        int nq = 0;
        auto reinsert = [&](uint32_t ow, uint32_t off) __attribute__((always_inline)) {  // up to 64 old entries into the new table at `off`
          bool opart = ow != EMPTY;
          double o[6];
          cand_tuple(p, l, r, orient, opart ? (ow & 0xFFFFu) : 0u, o);
          bool oplaced;
          uint32_t oslot;
          pyset_match<uint32_t, GT>(tabs + off, newsize - 1, opart, tuplehash6d(o), lane, false, oplaced, oslot,
                                    [&](uint32_t) { return false; });
          if (oplaced) tab_st<GT, uint32_t>(&tabs[off + oslot], ow);
          __syncthreads();
        };
        auto push = [&](uint32_t ow, uint32_t off) __attribute__((always_inline)) {  // one chunk of old slots
          const uint64_t m = __ballot(ow != EMPTY);
          if (ow != EMPTY) l.dd[nq + rank_below(m)] = ow;
          nq += __popcll(m);
          __syncthreads();
          if (nq >= 64) {
            const uint32_t w = l.dd[lane];
            const uint32_t mv = l.dd[lane + 64];
            __syncthreads();
            l.dd[lane] = mv;
            nq -= 64;
            __syncthreads();
            reinsert(w, off);
          }
        };
        auto drain = [&](uint32_t off) __attribute__((always_inline)) {
          if (nq > 0) {
            const uint32_t w = lane < nq ? l.dd[lane] : EMPTY;
            __syncthreads();
            reinsert(w, off);
          }
          l.dd[lane] = 0xFFFFFFFFu;  // back to the all-ones state the de-duplication expects
          l.dd[lane + 64] = 0xFFFFFFFFu;
          __syncthreads();
        };
        if (!GT && size <= 512) {
          // same LDS region: lift the old table (<= 512 slots = 8 words per lane) into registers, wipe, re-insert
          uint32_t oldw[8];
#pragma unroll
          for (int c = 0; c < 8; c++) {
            const uint32_t s2 = (uint32_t)c * 64u + lane;
            oldw[c] = (s2 < size) ? tabs[toff + s2] : EMPTY;
          }
          __syncthreads();
          uint32_t nchunks = (size + 63u) / 64u;
          if (size == 512u) {
            // 512 -> 2048: squeeze the 8 sparse chunks into 5 dense ones (slot order kept), through the old table's
            // own words, as the discrete kernel does
            uint32_t base = 0;
#pragma unroll
            for (int c = 0; c < 8; c++) {
              const uint64_t m = __ballot(oldw[c] != EMPTY);
              if (oldw[c] != EMPTY) tabs[noff + base + (uint32_t)rank_below(m)] = oldw[c];
              base += (uint32_t)__popcll(m);
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 8; c++) {
              const uint32_t q2 = (uint32_t)c * 64u + lane;
              oldw[c] = q2 < base ? tabs[noff + q2] : EMPTY;
            }
            nchunks = (base + 63u) / 64u;
            __syncthreads();
          }
          for (uint32_t s2 = lane; s2 < newsize; s2 += 64) tabs[noff + s2] = EMPTY;
          __syncthreads();
#pragma unroll
          for (int c = 0; c < 8; c++)
            if ((uint32_t)c < nchunks) reinsert(oldw[c], noff);
        } else if (!GT && p.gpark) {
          // an LDS table of 2048 slots growing to 8192 in the SAME region (round 5: 32 KB instead of 40, three 100^3 envs per CU
          // instead of two): the old entries -- at most 1229 -- are parked, dense and in slot order, in this env's HBM row (coalesced
          // stores, agent-scope so that the same wave reads them back), the region is wiped, and they return 64 at a time
          uint32_t* const park = p.gpark + (size_t)e * PCT_PARK_WORDS;
          int nold = 0;
          for (uint32_t sb = 0; sb < size; sb += 64) {
            const uint32_t ow = tabs[toff + sb + lane];
            const uint64_t m = __ballot(ow != EMPTY);
            if (ow != EMPTY) tab_st<true, uint32_t>(&park[nold + rank_below(m)], ow);
            nold += __popcll(m);
          }
          __syncthreads();
          for (uint32_t s2 = lane; s2 < newsize; s2 += 64) tabs[s2] = EMPTY;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          __syncthreads();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          for (int ob = 0; ob < nold; ob += 64) reinsert(ob + lane < nold ? tab_ld<true, uint32_t>(&park[ob + lane]) : EMPTY, 0u);
          toff = 0;
          size = newsize;
          tm.sub_tick(PH_SET_REBUILD);
          continue;
        } else if (!GT) {
          // (an LDS table beyond 512 slots grows through the parking row above; pct_create allocates it whenever such a table
          // can exist.  Rounds 1-4 kept a second LDS region behind the old table here: the union no longer has room for one, so a
          // handle without the row sends the env to the retry pass rather than write past the LDS union -- ADVICE r5)
          cand_overflow = true;
          break;
        } else {
          for (uint32_t s2 = lane; s2 < newsize; s2 += 64) tab_st<GT, uint32_t>(&tabs[noff + s2], EMPTY);
          __syncthreads();
          for (uint32_t sb = 0; sb < size; sb += 64) {
            uint32_t s2 = sb + lane;
            push((s2 < size) ? tab_ld<GT, uint32_t>(&tabs[toff + s2]) : EMPTY, noff);
          }
          drain(noff);
        }
        toff = noff;
        size = newsize;
        tm.sub_tick(PH_SET_REBUILD);
      }
    }
  };

  auto flush = [&](int cnt) __attribute__((always_inline)) {
    bool pending = lane < cnt;
    uint32_t g = pending ? (uint32_t)l.pend[lane] : 0u;
    uint16_t mv = (lane + 64 < npend) ? l.pend[lane + 64] : (uint16_t)0;
    __syncthreads();
    if (lane + 64 < npend) l.pend[lane] = mv;
    npend -= cnt;
    tm.sub_tick(PH_SET_GEN);
    double t[6];
    cand_tuple(p, l, r, orient, g, t);
    uint64_t hash = tuplehash6d(t);
    __syncthreads();
    // exact in-batch de-duplication (first occurrence stays): the first holder's tuple comes over cross-lane reads
    pending = pending && !batch_find_duplicates_t6<128>(l.dd, pending, hash, t, lane);
    tm.sub_tick(PH_SET_DEDUP);
    consume(pending, g, hash, t);
  };

  if (PIPE && helper) {
    // consumer side of the two-wave pipeline: open the phase for the producer wave, then take its batch records in order
    PipeCtl* const pc = l.pc;
    if (lane == 0) {
      pc->E = E; pc->b0 = r.b0; pc->b1 = r.b1; pc->b2 = r.b2;
      pipe_st(&pc->prod, 0); pipe_st(&pc->cons, 0); pipe_st(&pc->abort, 0);
    }
    pipe_post(pc, PIPE_JOB_SET, lane);
    int ncons = 0;
    while (true) {
      int pr;
      while (true) {
        pr = pipe_ld(&pc->prod);
        if (pr > ncons) break;
        if (pipe_ld(&pc->done)) { pr = pipe_ld(&pc->prod); break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (pr <= ncons) break;  // the producer is through and every record has been taken
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int slot = ncons % PIPE_K;
      const uint64_t hash = l.ring_hash[slot * 64 + lane];
      const uint32_t g = (uint32_t)l.ring_g[slot * 64 + lane];
      const uint64_t pm = l.ring_mask[slot];
      ncons++;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // (the record is in registers: the slot is free)
      if (lane == 0) pipe_st(&pc->cons, ncons);
      tm.add(ST_GENERATED, (uint64_t)__popcll(pm));
      double t[6];
      cand_tuple(p, l, r, orient, g, t);  // (the coordinates again, for the equality tests of the insertion)
      consume((pm >> lane) & 1ull, g, hash, t);
      if (cand_overflow) break;
    }
    if (cand_overflow) {  // stop the producer and wait until it has left the phase (it reads the EMS list)
      if (lane == 0) pipe_st(&pc->abort, 1);
      while (!pipe_ld(&pc->done)) __builtin_amdgcn_s_sleep(2);
    }
  } else {
  for (int pbase = 0; pbase < NP && !cand_overflow; pbase += 64) {
    int q = pbase + lane;
    bool pv = q < NP;
    int ei = q / orient, rot = q - ei * orient;
    double sx, sy, sz;
    bool skip = crot_size(r, rot, sx, sy, sz);
    if (pv) {
      double x0 = lat2d(l.emsk[0 * cap + ei]), y0 = lat2d(l.emsk[1 * cap + ei]), z0 = lat2d(l.emsk[2 * cap + ei]);
      double x1 = lat2d(l.emsk[3 * cap + ei]), y1 = lat2d(l.emsk[4 * cap + ei]), z1 = lat2d(l.emsk[5 * cap + ei]);
      pv = !skip && (x1 - x0 + 1e-6 >= sx) && (y1 - y0 + 1e-6 >= sy) && (z1 - z0 + 1e-6 >= sz);
    }
    uint64_t pm = __ballot(pv);
    const int nt = 4 * __popcll(pm);
    tm.add(ST_GENERATED, (uint64_t)nt);
    if (pv) l.vp[rank_below(pm)] = (uint16_t)q;
    __syncthreads();
    for (int tb = 0; tb < nt && !cand_overflow; tb += 64) {
      int tt = tb + lane;
      bool valid = tt < nt;
      uint32_t g = valid ? ((uint32_t)l.vp[tt >> 2] << 2 | (uint32_t)(tt & 3)) : 0u;
      // no separate membership pass: every generated tuple is queued, and one that is already in the set ends
      // its matching walk at its own entry (check_found in the flush) -- one probe walk per tuple instead of two
      bool fresh = valid;
      uint64_t nm = __ballot(fresh);
      if (fresh) l.pend[npend + rank_below(nm)] = (uint16_t)g;
      npend += __popcll(nm);
      __syncthreads();
      if (npend >= 64) flush(64);
    }
    __syncthreads();
  }
  while (npend > 0 && !cand_overflow) flush(npend < 64 ? npend : 64);
  }
  if (cand_overflow) {
    // the table outgrew this launch's capacity: hand the env to the large-capacity pass
    // (state untouched) if there is one, else record the overflow
    if (p.retry_ids != nullptr && !p.retry_mode) return true;
    r.flags |= PCT_FLAG_CANDIDATE_OVERFLOW;
  }
  __syncthreads();
  if (MT && COUNT_ONLY) {
    if (p.shuffle)
      for (int i = (int)fill - 1; i >= 1; i--) (void)mt_interval(l, r, lane, (uint32_t)i);
    return false;
  }

  // list(set): generator ids in slot order
  int norder = 0;
  for (uint32_t sb = 0; sb < size; sb += 64) {
    uint32_t s = sb + lane;
    uint32_t w = (s < size) ? tab_ld<GT, uint32_t>(&tabs[toff + s]) : EMPTY;
    uint64_t m = __ballot(w != EMPTY);
    if (w != EMPTY) tab_st<GT, uint16_t>(&order[norder + rank_below(m)], (uint16_t)(w & 0xFFFFu));
    norder += __popcll(m);
  }
  __syncthreads();
  tm.add(ST_EMS, (uint64_t)E);
  tm.add(ST_DISTINCT, (uint64_t)fill);
  tm.sub_tick(PH_SET_GEN);
  tm.tick(PH_SET);

  // feasibility in list order (C/space.py:380-425 drop_box_virtual), first L kept
  int nleaf = 0;
  uint32_t stab_err = 0;
  const int nb = r.n_boxes;
  const double next_den = !STAB ? 1.0 : (MT ? (p.setting == 3 ? r.den_cur : 1.0) : next_density(p, e, r.oc, r.traj, r.cursor - 1));  // C/bin3D.py:81-90
  // ALL 64 lanes call (`live`: this lane holds a candidate): the stability check of the lanes that need one is a
  // wave-cooperative task walk (pct_stab.cuh stab_virtual_wave)
  bool stab_ill = false;
  StabStats sstats = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (timed build only)
  bool unknown = false;  // the last call left this lane's candidate undecided (a capacity of its own was exceeded)
  auto feasible = [&](bool live, const double t[6]) __attribute__((always_inline)) -> bool {
    unknown = false;
    double lx = t[0], ly = t[1];
    double x = t[3] - t[0], y = t[4] - t[1], z = t[5] - t[2];
    bool ok = live;
    if (lx + x - 1e-6 > p.W || ly + y - 1e-6 > p.Ly) ok = false;
    if (lx + 1e-6 < 0 || ly + 1e-6 < 0) ok = false;
    // interSect2D (:305-314) on lattice indices; tops stay float64
    int c0 = klat(-lx), c1 = klat(-ly), c2 = klat(lx + x), c3 = klat(ly + y);
    double max_h = 0.0;
    if (live)
      for (int b2 = 0; b2 < nb; b2++) {
        int u0 = l.bk[0 * p.I + b2], u1 = l.bk[1 * p.I + b2], u2 = l.bk[2 * p.I + b2], u3 = l.bk[3 * p.I + b2];
        bool ov = (min(c0, u0) + min(c2, u2) > 0) && (min(c1, u1) + min(c3, u3) > 0);
        double top = l.top[b2];
        max_h = (ov && top > max_h) ? top : max_h;
      }
    if (max_h + z - 1e-6 > p.H) ok = false;
    if (STAB) {  // C/space.py:432-439 calculated_impact_virtual(True)
      const bool need = ok && !(fabs(max_h) < 1e-6);
      if (__ballot(need)) {
        const double cand[9] = {lx, ly, max_h, lx + x, ly + y, max_h + z, x, y, z};
        CGeo geo{l.box, l.bsz, p.I};
        uint32_t cap = 0;
        bool ill = false, lerr = false;
        const bool stable = stab_virtual_wave<true>(geo, l.st, nb, need, cand, next_den, l.sw, lane, cap, lerr, ill,
                                                    TM::on ? &sstats : nullptr);
        if (need) ok = stable && !cap;
        stab_err |= cap;
        unknown = need && lerr;
        if (__ballot(ill)) stab_ill = true;
      }
    }
    return ok;
  };
  if (MT && p.shuffle) {
    // C/bin3D.py:126-127 np.random.shuffle(allPostion), draw for draw: Fisher-Yates from the back with
    // j = random_interval(i) (legacy RandomState.shuffle) on the list(set) order; the plain sweep below follows
    for (int i = norder - 1; i >= 1; i--) {
      const int j = (int)mt_interval(l, r, lane, (uint32_t)i);
      if (j != i && lane == 0) {
        const uint16_t a = tab_ld<GT, uint16_t>(&order[i]), b = tab_ld<GT, uint16_t>(&order[j]);
        tab_st<GT, uint16_t>(&order[i], b);
        tab_st<GT, uint16_t>(&order[j], a);
      }
    }
    __syncthreads();
  }
  if (!MT && p.shuffle) {
    // C/bin3D.py:126-127 np.random.shuffle -> pct_shuffle_priority: every candidate is tested,
    // the feasible ones are ranked by (priority, list index), the first L ranks are kept
    uint32_t* const fpri = GT ? p.gfpri + gslot * (size_t)p.order_cap : l.fpri;
    int nf = 0;
    for (int base = 0; base < norder; base += 64) {
      int i = base + lane;
      bool live = i < norder;
      uint32_t g = live ? (uint32_t)tab_ld<GT, uint16_t>(&order[i]) : 0u;
      double t[6];
      cand_tuple(p, l, r, orient, g, t);
      bool ok = feasible(live, t);
      if (STAB && __ballot(unknown)) stab_err |= STAB_WHY_SPLIT;
      uint64_t m = __ballot(ok);
      int o = nf + rank_below(m);
      __syncthreads();
      if (ok) {
        tab_st<GT, uint16_t>(&order[o], (uint16_t)g);  // in-place compaction (o <= i)
        tab_st<GT, uint32_t>(&fpri[o], pct_shuffle_priority(p.shuffle_seed, (uint64_t)(p.env_id_base + e), (uint64_t)r.oc, (uint32_t)i));
      }
      nf += __popcll(m);
      __syncthreads();
    }
    for (int base = 0; base < nf; base += 64) {
      int a2 = base + lane;
      bool live = a2 < nf;
      uint32_t pa = live ? tab_ld<GT, uint32_t>(&fpri[a2]) : 0u;
      int rank = 0;
      for (int j = 0; j < nf; j++) {
        uint32_t pj = tab_ld<GT, uint32_t>(&fpri[j]);
        rank += (pj < pa || (pj == pa && j < a2)) ? 1 : 0;
      }
      if (live && rank < p.L) l.leafg[rank] = tab_ld<GT, uint16_t>(&order[a2]);
    }
    nleaf = nf;
  } else if (PIPE && !GT && !STAB && helper) {
    // two chunks of list(set) at a time: the second wave tests the odd one while this wave tests the even one; merged in list order
    PipeCtl* const pc = l.pc;
    for (int base = 0; base < norder && nleaf < p.L; base += 128) {
      const bool two = base + 64 < norder;
      if (two) {
        if (lane == 0) { pc->a0 = base + 64; pc->a1 = norder; pc->a2 = nb; pc->b0 = r.b0; pc->b1 = r.b1; pc->b2 = r.b2; }
        pipe_post(pc, PIPE_JOB_FEAS, lane);
      }
      const uint64_t m0 = cfeas_chunk(p, l, r, lane, order, base, norder, nb, orient);
      {
        const int i = base + lane;
        const int idx = nleaf + rank_below(m0);
        if (((m0 >> lane) & 1ull) && idx < p.L) l.leafg[idx] = order[i];
        nleaf += __popcll(m0);
      }
      if (two) {
        pipe_join(pc);
        const uint64_t m1 = pc->mask;
        if (nleaf < p.L) {
          const int i = base + 64 + lane;
          const int idx = nleaf + rank_below(m1);
          if (((m1 >> lane) & 1ull) && idx < p.L) l.leafg[idx] = order[i];
          nleaf += __popcll(m1);
        }
      }
    }
  } else {
    for (int base = 0; base < norder && nleaf < p.L; base += 64) {
      int i = base + lane;
      bool live = i < norder;
      double t[6];
      const uint32_t gid = live ? (uint32_t)tab_ld<GT, uint16_t>(&order[i]) : 0u;
      cand_tuple(p, l, r, orient, gid, t);
      bool ok = feasible(live, t);
      uint64_t m = __ballot(ok);
      int idx = nleaf + rank_below(m);
      if (STAB && __ballot(unknown && idx < p.L)) stab_err |= STAB_WHY_SPLIT;  // (beyond the L-th feasible one the reference never looks)
      if (ok && idx < p.L) l.leafg[idx] = (uint16_t)gid;
      nleaf += __popcll(m);
    }
  }
  r.oc++;
  if (STAB) r.stab_over |= stab_err;  // (wave-uniform)
  if (STAB && stab_ill) r.flags |= PCT_FLAG_ILL_CONDITIONED;
  r.n_leaf = nleaf < p.L ? nleaf : p.L;
  if (TM::on && STAB) {
    tm.add(ST_STAB_VPASSES, (uint64_t)__builtin_amdgcn_readfirstlane(sstats.v_passes));
    tm.add(ST_STAB_VTASKS, (uint64_t)__builtin_amdgcn_readfirstlane(sstats.v_tasks));
    tm.add(ST_STAB_VNARROW, (uint64_t)__builtin_amdgcn_readfirstlane(sstats.v_narrow));
    tm.add(ST_STAB_LEVEL0, (uint64_t)wave_sum_i64(sstats.v_level0));
    tm.add(ST_STAB_LSQ3, (uint64_t)wave_sum_i64(sstats.lsq3));
    tm.add(ST_STAB_LSQ4, (uint64_t)wave_sum_i64(sstats.lsq4));
    tm.add(ST_STAB_LSQ5, (uint64_t)wave_sum_i64(sstats.lsq5));
    tm.add(ST_STAB_LSQX, (uint64_t)wave_sum_i64(sstats.lsqx));
    tm.add(ST_STAB_LSQ_ROUNDS, (uint64_t)__builtin_amdgcn_readfirstlane(sstats.lsq_rounds));
    tm.add(ST_STAB_LSQ_ROUNDS_L0, (uint64_t)__builtin_amdgcn_readfirstlane(sstats.lsq_rounds_l0));
    tm.add(ST_STAB_VROUNDS, (uint64_t)__builtin_amdgcn_readfirstlane(sstats.v_rounds));
    tm.add(ST_STAB_VCALLS, (uint64_t)__builtin_amdgcn_readfirstlane(sstats.v_calls));
  }
  __syncthreads();
  tm.tick(PH_FEAS);
  return false;
}

// C/bin3D.py:78-100 observation rows, float32 (envs.py:180)
// `full` rewrites every row; otherwise only what changed since this env's previous observation:
// the row of the box just placed (`new_row`, or -1), the leaf rows and the next-item row
__device__ __forceinline__ void cwrite_obs(const ContinuousParams& p, int e, const CLds& l, const CRegs& r, int lane,
                                  float* __restrict__ obs, bool full, int new_row, const double newbox[6]) {
  const float nden = p.rng_numpy ? (float)(p.setting == 3 ? r.den_cur : 1.0)
                                 : (float)next_density(p, e, r.oc - 1, r.traj, r.cursor - 1);  // C/bin3D.py:81-90,98
  const int orient = (p.setting == 2) ? 6 : 2;
  double a = r.b0, b = r.b1, c = r.b2, tmp;
  if (a > b) { tmp = a; a = b; b = tmp; }
  if (b > c) { tmp = b; b = c; c = tmp; }
  if (a > b) { tmp = a; a = b; b = tmp; }
  if (!full && new_row >= 0 && lane < 9) {
    double v = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) v = lane == k ? newbox[k] : v;
    obs_st(&obs[new_row * 9 + lane], lane < 6 ? (float)v : (lane == 8 ? 1.0f : 0.f));
  }
  if (!full) {
    // incremental: lane = leaf row (nine strided stores), then the item row -- as in the discrete kernel; a
    // leaf's 6-tuple is recomputed from its generator id (the EMS list and the item it was made from are
    // the ones in LDS / in the registers)
    for (int jb = 0; jb < p.L; jb += 64) {
      const int j = jb + lane;
      if (j < p.L) {
        const bool on = j < r.n_leaf;
        double t[6] = {0, 0, 0, 0, 0, 0};
        if (on) cand_tuple(p, l, r, orient, (uint32_t)l.leafg[j], t);
        float* o = obs + (size_t)(p.I + j) * 9;
        obs_st(o + 0, (float)t[0]); obs_st(o + 1, (float)t[1]); obs_st(o + 2, (float)t[2]); obs_st(o + 3, (float)t[3]);
        obs_st(o + 4, (float)t[4]);
        obs_st(o + 5, on ? (float)p.H : 0.f);
        obs_st(o + 6, 0.f); obs_st(o + 7, 0.f);
        obs_st(o + 8, on ? 1.0f : 0.f);
      }
    }
    if (lane < 9)
      obs_st(&obs[(size_t)(p.I + p.L) * 9 + lane],
             lane == 0 ? nden : (lane == 3 ? (float)a : (lane == 4 ? (float)b : (lane == 5 ? (float)c : (lane == 8 ? 1.0f : 0.f)))));
    return;
  }
  // full rewrite: the placed boxes come from their HBM rows (written when they were placed; the one placed by
  // this very step from `newbox`), lane = row for the box and leaf parts
  const double* gb = p.boxes + (size_t)e * 6 * p.I;
  for (int rb = 0; rb < p.I; rb += 64) {
    const int row = rb + lane;
    if (row < p.I) {
      float* o = obs + (size_t)row * 9;
      const bool on = row < r.n_boxes;
      double g[6] = {0, 0, 0, 0, 0, 0};
      if (on && row != new_row) {
#pragma unroll
        for (int k = 0; k < 6; k++) g[k] = gb[k * p.I + row];
      } else if (on) {
#pragma unroll
        for (int k = 0; k < 6; k++) g[k] = newbox[k];
      }
#pragma unroll
      for (int k = 0; k < 6; k++) obs_st(o + k, (float)g[k]);
      obs_st(o + 6, 0.f); obs_st(o + 7, 0.f);  // density column is 0 (:372-373)
      obs_st(o + 8, (on || row == 0) ? 1.0f : 0.f);  // row 0: the dummy valid node after reset (C/space.py:285-286)
    }
  }
  for (int jb = 0; jb < p.L; jb += 64) {
    const int j = jb + lane;
    if (j < p.L) {
      const bool on = j < r.n_leaf;
      double t[6] = {0, 0, 0, 0, 0, 0};
      if (on) cand_tuple(p, l, r, orient, (uint32_t)l.leafg[j], t);
      float* o = obs + (size_t)(p.I + j) * 9;
      obs_st(o + 0, (float)t[0]); obs_st(o + 1, (float)t[1]); obs_st(o + 2, (float)t[2]); obs_st(o + 3, (float)t[3]);
      obs_st(o + 4, (float)t[4]);
      obs_st(o + 5, on ? (float)p.H : 0.f);
      obs_st(o + 6, 0.f); obs_st(o + 7, 0.f);
      obs_st(o + 8, on ? 1.0f : 0.f);
    }
  }
  if (lane < 9)
    obs_st(&obs[(size_t)(p.I + p.L) * 9 + lane],
           lane == 0 ? nden : (lane == 3 ? (float)a : (lane == 4 ? (float)b : (lane == 5 ? (float)c : (lane == 8 ? 1.0f : 0.f)))));
}

// Stand-in policy epilogue (pct_bind_policy_rows), as in the discrete kernel: the float32 leaf row the separate policy
// kernel would gather from the observation just written (leaf pct_mix32(g, t) % k of the k valid ones, else the zero row)
__device__ __forceinline__ void cpolicy_epilogue(const ContinuousParams& p, int e, const CLds& l, const CRegs& r, int lane) {
  const int k = r.n_leaf;
  const int li = k > 0 ? (int)(pct_mix32((uint32_t)(p.env_id_base + e), r.t) % (uint32_t)k) : 0;
  const int orient = (p.setting == 2) ? 6 : 2;
  double t[6] = {0, 0, 0, 0, 0, 0};
  if (k > 0) cand_tuple(p, l, r, orient, (uint32_t)l.leafg[li], t);
  if (lane < 9) {
    double v = 0;
#pragma unroll
    for (int c = 0; c < 5; c++) v = lane == c ? t[c] : v;
    const bool on = k > 0;
    p.policy_rows[(size_t)e * 9 + lane] = lane < 5 ? (float)v : (lane == 5 ? (on ? (float)p.H : 0.f) : (lane == 8 ? (on ? 1.0f : 0.f) : 0.f));
  }
}

__device__ __forceinline__ void cload(const ContinuousParams& p, int e, CLds& l, CRegs& r, int lane) {
  const int32_t* sc = p.scalars + (size_t)e * PCT_SCALARS;
  r.n_ems = sc[0]; r.n_boxes = sc[1]; r.n_leaf = sc[2];
  r.ik0 = sc[3]; r.ik1 = sc[4]; r.ik2 = sc[5];
  r.b0 = (double)r.ik0 / 1000.0; r.b1 = (double)r.ik1 / 1000.0; r.b2 = (double)r.ik2 / 1000.0;
  r.t = (uint32_t)sc[6];
  r.cursor = ((uint64_t)(uint32_t)sc[9] << 32) | (uint32_t)sc[8];
  r.flags = p.flags[e];
  r.volsum = p.volsum[e];
  r.traj = sc[12];
  r.oc = (uint32_t)sc[13];
  r.mt_pos = sc[7];
  r.mt_dirty = false;
  r.den_cur = 1.0;
  r.stab_over = 0;
  r.poly_from = 0;
  if (p.setting != 2) {  // the stability state: the used part of the pools (word 15: entries | vertices << 16)
    const uint32_t pw = (uint32_t)sc[15];
    r.stab_over = stab_load(p.sb, p.I, e, r.n_boxes, (int)(pw & 0xFFFFu), (int)(pw >> 16), l.st, lane) ? 0u : STAB_WHY_LOAD;
    r.poly_from = l.st.n_poly;
  }
  if (p.rng_numpy) {
    const uint32_t* gm = p.mt + (size_t)e * 624;
    for (int i = lane; i < 624; i += 64) l.mt[i] = gm[i];
    if (p.setting == 3) r.den_cur = p.mt_den[e];
  }
  const int32_t* ge = p.ems + (size_t)e * 6 * p.ems_stride;
  const double* gb = p.boxes + (size_t)e * 6 * p.I;
  const uint16_t* gl = p.leafg + (size_t)e * p.L;
  const bool stab = p.setting != 2;
  const int n_fit = r.n_ems < p.ems_cap ? r.n_ems : p.ems_cap;  // a longer list belongs to the retry pass (caller checks)
  for (int c = 0; c < 6; c++)
    for (int i = lane; i < n_fit; i += 64) l.emsk[c * p.ems_cap + i] = ge[c * p.ems_stride + i];
  for (int i = lane; i < r.n_leaf; i += 64) l.leafg[i] = gl[i];
  // placed boxes: the footprint on the lattice and the top for the per-candidate scan; the full rows (and the
  // sizes as placed) only where the stability check needs them
  for (int i = lane; i < r.n_boxes; i += 64) {
    const double lx = gb[0 * p.I + i], ly = gb[1 * p.I + i], xe = gb[3 * p.I + i], ye = gb[4 * p.I + i], top = gb[5 * p.I + i];
    l.bk[0 * p.I + i] = klat(-lx);
    l.bk[1 * p.I + i] = klat(-ly);
    l.bk[2 * p.I + i] = klat(xe);
    l.bk[3 * p.I + i] = klat(ye);
    l.top[i] = top;
    if (stab) {
      l.box[0 * p.I + i] = lx; l.box[1 * p.I + i] = ly; l.box[2 * p.I + i] = gb[2 * p.I + i];
      l.box[3 * p.I + i] = xe; l.box[4 * p.I + i] = ye;
      const double* gs = p.bsz + (size_t)e * 3 * p.I;
      l.bsz[0 * p.I + i] = gs[0 * p.I + i]; l.bsz[1 * p.I + i] = gs[1 * p.I + i]; l.bsz[2 * p.I + i] = gs[2 * p.I + i];
    }
  }
  __syncthreads();
}

// the box just placed goes to its HBM row at once (rows are only ever appended; the row count lives in the
// scalars, which cstore writes): lane 0, six strided stores
__device__ __forceinline__ void cstore_box(const ContinuousParams& p, int e, int bi, const double g[6], const double sz[3], int lane) {
  if (lane == 0) {
    double* gb = p.boxes + (size_t)e * 6 * p.I;
#pragma unroll
    for (int k = 0; k < 6; k++) gb[k * p.I + bi] = g[k];
    if (p.setting != 2) {
      double* gs = p.bsz + (size_t)e * 3 * p.I;
      gs[0 * p.I + bi] = sz[0]; gs[1 * p.I + bi] = sz[1]; gs[2 * p.I + bi] = sz[2];
    }
  }
}

__device__ __forceinline__ void cstore(const ContinuousParams& p, int e, const CLds& l, const CRegs& r, int lane) {
  int32_t* sc = p.scalars + (size_t)e * PCT_SCALARS;
  int32_t* ge = p.ems + (size_t)e * 6 * p.ems_stride;
  uint16_t* gl = p.leafg + (size_t)e * p.L;
  for (int c = 0; c < 6; c++)
    for (int i = lane; i < r.n_ems; i += 64) ge[c * p.ems_stride + i] = l.emsk[c * p.ems_cap + i];
  for (int i = lane; i < r.n_leaf; i += 64) gl[i] = l.leafg[i];
  if (p.rng_numpy && r.mt_dirty) {
    uint32_t* gm = p.mt + (size_t)e * 624;
    for (int i = lane; i < 624; i += 64) gm[i] = l.mt[i];
  }
  if (p.setting != 2) stab_store(p.sb, p.I, e, r.n_boxes, l.st, r.poly_from, lane);
  if (lane == 0) {
    if (p.setting != 2) sc[15] = (int32_t)((uint32_t)l.st.n_ent | ((uint32_t)l.st.n_poly << 16));
    if (p.rng_numpy) {
      sc[7] = r.mt_pos;
      if (p.setting == 3) p.mt_den[e] = r.den_cur;
    }
    sc[0] = r.n_ems; sc[1] = r.n_boxes; sc[2] = r.n_leaf;
    sc[3] = r.ik0; sc[4] = r.ik1; sc[5] = r.ik2;
    sc[6] = (int32_t)r.t;
    sc[8] = (int32_t)(uint32_t)r.cursor; sc[9] = (int32_t)(uint32_t)(r.cursor >> 32);
    sc[12] = r.traj;
    sc[13] = (int32_t)r.oc;
    p.flags[e] = r.flags;
    p.volsum[e] = r.volsum;
  }
}

// C/bin3D.py:169-207 step (+ the VecEnv worker's auto-reset).  a1/a2: raw position entries of
// the action, (bx,by,bz): the item as LeafNode2Action returns it.
// returns 0: the episode goes on, 1: it ended and the env was reset, 2 (NumPy-stream mode): the discarded
// observation's candidate set outgrew this launch's table -- requeue
template <bool GT, bool STAB, bool MT, typename TM>
__device__ __forceinline__ int ctransition(const ContinuousParams& p, int e, CLds& l, CRegs& r, int lane, int flag, double a1,
                                   double a2, double bx, double by, double bz, TM& tm, double newbox[6], bool giveup = false,
                                   bool helper = false) {
  r.t++;
  const double lx = around6(a1), ly = around6(a2);  // idx = [round(action[1], 6), round(action[2], 6)]
  const double x = flag ? by : bx, y = flag ? bx : by, z = bz;  // C/space.py:330-333
  bool ok = !giveup;  // giveup: a heuristic found no placement -- the episode ends without a step()
  if (lx + x - 1e-6 > p.W || ly + y - 1e-6 > p.Ly) ok = false;
  if (lx + 1e-6 < 0 || ly + 1e-6 < 0) ok = false;
  double max_h = 0.0;
  if (ok) {
    int c0 = klat(-lx), c1 = klat(-ly), c2 = klat(lx + x), c3 = klat(ly + y);
    double m = 0.0;
    for (int b = lane; b < r.n_boxes; b += 64) {
      int u0 = l.bk[0 * p.I + b], u1 = l.bk[1 * p.I + b], u2 = l.bk[2 * p.I + b], u3 = l.bk[3 * p.I + b];
      bool ov = (min(c0, u0) + min(c2, u2) > 0) && (min(c1, u1) + min(c3, u3) > 0);
      double top = l.top[b];
      m = (ov && top > m) ? top : m;
    }
    max_h = wave_max_f64(m);
    if (max_h + z - 1e-6 > p.H) ok = false;
  }
  if (STAB && ok && r.n_boxes < p.I) {
    // check_box :432-437: box_now.calculated_impact() (or True on the floor); the box is kept
    // only if the verdict is True, so it is written beyond n_boxes first
    const int bi = r.n_boxes;
    if (lane == 0) {
      l.box[0 * p.I + bi] = lx; l.box[1 * p.I + bi] = ly; l.box[2 * p.I + bi] = max_h;
      l.box[3 * p.I + bi] = lx + x; l.box[4 * p.I + bi] = ly + y; l.box[5 * p.I + bi] = max_h + z;
      l.bsz[0 * p.I + bi] = x; l.bsz[1 * p.I + bi] = y; l.bsz[2 * p.I + bi] = z;
    }
    __syncthreads();
    CGeo geo{l.box, l.bsz, p.I};
    bool ill = false;
    const double den = MT ? (p.setting == 3 ? r.den_cur : 1.0) : next_density(p, e, r.oc - 1, r.traj, r.cursor - 1);
    // lane 0 walks; a split over six and more supporters is solved by the whole wave (pct_stab.cuh stab_commit_wave)
    StabStats cstats = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (timed build only)
    const int rc = stab_commit_wave<true>(geo, l.st, bi, den, l.sw, lane, ill, TM::on ? &cstats : nullptr);
    if (TM::on) {
      tm.add(ST_STAB_COMMIT_VISITS, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.commit_visits));
      tm.add(ST_STAB_LSQ3, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsq3));
      tm.add(ST_STAB_LSQ4, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsq4));
      tm.add(ST_STAB_LSQ5, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsq5));
      tm.add(ST_STAB_LSQX, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsqx));
      tm.add(ST_STAB_LSQ_ROUNDS, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsq_rounds));
      tm.add(ST_STAB_COMMIT_ROUNDS, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsq_rounds));
    }
    l.st.n_ent = __builtin_amdgcn_readfirstlane(l.st.n_ent);
    l.st.n_poly = __builtin_amdgcn_readfirstlane(l.st.n_poly);
    if (__builtin_amdgcn_readfirstlane(ill ? 1 : 0)) r.flags |= PCT_FLAG_ILL_CONDITIONED | PCT_FLAG_ILL_COMMIT;
    if (rc < 0) r.stab_over |= STAB_WHY_COMMIT;
    ok = rc == 1;
    __syncthreads();
  }
  if (ok && r.n_boxes >= p.I) {  // IndexError at C/space.py:371
    ok = false;
    r.flags |= PCT_FLAG_INTERNAL_OVERFLOW;
  }
  const double mx = (double)((long long)p.W * (long long)p.Ly * (long long)p.H);
  float reward;
  uint8_t done;
  int counter;
  double ratio = 0.0;
  if (ok) {
    const double top = max_h + z, xe = lx + x, ye = ly + y;
    const int bi = r.n_boxes;
    newbox[0] = lx; newbox[1] = ly; newbox[2] = max_h; newbox[3] = xe; newbox[4] = ye; newbox[5] = top;
    if (lane == 0) {
      if (STAB) {
        l.box[0 * p.I + bi] = lx; l.box[1 * p.I + bi] = ly; l.box[2 * p.I + bi] = max_h;
        l.box[3 * p.I + bi] = xe; l.box[4 * p.I + bi] = ye;
        l.bsz[0 * p.I + bi] = x; l.bsz[1 * p.I + bi] = y; l.bsz[2 * p.I + bi] = z;
      }
      l.top[bi] = top;
      l.bk[0 * p.I + bi] = klat(-lx); l.bk[1 * p.I + bi] = klat(-ly);
      l.bk[2 * p.I + bi] = klat(xe); l.bk[3 * p.I + bi] = klat(ye);
    }
    {
      const double sz[3] = {x, y, z};
      cstore_box(p, e, bi, newbox, sz, lane);
    }
    r.n_boxes++;
    r.volsum = r.volsum + x * y * z;  // get_ratio's left fold (:316-321)
    __syncthreads();
    tm.tick(PH_DROP);
    // GENEMS([lx, ly, lz, round(lx+x,6), round(ly+y,6), round(lz+z,6)]) (C/bin3D.py:190-194)
    const double loc[6] = {lx, ly, max_h, around6(lx + x), around6(ly + y), around6(max_h + z)};
    cgenems(p, l, r, lane, loc, helper);
    tm.tick(PH_GENEMS);
    reward = (float)(((r.b0 * r.b1 * r.b2) / mx) * 10);
    done = 0;
    counter = r.n_boxes;
  } else {
    reward = 0.f;
    done = 1;
    counter = r.n_boxes;
    ratio = r.volsum / mx;
    if (!giveup) r.oc++;  // the terminal step's own (discarded) observation (C/bin3D.py:183)
    __syncthreads();
    if (MT) {
      // that observation draws a NEW item, a density, and shuffles the new item's candidates on the final packing
      cdraw_item_mt(p, l, r, lane);
      if (cleaf_nodes<GT, STAB, true, true>(p, e, l, r, lane, tm)) return 2;
      __syncthreads();
    }
    cspace_reset(p, l, r, lane);
    __syncthreads();
    tm.tick(PH_DROP);
  }
  if (MT) {
    cskip_creator_mt(p, l, r, lane);  // generate_box_size() after a success (:202) / in reset() (:73)
    cdraw_item_mt(p, l, r, lane);
  } else {
    cdraw_item(p, e, r);
  }
  if (lane == 0) {
    p.reward[e] = reward;
    p.done[e] = done;
    p.counter[e] = counter;
    p.ratio[e] = ratio;
    if (p.mask) p.mask[e] = done ? 0.f : 1.f;  // train_tools.py:70 masks = 1 - done
  }
  return done != 0 ? 1 : 0;
}

#ifndef PCT_CONT_MT
// ---- heuristic.py on PackingContinuous (tools.py:217-218: LSAH, OnlineBPH, BR) as in-env policies -------------------
// (The tests hold a sequential CPU restatement of these loops, pinned on the reference's own.)
// lane = (EMS, rotation) pair in the reference's loop order; float64 in the reference's operation order.
__device__ __forceinline__ void cheur_rot(const CRegs& r, int rot, double& x, double& y, double& z) {  // heuristic.py:171-182
  switch (rot) {
    case 0: x = r.b0; y = r.b1; z = r.b2; break;
    case 1: y = r.b0; x = r.b1; z = r.b2; break;
    case 2: z = r.b0; x = r.b1; y = r.b2; break;
    case 3: z = r.b0; y = r.b1; x = r.b2; break;
    case 4: x = r.b0; z = r.b1; y = r.b2; break;
    default: y = r.b0; z = r.b1; x = r.b2; break;
  }
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    double o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    int o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}
// returns false if there is no feasible placement.  (olx, oly): position, (ox, oy, oz): the item as rotated.
template <bool STAB>
__device__ __forceinline__ bool cheur_choose(const ContinuousParams& p, int e, CLds& l, CRegs& r, int lane, int kind, double& olx,
                                    double& oly, double& ox, double& oy, double& oz) {
  const int orient = (p.setting == 2) ? 6 : 2;
  const int E = r.n_ems, NQ = E * orient, nb = r.n_boxes;
  const double den = STAB ? next_density(p, e, r.oc - 1, r.traj, r.cursor - 1) : 1.0;
  // drop_box_virtual(..., returnH=True) (C/space.py:380-425) of size (x, y, z) at (lx, ly).  ALL 64 lanes call (`go`: this
  // lane has a placement to test): the stability check is wave-cooperative
  auto probe = [&](bool go, double x, double y, double z, double lx, double ly, double& height) __attribute__((always_inline)) -> bool {
    bool ok = go;
    if (lx + x - 1e-6 > p.W || ly + y - 1e-6 > p.Ly) ok = false;
    if (lx + 1e-6 < 0 || ly + 1e-6 < 0) ok = false;
    int c0 = klat(-lx), c1 = klat(-ly), c2 = klat(lx + x), c3 = klat(ly + y);
    double max_h = 0.0;
    if (go)
      for (int b2 = 0; b2 < nb; b2++) {
        int u0 = l.bk[0 * p.I + b2], u1 = l.bk[1 * p.I + b2], u2 = l.bk[2 * p.I + b2], u3 = l.bk[3 * p.I + b2];
        bool ov = (min(c0, u0) + min(c2, u2) > 0) && (min(c1, u1) + min(c3, u3) > 0);
        double top = l.top[b2];
        max_h = (ov && top > max_h) ? top : max_h;
      }
    if (max_h + z - 1e-6 > p.H) ok = false;
    height = max_h;
    if (STAB) {
      const bool need = ok && !(fabs(max_h) < 1e-6);
      if (__ballot(need)) {
        const double cand[9] = {lx, ly, max_h, lx + x, ly + y, max_h + z, x, y, z};
        CGeo geo{l.box, l.bsz, p.I};
        uint32_t cap = 0;
        bool ill = false, lerr = false;
        const bool stable = stab_virtual_wave<true>(geo, l.st, nb, need, cand, den, l.sw, lane, cap, lerr, ill);
        if (need) ok = stable && !cap;
        r.stab_over |= cap;
        if (__ballot(need && lerr)) r.stab_over |= STAB_WHY_SPLIT;
        if (__ballot(ill)) r.flags |= PCT_FLAG_ILL_CONDITIONED;  // a near-cut rank decision inside a heuristic's probe
      }
    }
    return ok;
  };
  auto ems_of = [&](int ei, double em[6]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 6; c++) em[c] = lat2d(l.emsk[c * p.ems_cap + ei]);
  };
  bool found = false;
  if (kind == PCT_HEUR_OBPH) {
    // :364-425: EMS sorted by (z, y, x), stable; first feasible (EMS, rotation); no fit-in-EMS test.  The lattice indices
    // order like the doubles: the winner is the lexicographic minimum of (kz, ky, kx, q) over the feasible pairs
    int bz = 0x7fffffff, by = 0x7fffffff, bx = 0x7fffffff, bq = 0x7fffffff;
    for (int pass = 0; pass < 4; pass++) {
      int best = 0x7fffffff;
      for (int base = 0; base < NQ; base += 64) {
        const int q = base + lane;
        const bool live = q < NQ;
        const int ei = live ? q / orient : 0, rot = q - ei * orient;
        const int kz = l.emsk[2 * p.ems_cap + ei], ky = l.emsk[1 * p.ems_cap + ei], kx = l.emsk[0 * p.ems_cap + ei];
        // later passes only look at the pairs that tie on the earlier keys
        const bool in = live && (pass < 1 || kz == bz) && (pass < 2 || ky == by) && (pass < 3 || kx == bx);
        double x, y, z, hh, em[6];
        cheur_rot(r, rot, x, y, z);
        ems_of(ei, em);
        // (feasibility is the expensive part: it is evaluated once, in the first pass, and kept as a bit in the scratch)
        bool feas;
        if (pass == 0) {
          feas = probe(live, x, y, z, em[0], em[1], hh);
          const uint64_t fm = __ballot(feas);
          if (lane == 0) { l.tab[(base >> 6) * 2] = (uint32_t)fm; l.tab[(base >> 6) * 2 + 1] = (uint32_t)(fm >> 32); }
        } else {
          const uint64_t fm = ((uint64_t)l.tab[(base >> 6) * 2 + 1] << 32) | l.tab[(base >> 6) * 2];
          feas = (fm >> lane) & 1ull;
        }
        const int key = pass == 0 ? kz : (pass == 1 ? ky : (pass == 2 ? kx : q));
        if (in && feas && key < best) best = key;
      }
      __syncthreads();
      best = wave_min_i32(best);
      if (best == 0x7fffffff) return false;
      if (pass == 0) bz = best; else if (pass == 1) by = best; else if (pass == 2) bx = best; else bq = best;
    }
    const int ei = bq / orient, rot = bq - ei * orient;
    double em[6];
    ems_of(ei, em);
    olx = em[0]; oly = em[1];
    cheur_rot(r, rot, ox, oy, oz);
    return true;
  }
  if (kind == PCT_HEUR_BR) {
    // :500-569: the EMS with the best eval_ems (volume + item types that fit unrotated, + 10 if all do); first best in
    // (EMS, rotation) order.  Two reductions: the best score among the feasible pairs, then the first pair holding it
    double* const esc = reinterpret_cast<double*>(l.tab);  // [E] per-EMS scores (the table region is idle between observations)
    for (int base = 0; base < E; base += 64) {
      const int ei = base + lane;
      if (ei < E) {
        double em[6];
        ems_of(ei, em);
        const double dx = em[3] - em[0], dy = em[4] - em[1], dz = em[5] - em[2];
        int valid = 0;
        for (int i = 0; i < p.n_items; i++)
          valid += (dx >= p.item_set[3 * i] / 1000.0 && dy >= p.item_set[3 * i + 1] / 1000.0 && dz >= p.item_set[3 * i + 2] / 1000.0) ? 1 : 0;
        double sc = 0;
        sc += dx * dy * dz;
        sc += valid;
        if (valid == p.n_items) sc += 10;
        esc[ei] = sc;
      }
    }
    __syncthreads();
    double smax = -1e10;
    uint32_t* const fbits = l.tab + 2 * ((E + 1) & ~1) + 2;  // feasibility bits of the chunks, behind the scores
    for (int base = 0; base < NQ; base += 64) {
      const int q = base + lane;
      const bool live = q < NQ;
      const int ei = live ? q / orient : 0, rot = q - ei * orient;
      double x, y, z, hh, em[6];
      cheur_rot(r, rot, x, y, z);
      ems_of(ei, em);
      const bool fits = live && (em[3] - em[0] >= x) && (em[4] - em[1] >= y) && (em[5] - em[2] >= z);
      const bool feas = probe(fits, x, y, z, em[0], em[1], hh);
      const uint64_t fm = __ballot(feas);
      if (lane == 0) { fbits[(base >> 6) * 2] = (uint32_t)fm; fbits[(base >> 6) * 2 + 1] = (uint32_t)(fm >> 32); }
      if (feas && esc[ei] > smax) smax = esc[ei];
    }
    __syncthreads();
    smax = wave_max_f64(smax);
    int bq = 0x7fffffff;
    for (int base = 0; base < NQ; base += 64) {
      const int q = base + lane;
      const uint64_t fm = ((uint64_t)fbits[(base >> 6) * 2 + 1] << 32) | fbits[(base >> 6) * 2];
      if (q < NQ && ((fm >> lane) & 1ull) && esc[q / orient] == smax && q < bq) bq = q;
    }
    bq = wave_min_i32(bq);
    __syncthreads();
    if (bq == 0x7fffffff) return false;
    const int ei = bq / orient, rot = bq - ei * orient;
    double em[6];
    ems_of(ei, em);
    olx = em[0]; oly = em[1];
    cheur_rot(r, rot, ox, oy, oz);
    return true;
  }
  // :138-226 LASH: least surface area of the bounding box of everything packed so far.  maxXY / minXY: lx + x and lx of
  // every placed box, as the reference accumulates them (the raw float64 sums of the HBM box rows)
  double maxX = 0, maxY = 0, minX = p.W, minY = p.Ly;
  {
    const double* gb = p.boxes + (size_t)e * 6 * p.I;
    double a = 0, b = 0, c = p.W, d = p.Ly;
    for (int i = lane; i < nb; i += 64) {
      const double lx = gb[0 * p.I + i], ly = gb[1 * p.I + i], xe = gb[3 * p.I + i], ye = gb[4 * p.I + i];
      a = xe > a ? xe : a; b = ye > b ? ye : b; c = lx < c ? lx : c; d = ly < d ? ly : d;
    }
    maxX = wave_max_f64(a); maxY = wave_max_f64(b); minX = wave_min_f64(c); minY = wave_min_f64(d);
  }
  const double init = p.W * p.Ly + p.Ly * p.H + p.H * p.W;
  double* const scq = reinterpret_cast<double*>(l.tab);  // [NQ] scores (+inf: not a candidate)
  double smin = INFINITY;
  for (int base = 0; base < NQ; base += 64) {
    const int q = base + lane;
    const bool live = q < NQ;
    const int ei = live ? q / orient : 0, rot = q - ei * orient;
    double x, y, z, height, em[6];
    cheur_rot(r, rot, x, y, z);
    ems_of(ei, em);
    const double dx = em[3] - em[0], dy = em[4] - em[1], dz = em[5] - em[2];
    const double lx = em[0], ly = em[1];
    double sc = INFINITY;
    if (probe(live && dx >= x && dy >= y && dz >= z, x, y, z, lx, ly, height)) {
      const double ex = fmax(lx + x, maxX) - fmin(lx, minX), ey = fmax(ly + y, maxY) - fmin(ly, minY);
      sc = ex * ey + (height + z) * ey + (height + z) * ex;
    }
    if (live) scq[q] = sc;
    smin = sc < smin ? sc : smin;
  }
  smin = wave_min_f64(smin);
  __syncthreads();
  if (smin < init) {  // a score equal to the initial bound is never taken (:195-199 needs a best already)
    double bd0 = 0, bd1 = 0, bd2 = 0;
    for (int base = 0; base < NQ; base += 64) {
      const int q = base + lane;
      uint64_t m = __ballot(q < NQ && scq[q] == smin);
      while (m) {  // the candidates that tie on the best score, in loop order (:195-199)
        const int bit = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        const int qq = base + bit;
        const int ei = qq / orient, rot = qq - ei * orient;
        double x, y, z, em[6];
        cheur_rot(r, rot, x, y, z);
        ems_of(ei, em);
        const double dx = em[3] - em[0], dy = em[4] - em[1], dz = em[5] - em[2];
        bool take = !found;
        if (found) take = fmin(fmin(dx - x, dy - y), dz - z) < fmin(fmin(bd0 - x, bd1 - y), bd2 - z);
        if (take) {
          found = true;
          olx = em[0]; oly = em[1]; ox = x; oy = y; oz = z;
          bd0 = dx; bd1 = dy; bd2 = dz;
        }
      }
    }
  }
  __syncthreads();
  return found;
}
#endif  // !PCT_CONT_MT

// C/bin3D.py:151-167 LeafNode2Action on a float64 row (a0,a1,_,a3,a4,_)
__device__ __forceinline__ void cdecode_leaf(const CRegs& r, bool zero_row, double a0, double a1, double a3, double a4, double& p1,
                                    double& p2, double& bx, double& by, double& bz) {
  if (zero_row) {
    p1 = 0; p2 = 0; bx = r.b0; by = r.b1; bz = r.b2;
    return;
  }
  double x = around6(a3 - a0), y = around6(a4 - a1);
  const double nb[3] = {r.b0, r.b1, r.b2};
  int rec[3] = {0, 1, 2}, nr = 3;
  for (int i = 0; i < nr; i++)
    if (fabs(x - nb[rec[i]]) < 1e-6) {
      for (int j = i; j < nr - 1; j++) rec[j] = rec[j + 1];
      nr--;
      break;
    }
  for (int i = 0; i < nr; i++)
    if (fabs(y - nb[rec[i]]) < 1e-6) {
      for (int j = i; j < nr - 1; j++) rec[j] = rec[j + 1];
      nr--;
      break;
    }
  p1 = a0; p2 = a1; bx = x; by = y; bz = nb[rec[0]];
}

enum { CACT_ROWS = 0, CACT_INDEX = 1, CACT_HASH = 2, CACT_RESET = 3, CACT_HEUR = 4 /* row_len = PCT_HEUR_* */ };

template <int ACT, bool TIMED, bool GT, bool STAB, bool MT, bool PIPE = false>
#ifndef PCT_CONT_WAVES
#define PCT_CONT_WAVES 2 /* waves per SIMD the plain kernel is compiled for.  The kernel needs 217 VGPRs: at two waves per SIMD (256) nothing
                            spills; at three (168; C3's 14.2 KB of LDS would admit 11 envs per CU instead of 8) 32 VGPRs spill.  Round 4
                            shipped three: C3 +3 % (28.8 against 27.9 M env-steps/s, +9 % at 8192 envs) for 2.8 x the HBM traffic
                            (71.7 against 25.7 MB per launch = 3.7 x against 1.3 x the algorithmic bytes) -- the trade round 4 itself
                            refused on C2 (write-through stores).  Round 5 re-measured one / two / three (scripts/r05_variants.sh:
                            C3 27.94 / 27.91 / 28.76 M, C5 2.186 / 2.160 / 2.159 M) and went back to two: a launch of C3 is its
                            slowest env's chain (profiles/r05_step_profile_c3.txt), which occupancy does not shorten */
#endif
#ifndef PCT_STAB_WAVES
#define PCT_STAB_WAVES 1 /* waves per SIMD the stability-check kernels are compiled for (2: 256 VGPRs, ~470 of them spilled -- slower, profiles/r03_stability_tuning.txt) */
#endif
__global__ void __launch_bounds__(PIPE ? 128 : 64) __attribute__((amdgpu_waves_per_eu(TIMED ? 1 : (STAB ? PCT_STAB_WAVES : PCT_CONT_WAVES))))
pct_continuous_kernel(ContinuousParams p_arg, const void* actions,
                                                            int row_len, int n_steps,
                                                            const int32_t* __restrict__ env_ids, int n_ids) {
  extern __shared__ __align__(16) unsigned char smem[];
  // (pipe: two waves per env -- wave 0 steps the env, wave 1 serves the candidate pipeline of its set phases)
  const int lane = PIPE ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;
  const int wave = PIPE ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
#if PCT_KERNARG_PTR
  // the parameter block is read where it is used, from the kernarg segment (pct_device.h pct_param_fence)
  const ContinuousParams& p = *(const ContinuousParams*)pct_param_fence((PctConstParams<ContinuousParams>)__builtin_amdgcn_kernarg_segment_ptr());
#else
  const ContinuousParams& p = p_arg;
#endif
  // work items: every env (normal pass), the listed envs (reset_specific), or -- in the
  // large-capacity retry pass -- the envs the normal pass queued, grid-strided
  const bool listed = (ACT == CACT_RESET && env_ids != nullptr);
  int limit = listed ? n_ids : p.N;
  // retry pass: p.retry_count is this step's counter of a ping-pong pair; the other one (the next step's) is
  // zeroed here, so that no memset sits between the launches (retry_mode = +1 / -1: offset of the other)
  if (p.retry_mode) {
    limit = *p.retry_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      p.retry_count[p.retry_mode] = 0;
      if (limit > 0 && p.retry_total) { p.retry_total[0] += limit; p.retry_total[1] += 1; }
    }
  }
  for (int work = blockIdx.x; work < limit; work += gridDim.x) {
  // (heavy-first dispatch: p.order is a permutation of 0..N-1, see pct_device.h)
  // (readfirstlane: a loaded id would otherwise live, with every address derived from it, in vector registers)
  const int e = __builtin_amdgcn_readfirstlane(
      p.retry_mode ? p.retry_ids[work] : (listed ? env_ids[work] : ((ACT != CACT_RESET && p.order) ? p.order[work] : work)));
  if (e < 0 || e >= p.N) continue;
  work_key_begin(smem);
  CLds l = carve(p, smem + PCT_LDS_STASH);
  if (PIPE) {
    // the ticket word starts at 0 for both waves (the only real workgroup barrier of the kernel: s_barrier, not the
    // wave-level fence __syncthreads() stands for in this translation unit); the pipe launcher runs ONE env per workgroup
    if (threadIdx.x == 0) pipe_st(&l.pc->go, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (wave == 1) {
      if (p.pipe == 2 && p.scalars[(size_t)e * PCT_SCALARS] < p.pipe_min_ems) return;  // (a light env: one wave steps it)
      int last = 0;
      while (true) {
        int t;
        while ((t = pipe_ld(&l.pc->go)) == last) __builtin_amdgcn_s_sleep(4);
        if (t < 0) break;
        last = t;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int kind = pipe_ld(&l.pc->kind);
        if (kind == PIPE_JOB_ELIM) {
          cgenems_elim(p, l, lane, pipe_ld(&l.pc->E), pipe_ld(&l.pc->a0), 1, 2);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) pipe_st(&l.pc->done, 1);
        } else if (kind == PIPE_JOB_FEAS) {
          CRegs rr;
          rr.b0 = l.pc->b0; rr.b1 = l.pc->b1; rr.b2 = l.pc->b2;
          const uint64_t m = cfeas_chunk(p, l, rr, lane, reinterpret_cast<const uint16_t*>(l.tab), pipe_ld(&l.pc->a0), pipe_ld(&l.pc->a1),
                                         pipe_ld(&l.pc->a2), (p.setting == 2) ? 6 : 2);
          if (lane == 0) l.pc->mask = m;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) pipe_st(&l.pc->done, 1);
        } else {
          cpipe_produce(p, l, lane);
        }
      }
      return;
    }
  }
  CRegs r;
  PhaseTimer<TIMED> tm;
  tm.start();
  cload(p, e, l, r, lane);
  tm.tick(PH_LOAD);
  wave_priority(r.n_ems, p.prio_t);
  // (pipe = 2: the second wave has stayed only if the env entered the step with pipe_min_ems EMS -- the count both waves read)
  const bool helper = PIPE && !(p.pipe == 2 && r.n_ems < p.pipe_min_ems);
  float* obs = p.obs + (size_t)e * p.row_len;
  // an env whose EMS list outgrows this launch's LDS list -- already at load, or in this step's GENEMS -- goes,
  // state untouched, to the large-capacity pass, like one whose candidate set outgrows the table
  const bool can_retry = p.retry_ids != nullptr && !p.retry_mode;
  const uint32_t flags_in = r.flags;
  bool requeue = can_retry && ACT != CACT_RESET && (r.n_ems > p.ems_cap || r.stab_over);
  if (requeue) {
    retry_enqueue(p.retry_count, p.retry_ids, e);
    __syncthreads();
    if (PIPE) {  // release the producer wave
      if (lane == 0) pipe_st(&l.pc->go, -1);
      break;
    }
    continue;
  }

  if (ACT == CACT_RESET) {
    cspace_reset(p, l, r, lane);
    __syncthreads();
    if (MT) {
      cskip_creator_mt(p, l, r, lane);
      cdraw_item_mt(p, l, r, lane);
    } else {
      cdraw_item(p, e, r);
    }
    requeue = cleaf_nodes<GT, STAB, MT, false, PhaseTimer<TIMED>, PIPE>(p, e, l, r, lane, tm, helper);
    if (STAB && r.stab_over) {
      if (can_retry) requeue = true;
      else r.flags |= PCT_FLAG_STABILITY_OVERFLOW | r.stab_over;
    }
    if (!requeue) {
      const double nobox[6] = {0, 0, 0, 0, 0, 0};
      cwrite_obs(p, e, l, r, lane, obs, true, -1, nobox);
      if (p.policy_rows) cpolicy_epilogue(p, e, l, r, lane);
      cstore(p, e, l, r, lane);
    }
  } else {
  const int orient0 = (p.setting == 2) ? 6 : 2;
  for (int it = 0; it < n_steps; it++) {
    int flag = 0;
    double p1 = 0, p2 = 0, bx = 0, by = 0, bz = 0;
    bool giveup = false;
#ifndef PCT_CONT_MT
    if (ACT == CACT_HEUR) {
      giveup = !cheur_choose<STAB>(p, e, l, r, lane, row_len, p1, p2, bx, by, bz);
      if (!giveup) { r.b0 = bx; r.b1 = by; r.b2 = bz; }  // env.next_box = [x, y, z] (heuristic.py:190, 399, 552)
    } else
#endif
    if (ACT == CACT_ROWS) {
      const float* row = reinterpret_cast<const float*>(actions) + (size_t)e * row_len;
      float v = lane < row_len ? row[lane] : 0.f;
      double a0 = (double)__shfl(v, 0, 64), a1 = (double)__shfl(v, 1, 64), a2 = (double)__shfl(v, 2, 64),
             a3 = (double)__shfl(v, 3, 64), a4 = (double)__shfl(v, 4, 64), a5 = (double)__shfl(v, 5, 64);
      if (row_len == 3) {
        flag = (int)a0; p1 = a1; p2 = a2;
        bx = r.b0; by = r.b1; bz = r.b2;
      } else {
        double sum = ((((a0 + a1) + a2) + a3) + a4) + a5;  // np.sum(leaf_node[0:6]) == 0
        // A float32 row cannot carry the 1e-6 resolution the reference's round(.,6) decode
        // assumes (float32 spacing is 9.5e-7 above 8).  The trainer's row is the float32 cast
        // of one of this env's current leaf rows (train_tools.py:66), so it is matched back to
        // that leaf and decoded from the leaf's own float64 values; a row that matches no leaf
        // is decoded from the widened floats.
        int match = -1;
        for (int base = 0; base < r.n_leaf && match < 0; base += 64) {
          int j = base + lane;
          double t[6] = {0, 0, 0, 0, 0, 0};
          if (j < r.n_leaf) cand_tuple(p, l, r, orient0, (uint32_t)l.leafg[j], t);
          bool eq = j < r.n_leaf && (float)t[0] == (float)a0 && (float)t[1] == (float)a1 && (float)t[2] == (float)a2 &&
                    (float)t[3] == (float)a3 && (float)t[4] == (float)a4;
          uint64_t m = __ballot(eq);
          if (m) match = base + __ffsll((unsigned long long)m) - 1;
        }
        if (match >= 0 && sum != 0.0) {
          double t[6];
          cand_tuple(p, l, r, orient0, (uint32_t)l.leafg[match], t);
          a0 = t[0]; a1 = t[1]; a3 = t[3]; a4 = t[4];
        }
        cdecode_leaf(r, sum == 0.0, a0, a1, a3, a4, p1, p2, bx, by, bz);
      }
    } else {
      int64_t li;
      if (ACT == CACT_INDEX) li = reinterpret_cast<const int64_t*>(actions)[e];
      else li = r.n_leaf > 0 ? (int64_t)(pct_mix32((uint32_t)(p.env_id_base + e), r.t) % (uint32_t)r.n_leaf) : 0;
      bool zero_row = !(li >= 0 && li < r.n_leaf);
      double t[6] = {0, 0, 0, 0, 0, 0};
      if (!zero_row) cand_tuple(p, l, r, orient0, (uint32_t)l.leafg[(int)li], t);
      double a0 = t[0], a1 = t[1], a3 = t[3], a4 = t[4];
      if (!zero_row) {  // a valid leaf row sums to > 0 unless it is the all-zero row
        double sum = ((((a0 + a1) + t[2]) + a3) + a4) + p.H;
        zero_row = (sum == 0.0);
      }
      cdecode_leaf(r, zero_row, a0, a1, a3, a4, p1, p2, bx, by, bz);
    }
    double newbox[6] = {0, 0, 0, 0, 0, 0};
    const int tr = ctransition<GT, STAB, MT>(p, e, l, r, lane, flag, p1, p2, bx, by, bz, tm, newbox, giveup, helper);
    if (tr == 2) { requeue = true; break; }
    const bool ended = tr != 0;
    if (can_retry && ((r.flags & ~flags_in) & PCT_FLAG_EMS_OVERFLOW)) { requeue = true; break; }
    requeue = cleaf_nodes<GT, STAB, MT, false, PhaseTimer<TIMED>, PIPE>(p, e, l, r, lane, tm, helper);
    // a stability capacity exceeded (pools, workspace, queue -- in the commit or in a virtual check): the step goes, state
    // untouched (the stability state is LDS-resident, nothing of it has been stored), to the large-capacity pass
    if (STAB && r.stab_over) {
      if (can_retry) requeue = true;
      else { r.flags |= PCT_FLAG_STABILITY_OVERFLOW | r.stab_over; r.stab_over = 0; }
    }
    if (requeue) break;
    cwrite_obs(p, e, l, r, lane, obs, ended || (p.full_obs != 0 && it == 0), ended ? -1 : r.n_boxes - 1, newbox);
    __syncthreads();
    tm.tick(PH_OBS);
  }
  if (!requeue) {
    if (p.policy_rows) cpolicy_epilogue(p, e, l, r, lane);
    cstore(p, e, l, r, lane);
    tm.tick(PH_STORE);
    if (TIMED && lane == 0) tm.flush(p.timing + (size_t)e * PCT_TIMING_SLOTS, n_steps);
  }
  }  // step / reset
  if (requeue) retry_enqueue(p.retry_count, p.retry_ids, e);
  work_key_end(smem, p.scalars, p.N, e, ACT == CACT_RESET, r.n_ems);
  __syncthreads();
  if (PIPE) {  // release the producer wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) pipe_st(&l.pc->go, -1);
    break;
  }
  }  // work items
}

#if !defined(PCT_CONT_MT) && !defined(PCT_CONT_PIPE)
// stand-in policy kernel on the float32 observation (same as the discrete one)
__global__ void __launch_bounds__(64) pct_cpolicy_hash_rows_kernel(ContinuousParams p, float* __restrict__ rows_out) {
  const int lane = threadIdx.x;
  const int e = blockIdx.x;
  const float* obs = p.obs + (size_t)e * p.row_len;
  int k = 0;
  for (int base = 0; base < p.L; base += 64) {
    int j = base + lane;
    bool v = j < p.L && obs[(p.I + j) * 9 + 8] != 0.f;
    k += __popcll(__ballot(v));
  }
  uint32_t t = (uint32_t)p.scalars[(size_t)e * PCT_SCALARS + 6];
  int li = k > 0 ? (int)(pct_mix32((uint32_t)(p.env_id_base + e), t) % (uint32_t)k) : 0;
  if (lane < 9) rows_out[(size_t)e * 9 + lane] = obs[(p.I + li) * 9 + lane];
}

hipError_t launch_cpolicy_hash_rows(const ContinuousParams& p, float* rows_out, hipStream_t stream) {
  hipLaunchKernelGGL(pct_cpolicy_hash_rows_kernel, dim3(p.N), dim3(64), 0, stream, p, rows_out);
  return hipGetLastError();
}

#endif  // !PCT_CONT_MT

// This file is compiled twice: as is (counter-based streams), and through pct_continuous_mt.hip with PCT_CONT_MT
// defined (strict NumPy-stream mode: the MT19937 variants of the same kernels, launch_continuous_mt).
#ifdef PCT_CONT_MT
#define PCT_CONT_LAUNCH launch_continuous_mt
#define PCT_CONT_MTV true
#else
#define PCT_CONT_LAUNCH launch_continuous
#define PCT_CONT_MTV false
#endif
#ifdef PCT_CONT_PIPE
// the two-wave pipeline kernels: the normal pass of setting 2 with the table in LDS, one env per 128-thread workgroup
hipError_t launch_continuous_pipe(const ContinuousParams& p, int act, const void* actions, int row_len, int n_steps, const int32_t* env_ids,
                                  int n_ids, hipStream_t stream) {
  const size_t lds = continuous_lds_bytes(p);
  const int grid = p.N;
  if (grid <= 0) return hipSuccess;
  void (*kern)(ContinuousParams, const void*, int, int, const int32_t*, int) = nullptr;
  switch (act) {
    case CACT_ROWS: kern = pct_continuous_kernel<CACT_ROWS, false, false, false, false, true>; break;
    case CACT_INDEX: kern = pct_continuous_kernel<CACT_INDEX, false, false, false, false, true>; break;
    case CACT_HASH: kern = pct_continuous_kernel<CACT_HASH, false, false, false, false, true>; break;
    case CACT_RESET: kern = pct_continuous_kernel<CACT_RESET, false, false, false, false, true>; break;
    default: return hipErrorNotSupported;
  }
  if (lds > 48 * 1024) {
    hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (er != hipSuccess) return er;
  }
  hipExtLaunchKernelGGL(kern, dim3(grid), dim3(128), lds, stream, (hipEvent_t)p.launch_ev_start, (hipEvent_t)p.launch_ev_stop, 0, p, actions,
                        row_len, n_steps, env_ids, n_ids);
  return hipGetLastError();
}
#else
hipError_t PCT_CONT_LAUNCH(const ContinuousParams& p, int act, const void* actions, int row_len, int n_steps,
                           const int32_t* env_ids, int n_ids, hipStream_t stream) {
#ifndef PCT_CONT_MT
  if (p.rng_numpy) return launch_continuous_mt(p, act, actions, row_len, n_steps, env_ids, n_ids, stream);
  // the two-wave candidate pipeline (pct_continuous_pipe.hip): the whole-batch normal pass of setting 2 with the table in LDS
  if (p.pipe && !p.retry_mode && p.setting == 2 && !p.table_global && p.timing == nullptr && act != CACT_HEUR && !(act == CACT_RESET && env_ids))
    return launch_continuous_pipe(p, act, actions, row_len, n_steps, env_ids, n_ids, stream);
#endif
  size_t lds = continuous_lds_bytes(p);
  const bool timed = !PCT_CONT_MTV && p.timing != nullptr && act != CACT_RESET;
  const bool stab = p.setting != 2;
  int grid = p.retry_mode ? n_ids : ((act == CACT_RESET && env_ids) ? n_ids : p.N);
  if (grid <= 0) return hipSuccess;
#ifdef PCT_FEW_KERNELS
  // kernel experiments (scripts/build_variant.py): only the untimed LDS-table kernels of the action kinds the benchmark uses
  if (timed || act == CACT_HEUR || act == CACT_INDEX) return hipErrorNotSupported;
#define PCT_CKERN(A) (stab ? (p.table_global ? pct_continuous_kernel<A, false, true, true, PCT_CONT_MTV> : pct_continuous_kernel<A, false, false, true, PCT_CONT_MTV>) \
                           : (p.table_global ? pct_continuous_kernel<A, false, true, false, PCT_CONT_MTV> : pct_continuous_kernel<A, false, false, false, PCT_CONT_MTV>))
#endif
#define PCT_CLAUNCH(A)                                                                                         \
  do {                                                                                                         \
    void (*kern)(ContinuousParams, const void*, int, int, const int32_t*, int);                                \
    PCT_CLAUNCH_PICK(A)                                                                                        \
    if (lds > 48 * 1024) {                                                                                     \
      hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                 \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
      if (er != hipSuccess) return er;                                                                         \
    }                                                                                                          \
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, stream, (hipEvent_t)p.launch_ev_start,             \
                          (hipEvent_t)p.launch_ev_stop, 0, p, actions, row_len, n_steps, env_ids, n_ids); \
  } while (0)
#ifdef PCT_FEW_KERNELS
#define PCT_CLAUNCH_PICK(A) kern = PCT_CKERN(A);
#else
#define PCT_CLAUNCH_PICK(A)                                                                                    \
    if (stab && timed && A == CACT_ROWS && !p.table_global)                                                   \
      kern = pct_continuous_kernel<(A == CACT_ROWS ? A : CACT_ROWS), !PCT_CONT_MTV, false, true, PCT_CONT_MTV>; \
    else if (stab) kern = p.table_global ? pct_continuous_kernel<A, false, true, true, PCT_CONT_MTV>           \
                                    : pct_continuous_kernel<A, false, false, true, PCT_CONT_MTV>;              \
    else if (p.table_global) kern = pct_continuous_kernel<A, false, true, false, PCT_CONT_MTV>;                \
    else kern = timed ? pct_continuous_kernel<A, !PCT_CONT_MTV, false, false, PCT_CONT_MTV>                    \
                      : pct_continuous_kernel<A, false, false, false, PCT_CONT_MTV>;
#endif
  switch (act) {
    case CACT_ROWS: PCT_CLAUNCH(CACT_ROWS); break;
#ifndef PCT_FEW_KERNELS
    case CACT_INDEX: PCT_CLAUNCH(CACT_INDEX); break;
#endif
    case CACT_HASH: PCT_CLAUNCH(CACT_HASH); break;
#if !defined(PCT_CONT_MT) && !defined(PCT_FEW_KERNELS)
    case CACT_HEUR: PCT_CLAUNCH(CACT_HEUR); break;
#endif
    default: PCT_CLAUNCH(CACT_RESET); break;
  }
#undef PCT_CLAUNCH
#undef PCT_CLAUNCH_PICK
  return hipGetLastError();
}
#endif  // PCT_CONT_PIPE

}  // namespace pct
