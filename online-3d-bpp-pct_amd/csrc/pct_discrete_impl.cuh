// pct_discrete_impl.cuh -- gfx950 kernels for the batched PctDiscrete0 environment (included by
// pct_discrete.hip [32-bit keys] and pct_discrete_u64.hip [64-bit keys], which only instantiate).
//
// One 64-lane wavefront (= one 64-thread workgroup) owns one environment for a whole
// transition; the env's heightmap, EMS list, candidate hash table, placed boxes and leaf
// list are staged in LDS, the persistent state lives in HBM as struct-of-arrays over envs
// (each env's slice contiguous, so the wave's lanes read consecutive addresses).
// Integer / branchy AABB, scan and compaction work: no MFMA anywhere on this path.
//
// Reference semantics restated here (paths under the reference repo,
// D/ = pct_envs/PctDiscrete0/):
//   step / auto-reset        D/bin3D.py:151-188, wrapper/shmem_vec_env.py:139-143
//   LeafNode2Action          D/bin3D.py:139-149
//   drop_box / check_box     D/space.py:347-389, 436-454
//   GENEMS / Difference      D/space.py:457-483, 498-512
//   EliminateInscribedEMS    D/space.py:518-531
//   EMSPoint                 D/space.py:534-570  (a CPython `set`: its iteration order is
//                            reproduced exactly -- Objects/setobject.c set_add_entry /
//                            set_table_resize / set_insert_clean, tupleobject.c tuplehash)
//   get_possible_position    D/bin3D.py:100-136
//   cur_observation          D/bin3D.py:70-93
#ifndef PCT_DISCRETE_IMPL_CUH
#define PCT_DISCRETE_IMPL_CUH
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/pct_env.h"
#include "pct_device.h"

#include <type_traits>

#include "pct_set.cuh"
#include "pct_stab.cuh"
#include "pct_mt.cuh"

#ifndef PCT_SET_V
#define PCT_SET_V 1   /* tuples per lane and insertion batch of the EMS expansion (measured on MI355X: 1 is fastest -- a lone
                         wave is bound by its instruction issue, not by dependence latency, so wider batches only add
                         predicated-off work; 2 and 4 are kept for experiments) */
#endif
#ifndef PCT_STAB_WAVES
#define PCT_STAB_WAVES 1 /* waves per SIMD the stability-check kernels are compiled for (2: 256 VGPRs, ~470 of them spilled -- slower, profiles/r03_stability_tuning.txt) */
#endif
#ifndef PCT_SET_CARRY
#define PCT_SET_CARRY 1 /* 1: insertion batches of exactly 64 tuples across the 64-pair chunks of the EMS expansion (one key per lane) */
#endif
#ifndef PCT_SET_RV
#define PCT_SET_RV 1  /* old slots per lane and matching pass of a table rebuild (32-bit keys) */
#endif

namespace pct {

// ----------------------------------------------------------------------------------------
// packed boxes: six coordinates, BITS bits each, in one key word; the top bits stay clear
// (30 of 32 / 60 of 64 bits used), which the hash-table slot encoding relies on.
// ----------------------------------------------------------------------------------------
template <typename K, int BITS>
struct Pack {
  static constexpr uint32_t M = (1u << BITS) - 1u;
  __device__ static inline K pack(int a, int b, int c, int d, int e, int f) {
    return (K)a | ((K)b << BITS) | ((K)c << (2 * BITS)) | ((K)d << (3 * BITS)) | ((K)e << (4 * BITS)) |
           ((K)f << (5 * BITS));
  }
  __device__ static inline int get(K k, int i) { return (int)((k >> (i * BITS)) & (K)M); }
};

template <typename K, int BITS>
__device__ __forceinline__ uint64_t tuplehash6(K key) {
  uint64_t acc = tuplehash_begin();
#pragma unroll
  for (int i = 0; i < 6; i++) acc = tuplehash_lane(acc, (uint64_t)Pack<K, BITS>::get(key, i));  // hash(int) == int
  return tuplehash_end6(acc);
}

struct EnvRegs {  // wave-uniform per-env scalars
  int n_ems, n_boxes, n_leaf;
  int item0, item1, item2;
  uint64_t cursor;
  uint32_t t;
  int64_t vol;
  uint32_t flags;
  int traj;  // dataset mode: current trajectory (LoadBoxCreator.index)
  uint32_t oc;  // observations produced so far (shuffle key)
  int box_from;  // placed boxes [box_from, n_boxes) are newer than the HBM copy
  int poly_from;  // ... and so are the polygon-pool vertices from this one on (stability settings)
  uint32_t stab_over;  // STAB_WHY_* bits: a stability capacity (pools, workspace, queue) was exceeded -- the step belongs to the retry pass
  // the item that the draw number pre_cursor (of trajectory pre_traj) yields, fetched at kernel start
  int pre0, pre1, pre2, pre_traj;
  uint64_t pre_cursor;
  // strict NumPy-stream mode: position in the MT19937 block, the density drawn for the current observation, and the
  // number of candidates of the current observation (a failed step shuffles that list once more, D/bin3D.py:165)
  int mt_pos, n_cand;
  double den_cur;
};

template <typename K, int BITS>
struct Lds {
  typedef typename std::conditional<sizeof(K) == 4, uint8_t, int16_t>::type HT;  // heights <= 31 fit a byte
  K* tab0;     // candidate hash table(s): table_words_compact(cand_cap) key words
  K* ems_a;    // [ems_cap] current EMS list
  K* ems_b;    // GENEMS scratch list: aliases the table region (idle during GENEMS)
  uint32_t* dd;  // [128] bucket words of the batch de-duplication
  K* box;
  K* leaf;
  HT* hmap;
  uint16_t* vp; /* [64] valid (ems, rotation) pairs of the current chunk */
  uint32_t* cp; /* scheme scratch: CP / EP levels (6 arrays of I+2 words), EV tables (288 words) */
  K* fkey;       /* shuffle: feasible candidates in list order ... */
  uint32_t* fpri; /* ... and their priorities (only when shuffle) */
  uint32_t* mt;   /* [624] the env's MT19937 state (strict NumPy-stream mode only) */
  StabState st;   /* settings 1 / 3: the env's stability state, resident for the whole transition (pct_stab.cuh) ... */
  StabWave sw;    /* ... and the wave's hull workspace + task queue */
};

__host__ __device__ __forceinline__ int discrete_scheme_words(const DiscreteParams& p) {
  if (p.lnes == PCT_LNES_CP || p.lnes == PCT_LNES_EP) return 6 * (p.I + 2);
  if (p.lnes == PCT_LNES_EV) return 288;
  return 0;
}
__host__ __device__ __forceinline__ int discrete_scratch_words(const DiscreteParams& p) {
  return (int)(128 * sizeof(uint32_t) / p.key_bytes);  // dd[128 x u32], in key words
}

// LDS bytes of everything but the stability state (which follows, 16-byte aligned)
__host__ __device__ __forceinline__ size_t discrete_lds_base_bytes(const DiscreteParams& p) {
  size_t k = p.key_bytes;
  size_t n = (size_t)table_words_compact((uint32_t)p.cand_cap) + p.ems_cap + discrete_scratch_words(p) + p.I + p.L;
  size_t hb = ((size_t)p.AA * (k == 4 ? 1 : 2) + 3) & ~(size_t)3;
  size_t cp = (size_t)discrete_scheme_words(p) * sizeof(uint32_t);
  if (p.shuffle && !p.rng_numpy) cp += ((size_t)(p.cand_cap * 3) / 5 + 4) * (sizeof(uint32_t) + k);
  if (p.rng_numpy) cp += 624 * sizeof(uint32_t);
  return (n * k + hb + 64 * sizeof(uint16_t) + cp + 15) & ~(size_t)15;
}

template <typename K, int BITS>
__device__ __forceinline__ Lds<K, BITS> carve_lds(const DiscreteParams& p, unsigned char* base) {
  Lds<K, BITS> l;
  K* q = reinterpret_cast<K*>(base);
  l.tab0 = q; q += table_words_compact((uint32_t)p.cand_cap);
  l.ems_a = q; q += p.ems_cap;
  l.ems_b = l.tab0;
  l.dd = reinterpret_cast<uint32_t*>(q);
  q += discrete_scratch_words(p);
  l.box = q; q += p.I;
  l.leaf = q; q += p.L;
  l.hmap = reinterpret_cast<typename Lds<K, BITS>::HT*>(q);
  size_t hb = ((size_t)p.AA * sizeof(typename Lds<K, BITS>::HT) + 3) & ~(size_t)3;
  l.vp = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(q) + hb);
  l.cp = reinterpret_cast<uint32_t*>(l.vp + 64);
  uint32_t* after_cp = l.cp + discrete_scheme_words(p);
  const int fcap = (p.cand_cap * 3) / 5 + 2;
  l.fpri = after_cp;
  l.fkey = reinterpret_cast<K*>(after_cp + fcap + (fcap & 1));
  // the MT19937 words sit behind the shuffle arrays (counter-keyed shuffle) or directly behind the scheme scratch
  l.mt = (p.shuffle && !p.rng_numpy) ? reinterpret_cast<uint32_t*>(l.fkey + fcap + (fcap & 1)) : after_cp;
  if (p.setting != 2) {
    unsigned char* sbase = base + discrete_lds_base_bytes(p);
    l.st = stab_carve(sbase, p.I, p.sb.caps);
    l.sw = stab_wave_carve(sbase + ((stab_state_bytes(p.I, p.sb.caps) + 15) & ~(size_t)15), p.sb.caps);
  }
  return l;
}

// placed-box geometry for the stability code: lx,ly,lz,xe,ye,ze from the packed LDS boxes
template <typename K, int BITS>
struct BoxGeo {
  const K* box;
  static constexpr bool kSquareIsPow = BITS <= 5;  // (pct_stab.cuh, the lever rule's `tri_base_len ** 2`)
  __device__ __forceinline__ void operator()(int i, double g[9]) const {
    K k = box[i];
#pragma unroll
    for (int c = 0; c < 6; c++) g[c] = (double)Pack<K, BITS>::get(k, c);
    g[6] = g[3] - g[0]; g[7] = g[4] - g[1]; g[8] = g[5] - g[2];  // exact for integers
  }
};
__device__ __forceinline__ void draw_item(const DiscreteParams& p, int e, EnvRegs& r) {
  // binCreator.py:37-39 generate_box_size, through the scripted / counter-based sources
  uint64_t c = r.cursor++;
  const int32_t* it;
  if (p.source == PCT_ITEMS_DATASET) {  // binCreator.py:64-72 generate_box_size
    int t = r.traj < p.ds_ntraj ? r.traj : p.ds_ntraj - 1;
    int len = p.ds_len[t];
    if (c < (uint64_t)len) {
      it = p.stream + ((size_t)t * p.ds_maxlen + (size_t)c) * 3;
    } else {
      int v = (c == (uint64_t)len) ? 100 : 10;
      r.item0 = v; r.item1 = v; r.item2 = v;
      return;
    }
  } else if (p.source == PCT_ITEMS_STREAM) {
    it = p.stream + ((size_t)e * (size_t)p.T + (size_t)(c % (uint64_t)p.T)) * 3;
  } else {
    uint64_t g = (uint64_t)(p.env_id_base + e);
    it = p.item_set + (size_t)(pct_pick(p.seed, g, c, (uint32_t)p.n_items)) * 3;
  }
  r.item0 = it[0];
  r.item1 = it[1];
  r.item2 = it[2];
}

// binCreator.py:37-39 in strict NumPy-stream mode: idx = np.random.randint(0, len(box_set))
template <typename L>
__device__ __forceinline__ void draw_item_mt(const DiscreteParams& p, L& l, EnvRegs& r, int lane) {
  r.cursor++;
  const uint32_t idx = mt_interval(l, r, lane, (uint32_t)p.n_items - 1u);
  const int32_t* it = p.item_set + (size_t)idx * 3;
  r.item0 = it[0];
  r.item1 = it[1];
  r.item2 = it[2];
}

// D/space.py:290-314 Space.reset on the LDS-resident state
template <typename K, int BITS>
__device__ __forceinline__ void space_reset(const DiscreteParams& p, Lds<K, BITS>& l, EnvRegs& r, int lane) {
  for (int c = lane; c < p.AA; c += 64) l.hmap[c] = 0;
  if (lane == 0) l.ems_a[0] = Pack<K, BITS>::pack(0, 0, 0, p.W, p.Ly, p.H);
  r.n_ems = 1;
  r.n_boxes = 0;
  r.box_from = 0;
  r.vol = 0;
  l.st.n_ent = 0;
  l.st.n_poly = 0;
  r.poly_from = 0;
  if (p.source == PCT_ITEMS_DATASET) {  // LoadBoxCreator.reset (binCreator.py:51-62)
    r.traj++;
    r.cursor = 0;
    if (r.traj >= p.ds_ntraj) r.flags |= PCT_FLAG_DATASET_EXHAUSTED;
  }
}

// D/space.py:457-483 GENEMS + :518-531 EliminateInscribedEMS.  ems_a -> ems_a.
template <typename K, int BITS>
__device__ __forceinline__ void genems(const DiscreteParams& p, Lds<K, BITS>& l, EnvRegs& r, int lane, int bx0, int by0,
                              int bz0, int bx1, int by1, int bz1) {
  typedef Pack<K, BITS> P;
  const int E = r.n_ems;
  const int scap = (int)table_words_compact((uint32_t)p.cand_cap);  // pre-elimination list capacity
  const int lb = p.low_bound <= 0 ? 1 : p.low_bound;
  // children of the intersected EMS held by this lane, written at `first` + (this lane's offset
  // among the chunk's children): parent index, then branch order.  Returns the chunk's child count.
  auto emit_children = [&](bool live, K k, int first) __attribute__((always_inline)) -> int {
    int x1 = P::get(k, 0), y1 = P::get(k, 1), z1 = P::get(k, 2), x2 = P::get(k, 3), y2 = P::get(k, 4),
        z2 = P::get(k, 5);
    int x3 = max(bx0, x1), y3 = max(by0, y1), z3 = max(bz0, z1);
    int x4 = min(bx1, x2), y4 = min(by1, y2), z4 = min(bz1, z2);
    bool inter = live && (x3 < x4) && (y3 < y4) && (z3 < z4);
    bool ylz = (y2 - y1 >= lb) && (z2 - z1 >= lb);
    bool xlz = (x2 - x1 >= lb) && (z2 - z1 >= lb);
    bool c0 = inter && (x3 - x1 >= lb) && ylz;                       // [x1,y1,z1,x3,y2,z2]
    bool c1 = inter && (x2 - x4 >= lb) && ylz;                       // [x4,y1,z1,x2,y2,z2]
    bool c2 = inter && (y3 - y1 >= lb) && xlz;                       // [x1,y1,z1,x2,y3,z2]
    bool c3 = inter && (y2 - y4 >= lb) && xlz;                       // [x1,y4,z1,x2,y2,z2]
    bool c4 = inter && (z2 - z4 >= lb) && (x2 - x1 >= lb) && (y2 - y1 >= lb);  // [x1,y1,z4,x2,y2,z2]
    uint64_t m0 = __ballot(c0), m1 = __ballot(c1), m2 = __ballot(c2), m3 = __ballot(c3), m4 = __ballot(c4);
    int pos = first + rank_below(m0) + rank_below(m1) + rank_below(m2) + rank_below(m3) +
              rank_below(m4);
    if (c0) { if (pos < scap) l.ems_b[pos] = P::pack(x1, y1, z1, x3, y2, z2); pos++; }
    if (c1) { if (pos < scap) l.ems_b[pos] = P::pack(x4, y1, z1, x2, y2, z2); pos++; }
    if (c2) { if (pos < scap) l.ems_b[pos] = P::pack(x1, y1, z1, x2, y3, z2); pos++; }
    if (c3) { if (pos < scap) l.ems_b[pos] = P::pack(x1, y4, z1, x2, y2, z2); pos++; }
    if (c4) { if (pos < scap) l.ems_b[pos] = P::pack(x1, y1, z4, x2, y2, z2); pos++; }
    return __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3) + __popcll(m4);
  };
  auto intersects = [&](K k) __attribute__((always_inline)) -> bool {
    int t1 = max(bx0, P::get(k, 0)), u1 = max(by0, P::get(k, 1)), v1 = max(bz0, P::get(k, 2));
    int t2 = min(bx1, P::get(k, 3)), u2 = min(by1, P::get(k, 4)), v2 = min(bz1, P::get(k, 5));
    return (t1 < t2) && (u1 < u2) && (v1 < v2);
  };
  // survivors (EMS not intersected by the box) keep their order at the front, then the children of
  // every intersected EMS
  int S = 0, C = 0;
  bool overflow = false;
  if (E <= 64) {  // the usual case: one read of the list serves both
    bool live = lane < E;
    K k = live ? l.ems_a[lane] : (K)0;
    bool surv = live && !intersects(k);
    uint64_t m = __ballot(surv);
    if (surv) l.ems_b[rank_below(m)] = k;
    S = __popcll(m);
    C = emit_children(live, k, S);
  } else {
    for (int base = 0; base < E; base += 64) {
      int i = base + lane;
      bool live = i < E;
      K k = live ? l.ems_a[i] : (K)0;
      bool surv = live && !intersects(k);
      uint64_t m = __ballot(surv);
      if (surv) l.ems_b[S + rank_below(m)] = k;
      S += __popcll(m);
    }
    for (int base = 0; base < E; base += 64) {
      int i = base + lane;
      bool live = i < E;
      K k = live ? l.ems_a[i] : (K)0;
      C += emit_children(live, k, S + C);
    }
  }
  int n = S + C;
  if (n > scap) {
    overflow = true;
    n = scap;
  }
  if (overflow) r.flags |= PCT_FLAG_EMS_OVERFLOW;
  __syncthreads();
  // elimination: i is deleted iff some j != i contains it (non-strict, pre-deletion list).
  // The list before GENEMS is containment-free (it is the output of the previous
  // elimination, or the single initial EMS), and a child is a subset of its intersected
  // parent, so a survivor can neither lie inside a child nor equal one: only children can be
  // deleted.  Survivors are copied, children are tested against the whole list.
  for (int i = lane; i < (S < n ? S : n); i += 64) l.ems_a[i] = l.ems_b[i];
  int out = S < n ? S : n;
  // 5-bit coordinates: containment as two packed subtractions.  A triple of 5-bit fields is spread to
  // 6-bit spacing; with the guard bit G set in the minuend, (x | G) - y keeps G in a field iff
  // x >= y there (no borrow crosses a field).  i inside j  <=>  lo(i) >= lo(j) and hi(j) >= hi(i)
  // field by field; an entry always contains itself, so "some j != i" is "at least two j".
  const bool swar = (BITS == 5) && (3 * n <= scap) && (out < n);
  uint32_t* cmpw = reinterpret_cast<uint32_t*>(l.ems_b) + n;  // [n][2]: spread lo, spread hi | G
  constexpr uint32_t G = 0x20820u;
  auto spread = [](uint32_t t) { return (t & 0x1Fu) | ((t & 0x3E0u) << 1) | ((t & 0x7C00u) << 2); };
  if (swar) {
    for (int i = lane; i < n; i += 64) {
      uint32_t kk = (uint32_t)l.ems_b[i];
      cmpw[2 * i] = spread(kk & 0x7FFFu);
      cmpw[2 * i + 1] = spread((kk >> 15) & 0x7FFFu) | G;
    }
    __syncthreads();
  }
  if (swar && n - out <= 32) {
    // few children: several lanes share a child and split the list between them
    const int Cn = n - out;
    int Cp = 1;
    while (Cp < Cn) Cp <<= 1;
    const int parts = 64 / Cp;
    const int ci = lane & (Cp - 1), part = lane / Cp;
    const bool live = ci < Cn;
    const K k = live ? l.ems_b[out + ci] : (K)0;
    const uint32_t alo_g = spread((uint32_t)k & 0x7FFFu) | G, ahi = spread(((uint32_t)k >> 15) & 0x7FFFu);
    int cnt = 0;
    for (int j = part; j < n; j += parts) {
      const uint32_t blo = cmpw[2 * j], bhi_g = cmpw[2 * j + 1];
      const uint32_t t = (alo_g - blo) & (bhi_g - ahi) & G;
      cnt += (t == G) ? 1 : 0;
    }
    for (int off = Cp; off < 64; off <<= 1) cnt += __shfl_xor(cnt, off, 64);
    const bool keep = live && part == 0 && cnt < 2;
    const uint64_t m = __ballot(keep);
    const int o = out + rank_below(m);
    if (keep && o < p.ems_cap) l.ems_a[o] = k;
    out += __popcll(m);
  } else
  for (int base = out; base < n; base += 64) {
    int i = base + lane;
    bool live = i < n;
    K k = live ? l.ems_b[i] : (K)0;
    bool del = false;
    if (swar) {
      const uint32_t alo_g = spread((uint32_t)k & 0x7FFFu) | G, ahi = spread(((uint32_t)k >> 15) & 0x7FFFu);
      int cnt = 0;
#pragma unroll 4
      for (int j = 0; j < n; j++) {
        const uint32_t blo = uniform_key<uint32_t>(cmpw[2 * j]), bhi_g = uniform_key<uint32_t>(cmpw[2 * j + 1]);
        const uint32_t t = (alo_g - blo) & (bhi_g - ahi) & G;
        cnt += (t == G) ? 1 : 0;
      }
      del = cnt >= 2;
    } else {
      int a0 = P::get(k, 0), a1 = P::get(k, 1), a2 = P::get(k, 2), a3 = P::get(k, 3), a4 = P::get(k, 4),
          a5 = P::get(k, 5);
      for (int j = 0; j < n; j++) {
        K kj = uniform_key<K>(l.ems_b[j]);
        int b0 = P::get(kj, 0), b1 = P::get(kj, 1), b2 = P::get(kj, 2), b3 = P::get(kj, 3), b4 = P::get(kj, 4),
            b5 = P::get(kj, 5);
        bool inside = (a0 >= b0) & (a1 >= b1) & (a2 >= b2) & (a3 <= b3) & (a4 <= b4) & (a5 <= b5);
        del |= inside & (j != i);
      }
    }
    bool keep = live && !del;
    uint64_t m = __ballot(keep);
    int o = out + rank_below(m);
    if (keep && o < p.ems_cap) l.ems_a[o] = k;
    out += __popcll(m);
  }
  if (out > p.ems_cap) {  // the list that survives elimination must fit the state array
    out = p.ems_cap;
    r.flags |= PCT_FLAG_EMS_OVERFLOW;
  }
  r.n_ems = out;
  __syncthreads();
}

// The candidate set under construction: a CPython `set` whose table lives in LDS (l.tab0).
template <typename K>
struct SetState {
  K* tabs;         // the table region
  uint32_t* dd;    // [128] bucket words of the batch de-duplication, all ones between uses
  uint32_t cap;    // candidate_capacity (largest table)
  uint32_t toff, size, fill;
  bool overflow;
};

// set.add of up to V*64 tuples, V per lane, in batch-position order (position of (v, lane) = v*64 + lane):
// membership test against the table, exact in-batch de-duplication, then insertion exactly as CPython
// would do it one key at a time -- growth when fill*5 >= mask*3 right after an insertion, re-insertion
// in old-slot order (Objects/setobject.c set_add_entry / set_table_resize / set_insert_clean).
// A table rebuild re-inserts RV*64 old slots per matching pass.
template <typename K, int BITS, int V, typename TM>
__device__ __forceinline__ void set_insert(SetState<K>& st, const K (&key)[V], const bool (&valid)[V], int lane, TM& tm,
                                  int* mst) {
  const K EMPTY = SlotWord<K>::EMPTY;
  K* const tabs = st.tabs;
  uint32_t* const dd = st.dd;
  bool pending[V];
  uint64_t hash0 = 0;  // slice 0's hashes, for the fast start
  uint64_t hash[V];    // (V > 1: recomputed per matching pass below rather than kept live across a table rebuild)
  {
#pragma unroll
    for (int v = 0; v < V; v++) hash[v] = tuplehash6<K, BITS>(key[v]);
    hash0 = hash[0];
    // no separate membership pass: a key that is already in the table ends its matching walk at its own entry
    // (pyset_match_v CHECK) -- one probe walk per key instead of two
#pragma unroll
    for (int v = 0; v < V; v++) pending[v] = valid[v];
    tm.sub_tick(PH_SET_GEN);
    if (!__ballot(any_of<V>(pending))) return;
    tm.add(ST_FLUSHES, 1);
    // set.add of a key that an earlier position of the batch holds is a no-op
    bool dup[V];
    batch_find_duplicates_v<V, 128, K>(dd, pending, key, hash, lane, dup);
#pragma unroll
    for (int v = 0; v < V; v++) pending[v] = pending[v] && !dup[v];
  }
  tm.sub_tick(PH_SET_DEDUP);
  if (sizeof(K) == 4 && st.fill == 0 && st.size == 8 && st.cap >= 128) {
    const uint64_t pm0 = __ballot(pending[0]);
    if (__popcll(pm0) >= 19) {
      // Fast start of a fresh set whose first 19 new keys sit in slice 0 (every key of this batch is new to
      // the empty table and to the rest of the batch).  CPython puts the first 5 into the 8-slot table, grows
      // it to 32 slots (re-inserting in slot order), adds keys 6..19 and grows again to 128.  The two small
      // tables are replayed on the scalar unit -- a table is a VGPR whose lane s holds the source lane of the
      // key in slot s, occupancy is a scalar bit mask, hashes come over v_readlane -- and only their outcome
      // is materialised: the 128-slot table receives, in ONE pass, the 19 keys in 32-table slot order
      // followed by the rest of slice 0 (at most 64 keys, below the 128-slot table's growth point of 77).
      uint64_t rem = pm0;
      int t8 = 0xFF, t32 = 0xFF;  // per lane: source lane of the key in that slot
      uint32_t occ8 = 0, occ32 = 0;
      const uint64_t h0 = hash0;
      auto lane_hash = [&](int src) __attribute__((always_inline)) -> uint64_t {
        uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h0, src);
        uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h0 >> 32), src);
        return ((uint64_t)hi << 32) | lo;
      };
      for (int o = 0; o < 5; o++) {  // mask 7: no linear probes (i + 9 > mask)
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
      auto insert32 = [&](int src) __attribute__((always_inline)) {  // set_insert_clean / set_add_entry on the 32-slot table
        const uint64_t h = lane_hash(src);
        uint32_t i = (uint32_t)h & 31u;
        uint64_t perturb = h;
        while (true) {
          const uint32_t span = (i + 9u <= 31u) ? 10u : 1u;  // slot i, plus 9 linear probes if they fit
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
      // insertion order into the 128-slot table: 32-table slot order, then the rest of slice 0
      K* fin = reinterpret_cast<K*>(dd);
      const K from_slot = shfl_key<K>(key[0], t32 & 63);
      const bool in32 = lane < 32 && ((occ32 >> lane) & 1u);
      const bool later = (rem >> lane) & 1ull;
      if (in32) fin[rank_below((uint64_t)occ32)] = from_slot;
      if (later) fin[19 + rank_below(rem)] = key[0];
      const int total = 19 + __popcll(rem);
      tabs[lane] = EMPTY;
      tabs[64 + lane] = EMPTY;
      __syncthreads();
      const K mk = lane < total ? fin[lane] : (K)0;
      __syncthreads();
      dd[lane] = 0xFFFFFFFFu;  // back to the all-ones state the de-duplication expects
      if (sizeof(K) == 8) dd[64 + lane] = 0xFFFFFFFFu;
      if (TM::on) tm.sub_tick(PH_FAST_START);
      const bool mpart[1] = {lane < total};
      const uint64_t mhash[1] = {tuplehash6<K, BITS>(mk)};
      uint32_t mslot[1];
      pyset_match_v<1, K>(tabs, 127u, mpart, mhash, lane, mslot, mst);
      if (mpart[0]) tabs[mslot[0]] = mk;
      st.toff = 0;  // table_offset_compact(cap, 128) for every cap >= 128
      st.size = 128;
      st.fill = (uint32_t)total;
      pending[0] = false;
      __syncthreads();
      tm.sub_tick(PH_SET_MATCH);
    }
  }
  while (true) {
    uint64_t pm[V];
    int total = 0;
#pragma unroll
    for (int v = 0; v < V; v++) {
      pm[v] = __ballot(pending[v]);
      total += __popcll(pm[v]);
    }
    // set_add_entry grows the table when fill*5 >= mask*3, checked right after each insertion: at most
    // thr - fill more keys go into this table -- the first thr - fill pending positions
    const uint32_t mask = st.size - 1;
    const uint32_t thr = (mask * 3u + 4u) / 5u;
    if (st.fill < thr) {
      if (!total) break;
      // the first thr - fill pending positions: even if every one of them is new the table does not outgrow its
      // threshold inside this pass (those that turn out to be members already leave room for a further pass)
      const int budget = (int)(thr - st.fill);
      bool part[V], placed[V];
      int before = 0, nplaced = 0;
#pragma unroll
      for (int v = 0; v < V; v++) {
        part[v] = pending[v] && before + rank_below(pm[v]) < budget;
        before += __popcll(pm[v]);
      }
      uint32_t slot[V];
      if (V > 1) {
#pragma unroll
        for (int v = 0; v < V; v++) hash[v] = tuplehash6<K, BITS>(key[v]);
      }
      pyset_match_v<V, K, true>(tabs + st.toff, mask, part, hash, lane, slot, mst, key, placed);
#pragma unroll
      for (int v = 0; v < V; v++) {
        if (placed[v]) tabs[st.toff + slot[v]] = key[v];
        pending[v] = pending[v] && !part[v];
        nplaced += __popcll(__ballot(placed[v]));
      }
      st.fill += (uint32_t)nplaced;
      __syncthreads();
      tm.sub_tick(PH_SET_MATCH);
    }
    if (st.fill >= thr) {  // set_table_resize(used * 4): re-insert in old-slot order
      uint32_t newsize = 8;
      while (newsize <= st.fill * 4u) newsize <<= 1;
      if (newsize > st.cap) {
        st.overflow = true;
        break;
      }
      const uint32_t noff = table_offset_compact(st.cap, newsize);
      tm.add(ST_REBUILDS, 1);
      constexpr int RV = sizeof(K) == 4 ? PCT_SET_RV : 4;  // old slots re-inserted per lane and pass
      if (noff == st.toff) {
        // same region: lift the old table (<= 512 slots = 8 per lane) into registers, wipe, re-insert
        K oldk[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const uint32_t s2 = (uint32_t)c * 64u + lane;
          oldk[c] = (s2 < st.size) ? tabs[st.toff + s2] : EMPTY;
        }
        __syncthreads();
        uint32_t nchunks = (st.size + 63u) / 64u;
        if (st.size == 512u) {
          // 512 -> 2048: the 307 keys sit in 8 sparse chunks of 64 slots; squeezed (slot order kept) they are 5 dense
          // chunks = 5 matching passes instead of 8.  The dense list goes through the old table's own words.
          uint32_t base = 0;
#pragma unroll
          for (int c = 0; c < 8; c++) {
            const uint64_t m = __ballot(oldk[c] != EMPTY);
            if (oldk[c] != EMPTY) tabs[noff + base + (uint32_t)rank_below(m)] = oldk[c];
            base += (uint32_t)__popcll(m);
          }
          __syncthreads();
#pragma unroll
          for (int c = 0; c < 8; c++) {
            const uint32_t q = (uint32_t)c * 64u + lane;
            oldk[c] = q < base ? tabs[noff + q] : EMPTY;
          }
          nchunks = (base + 63u) / 64u;
          __syncthreads();
        }
        for (uint32_t s2 = lane; s2 < newsize; s2 += 64) tabs[noff + s2] = EMPTY;
        __syncthreads();
#pragma unroll
        for (int c0 = 0; c0 < 8; c0 += RV) {
          if ((uint32_t)c0 < nchunks) {
            bool opart[RV];
            uint64_t ohash[RV];
            uint32_t oslot[RV];
#pragma unroll
            for (int c = 0; c < RV; c++) {
              opart[c] = oldk[c0 + c] != EMPTY;
              ohash[c] = tuplehash6<K, BITS>(oldk[c0 + c]);
            }
            pyset_match_v<RV, K>(tabs + noff, newsize - 1, opart, ohash, lane, oslot, mst);
#pragma unroll
            for (int c = 0; c < RV; c++)
              if (opart[c]) tabs[noff + oslot[c]] = oldk[c0 + c];
            __syncthreads();
          }
        }
      } else {
        for (uint32_t s2 = lane; s2 < newsize; s2 += 64) tabs[noff + s2] = EMPTY;
        __syncthreads();
        for (uint32_t sb = 0; sb < st.size; sb += 64u * RV) {
          K ok[RV];
          bool opart[RV];
          uint64_t ohash[RV];
          uint32_t oslot[RV];
#pragma unroll
          for (int c = 0; c < RV; c++) {
            const uint32_t s2 = sb + (uint32_t)c * 64u + lane;
            ok[c] = (s2 < st.size) ? tabs[st.toff + s2] : EMPTY;
            opart[c] = ok[c] != EMPTY;
            ohash[c] = tuplehash6<K, BITS>(ok[c]);
          }
          pyset_match_v<RV, K>(tabs + noff, newsize - 1, opart, ohash, lane, oslot, mst);
#pragma unroll
          for (int c = 0; c < RV; c++)
            if (opart[c]) tabs[noff + oslot[c]] = ok[c];
          __syncthreads();
        }
      }
      st.toff = noff;
      st.size = newsize;
      tm.sub_tick(PH_SET_REBUILD);
    }
  }
}

// rotation `rot` of the item (b0, b1, b2) (D/space.py:540-562): extents and the skip rule
__device__ __forceinline__ bool item_rot_size(int b0, int b1, int b2, int rot, int& sx, int& sy, int& sz) {
  switch (rot) {
    case 0: sx = b0; sy = b1; sz = b2; return false;
    case 1: sx = b1; sy = b0; sz = b2; return sx == sy;
    case 2: sx = b0; sy = b2; sz = b1; return sx == sy && sy == sz;
    case 3: sx = b1; sy = b2; sz = b0; return sx == sy && sy == sz;
    case 4: sx = b2; sy = b0; sz = b1; return sx == sy;
    default: sx = b2; sy = b1; sz = b0; return sx == sy;
  }
}

#ifndef PCT_SET_WHOLE
#define PCT_SET_WHOLE 0 /* 1: the EMS candidate set is built from the whole tuple list at once (below); 0: batch by batch.
                           Bit-exact (every discrete GPU test passes with it) but SLOWER on MI355X, measured round 3
                           (profiles/r03_whole_set_experiment.txt): C2 74.3 vs 68.0 us per launch.  It cuts the matching
                           calls of the EMS-richest env from 22.8 to 5.9 and its summed longest walks from 135 to 78 steps,
                           but with several keys per lane every walk step issues that many times the instructions, and with
                           four waves per SIMD the kernel is bound by instruction issue, not by the dependence chain. */
#endif
#ifndef PCT_SET_WHOLE_MIN
#define PCT_SET_WHOLE_MIN 0 /* (EMS, rotation) pairs from which on an env takes the whole-list path (experiments: only the EMS-rich
                               envs that set a launch's length, the others batch by batch) */
#endif
// ----------------------------------------------------------------------------------------------------------------
// The EMS candidate set built FROM THE WHOLE TUPLE LIST AT ONCE (32-bit keys, a table region of >= 2048 words).
//
// Sequential set.add of the generated tuples is: drop the tuples an earlier one equals (no-ops), then insert the DISTINCT
// ones d0, d1, ... in order, growing the table 8 -> 32 -> 128 -> 512 -> 2048 right after the 5th, 19th, 77th and 307th
// distinct key with a re-insertion in old-slot order (Objects/setobject.c set_add_entry / set_table_resize).  The final
// table is therefore a function of the distinct sequence alone, and it can be had stage by stage:
//     32-table  <- the first 5 in 8-table slot order, then d5..d18               (scalar replay, as the fast start)
//     128-table <- the first 19 in 32-table slot order, then d19..d76            ONE matching of <= 77 keys (2 per lane)
//     512-table <- the first 77 in 128-table slot order, then d77..d306          ONE matching of <= 307 keys (5 per lane)
//     2048-table<- the first 307 in 512-table slot order, then d307..            passes of 256 keys (4 per lane)
// where a "matching" is the wave-parallel stable matching of pct_set.cuh with the insertion position as the priority --
// any number of distinct keys at once.  Instead of one matching (a chain of dependent LDS round trips) per 64-tuple
// batch -- 23 of them in the EMS-rich env that sets a launch's duration -- the env pays one per table stage, each with
// several keys per lane in flight.  The de-duplication runs once over the whole list (bucket scatter by position with
// a cheap multiplicative hash; the first holder of a key is the minimum position of its bucket in some round).
// Layout inside the table region (words): [0, 512) tables up to 512 slots, [512, 1024) de-duplication buckets, later the
// 32-table slot order, [1024, 1024 + WS_KMAX) the tuples, compacted in place to the distinct ones.  The 2048-slot table
// takes the whole region: what it is built from waits in an HBM scratch row of the workgroup.
// Returns false (nothing of the caller's set state touched beyond scratch) when the list does not fit: the caller then
// inserts batch by batch.
constexpr int WS_KMAX = 768;  // generated tuples the whole-set path takes (the 99.9th percentile at 10^3 is ~600)
template <typename K, int BITS, typename TM>
__device__ __forceinline__ bool ems_set_whole(const DiscreteParams& p, Lds<K, BITS>& l, SetState<K>& st, int lane, TM& tm, int* mst,
                                     int E, int orient, uint32_t rotmask, int b0, int b1, int b2) {
  typedef Pack<K, BITS> P;
  const K EMPTY = SlotWord<K>::EMPTY;
  K* const tabs = st.tabs;
  constexpr uint32_t WB = 512, WD = 1024;
  const int NP = E * orient;
  // 1. every tuple, in generation order: pair (EMS, rotation) by pair, four bottom corners each (D/space.py:565-568)
  int NT = 0;
  for (int pbase = 0; pbase < NP; pbase += 64) {
    const int q = pbase + lane;
    bool pv = q < NP;
    const int ei = q / orient, rot = q - ei * orient;
    int sx, sy, sz;
    const bool skip = item_rot_size(b0, b1, b2, rot, sx, sy, sz) || !((rotmask >> rot) & 1u);
    const K ek = pv ? l.ems_a[ei] : (K)0;
    const int x0 = P::get(ek, 0), y0 = P::get(ek, 1), z0 = P::get(ek, 2), x1 = P::get(ek, 3), y1 = P::get(ek, 4);
    pv = pv && !skip && (x1 - x0 >= sx) && (y1 - y0 >= sy) && (P::get(ek, 5) - z0 >= sz);
    const uint64_t pm = __ballot(pv);
    const int at = NT + 4 * rank_below(pm);
    if (pv && at + 3 < WS_KMAX) {
      tabs[WD + at + 0] = P::pack(x0, y0, z0, x0 + sx, y0 + sy, z0 + sz);
      tabs[WD + at + 1] = P::pack(x1 - sx, y0, z0, x1, y0 + sy, z0 + sz);
      tabs[WD + at + 2] = P::pack(x0, y1 - sy, z0, x0 + sx, y1, z0 + sz);
      tabs[WD + at + 3] = P::pack(x1 - sx, y1 - sy, z0, x1, y1, z0 + sz);
    }
    NT += 4 * __popcll(pm);
  }
  if (NT > WS_KMAX) return false;
  tm.add(ST_GENERATED, (uint64_t)NT);
  if (NT == 0) return true;  // the empty set: the caller's fresh 8-slot table
  const int NC = (NT + 63) >> 6;
  // 2. exact de-duplication of the whole list, first occurrence wins
  for (int i = lane; i < 512; i += 64) tabs[WB + i] = EMPTY;
  uint32_t unres = 0, dupm = 0;  // per lane: bit c = tuple c * 64 + lane
  for (int c = 0; c < NC; c++)
    if (c * 64 + lane < NT) unres |= 1u << c;
  __syncthreads();
  for (int round = 0; round < 10; round++) {
    if (!__ballot(unres != 0)) break;
    const uint32_t mul = 0x9E3779B1u + (uint32_t)round * 0x3C6EF372u;  // (odd)
    const uint32_t part = unres;
    for (int c = 0; c < NC; c++)
      if ((part >> c) & 1u) {
        const uint32_t t = (uint32_t)(c * 64 + lane);
        const uint32_t b = ((uint32_t)tabs[WD + t] * mul) >> 23;
        atomicMin(reinterpret_cast<uint32_t*>(&tabs[WB + b]), t);
      }
    __syncthreads();
    for (int c = 0; c < NC; c++)
      if ((part >> c) & 1u) {
        const uint32_t t = (uint32_t)(c * 64 + lane);
        const K key = tabs[WD + t];
        const uint32_t w = (uint32_t)tabs[WB + (((uint32_t)key * mul) >> 23)];
        if (w == t) unres &= ~(1u << c);  // the minimum position of its bucket: nobody earlier holds this key
        else if (tabs[WD + w] == key) { unres &= ~(1u << c); dupm |= 1u << c; }
      }
    __syncthreads();
    for (int c = 0; c < NC; c++)
      if ((part >> c) & 1u) tabs[WB + (((uint32_t)tabs[WD + c * 64 + lane] * mul) >> 23)] = EMPTY;
    __syncthreads();
  }
  if (__ballot(unres != 0)) return false;  // (ten rounds of pure collisions between distinct keys)
  tm.sub_tick(PH_SET_DEDUP);
  // 3. the distinct tuples d0, d1, ... compacted to the front of the list, order kept
  int n = 0;
  for (int c = 0; c < NC; c++) {
    const int t = c * 64 + lane;
    const K key = t < NT ? tabs[WD + t] : (K)0;
    const bool first = t < NT && !((dupm >> c) & 1u);
    const uint64_t m = __ballot(first);
    __syncthreads();
    if (first) tabs[WD + n + rank_below(m)] = key;
    n += __popcll(m);
    __syncthreads();
  }
  if (n < 19) {  // up to the 32-slot table: one ordinary batch
    const K key1[1] = {lane < n ? tabs[WD + lane] : (K)0};
    const bool valid1[1] = {lane < n};
    __syncthreads();
    set_insert<K, BITS, 1>(st, key1, valid1, lane, tm, mst);
    return true;
  }
  // 4a. the 8- and 32-slot tables of d0..d18, replayed on the scalar unit (as the fast start of set_insert): lane s of t32
  // holds the position of the key in 32-table slot s
  const K key0 = tabs[WD + lane];  // (n >= 19; positions >= n are never used)
  const uint64_t h0 = tuplehash6<K, BITS>(key0);
  int t8 = 0xFF, t32 = 0xFF;
  uint32_t occ8 = 0, occ32 = 0;
  auto lane_hash = [&](int src) __attribute__((always_inline)) -> uint64_t {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h0, src);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h0 >> 32), src);
    return ((uint64_t)hi << 32) | lo;
  };
  for (int o = 0; o < 5; o++) {  // mask 7: no linear probes (i + 9 > mask)
    const uint64_t h = lane_hash(o);
    uint32_t i = (uint32_t)h & 7u;
    uint64_t perturb = h;
    while ((occ8 >> i) & 1u) {
      perturb >>= 5;
      i = (i * 5u + 1u + (uint32_t)perturb) & 7u;
    }
    occ8 |= 1u << i;
    t8 = lane == (int)i ? o : t8;
  }
  auto insert32 = [&](int src) __attribute__((always_inline)) {  // set_insert_clean / set_add_entry on the 32-slot table
    const uint64_t h = lane_hash(src);
    uint32_t i = (uint32_t)h & 31u;
    uint64_t perturb = h;
    while (true) {
      const uint32_t span = (i + 9u <= 31u) ? 10u : 1u;  // slot i, plus 9 linear probes if they fit
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
  for (int o = 5; o < 19; o++) insert32(o);
  {
    const K from_slot = shfl_key<K>(key0, t32 & 63);
    if (lane < 32 && ((occ32 >> lane) & 1u)) tabs[WB + rank_below((uint64_t)occ32)] = from_slot;  // 32-table slot order
  }
  tabs[lane] = EMPTY;
  tabs[64 + lane] = EMPTY;
  __syncthreads();
  // 4b. the 128-slot table: those 19, then d19..d76
  const int m1 = n < 77 ? n : 77;
  {
    K k2[2];
    bool part2[2];
    uint64_t hash2[2];
    uint32_t slot2[2];
#pragma unroll
    for (int v = 0; v < 2; v++) {
      const int pos = v * 64 + lane;
      part2[v] = pos < m1;
      k2[v] = pos < 19 ? tabs[WB + pos] : (part2[v] ? tabs[WD + pos] : (K)0);
      hash2[v] = tuplehash6<K, BITS>(k2[v]);
    }
    pyset_match_v<2, K>(tabs, 127u, part2, hash2, lane, slot2, mst);
#pragma unroll
    for (int v = 0; v < 2; v++)
      if (part2[v]) tabs[slot2[v]] = k2[v];
    __syncthreads();
  }
  if (n < 77) {
    st.toff = 0; st.size = 128; st.fill = (uint32_t)n;
    tm.sub_tick(PH_SET_MATCH);
    return true;
  }
  // 4c. the 512-slot table: the 77 in 128-table slot order, then d77..d306 -- the insertion-order list is completed in
  // place (the 77 go in front of d77.. in the list region), then matched in passes of 192 positions (3 keys per lane)
  {
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const K k = tabs[c * 64 + lane];
      const bool occ = k != EMPTY;
      const uint64_t m = __ballot(occ);
      if (occ) tabs[WD + cnt + rank_below(m)] = k;  // (d0..d76 themselves are no longer needed in generation order)
      cnt += __popcll(m);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; c++) tabs[c * 64 + lane] = EMPTY;
    __syncthreads();
    const int m2 = n < 307 ? n : 307;
    for (int base = 0; base < m2; base += 192) {
      K k3[3];
      bool part3[3];
      uint64_t hash3[3];
      uint32_t slot3[3];
#pragma unroll
      for (int v = 0; v < 3; v++) {
        const int pos = base + v * 64 + lane;
        part3[v] = pos < m2;
        k3[v] = part3[v] ? tabs[WD + pos] : (K)0;
        hash3[v] = tuplehash6<K, BITS>(k3[v]);
      }
      pyset_match_v<3, K>(tabs, 511u, part3, hash3, lane, slot3, mst, nullptr, nullptr, (uint32_t)base);
#pragma unroll
      for (int v = 0; v < 3; v++)
        if (part3[v]) tabs[slot3[v]] = k3[v];
      __syncthreads();
    }
  }
  if (n < 307) {
    st.toff = 0; st.size = 512; st.fill = (uint32_t)n;
    tm.sub_tick(PH_SET_MATCH);
    return true;
  }
  // 4d. the 2048-slot table takes the whole region: the insertion-order list -- the 307 in 512-table slot order, then
  // d307.. -- waits in this env's HBM scratch row while the region is wiped (one store / load round trip, only in the
  // 0.5 % of the steps that reach this stage); passes of 192 positions
  {
    uint32_t* const gs = p.set_scratch + (size_t)blockIdx.x * WS_KMAX;
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const K k = tabs[c * 64 + lane];
      const bool occ = k != EMPTY;
      const uint64_t m = __ballot(occ);
      if (occ) __hip_atomic_store(&gs[cnt + rank_below(m)], (uint32_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cnt += __popcll(m);
    }
    for (int pos = 307 + lane; pos < n; pos += 64)
      __hip_atomic_store(&gs[pos], (uint32_t)tabs[WD + pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int i = lane; i < 2048; i += 64) tabs[i] = EMPTY;
    __threadfence_block();
    __syncthreads();
    for (int base = 0; base < n; base += 192) {
      K k3[3];
      bool part3[3];
      uint64_t hash3[3];
      uint32_t slot3[3];
#pragma unroll
      for (int v = 0; v < 3; v++) {
        const int pos = base + v * 64 + lane;
        part3[v] = pos < n;
        k3[v] = part3[v] ? (K)__hip_atomic_load(&gs[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (K)0;  // (past the L1)
        hash3[v] = tuplehash6<K, BITS>(k3[v]);
      }
      pyset_match_v<3, K>(tabs, 2047u, part3, hash3, lane, slot3, mst, nullptr, nullptr, (uint32_t)base);
#pragma unroll
      for (int v = 0; v < 3; v++)
        if (part3[v]) tabs[slot3[v]] = k3[v];
      __syncthreads();
    }
  }
  st.toff = 0; st.size = 2048; st.fill = (uint32_t)n;
  tm.sub_tick(PH_SET_MATCH);
  return true;
}

// D/space.py:534-570 EMSPoint (CPython-set order) + D/bin3D.py:100-136
// get_possible_position: fills l.leaf[0..n_leaf) with the first <= L feasible candidates.
// RNG: bit 0 = the candidate list is shuffled before the first-L cut, bit 1 = strict NumPy-stream mode (the shuffle, the
// item picks and the densities consume the env's MT19937 stream; otherwise they are counter-keyed)
template <typename K, int BITS, bool STAB, int SCHEME, int RNG, typename TM>
__device__ __forceinline__ void leaf_nodes(const DiscreteParams& p, int e, Lds<K, BITS>& l, EnvRegs& r, int lane, TM& tm) {
  typedef Pack<K, BITS> P;
  constexpr bool SHUFFLE = (RNG & 1) != 0, MT = (RNG & 2) != 0;
  // strict mode: the observation's density is drawn first (D/bin3D.py:80-84), the shuffle draws follow below
  if (MT && p.setting == 3) r.den_cur = mt_density(l, r, lane);
  const int E = r.n_ems;
  const int b0 = r.item0, b1 = r.item1, b2 = r.item2;
  constexpr int orient = STAB ? 2 : 6;  // setting 2 <=> no stability check <=> 6 orientations (D/space.py:536-537)
  const int NP = E * orient;  // (EMS, rotation) pairs in set-insertion order

  // fresh set: PySet_MINSIZE = 8 slots
  const K EMPTY = SlotWord<K>::EMPTY;
  K* const tabs = l.tab0;
  SetState<K> st;
  st.tabs = l.tab0;
  st.dd = l.dd;
  st.cap = (uint32_t)p.cand_cap;
  st.size = 8;
  st.fill = 0;
  st.toff = table_offset_compact((uint32_t)p.cand_cap, 8u);
  st.overflow = false;
  if (lane < 8) tabs[st.toff + lane] = EMPTY;
  l.dd[lane] = 0xFFFFFFFFu;
  l.dd[64 + lane] = 0xFFFFFFFFu;
  __syncthreads();
  int mstat[4] = {0, 0, 0, 0};  // timed build only: match calls, outer rounds, sum over calls of the longest walk, cycles inside the walk loops
  int* const mst = TM::on ? mstat : nullptr;
  tm.sub_start();

  // rotation r of the item (D/space.py:540-562): extents and the skip rule
  auto rot_size = [&](int rot, int& sx, int& sy, int& sz) __attribute__((always_inline)) -> bool {
    switch (rot) {
      case 0: sx = b0; sy = b1; sz = b2; return false;
      case 1: sx = b1; sy = b0; sz = b2; return sx == sy;
      case 2: sx = b0; sy = b2; sz = b1; return sx == sy && sy == sz;
      case 3: sx = b1; sy = b2; sz = b0; return sx == sy && sy == sz;
      case 4: sx = b2; sy = b0; sz = b1; return sx == sy;
      default: sx = b2; sy = b1; sz = b0; return sx == sy;
    }
  };

  // SCHEME 1 = every expansion other than EMS, selected at run time (coverage paths)
  const bool FC = SCHEME == 1 && p.lnes == PCT_LNES_FC;
  const bool EV = SCHEME == 1 && p.lnes == PCT_LNES_EV;
  const bool EP = SCHEME == 1 && p.lnes == PCT_LNES_EP;
  const bool CP = SCHEME == 1 && (p.lnes == PCT_LNES_CP || EP);  // CP and EP share the level skeleton
  int n_ev = 0;  // EV: number of ordered candidates left in l.cp
  if (FC) {
    // D/space.py:573-610 FullCoord: rotation-major, then lx, then ly; lz = the cell's own height
    const int NQ = orient * p.W * p.Ly;
    for (int base = 0; base < NQ && !st.overflow; base += 64) {
      int q = base + lane;
      bool valid = q < NQ;
      int rot = q / (p.W * p.Ly);
      int rem = q - rot * (p.W * p.Ly);
      int px = rem / p.Ly, py = rem - px * p.Ly;
      int sx, sy, sz;
      bool skip = rot_size(rot, sx, sy, sz);
      int pz = valid ? (int)l.hmap[px * p.A + py] : 0;
      valid = valid && !skip && (px + sx <= p.W) && (py + sy <= p.Ly) && (pz + sz <= p.H);
      const K key1[1] = {P::pack(px, py, pz, px + sx, py + sy, pz + sz)};
      const bool valid1[1] = {valid};
      set_insert<K, BITS, 1>(st, key1, valid1, lane, tm, mst);
      __syncthreads();
    }
  } else if (EV) {
    // D/space.py:613-693 EventPoint.  bin3D.py:171 runs GENEMS only under LNES == 'EMS', so under
    // 'EV' ZMAP and the EMS list stay as Space.reset left them: level 0 with x_up=[0], y_left=[0],
    // x_bottom=[W], y_right=[Ly] and the whole-bin EMS.  posVec = the four bin corners of every
    // rotation (<= 24 tuples, possibly with negative coordinates, which is why they do not go
    // through the packed-key table); one lane replays set.add on a 128-slot table of candidate ids.
    uint32_t* tb = l.cp;  // [128] ids (rot * 4 + corner), 0xFF = empty
    auto ev_tuple = [&](int id, int t[6]) __attribute__((always_inline)) {
      int sx, sy, sz;
      rot_size(id >> 2, sx, sy, sz);
      int xs = (id & 2) ? p.W - sx : 0, ys = (id & 1) ? p.Ly - sy : 0;  // add order :654-674
      t[0] = xs; t[1] = ys; t[2] = 0; t[3] = xs + sx; t[4] = ys + sy; t[5] = sz;
    };
    auto ev_hash = [&](const int t[6]) __attribute__((always_inline)) {
      uint64_t acc = tuplehash_begin();
      for (int c = 0; c < 6; c++) acc = tuplehash_lane(acc, pyhash_int(t[c]));
      return tuplehash_end6(acc);
    };
    if (lane == 0) {
      uint32_t sz_t = 8, fill = 0;
      for (int i = 0; i < 8; i++) tb[i] = 0xFFu;
      // set_insert_clean / set_add_entry probe sequence (LINEAR_PROBES 9, PERTURB_SHIFT 5)
      auto probe_insert = [&](uint32_t* tab, uint32_t mask, int id, bool check) __attribute__((always_inline)) -> bool {
        int t[6];
        ev_tuple(id, t);
        Walk w;
        w.start(ev_hash(t), mask);
        while (true) {
          uint32_t cur = tab[w.i + w.j];
          if (cur == 0xFFu) { tab[w.i + w.j] = (uint32_t)id; return true; }
          if (check) {
            int u[6];
            ev_tuple((int)cur, u);
            bool eq = true;
            for (int c = 0; c < 6; c++) eq = eq && (u[c] == t[c]);
            if (eq) return false;
          }
          w.next(mask);
        }
      };
      for (int id = 0; id < orient * 4; id++) {
        int sx, sy, sz;
        if (rot_size(id >> 2, sx, sy, sz)) continue;
        if (probe_insert(tb, sz_t - 1, id, true)) {
          fill++;
          uint32_t mask = sz_t - 1;
          if (fill * 5u >= mask * 3u) {  // set_table_resize(used * 4) into the other half of the scratch
            uint32_t ns = 8;
            while (ns <= fill * 4u) ns <<= 1;
            uint32_t* nt = (tb == l.cp) ? l.cp + 128 : l.cp;
            for (uint32_t i = 0; i < ns; i++) nt[i] = 0xFFu;
            for (uint32_t i = 0; i < sz_t; i++)
              if (tb[i] != 0xFFu) probe_insert(nt, ns - 1, (int)tb[i], false);
            tb = nt;
            sz_t = ns;
          }
        }
      }
      // list(posVec) in slot order, kept if the footprint lies inside the EMS (:677-688)
      uint32_t* ord = l.cp + 256;
      int m = 0;
      for (uint32_t i = 0; i < sz_t; i++) {
        if (tb[i] == 0xFFu) continue;
        int t[6];
        ev_tuple((int)tb[i], t);
        if (t[0] >= 0 && t[1] >= 0 && t[3] <= p.W && t[4] <= p.Ly) ord[m++] = tb[i];
      }
      ord[31] = (uint32_t)m;
    }
    __syncthreads();
    n_ev = (int)l.cp[256 + 31];
    // the ordered survivors become the "table": slot i of a fresh region holds candidate i
    {
      uint32_t id = lane < n_ev ? l.cp[256 + lane] : 0u;
      int t[6];
      ev_tuple((int)id, t);
      __syncthreads();
      st.size = 64;
      st.toff = 0;
      tabs[lane] = lane < n_ev ? P::pack(t[0], t[1], t[2], t[3], t[4], t[5]) : EMPTY;
      __syncthreads();
    }
  } else if (CP && r.n_boxes == 0) {
    // D/space.py:756-757 (and :700-701 for EP): an empty bin yields a plain two-element LIST
    // (unrotated, x/y swapped; no set, no in-bin test): slots 0 and 1 of the fresh 8-slot table hold
    // them in list order, duplicates included
    if (lane == 0) {
      tabs[st.toff + 0] = P::pack(0, 0, 0, b0, b1, b2);
      tabs[st.toff + 1] = P::pack(0, 0, 0, b1, b0, b2);
    }
    __syncthreads();
  } else if (CP) {
    // D/space.py:758-774 + D/PctTools.py:137-158 (CP) / :696-716 + PctTools.py:114-136 (EP): per level
    // k (sorted distinct tops, 0 first) the corner / extreme points of the boxes reaching above k,
    // minus those of the previous level
    const int n = r.n_boxes;
    const int cw = p.I + 2;
    uint32_t* T = l.cp;              // [I+2] levels
    uint32_t* srt = l.cp + cw;       // [I+2] box ids in the level's sort order
    uint32_t* cik = l.cp + 2 * cw;   // [2(I+2)] points of this level: x | y << 16
    uint32_t* last = l.cp + 4 * cw;  // [2(I+2)] points of the previous level
    uint32_t* CI = reinterpret_cast<uint32_t*>(l.ems_a);  // EMS are not kept under CP/EP: x | y<<10 | k<<20
    const int ci_cap = (int)(p.ems_cap * sizeof(K) / sizeof(uint32_t));
    // distinct tops, ascending (Tset, D/space.py:758-761)
    int nT = 1;
    if (lane == 0) T[0] = 0;
    for (int base = 0; base < n; base += 64) {  // cik[i] = 1 iff box i is the first with its top
      int i = base + lane;
      int top = i < n ? P::get(l.box[i], 5) : 0;
      bool first = i < n;
      for (int j = 0; j < n; j++) first = first && !(j < i && P::get(uniform_key<K>(l.box[j]), 5) == top);
      if (i < n) cik[i] = first ? 1u : 0u;
    }
    __syncthreads();
    for (int base = 0; base < n; base += 64) {
      int i = base + lane;
      int top = i < n ? P::get(l.box[i], 5) : 0;
      bool first = i < n && cik[i] != 0u;
      int less = 0;
      for (int j = 0; j < n; j++)
        less += (uniform_key<uint32_t>(cik[j]) != 0u && P::get(uniform_key<K>(l.box[j]), 5) < top) ? 1 : 0;
      if (first) T[1 + less] = (uint32_t)top;
      nT += __popcll(__ballot(first));
    }
    __syncthreads();
    int nCI = 0, nlast = 0;
    bool ci_overflow = false;
    for (int ti = 0; ti < nT; ti++) {
      const int k = (int)uniform_key<uint32_t>(T[ti]);
      // stable sort of the active rectangles, rank by counting: CP descending by (ye, xe)
      // (PctTools.py:143), EP ascending by (ly, lxe) (PctTools.py:117)
      int nact = 0;
      for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        K bi = i < n ? l.box[i] : (K)0;
        bool act = i < n && P::get(bi, 5) > k;
        int k1 = EP ? P::get(bi, 1) : P::get(bi, 4), k2 = P::get(bi, 3);
        int rank = 0;
        for (int j = 0; j < n; j++) {
          K bj = uniform_key<K>(l.box[j]);
          bool actj = P::get(bj, 5) > k;
          int j1 = EP ? P::get(bj, 1) : P::get(bj, 4), j2 = P::get(bj, 3);
          bool before = EP ? ((j1 < k1) || (j1 == k1 && j2 < k2) || (j1 == k1 && j2 == k2 && j < i))
                           : ((j1 > k1) || (j1 == k1 && j2 > k2) || (j1 == k1 && j2 == k2 && j < i));
          rank += (actj && before) ? 1 : 0;
        }
        if (act) srt[rank] = (uint32_t)i;
        nact += __popcll(__ballot(act));
      }
      __syncthreads();
      int nc = 0;
      if (nact == 0) {
        if (lane == 0) cik[0] = 0;  // corners2D([]) == [(0, 0)]; extreme2D([]) == [(0, 0, 0)]
        nc = 1;
      } else if (EP) {
        // extreme2D (PctTools.py:114-136): item i of the sorted list projects onto the walls / the
        // items before it -> up to two points, in the order of a 2-element set; a point is dropped
        // if a LATER item's footprint covers it (deleteEps2D runs before the item adds its own)
        for (int base = 0; base < nact; base += 64) {
          int pos = base + lane;
          bool live = pos < nact;
          K bb = live ? l.box[srt[pos]] : (K)0;
          const int lx = P::get(bb, 0), ly = P::get(bb, 1), lxe = P::get(bb, 3), lye = P::get(bb, 4);
          // demo walls smallBox(-1,0,0,10) and smallBox(0,-1,10,0): the literal 10 of PctTools.py:118
          bool have0 = lx >= 0 && lye < 10, have2 = ly >= 0 && lxe < 10;
          int max0 = 0, max2 = 0;             // projectedX / projectedY maxima (both walls project 0)
          int idx0 = have0 ? -2 : 0x7fffffff;  // position (in demo + earlier items) of the first valid box
          int idx2 = have2 ? -1 : 0x7fffffff;
          for (int q = 0; q < nact; q++) {
            K bq = uniform_key<K>(l.box[uniform_key<uint32_t>(srt[q])]);
            int qxe = P::get(bq, 3), qye = P::get(bq, 4);
            bool before = q < pos;
            if (before && lx >= qxe && lye < qye) {  // IsProjectionValid2D(newItem, box, 0)
              if (!have0 || qxe > max0) max0 = qxe;
              if (!have0) { have0 = true; idx0 = q; }
            }
            if (before && ly >= qye && lxe < qxe) {  // direction 2
              if (!have2 || qye > max2) max2 = qye;
              if (!have2) { have2 = true; idx2 = q; }
            }
          }
          // maxBound starts at -10 and every projection is >= 0, so the first valid box always sets it
          uint32_t e0 = (uint32_t)max0 | ((uint32_t)lye << 16);  // (projectedX, newItem.ly + newItem.y)
          uint32_t e2 = (uint32_t)lxe | ((uint32_t)max2 << 16);  // (newItem.lx + newItem.x, projectedY)
          uint32_t ea = 0, eb = 0;
          int ne = 0;
          if (live) {
            if (have0 && have2 && e0 != e2) {
              uint32_t fa = idx0 < idx2 ? e0 : e2, fb = idx0 < idx2 ? e2 : e0;  // dict insertion order
              uint64_t ha = tuplehash_end(tuplehash_lane(tuplehash_lane(tuplehash_begin(), fa & 0xFFFFu), fa >> 16), 2ULL);
              uint64_t hb = tuplehash_end(tuplehash_lane(tuplehash_lane(tuplehash_begin(), fb & 0xFFFFu), fb >> 16), 2ULL);
              uint32_t ia = (uint32_t)ha & 7u, ib = (uint32_t)hb & 7u;
              uint64_t perturb = hb;
              while (ib == ia) {  // 8-slot table: no linear probes
                perturb >>= 5;
                ib = (ib * 5u + 1u + (uint32_t)perturb) & 7u;
              }
              ea = ia < ib ? fa : fb;
              eb = ia < ib ? fb : fa;
              ne = 2;
            } else if (have0 || have2) {
              ea = have0 ? e0 : e2;
              ne = 1;
            }
          }
          bool keep_a = ne >= 1, keep_b = ne >= 2;
          for (int q = 0; q < nact; q++) {
            K bq = uniform_key<K>(l.box[uniform_key<uint32_t>(srt[q])]);
            int qx = P::get(bq, 0), qy = P::get(bq, 1), qxe = P::get(bq, 3), qye = P::get(bq, 4);
            bool later = q > pos;
            int ax = (int)(ea & 0xFFFFu), ay = (int)(ea >> 16), bx = (int)(eb & 0xFFFFu), by = (int)(eb >> 16);
            if (later && ax >= qx && ax < qxe && ay >= qy && ay < qye) keep_a = false;
            if (later && bx >= qx && bx < qxe && by >= qy && by < qye) keep_b = false;
          }
          uint64_t ma = __ballot(keep_a), mb = __ballot(keep_b);
          int o = nc + rank_below(ma) + rank_below(mb);
          if (keep_a) cik[o] = ea;
          if (keep_b) cik[o + (keep_a ? 1 : 0)] = eb;
          nc += __popcll(ma) + __popcll(mb);
        }
      } else {
        // extreme items (PctTools.py:145-151): xe above the running maximum of everything sorted
        // before; the corner an extreme item contributes is (that running maximum, its ye) --
        // the running maximum is 0 for the first one and the previous extreme item's xe after
        int m = 0, xmax = 0;
        for (int base = 0; base < nact; base += 64) {
          int pos = base + lane;
          bool live = pos < nact;
          K bb = live ? l.box[srt[pos]] : (K)0;
          int xe = P::get(bb, 3), ye = P::get(bb, 4);
          int run = 0;
          for (int q = 0; q < nact; q++) {
            int xq = P::get(uniform_key<K>(l.box[srt[q]]), 3);
            run = (q < pos && xq > run) ? xq : run;
          }
          bool ext = live && xe > run;
          uint64_t em_mask = __ballot(ext);
          if (ext) cik[m + rank_below(em_mask)] = (uint32_t)run | ((uint32_t)ye << 16);
          m += __popcll(em_mask);
          int passmax = wave_max_i32(live ? xe : 0);
          xmax = passmax > xmax ? passmax : xmax;
        }
        // closing corner (xe of the last extreme item, 0): the running maximum of all xe
        if (lane == 0) cik[m] = (uint32_t)xmax;
        nc = m + 1;
      }
      __syncthreads();
      // CI += points not present at the previous level (order kept).  EP: the empty level's point is
      // the 3-tuple (0,0,0), which no 2-tuple of the previous level equals (and vice versa)
      const bool ep_empty = EP && nact == 0;
      for (int base = 0; base < nc; base += 64) {
        int c = base + lane;
        bool live = c < nc;
        uint32_t v = live ? cik[c] : 0u;
        bool seen = false;
        for (int q = 0; q < nlast; q++) seen = seen || (last[q] == v);
        bool add = live && (!seen || ep_empty);
        uint64_t m2 = __ballot(add);
        int o = nCI + rank_below(m2);
        if (add) {
          if (o < ci_cap) CI[o] = (v & 0xFFFFu) | ((v >> 16) << 10) | ((uint32_t)k << 20);
          else ci_overflow = true;
        }
        nCI += __popcll(m2);
      }
      ci_overflow = __ballot(ci_overflow) != 0;
      if (nCI > ci_cap) nCI = ci_cap;
      __syncthreads();
      for (int c = lane; c < nc; c += 64) last[c] = cik[c];
      nlast = ep_empty ? 0 : nc;
      __syncthreads();
    }
    if (ci_overflow) r.flags |= PCT_FLAG_EMS_OVERFLOW;
    // candidates: corner x rotation, in-bin test (D/space.py:776-803), into the set
    const int NQ = nCI * orient;
    for (int base = 0; base < NQ && !st.overflow; base += 64) {
      int q = base + lane;
      bool valid = q < NQ;
      int ci = q / orient, rot = q - ci * orient;
      int sx, sy, sz;
      bool skip = rot_size(rot, sx, sy, sz);
      uint32_t cv = valid ? CI[ci] : 0u;
      int px = (int)(cv & 0x3FFu), py = (int)((cv >> 10) & 0x3FFu), pz = (int)(cv >> 20);
      valid = valid && !skip && (px + sx <= p.W) && (py + sy <= p.Ly) && (pz + sz <= p.H);
      const K key1[1] = {P::pack(px, py, pz, px + sx, py + sy, pz + sz)};
      const bool valid1[1] = {valid};
      set_insert<K, BITS, 1>(st, key1, valid1, lane, tm, mst);
      __syncthreads();
    }
  } else {
    // rotations worth generating: not skipped by the reference's rule, and not a repeat of an
    // earlier generated rotation with the same (sx, sy, sz) -- for one EMS that repeat yields the
    // same four tuples right after the first, i.e. set.add no-ops
    // (closed form of "first occurrence of (sx,sy,sz) among the rotations the reference generates")
    const bool e01 = b0 == b1, e02 = b0 == b2, e12 = b1 == b2;
    const bool g1 = !e01, g2 = !e12, g3 = !(e01 && e12) && !(g1 && e02) && !(g2 && e01);
    const bool g4 = !e02 && !(g1 && e12), g5 = !e12 && !e02 && !(g4 && e01);
    const uint32_t rotmask = 1u | (g1 ? 2u : 0u) | (g2 ? 4u : 0u) | (g3 ? 8u : 0u) | (g4 ? 16u : 0u) | (g5 ? 32u : 0u);
    constexpr int V = PCT_SET_V;  // a chunk of 64 (EMS, rotation) pairs = up to 256 tuples = 4 / V batches, V tuples per lane
    bool whole = false;
    if (PCT_SET_WHOLE && sizeof(K) == 4 && p.cand_cap >= 2048 && NP >= PCT_SET_WHOLE_MIN)
      whole = ems_set_whole<K, BITS>(p, l, st, lane, tm, mst, E, orient, rotmask, b0, b1, b2);
    if (!whole) {  // (the list did not fit the scratch, or a small table region: batch by batch)
      if (lane < 8) tabs[st.toff + lane] = EMPTY;
      l.dd[lane] = 0xFFFFFFFFu;
      l.dd[64 + lane] = 0xFFFFFFFFu;
      __syncthreads();
    }
#if PCT_SET_CARRY
    // Batches of exactly 64 tuples across chunk boundaries (round 4): a chunk of 64 (EMS, rotation) pairs yields 4 x (pairs that
    // fit) tuples, and its last batch used to go out part full -- 2.8 batches per step for 117 tuples, 10.4 for the 545 of the
    // EMS-richest env whose candidate set sets a launch's length.  The tuples a chunk leaves over (< 64) wait in a register,
    // one per lane, in front of the next chunk's; batch position = generation order, as before.
    K carry = (K)0;
    int nc = 0;  // lanes [0, nc) hold a carried tuple
    auto tuple_of = [&](int tt) __attribute__((always_inline)) -> K {  // tuple number tt of the current chunk (l.vp)
      const int qq = (int)l.vp[(tt >> 2) & 63];
      const int corner = tt & 3;
      const int e2 = qq / orient;
      int tx, ty, tz;
      rot_size(qq - e2 * orient, tx, ty, tz);
      const K k2 = l.ems_a[e2];
      const int x0 = P::get(k2, 0), y0 = P::get(k2, 1), z0 = P::get(k2, 2), x1 = P::get(k2, 3), y1 = P::get(k2, 4);
      const int xs = (corner & 1) ? x1 - tx : x0;
      const int ys = (corner & 2) ? y1 - ty : y0;
      return P::pack(xs, ys, z0, xs + tx, ys + ty, z0 + tz);
    };
    for (int pbase = 0; pbase < NP && !st.overflow && !whole && V == 1; pbase += 64) {
      const uint64_t tpair = tm.now();
      int q = pbase + lane;
      bool pv = q < NP;
      int ei = q / orient, rot = q - ei * orient;
      int sx, sy, sz;
      bool skip = rot_size(rot, sx, sy, sz) || !((rotmask >> rot) & 1u);
      K ek = pv ? l.ems_a[ei] : (K)0;
      pv = pv && !skip && (P::get(ek, 3) - P::get(ek, 0) >= sx) && (P::get(ek, 4) - P::get(ek, 1) >= sy) &&
           (P::get(ek, 5) - P::get(ek, 2) >= sz);
      uint64_t pm = __ballot(pv);
      const int nt = 4 * __popcll(pm);  // four bottom-corner placements per pair (:565-568)
      if (!nt) continue;
      if (pv) l.vp[rank_below(pm)] = (uint16_t)q;
      tm.add(ST_GENERATED, (uint64_t)nt);
      __syncthreads();
      if (TM::on) tm.add(PH_GEN_PAIRS, tm.now() - tpair);
      int off = 0;
      while (nc + (nt - off) >= 64 && !st.overflow) {
        const K key1[1] = {lane < nc ? carry : tuple_of(off + lane - nc)};
        const bool valid1[1] = {true};
        set_insert<K, BITS, 1>(st, key1, valid1, lane, tm, mst);
        off += 64 - nc;
        nc = 0;
      }
      const int rem = nt - off;  // (< 64 - nc) joins the carry
      if (lane >= nc && lane < nc + rem) carry = tuple_of(off + lane - nc);
      nc += rem;
      __syncthreads();
    }
    if (V == 1 && nc > 0 && !st.overflow && !whole) {
      const K key1[1] = {carry};
      const bool valid1[1] = {lane < nc};
      set_insert<K, BITS, 1>(st, key1, valid1, lane, tm, mst);
      __syncthreads();
    }
    for (int pbase = 0; pbase < NP && !st.overflow && !whole && V != 1; pbase += 64) {
#else
    for (int pbase = 0; pbase < NP && !st.overflow && !whole; pbase += 64) {
#endif
      // which (EMS, rotation) pairs of this chunk can hold the item at all
      const uint64_t tpair = tm.now();
      int q = pbase + lane;
      bool pv = q < NP;
      int ei = q / orient, rot = q - ei * orient;
      int sx, sy, sz;
      bool skip = rot_size(rot, sx, sy, sz) || !((rotmask >> rot) & 1u);
      K ek = pv ? l.ems_a[ei] : (K)0;
      pv = pv && !skip && (P::get(ek, 3) - P::get(ek, 0) >= sx) && (P::get(ek, 4) - P::get(ek, 1) >= sy) &&
           (P::get(ek, 5) - P::get(ek, 2) >= sz);
      uint64_t pm = __ballot(pv);
      const int nt = 4 * __popcll(pm);  // four bottom-corner placements per pair (:565-568)
      if (!nt) continue;
      if (pv) l.vp[rank_below(pm)] = (uint16_t)q;
      tm.add(ST_GENERATED, (uint64_t)nt);
      __syncthreads();
      if (TM::on) tm.add(PH_GEN_PAIRS, tm.now() - tpair);
      for (int tb = 0; tb < nt && !st.overflow; tb += 64 * V) {
      K key[V];
      bool valid[V];
#pragma unroll
      for (int v = 0; v < V; v++) {
        const int tt = tb + v * 64 + lane;  // batch position = generation order
        valid[v] = tt < nt;
        const int qq = valid[v] ? (int)l.vp[tt >> 2] : 0;
        const int corner = tt & 3;
        const int e2 = qq / orient;
        int tx, ty, tz;
        rot_size(qq - e2 * orient, tx, ty, tz);
        const K k2 = l.ems_a[e2];
        const int x0 = P::get(k2, 0), y0 = P::get(k2, 1), z0 = P::get(k2, 2), x1 = P::get(k2, 3), y1 = P::get(k2, 4);
        const int xs = (corner & 1) ? x1 - tx : x0;
        const int ys = (corner & 2) ? y1 - ty : y0;
        key[v] = P::pack(xs, ys, z0, xs + tx, ys + ty, z0 + tz);
      }
      set_insert<K, BITS, V>(st, key, valid, lane, tm, mst);
      }
      __syncthreads();
    }
  }
  if (st.overflow) r.flags |= PCT_FLAG_CANDIDATE_OVERFLOW;
  uint32_t size = st.size;
  const uint32_t toff = st.toff, fill = st.fill;
  __syncthreads();
  tm.add(ST_EMS, (uint64_t)E);
  tm.add(ST_DISTINCT, (uint64_t)fill);
  if (TM::on) { tm.add(ST_MATCH_CALLS, (uint64_t)mstat[0]); tm.add(ST_MATCH_ROUNDS, (uint64_t)mstat[1]); tm.add(ST_MATCH_PROBES, (uint64_t)mstat[2]); tm.add(ST_CONTAINS_CALLS, (uint64_t)mstat[3]); }
  tm.sub_tick(PH_SET_GEN);
  tm.tick(PH_SET);

  // iterate the table in slot order (= list(set)), test feasibility, keep the first L
  int nleaf = 0;
  uint32_t stab_err = 0;
  // D/bin3D.py:75-84: the density drawn for this observation (setting 3), else 1
  const double next_den = !STAB ? 1.0 : (MT ? (p.setting == 3 ? r.den_cur : 1.0) : next_density(p, e, r.oc, r.traj, r.cursor - 1));
  // feasibility of the candidate every lane holds (EMPTY: none).  All 64 lanes call: the stability check of the lanes
  // that need one is a wave-cooperative task walk (pct_stab.cuh stab_virtual_wave).
  bool stab_ill = false;
  StabStats sstats = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (timed build only)
  bool unknown = false;  // the last call left this lane's candidate undecided (a capacity of its own was exceeded)
  auto feasible = [&](K k) __attribute__((always_inline)) -> bool {
    unknown = false;
    const bool occ = k != SlotWord<K>::EMPTY;
    int xs = P::get(k, 0), ys = P::get(k, 1), xe = P::get(k, 3), ye = P::get(k, 4);
    int z = P::get(k, 5) - P::get(k, 2);
    int mh = 0;  // D/space.py:400-401 footprint maximum (candidate's own zs is ignored)
    if (occ)
      for (int x = xs; x < xe; x++)
        for (int y = ys; y < ye; y++) {
          int h = l.hmap[x * p.A + y];
          mh = h > mh ? h : mh;
        }
    // check_box :436-446: EMS candidates are inside the bin by construction
    bool feas = occ && (xe <= p.W) && (ye <= p.Ly) && (mh + z <= p.H);
    if (STAB) {  // :447-454 calculated_impact_virtual(first=True)
      const bool need = feas && mh != 0;
      if (__ballot(need)) {
        const double cand[9] = {(double)xs, (double)ys, (double)mh, (double)xe, (double)ye, (double)(mh + z),
                                (double)(xe - xs), (double)(ye - ys), (double)z};
        BoxGeo<K, BITS> geo{l.box};
        uint32_t cap = 0;
        bool ill = false, lerr = false;
        const bool stable = stab_virtual_wave<false>(geo, l.st, r.n_boxes, need, cand, next_den, l.sw, lane, cap, lerr, ill,
                                                     TM::on ? &sstats : nullptr);
        if (need) feas = stable && !cap;
        stab_err |= cap;
        unknown = need && lerr;
        if (__ballot(ill)) stab_ill = true;
      }
    }
    return feas;
  };
  if (SHUFFLE && MT) {
    // bin3D.py:114-115 np.random.shuffle(allPostion), draw for draw: the keys are squeezed to the front of the table
    // region in slot order (= the array the reference shuffles), then Fisher-Yates from the back with
    // j = random_interval(i) (legacy RandomState.shuffle), then the plain first-L sweep over the permuted list
    uint32_t cnt = 0;
    for (uint32_t sb = 0; sb < size; sb += 64) {
      K k = tabs[toff + sb + lane];
      bool occ = (sb + lane < size) && k != SlotWord<K>::EMPTY;
      uint64_t m = __ballot(occ);
      __syncthreads();
      if (occ) tabs[toff + cnt + rank_below(m)] = k;
      cnt += (uint32_t)__popcll(m);
      __syncthreads();
    }
    r.n_cand = (int)cnt;
    for (int i = (int)cnt - 1; i >= 1; i--) {
      const int j = (int)mt_interval(l, r, lane, (uint32_t)i);
      if (j != i && lane == 0) {
        const K a = tabs[toff + i], b = tabs[toff + j];
        tabs[toff + i] = b;
        tabs[toff + j] = a;
      }
    }
    __syncthreads();
    const uint32_t padded = (cnt + 63u) & ~63u;
    if (cnt + lane < padded) tabs[toff + cnt + lane] = SlotWord<K>::EMPTY;
    size = padded;
    __syncthreads();
    for (uint32_t sb = 0; sb < size && nleaf < p.L; sb += 64) {
      K k = tabs[toff + sb + lane];
      bool feas = feasible(k);
      uint64_t m = __ballot(feas);
      int idx = nleaf + rank_below(m);
      if (feas && idx < p.L) l.leaf[idx] = k;
      if (STAB && __ballot(unknown && idx < p.L)) stab_err |= STAB_WHY_SPLIT;  // (beyond the L-th feasible one the reference never looks)
      nleaf += __popcll(m);
    }
  } else if (SHUFFLE) {
    // bin3D.py:114-115 np.random.shuffle(allPostion) -> pct_shuffle_priority: every candidate is
    // tested, the feasible ones are ranked by (priority, list index), the first L ranks are kept
    if (STAB && size > 64) {  // (keys instead of slots, as in the plain sweep below: one call of the stability check per 64 candidates)
      uint32_t cnt = 0;
      for (uint32_t sb = 0; sb < size; sb += 64) {
        K k = tabs[toff + sb + lane];
        bool occ = k != SlotWord<K>::EMPTY;
        uint64_t m = __ballot(occ);
        if (occ) tabs[toff + cnt + rank_below(m)] = k;
        cnt += (uint32_t)__popcll(m);
      }
      const uint32_t padded = (cnt + 63u) & ~63u;
      if (cnt + lane < padded) tabs[toff + cnt + lane] = SlotWord<K>::EMPTY;
      size = padded;
      __syncthreads();
    }
    int nlist = 0, nf = 0;
    for (uint32_t sb = 0; sb < size; sb += 64) {
      uint32_t s2 = sb + lane;
      K k = (s2 < size) ? tabs[toff + s2] : SlotWord<K>::EMPTY;
      bool occ = k != SlotWord<K>::EMPTY;
      uint64_t om = __ballot(occ);
      uint32_t li = (uint32_t)(nlist + rank_below(om));
      bool feas = feasible(k);
      if (STAB && __ballot(unknown)) stab_err |= STAB_WHY_SPLIT;
      uint64_t fm = __ballot(feas);
      int o = nf + rank_below(fm);
      if (feas) {
        l.fkey[o] = k;
        l.fpri[o] = pct_shuffle_priority(p.shuffle_seed, (uint64_t)(p.env_id_base + e), (uint64_t)r.oc, li);
      }
      nlist += __popcll(om);
      nf += __popcll(fm);
    }
    __syncthreads();
    for (int base = 0; base < nf; base += 64) {
      int a2 = base + lane;
      bool live = a2 < nf;
      uint32_t pa = live ? l.fpri[a2] : 0u;
      int rank = 0;
      for (int j = 0; j < nf; j++) {
        uint32_t pj = uniform_key<uint32_t>(l.fpri[j]);
        rank += (pj < pa || (pj == pa && j < a2)) ? 1 : 0;
      }
      if (live && rank < p.L) l.leaf[rank] = l.fkey[a2];
    }
    nleaf = nf;
  } else {
    if (size >= 512 || (STAB && size > 64)) {
      // a big table is at most 60 % full (usually far less): squeeze the keys to the front, in slot
      // order, so that the feasibility sweep below walks keys instead of slots (the table is not
      // needed as a table any more; a chunk is read whole before it is written, leftwards).
      // Stability settings (round 5): from 128 slots on -- the ~40 candidates of a 128-slot table then make ONE call of the
      // wave-cooperative check instead of two half-empty ones, and every call is a chain of level-0 rounds, walk passes and
      // solve rounds (the slowest env of a c1 launch: 1.98 calls, 3.6 level-0 rounds, 4.7 solve rounds per step before)
      uint32_t cnt = 0;
      for (uint32_t sb = 0; sb < size; sb += 64) {
        K k = tabs[toff + sb + lane];
        bool occ = k != SlotWord<K>::EMPTY;
        uint64_t m = __ballot(occ);
        if (occ) tabs[toff + cnt + rank_below(m)] = k;
        cnt += (uint32_t)__popcll(m);
      }
      const uint32_t padded = (cnt + 63u) & ~63u;
      if (cnt + lane < padded) tabs[toff + cnt + lane] = SlotWord<K>::EMPTY;
      size = padded;
      __syncthreads();
    }
    for (uint32_t sb = 0; sb < size && nleaf < p.L; sb += 64) {
      uint32_t s2 = sb + lane;
      K k = (s2 < size) ? tabs[toff + s2] : SlotWord<K>::EMPTY;
      bool feas = feasible(k);
      uint64_t m = __ballot(feas);
      int idx = nleaf + rank_below(m);
      if (feas && idx < p.L) l.leaf[idx] = k;
      if (STAB && __ballot(unknown && idx < p.L)) stab_err |= STAB_WHY_SPLIT;  // (beyond the L-th feasible one the reference never looks)
      nleaf += __popcll(m);
    }
  }
  r.oc++;
  if (STAB) r.stab_over |= stab_err;  // (wave-uniform)
  if (STAB && stab_ill) r.flags |= PCT_FLAG_ILL_CONDITIONED;
  r.n_leaf = nleaf < p.L ? nleaf : p.L;
  if (TM::on && STAB) {
    // pass / task counters are kept by lane 0, the per-lane ones (level-0 tasks, solves) are summed over the wave
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
}

// D/bin3D.py:70-93: the [I+L+1, 9] float32 observation, written once, coalesced
template <typename K, int BITS>
// `full` rewrites every row (reset, end of an episode, freshly bound buffer); otherwise the buffer
// still holds this env's previous observation and only what changed is written: the row of the
// box just placed (`new_row`, or -1), the L leaf rows and the next-item row.
__device__ __forceinline__ void write_obs(const DiscreteParams& p, int e, const Lds<K, BITS>& l, const EnvRegs& r, int lane,
                                 float* __restrict__ obs, bool full, int new_row) {
  typedef Pack<K, BITS> P;
  int a = r.item0, b = r.item1, c = r.item2, tmp;
  if (a > b) { tmp = a; a = b; b = tmp; }
  if (b > c) { tmp = b; b = c; c = tmp; }
  if (a > b) { tmp = a; a = b; b = tmp; }
  // 7 rows (63 consecutive floats) per pass: a lane keeps its column, only the row advances
  const int col = lane % 9, rsub = lane / 9;
  const bool lane_on = lane < 63;
  const int rows = p.I + p.L + 1;
  const bool dens = p.setting == 3;  // densities other than 1 (D/space.py:386, D/bin3D.py:91)
  const double* bden = l.st.den;  // the LDS-resident stability state (only read under setting 3)
  const float nden = !dens ? 1.0f : (p.rng_numpy ? (float)r.den_cur : (float)next_density(p, e, r.oc - 1, r.traj, r.cursor - 1));
  if (!full && new_row >= 0 && lane < 9) {
    K k = l.box[new_row];
    float v = lane < 6 ? (float)P::get(k, lane) : (lane == 7 ? 0.f : 1.0f);
    if (dens && lane == 6) v = (float)bden[new_row];
    obs_st(&obs[new_row * 9 + lane], v);
  }
  if (!full) {
    // incremental: lane = leaf row, nine strided stores per lane (far fewer instructions than the
    // coalesced row-major sweep below; the launch is latency-bound, not store-bound), then the item row
    for (int jb = 0; jb < p.L; jb += 64) {
      const int j = jb + lane;
      if (j < p.L) {
        const bool on = j < r.n_leaf;
        const K k = on ? l.leaf[j] : (K)0;
        float* o = obs + (size_t)(p.I + j) * 9;
        obs_st(o + 0, (float)P::get(k, 0)); obs_st(o + 1, (float)P::get(k, 1)); obs_st(o + 2, (float)P::get(k, 2));
        obs_st(o + 3, (float)P::get(k, 3)); obs_st(o + 4, (float)P::get(k, 4));
        obs_st(o + 5, on ? (float)p.H : 0.f);
        obs_st(o + 6, 0.f); obs_st(o + 7, 0.f);
        obs_st(o + 8, on ? 1.0f : 0.f);
      }
    }
    if (lane < 9) {
      const int col2 = lane;
      obs_st(&obs[(size_t)(p.I + p.L) * 9 + col2],
             col2 == 0 ? nden : (col2 == 3 ? (float)a : (col2 == 4 ? (float)b : (col2 == 5 ? (float)c : (col2 == 8 ? 1.0f : 0.f)))));
    }
    return;
  }
  for (int rbase = 0; rbase < rows; rbase += 7) {
    const int row = rbase + rsub;
    if (!lane_on || row >= rows) continue;
    float v = 0.f;
    if (row < p.I) {
      if (row < r.n_boxes) {
        K k = l.box[row];
        v = col < 6 ? (float)P::get(k, col) : (col == 7 ? 0.f : 1.0f);  // density 1, pad 0, mask 1
        if (dens && col == 6) v = (float)bden[row];
      } else if (row == 0 && col == 8) {
        v = 1.0f;  // D/space.py:294-295 dummy valid node after reset
      }
    } else if (row < p.I + p.L) {
      int j = row - p.I;
      if (j < r.n_leaf) {
        K k = l.leaf[j];
        v = col < 5 ? (float)P::get(k, col) : (col == 5 ? (float)p.H : (col == 8 ? 1.0f : 0.f));
      }
    } else {
      v = col == 0 ? nden : (col == 3 ? (float)a : (col == 4 ? (float)b : (col == 5 ? (float)c : (col == 8 ? 1.0f : 0.f))));
    }
    obs_st(&obs[row * 9 + col], v);
  }
}

template <typename K, int BITS>
__device__ __forceinline__ void load_state(const DiscreteParams& p, int e, Lds<K, BITS>& l, EnvRegs& r, int lane,
                                  bool need_boxes, bool need_leaves) {
  const K* g_ems = reinterpret_cast<const K*>(p.ems) + (size_t)e * p.ems_stride;
  const K* g_box = reinterpret_cast<const K*>(p.boxes) + (size_t)e * p.I;
  const K* g_leaf = reinterpret_cast<const K*>(p.leaves) + (size_t)e * p.L;
  const int16_t* g_h = p.hmap + (size_t)e * p.AA;
  const int32_t* sc = p.scalars + (size_t)e * PCT_SCALARS;
  // the first 128 EMS words and the heightmap do not wait for the scalars: their loads go out
  // first (reading past n_ems stays inside the env's slice), so that one memory round trip covers
  // the whole state
  const K e0 = g_ems[lane];
  const K e1 = (lane + 64 < p.ems_stride) ? g_ems[lane + 64] : (K)0;
  const int16_t h0 = lane < p.AA ? g_h[lane] : (int16_t)0;
  const int16_t h1 = lane + 64 < p.AA ? g_h[lane + 64] : (int16_t)0;
  r.n_ems = sc[0]; r.n_boxes = sc[1]; r.n_leaf = sc[2];
  r.item0 = sc[3]; r.item1 = sc[4]; r.item2 = sc[5];
  r.t = (uint32_t)sc[6];
  r.flags = p.flags[e];
  r.cursor = ((uint64_t)(uint32_t)sc[9] << 32) | (uint32_t)sc[8];
  r.vol = (int64_t)(((uint64_t)(uint32_t)sc[11] << 32) | (uint32_t)sc[10]);
  r.traj = sc[12];
  r.oc = (uint32_t)sc[13];
  r.mt_pos = sc[7];
  r.n_cand = sc[14];
  r.den_cur = 1.0;
  if (p.rng_numpy) {
    const uint32_t* gm = p.mt + (size_t)e * 624;
    for (int i = lane; i < 624; i += 64) l.mt[i] = gm[i];
    if (p.setting == 3) r.den_cur = p.mt_den[e];
  }
  const int n_fit = r.n_ems < p.ems_cap ? r.n_ems : p.ems_cap;  // a longer list belongs to the retry pass (caller checks)
  if (lane < n_fit) l.ems_a[lane] = e0;
  if (lane + 64 < n_fit) l.ems_a[lane + 64] = e1;
  for (int i = 128 + lane; i < n_fit; i += 64) l.ems_a[i] = g_ems[i];
  if (need_boxes)
    for (int i = lane; i < r.n_boxes; i += 64) l.box[i] = g_box[i];
  if (need_leaves)
    for (int i = lane; i < r.n_leaf; i += 64) l.leaf[i] = g_leaf[i];
  if (lane < p.AA) l.hmap[lane] = (typename Lds<K, BITS>::HT)h0;
  if (lane + 64 < p.AA) l.hmap[lane + 64] = (typename Lds<K, BITS>::HT)h1;
  r.box_from = r.n_boxes;
  for (int i = 128 + lane; i < p.AA; i += 64) l.hmap[i] = (typename Lds<K, BITS>::HT)g_h[i];
  r.stab_over = 0;
  r.poly_from = 0;
  if (p.setting != 2) {  // the stability state: the used part of the pools (word 15: entries | vertices << 16)
    const uint32_t pw = (uint32_t)sc[15];
    r.stab_over = stab_load(p.sb, p.I, e, r.n_boxes, (int)(pw & 0xFFFFu), (int)(pw >> 16), l.st, lane) ? 0u : STAB_WHY_LOAD;
    r.poly_from = l.st.n_poly;
  }
  __syncthreads();
}

template <typename K, int BITS>
__device__ __forceinline__ void store_state(const DiscreteParams& p, int e, const Lds<K, BITS>& l, const EnvRegs& r, int lane) {
  K* g_ems = reinterpret_cast<K*>(p.ems) + (size_t)e * p.ems_stride;
  K* g_box = reinterpret_cast<K*>(p.boxes) + (size_t)e * p.I;
  K* g_leaf = reinterpret_cast<K*>(p.leaves) + (size_t)e * p.L;
  int16_t* g_h = p.hmap + (size_t)e * p.AA;
  int32_t* sc = p.scalars + (size_t)e * PCT_SCALARS;
  for (int i = lane; i < r.n_ems; i += 64) g_ems[i] = l.ems_a[i];
  for (int i = r.box_from + lane; i < r.n_boxes; i += 64) g_box[i] = l.box[i];
  for (int i = lane; i < r.n_leaf; i += 64) g_leaf[i] = l.leaf[i];
  for (int i = lane; i < p.AA; i += 64) g_h[i] = (int16_t)l.hmap[i];
  if (p.rng_numpy) {
    uint32_t* gm = p.mt + (size_t)e * 624;
    for (int i = lane; i < 624; i += 64) gm[i] = l.mt[i];
  }
  if (p.setting != 2) stab_store(p.sb, p.I, e, r.n_boxes, l.st, r.poly_from, lane);
  // the 16 scalar words go out as ONE 64-byte store (lane = word; the values are wave-uniform, a select chain puts each into
  // its lane) instead of 16 single-lane stores
  {
    const int32_t w[PCT_SCALARS] = {
        r.n_ems, r.n_boxes, r.n_leaf, r.item0, r.item1, r.item2, (int32_t)r.t, r.mt_pos,
        (int32_t)(uint32_t)r.cursor, (int32_t)(uint32_t)(r.cursor >> 32), (int32_t)(uint32_t)(uint64_t)r.vol,
        (int32_t)(uint32_t)((uint64_t)r.vol >> 32), r.traj, (int32_t)r.oc, r.n_cand,
        p.setting != 2 ? (int32_t)((uint32_t)l.st.n_ent | ((uint32_t)l.st.n_poly << 16)) : 0};
    int32_t v = 0;
#pragma unroll
    for (int i = 0; i < PCT_SCALARS; i++) v = lane == i ? w[i] : v;
    if (lane < PCT_SCALARS) sc[lane] = v;
  }
  if (lane == 0) {
    if (p.rng_numpy && p.setting == 3) p.mt_den[e] = r.den_cur;
    p.flags[e] = r.flags;
  }
}

// ---- heuristic.py: the placement rules of the heuristic baselines as in-env policies ----------
// rotation convention of heuristic.py:260-271 (shared by all of them; not EMSPoint's)
__device__ __forceinline__ void heur_rot(int b0, int b1, int b2, int rot, int& x, int& y, int& z) {
  switch (rot) {
    case 0: x = b0; y = b1; z = b2; break;
    case 1: y = b0; x = b1; z = b2; break;
    case 2: z = b0; x = b1; y = b2; break;
    case 3: z = b0; y = b1; x = b2; break;
    case 4: x = b0; z = b1; y = b2; break;
    default: y = b0; z = b1; x = b2; break;
  }
}

// Picks the placement heuristic `kind` would step with (heuristic.py; the tests hold a sequential
// CPU restatement of the same loops).  Every lane evaluates its own candidates; the winner
// is the lexicographic minimum of (score, position in the reference's loop order), which is what
// "replace on strictly better" leaves.  Returns false if there is no feasible placement.
template <typename K, int BITS, bool STAB>
__device__ __forceinline__ bool heur_choose(const DiscreteParams& p, int e, Lds<K, BITS>& l, const EnvRegs& r, int lane, int kind,
                                   int& olx, int& oly, int& ox, int& oy, int& oz, uint32_t& stab_err) {
  typedef Pack<K, BITS> P;
  const int orient = STAB ? 2 : 6;
  const int b0 = r.item0, b1 = r.item1, b2 = r.item2;
  const double den = STAB ? next_density(p, e, r.oc - 1, r.traj, r.cursor - 1) : 1.0;
  uint32_t* scratch = reinterpret_cast<uint32_t*>(l.tab0);  // the table region is idle between observations
  // drop_box_virtual (D/space.py:393-433) of size (x,y,z) at (lx,ly): feasibility, height, and the sum of
  // the heightmap under the footprint (NumPy clips the slice to the array)
  // ALL 64 lanes call (`go`: this lane has a placement to test): the stability check is wave-cooperative
  auto probe = [&](bool go, int x, int y, int z, int lx, int ly, int& mh, long long& under) __attribute__((always_inline)) -> bool {
    mh = 0;
    under = 0;
    if (go) {
      const int xe = min(lx + x, p.A), ye = min(ly + y, p.A);
      for (int cx = lx; cx < xe; cx++)
        for (int cy = ly; cy < ye; cy++) {
          int hh = l.hmap[cx * p.A + cy];
          mh = hh > mh ? hh : mh;
          under += hh;
        }
    }
    bool feas = go && (lx + x <= p.W) && (ly + y <= p.Ly) && (mh + z <= p.H);
    if (STAB) {
      const bool need = feas && mh != 0;
      if (__ballot(need)) {
        const double cand[9] = {(double)lx, (double)ly, (double)mh, (double)(lx + x), (double)(ly + y), (double)(mh + z),
                                (double)x, (double)y, (double)z};
        BoxGeo<K, BITS> geo{l.box};
        uint32_t cap = 0;
        bool ill = false, lerr = false;
        const bool stable = stab_virtual_wave<false>(geo, l.st, r.n_boxes, need, cand, den, l.sw, lane, cap, lerr, ill);
        if (need) feas = stable && !cap;
        stab_err |= cap;
        if (__ballot(need && lerr)) stab_err |= STAB_WHY_SPLIT;
        if (__ballot(ill)) stab_err |= STAB_NOTE_ILL;  // a near-cut rank decision inside a heuristic's probe
      }
    }
    return feas;
  };
  const uint64_t NONE = ~0ull;
  uint64_t best = NONE;
  if (kind == PCT_HEUR_DBL || kind == PCT_HEUR_HM || kind == PCT_HEUR_RANDOM) {
    // :431-498 / :232-298: every (lx, ly) the UNROTATED item fits at, every rotation
    long long total = 0;
    if (kind == PCT_HEUR_HM) {
      long long part = 0;
      for (int c = lane; c < p.W * p.Ly; c += 64) part += l.hmap[(c / p.Ly) * p.A + (c % p.Ly)];
      total = wave_sum_i64(part);
    }
    const int nx = p.W - b0 + 1, ny = p.Ly - b1 + 1;
    const int NQ = (nx > 0 && ny > 0) ? nx * ny * orient : 0;
    if (kind == PCT_HEUR_RANDOM) {
      // :300-362 random: the pct_mix32(g, t) % n -th feasible placement of the same enumeration (the
      // reference's np.random.randint(0, n)); the feasibility masks of the chunks wait in the scratch
      int n = 0;
      for (int base = 0; base < NQ; base += 64) {
        int q = base + lane;
        bool feas = false;
        {
          int rot = q % orient, cell = q / orient;
          int lx = cell / ny, ly = cell - lx * ny;
          int x, y, z, mh;
          long long under;
          heur_rot(b0, b1, b2, rot, x, y, z);
          feas = probe(q < NQ, x, y, z, lx, ly, mh, under);
        }
        const uint64_t m = __ballot(feas);
        if (lane == 0) { scratch[(base >> 6) * 2] = (uint32_t)m; scratch[(base >> 6) * 2 + 1] = (uint32_t)(m >> 32); }
        n += __popcll(m);
      }
      __syncthreads();
      if (n == 0) return false;
      int pick = (int)(pct_mix32((uint32_t)(p.env_id_base + e), r.t) % (uint32_t)n);
      int q = -1;
      for (int base = 0; base < NQ && q < 0; base += 64) {
        uint64_t m = ((uint64_t)scratch[(base >> 6) * 2 + 1] << 32) | scratch[(base >> 6) * 2];
        int c = __popcll(m);
        if (pick < c) {
          for (int k2 = 0; k2 < pick; k2++) m &= m - 1;  // drop the `pick` lowest set bits
          q = base + __ffsll((unsigned long long)m) - 1;
        } else {
          pick -= c;
        }
      }
      __syncthreads();
      int rot = q % orient, cell = q / orient;
      olx = cell / ny;
      oly = cell - olx * ny;
      heur_rot(b0, b1, b2, rot, ox, oy, oz);
      return true;
    }
    for (int base = 0; base < NQ; base += 64) {
      int q = base + lane;
      {
        int rot = q % orient, cell = q / orient;
        int lx = cell / ny, ly = cell - lx * ny;
        int x, y, z, mh;
        long long under;
        heur_rot(b0, b1, b2, rot, x, y, z);
        if (probe(q < NQ, x, y, z, lx, ly, mh, under)) {
          long long score = kind == PCT_HEUR_DBL ? (long long)(lx + ly) + 100ll * mh
                                                 : (long long)(lx + ly) + 100ll * (total - under + (long long)(mh + z) * x * y);
          uint64_t key = ((uint64_t)score << 24) | (uint64_t)q;
          best = key < best ? key : best;
        }
      }
    }
    best = wave_min_u64(best);
    if (best == NONE) return false;
    int q = (int)(best & 0xFFFFFFull);
    int rot = q % orient, cell = q / orient;
    olx = cell / ny;
    oly = cell - olx * ny;
    heur_rot(b0, b1, b2, rot, ox, oy, oz);
    return true;
  }
  const int E = r.n_ems;
  const int NQ = E * orient;
  if (kind == PCT_HEUR_OBPH) {
    // :364-425: EMS in (z, y, x) order (stable), first feasible (corner, rotation); no fit-in-EMS test
    for (int base = 0; base < NQ; base += 64) {
      int q = base + lane;
      {
        const bool live = q < NQ;
        int ei = q / orient, rot = q - ei * orient;
        K ek = live ? l.ems_a[ei] : (K)0;
        int x, y, z, mh;
        long long under;
        heur_rot(b0, b1, b2, rot, x, y, z);
        if (probe(live, x, y, z, P::get(ek, 0), P::get(ek, 1), mh, under)) {
          uint64_t key = ((uint64_t)P::get(ek, 2) << 50) | ((uint64_t)P::get(ek, 1) << 40) | ((uint64_t)P::get(ek, 0) << 30) |
                         (uint64_t)q;
          best = key < best ? key : best;
        }
      }
    }
    best = wave_min_u64(best);
    if (best == NONE) return false;
    int q = (int)(best & 0x3FFFFFFFull);
    int ei = q / orient, rot = q - ei * orient;
    K ek = l.ems_a[ei];
    olx = P::get(ek, 0);
    oly = P::get(ek, 1);
    heur_rot(b0, b1, b2, rot, ox, oy, oz);
    return true;
  }
  if (kind == PCT_HEUR_BR) {
    // :500-569: best eval_ems (volume + item types that fit unrotated + 10 if all do), first in order
    for (int base = 0; base < E; base += 64) {
      int ei = base + lane;
      if (ei < E) {
        K ek = l.ems_a[ei];
        int dx = P::get(ek, 3) - P::get(ek, 0), dy = P::get(ek, 4) - P::get(ek, 1), dz = P::get(ek, 5) - P::get(ek, 2);
        int fits = 0;
        for (int i = 0; i < p.n_items; i++)
          fits += (dx >= p.item_set[3 * i] && dy >= p.item_set[3 * i + 1] && dz >= p.item_set[3 * i + 2]) ? 1 : 0;
        scratch[ei] = (uint32_t)(dx * dy * dz + fits + (fits == p.n_items ? 10 : 0));
      }
    }
    __syncthreads();
    for (int base = 0; base < NQ; base += 64) {
      int q = base + lane;
      {
        const bool live = q < NQ;
        int ei = q / orient, rot = q - ei * orient;
        K ek = live ? l.ems_a[ei] : (K)0;
        int dx = P::get(ek, 3) - P::get(ek, 0), dy = P::get(ek, 4) - P::get(ek, 1), dz = P::get(ek, 5) - P::get(ek, 2);
        int x, y, z, mh;
        long long under;
        heur_rot(b0, b1, b2, rot, x, y, z);
        if (probe(live && dx >= x && dy >= y && dz >= z, x, y, z, P::get(ek, 0), P::get(ek, 1), mh, under)) {
          uint64_t key = ((uint64_t)(0xFFFFFFFFu - scratch[live ? ei : 0]) << 24) | (uint64_t)q;
          best = key < best ? key : best;
        }
      }
    }
    best = wave_min_u64(best);
    __syncthreads();
    if (best == NONE) return false;
    int q = (int)(best & 0xFFFFFFull);
    int ei = q / orient, rot = q - ei * orient;
    K ek = l.ems_a[ei];
    olx = P::get(ek, 0);
    oly = P::get(ek, 1);
    heur_rot(b0, b1, b2, rot, ox, oy, oz);
    return true;
  }
  if (kind == PCT_HEUR_MACS) {
    // :11-136 MACS: maximise the sum, over the levels below the item's base, of the largest empty
    // rectangle of the level after the placement.  The heuristic's voxel container is nonzero exactly
    // below the heightmap, so (i, j) is empty at level h iff hmap[i][j] <= h -- one row mask per
    // (level, row), built once per step in the idle table region; the candidate's footprint is cleared
    // from the rows it covers.  Largest rectangle of a level: for every band of rows i1..i2 the AND of
    // their masks, times the longest run of ones.
    // Row masks are 32 bits wide for Ly <= 32, 64 bits wide up to Ly = 64 and multi-word beyond (macs_wide).
    auto macs = [&](auto mask_tag) __attribute__((always_inline)) {
      typedef decltype(mask_tag) M;
      constexpr int MB = (int)sizeof(M) * 8;
      M* rows = reinterpret_cast<M*>(scratch);  // [H][W]
      for (int c = lane; c < p.H * p.W; c += 64) {
        const int lv = c / p.W, i = c - lv * p.W;
        M m = 0;
        for (int j = 0; j < p.Ly; j++) m |= ((int)l.hmap[i * p.A + j] <= lv) ? ((M)1 << j) : (M)0;
        rows[c] = m;
      }
      __syncthreads();
      const int NC = NQ * 4;
      for (int base = 0; base < NC; base += 64) {
        int qc = base + lane;
        const bool live = qc < NC;
        int q = qc >> 2, corner = qc & 3;
        int ei = q / orient, rot = q - ei * orient;
        K ek = live ? l.ems_a[ei] : (K)0;
        int dx = P::get(ek, 3) - P::get(ek, 0), dy = P::get(ek, 4) - P::get(ek, 1), dz = P::get(ek, 5) - P::get(ek, 2);
        int x, y, z, mh;
        long long under;
        heur_rot(b0, b1, b2, rot, x, y, z);
        int lx = (corner & 1) ? P::get(ek, 3) - x : P::get(ek, 0);
        int ly = (corner & 2) ? P::get(ek, 4) - y : P::get(ek, 1);
        if (probe(live && dx >= x && dy >= y && dz >= z, x, y, z, lx, ly, mh, under)) {
          const M foot = (M)((y >= MB ? (M)0 : ((M)1 << y)) - (M)1) << ly;
          uint32_t score = 0;
          for (int lv = 0; lv < mh; lv++) {
            const M* rw = rows + lv * p.W;
            uint32_t level_max = 0;
            for (int i1 = 0; i1 < p.W; i1++) {
              M band = ~(M)0;
              for (int i2 = i1; i2 < p.W; i2++) {
                M rm = rw[i2];
                if (i2 >= lx && i2 < lx + x) rm &= ~foot;
                band &= rm;
                if (!band) break;
                M t = band;
                uint32_t run = 0;
                while (t) { t &= t << 1; run++; }
                const uint32_t area = run * (uint32_t)(i2 - i1 + 1);
                level_max = area > level_max ? area : level_max;
              }
            }
            score += level_max;
          }
          uint64_t key = ((uint64_t)(0xFFFFFFFFu - score) << 24) | (uint64_t)qc;
          best = key < best ? key : best;
        }
      }
    };
    // Beyond 64 cells along y (round 4): the same sweep on multi-word masks, NW = ceil(Ly / 64) words of 64 bits per (level,
    // row); the band, its run-length copy and the footprint are small per-lane arrays (a coverage path, not a tuned one).
    auto macs_wide = [&]() __attribute__((always_inline)) {
      constexpr int MAXW = 16;  // Ly <= 1023
      const int NW = (p.Ly + 63) >> 6;
      uint64_t* rows = reinterpret_cast<uint64_t*>(scratch);  // [H][W][NW]
      for (int c = lane; c < p.H * p.W * NW; c += 64) {
        const int lv = c / (p.W * NW), rem = c - lv * p.W * NW, i = rem / NW, wd = rem - i * NW;
        uint64_t m = 0;
        for (int j = wd * 64; j < p.Ly && j < wd * 64 + 64; j++) m |= ((int)l.hmap[i * p.A + j] <= lv) ? (1ull << (j - wd * 64)) : 0ull;
        rows[c] = m;
      }
      __syncthreads();
      const int NC = NQ * 4;
      for (int base = 0; base < NC; base += 64) {
        int qc = base + lane;
        const bool live = qc < NC;
        int q = qc >> 2, corner = qc & 3;
        int ei = q / orient, rot = q - ei * orient;
        K ek = live ? l.ems_a[ei] : (K)0;
        int dx = P::get(ek, 3) - P::get(ek, 0), dy = P::get(ek, 4) - P::get(ek, 1), dz = P::get(ek, 5) - P::get(ek, 2);
        int x, y, z, mh;
        long long under;
        heur_rot(b0, b1, b2, rot, x, y, z);
        int lx = (corner & 1) ? P::get(ek, 3) - x : P::get(ek, 0);
        int ly = (corner & 2) ? P::get(ek, 4) - y : P::get(ek, 1);
        if (probe(live && dx >= x && dy >= y && dz >= z, x, y, z, lx, ly, mh, under)) {
          uint64_t foot[MAXW], band[MAXW], tt[MAXW];
          for (int wd = 0; wd < NW; wd++) {  // bits [ly, ly + y) of the row
            const int lo = ly - wd * 64, hi = ly + y - wd * 64;
            const uint64_t a = lo <= 0 ? ~0ull : (lo >= 64 ? 0ull : (~0ull << lo));
            const uint64_t b = hi >= 64 ? ~0ull : (hi <= 0 ? 0ull : ((1ull << hi) - 1ull));
            foot[wd] = a & b;
          }
          uint32_t score = 0;
          for (int lv = 0; lv < mh; lv++) {
            const uint64_t* rw = rows + (size_t)lv * p.W * NW;
            uint32_t level_max = 0;
            for (int i1 = 0; i1 < p.W; i1++) {
              for (int wd = 0; wd < NW; wd++) band[wd] = ~0ull;
              for (int i2 = i1; i2 < p.W; i2++) {
                uint64_t any = 0;
                for (int wd = 0; wd < NW; wd++) {
                  uint64_t rm = rw[i2 * NW + wd];
                  if (i2 >= lx && i2 < lx + x) rm &= ~foot[wd];
                  band[wd] &= rm;
                  any |= band[wd];
                  tt[wd] = band[wd];
                }
                if (!any) break;
                uint32_t run = 0;
                while (any) {  // longest run of ones: t &= t << 1 across the words until nothing is left
                  any = 0;
                  for (int wd = NW - 1; wd >= 0; wd--) {
                    const uint64_t sh = (tt[wd] << 1) | (wd > 0 ? (tt[wd - 1] >> 63) : 0ull);
                    tt[wd] &= sh;
                    any |= tt[wd];
                  }
                  run++;
                }
                const uint32_t area = run * (uint32_t)(i2 - i1 + 1);
                level_max = area > level_max ? area : level_max;
              }
            }
            score += level_max;
          }
          uint64_t key = ((uint64_t)(0xFFFFFFFFu - score) << 24) | (uint64_t)qc;
          best = key < best ? key : best;
        }
      }
    };
    if (p.Ly <= 32) macs((uint32_t)0);
    else if (p.Ly <= 64) macs((uint64_t)0);
    else macs_wide();
    best = wave_min_u64(best);
    __syncthreads();
    if (best == NONE) return false;
    int qc = (int)(best & 0xFFFFFFull);
    int q = qc >> 2, corner = qc & 3;
    int ei = q / orient, rot = q - ei * orient;
    K ek = l.ems_a[ei];
    heur_rot(b0, b1, b2, rot, ox, oy, oz);
    olx = (corner & 1) ? P::get(ek, 3) - ox : P::get(ek, 0);
    oly = (corner & 2) ? P::get(ek, 4) - oy : P::get(ek, 1);
    return true;
  }
  // :138-226 LASH: least surface area of the bounding box of everything packed so far
  int maxX = 0, maxY = 0, minX = p.W, minY = p.Ly;
  for (int i = 0; i < r.n_boxes; i++) {  // wave-uniform scan of the placed boxes
    K bk = uniform_key<K>(l.box[i]);
    maxX = max(maxX, P::get(bk, 3)); maxY = max(maxY, P::get(bk, 4));
    minX = min(minX, P::get(bk, 0)); minY = min(minY, P::get(bk, 1));
  }
  const uint32_t init = (uint32_t)(p.W * p.Ly + p.Ly * p.H + p.H * p.W);
  uint32_t smin = 0xFFFFFFFFu;
  for (int base = 0; base < NQ; base += 64) {
    int q = base + lane;
    uint32_t sc = 0xFFFFFFFFu;
    {
      const bool live = q < NQ;
      int ei = q / orient, rot = q - ei * orient;
      K ek = live ? l.ems_a[ei] : (K)0;
      int dx = P::get(ek, 3) - P::get(ek, 0), dy = P::get(ek, 4) - P::get(ek, 1), dz = P::get(ek, 5) - P::get(ek, 2);
      int x, y, z, mh;
      long long under;
      heur_rot(b0, b1, b2, rot, x, y, z);
      int lx = P::get(ek, 0), ly = P::get(ek, 1);
      if (probe(live && dx >= x && dy >= y && dz >= z, x, y, z, lx, ly, mh, under)) {
        int ex = max(lx + x, maxX) - min(lx, minX), ey = max(ly + y, maxY) - min(ly, minY);
        sc = (uint32_t)(ex * ey + (mh + z) * ey + (mh + z) * ex);
      }
      if (live) scratch[q] = sc;
    }
    smin = sc < smin ? sc : smin;
  }
  smin = (uint32_t)(wave_min_u64((uint64_t)smin));
  __syncthreads();
  bool found = false;
  if (smin < init) {  // a score equal to the initial bound is never taken (:195-199 needs a best already)
    int bd0 = 0, bd1 = 0, bd2 = 0;
    for (int base = 0; base < NQ; base += 64) {
      int q = base + lane;
      uint64_t m = __ballot(q < NQ && scratch[q] == smin);
      while (m) {  // the candidates that tie on the best score, in loop order (:195-199)
        int bit = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        int qq = base + bit;
        int ei = qq / orient, rot = qq - ei * orient;
        K ek = uniform_key<K>(l.ems_a[ei]);
        int dx = P::get(ek, 3) - P::get(ek, 0), dy = P::get(ek, 4) - P::get(ek, 1), dz = P::get(ek, 5) - P::get(ek, 2);
        int x, y, z;
        heur_rot(b0, b1, b2, rot, x, y, z);
        bool take = !found;
        if (found) take = min(min(dx - x, dy - y), dz - z) < min(min(bd0 - x, bd1 - y), bd2 - z);
        if (take) {
          found = true;
          olx = P::get(ek, 0); oly = P::get(ek, 1); ox = x; oy = y; oz = z;
          bd0 = dx; bd1 = dy; bd2 = dz;
        }
      }
    }
  }
  __syncthreads();
  return found;
}

// One transition of one env with the state resident in LDS (D/bin3D.py:151-188 plus the
// VecEnv worker's auto-reset).  (flag, lx, ly) + (bx, by, bz) is the decoded action.
template <typename K, int BITS, bool STAB, int SCHEME, int RNG, typename TM>
__device__ __forceinline__ bool transition(const DiscreteParams& p, int e, Lds<K, BITS>& l, EnvRegs& r, int lane, bool bad,
                                  int flag, int lx, int ly, int bx, int by, int bz, TM& tm, bool giveup = false) {
  typedef Pack<K, BITS> P;
  constexpr bool SHUFFLE = (RNG & 1) != 0, MT = (RNG & 2) != 0;
  r.t++;
  int x = flag ? by : bx, y = flag ? bx : by, z = bz;  // D/space.py:348-351
  bool ok = !bad && !giveup;  // giveup: a heuristic found no placement -- the episode ends without a step()
  int max_h = 0;
  // the density shown with the observation this action answers (D/bin3D.py:158 self.next_den)
  const double item_den = !STAB ? 1.0 : (MT ? (p.setting == 3 ? r.den_cur : 1.0) : next_density(p, e, r.oc - 1, r.traj, r.cursor - 1));
  if (ok) {
    // np.max(plain[lx:lx+x, ly:ly+y]) with Python slice normalisation (D/space.py:354-355)
    int xa = lx, xb = lx + x, ya = ly, yb = ly + y;
    if (xa < 0) { xa += p.A; if (xa < 0) xa = 0; }
    if (xb < 0) { xb += p.A; if (xb < 0) xb = 0; }
    if (ya < 0) { ya += p.A; if (ya < 0) ya = 0; }
    if (yb < 0) { yb += p.A; if (yb < 0) yb = 0; }
    xa = min(xa, p.A); xb = min(xb, p.A); ya = min(ya, p.A); yb = min(yb, p.A);
    if (xb <= xa || yb <= ya) {
      ok = false;  // empty slice: np.max raises ValueError
      r.flags |= PCT_FLAG_BAD_ACTION;
    } else if (lx < 0 || ly < 0) {
      ok = false;  // check_box D/space.py:440-441 rejects it whatever max_h is
    } else {
      int m = 0;  // 8x8 tiles of the footprint, one cell per lane
      for (int tx = xa; tx < xb; tx += 8)
        for (int ty = ya; ty < yb; ty += 8) {
          int cx = tx + (lane >> 3), cy = ty + (lane & 7);
          if (cx < xb && cy < yb) {
            int h = l.hmap[cx * p.A + cy];
            m = h > m ? h : m;
          }
        }
      max_h = wave_max_i32(m);
      // check_box D/space.py:436-446 (setting 2)
      ok = !(lx + x > p.W || ly + y > p.Ly) && !(max_h + z > p.H);
    }
  }
  if (STAB && ok && r.n_boxes < p.I) {
    // check_box :447-454: box_now.calculated_impact() -- one lane records the new box in the LDS-resident stability
    // state, walks the support graph and commits the new shares / stacks (the box is only kept if the verdict is True;
    // a box on the floor is recorded and accepted without a walk)
    if (lane == 0) l.box[r.n_boxes] = P::pack(lx, ly, max_h, lx + x, ly + y, max_h + z);
    __syncthreads();
    StabStats cstats = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (timed build only)
    BoxGeo<K, BITS> geo{l.box};
    bool ill = false;
    // lane 0 walks; a split over six and more supporters is solved by the whole wave (pct_stab.cuh stab_commit_wave)
    const int rc = stab_commit_wave<false>(geo, l.st, r.n_boxes, item_den, l.sw, lane, ill, TM::on ? &cstats : nullptr);
    l.st.n_ent = __builtin_amdgcn_readfirstlane(l.st.n_ent);
    l.st.n_poly = __builtin_amdgcn_readfirstlane(l.st.n_poly);
    if (__builtin_amdgcn_readfirstlane(ill ? 1 : 0)) r.flags |= PCT_FLAG_ILL_CONDITIONED | PCT_FLAG_ILL_COMMIT;
    if (rc < 0) r.stab_over |= STAB_WHY_COMMIT;
    ok = rc == 1;
    if (TM::on) {
      tm.add(ST_STAB_COMMIT_VISITS, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.commit_visits));
      tm.add(ST_STAB_LSQ3, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsq3));
      tm.add(ST_STAB_LSQ4, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsq4));
      tm.add(ST_STAB_LSQ5, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsq5));
      tm.add(ST_STAB_LSQX, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsqx));
      tm.add(ST_STAB_LSQ_ROUNDS, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsq_rounds));
      tm.add(ST_STAB_COMMIT_ROUNDS, (uint64_t)__builtin_amdgcn_readfirstlane(cstats.lsq_rounds));
    }
    __syncthreads();
  }
  if (ok && r.n_boxes >= p.I) {  // IndexError at D/space.py:385
    ok = false;
    r.flags |= PCT_FLAG_INTERNAL_OVERFLOW;
  }
  float reward;
  uint8_t done;
  int counter;
  double ratio = 0.0;
  const double binvol = (double)((int64_t)p.W * p.Ly * p.H);
  if (ok) {
    int top = max_h + z;
    for (int tx = lx; tx < lx + x; tx += 8)
      for (int ty = ly; ty < ly + y; ty += 8) {
        int cx = tx + (lane >> 3), cy = ty + (lane & 7);
        if (cx < lx + x && cy < ly + y) l.hmap[cx * p.A + cy] = (typename Lds<K, BITS>::HT)top;
      }
    if (lane == 0) l.box[r.n_boxes] = P::pack(lx, ly, max_h, lx + x, ly + y, top);
    r.n_boxes++;
    r.vol += (int64_t)x * y * z;
    __syncthreads();
    tm.tick(PH_DROP);
    if (SCHEME == 0) genems<K, BITS>(p, l, r, lane, lx, ly, max_h, lx + x, ly + y, top);  // D/bin3D.py:172-175
    tm.tick(PH_GENEMS);
    // D/bin3D.py:57-59,183: 10 * vol(item) / vol(bin), float64 then envs.py:181 .float()
    reward = (float)(((double)((int64_t)r.item0 * r.item1 * r.item2) / binvol) * 10.0);
    done = 0;
    counter = r.n_boxes;
  } else {
    reward = 0.f;
    done = 1;
    counter = r.n_boxes;
    ratio = (double)r.vol / binvol;  // D/space.py:334-339
    if (!giveup) r.oc++;  // the terminal step's own (discarded) observation consumed a shuffle too (D/bin3D.py:165)
    if (MT && !giveup) {
      // ... and in strict NumPy-stream mode its draws: cur_observation() of the unchanged bin and item redraws the
      // density (setting 3) and shuffles the same candidate list once more -- n_cand - 1 random_interval draws
      if (p.setting == 3) (void)mt_density(l, r, lane);
      if (SHUFFLE)
        for (int i = r.n_cand - 1; i >= 1; i--) (void)mt_interval(l, r, lane, (uint32_t)i);
    }
    __syncthreads();
    space_reset<K, BITS>(p, l, r, lane);  // shmem_vec_env.py:141-143 -> D/bin3D.py:61-67
    __syncthreads();
    tm.tick(PH_DROP);
  }
  if (MT) {
    draw_item_mt(p, l, r, lane);  // np.random.randint(0, len(box_set)): after a success and after the reset alike
  } else if (r.cursor == r.pre_cursor && r.traj == r.pre_traj) {  // the usual case: the prefetched draw
    r.item0 = r.pre0; r.item1 = r.pre1; r.item2 = r.pre2;
    r.cursor++;
  } else {
    draw_item(p, e, r);
  }
  if (lane == 0) {
    p.reward[e] = reward;
    p.done[e] = done;
    p.counter[e] = counter;
    p.ratio[e] = ratio;
    if (p.mask) p.mask[e] = done ? 0.f : 1.f;  // train_tools.py:70 masks = 1 - done
  }
  return done != 0;  // true: the episode ended and the env was reset
}

// D/bin3D.py:139-149 LeafNode2Action for a leaf given as six integers (zero row -> (0,0,0)
// with the unrotated item)
__device__ __forceinline__ void decode_leaf(const EnvRegs& r, bool zero_row, int xs, int ys, int xe, int ye, bool& bad,
                                   int& lx, int& ly, int& bx, int& by, int& bz) {
  bad = false;
  if (zero_row) {
    lx = 0; ly = 0; bx = r.item0; by = r.item1; bz = r.item2;
    return;
  }
  int x = xe - xs, y = ye - ys;
  int z0 = r.item0, z1 = r.item1, z2 = r.item2;
  // z = list(next_box); z.remove(x); z.remove(y); z = z[0]
  int a, b;  // the two survivors after removing x
  if (z0 == x) { a = z1; b = z2; }
  else if (z1 == x) { a = z0; b = z2; }
  else if (z2 == x) { a = z0; b = z1; }
  else { bad = true; a = b = 0; }
  int zz = 0;
  if (!bad) {
    if (a == y) zz = b;
    else if (b == y) zz = a;
    else bad = true;
  }
  lx = xs; ly = ys; bx = x; by = y; bz = zz;
}

enum { ACT_ROWS = 0, ACT_INDEX = 1, ACT_HASH = 2, ACT_RESET = 3, ACT_HEUR = 4 /* row_len = PCT_HEUR_* */ };

// Stand-in policy epilogue (pct_bind_policy_rows): the float32 [9] leaf row pct_policy_hash_rows_kernel would gather from
// the observation just written -- k = number of valid leaves (the mask column, tools.py:103), leaf pct_mix32(g, t) % k,
// the all-zero row when there is none (train_tools.py:66) -- written by the transition itself, so that a benchmark
// loop needs no policy dispatch between two transitions.  Same bytes as the separate kernel's.
template <typename K, int BITS>
__device__ __forceinline__ void policy_epilogue(const DiscreteParams& p, int e, const Lds<K, BITS>& l, const EnvRegs& r, int lane) {
  typedef Pack<K, BITS> P;
  const int k = r.n_leaf;
  const int li = k > 0 ? (int)(pct_mix32((uint32_t)(p.env_id_base + e), r.t) % (uint32_t)k) : 0;
  if (lane < 9) {
    const bool on = k > 0;
    const K kk = on ? l.leaf[li] : (K)0;
    const float v = lane < 5 ? (float)P::get(kk, lane) : (lane == 5 ? (on ? (float)p.H : 0.f) : (lane == 8 ? (on ? 1.0f : 0.f) : 0.f));
    p.policy_rows[(size_t)e * 9 + lane] = v;
  }
}

// one env, one launch's worth of transitions (the body of the kernel below)
template <typename K, int BITS, int ACT, bool TIMED, bool STAB, int SCHEME, int RNG>
__device__ __forceinline__ void discrete_env_steps(const DiscreteParams& p, const void* actions, int row_len, int n_steps,
                                          int e, unsigned char* smem, int& ems_out) {
  ems_out = 0;  // the live EMS count the env is left with (half of the heavy-first dispatch's sort key)
  const int lane = threadIdx.x;
  Lds<K, BITS> l = carve_lds<K, BITS>(p, smem + PCT_LDS_STASH);
  EnvRegs r;
  PhaseTimer<TIMED> tm;
  tm.start();
  // the placed-box list is only read by the stability check, the CP / EP schemes, LASH and a full
  // rewrite of the observation; the leaf list only by the index / stand-in actions
  // the action row is fetched before the state so that both loads share one memory round trip
  float act_v = 0.f;
  if (ACT == ACT_ROWS) {
    const float* row = reinterpret_cast<const float*>(actions) + (size_t)e * row_len;
    act_v = lane < row_len ? row[lane] : 0.f;
  }
  const bool need_boxes = STAB || SCHEME != 0 || ACT == ACT_HEUR || p.full_obs != 0;
  load_state<K, BITS>(p, e, l, r, lane, need_boxes, ACT == ACT_INDEX || ACT == ACT_HASH);
  {  // the next draw of the item source does not depend on this step's outcome (except after a
     // dataset reset): issue its loads now, use them at the end of the transition
    EnvRegs nx = r;
    if (!(RNG & 2)) draw_item(p, e, nx);  // (the NumPy stream is consumed strictly in order: no look-ahead)
    r.pre0 = nx.item0; r.pre1 = nx.item1; r.pre2 = nx.item2;
    r.pre_cursor = r.cursor;
    r.pre_traj = r.traj;
  }
  tm.tick(PH_LOAD);
  wave_priority(r.n_ems, p.prio_t);
  float* obs = p.obs + (size_t)e * p.row_len;
  // An env whose EMS list or candidate set outgrows this launch's LDS capacities is handed, state untouched,
  // to the large-capacity retry pass (if there is one) instead of being flagged and terminated: nothing of
  // its state is stored here, and whatever outputs it already wrote are written again, identically up to the
  // point of the overflow and correctly beyond, by the retry.
  const uint32_t flags_in = r.flags;
  // (The stability settings take part since round 3: their state is LDS-resident during the transition, so a requeued
  // env has left nothing of a half-done commit behind.  A stability capacity -- pools, hull workspace, task queue --
  // that is exceeded requeues the env likewise; only where no larger pass exists does it raise the flag.)
  const bool can_retry = p.retry_count != nullptr && !p.retry_mode;
  auto overflowed = [&]() __attribute__((always_inline)) -> bool {
    if (STAB && r.stab_over && !can_retry) {
      r.flags |= PCT_FLAG_STABILITY_OVERFLOW | r.stab_over;
      r.stab_over = 0;
    }
    return can_retry && ((((r.flags & ~flags_in) & (PCT_FLAG_EMS_OVERFLOW | PCT_FLAG_CANDIDATE_OVERFLOW)) != 0) || (STAB && r.stab_over));
  };
  if (can_retry && ((r.n_ems > p.ems_cap && ACT != ACT_RESET) || (STAB && r.stab_over && ACT != ACT_RESET))) {
    retry_enqueue(p.retry_count, p.retry_ids, e);
    return;
  }

  if (ACT == ACT_RESET) {
    space_reset<K, BITS>(p, l, r, lane);
    __syncthreads();
    if (RNG & 2) draw_item_mt(p, l, r, lane);
    else draw_item(p, e, r);
    leaf_nodes<K, BITS, STAB, SCHEME, RNG>(p, e, l, r, lane, tm);
    if (overflowed()) {
      retry_enqueue(p.retry_count, p.retry_ids, e);
      return;
    }
    write_obs<K, BITS>(p, e, l, r, lane, obs, true, -1);
    if (p.policy_rows) policy_epilogue<K, BITS>(p, e, l, r, lane);
    store_state<K, BITS>(p, e, l, r, lane);
    ems_out = r.n_ems;
    return;
  }

  for (int it = 0; it < n_steps; it++) {
    bool bad = false, zero_row = false, giveup = false;
    int flag = 0, lx = 0, ly = 0, bx = 0, by = 0, bz = 0;
    if (ACT == ACT_HEUR) {
      uint32_t serr = 0;
      giveup = !heur_choose<K, BITS, STAB>(p, e, l, r, lane, row_len, lx, ly, bx, by, bz, serr);
      if (STAB && (serr & STAB_NOTE_ILL)) r.flags |= PCT_FLAG_ILL_CONDITIONED;
      if (STAB) r.stab_over |= serr & ~STAB_NOTE_ILL;
    } else if (ACT == ACT_ROWS) {
      float v = act_v;
      float a0 = __shfl(v, 0, 64), a1 = __shfl(v, 1, 64), a2 = __shfl(v, 2, 64), a3 = __shfl(v, 3, 64),
            a4 = __shfl(v, 4, 64), a5 = __shfl(v, 5, 64);
      if (row_len == 3) {  // (flag, lx, ly) with the unrotated item, D/bin3D.py:152-153
        flag = (int)a0; lx = (int)a1; ly = (int)a2;
        bx = r.item0; by = r.item1; bz = r.item2;
      } else {
        float sum = ((((a0 + a1) + a2) + a3) + a4) + a5;  // np.sum(leaf_node[0:6]) == 0
        zero_row = (sum == 0.f);
        decode_leaf(r, zero_row, (int)a0, (int)a1, (int)a0 + (int)(a3 - a0), (int)a1 + (int)(a4 - a1), bad, lx, ly,
                    bx, by, bz);
      }
    } else {
      int64_t li;
      if (ACT == ACT_INDEX) {
        li = reinterpret_cast<const int64_t*>(actions)[e];
      } else {  // stand-in policy: leaf = pct_mix32(g, t) % k over the k valid leaves
        li = r.n_leaf > 0 ? (int64_t)(pct_mix32((uint32_t)(p.env_id_base + e), r.t) % (uint32_t)r.n_leaf) : 0;
      }
      zero_row = !(li >= 0 && li < r.n_leaf);
      K k = zero_row ? (K)0 : l.leaf[li];
      typedef Pack<K, BITS> P;
      decode_leaf(r, zero_row, P::get(k, 0), P::get(k, 1), P::get(k, 3), P::get(k, 4), bad, lx, ly, bx, by, bz);
    }
    if (bad) r.flags |= PCT_FLAG_BAD_ACTION;  // ValueError in list.remove, D/bin3D.py:144-145
    const bool ended = transition<K, BITS, STAB, SCHEME, RNG>(p, e, l, r, lane, bad, flag, lx, ly, bx, by, bz, tm, giveup);
    leaf_nodes<K, BITS, STAB, SCHEME, RNG>(p, e, l, r, lane, tm);
    if (overflowed()) {
      retry_enqueue(p.retry_count, p.retry_ids, e);
      return;
    }
    write_obs<K, BITS>(p, e, l, r, lane, obs, ended || (p.full_obs != 0 && it == 0), ended ? -1 : r.n_boxes - 1);
    __syncthreads();
    tm.tick(PH_OBS);
  }
  if (p.policy_rows) policy_epilogue<K, BITS>(p, e, l, r, lane);
  store_state<K, BITS>(p, e, l, r, lane);
  ems_out = r.n_ems;
  tm.tick(PH_STORE);
  if (TIMED && lane == 0) tm.flush(p.timing + (size_t)e * PCT_TIMING_SLOTS, n_steps);
}

template <typename K, int BITS, int ACT, bool TIMED, bool STAB, int SCHEME, int RNG>
// the plain setting-2 kernels are held to 128 VGPRs (4 waves per SIMD = 16 resident envs per CU, the
// occupancy the LDS layout is sized for); the float64 stability code and the timed build are not
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(STAB ? PCT_STAB_WAVES : 4)))
pct_discrete_kernel(DiscreteParams p_arg, const void* actions,
                                                          int row_len, int n_steps,
                                                          const int32_t* __restrict__ env_ids, int n_ids) {
  extern __shared__ __align__(16) unsigned char smem[];
#if PCT_KERNARG_PTR
  // The parameter block is read where it is used, straight from the kernarg segment (scalar loads that hit the scalar
  // cache), instead of through the by-value argument, which the backend loads whole in the entry block and then has to
  // keep alive -- or spill to VGPR lanes -- for the length of the kernel.
  const DiscreteParams& p = *(const DiscreteParams*)pct_param_fence((PctConstParams<DiscreteParams>)__builtin_amdgcn_kernarg_segment_ptr());
#else
  const DiscreteParams& p = p_arg;
#endif
  // One body for both passes (two inlined copies of the transition doubled the kernel's code).  Normal pass: workgroup b
  // steps env b (or order[b], or env_ids[b] for a partial reset), once.  Large-capacity retry pass: a small grid strides
  // over the envs the normal pass queued (usually none).  p.retry_count points at this step's counter of a ping-pong
  // pair; the other one (used by the next step's normal pass, which cannot start before this kernel ends) is zeroed
  // here -- no memset between launches.
  const int rm = p.retry_mode;  // +1 / -1: offset of the other counter
  int limit = 0;
  if (rm) {
    limit = *p.retry_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      p.retry_count[rm] = 0;
      if (limit > 0 && p.retry_total) { p.retry_total[0] += limit; p.retry_total[1] += 1; }
    }
    // (a handle whose plain steps run the retry pass as the tail of their own launch: its sub-counters of the other step, too)
    if (blockIdx.x == 0 && p.tail_done) p.tail_done[p.tail_rm + (int)threadIdx.x] = 0;
  }
  int w = blockIdx.x;
  while (true) {
    int e = w;
    if (rm) {
      if (w >= limit) break;
      e = __builtin_amdgcn_readfirstlane(p.retry_ids[w]);
    } else if (ACT == ACT_RESET && env_ids) {
      if (e >= n_ids) break;
      e = __builtin_amdgcn_readfirstlane(env_ids[e]);
      if (e < 0 || e >= p.N) break;
    } else if (ACT != ACT_RESET && p.order) {
      // heavy-first dispatch: a permutation of 0..N-1.  (readfirstlane: the compiler would otherwise keep the loaded id,
      // and every address derived from it, in vector registers -- 34 more spilled VGPRs in the setting-2 kernel)
      e = __builtin_amdgcn_readfirstlane(p.order[e]);
    }
    work_key_begin(smem);
    int n_ems = 0;
    discrete_env_steps<K, BITS, ACT, TIMED, STAB, SCHEME, RNG>(p, actions, row_len, n_steps, e, smem, n_ems);
    // (a large-capacity run is what this env's next step will look like: its key replaces the normal pass's)
    work_key_end(smem, p.scalars, p.N, e, !rm && ACT == ACT_RESET, n_ems);
    if (!rm) break;
    __syncthreads();
    w += gridDim.x;
  }
}


// ---- one dispatch per step: the retry pass as the TAIL of the normal pass's launch (round 6) -----------------------------------
// The plain setting-2 transition (LNES = EMS, counter-keyed draws, no shuffle): workgroups 0 .. N - 1 are the normal pass (as
// pct_discrete_kernel), workgroups N .. N + R - 1 the retry pass -- dispatched last, they wait until the N others have left (64
// sub-counters, one lane polls each), then stride over the queue with the retry pass's capacities (`q`: the parameter block the
// separate retry dispatch would get), their "LDS" lists in a per-workgroup row of HBM.  The transition is inlined a second
// time for that (generic address space): cold code that an ordinary step never enters.  What it buys: the 16-block retry kernel
// that follows every transition otherwise -- idle, but a dependent dispatch -- was 5.5 us of C2's 61.8 us step.
template <typename K, int BITS, int ACT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
pct_discrete_tail_kernel(DiscreteParams p_arg, const void* actions, int row_len, int n_steps, DiscreteParams q_arg) {
  extern __shared__ __align__(16) unsigned char smem[];
  const char __attribute__((address_space(4)))* const ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  const DiscreteParams& p = *(const DiscreteParams*)pct_param_fence((PctConstParams<DiscreteParams>)ka);
  constexpr size_t q_off = (sizeof(DiscreteParams) + sizeof(void*) + 2 * sizeof(int) + 7) & ~(size_t)7;
  (void)p_arg; (void)q_arg;
  if (__builtin_expect((int)blockIdx.x >= p.N, 0)) {
    // ---- the retry pass -------------------------------------------------------------------------------------------------
    const DiscreteParams& q = *(const DiscreteParams*)pct_param_fence((PctConstParams<DiscreteParams>)(ka + q_off));
    const int tb = (int)blockIdx.x - p.N, R = (int)gridDim.x - p.N;
    const int lane = threadIdx.x;
    while (true) {  // every normal-pass workgroup has added one to its sub-counter when it left
      int c = __hip_atomic_load(&p.tail_done[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
      if (c >= p.N) break;
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int limit = __hip_atomic_load(p.retry_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tb == 0) {  // the OTHER step's counters (the next launch's, which cannot start before this kernel ends)
      p.tail_done[p.tail_rm + lane] = 0;
      if (lane == 0) {
        p.retry_count[p.tail_rm > 0 ? 1 : -1] = 0;
        if (limit > 0 && p.retry_total) { p.retry_total[0] += limit; p.retry_total[1] += 1; }
      }
    }
    unsigned char* const base = p.tail_scratch + (size_t)tb * (size_t)p.tail_scratch_bytes;
    for (int w = tb; w < limit; w += R) {
      const int e = __builtin_amdgcn_readfirstlane(p.retry_ids[w]);
      work_key_begin(smem);
      int n_ems = 0;
      discrete_env_steps<K, BITS, ACT, false, false, 0, 0>(q, actions, row_len, n_steps, e, base, n_ems);
      work_key_end(smem, p.scalars, p.N, e, false, n_ems);
      __syncthreads();
    }
    return;
  }
  // ---- the normal pass --------------------------------------------------------------------------------------------------
  int e = blockIdx.x;
  if (p.order) e = __builtin_amdgcn_readfirstlane(p.order[e]);
  work_key_begin(smem);
  int n_ems = 0;
  discrete_env_steps<K, BITS, ACT, false, false, 0, 0>(p, actions, row_len, n_steps, e, smem, n_ems);
  work_key_end(smem, p.scalars, p.N, e, false, n_ems);
  // (an env this workgroup queued: retry_enqueue has released its queue entry at agent scope; a workgroup that queued nothing has
  // nothing to publish -- what a queued env had already written, reward / done / counter, the tail writes again, identically)
  if (threadIdx.x == 0) __hip_atomic_fetch_add(&p.tail_done[blockIdx.x & 63], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace pct

// ----------------------------------------------------------------------------------------
// launchers (called from pct_env.hip)
// ----------------------------------------------------------------------------------------
namespace pct {

inline size_t discrete_lds_bytes_impl(const DiscreteParams& p) {
  size_t b = PCT_LDS_STASH + discrete_lds_base_bytes(p);
  if (p.setting != 2) b += ((stab_state_bytes(p.I, p.sb.caps) + 15) & ~(size_t)15) + stab_wave_bytes(p.sb.caps);
  return b;
}

// STAB selects the half of the instantiations a translation unit carries: the setting-2 kernels (false) or the
// stability-check kernels of settings 1 / 3 (true); MTSEL the random source: the counter-keyed kernels (false) or the
// strict NumPy-stream kernels (true, per-env MT19937: pct_set_numpy_rng) -- eight translation units (u32 / u64 keys x
// plain / stability x counter / NumPy stream) compile in parallel
template <typename K, int BITS, bool STAB, bool MTSEL = false>
inline hipError_t launch_typed(const DiscreteParams& p, int act, const void* actions, int row_len, int n_steps,
                               const int32_t* env_ids, int n_ids, hipStream_t stream) {
  size_t lds = discrete_lds_bytes_impl(p);
  const bool timed = p.timing != nullptr && act != ACT_RESET;
  const int scheme = p.lnes == PCT_LNES_EMS ? 0 : 1;  // 1: every other expansion, dispatched inside the kernel
  int grid = p.retry_mode ? n_ids : ((act == ACT_RESET && env_ids) ? n_ids : p.N);
  if (grid <= 0) return hipSuccess;
  if (MTSEL != (p.rng_numpy != 0)) return hipErrorInvalidValue;  // (launch_discrete picks the translation unit)
  // RNG mode: bit 0 shuffle, bit 1 strict NumPy stream (no timed build, no heuristic kernels: checked by the host)
  constexpr int R_SHUF = MTSEL ? 3 : 1, R_PLAIN = MTSEL ? 2 : 0;
#ifdef PCT_FEW_KERNELS
  // kernel experiments (scripts/build_variant.py): only the untimed LNES = EMS kernels without shuffle -- a stability translation unit
  // then compiles in a minute instead of five; everything else is refused
#ifdef PCT_FEW_TIMED
  if (p.shuffle || scheme == 1 || act == ACT_HEUR || act == ACT_INDEX) return hipErrorNotSupported;
#define PCT_KERN(A, T, S, C) pct_discrete_kernel<K, BITS, A, T, S, 0, R_PLAIN>
#else
  if (p.shuffle || scheme == 1 || timed || act == ACT_HEUR || act == ACT_INDEX) return hipErrorNotSupported;
#define PCT_KERN(A, T, S, C) pct_discrete_kernel<K, BITS, A, false, S, 0, R_PLAIN>
#endif
#else
#define PCT_KERN(A, T, S, C) (p.shuffle ? pct_discrete_kernel<K, BITS, A, false, S, C, R_SHUF> : pct_discrete_kernel<K, BITS, A, (MTSEL ? false : T), S, C, R_PLAIN>)
#endif
#define PCT_LAUNCH(A)                                                                                        \
  do {                                                                                                       \
    void (*kern)(DiscreteParams, const void*, int, int, const int32_t*, int);                                \
    if (scheme == 1) kern = PCT_KERN(A, false, STAB, 1);                                                     \
    else kern = timed ? PCT_KERN(A, true, STAB, 0) : PCT_KERN(A, false, STAB, 0);                            \
    if (lds > 48 * 1024) {                                                                                   \
      hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                               \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
      if (er != hipSuccess) return er;                                                                       \
    }                                                                                                        \
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, stream, (hipEvent_t)p.launch_ev_start,           \
                          (hipEvent_t)p.launch_ev_stop, 0, p, actions, row_len, n_steps, env_ids, n_ids); \
  } while (0)
  if (act == ACT_HEUR) {  // heuristic policies read the EMS list: LNES == EMS only (checked by the caller)
#ifdef PCT_FEW_KERNELS
    if constexpr (true) {
#else
    if constexpr (MTSEL) {
#endif
      return hipErrorInvalidValue;
    } else {
    void (*kern)(DiscreteParams, const void*, int, int, const int32_t*, int);
    kern = p.shuffle ? pct_discrete_kernel<K, BITS, ACT_HEUR, false, STAB, 0, 1>
                     : pct_discrete_kernel<K, BITS, ACT_HEUR, false, STAB, 0, 0>;
    if (lds > 48 * 1024) {
      hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (er != hipSuccess) return er;
    }
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, stream, (hipEvent_t)p.launch_ev_start, (hipEvent_t)p.launch_ev_stop,
                          0, p, actions, row_len, n_steps, env_ids, n_ids);
    return hipGetLastError();
    }
  }
  if constexpr (!STAB && !MTSEL) {
    if (discrete_tail_eligible(p, act, env_ids) && p.tail_q) {
      // one dispatch: the normal pass + the retry pass as its tail (pct_discrete_tail_kernel); q = the retry pass's parameter block
      void (*kern)(DiscreteParams, const void*, int, int, DiscreteParams) = nullptr;
      switch (act) {
        case ACT_ROWS: kern = pct_discrete_tail_kernel<K, BITS, ACT_ROWS>; break;
        case ACT_INDEX: kern = pct_discrete_tail_kernel<K, BITS, ACT_INDEX>; break;
        default: kern = pct_discrete_tail_kernel<K, BITS, ACT_HASH>; break;
      }
      hipExtLaunchKernelGGL(kern, dim3(p.N + p.tail_blocks), dim3(64), lds, stream, (hipEvent_t)p.launch_ev_start,
                            (hipEvent_t)p.launch_ev_stop, 0, p, actions, row_len, n_steps, *p.tail_q);
      return hipGetLastError();
    }
  }
  switch (act) {
    case ACT_ROWS: PCT_LAUNCH(ACT_ROWS); break;
#ifndef PCT_FEW_KERNELS
    case ACT_INDEX: PCT_LAUNCH(ACT_INDEX); break;
#endif
    case ACT_HASH: PCT_LAUNCH(ACT_HASH); break;
    default: PCT_LAUNCH(ACT_RESET); break;
  }
#undef PCT_LAUNCH
  return hipGetLastError();
}


}  // namespace pct
#endif
