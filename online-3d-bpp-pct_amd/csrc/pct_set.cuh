// pct_set.cuh -- wave-level helpers and the exact, wave-parallel CPython `set` emulation shared
// by the discrete and continuous transition kernels (gfx950, 64-lane wavefronts).
#ifndef PCT_SET_CUH
#define PCT_SET_CUH
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pct {

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    int o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, 64);
    uint64_t o = ((uint64_t)hi << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)(uint64_t)v, off, 64);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)((uint64_t)v >> 32), off, 64);
    v += (long long)(((uint64_t)hi << 32) | lo);
  }
  return v;
}
// number of set bits of a wave-uniform mask below this lane (the rank of the lane among the set
// lanes): v_mbcnt_lo/hi, no per-lane mask register needed
__device__ __forceinline__ int rank_below(uint64_t m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int src) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src);
  uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
template <typename K>
__device__ __forceinline__ K bcast_key(K v, int src);
template <>
__device__ __forceinline__ uint32_t bcast_key<uint32_t>(uint32_t v, int src) {
  return __builtin_amdgcn_readlane(v, src);
}
template <>
__device__ __forceinline__ uint64_t bcast_key<uint64_t>(uint64_t v, int src) {
  return bcast_u64(v, src);
}
template <typename K>
__device__ __forceinline__ K uniform_key(K v);
template <>
__device__ __forceinline__ uint32_t uniform_key<uint32_t>(uint32_t v) {
  return __builtin_amdgcn_readfirstlane(v);
}
template <>
__device__ __forceinline__ uint64_t uniform_key<uint64_t>(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}


// CPython tuple hash (Objects/tupleobject.c, xxHash-derived, CPython >= 3.8): acc over the
// element hashes ("lanes").
#define PCT_XXPRIME_1 11400714785074694791ULL
#define PCT_XXPRIME_2 14029467366897019727ULL
#define PCT_XXPRIME_5 2870177450012600261ULL
__device__ __forceinline__ uint64_t tuplehash_begin() { return PCT_XXPRIME_5; }
__device__ __forceinline__ uint64_t tuplehash_lane(uint64_t acc, uint64_t lane) {
  acc += lane * PCT_XXPRIME_2;
  acc = (acc << 31) | (acc >> 33);
  acc *= PCT_XXPRIME_1;
  return acc;
}
__device__ __forceinline__ uint64_t tuplehash_end(uint64_t acc, uint64_t len) {
  acc += len ^ (PCT_XXPRIME_5 ^ 3527539ULL);
  if (acc == (uint64_t)-1) return 1546275796ULL;
  return acc;
}
__device__ __forceinline__ uint64_t tuplehash_end6(uint64_t acc) { return tuplehash_end(acc, 6ULL); }
// hash(int) as the tuple hash sees it: the value itself (two's complement), except hash(-1) == -2
__device__ __forceinline__ uint64_t pyhash_int(int v) { return v == -1 ? (uint64_t)(int64_t)-2 : (uint64_t)(int64_t)v; }

// ----------------------------------------------------------------------------------------
// CPython set emulation, wave-parallel and exact.
//
// Slot words: EMPTY (all ones) | a key (tag bit clear) | TAG|lane (tentative holder, only
// while a batch is being matched).  Sequential set.add of keys k0,k1,.. (each takes the first
// free slot on its own probe path: LINEAR_PROBES 9, PERTURB_SHIFT 5) is a serial
// dictatorship; because every slot ranks keys by the same priority (insertion order) its
// outcome is the unique stable matching, which the lanes reach in parallel by proposing
// along their paths with atomicMin on TAG|lane: a lower lane evicts a higher one, the
// evicted lane walks on.  A slot a lane has walked past stays held by a higher-priority key
// for ever, so the fixed point equals the sequential result, slot for slot.
// ----------------------------------------------------------------------------------------
template <typename K>
struct SlotWord;
template <>
struct SlotWord<uint32_t> {
  static constexpr uint32_t EMPTY = 0xFFFFFFFFu, TAG = 0x80000000u;
};
template <>
struct SlotWord<uint64_t> {
  static constexpr uint64_t EMPTY = ~0ull, TAG = 1ull << 63;
};
__device__ __forceinline__ uint32_t lds_atomic_min(uint32_t* p, uint32_t v) { return atomicMin(p, v); }
__device__ __forceinline__ uint64_t lds_atomic_min(uint64_t* p, uint64_t v) {
  return (uint64_t)atomicMin(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
}

#ifndef PCT_MATCH_MASKS
#define PCT_MATCH_MASKS 1 /* 1: the one-key-per-lane matching walk keeps its lane sets as scalar masks (pyset_match_v) */
#endif
#define PCT_PEND_SLOTS 128
// Table accessors.  GT == false: the table is in LDS (plain accesses).  GT == true: the table
// is a per-env slice of HBM (capacities that do not fit in LDS): every access goes to L2
// (agent-scope relaxed atomics) because a vector L1 line is not refreshed by this wave's own
// atomics.
template <bool GT, typename K>
__device__ __forceinline__ K tab_ld(const K* p) {
  if (GT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool GT, typename K>
__device__ __forceinline__ void tab_st(K* p, K v) {
  if (GT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

struct Walk {  // position on a key's probe path: slot = i + j
  uint32_t i;
  int j;
  uint64_t perturb;
  __device__ __forceinline__ void start(uint64_t hash, uint32_t mask) {
    perturb = hash;
    i = (uint32_t)hash & mask;
    j = 0;
  }
  __device__ __forceinline__ void next(uint32_t mask) {
    int probes = (i + 9u <= mask) ? 9 : 0;
    if (j < probes) {
      j++;
    } else {
      perturb >>= 5;
      i = (i * 5u + 1u + (uint32_t)perturb) & mask;
      j = 0;
    }
  }
};

// One step along a probe path, without branches: the next linear probe, or the next perturbed jump
__device__ __forceinline__ void walk_advance(uint32_t& i, int& j, uint64_t& perturb, uint32_t mask) {
  const bool lin = j < ((i + 9u <= mask) ? 9 : 0);
  const uint64_t pn = perturb >> 5;
  const uint32_t in = (i * 5u + 1u + (uint32_t)pn) & mask;
  j = lin ? j + 1 : 0;
  i = lin ? i : in;
  perturb = lin ? perturb : pn;
}

// Read-only membership test (no tags may be present).
// `same(word)` decides whether the table entry `word` holds the caller's key (set_add_entry:
// entry->hash == hash and the keys compare equal).  Wave-uniform loop, predicated body: the whole wave
// runs as many iterations as its longest walk and an iteration is a handful of selects around one
// LDS read.  All 64 lanes must call; a lane without a key passes active = false.
template <typename K, bool GT = false, typename Same>
__device__ __forceinline__ bool pyset_contains(const K* tab, uint32_t mask, uint64_t hash, bool active, Same same,
                                      int* probes = nullptr) {
  uint32_t i = (uint32_t)hash & mask;
  int j = 0;
  uint64_t perturb = hash;
  bool found = false;
  while (__ballot(active)) {
    K cur = SlotWord<K>::EMPTY;
    if (active) cur = tab_ld<GT, K>(&tab[i + (uint32_t)j]);
    if (probes) (*probes)++;
    const bool eq = active && cur != SlotWord<K>::EMPTY && same(cur);
    found = found || eq;
    active = active && cur != SlotWord<K>::EMPTY && !eq;
    walk_advance(i, j, perturb, mask);
  }
  return found;
}

// Matches the participating lanes' keys into the table in lane-priority order.  The
// participating keys must be pairwise distinct (the caller de-duplicates a batch first: two
// equal keys can overtake one another on their common path, distinct keys cannot matter to
// each other except through the slots they hold).  On return a lane with `placed` holds
// TAG|lane in tab[slot]; a participating lane that is not placed found its key already in the
// table (check_found).  All 64 lanes must call.
template <typename K, bool GT = false, typename Same>
__device__ __forceinline__ void pyset_match(K* tab, uint32_t mask, bool part, uint64_t hash, int lane, bool check_found,
                                   bool& placed, uint32_t& slot, Same same, int* stats = nullptr) {
  const K TAG = SlotWord<K>::TAG;
  const K mytag = TAG | (K)lane;
  placed = false;
  slot = 0;
  bool walking = part;
  int my_probes = 0;
  if (GT) {
    Walk w;
    w.start(hash, mask);
    while (true) {
      if (stats) stats[1]++;
      while (walking) {
        my_probes++;
        uint32_t cur = w.i + w.j;
        K v = tab_ld<GT, K>(&tab[cur]);
        if ((v & TAG) && v > mytag) {  // empty, or tentatively held by a later lane
          K old = lds_atomic_min(&tab[cur], mytag);
          if (old > mytag) {
            slot = cur;
            placed = true;
            walking = false;
          }
        } else if (check_found && same(v)) {
          walking = false;  // already a member
        } else {
          w.next(mask);  // a different key, or an earlier lane's tentative hold
        }
      }
      __syncthreads();
      if (placed && tab_ld<GT, K>(&tab[slot]) != mytag) {  // evicted by an earlier lane: walk on
        placed = false;
        walking = true;
        w.next(mask);
      }
      if (!__ballot(walking)) break;
    }
  } else {
    // LDS: propose straight away -- a real key (tag bit clear) or an earlier lane's tag is numerically
    // smaller than mytag and stays, so the atomic doubles as the read.  Wave-uniform loops with a
    // predicated body (an iteration = one LDS atomic and a handful of selects); a lane's position always
    // moves past the slot it has just tried, so an evicted lane simply resumes.
#if PCT_MATCH_MASKS
    {
      // the lane sets as scalar masks, the next position computed in the atomic's shadow: see pyset_match_v
      const K EMPTYW = SlotWord<K>::EMPTY;
      uint32_t ci = (uint32_t)hash & mask;
      int cj = 0;
      uint64_t cp = hash;
      uint32_t pos = ci;
      uint64_t wm = __ballot(part), pm = 0;
      while (true) {
        if (stats) stats[1]++;
        while (wm) {
          const bool w = __builtin_amdgcn_inverse_ballot_w64(wm);
          const uint32_t tried = pos;
          K old = lds_atomic_min(&tab[pos], w ? mytag : EMPTYW);
          {
            uint32_t ni = ci;
            int nj = cj;
            uint64_t np = cp;
            walk_advance(ni, nj, np, mask);
            const uint32_t npos = ni + (uint32_t)nj;
            ci = w ? ni : ci;
            cj = w ? nj : cj;
            cp = w ? np : cp;
            pos = w ? npos : pos;
            uint32_t plo = (uint32_t)cp, phi = (uint32_t)(cp >> 32);
            if (sizeof(K) == 4) {
              uint32_t o32 = (uint32_t)old;
              asm volatile("" : "+v"(o32), "+v"(ci), "+v"(cj), "+v"(plo), "+v"(phi), "+v"(pos));
              old = (K)o32;
            } else {
              uint32_t olo = (uint32_t)old, ohi = (uint32_t)((uint64_t)old >> 32);
              asm volatile("" : "+v"(olo), "+v"(ohi), "+v"(ci), "+v"(cj), "+v"(plo), "+v"(phi), "+v"(pos));
              old = (K)(((uint64_t)ohi << 32) | olo);
            }
            cp = ((uint64_t)phi << 32) | plo;
          }
          my_probes++;
          const uint64_t won = __ballot(old > mytag) & wm;  // was empty, or tentatively held by a later lane
          uint64_t stop = won;
          if (check_found) {
            const bool cand = w && !(old & TAG);  // a real key: is it this lane's own?
            if (__ballot(cand)) stop |= __ballot(cand && same(old));
          }
          slot = __builtin_amdgcn_inverse_ballot_w64(won) ? tried : slot;
          pm |= won;
          wm &= ~stop;
        }
        __syncthreads();
        const bool held = __builtin_amdgcn_inverse_ballot_w64(pm);
        const K now = tab[slot];
        wm = __ballot(held && now != mytag);  // evicted by an earlier lane: walk on from the next slot
        pm &= ~wm;
        if (!wm) break;
      }
      placed = __builtin_amdgcn_inverse_ballot_w64(pm);
      if (stats) { stats[0]++; stats[2] += wave_max_i32(my_probes); }
      return;
    }
#endif
    uint32_t i = (uint32_t)hash & mask;
    int j = 0;
    uint64_t perturb = hash;
    while (true) {
      if (stats) stats[1]++;
      while (__ballot(walking)) {
        const uint32_t cur = i + (uint32_t)j;
        // every lane issues the atomic: one that is not walking proposes the all-ones word (EMPTY), which changes
        // no slot, and stays where it is -- no exec-mask juggling around the LDS operation
        const K old = lds_atomic_min(&tab[cur], walking ? mytag : SlotWord<K>::EMPTY);
        my_probes++;
        const bool won = walking & (old > mytag);  // was empty, or tentatively held by a later lane
        const bool member = walking && check_found && !(old & TAG) && same(old);
        slot = won ? cur : slot;
        placed = placed | won;
        uint32_t ni = i;
        int nj = j;
        uint64_t np = perturb;
        walk_advance(ni, nj, np, mask);
        i = walking ? ni : i;
        j = walking ? nj : j;
        perturb = walking ? np : perturb;
        walking = walking & !won & !member;
      }
      __syncthreads();
      if (placed && tab[slot] != mytag) {  // evicted by an earlier lane: walk on from the next slot
        placed = false;
        walking = true;
      }
      if (!__ballot(walking)) break;
    }
  }
  if (stats) { stats[0]++; stats[2] += wave_max_i32(my_probes); }
}

// Exact removal of duplicates inside a batch of <= 64 keys (lane = position in insertion order):
// returns true for a lane whose key also sits in an EARLIER active lane.  Lanes scatter their
// lane id into a bucket chosen by hash bits (LDS atomicMin); the minimum lane of a bucket is the
// first holder of every key hashing there, so it is unique; any other lane compares with that
// minimum lane (`same_as(w)`): equal -> duplicate, different -> the two merely collided and both
// the later lane and all its potential duplicates (same hash, hence same bucket in every round)
// stay for the next round, which uses other hash bits.  `dd` = NB words of LDS, all ones.
template <int NB, typename SameAs>
__device__ __forceinline__ bool batch_find_duplicates(uint32_t* dd, bool active, uint64_t hash, int lane, int cnt,
                                             SameAs same_as) {
  bool unresolved = active, dup = false;
  for (int round = 0; round < 8; round++) {
    if (!__ballot(unresolved)) return dup;
    const uint32_t b = (uint32_t)(hash >> (3 + 7 * round)) & (uint32_t)(NB - 1);
    if (unresolved) atomicMin(&dd[b], (uint32_t)lane);
    __syncthreads();
    const uint32_t w = unresolved ? dd[b] : (uint32_t)lane;
    __syncthreads();
    if (unresolved) {
      dd[b] = 0xFFFFFFFFu;
      if (w == (uint32_t)lane) {
        unresolved = false;
      } else if (same_as((int)w)) {
        dup = true;
        unresolved = false;
      }
    }
    __syncthreads();
  }
  // eight rounds of pure collisions between distinct keys: fall back to the exhaustive scan
  for (int i = 0; i < cnt; i++)
    if (unresolved && i < lane && same_as(i)) dup = true;
  return dup;
}

// Same as batch_find_duplicates, but the earlier lane's hash and payload (e.g. a generator id the key can be
// rebuilt from) come over cross-lane shuffles instead of an LDS copy of the batch: `same(hash_w, payload_w)`.
template <int NB, typename Same>
__device__ __forceinline__ bool batch_find_duplicates_shfl(uint32_t* dd, bool active, uint64_t hash, uint32_t payload, int lane,
                                                  Same same) {
  bool unresolved = active, dup = false;
  for (int round = 0; round < 8; round++) {
    if (!__ballot(unresolved)) return dup;
    const uint32_t b = (uint32_t)(hash >> (3 + 7 * round)) & (uint32_t)(NB - 1);
    if (unresolved) atomicMin(&dd[b], (uint32_t)lane);
    __syncthreads();
    const uint32_t w = unresolved ? dd[b] : (uint32_t)lane;
    __syncthreads();
    const uint32_t hlo = (uint32_t)__shfl((int)(uint32_t)hash, (int)(w & 63u), 64);
    const uint32_t hhi = (uint32_t)__shfl((int)(uint32_t)(hash >> 32), (int)(w & 63u), 64);
    const uint32_t pw = (uint32_t)__shfl((int)payload, (int)(w & 63u), 64);
    if (unresolved) {
      dd[b] = 0xFFFFFFFFu;
      if (w == (uint32_t)lane) {
        unresolved = false;
      } else if (same(((uint64_t)hhi << 32) | hlo, pw)) {
        dup = true;
        unresolved = false;
      }
    }
    __syncthreads();
  }
  // eight rounds of pure collisions between distinct keys: exhaustive scan over the earlier lanes
  const uint64_t am = __ballot(active);
  for (int i = 0; i < 64; i++) {
    const uint32_t hlo = (uint32_t)__shfl((int)(uint32_t)hash, i, 64);
    const uint32_t hhi = (uint32_t)__shfl((int)(uint32_t)(hash >> 32), i, 64);
    const uint32_t pw = (uint32_t)__shfl((int)payload, i, 64);
    if (unresolved && ((am >> i) & 1ull) && i < lane && same(((uint64_t)hhi << 32) | hlo, pw)) dup = true;
  }
  return dup;
}

// Same again for keys that are six doubles held in registers (the continuous env's candidate tuples): the first holder's
// tuple itself comes over twelve cross-lane reads and is compared value for value -- no hash comparison, and no recomputation
// of the other lane's tuple from its generator id (float64 lattice conversions and the rotation switch: ~120 instructions that
// every lane of the wave paid in every round that held a duplicate).
template <int NB>
__device__ __forceinline__ bool batch_find_duplicates_t6(uint32_t* dd, bool active, uint64_t hash, const double (&t)[6], int lane) {
  bool unresolved = active, dup = false;
  for (int round = 0; round < 8; round++) {
    if (!__ballot(unresolved)) return dup;
    const uint32_t b = (uint32_t)(hash >> (3 + 7 * round)) & (uint32_t)(NB - 1);
    if (unresolved) atomicMin(&dd[b], (uint32_t)lane);
    __syncthreads();
    const uint32_t w = unresolved ? dd[b] : (uint32_t)lane;
    __syncthreads();
    bool eq = true;
#pragma unroll
    for (int c = 0; c < 6; c++) eq = eq && (__shfl(t[c], (int)(w & 63u), 64) == t[c]);
    if (unresolved) {
      dd[b] = 0xFFFFFFFFu;
      if (w == (uint32_t)lane) unresolved = false;
      else if (eq) { dup = true; unresolved = false; }
    }
    __syncthreads();
  }
  // eight rounds of pure collisions between distinct keys: exhaustive scan over the earlier lanes
  const uint64_t am = __ballot(active);
  for (int i = 0; i < 64; i++) {
    bool eq = true;
#pragma unroll
    for (int c = 0; c < 6; c++) eq = eq && (__shfl(t[c], i, 64) == t[c]);
    if (unresolved && ((am >> i) & 1ull) && i < lane && eq) dup = true;
  }
  return dup;
}

template <typename K>
__device__ __forceinline__ K shfl_key(K v, int src);
template <>
__device__ __forceinline__ uint32_t shfl_key<uint32_t>(uint32_t v, int src) {
  return (uint32_t)__shfl((int)v, src, 64);
}
template <>
__device__ __forceinline__ uint64_t shfl_key<uint64_t>(uint64_t v, int src) {
  uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64);
  uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
  return ((uint64_t)hi << 32) | lo;
}

// Same, for keys that fit a register: the minimum lane's key is fetched with a cross-lane
// shuffle instead of an LDS copy of the batch.
template <int NB, typename K>
__device__ __forceinline__ bool batch_find_duplicates_reg(uint32_t* dd, bool active, K key, uint64_t hash, int lane) {
  bool unresolved = active, dup = false;
  for (int round = 0; round < 9; round++) {
    if (!__ballot(unresolved)) return dup;
    const uint32_t b = (uint32_t)(hash >> (3 + 6 * round)) & (uint32_t)(NB - 1);
    if (unresolved) atomicMin(&dd[b], (uint32_t)lane);
    __syncthreads();
    const uint32_t w = unresolved ? dd[b] : (uint32_t)lane;
    __syncthreads();
    const K kw = shfl_key<K>(key, (int)(w & 63u));
    if (unresolved) {
      dd[b] = 0xFFFFFFFFu;
      if (w == (uint32_t)lane) unresolved = false;
      else if (kw == key) { dup = true; unresolved = false; }
    }
    __syncthreads();
  }
  // nine rounds of pure collisions between distinct keys: exhaustive scan
  for (int i = 0; i < 64; i++) {
    const K ki = shfl_key<K>(key, i);
    const bool ai = (__ballot(active) >> i) & 1ull;
    if (unresolved && ai && i < lane && ki == key) dup = true;
  }
  return dup;
}

// ----------------------------------------------------------------------------------------
// V keys per lane ("virtual lanes").  One wave alone on its SIMD waits ~7 cycles for a dependent VALU
// result and ~100 for an LDS round trip; the probe loops are chains of exactly such dependences, so a
// lane that carries V independent keys through the same loop gets V times the work done per iteration
// at little more than the latency of one.  Batch position of (v, lane) is v * 64 + lane: slice 0 comes
// first in insertion order, then slice 1, ...  LDS tables only (integer keys compared by value).
// ----------------------------------------------------------------------------------------
template <int V>
__device__ __forceinline__ bool any_of(const bool (&b)[V]) {
  bool a = false;
#pragma unroll
  for (int v = 0; v < V; v++) a = a || b[v];
  return a;
}

// Read-only membership test of V keys per lane (no tags may be present): found[v] for active[v] keys.
template <int V, typename K>
__device__ __forceinline__ void pyset_contains_v(const K* tab, uint32_t mask, const uint64_t (&hash)[V], const K (&key)[V],
                                        const bool (&valid)[V], bool (&found)[V], int* probes = nullptr) {
  uint32_t i[V];
  int j[V];
  uint64_t perturb[V];
  bool active[V];
#pragma unroll
  for (int v = 0; v < V; v++) {
    i[v] = (uint32_t)hash[v] & mask;
    j[v] = 0;
    perturb[v] = hash[v];
    active[v] = valid[v];
    found[v] = false;
  }
  while (__ballot(any_of<V>(active))) {
    K cur[V];
#pragma unroll
    for (int v = 0; v < V; v++) {
      cur[v] = SlotWord<K>::EMPTY;
      if (active[v]) cur[v] = tab[i[v] + (uint32_t)j[v]];
    }
    if (probes) (*probes)++;
#pragma unroll
    for (int v = 0; v < V; v++) {
      const bool eq = active[v] && cur[v] == key[v];
      found[v] = found[v] || eq;
      active[v] = active[v] && cur[v] != SlotWord<K>::EMPTY && !eq;
      walk_advance(i[v], j[v], perturb[v], mask);
    }
  }
}

// Matches the participating keys (pairwise distinct, none of them in the table yet) into the table in
// batch-position order, exactly as sequential set.add calls would place them.  On return every
// participating (v, lane) holds TAG | (v*64 + lane) in tab[slot[v]]; the caller overwrites the tags
// with the keys.  All 64 lanes must call.
// With `key` (CHECK): a participating key may already be a member of the table -- its walk then ends at its own
// entry and it is not placed (set.add of a member is a no-op); placed[v] tells the two outcomes apart.
// `base`: batch position of (0, lane 0) -- the priority of (v, lane) is base + v * 64 + lane (a whole-set matching of more
// than V * 64 keys runs in passes over one table).
template <int V, typename K, bool CHECK = false>
__device__ __forceinline__ void pyset_match_v(K* tab, uint32_t mask, const bool (&part)[V], const uint64_t (&hash)[V], int lane,
                                     uint32_t (&slot)[V], int* stats = nullptr, const K* key = nullptr,
                                     bool* placed_out = nullptr, uint32_t base = 0) {
  const K TAG = SlotWord<K>::TAG;
#if PCT_MATCH_MASKS
  if constexpr (V == 1) {
    // One key per lane (the shipped configuration): the walk with its lane sets -- walking, won, placed -- kept as 64-bit
    // masks in SCALAR registers (round 4).  A walk step is a chain: LDS atomic -> compare -> who is still walking -> the next
    // atomic; scripts/microbench_walk.hip measures 85 cycles for the returning LDS atomic alone and 270 for a step of the
    // per-lane-predicate formulation below, most of it the compiler's select / re-test sequences around the wave-uniform
    // loop condition.  Here the chain behind the atomic is v_cmp (into an SGPR pair), two s_and / s_andn2, the loop branch
    // and one v_cndmask that takes the mask straight from the SGPRs (inverse ballot); the probe-sequence arithmetic of the
    // NEXT position does not depend on the value returned and runs in the atomic's shadow.  Same proposals, same order,
    // same fixed point as the formulation below.
    const K mytag = TAG | (K)(base + (uint32_t)lane);
    const K EMPTYW = SlotWord<K>::EMPTY;
    uint32_t ci = (uint32_t)hash[0] & mask;
    int cj = 0;
    uint64_t cp = hash[0];
    uint32_t pos = ci;
    uint32_t myslot = 0;
    uint64_t wm = __ballot(part[0]);  // lanes that are walking
    uint64_t pm = 0;                  // lanes that hold a slot (tentatively, until the fixed point)
    int my_probes = 0;
    while (true) {
      if (stats) stats[1]++;
      const uint64_t t_loop = stats ? __builtin_readcyclecounter() : 0;  // (timed build only)
      while (wm) {
        const bool w = __builtin_amdgcn_inverse_ballot_w64(wm);
        // every lane issues the atomic: one that is not walking proposes the all-ones word, which changes no slot
        const uint32_t tried = pos;
        K old = lds_atomic_min(&tab[pos], w ? mytag : EMPTYW);
        // past the slot just tried (an evicted key resumes there; a lane that is not walking stays where it is): none of
        // this depends on the value returned -- it is pinned in front of its first use, in the atomic's shadow
        {
          uint32_t ni = ci;
          int nj = cj;
          uint64_t np = cp;
          walk_advance(ni, nj, np, mask);
          const uint32_t npos = ni + (uint32_t)nj;
          ci = w ? ni : ci;
          cj = w ? nj : cj;
          cp = w ? np : cp;
          pos = w ? npos : pos;
          uint32_t plo = (uint32_t)cp, phi = (uint32_t)(cp >> 32);
          if (sizeof(K) == 4) {
            uint32_t o32 = (uint32_t)old;
            asm volatile("" : "+v"(o32), "+v"(ci), "+v"(cj), "+v"(plo), "+v"(phi), "+v"(pos));
            old = (K)o32;
          } else {
            uint32_t olo = (uint32_t)old, ohi = (uint32_t)((uint64_t)old >> 32);
            asm volatile("" : "+v"(olo), "+v"(ohi), "+v"(ci), "+v"(cj), "+v"(plo), "+v"(phi), "+v"(pos));
            old = (K)(((uint64_t)ohi << 32) | olo);
          }
          cp = ((uint64_t)phi << 32) | plo;
        }
        const uint64_t won = __ballot(old > mytag) & wm;  // was empty, or tentatively held by a later position
        uint64_t stop = won;
        if (CHECK) stop |= __ballot(!(old & TAG) && old == key[0]) & wm;  // its own entry: set.add of a member is a no-op
        myslot = __builtin_amdgcn_inverse_ballot_w64(won) ? tried : myslot;
        pm |= won;
        if (stats) my_probes++;
        wm &= ~stop;
      }
      if (stats) stats[3] += (int)(__builtin_readcyclecounter() - t_loop);
      __syncthreads();
      // evicted by an earlier position: walk on
      const bool held = __builtin_amdgcn_inverse_ballot_w64(pm);
      const K now = tab[myslot];
      wm = __ballot(held && now != mytag);
      pm &= ~wm;
      if (!wm) break;
    }
    if (stats) { stats[0]++; stats[2] += wave_max_i32(my_probes); }
    slot[0] = myslot;
    if (CHECK) placed_out[0] = __builtin_amdgcn_inverse_ballot_w64(pm);
    return;
  }
#endif
  uint32_t i[V];
  int j[V];
  uint64_t perturb[V];
  bool walking[V], placed[V];
#define PCT_MYTAG(v) (TAG | (K)(base + (uint32_t)((v) * 64 + lane)))
#pragma unroll
  for (int v = 0; v < V; v++) {
    i[v] = (uint32_t)hash[v] & mask;
    j[v] = 0;
    perturb[v] = hash[v];
    walking[v] = part[v];
    placed[v] = false;
    slot[v] = 0;
  }
  int my_probes = 0;
  while (true) {
    if (stats) stats[1]++;
    const uint64_t t_loop = stats ? __builtin_readcyclecounter() : 0;  // (timed build only)
    while (__ballot(any_of<V>(walking))) {
      K old[V];
      uint32_t cur[V];
#pragma unroll
      for (int v = 0; v < V; v++) {  // V atomics in flight: a real key or an earlier position's tag is
        cur[v] = i[v] + (uint32_t)j[v];  // numerically smaller and stays, so the atomic doubles as the read.
        // Straight-line body: EVERY lane issues the atomic -- one that is not walking proposes the all-ones word
        // (EMPTY), which changes no slot -- so the step needs no exec-mask juggling around the LDS operation
        old[v] = lds_atomic_min(&tab[cur[v]], walking[v] ? PCT_MYTAG(v) : SlotWord<K>::EMPTY);
      }
      my_probes++;
#pragma unroll
      for (int v = 0; v < V; v++) {
        const bool won = walking[v] & (old[v] > PCT_MYTAG(v));  // was empty, or tentatively held by a later position
        const bool member = CHECK && (walking[v] & !(old[v] & TAG) & (old[v] == key[v]));
        slot[v] = won ? cur[v] : slot[v];
        placed[v] = placed[v] | won;
        // past the slot just tried: an evicted key resumes here.  A lane that has stopped walking is not moved (its
        // position must stay inside the table for the no-op proposals above; its placed slot is kept in slot[v])
        uint32_t ni = i[v];
        int nj = j[v];
        uint64_t np = perturb[v];
        walk_advance(ni, nj, np, mask);
        i[v] = walking[v] ? ni : i[v];
        j[v] = walking[v] ? nj : j[v];
        perturb[v] = walking[v] ? np : perturb[v];
        walking[v] = walking[v] & !won & !member;
      }
    }
    if (stats) stats[3] += (int)(__builtin_readcyclecounter() - t_loop);
    __syncthreads();
#pragma unroll
    for (int v = 0; v < V; v++) {
      if (placed[v] && tab[slot[v]] != PCT_MYTAG(v)) {  // evicted by an earlier position: walk on
        placed[v] = false;
        walking[v] = true;
      }
    }
    if (!__ballot(any_of<V>(walking))) break;
  }
  if (stats) { stats[0]++; stats[2] += wave_max_i32(my_probes); }
  if (CHECK) {
#pragma unroll
    for (int v = 0; v < V; v++) placed_out[v] = placed[v];
  }
#undef PCT_MYTAG
}

// Exact removal of duplicates inside a batch of V*64 keys held V per lane: dup[v] for a key that also sits
// at an EARLIER active batch position.  Same bucket scatter as batch_find_duplicates_reg (the minimum
// position of a bucket is the first holder of every key hashing there; others compare with it: equal ->
// duplicate, different -> both stay for the next round, which uses other hash bits); the key of a position
// comes over a cross-lane shuffle of the slice it lives in.  `dd` = NB words of LDS, all ones.
template <int V, int NB, typename K>
__device__ __forceinline__ void batch_find_duplicates_v(uint32_t* dd, const bool (&active)[V], const K (&key)[V],
                                               const uint64_t (&hash)[V], int lane, bool (&dup)[V]) {
  bool unresolved[V];
#pragma unroll
  for (int v = 0; v < V; v++) {
    unresolved[v] = active[v];
    dup[v] = false;
  }
  for (int round = 0; round < 8; round++) {
    if (!__ballot(any_of<V>(unresolved))) return;
    uint32_t b[V], w[V];
#pragma unroll
    for (int v = 0; v < V; v++) {
      b[v] = (uint32_t)(hash[v] >> (3 + 7 * round)) & (uint32_t)(NB - 1);
      if (unresolved[v]) atomicMin(&dd[b[v]], (uint32_t)(v * 64 + lane));
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < V; v++) w[v] = unresolved[v] ? dd[b[v]] : (uint32_t)(v * 64 + lane);
    __syncthreads();
#pragma unroll
    for (int v = 0; v < V; v++) {
      K kw = shfl_key<K>(key[0], (int)(w[v] & 63u));
#pragma unroll
      for (int u = 1; u < V; u++) {
        const K ku = shfl_key<K>(key[u], (int)(w[v] & 63u));
        kw = ((w[v] >> 6) == (uint32_t)u) ? ku : kw;
      }
      if (unresolved[v]) {
        dd[b[v]] = 0xFFFFFFFFu;
        if (w[v] == (uint32_t)(v * 64 + lane)) unresolved[v] = false;
        else if (kw == key[v]) { dup[v] = true; unresolved[v] = false; }
      }
    }
    __syncthreads();
  }
  // eight rounds of pure collisions between distinct keys: exhaustive scan over the earlier positions
#pragma unroll
  for (int u = 0; u < V; u++) {
    const uint64_t au = __ballot(active[u]);
    for (int i = 0; i < 64; i++) {
      const K ki = shfl_key<K>(key[u], i);
      const bool ai = (au >> i) & 1ull;
#pragma unroll
      for (int v = 0; v < V; v++)
        if (unresolved[v] && ai && (u * 64 + i) < (v * 64 + lane) && ki == key[v]) dup[v] = true;
    }
  }
}

// Tables of a set with capacity `cap` in ONE LDS region: every size up to 2048 slots starts at
// offset 0 (a rebuild first lifts the <= 512 old slots into registers, then reuses the space);
// only a final table larger than 2048 slots lives behind the quarter-size region it grows from.
__device__ __forceinline__ uint32_t table_offset_compact(uint32_t cap, uint32_t size) {
  return (cap > 2048u && size == cap) ? cap / 4u : 0u;
}
__device__ __host__ inline uint32_t table_words_compact(uint32_t cap) { return cap > 2048u ? cap + cap / 4u : cap; }

// Which LDS region holds a table of `size` slots (ping-pong so that a resize can stream
// old -> new without a temporary): cap in region 0, cap/4 in region 1, cap/16 in 0, ...
// Returned as a slot OFFSET from tab0 (tab1 follows tab0 in LDS) so that every table access
// stays a plain LDS access off one base pointer.
__device__ __forceinline__ uint32_t table_region(uint32_t cap, uint32_t size) {
  int lv = 0;
  while (size < cap) {
    size <<= 2;
    lv++;
  }
  return (lv & 1) ? cap : 0u;
}

#define PCT_TIMING_SLOTS 44
// optional per-phase cycle accounting (pct_debug_phase_timing): s_memtime deltas per env.
// The untimed specialisation is empty, so production kernels carry no extra registers.
template <bool ON>
struct PhaseTimer {
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void tick(int) {}
  __device__ __forceinline__ void flush(unsigned long long*, int) {}
  __device__ __forceinline__ void sub_start() {}
  __device__ __forceinline__ void sub_tick(int) {}
  __device__ __forceinline__ void add(int, uint64_t) {}
  __device__ __forceinline__ uint64_t now() { return 0; }
  static constexpr bool on = false;
};
template <>
struct PhaseTimer<true> {
  static constexpr bool on = true;
  uint64_t last;
  uint64_t acc[PCT_TIMING_SLOTS];
  __device__ __forceinline__ void add(int i, uint64_t v) { acc[i] += v; }  // plain statistics (slots 12..)
  __device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }
  __device__ __forceinline__ void start() {
    for (int i = 0; i < PCT_TIMING_SLOTS; i++) acc[i] = 0;
    last = __builtin_readcyclecounter();
  }
  __device__ __forceinline__ void tick(int i) {
    uint64_t now = __builtin_readcyclecounter();
    acc[i] += now - last;
    last = now;
  }
  uint64_t sub_last;
  __device__ __forceinline__ void sub_start() { sub_last = __builtin_readcyclecounter(); }
  __device__ __forceinline__ void sub_tick(int i) {
    uint64_t now = __builtin_readcyclecounter();
    acc[i] += now - sub_last;
    sub_last = now;
  }
  __device__ __forceinline__ void flush(unsigned long long* o, int n_steps) {
    for (int i = 0; i < PCT_TIMING_SLOTS; i++)
      if (i != 7) o[i] += acc[i];
    o[7] += (unsigned long long)n_steps;
  }
};
enum { PH_LOAD = 0, PH_DROP = 1, PH_GENEMS = 2, PH_SET = 3, PH_FEAS = 4, PH_OBS = 5, PH_STORE = 6, PH_STEPS = 7,
       // detail of PH_SET (they sum to it): tuple generation + membership probes, batch de-duplication,
       // matching passes, table rebuilds
       PH_SET_GEN = 8, PH_SET_DEDUP = 9, PH_SET_MATCH = 10, PH_SET_REBUILD = 11,
       // plain per-step statistics (not cycles)
       ST_EMS = 12, ST_DISTINCT = 13, ST_GENERATED = 14, ST_MATCH_CALLS = 16, ST_MATCH_ROUNDS = 17, ST_MATCH_PROBES = 18,
       ST_CONTAINS_CALLS = 19, ST_CONTAINS_PROBES = 20, ST_FLUSHES = 21, PH_FAST_START = 22, ST_REBUILDS = 23,
       PH_SET_HASH = 24, PH_GEN_TUPLE = 25, PH_GEN_HASH = 26, PH_GEN_CONTAINS = 27, PH_GEN_PEND = 28, PH_GEN_PAIRS = 29,
       // stability settings, slots of their own (round 4 shared slots 15 / 19 / 20 / 24..27 with the set statistics, which a timed
       // stability build also fills -- ADVICE r4): commit walk visits, virtual-check passes / tasks / single-task passes / level-0
       // candidates, least-squares splits by supporter count
       ST_STAB_COMMIT_VISITS = 30, ST_STAB_VPASSES = 31, ST_STAB_VTASKS = 32, ST_STAB_VNARROW = 33, ST_STAB_LSQ3 = 34,
       ST_STAB_LSQ4 = 35, ST_STAB_LSQ5 = 36, ST_STAB_LSQX = 37, ST_STAB_LEVEL0 = 38, ST_STAB_LSQ_ROUNDS = 39,
       ST_STAB_LSQ_ROUNDS_L0 = 40, ST_STAB_COMMIT_ROUNDS = 41, ST_STAB_VROUNDS = 42, ST_STAB_VCALLS = 43 };


}  // namespace pct
#endif
