// Multi-wave candidate set: the EMS-rich envs of a launch -- the ones whose single wave would still be running
// long after every other env of the launch has finished -- build their CPython-ordered candidate set with the
// four waves of a 256-thread workgroup instead of one (pct_discrete_kernel_mw).  The result is the same table,
// bit for bit, as set_insert's: the same fresh-set fast start, the same growth points, the same re-insertion
// order; only the batch is 256 positions wide (one chunk of 64 (EMS, rotation) pairs = up to 256 tuples),
// a table rebuild is ONE matching pass over all old slots, and the wave collectives (ballot counts, "anybody
// still walking") go through a few LDS control words and real barriers.
//
// Rules of the cooperative section: all four waves execute it with IDENTICAL control flow (every loop bound and
// branch below depends only on values every wave holds identically), __syncthreads() is a real s_barrier here
// and appears nowhere else in the env code (PCT_SYNC is wave-local).
// Included from pct_discrete_impl.cuh (needs Pack, tuplehash6, Lds, SetState, walk_advance, ...).
#pragma once

namespace pct {

enum { MW_CMD_BUILD = 1, MW_CMD_EXIT = 2 };
// control words in LDS: [0] command, [1] EMS count, [2] item (10 bits per edge), [3..5] rotating gather words
struct MwCtl {
  uint32_t* w;
  uint32_t k;  // gathers done so far (identical in every wave)
};

// every wave contributes one byte, every wave gets all four (byte i = wave i's).  One barrier.
// Three words in rotation: the word of gather k is zeroed by wave 0 after the barrier of gather k + 1, when every
// wave has read it, and is used again by gather k + 3, whose writes follow the barrier of gather k + 2.
__device__ inline uint32_t mw_gather(MwCtl& c, int wv, int lane, uint32_t byte) {
  uint32_t* W = c.w + 3 + (c.k % 3u);
  if (lane == 0 && byte) atomicOr(W, byte << (8 * wv));
  __syncthreads();
  const uint32_t v = *W;
  if (wv == 0 && lane == 0) c.w[3 + ((c.k + 2u) % 3u)] = 0u;
  c.k++;
  return v;
}
__device__ inline uint32_t mw_bytes_sum(uint32_t v) { return (v & 0xFFu) + ((v >> 8) & 0xFFu) + ((v >> 16) & 0xFFu) + (v >> 24); }
__device__ inline uint32_t mw_bytes_below(uint32_t v, int wv) {  // sum of the bytes of the waves before wv
  uint32_t s = 0;
  s += wv > 0 ? (v & 0xFFu) : 0u;
  s += wv > 1 ? ((v >> 8) & 0xFFu) : 0u;
  s += wv > 2 ? ((v >> 16) & 0xFFu) : 0u;
  return s;
}

// pyset_match_v across the waves: position `pos[v]` (its rank in insertion order) is what the tag carries.
template <int V, typename K, bool CHECK>
__device__ inline void mw_match(MwCtl& c, int wv, int lane, K* tab, uint32_t mask, const bool (&part)[V],
                                const uint64_t (&hash)[V], const uint32_t (&pos)[V], const K (&key)[V],
                                uint32_t (&slot)[V], bool (&placed)[V]) {
  const K TAG = SlotWord<K>::TAG;
  uint32_t i[V];
  int j[V];
  uint64_t perturb[V];
  bool walking[V];
#pragma unroll
  for (int v = 0; v < V; v++) {
    i[v] = (uint32_t)hash[v] & mask;
    j[v] = 0;
    perturb[v] = hash[v];
    walking[v] = part[v];
    placed[v] = false;
    slot[v] = 0;
  }
  while (true) {
    while (__ballot(any_of<V>(walking))) {
      K old[V];
      uint32_t cur[V];
#pragma unroll
      for (int v = 0; v < V; v++) {
        cur[v] = i[v] + (uint32_t)j[v];
        old[v] = 0;
        if (walking[v]) old[v] = lds_atomic_min(&tab[cur[v]], TAG | (K)pos[v]);
      }
#pragma unroll
      for (int v = 0; v < V; v++) {
        const bool won = walking[v] && old[v] > (TAG | (K)pos[v]);
        const bool member = CHECK && walking[v] && !(old[v] & TAG) && old[v] == key[v];
        slot[v] = won ? cur[v] : slot[v];
        placed[v] = placed[v] || won;
        if (walking[v]) walk_advance(i[v], j[v], perturb[v], mask);
        walking[v] = walking[v] && !won && !member;
      }
    }
    __syncthreads();  // every wave's walkers have settled: nobody touches the table until the gather below
#pragma unroll
    for (int v = 0; v < V; v++) {
      if (placed[v] && tab[slot[v]] != (TAG | (K)pos[v])) {  // evicted by an earlier position: walk on
        placed[v] = false;
        walking[v] = true;
      }
    }
    if (!mw_gather(c, wv, lane, __ballot(any_of<V>(walking)) ? 1u : 0u)) break;
  }
}

// The cooperative build.  Wave 0 publishes E and the item in c.w[1..2] before the barrier that starts it.
// On return every wave holds the same st.{toff,size,fill,overflow}; the table is complete and visible.
template <typename K, int BITS>
__device__ inline void mw_build_set(const DiscreteParams& p, Lds<K, BITS>& l, MwCtl& c, int wv, int lane, SetState<K>& st) {
  typedef Pack<K, BITS> P;
  const K EMPTY = SlotWord<K>::EMPTY;
  K* const tabs = l.tab0;
  uint32_t* const dd = l.dd;
  const int E = (int)c.w[1];
  const uint32_t itw = c.w[2];
  const int b0 = (int)(itw & 0x3FFu), b1 = (int)((itw >> 10) & 0x3FFu), b2 = (int)(itw >> 20);
  constexpr int orient = 6;
  const int NP = E * orient;
  const int gl = wv * 64 + lane;
  c.k = 0;
  st.tabs = l.tab0;
  st.dd = l.dd;
  st.cap = (uint32_t)p.cand_cap;
  st.size = 8;
  st.fill = 0;
  st.toff = 0;
  st.overflow = false;
  if (gl < 8) tabs[gl] = EMPTY;
  if (gl < 128) dd[gl] = 0xFFFFFFFFu;
  auto rot_size = [&](int rot, int& sx, int& sy, int& sz) -> bool {
    switch (rot) {
      case 0: sx = b0; sy = b1; sz = b2; return false;
      case 1: sx = b1; sy = b0; sz = b2; return sx == sy;
      case 2: sx = b0; sy = b2; sz = b1; return sx == sy && sy == sz;
      case 3: sx = b1; sy = b2; sz = b0; return sx == sy && sy == sz;
      case 4: sx = b2; sy = b0; sz = b1; return sx == sy;
      default: sx = b2; sy = b1; sz = b0; return sx == sy;
    }
  };
  const bool e01 = b0 == b1, e02 = b0 == b2, e12 = b1 == b2;
  const bool g1 = !e01, g2 = !e12, g3 = !(e01 && e12) && !(g1 && e02) && !(g2 && e01);
  const bool g4 = !e02 && !(g1 && e12), g5 = !e12 && !e02 && !(g4 && e01);
  const uint32_t rotmask = 1u | (g1 ? 2u : 0u) | (g2 ? 4u : 0u) | (g3 ? 8u : 0u) | (g4 ? 16u : 0u) | (g5 ? 32u : 0u);
  // tuple at batch position tt of the current chunk (l.vp: the chunk's valid pairs)
  auto tuple_at = [&](int tt) -> K {
    const int qq = (int)l.vp[tt >> 2];
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
  __syncthreads();

  for (int pbase = 0; pbase < NP && !st.overflow; pbase += 64) {
    // the chunk's (EMS, rotation) pairs that can hold the item: every wave computes the same mask
    const int q = pbase + lane;
    bool pv = q < NP;
    const int ei = q / orient, rot = q - ei * orient;
    int sx, sy, sz;
    const bool skip = rot_size(rot, sx, sy, sz) || !((rotmask >> rot) & 1u);
    const K ek = pv ? l.ems_a[ei] : (K)0;
    pv = pv && !skip && (P::get(ek, 3) - P::get(ek, 0) >= sx) && (P::get(ek, 4) - P::get(ek, 1) >= sy) &&
         (P::get(ek, 5) - P::get(ek, 2) >= sz);
    const uint64_t pm = __ballot(pv);
    const int nt = 4 * __popcll(pm);
    if (!nt) continue;
    __syncthreads();  // the previous chunk's readers of l.vp are done
    if (wv == 0 && pv) l.vp[rank_below(pm)] = (uint16_t)q;
    __syncthreads();
    // one tuple per thread, batch position = generation order
    const bool valid = gl < nt;
    const K key = valid ? tuple_at(gl) : (K)0;
    const uint64_t hash = tuplehash6<K, BITS>(key);
    bool pending = valid;
    {
      // exact removal of in-batch duplicates (the first occurrence stays): bucket scatter as in
      // batch_find_duplicates_v; the key of another position is recomputed from the pair list
      bool unresolved = pending;
      int round = 0;
      for (; round < 8; round++) {
        if (round > 0 && !mw_gather(c, wv, lane, __ballot(unresolved) ? 1u : 0u)) break;
        const uint32_t b = (uint32_t)(hash >> (3 + 7 * round)) & 127u;
        if (unresolved) atomicMin(&dd[b], (uint32_t)gl);
        __syncthreads();
        const uint32_t wpos = unresolved ? dd[b] : (uint32_t)gl;
        __syncthreads();
        const K kw = tuple_at(unresolved ? (int)wpos : 0);
        if (unresolved) {
          dd[b] = 0xFFFFFFFFu;
          if (wpos == (uint32_t)gl) unresolved = false;
          else if (kw == key) { pending = false; unresolved = false; }
        }
      }
      if (round == 8) {
        // (the resets of round 7 must land before anybody's next scatter)
        if (mw_gather(c, wv, lane, __ballot(unresolved) ? 1u : 0u)) {
          // eight rounds of pure collisions between distinct keys: exhaustive scan over the earlier positions
          for (int i2 = 0; i2 < nt; i2++) {
            const K ki = tuple_at(i2);
            if (unresolved && i2 < gl && ki == key) pending = false;
          }
        }
      }
    }
    uint32_t cnts = mw_gather(c, wv, lane, (uint32_t)__popcll(__ballot(pending)));
    if (st.fill == 0 && st.size == 8 && st.cap >= 128 && (cnts & 0xFFu) >= 19u) {
      // fresh-set fast start on wave 0's slice, exactly as set_insert does it: the 8- and 32-slot tables are
      // replayed on the scalar unit, the 128-slot table receives the 19 keys in 32-table slot order and then the
      // rest of the slice in one single-wave pass
      if (wv == 0) {
        const uint64_t pm0 = __ballot(pending);
        uint64_t rem = pm0;
        int t8 = 0xFF, t32 = 0xFF;
        uint32_t occ8 = 0, occ32 = 0;
        auto lane_hash = [&](int src) -> uint64_t {
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
        auto insert32 = [&](int src) {
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
        K* fin = reinterpret_cast<K*>(dd);
        const K from_slot = shfl_key<K>(key, t32 & 63);
        const bool in32 = lane < 32 && ((occ32 >> lane) & 1u);
        const bool later = (rem >> lane) & 1ull;
        if (in32) fin[rank_below((uint64_t)occ32)] = from_slot;
        if (later) fin[19 + rank_below(rem)] = key;
        const int total = 19 + __popcll(rem);
        tabs[lane] = EMPTY;
        tabs[64 + lane] = EMPTY;
        PCT_SYNC();
        const K mk = lane < total ? fin[lane] : (K)0;
        PCT_SYNC();
        dd[lane] = 0xFFFFFFFFu;
        PCT_SYNC();
        const bool mpart[1] = {lane < total};
        const uint64_t mhash[1] = {tuplehash6<K, BITS>(mk)};
        uint32_t mslot[1];
        pyset_match_v<1, K>(tabs, 127u, mpart, mhash, lane, mslot);
        if (mpart[0]) tabs[mslot[0]] = mk;
        pending = false;
      }
      st.toff = 0;
      st.size = 128;
      st.fill = cnts & 0xFFu;  // every pending key of wave 0's slice went in
      cnts &= ~0xFFu;
      __syncthreads();
    }
    bool first = true;
    while (true) {
      // (the counts of the first pass are already known)
      if (!first) cnts = mw_gather(c, wv, lane, (uint32_t)__popcll(__ballot(pending)));
      first = false;
      const uint32_t total = mw_bytes_sum(cnts);
      const uint32_t mask = st.size - 1;
      const uint32_t thr = (mask * 3u + 4u) / 5u;
      if (st.fill < thr) {
        if (!total) break;
        // at most thr - fill more keys go into this table: the first thr - fill pending positions
        const int budget = (int)(thr - st.fill);
        const uint64_t pmk = __ballot(pending);
        const bool part1 = pending && (int)mw_bytes_below(cnts, wv) + rank_below(pmk) < budget;
        const bool part[1] = {part1};
        const uint64_t h1[1] = {hash};
        const uint32_t pos[1] = {(uint32_t)gl};
        const K k1[1] = {key};
        uint32_t slot[1];
        bool placed[1];
        mw_match<1, K, true>(c, wv, lane, tabs + st.toff, mask, part, h1, pos, k1, slot, placed);
        if (placed[0]) tabs[st.toff + slot[0]] = key;
        pending = pending && !part1;
        st.fill += mw_bytes_sum(mw_gather(c, wv, lane, (uint32_t)__popcll(__ballot(placed[0]))));
      }
      if (st.fill >= thr) {  // set_table_resize(used * 4): every old slot re-inserted, in slot order, in one pass
        uint32_t newsize = 8;
        while (newsize <= st.fill * 4u) newsize <<= 1;
        if (newsize > st.cap) {
          st.overflow = true;
          break;
        }
        // old table: at most 512 slots (the next size is the capacity, 2048) = 2 per thread
        K oldk[2];
        uint32_t opos[2];
        bool opart[2];
        uint64_t ohash[2];
#pragma unroll
        for (int v = 0; v < 2; v++) {
          opos[v] = (uint32_t)(v * 256 + gl);
          oldk[v] = opos[v] < st.size ? tabs[st.toff + opos[v]] : EMPTY;
          opart[v] = oldk[v] != EMPTY;
          ohash[v] = tuplehash6<K, BITS>(oldk[v]);
        }
        __syncthreads();
        for (uint32_t s2 = (uint32_t)gl; s2 < newsize; s2 += 256u) tabs[s2] = EMPTY;
        __syncthreads();
        uint32_t oslot[2];
        bool oplaced[2];
        mw_match<2, K, false>(c, wv, lane, tabs, newsize - 1, opart, ohash, opos, oldk, oslot, oplaced);
#pragma unroll
        for (int v = 0; v < 2; v++)
          if (opart[v]) tabs[oslot[v]] = oldk[v];
        st.toff = 0;
        st.size = newsize;
        __syncthreads();
      }
    }
  }
  __syncthreads();
}

}  // namespace pct
