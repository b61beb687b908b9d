// GROUP BY <one-word key> -> COUNT(*) for group sets far beyond LDS, in two passes over the
// rows and without a histogram: the shape prism-ql's `count by <column>` lowers to
// (src/prism-ql/src/lower.rs:83-95) and ClickBench's GROUP BY "UserID".
//
// The general partitioned path (kernels_part.hip) reads the key column twice -- a histogram
// to place every bucket's rows exactly, then the scatter -- and sorts each tile of rows by
// bucket in LDS with four barriers per tile, its sixteen waves in step.  Here:
//
//   k_wc_sample    a sample of the scan (32 tiles) is counted into a hash table of 2^19 keys;
//                  a key whose count reaches the threshold becomes a candidate
//   k_wc_build     the candidates, most frequent first, fill a small table of HOT keys
//   k_wc_scatter   one pass over the rows.  A row whose key is hot is counted in LDS and goes
//                  nowhere (with skewed keys -- a few ids in a large share of the rows -- that
//                  is a third to a half of all rows, and it takes the heavy hitters out of the
//                  buckets).  Every other row is appended to its bucket's staging line in LDS
//                  (2048 buckets x 8 records); the lane that fills a line copies it out as 64
//                  contiguous bytes.  No tile, no sort, no barrier: a wave takes a ticket per
//                  row from the bucket's counter and waits only for that bucket's line.
//                  Every (workgroup, bucket) pair owns a fixed segment of the record array,
//                  sized from the expected rows with room to spare, so no offsets need
//                  computing beforehand; a segment that overflows (rows of one cold key
//                  clustered in one workgroup's share) voids the run and the caller falls back
//                  to the histogram path.
//   k_wc_agg       a workgroup per bucket: the bucket's segments are read (16 lanes a segment),
//                  the keys counted in an LDS table of two-entry buckets with 32-bit counters
//                  (12 bytes a group: 13 000 groups per bucket, 2.7e7 per query), the groups
//                  appended to the result list
//                  appended to the result list; the hot keys' totals join the list at the end
//
// Traffic per row: the key read once, the record written once and read once (24 bytes for an
// 8-byte key, less the hot rows' share of the last two).
#include <algorithm>

#include "part_device.hpp"

namespace prism {

namespace {

constexpr uint32_t kWcBlock = 1024;
constexpr uint32_t kWcBucketBits = 11, kWcBuckets = 1u << kWcBucketBits;
constexpr uint32_t kWcStage = 8;          // records of a staging line (64 bytes)
constexpr uint32_t kWcHotSlots = 1024;    // hot table: 512 buckets of two entries
constexpr uint32_t kWcCmBits = 19;        // entries of the sample's table (32 tiles: a quarter full at most)
constexpr uint32_t kWcMaxCand = 4096;
constexpr unsigned long long kWcEmpty = ~0ull;
constexpr int kWcMaxProbes = 128;
constexpr size_t kWcLdsBytes = 160 * 1024 - 1024;

typedef const __attribute__((address_space(1))) unsigned long long* WcWordPtr;

// One 32-bit mix of the key serves the scatter pass twice: its top bits are the bucket, and
// the two home buckets of the key in the hot table (kWcHotSlots / 2 = 2^9 buckets of two
// entries) come from its lower bits.  A key takes the first free entry of its first home
// bucket, else of its second: with one home bucket three of the hundred most frequent keys
// were likely to share one, and the one turned away overflowed its segments.  A look-up reads
// the second bucket only when the first is full and holds another two keys.
__device__ __forceinline__ uint32_t wc_hot_b1(uint32_t mix) { return (mix >> 3) & (kWcHotSlots / 2 - 1); }
__device__ __forceinline__ uint32_t wc_hot_b2(uint32_t mix) { return (mix * 0x9E3779B1u) >> (32 - 9); }
__device__ __forceinline__ uint32_t wc_cm_index(uint64_t key) {
  const uint32_t x = ((uint32_t)key ^ (uint32_t)(key >> 32)) * 0x9E3779B1u;
  return ((x ^ (x >> 16)) * 0x85EBCA6Bu) >> (32 - kWcCmBits);
}

// ---- the hot keys -----------------------------------------------------------------------

// (The workgroup counts its tile in an LDS table first and adds every key's count to the
// global counter once: a frequent key is a twentieth of the sample, and ten thousand updates of
// one address from all over the chip are carried out one after the other -- 0.14 ms.)
__global__ __launch_bounds__(kWcBlock) void k_wc_sample(AggParams p, WcParams w) {
  constexpr uint32_t S = 8192;  // a tile has 4096 rows: half load at most
  __shared__ unsigned long long skey[S];
  __shared__ uint32_t scnt[S];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < S; i += kWcBlock) { skey[i] = kWcEmpty; scnt[i] = 0; }
  const uint64_t tile = (uint64_t)blockIdx.x * p.sd.n_tiles / gridDim.x;
  PartSegPos pos;
  part_seg_init(p, pos, tile);
  PartTile tl[1];
  tl[0] = part_tile_next(p, pos, tile, p.sd.n_tiles);
  uint64_t lo[4], hi[4];
  uint32_t rowc[4], live;
  part_load_keys<1, 4>(p, tl, tid, lo, hi, rowc, live);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (!((live >> r) & 1u) || lo[r] == kWcEmpty) continue;
    uint32_t e = wc_cm_index(lo[r]) & (S - 1);
    for (;;) {
      unsigned long long k = __hip_atomic_load(&skey[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (k == kWcEmpty) {
        const unsigned long long old = atomicCAS(&skey[e], kWcEmpty, (unsigned long long)lo[r]);
        k = old == kWcEmpty ? lo[r] : old;
      }
      if (k == lo[r]) { atomicAdd(&scnt[e], 1u); break; }
      e = (e + 1) & (S - 1);
    }
  }
  __syncthreads();
  // The workgroup's counts join the sample's table in HBM, keyed by the whole key: the add that
  // takes a key's count across the threshold makes it a candidate, exactly once.  (Counters
  // shared by whatever keys hash alike would do for the counting, but then the add that
  // crosses the threshold may be a rare key's, and the frequent key beside it never becomes a
  // candidate: one run in three lost a key that way and overflowed its segments.)
  const uint32_t G = 1u << kWcCmBits;
  // (a thread's eight entries are taken through the steps together -- the claims, then the
  // additions -- so that it waits for the memory side twice, not sixteen times)
  constexpr int PER = S / kWcBlock;
  unsigned long long k[PER], g[PER];
  uint32_t n[PER], e[PER], before[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    k[j] = skey[tid + (uint32_t)j * kWcBlock];
    n[j] = scnt[tid + (uint32_t)j * kWcBlock];
    e[j] = wc_cm_index(k[j]);
    g[j] = 0;
    if (k[j] != kWcEmpty) g[j] = atomicCAS((unsigned long long*)&w.s_keys[e[j]], kWcEmpty, k[j]);
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (k[j] == kWcEmpty) continue;
    // not the key's entry: on along the probe path (rare at a quarter load)
    for (uint32_t probes = 0; g[j] != kWcEmpty && g[j] != k[j] && probes < 1024; ++probes) {
      e[j] = (e[j] + 1) & (G - 1);
      g[j] = atomicCAS((unsigned long long*)&w.s_keys[e[j]], kWcEmpty, k[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    before[j] = 0;
    if (k[j] != kWcEmpty && (g[j] == kWcEmpty || g[j] == k[j])) before[j] = atomicAdd(&w.s_cnt[e[j]], n[j]);
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (k[j] == kWcEmpty || !(g[j] == kWcEmpty || g[j] == k[j])) continue;
    if (before[j] < w.hot_threshold && before[j] + n[j] >= w.hot_threshold) {
      const uint32_t c = atomicAdd(w.n_cand, 1u);
      if (c < kWcMaxCand) w.cand[c] = e[j];
    }
  }
}

// One workgroup: the candidates whose counters ended highest take their places in the hot
// table first.  A key that finds both its buckets full is simply not hot.
__global__ __launch_bounds__(kWcBlock) void k_wc_build(WcParams w) {
  __shared__ unsigned long long t[kWcHotSlots];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < kWcHotSlots; i += kWcBlock) t[i] = kWcEmpty;
  const uint32_t n = *w.n_cand < kWcMaxCand ? *w.n_cand : kWcMaxCand;
  // a thread's candidates (at most four), fetched once for all the passes
  constexpr int PER = kWcMaxCand / kWcBlock;
  unsigned long long key[PER];
  uint32_t cnt[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const uint32_t i = tid + (uint32_t)j * kWcBlock;
    const uint32_t at = i < n ? w.cand[i] : 0u;
    key[j] = i < n ? w.s_keys[at] : kWcEmpty;
    cnt[j] = i < n ? w.s_cnt[at] : 0u;
  }
  __syncthreads();
  uint32_t upper = ~0u;
  for (int pass = 0; pass < 4; ++pass) {
    const uint32_t lower = w.hot_threshold << (2 * (3 - pass));  // 64, 16, 4, 1 times the threshold
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (cnt[j] < lower || cnt[j] >= upper || key[j] == kWcEmpty) continue;
      const uint32_t mix = bucket_hash32(key[j]), b1 = wc_hot_b1(mix), b2 = wc_hot_b2(mix);
      const uint32_t order[4] = {2 * b1, 2 * b1 + 1, 2 * b2, 2 * b2 + 1};
      for (int o = 0; o < 4; ++o) {
        const unsigned long long old = atomicCAS(&t[order[o]], kWcEmpty, key[j]);
        if (old == kWcEmpty || old == key[j]) break;
      }
    }
    upper = lower;
    __syncthreads();
  }
  for (uint32_t i = tid; i < kWcHotSlots; i += kWcBlock) w.hot_keys[i] = t[i];
}

// ---- the scatter pass ---------------------------------------------------------------------
//
// Per bucket: `fill` hands out tickets (ticket t: place t % 8 of generation t / 8 of the
// bucket's staging line), `prog` holds the records of the current generation already in
// place (low half) and the generations copied out so far (high half, modulo 2^16).  The
// holder of a ticket writes its record once its generation is the current one; whoever
// writes a generation's eighth record copies the line out and opens the next generation.
// The only wait is for a line that still holds the generation before, and it is done in one
// loop that tries every pending row of every lane in turn, so a lane never spins in front of
// a step that another lane of its own wave has to take first.

// NK: key columns requested a tile ahead (no validity bitmaps); 0: keys loaded where they are
// used, with their validity bits; kWcPlain8: ONE 8-byte key column, no NULLs, no selection
// bitmap, not a date_bin -- the key is (value - bias) << bit_pos, and the loop carries none of
// the scalar branches on widths and key kinds that the general loaders take per row (they were
// a third of the wave's issue cycles).
constexpr int kWcPlain8 = 9;

template <int NK>
__global__ __launch_bounds__(kWcBlock) __attribute__((amdgpu_waves_per_eu(4))) void k_wc_scatter(AggParams p, WcParams w) {
  constexpr bool PLAIN = NK == kWcPlain8;
  constexpr bool AHEAD = NK > 0 && !PLAIN;
  constexpr int NA = AHEAD ? NK : 1;
  constexpr int R = 4;  // rows of a scan tile per thread: a workgroup takes one tile per step
  extern __shared__ __align__(16) unsigned long long wc_lds[];
  unsigned long long* stage = wc_lds;                                        // [kWcBuckets][kWcStage]
  uint32_t* fill = (uint32_t*)(stage + (size_t)kWcBuckets * kWcStage);       // [kWcBuckets]
  uint32_t* prog = fill + kWcBuckets;                                        // [kWcBuckets]
  unsigned long long* hot_keys = (unsigned long long*)(prog + kWcBuckets);   // [kWcHotSlots]
  uint32_t* hot_cnt = (uint32_t*)(hot_keys + kWcHotSlots);                   // [kWcHotSlots]
  const uint32_t tid = threadIdx.x, wg = blockIdx.x;
  for (uint32_t i = tid; i < kWcBuckets; i += kWcBlock) { fill[i] = 0; prog[i] = 0; }
  for (uint32_t i = tid; i < kWcHotSlots; i += kWcBlock) { hot_keys[i] = w.hot_keys[i]; hot_cnt[i] = 0; }
  __syncthreads();
  const uint64_t t0 = (uint64_t)wg * w.tiles_per_wg;
  const uint64_t t1 = t0 + w.tiles_per_wg < p.sd.n_tiles ? t0 + w.tiles_per_wg : p.sd.n_tiles;
  const uint32_t cap = 1u << w.cap_shift;
  unsigned long long raw[NA][R];
  uint32_t live = 0, live_next = 0;
  PartCursor<NA> cur;
  PartSegPos pos;
  // kWcPlain8: the four rows of the lane in tile `t` -- two pairs of neighbouring rows, 16 bytes
  // a load (which rows a lane takes is of no consequence here) -- requested two tiles before they
  // are used: with 8-byte loads one tile ahead a CU had 32 KiB in flight and the pass read at
  // 3.5 TB/s with everything else switched off.
  typedef unsigned long long __attribute__((ext_vector_type(2))) wc_u64x2;
  wc_u64x2 pre[2][2];      // [tile parity][pair]
  uint32_t pre_live[2] = {0, 0};
  auto plain_issue = [&](uint64_t t, int slot) {
    if constexpr (!PLAIN) return;
    else {
    while (t >= cur.next_base) part_cursor_load<1, 0>(p, cur, cur.seg + 1);
    const uint64_t first = (t - cur.base) * kTileRows, left = cur.rows - first;
    const uint32_t n = left > kTileRows ? kTileRows : (uint32_t)left, row0 = (uint32_t)first;
    uint32_t lv = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t in_tile = (uint32_t)h * 2u * kWcBlock + 2u * tid;  // the pair's first row
      lv |= (in_tile < n ? 1u : 0u) << (2 * h);
      lv |= (in_tile + 1 < n ? 1u : 0u) << (2 * h + 1);
      // (a pair beyond the tile's rows reads the tile's first pair; one that straddles its end reads
      // eight bytes into the column's padding)
      pre[slot][h] = __builtin_nontemporal_load(
          (const __attribute__((address_space(1))) wc_u64x2*)(cur.col[0] + (row0 + (in_tile < n ? in_tile : 0u)) * 8u));
    }
    pre_live[slot] = lv;
    }
  };
  const uint64_t plain_bias = (uint64_t)p.keys[0].bias;
  const int plain_shift = p.keys[0].bit_pos;
  if constexpr (PLAIN) {
    if (t0 < t1) {
      part_cursor_init<1, 0>(p, cur, t0);
      plain_issue(t0, 0);
      if (t0 + 1 < t1) plain_issue(t0 + 1, 1);
    }
  } else if constexpr (AHEAD) {
    if (t0 < t1) {
      part_cursor_init<NA, 0>(p, cur, t0);
      part_issue<NA, 0, R>(p, cur, t0, t1, tid, raw, live_next);
    }
  } else {
    if (t0 < t1) part_seg_init(p, pos, t0);
  }
  for (uint64_t t = t0; t < t1; ++t) {
    uint64_t key[R];
    if constexpr (PLAIN) {
      // (the two slots are taken in turn: the loop is unrolled by hand so that they stay registers)
      const int slot = (int)((t - t0) & 1u);
      if (slot == 0) {
        key[0] = pre[0][0].x; key[1] = pre[0][0].y; key[2] = pre[0][1].x; key[3] = pre[0][1].y;
        live = pre_live[0];
        if (t + 2 < t1) plain_issue(t + 2, 0);
      } else {
        key[0] = pre[1][0].x; key[1] = pre[1][0].y; key[2] = pre[1][1].x; key[3] = pre[1][1].y;
        live = pre_live[1];
        if (t + 2 < t1) plain_issue(t + 2, 1);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) key[r] = (key[r] - plain_bias) << plain_shift;
    } else if constexpr (AHEAD) {
      uint64_t rec[1][R];
      part_finish<1, NA, 0, R>(p, raw, rec);
#pragma unroll
      for (int r = 0; r < R; ++r) key[r] = rec[0][r];
      live = live_next;
      if (t + 1 < t1) part_issue<NA, 0, R>(p, cur, t + 1, t1, tid, raw, live_next);
    } else {
      PartTile tl[1];
      uint64_t hi_unused[R];
      uint32_t rowc[R];
      tl[0] = part_tile_next(p, pos, t, t1);
      part_load_keys<1, R>(p, tl, tid, key, hi_unused, rowc, live);
    }
    // hot keys are counted where they are; the other rows take a ticket of their bucket
    uint32_t mix[R], bkt[R], tk[R], needw = 0;
    ulonglong2 hc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      mix[r] = bucket_hash32(key[r]);
      bkt[r] = mix[r] >> (32 - kWcBucketBits);
      hc1[r] = *(const ulonglong2*)&hot_keys[2 * wc_hot_b1(mix[r])];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      tk[r] = 0;
      if (!((live >> r) & 1u)) continue;
      int hot = -1;
      hot = hc1[r].y == key[r] ? (int)(2 * wc_hot_b1(mix[r]) + 1) : hot;
      hot = hc1[r].x == key[r] ? (int)(2 * wc_hot_b1(mix[r])) : hot;
      if (hot < 0 && hc1[r].y != kWcEmpty) {  // the first home bucket is full of other keys: the key may sit in its second
        // (taken row by row: gathering a lane's rows that need it for one look-up measured slower)
        const uint32_t b2 = wc_hot_b2(mix[r]);
        const ulonglong2 hc2 = *(const ulonglong2*)&hot_keys[2 * b2];
        hot = hc2.y == key[r] ? (int)(2 * b2 + 1) : hot;
        hot = hc2.x == key[r] ? (int)(2 * b2) : hot;
      }
      if (hot >= 0) {
        atomicAdd(&hot_cnt[hot], 1u);
      } else {
        tk[r] = atomicAdd(&fill[bkt[r]], 1u);
        needw |= 1u << r;
      }
    }
    // Each round takes the pending rows of the lane through the same steps together -- the four
    // look-ups of `prog`, then the stores, then the four additions -- so that their LDS round
    // trips overlap instead of adding up row after row.  (LDS carries out a wave's
    // instructions in the order they were issued; the compiler is kept from reordering them,
    // and nothing has to wait for a store before the addition that follows it.)
    for (;;) {
      uint32_t pr[R], before[R], ok = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) pr[r] = __hip_atomic_load(&prog[bkt[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t gen = tk[r] / kWcStage, place = tk[r] % kWcStage;
        if (((needw >> r) & 1u) && (pr[r] >> 16) == (gen & 0xffffu)) {  // (else the line still holds the generation before)
          ok |= 1u << r;
          __hip_atomic_store(&stage[(size_t)bkt[r] * kWcStage + place], (unsigned long long)key[r], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < R; ++r) {
        before[r] = 0;
        if ((ok >> r) & 1u) before[r] = __hip_atomic_fetch_add(&prog[bkt[r]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      needw &= ~ok;
      // The rows that put the eighth record of a line in place copy the line out, 64 contiguous
      // bytes of the segment (nobody waits for anybody: whoever writes last, copies).  A lane
      // seldom has two such rows in a round, so the copy-out is issued once per lane, for its
      // first such row, and again only while some lane has another: a store instruction costs the
      // address unit the same sixteen cycles whether four of its lanes are active or sixty-four,
      // and one copy-out per ROW of the round made this a third of the pass.
      uint32_t fl = 0;
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (((ok >> r) & 1u) && (before[r] & 0xffffu) == kWcStage - 1) fl |= 1u << r;
      while (__ballot(fl != 0u) != 0ull) {
        if (fl != 0u) {
          const int r = __ffs((int)fl) - 1;
          fl &= fl - 1u;
          const uint32_t fb = r == 0 ? bkt[0] : r == 1 ? bkt[1] : r == 2 ? bkt[2] : bkt[3];
          const uint32_t ft = r == 0 ? tk[0] : r == 1 ? tk[1] : r == 2 ? tk[2] : tk[3];
          const ulonglong2* s = (const ulonglong2*)&stage[(size_t)fb * kWcStage];
          asm volatile("" ::: "memory");
          const ulonglong2 a = s[0], b = s[1], c = s[2], d = s[3];
          asm volatile("" ::: "memory");
          // the records are on their way to registers: the next generation may come (-8 records, +1 generation)
          __hip_atomic_fetch_add(&prog[fb], 0x10000u - kWcStage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const uint64_t off = (uint64_t)(ft / kWcStage) * kWcStage;
          if (off + kWcStage <= cap) {
            ulonglong2* dst = (ulonglong2*)(w.rec + ((((uint64_t)fb * w.n_wg + wg) << w.cap_shift) + off));
            dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
          } else {
            w.flags[1] = 1u;
          }
        }
      }
      if (__ballot(needw != 0u) == 0ull) break;
    }
  }
  __syncthreads();
  // what is left in the staging lines, and how many records every segment got
  for (uint32_t b = tid; b < kWcBuckets; b += kWcBlock) {
    const uint32_t n = fill[b], rem = n % kWcStage;
    const uint64_t off = (uint64_t)(n / kWcStage) * kWcStage;
    const bool fits = n <= cap;
    if (!fits) w.flags[1] = 1u;
    if (fits && rem) {
      unsigned long long* dst = (unsigned long long*)w.rec + ((((uint64_t)b * w.n_wg + wg) << w.cap_shift) + off);
      for (uint32_t j = 0; j < rem; ++j) dst[j] = stage[(size_t)b * kWcStage + j];
    }
    w.seg_count[(uint64_t)b * w.n_wg + wg] = fits ? n : cap;
  }
  for (uint32_t i = tid; i < kWcHotSlots; i += kWcBlock)
    if (hot_cnt[i]) atomicAdd(&w.hot_count[i], (unsigned long long)hot_cnt[i]);
}

// ---- the aggregation pass -------------------------------------------------------------------

// Entry of the key in the table of two-entry buckets (two home buckets, then the buckets
// after the second), inserted when absent; -1: no room on its path.
__device__ __forceinline__ int wc_entry(unsigned long long* keys, uint32_t nb, uint64_t key, bool& inserted) {
  const uint32_t m1 = ((uint32_t)key ^ (uint32_t)(key >> 32)) * 0x9E3779B1u;
  const uint32_t m2 = (m1 ^ (m1 >> 15)) * 0x85EBCA6Bu;
  uint32_t b = __umulhi(m1, nb);
  int step = 0;
  while (step < kWcMaxProbes) {
    asm volatile("" ::: "memory");  // the bucket is read afresh in every step
    const ulonglong2 c = *(const ulonglong2*)&keys[2 * (size_t)b];
    if (c.x == key) return (int)(2 * b);
    if (c.y == key) return (int)(2 * b + 1);
    if (c.x == kWcEmpty || c.y == kWcEmpty) {
      const uint32_t e = 2 * b + (c.x == kWcEmpty ? 0u : 1u);
      const unsigned long long old = atomicCAS(&keys[e], kWcEmpty, (unsigned long long)key);
      if (old == kWcEmpty) { inserted = true; return (int)e; }
      if (old == key) return (int)e;
      continue;  // somebody else took the entry: look at the bucket again
    }
    b = step == 0 ? __umulhi(m2, nb) : (b + 1 == nb ? 0u : b + 1);
    ++step;
  }
  return -1;
}

// The hot keys' totals join the list (one workgroup; run by the aggregation pass' first
// workgroup once its buckets are done: the list has no order, and a launch of its own for a
// thousand entries cost more than the work).
__device__ __forceinline__ void wc_hot_emit(const WcParams& w, uint32_t cut) {
  const uint32_t i = threadIdx.x, lane = i & 63u;
  const unsigned long long k = i < kWcHotSlots ? w.hot_keys[i] : kWcEmpty;
  const unsigned long long c = i < kWcHotSlots ? w.hot_count[i] : 0ull;
  const bool used = k != kWcEmpty && c != 0ull;
  const bool occ = used && c >= (unsigned long long)cut;
  // (the rows the hot keys took: the caller sizes the next run's segments for the rest)
  const unsigned long long wave_rows = wave_sum_u64(used ? c : 0ull);
  if (lane == 0 && wave_rows) atomicAdd(w.hot_rows, wave_rows);
  const uint64_t m = __ballot(occ);
  unsigned long long wave_first = 0;
  if (lane == 0 && m) wave_first = atomicAdd(w.list_cursor, (unsigned long long)__popcll(m));
  wave_first = __shfl(wave_first, 0);
  if (!occ) return;
  const uint64_t at = wave_first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  if (at < w.list_capacity) {
    w.list_keys[at] = k;
    w.list_state[at] = (long long)c;
  } else {
    w.flags[0] = 1u;
  }
}

__global__ __launch_bounds__(kWcBlock) void k_wc_agg(AggParams p, WcParams w) {
  extern __shared__ __align__(16) unsigned long long wc_lds[];
  __shared__ uint32_t n_entries, wg_pos, cut_found;
  __shared__ unsigned long long out_base;
  const uint32_t E = w.lds_entries, nb = E / 2, tid = threadIdx.x, lane = tid & 63u;
  unsigned long long* keys = wc_lds;              // [E]
  uint32_t* cnt = (uint32_t*)(keys + E);          // [E]
  uint32_t* segcnt = cnt + E;                     // [n_wg]
  // The count a group must reach to be listed (see WcParams::leading): every workgroup ranks the
  // hot keys' totals itself -- a thousand numbers, each thread its own against all of them --
  // and all arrive at the same one.
  uint32_t cut = 0;
  if (w.leading) {
    static_assert(kWcHotSlots == kWcBlock, "one hot entry per thread");
    uint32_t* hc = (uint32_t*)keys;  // (the table's place, before the first bucket needs it)
    const uint32_t mine = w.hot_keys[tid] != kWcEmpty ? (uint32_t)w.hot_count[tid] : 0u;  // (fewer than 2^32 rows in a scan)
    hc[tid] = mine;
    if (tid == 0) cut_found = 0;
    __syncthreads();
    uint32_t rank = 0;
    for (uint32_t j = 0; j < kWcHotSlots; j += 4) {
      const uint4 o = *(const uint4*)&hc[j];
      rank += (o.x > mine || (o.x == mine && j + 0 < tid)) ? 1u : 0u;
      rank += (o.y > mine || (o.y == mine && j + 1 < tid)) ? 1u : 0u;
      rank += (o.z > mine || (o.z == mine && j + 2 < tid)) ? 1u : 0u;
      rank += (o.w > mine || (o.w == mine && j + 3 < tid)) ? 1u : 0u;
    }
    if (rank == w.leading - 1) cut_found = mine;  // (zero when fewer hot keys are in use)
    __syncthreads();
    cut = cut_found;
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) *w.cut = cut;
  }
  // 32 lanes a segment and five rows a lane: a step covers 160 records of each of the wave's two
  // segments -- a segment of the usual 100 to 130 records in one step with a quarter of the
  // lanes idle (16 lanes x 8 rows covered 128, and most waves went round a second time for the
  // one segment of their four that held a few more)
  constexpr uint32_t LPS = 32;
  const uint32_t sub = tid & (LPS - 1), seg_lane = tid / LPS;
  constexpr int U = 5;
  for (uint32_t b = blockIdx.x; b < kWcBuckets; b += gridDim.x) {
    for (uint32_t i = tid; i < E; i += kWcBlock) { keys[i] = kWcEmpty; cnt[i] = 0; }
    for (uint32_t i = tid; i < w.n_wg; i += kWcBlock) segcnt[i] = w.seg_count[(uint64_t)b * w.n_wg + i];
    if (tid == 0) { n_entries = 0; wg_pos = 0; }
    __syncthreads();
    const unsigned long long* region = (const unsigned long long*)w.rec + (((uint64_t)b * w.n_wg) << w.cap_shift);
    bool lost = false;
    uint32_t my_new = 0;  // keys this lane put into the table (added up per wave when the bucket is done)
    for (uint32_t seg0 = 0; seg0 < w.n_wg; seg0 += kWcBlock / LPS) {
      const uint32_t seg = seg0 + seg_lane;
      const uint32_t c = seg < w.n_wg ? segcnt[seg] : 0u;
      const unsigned long long* s = region + ((uint64_t)(seg < w.n_wg ? seg : 0u) << w.cap_shift);
      uint32_t cmax = c;  // the longer of the wave's two segments: the wave's trip count
      { const uint32_t o = __shfl_xor(cmax, 32, 64); cmax = o > cmax ? o : cmax; }
      cmax = __builtin_amdgcn_readfirstlane(cmax);
      // the keys of the next step are requested before those of this step are looked up
      uint64_t next[U];
      auto request = [&](uint32_t i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t j = i0 + (uint32_t)u * LPS + sub;
          next[u] = __builtin_nontemporal_load((WcWordPtr)(uint64_t)(s + (j < c ? j : 0u)));
        }
      };
      if (cmax) request(0);
      for (uint32_t i0 = 0; i0 < cmax; i0 += LPS * U) {
        uint64_t key[U];
        bool in[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          key[u] = next[u];
          in[u] = i0 + (uint32_t)u * LPS + sub < c;
        }
        if (i0 + LPS * U < cmax) request(i0 + LPS * U);
        // the first home bucket of all the step's rows is read before any is looked at, and the
        // claims of the new keys among them go out together: the LDS round trips overlap
        ulonglong2 c1[U];
        uint32_t h1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          h1[u] = __umulhi(((uint32_t)key[u] ^ (uint32_t)(key[u] >> 32)) * 0x9E3779B1u, nb);
          c1[u] = *(const ulonglong2*)&keys[2 * (size_t)h1[u]];
        }
        int e[U];
        unsigned long long old[U];
        bool claim[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          e[u] = c1[u].x == key[u] ? (int)(2 * h1[u]) : (c1[u].y == key[u] ? (int)(2 * h1[u] + 1) : -1);
          // a new key, as a rule, and a free entry in its first home bucket: one attempt to claim it
          // (the entry was free when the bucket was read and entries only fill up: if the claim
          // succeeds or meets the same key, the entry is the key's)
          claim[u] = in[u] && e[u] < 0 && c1[u].y == kWcEmpty;
          old[u] = 0;
          if (claim[u])
            old[u] = atomicCAS(&keys[2 * h1[u] + (c1[u].x == kWcEmpty ? 0u : 1u)], kWcEmpty, (unsigned long long)key[u]);
        }
        uint32_t miss = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool ins = claim[u] && old[u] == kWcEmpty;
          if (claim[u] && (old[u] == kWcEmpty || old[u] == key[u])) e[u] = (int)(2 * h1[u] + (c1[u].x == kWcEmpty ? 0u : 1u));
          if (in[u] && e[u] < 0) miss |= 1u << u;
          my_new += ins ? 1u : 0u;
          if (in[u] && e[u] >= 0) atomicAdd(&cnt[e[u]], 1u);
        }
        // The rows whose first home bucket held two other keys, or whose claim another lane won:
        // a few lanes of the wave, seldom two rows of one lane.  The walk along the probe path is
        // taken once for all the step's rows (a lane's first such row, then its next while any lane
        // has one): taken row by row, nearly every row of the step had some lane in it, and the
        // whole wave went through the walk five times.
        while (__ballot(miss != 0u) != 0ull) {
          bool ins = false;
          if (miss != 0u) {
            const int u = __ffs((int)miss) - 1;
            miss &= miss - 1u;
            uint64_t k = key[0];
#pragma unroll
            for (int v = 1; v < U; ++v) k = u == v ? key[v] : k;
            const int at = wc_entry(keys, nb, k, ins);
            if (at < 0) lost = true;
            else atomicAdd(&cnt[at], 1u);
          }
          my_new += ins ? 1u : 0u;
        }
      }
    }
    if (lost) w.flags[2] = 1u;
    {
      const uint32_t wave_new = (uint32_t)wave_sum_u64(my_new);
      if (lane == 0 && wave_new) atomicAdd(&n_entries, wave_new);
    }
    __syncthreads();
    const uint32_t n = n_entries;
    if (cut) {
      // only the few groups that reach the cut are listed: a place each from the list's cursor
      if (tid == 0) atomicAdd(w.groups_total, (unsigned long long)n);
      for (uint32_t i0 = 0; i0 < E; i0 += kWcBlock) {
        const uint32_t i = i0 + tid;
        const uint32_t c = i < E ? cnt[i] : 0u;
        if (c < cut) continue;
        const unsigned long long at = atomicAdd(w.list_cursor, 1ull);
        if (at < w.list_capacity) {
          w.list_keys[at] = keys[i];
          w.list_state[at] = (long long)c;
        } else {
          w.flags[0] = 1u;
        }
      }
      __syncthreads();
      continue;
    }
    if (tid == 0) {
      out_base = atomicAdd(w.list_cursor, (unsigned long long)n);
      atomicAdd(w.groups_total, (unsigned long long)n);
    }
    __syncthreads();
    const uint64_t first = out_base;
    const bool fits = first + n <= w.list_capacity;
    if (!fits && tid == 0) w.flags[0] = 1u;
    for (uint32_t i0 = 0; i0 < E && fits; i0 += kWcBlock) {
      const uint32_t i = i0 + tid;
      const unsigned long long k = i < E ? keys[i] : kWcEmpty;
      const bool occ = k != kWcEmpty;
      const uint64_t m = __ballot(occ);
      uint32_t wave_first = 0;
      if (lane == 0 && m) wave_first = atomicAdd(&wg_pos, (uint32_t)__popcll(m));
      wave_first = __shfl(wave_first, 0);
      if (!occ) continue;
      const uint64_t at = first + wave_first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      w.list_keys[at] = k;
      w.list_state[at] = (long long)cnt[i];
    }
    __syncthreads();
  }
  if (blockIdx.x == 0) wc_hot_emit(w, cut);
}

void wc_allow_lds(const void* k, size_t bytes) {
  PRISM_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

}  // namespace

bool wc_plan_fits(const AggParams& p) {
  if (p.key_words != 1 || p.keys_only || p.key_bits > 63 || p.n_keys < 1) return false;
  if (p.n_slots != 1 || p.slots[0].op != kSlotAddOne) return false;
  return p.row_mode == kRowsAll || p.row_mode == kRowsBitmap;
}

uint32_t wc_lds_entries(uint32_t n_wg) { return (uint32_t)((kWcLdsBytes - (size_t)n_wg * 4 - 256) / 12) & ~1u; }
uint32_t wc_buckets() { return kWcBuckets; }
uint32_t wc_hot_slots() { return kWcHotSlots; }
uint32_t wc_max_candidates() { return kWcMaxCand; }
uint64_t wc_sample_counters() { return 1ull << kWcCmBits; }

// What the passes expect on entry, in one launch: the sample's keys all ones, its counts and the
// small counters behind them zero, the list's flag word zero.
__global__ __launch_bounds__(256) void k_wc_clear(unsigned long long* ones, uint64_t n_ones, unsigned long long* zeros, uint64_t n_zeros,
                                                  uint32_t* flag) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && flag) *flag = 0u;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_ones + n_zeros; i += (uint64_t)gridDim.x * 256) {
    if (i < n_ones) ones[i] = ~0ull;
    else zeros[i - n_ones] = 0ull;
  }
}

void launch_wc_clear(prism_gpu_ctx* ctx, unsigned long long* ones, uint64_t n_ones, unsigned long long* zeros, uint64_t n_zeros, uint32_t* flag) {
  const uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(n_ones + n_zeros, (uint64_t)256 * 8), (uint64_t)ctx->n_cus * 8);
  k_wc_clear<<<dim3(std::max(1u, grid)), dim3(256), 0, ctx->stream>>>(ones, n_ones, zeros, n_zeros, flag);
  PRISM_HIP(hipGetLastError());
}

void launch_wc_group_by(prism_gpu_ctx* ctx, const AggParams& p, const WcParams& w, uint64_t key_bytes) {
  if (!wc_plan_fits(p)) fail(PRISM_GPU_EINTERNAL, "plan shape does not fit the histogram-free GROUP BY");
  if (w.n_wg == 0 || (uint64_t)w.n_wg * w.tiles_per_wg < p.sd.n_tiles || w.cap_shift < 3 || w.cap_shift > 28 ||
      w.lds_entries > wc_lds_entries(w.n_wg) || w.lds_entries < 64 || (w.leading && (w.lds_entries < kWcHotSlots || w.leading > kWcHotSlots)) || w.sample_tiles == 0 || w.sample_tiles > p.sd.n_tiles)
    fail(PRISM_GPU_EINTERNAL, "bad work split for the histogram-free GROUP BY");
  const uint64_t rows = p.sd.n_tiles * kTileRows;
  {
    KernelTimer timer(ctx, "wc_sample", (uint64_t)w.sample_tiles * kTileRows * key_bytes, true);
    k_wc_sample<<<dim3(w.sample_tiles), dim3(kWcBlock), 0, ctx->stream>>>(p, w);
    PRISM_HIP(hipGetLastError());
    k_wc_build<<<dim3(1), dim3(kWcBlock), 0, ctx->stream>>>(w);
    PRISM_HIP(hipGetLastError());
  }
  {
    KernelTimer timer(ctx, "wc_scatter", rows * (key_bytes + 8), true);
    const size_t lds = (size_t)kWcBuckets * kWcStage * 8 + (size_t)kWcBuckets * 8 + (size_t)kWcHotSlots * 12;
    const int ahead = part_keys_ahead(p);
    const bool plain8 = ahead == 1 && p.n_keys == 1 && p.keys[0].width == 8 && !p.keys[0].date_bin && p.row_mode == kRowsAll;
    auto kernel = plain8 ? k_wc_scatter<kWcPlain8> : ahead == 1 ? k_wc_scatter<1> : ahead == 2 ? k_wc_scatter<2> : k_wc_scatter<0>;
    wc_allow_lds((const void*)kernel, lds);
    kernel<<<dim3(w.n_wg), dim3(kWcBlock), lds, ctx->stream>>>(p, w);
    PRISM_HIP(hipGetLastError());
  }
  {
    KernelTimer timer(ctx, "wc_agg", rows * 8, true);
    const size_t lds = (size_t)w.lds_entries * 12 + (size_t)w.n_wg * 4;
    wc_allow_lds((const void*)k_wc_agg, lds);
    const uint32_t grid = std::min<uint32_t>(kWcBuckets, (uint32_t)ctx->n_cus);
    k_wc_agg<<<dim3(grid), dim3(kWcBlock), lds, ctx->stream>>>(p, w);
    PRISM_HIP(hipGetLastError());
  }
}

}  // namespace prism
