// Partitioned GROUP BY: the path for group sets that fit neither a workgroup's LDS table
// nor a time window.
//
// A device-scope atomic on this part is carried out at the memory side, at roughly 20 G
// requests per second for the whole chip, whatever the size of the table it lands in.  A
// GROUP BY whose rows mostly miss the LDS table therefore runs at that rate (about 0.2 TB/s
// of input) instead of the 6 TB/s the columns could be read at.  This file removes the
// per-row atomics: the rows are first partitioned by a hash of their packed key into
// buckets, in HBM, with plain stores; then every bucket is aggregated by one workgroup in an
// LDS table (LDS atomics only), and because a key lives in exactly one bucket the finished
// groups are written out with plain stores as well, as a compact list.
//
//   k_part_hist     rows of every workgroup per bucket (LDS histogram)
//   k_part_offsets  per bucket: where each workgroup's rows start inside the bucket
//   k_part_plan     bucket starts, and the work items of the aggregation pass: a bucket, or
//                   a slice of a bucket that holds more than chunk_rows rows (a heavy hitter)
//   k_part_scatter  the same rows again: packed key and widened aggregate inputs of a row
//                   form a record of W 8-byte words; a tile of records is counting-sorted by
//                   bucket inside LDS and then copied out, so that neighbouring lanes store
//                   to neighbouring addresses of one bucket
//   k_part_agg      one work item per workgroup at a time: LDS open-addressing table over the
//                   item's records; a whole bucket appends its groups to the list (one atomic
//                   per bucket to reserve the space), a slice merges them into a small HBM
//                   hash table with atomics, as do rows that find the LDS table full
//
// Traffic per row: key and value columns twice (histogram and scatter), the record written
// once and read once.
#include <algorithm>

#include "part_device.hpp"

namespace prism {

namespace {

// ---- pass 1a: histogram ------------------------------------------------------------

template <int KW, int NK>
__global__ __launch_bounds__(kPartBlock) void k_part_hist(AggParams p, PartParams q) {
  constexpr int R = 8;
  extern __shared__ __align__(16) unsigned long long part_lds[];
  uint32_t* h = (uint32_t*)part_lds;
  const uint32_t tid = threadIdx.x, P = q.n_buckets;
  for (uint32_t b = tid; b < P; b += kPartBlock) h[b] = 0;
  __syncthreads();
  const uint64_t t0 = (uint64_t)blockIdx.x * q.tiles_per_wg;
  const uint64_t t1 = t0 + q.tiles_per_wg < p.sd.n_tiles ? t0 + q.tiles_per_wg : p.sd.n_tiles;
  const int shift = 64 - q.p_bits;
  constexpr bool ahead = NK > 0;
  constexpr int NA = NK > 0 ? NK : 1;
  PartTile tl[R / 4];
  unsigned long long raw[NA][R];
  // a partition by the high word: the columns of the low word are not read
  uint32_t skip = 0;
  if (KW == 2 && q.by_hi)
    for (int k = 0; k < p.n_keys; ++k) skip |= (p.keys[k].word ? 0u : 1u) << k;
  skip = __builtin_amdgcn_readfirstlane(skip);
#pragma unroll
  for (int k = 0; k < NA; ++k)
#pragma unroll
    for (int r = 0; r < R; ++r) raw[k][r] = 0;
  uint32_t live_next = 0;
  PartCursor<NA> cur;
  PartSegPos pos;
  if (!ahead && t0 < t1) part_seg_init(p, pos, t0);
  if (ahead && t0 < t1) {
    part_cursor_init<NA, 0>(p, cur, t0);
    part_issue<NA, 0, R>(p, cur, t0, t1, tid, raw, live_next, skip);
  }
  for (uint64_t t = t0; t < t1; t += R / 4) {
    uint64_t lo[R], hi[R];
    uint32_t live;
    if constexpr (ahead) {
      uint64_t rec[KW][R];
      part_finish<KW, NA, 0, R>(p, raw, rec);
#pragma unroll
      for (int r = 0; r < R; ++r) { lo[r] = rec[0][r]; hi[r] = KW == 2 ? rec[KW - 1][r] : 0ull; }
      live = live_next;
      if (t + R / 4 < t1) part_issue<NA, 0, R>(p, cur, t + R / 4, t1, tid, raw, live_next, skip);
    } else {
      uint32_t rowc[R];
#pragma unroll
      for (int j = 0; j < R / 4; ++j) tl[j] = part_tile_next(p, pos, t + j, t1);
      part_load_keys<KW, R>(p, tl, tid, lo, hi, rowc, live, skip);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if ((live >> r) & 1u) atomicAdd(&h[(uint32_t)(part_bucket_hash<KW, true>(q, lo[r], hi[r]) >> shift)], 1u);
  }
  __syncthreads();
  for (uint32_t b = tid; b < P; b += kPartBlock) q.hist[(uint64_t)blockIdx.x * P + b] = h[b];
}

// hist[wg][b] becomes the number of rows of bucket b that workgroups before wg hold;
// totals[b] the rows of the bucket.
__global__ __launch_bounds__(kBlock) void k_part_offsets(PartParams q) {
  const uint32_t b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= q.n_buckets) return;
  uint32_t run = 0;
  constexpr uint32_t kBatch = 8;  // independent loads in flight (the running sum is the only chain)
  for (uint32_t wg0 = 0; wg0 < q.n_wg; wg0 += kBatch) {
    uint32_t c[kBatch];
#pragma unroll
    for (uint32_t j = 0; j < kBatch; ++j) c[j] = wg0 + j < q.n_wg ? q.hist[(uint64_t)(wg0 + j) * q.n_buckets + b] : 0u;
#pragma unroll
    for (uint32_t j = 0; j < kBatch; ++j) {
      if (wg0 + j < q.n_wg) q.hist[(uint64_t)(wg0 + j) * q.n_buckets + b] = run;
      run += c[j];
    }
  }
  q.totals[b] = run;
}

// One workgroup: bucket starts and the work items of the aggregation pass.
__global__ __launch_bounds__(kPartBlock) void k_part_plan(PartParams q) {
  __shared__ uint32_t a[4 * kPartBlock + 1];
  __shared__ uint32_t wave_tot[kPartBlock / 64];
  const uint32_t tid = threadIdx.x, P = q.n_buckets;
  for (uint32_t b = tid; b < P; b += kPartBlock) a[b] = q.totals[b];
  __syncthreads();
  part_block_scan(a, P, wave_tot);
  __syncthreads();
  for (uint32_t b = tid; b <= P; b += kPartBlock) q.bucket_base[b] = a[b];
  __syncthreads();
  // Items: the slices of the buckets that are cut up come first (they are the long ones, and
  // the aggregation pass hands items out in order), then one item per remaining bucket.
  auto chunks_of = [&](uint32_t b) { return (uint32_t)ceil_div((uint64_t)q.totals[b], (uint64_t)q.chunk_rows); };
  for (uint32_t b = tid; b < P; b += kPartBlock) { const uint32_t n = chunks_of(b); a[b] = n > 1 ? n : 0u; }
  __syncthreads();
  part_block_scan(a, P, wave_tot);
  __syncthreads();
  const uint32_t n_sliced = a[P];
  auto emit_items = [&](uint32_t b, uint32_t first, uint32_t n) {
    const uint64_t lo = q.bucket_base[b], hi = q.bucket_base[b + 1];
    for (uint32_t c = 0; c < n; ++c) {
      if (first + c >= q.max_items) break;
      PartItem it;
      it.bucket = b;
      it.exclusive = n == 1 ? 1u : 0u;
      it.lo = lo + (uint64_t)c * q.chunk_rows;
      it.hi = it.lo + q.chunk_rows < hi ? it.lo + q.chunk_rows : hi;
      q.items[first + c] = it;
    }
  };
  for (uint32_t b = tid; b < P; b += kPartBlock) {
    const uint32_t n = chunks_of(b);
    if (n > 1) emit_items(b, a[b], n);
  }
  __syncthreads();
  for (uint32_t b = tid; b < P; b += kPartBlock) a[b] = chunks_of(b) == 1 ? 1u : 0u;
  __syncthreads();
  part_block_scan(a, P, wave_tot);
  __syncthreads();
  for (uint32_t b = tid; b < P; b += kPartBlock)
    if (chunks_of(b) == 1) emit_items(b, n_sliced + a[b], 1);
  if (tid == 0) {
    const uint32_t total = n_sliced + a[P];
    *q.n_items = total < q.max_items ? total : q.max_items;
  }
}

// ---- pass 1b: scatter ------------------------------------------------------------------

template <int W>
constexpr int part_rows_per_thread() { return W == 1 ? 16 : (W == 2 ? 8 : 4); }

// Per super-tile (R rows a thread): the rows are ranked inside their buckets with LDS counters,
// the counters are scanned, the records are put in bucket order in LDS and leave as runs.  The
// counters are double buffered (the next tile's are cleared while this tile's are still read)
// and every thread keeps the output cursors of its own buckets in registers, which leaves
// four barriers per tile.
//
// NK > 0 (NK key columns, no validity bitmaps to read): the column words
// of a tile are requested two tiles before they are used and turned into records one tile
// before, right ahead of the copy-out of the tile in between.  The wait for them is then also
// the wait for the stores of the copy-out a whole tile earlier (one counter covers loads and
// stores, in order), which have long completed; waiting right after a copy-out instead was
// half of this kernel's time.
//
// BLOCK = 512 (records of one or two words, few enough buckets): two workgroups share a CU,
// each with half the tile, so that one's LDS phases run while the other waits for its loads
// and stores; the sixteen waves of one 1024-thread workgroup go through a tile's phases in
// step, and then nothing overlaps.
template <int KW, int NV, bool VALID, int NK, int BLOCK>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void k_part_scatter(AggParams p, PartParams q) {
  constexpr bool AHEAD = NK > 0;
  static_assert(!(AHEAD && VALID), "validity bitmaps are read with the values, not ahead of them");
  constexpr int W = KW + NV + (VALID ? 1 : 0);
  constexpr int NA = AHEAD ? NK : 1;       // key columns requested ahead
  constexpr int WA = AHEAD ? NK + NV : 1;  // and all the columns
  constexpr int R = part_rows_per_thread<W>();
  constexpr uint32_t T = (uint32_t)R * BLOCK;
  constexpr int RT = (int)(kTileRows / BLOCK);  // rows of a scan tile per thread
  extern __shared__ __align__(16) unsigned long long part_lds[];
  __shared__ uint32_t wave_tot[BLOCK / 64];
  unsigned long long* sorted = part_lds;                        // [W][T]
  uint32_t* counters = (uint32_t*)(part_lds + (size_t)W * T);   // [2][P + 2]: rows of the tile per bucket, then their first slot
  const uint32_t tid = threadIdx.x, P = q.n_buckets;
  uint32_t* delta = counters + 2 * (P + 2);                     // [P]: a bucket's place in the output minus its place in the tile
  uint32_t cursor[4];  // next record of this workgroup in the buckets tid, tid + 1024, ...
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) {
    const uint32_t b = tid + i * BLOCK;
    cursor[i] = 0;
    if (b < P) {
      cursor[i] = (uint32_t)q.bucket_base[b] + q.hist[(uint64_t)blockIdx.x * P + b];
      counters[b] = 0;
      counters[P + 2 + b] = 0;
    }
  }
  __syncthreads();
  const uint64_t t0 = (uint64_t)blockIdx.x * q.tiles_per_wg;
  const uint64_t t1 = t0 + q.tiles_per_wg < p.sd.n_tiles ? t0 + q.tiles_per_wg : p.sd.n_tiles;
  const int shift = 64 - q.p_bits;
  constexpr uint64_t kStep = R / RT;
  static_assert(R % RT == 0 && R >= RT, "a super-tile is a whole number of scan tiles");
  uint64_t rec[W][R];
  uint32_t live = 0, live_next = 0;
  unsigned long long raw[AHEAD ? WA : 1][R];
  PartCursor<WA> cur;
  PartSegPos pos;
  if constexpr (!AHEAD) {
    if (t0 < t1) part_seg_init(p, pos, t0);
  }
  if constexpr (AHEAD) {
    if (t0 < t1) {
      part_cursor_init<NA, NV>(p, cur, t0);
      part_issue<NA, NV, R, BLOCK>(p, cur, t0, t1, tid, raw, live);
      part_finish<KW, NA, NV, R>(p, raw, rec);
    }
  }
  // `pending`: the next super-tile exists and the loads of its first scan tile are on their way;
  // the loads of its other tiles are issued between the phases of the loop body, a few at a
  // time (a burst of all of them only stalls in the issue: a CU holds so many misses)
  bool pending = false;
  if constexpr (AHEAD) {
    pending = t0 + kStep < t1;
    if (pending) part_issue_group<NA, NV, R, 0, BLOCK>(p, cur, t0 + kStep, t1, tid, raw, live_next);
  }
  uint32_t flip = 0;
  for (uint64_t t = t0; t < t1; t += kStep, flip ^= 1u) {
    uint32_t* start = counters + flip * (P + 2);
    uint32_t* start_next = counters + (flip ^ 1u) * (P + 2);
    if constexpr (!AHEAD) {
      PartTile tl[R / RT];
      uint32_t rowc[R];
      uint64_t hi_unused[R];
#pragma unroll
      for (int j = 0; j < R / RT; ++j) tl[j] = part_tile_next(p, pos, t + j, t1);
      part_load_keys<KW, R, BLOCK>(p, tl, tid, rec[0], KW == 2 ? rec[KW - 1] : hi_unused, rowc, live);
      if constexpr (NV > 0) {
        uint64_t val[NV][R], validw[R];
        part_load_vals<NV, VALID, R, BLOCK>(p, tl, rowc, val, validw);
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int r = 0; r < R; ++r) rec[KW + v][r] = val[v][r];
        if constexpr (VALID) {
#pragma unroll
          for (int r = 0; r < R; ++r) rec[W - 1][r] = validw[r];
        }
      }
    }
    if constexpr (AHEAD && R >= 8) {
      if (pending) part_issue_group<NA, NV, R, 1, BLOCK>(p, cur, t + kStep, t1, tid, raw, live_next);
    }
    // rank of every row among the tile's rows of its bucket
    uint32_t slot[R];  // bucket (low 12 bits) and rank inside it (a tile holds at most 2^14 rows)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t b = (uint32_t)(part_bucket_hash<KW>(q, rec[0][r], KW == 2 ? rec[KW - 1][r] : 0ull) >> shift);
      slot[r] = b;
      if ((live >> r) & 1u) slot[r] |= atomicAdd(&start[b], 1u) << 12;
    }
    __syncthreads();
    if constexpr (AHEAD && R >= 16) {
      if (pending) part_issue_group<NA, NV, R, 2, BLOCK>(p, cur, t + kStep, t1, tid, raw, live_next);
    }
    part_block_scan<BLOCK>(start, P, wave_tot);
    __syncthreads();
    if constexpr (AHEAD && R >= 16) {
      if (pending) part_issue_group<NA, NV, R, 3, BLOCK>(p, cur, t + kStep, t1, tid, raw, live_next);
    }
    // the tile's records, sorted by bucket, in LDS
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!((live >> r) & 1u)) continue;
      const uint32_t pos = start[slot[r] & 0xfffu] + (slot[r] >> 12);
#pragma unroll
      for (int w = 0; w < W; ++w) sorted[(size_t)w * T + pos] = rec[w][r];
    }
    // the thread's own buckets: where their runs go, the cursors behind them, and the other
    // set of counters cleared for the next tile
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      const uint32_t b = tid + i * BLOCK;
      if (b < P) {
        const uint32_t s0 = start[b], s1 = start[b + 1];
        delta[b] = cursor[i] - s0;
        cursor[i] += s1 - s0;
        start_next[b] = 0;
      }
    }
    if constexpr (AHEAD) {
      // the next tile's records, and the first requests for the tile after it
      if (pending) {
        part_finish<KW, NA, NV, R>(p, raw, rec);
        live = live_next;
        pending = t + 2 * kStep < t1;
        if (pending) {
          live_next = 0;
          part_issue_group<NA, NV, R, 0, BLOCK>(p, cur, t + 2 * kStep, t1, tid, raw, live_next);
        }
      }
    }
    __syncthreads();
    // out: neighbouring lanes hold neighbouring records of one bucket
    const uint32_t n_live = start[P];
    for (uint32_t i = tid; i < n_live; i += BLOCK) {
      uint64_t w0 = sorted[i], w1 = KW == 2 ? sorted[(size_t)T + i] : 0ull;
      const uint32_t b = (uint32_t)(part_bucket_hash<KW>(q, w0, w1) >> shift);
      const uint64_t g = (uint64_t)(delta[b] + i);
      if (g < q.rec_capacity) {
        q.rec[0][g] = w0;
        if (KW == 2) q.rec[1][g] = w1;
#pragma unroll
        for (int w = KW; w < W; ++w) q.rec[w][g] = sorted[(size_t)w * T + i];
      }
    }
    // (no barrier here: the next tile's ranks go to the other set of counters, and nothing
    // of this tile is overwritten before the next tile's first barrier)
  }
}

// ---- pass 2: aggregation of one bucket (or slice) in LDS -----------------------------------

constexpr unsigned long long kPartEmpty = ~0ull;
constexpr int kPartMaxProbes = 128;

__device__ __forceinline__ unsigned long long lds_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Entry of the key in the LDS table, inserted when absent; -1: the table is full for this key.
template <int KW>
__device__ __forceinline__ int part_lds_entry(unsigned long long* keys_lo, unsigned long long* keys_hi, uint32_t S,
                                              int limit, int* n_entries, uint64_t lo, uint64_t hi, uint64_t h, bool& inserted) {
  uint32_t idx = __umulhi((uint32_t)h, S);
  int probes = 0;
  while (probes < kPartMaxProbes) {
    unsigned long long k = lds_load(&keys_lo[idx]);
    if (k == kPartEmpty) {
      if (__hip_atomic_load(n_entries, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= limit) return -1;
      const unsigned long long old = atomicCAS(&keys_lo[idx], kPartEmpty, (unsigned long long)lo);
      if (old == kPartEmpty) {
        if (KW == 2) __hip_atomic_store(&keys_hi[idx], (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        inserted = true;  // the caller counts it (once per wave: the counter is one LDS word)
        return (int)idx;
      }
      k = old;
    }
    if (k == lo) {
      if (KW == 1) return (int)idx;
      // the claimant of the entry publishes the high word right after its CAS: until then it reads all ones
      const unsigned long long kh = lds_load(&keys_hi[idx]);
      if (kh == hi) return (int)idx;
      if (kh == kPartEmpty) continue;  // not published yet: look again
    }
    idx = idx + 1 == S ? 0u : idx + 1;
    ++probes;
  }
  return -1;
}

// One-word keys: the table is cut into buckets of two entries, read with one 16-byte load,
// and a key has two home buckets (tried in order, then the buckets after the second).  A
// wave's look-up takes as many steps as its slowest lane: with one home slot and linear
// probing that was eight or nine steps at half load, with two choices of two entries it is
// two or three.  A key is only ever placed in the first free entry on its path and entries
// are never removed, so a free entry on the path proves the key absent.
__device__ __forceinline__ int part_lds_entry2way(unsigned long long* keys, uint32_t n_buckets, int limit, int* n_entries,
                                                  uint64_t key, bool& inserted) {
  const uint32_t m1 = ((uint32_t)key ^ (uint32_t)(key >> 32)) * 0x9E3779B1u;
  const uint32_t m2 = (m1 ^ (m1 >> 15)) * 0x85EBCA6Bu;
  uint32_t b = __umulhi(m1, n_buckets);
  int step = 0;
  while (step < kPartMaxProbes) {
    asm volatile("" ::: "memory");  // the bucket is read afresh in every step
    const ulonglong2 c = *(const ulonglong2*)&keys[2 * (size_t)b];
    if (c.x == key) return (int)(2 * b);
    if (c.y == key) return (int)(2 * b + 1);
    if (c.x == kPartEmpty || c.y == kPartEmpty) {
      if (__hip_atomic_load(n_entries, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= limit) return -1;
      const uint32_t e = 2 * b + (c.x == kPartEmpty ? 0u : 1u);
      const unsigned long long old = atomicCAS(&keys[e], kPartEmpty, (unsigned long long)key);
      if (old == kPartEmpty) { inserted = true; return (int)e; }
      if (old == key) return (int)e;
      continue;  // somebody else took the entry: look at the bucket again
    }
    b = step == 0 ? __umulhi(m2, n_buckets) : (b + 1 == n_buckets ? 0u : b + 1);
    ++step;
  }
  return -1;
}

template <int KW>
__device__ __forceinline__ int part_find(unsigned long long* keys_lo, unsigned long long* keys_hi, uint32_t S, int limit,
                                         int* n_entries, uint64_t lo, uint64_t hi, bool& inserted) {
  if (KW == 1) return part_lds_entry2way(keys_lo, S / 2, limit, n_entries, lo, inserted);
  return part_lds_entry<KW>(keys_lo, keys_hi, S, limit, n_entries, lo, hi, part_hash<KW>(lo, hi), inserted);
}

template <int KW, int NV>
__device__ __forceinline__ void part_hbm_update(const AggParams& p, uint64_t lo, uint64_t hi, const RowValsT<NV>& rv) {
  if (KW == 1) {
    if (p.keys_only) { (void)global_entry(p, lo); return; }
    global_update(p, lo, rv);
    return;
  }
  const uint64_t e = global_entry2(p, lo, hi);
  if (e == ~0ull || p.keys_only) return;
  for (int s = 0; s < p.n_slots; ++s) {
    long long x;
    if (slot_input(p, s, rv, x)) atomic_apply(&p.g_state[(uint64_t)s * (p.capacity + 1) + e], p.slots[s].op, x);
  }
}

template <int KW, int NV, bool VALID>
__global__ __launch_bounds__(kPartBlock) void k_part_agg(AggParams p, PartParams q) {
  constexpr int W = KW + NV + (VALID ? 1 : 0);
  extern __shared__ __align__(16) unsigned long long part_lds[];
  __shared__ int n_entries;
  __shared__ uint32_t wg_pos, gave_up, n_rejects, next_item;
  __shared__ unsigned long long out_base;
  const uint32_t S = q.lds_entries & ~1u, tid = threadIdx.x, lane = tid & 63u;
  const int n_lds_slots = p.keys_only ? 0 : p.n_slots;
  const bool count_only = n_lds_slots == 1 && p.slots[0].op == kSlotAddOne;  // COUNT(*) and nothing else
  unsigned long long* keys_lo = part_lds;
  unsigned long long* keys_hi = part_lds + (KW == 2 ? S : 0);
  long long* state = (long long*)(part_lds + (size_t)KW * S);  // [n_lds_slots][S]
  const uint32_t n_items = *q.n_items;
  for (;;) {
    // Items are handed out in order through a counter (long ones first).  Thread 0 takes the
    // next one and looks whether the HBM table has overflowed (then the whole pass will be run
    // again), so that the workgroup leaves as a whole.
    if (tid == 0) {
      next_item = atomicAdd(q.queue, 1u);
      gave_up = table_gave_up(p) ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t item = next_item;
    if (gave_up || item >= n_items) return;
    const PartItem it = q.items[item];
    for (uint32_t i = tid; i < S; i += kPartBlock) {
      keys_lo[i] = kPartEmpty;
      if (KW == 2) keys_hi[i] = kPartEmpty;
      for (int s = 0; s < n_lds_slots; ++s) state[(size_t)s * S + i] = slot_identity(p.slots[s].op);
    }
    if (tid == 0) { n_entries = 0; wg_pos = 0; n_rejects = 0; }
    __syncthreads();
    // Software pipelined: the records of the next step are requested before those of this
    // step are looked up (all waves of the workgroup start an item together, so without this
    // they would also wait for memory together).
    constexpr uint64_t kStep = (uint64_t)kPartAggInFlight * kPartBlock;
    uint64_t next[kPartAggInFlight][W];
    auto request = [&](uint64_t base) {
#pragma unroll
      for (int u = 0; u < kPartAggInFlight; ++u) {
        const uint64_t i = base + (uint64_t)u * kPartBlock + tid;
        const uint64_t ic = i < it.hi ? i : it.lo;
#pragma unroll
        for (int w = 0; w < W; ++w) next[u][w] = __builtin_nontemporal_load((PartWordPtr)(uint64_t)(q.rec[w] + ic));
      }
    };
    request(it.lo);
    for (uint64_t base = it.lo; base < it.hi; base += kStep) {
      uint64_t rec[kPartAggInFlight][W];
      bool in[kPartAggInFlight];
#pragma unroll
      for (int u = 0; u < kPartAggInFlight; ++u) {
        in[u] = base + (uint64_t)u * kPartBlock + tid < it.hi;
#pragma unroll
        for (int w = 0; w < W; ++w) rec[u][w] = next[u][w];
      }
      if (base + kStep < it.hi) request(base + kStep);
      // One-word keys: the two home buckets of all the step's rows are read before any of
      // them is looked at, so the LDS latencies overlap instead of adding up row after row (a
      // key that another row of the step inserts meanwhile is simply not seen here: the row
      // then takes the search-and-insert path, which reads again).
      int found[kPartAggInFlight], vacant[kPartAggInFlight];
      if (KW == 1) {
        ulonglong2 c1[kPartAggInFlight], c2[kPartAggInFlight];
        uint32_t hb1[kPartAggInFlight], hb2[kPartAggInFlight];
        const uint32_t nb = S / 2;
#pragma unroll
        for (int u = 0; u < kPartAggInFlight; ++u) {
          const uint64_t lo = rec[u][0];
          const uint32_t m1 = ((uint32_t)lo ^ (uint32_t)(lo >> 32)) * 0x9E3779B1u;
          const uint32_t m2 = (m1 ^ (m1 >> 15)) * 0x85EBCA6Bu;
          hb1[u] = __umulhi(m1, nb);
          hb2[u] = __umulhi(m2, nb);
          c1[u] = *(const ulonglong2*)&keys_lo[2 * (size_t)hb1[u]];
          c2[u] = *(const ulonglong2*)&keys_lo[2 * (size_t)hb2[u]];
        }
#pragma unroll
        for (int u = 0; u < kPartAggInFlight; ++u) {
          const uint64_t lo = rec[u][0];
          int f = -1;
          f = c2[u].y == lo ? (int)(2 * hb2[u] + 1) : f;
          f = c2[u].x == lo ? (int)(2 * hb2[u]) : f;
          f = c1[u].y == lo ? (int)(2 * hb1[u] + 1) : f;
          f = c1[u].x == lo ? (int)(2 * hb1[u]) : f;
          found[u] = f;
          // the first free entry on the key's path, as far as the two home buckets go
          int v = -1;
          v = c2[u].y == kPartEmpty ? (int)(2 * hb2[u] + 1) : v;
          v = c2[u].x == kPartEmpty ? (int)(2 * hb2[u]) : v;
          v = c1[u].y == kPartEmpty ? (int)(2 * hb1[u] + 1) : v;
          v = c1[u].x == kPartEmpty ? (int)(2 * hb1[u]) : v;
          vacant[u] = v;
        }
      }
      auto row_values = [&](int u) {
        RowValsT<NV> rv;
        rv.valid = VALID ? (uint32_t)rec[u][W - 1] : 0xffffffffu;
#pragma unroll
        for (int v = 0; v < NV; ++v) rv.v[v] = (long long)rec[u][KW + v];
        return rv;
      };
      auto apply_row = [&](int u, int e) {
        if (count_only) {
          atomicAdd((unsigned long long*)&state[(uint32_t)e], 1ull);
        } else {
          const RowValsT<NV> rv = row_values(u);
          for (int s = 0; s < n_lds_slots; ++s) {
            long long x;
            if (slot_input(p, s, rv, x)) atomic_apply(&state[(size_t)s * S + (uint32_t)e], p.slots[s].op, x);
          }
        }
      };
      int e[kPartAggInFlight];
      bool pend[kPartAggInFlight];
      uint32_t n_new = 0;
#pragma unroll
      for (int u = 0; u < kPartAggInFlight; ++u) {
        const uint64_t lo = rec[u][0];
        // (a one-word key of all ones is the table's empty marker: it lives in the HBM table)
        bool search = in[u] && lo != kPartEmpty;
        bool ins = false;
        e[u] = -1;
        if (KW == 1) {
          // Most rows find their key in one of its two home buckets (read above, compared
          // without a branch).  A new key, as a rule, finds a free entry there: one attempt to
          // claim it.  The entry was free when the buckets were read and the ones before it
          // on the path were taken, and entries only ever fill up: if the claim succeeds, or
          // meets the same key, the entry is the key's.  Anything else (another key got there
          // first, no free entry) is for the probe loop.
          if (search && found[u] >= 0) { e[u] = found[u]; search = false; }
          if (search && vacant[u] >= 0) {
            const unsigned long long old = atomicCAS(&keys_lo[vacant[u]], kPartEmpty, (unsigned long long)lo);
            if (old == kPartEmpty) { ins = true; e[u] = vacant[u]; search = false; }
            else if (old == lo) { e[u] = vacant[u]; search = false; }
          }
        }
        pend[u] = search;
        n_new += (uint32_t)__popcll(__ballot(ins));
      }
#pragma unroll
      for (int u = 0; u < kPartAggInFlight; ++u) {
        if (!__ballot(pend[u])) continue;
        bool ins = false;
        if (pend[u]) e[u] = part_find<KW>(keys_lo, keys_hi, S, (int)q.lds_limit, &n_entries, rec[u][0], KW == 2 ? rec[u][KW - 1] : 0ull, ins);
        n_new += (uint32_t)__popcll(__ballot(ins));
      }
      if (n_new && lane == 0) atomicAdd(&n_entries, (int)n_new);
      bool any_rejected = false;
#pragma unroll
      for (int u = 0; u < kPartAggInFlight; ++u) any_rejected = any_rejected || (in[u] && e[u] < 0);
      if (__ballot(any_rejected)) {
#pragma unroll
        for (int u = 0; u < kPartAggInFlight; ++u) {
          const bool rejected = in[u] && e[u] < 0;
          const uint64_t m_rej = __ballot(rejected);
          if (!m_rej) continue;
          if ((int)lane == __ffsll((long long)m_rej) - 1) atomicAdd(&n_rejects, (uint32_t)__popcll(m_rej));
          if (rejected) part_hbm_update<KW, NV>(p, rec[u][0], KW == 2 ? rec[u][KW - 1] : 0ull, row_values(u));
        }
      }
#pragma unroll
      for (int u = 0; u < kPartAggInFlight; ++u) {
        bool todo = in[u] && e[u] >= 0;
        if (count_only) {
          // a frequent key: the lanes that hold the entry of the first one are counted and
          // added once (LDS atomics of a wave on one address are carried out one by one)
          const uint64_t m_todo = __ballot(todo);
          if (m_todo) {
            const int leader = __ffsll((long long)m_todo) - 1;
            const int e0 = __shfl(e[u], leader);
            const uint64_t m_same = __ballot(todo && e[u] == e0);
            if (__popcll(m_same) >= 8) {
              if ((int)lane == leader) atomicAdd((unsigned long long*)&state[(uint32_t)e0], (unsigned long long)__popcll(m_same));
              todo = todo && e[u] != e0;
            }
          }
        }
        if (todo) apply_row(u, e[u]);
      }
    }
    __syncthreads();
    const uint32_t n = (uint32_t)n_entries;
    // Rows that found the table full went to the HBM table.  Whether a key is turned away
    // depends on when its row looks (the fill count moves while other waves insert), so a key
    // may sit in both tables: a bucket that turned rows away merges all its groups into the
    // HBM table instead of appending them to the list.
    const bool to_list = it.exclusive && n_rejects == 0;
    if (to_list) {
      // the bucket's groups are nobody else's: append them to the list
      if (tid == 0) out_base = atomicAdd(q.list_cursor, (unsigned long long)n);
      __syncthreads();
      const uint64_t first = out_base;
      const bool fits = first + n <= q.list_capacity;
      if (!fits && tid == 0) q.flags[0] = 1u;
      const uint64_t stride = q.list_capacity + 1;
      for (uint32_t i0 = 0; i0 < S && fits; i0 += kPartBlock) {
        const uint32_t i = i0 + tid;
        const unsigned long long k = i < S ? keys_lo[i] : kPartEmpty;
        const bool occ = k != kPartEmpty;
        const uint64_t m = __ballot(occ);
        uint32_t wave_first = 0;
        if (lane == 0 && m) wave_first = atomicAdd(&wg_pos, (uint32_t)__popcll(m));
        wave_first = __shfl(wave_first, 0);
        if (!occ) continue;
        const uint64_t pos = first + wave_first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        q.list_keys[pos] = k;
        if (KW == 2) q.list_keys_hi[pos] = keys_hi[i];
        for (int s = 0; s < n_lds_slots; ++s) q.list_state[(uint64_t)s * stride + pos] = state[(size_t)s * S + i];
      }
    } else {
      // a slice of a large bucket: other slices hold the same keys, merge through the HBM table
      for (uint32_t i = tid; i < S; i += kPartBlock) {
        const unsigned long long k = keys_lo[i];
        if (k == kPartEmpty) continue;
        const uint64_t e = KW == 2 ? global_entry2(p, k, keys_hi[i]) : global_entry(p, k);
        if (e == ~0ull) continue;
        for (int s = 0; s < n_lds_slots; ++s) {
          const int op = p.slots[s].op;
          const long long v = state[(size_t)s * S + i];
          if (v != slot_identity(op)) atomic_apply(&p.g_state[(uint64_t)s * (p.capacity + 1) + e], op, v);
        }
      }
      if (tid == 0) atomicAdd(&q.flags[1], n + n_rejects);
    }
    __syncthreads();
  }
}

// (Tried and dropped for rows that arrive clustered by bucket -- time-ordered rows partitioned
// by a group key that holds the time bucket: no tile at all, one output cursor per bucket in
// LDS, a row takes its place with one LDS atomic and is stored there straight from the
// registers, in the hope that the stores of neighbouring places meet in the L2.  No barriers,
// but 6.7 against 4.9 ms at 4e8 rows and 21.6 against 15.1 ms at 1.25e9.)
using PartKernel = void (*)(AggParams, PartParams);

struct PartKernels {
  PartKernel hist, scatter, agg;
  int words;
  PartKernel scatter_half;  // the scatter pass for workgroups of 512 threads (records of one or two words), or null
};

constexpr int kHalfBlock = 512;

template <int KW, int NV, bool VALID, int NK>
PartKernel part_scatter_half() {
  constexpr int W = KW + NV + (VALID ? 1 : 0);
  if constexpr (W <= 2) return k_part_scatter<KW, NV, VALID, NK, kHalfBlock>;
  else return nullptr;
}

template <int KW, int NV, bool VALID, int NK>
PartKernels part_kernels_ahead() {
  constexpr int W = KW + NV + (VALID ? 1 : 0);
  if constexpr (VALID) return {k_part_hist<KW, NK>, k_part_scatter<KW, NV, true, 0, (int)kPartBlock>, k_part_agg<KW, NV, true>, W, part_scatter_half<KW, NV, true, 0>()};
  else return {k_part_hist<KW, NK>, k_part_scatter<KW, NV, false, NK, (int)kPartBlock>, k_part_agg<KW, NV, false>, W, part_scatter_half<KW, NV, false, NK>()};
}

// nk: key columns the scan kernels request a super-tile ahead (0: none; the kernels exist
// for one column per key word, and for two columns in one key word).
template <int KW, int NV, bool VALID>
PartKernels part_kernels(int nk) {
  constexpr int W = KW + NV + (VALID ? 1 : 0);
  if (nk == KW) return part_kernels_ahead<KW, NV, VALID, KW>();
  // (two key words and three columns: the kernels run out of registers -- the scatter pass
  // with its three columns requested ahead spills 60 to 80 registers and is slower than the
  // one that waits for its loads, 16.3 against 14.1 ms on the benchmark's COUNT(DISTINCT))
  if constexpr (W >= 2 && KW == 1) {
    if (nk == KW + 1) return part_kernels_ahead<KW, NV, VALID, KW + 1>();
  }
  return {k_part_hist<KW, 0>, k_part_scatter<KW, NV, VALID, 0, (int)kPartBlock>, k_part_agg<KW, NV, VALID>, W, part_scatter_half<KW, NV, VALID, 0>()};
}

bool pick_part_kernels(int kw, int nv, bool valid, int ahead, PartKernels* out) {
  if (valid && nv == 0) valid = false;
  const int code = kw * 100 + nv * 10 + (valid ? 1 : 0);
  switch (code) {
    case 100: *out = part_kernels<1, 0, false>(ahead); return true;
    case 110: *out = part_kernels<1, 1, false>(ahead); return true;
    case 111: *out = part_kernels<1, 1, true>(ahead); return true;
    case 120: *out = part_kernels<1, 2, false>(ahead); return true;
    case 121: *out = part_kernels<1, 2, true>(ahead); return true;
    case 130: *out = part_kernels<1, 3, false>(ahead); return true;
    case 200: *out = part_kernels<2, 0, false>(ahead); return true;
    case 210: *out = part_kernels<2, 1, false>(ahead); return true;
    case 211: *out = part_kernels<2, 1, true>(ahead); return true;
    case 220: *out = part_kernels<2, 2, false>(ahead); return true;
    default: return false;
  }
}

constexpr size_t kPartLdsBytes = 160 * 1024 - 1024;  // dynamic LDS of the scatter and aggregation kernels' workgroups (one per CU)

void allow_lds(PartKernel k, size_t bytes) {
  if (bytes > 48 * 1024) PRISM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

}  // namespace

int part_record_words(const AggParams& p, bool vals_nullable) {
  PartKernels k;
  if (!pick_part_kernels(p.key_words, p.keys_only ? 0 : p.n_vals, vals_nullable, 0, &k)) return 0;
  return k.words;
}

uint32_t part_max_buckets(int words) {
  // scatter workgroup: W * R * 1024 records of 8 bytes, 2 * (P + 2) + P counters of 4 bytes
  const size_t tile = (size_t)words * (words == 1 ? 16 : (words == 2 ? 8 : 4)) * kPartBlock * 8;
  size_t p = 64;
  while (p * 2 <= 4 * kPartBlock && tile + (3 * (p * 2) + 4) * 4 <= kPartLdsBytes) p *= 2;
  return (uint32_t)p;
}

uint32_t part_lds_entries(const AggParams& p) {
  const size_t per_entry = 8 * ((size_t)p.key_words + (p.keys_only ? 0 : (size_t)p.n_slots));
  return (uint32_t)(kPartLdsBytes / per_entry);
}

// Bytes of LDS a scatter workgroup of `block` threads takes for records of `words` words and P buckets.
static size_t part_scatter_lds(int words, uint32_t P, uint32_t block) {
  const int R = words == 1 ? 16 : (words == 2 ? 8 : 4);
  return (size_t)words * R * block * 8 + (size_t)(3 * P + 4) * 4;
}

bool part_scatter_halves(int words, uint32_t P) {
  // Two workgroups of 512 threads per CU: both tiles with their counters must fit the LDS,
  // and the records must be two words wide.  With one-word records the runs that leave a
  // tile are so short already (8 to 32 rows of 8 bytes) that halving them costs more than
  // the overlap of the two workgroups gains (C3 with 512 buckets: 0.52 -> 0.61 ms); with two
  // words the pass got 9 % faster.  (The kernels exist for both widths.)
  return words == 2 && 2 * (part_scatter_lds(words, P, kHalfBlock) + 256) <= kPartLdsBytes;
}

uint32_t part_scan_workgroups(prism_gpu_ctx* ctx, const AggParams& p, uint32_t* tiles_per_wg, bool two_per_cu) {
  // whole scan tiles per workgroup, a multiple of 4 (the widest super-tile); one workgroup per CU, or two
  uint64_t per = ceil_div(p.sd.n_tiles, (uint64_t)ctx->n_cus * (two_per_cu ? 2 : 1));
  per = std::max<uint64_t>(4, ceil_div(per, 4) * 4);  // (an empty scan still gets one workgroup)
  *tiles_per_wg = (uint32_t)per;
  return (uint32_t)std::max<uint64_t>(1, ceil_div(p.sd.n_tiles, per));
}

void launch_partitioned_group_by(prism_gpu_ctx* ctx, const AggParams& p_scan, const PartParams& q, bool vals_nullable,
                                 uint64_t hist_bytes, uint64_t scatter_bytes, uint64_t agg_bytes, bool scatter_only) {
  const AggParams& p_table = p_scan;  // the same descriptor carries the scan and the HBM table the stragglers go to
  PartKernels k;
  if (!pick_part_kernels(p_scan.key_words, p_scan.keys_only ? 0 : p_scan.n_vals, vals_nullable, part_keys_ahead(p_scan), &k))
    fail(PRISM_GPU_EINTERNAL, "no partitioned GROUP BY kernel for this plan shape");
  const uint32_t P = q.n_buckets;
  if (P < 64 || P > part_max_buckets(k.words) || (P & (P - 1)) != 0 || (1u << q.p_bits) != P)
    fail(PRISM_GPU_EINTERNAL, "bad bucket count for the partitioned GROUP BY");
  if (q.n_wg == 0 || (uint64_t)q.n_wg * q.tiles_per_wg < p_scan.sd.n_tiles || q.tiles_per_wg % 4 != 0)
    fail(PRISM_GPU_EINTERNAL, "bad work split for the partitioned GROUP BY");
  const bool halves = q.half_blocks != 0;
  if (halves && (!k.scatter_half || !part_scatter_halves(k.words, P))) fail(PRISM_GPU_EINTERNAL, "no half-size scatter workgroups for this shape");
  const uint32_t scatter_block = halves ? kHalfBlock : kPartBlock;
  const size_t scatter_lds = part_scatter_lds(k.words, P, scatter_block);
  const size_t agg_lds = (size_t)q.lds_entries * 8 * ((size_t)p_scan.key_words + (p_scan.keys_only ? 0 : (size_t)p_scan.n_slots));
  if (scatter_lds > kPartLdsBytes || agg_lds > kPartLdsBytes) fail(PRISM_GPU_EINTERNAL, "partitioned GROUP BY exceeds LDS");
  allow_lds(halves ? k.scatter_half : k.scatter, scatter_lds);
  allow_lds(k.agg, agg_lds);
  {
    KernelTimer timer(ctx, "part_hist", hist_bytes, true);
    if (q.by_hi && p_scan.key_words == 2) {
      // The buckets follow from the high word alone: the histogram runs as a one-word plan over
      // that word's columns (its packed value is the group key), with their loads requested
      // ahead when they are at most two and not nullable; the columns of the low word are not read.
      AggParams p_hist = p_scan;
      p_hist.n_keys = 0;
      bool nullable = false;
      for (int i = 0; i < p_scan.n_keys; ++i) {
        if (!p_scan.keys[i].word) continue;
        p_hist.keys[p_hist.n_keys++] = p_scan.keys[i];
        nullable = nullable || p_scan.keys[i].null_bit >= 0;
      }
      p_hist.key_words = 1;
      PartKernel hist = (nullable || p_hist.n_keys > 2 || p_hist.n_keys == 0) ? k_part_hist<1, 0>
                        : p_hist.n_keys == 2 ? k_part_hist<1, 2> : k_part_hist<1, 1>;
      hist<<<dim3(q.n_wg), dim3(kPartBlock), (size_t)P * 4, ctx->stream>>>(p_hist, q);
    } else {
      k.hist<<<dim3(q.n_wg), dim3(kPartBlock), (size_t)P * 4, ctx->stream>>>(p_scan, q);
    }
    PRISM_HIP(hipGetLastError());
  }
  {
    KernelTimer timer(ctx, "part_plan", 0, true);
    k_part_offsets<<<dim3((uint32_t)ceil_div(P, kBlock)), dim3(kBlock), 0, ctx->stream>>>(q);
    PRISM_HIP(hipGetLastError());
    k_part_plan<<<dim3(1), dim3(kPartBlock), 0, ctx->stream>>>(q);  // (the bucket starts; the work items go unused when scatter_only)
    PRISM_HIP(hipGetLastError());
  }
  {
    KernelTimer timer(ctx, "part_scatter", scatter_bytes, true);
    (halves ? k.scatter_half : k.scatter)<<<dim3(q.n_wg), dim3(scatter_block), scatter_lds, ctx->stream>>>(p_scan, q);
    PRISM_HIP(hipGetLastError());
  }
  if (scatter_only) return;
  {
    KernelTimer timer(ctx, "part_agg", agg_bytes, true);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(q.max_items, (uint64_t)ctx->n_cus);
    k.agg<<<dim3(grid), dim3(kPartBlock), agg_lds, ctx->stream>>>(p_table, q);
    PRISM_HIP(hipGetLastError());
  }
}

}  // namespace prism
