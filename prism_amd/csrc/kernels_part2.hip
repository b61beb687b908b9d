// Second level of the partitioned COUNT(DISTINCT x) GROUP BY k (two-word (x, k) pairs).
//
// The first level (kernels_part.hip, bucket = hash of k) leaves the pairs sorted into coarse
// buckets: all pairs of a group share one.  Here every coarse bucket is sorted once more by
// a hash of the whole pair, into fine buckets small enough for LDS; a fine bucket is
// deduplicated in an LDS table and its distinct pairs are counted per group -- a fine
// bucket sees only the few groups of its coarse bucket -- into the outer table.  The set of
// pairs (hundreds of millions of 16-byte keys for the benchmark query) is never built in
// HBM: that set was what made the one-level version bound by random cache-line fetches.
//
//   k_part2_hist     rows per fine bucket (slices of the record array; LDS counters for the
//                    coarse buckets a slice touches, flushed with one atomic per counter)
//   k_part2_scan     exclusive scan of the counts (one workgroup)
//   k_part2_scatter  the same slices again, tile by tile: ranks inside the tile from LDS
//                    counters, one reservation per (tile, fine bucket) in the bucket's
//                    cursor, records written to their places
//   k_part2_agg      fine buckets handed out through a counter: LDS set, then per-group counts
//
// Algorithmic bytes per row: 16 read (hist) + 16 read + 16 written (scatter) + 16 read (agg).
#include <algorithm>
#include <cstring>

#include "agg_device.hpp"
#include "context.hpp"
#include "device_utils.hpp"
#include "kernels.hpp"
#include "part_device.hpp"

namespace prism {

namespace {

constexpr uint32_t kP2Block = 1024;
constexpr uint32_t kP2Counters = 2048;   // LDS counters of a slice: (coarse buckets covered) x 2^p2_bits
constexpr uint32_t kP2MaxSpan = 8;       // coarse buckets covered at most
constexpr uint32_t kP2TileRows = 8;      // rows per thread and tile of the scatter pass
constexpr unsigned long long kP2Empty = ~0ull;
constexpr uint32_t kP2Groups = 512;      // entries of the per-bucket group counters in LDS
constexpr uint32_t kP2AggBlock = 512;    // threads of the last pass; two of its workgroups share a CU, so that
constexpr uint32_t kP2AggPerCu = 2;      // one's waits for HBM overlap with the other's work in LDS
constexpr size_t kP2LdsBytes = 160 * 1024 - 1024;

typedef const __attribute__((address_space(1))) unsigned long long* P2WordPtr;

__device__ __forceinline__ uint32_t coarse_of(const Part2Params& q, uint64_t hi) {
  if (q.direct) return (uint32_t)(hi >> (q.hi_bits - (int)q.p1_bits));
  if (q.compact) return (uint32_t)(bij_mix(hi, q.hi_bits) >> (q.hi_bits - (int)q.p1_bits));
  return q.p1_bits ? (uint32_t)(part_hash_hi(hi) >> (64 - q.p1_bits)) : 0u;
}
__device__ __forceinline__ uint32_t span_of(const Part2Params& q, uint32_t counters = kP2Counters) {
  const uint32_t s = counters >> q.p2_bits;
  return s < kP2MaxSpan ? s : kP2MaxSpan;
}
// Compact form: the low word times an odd constant (a bijection modulo 2^lo_bits whose top
// bits depend on every bit of the word), its upper 32 bits then XORed with a hash of the high
// word -- still a bijection of the low word for any one high word, which is all a record
// needs, and a value of x that is frequent in many groups lands in a different sub-bucket
// for each of them.  (Without the XOR the rows of the most frequent x met in ONE sub-bucket
// of every coarse bucket, six times the average size, and the workgroups that those fell to
// made the last pass six times as long.)  One 64-bit multiply and two 32-bit ones; the first
// version (two 64-bit multiplies and a 64-bit hash, computed in the scatter pass twice)
// spent more instructions on mixing than on everything else.
__device__ __forceinline__ uint64_t mixed_lo(const Part2Params& q, uint64_t lo, uint64_t hi) {
  const uint64_t t = (lo * kBijMul1) & ((1ull << q.lo_bits) - 1ull);
  uint32_t h = ((uint32_t)hi ^ (uint32_t)(hi >> 32) * 0x85EBCA6Bu) * 0x7FEB352Du;
  h = (h ^ (h >> 15)) * 0x846CA68Bu;
  h ^= h >> 16;
  const uint64_t mask = q.lo_bits >= 32 ? (uint64_t)h << (q.lo_bits - 32) : (uint64_t)(h >> (32 - q.lo_bits));
  return t ^ mask;
}
__device__ __forceinline__ uint32_t sub_of(const Part2Params& q, uint64_t lo, uint64_t hi) {
  if (q.compact) return (uint32_t)(mixed_lo(q, lo, hi) >> (q.lo_bits - (int)q.p2_bits));
  return q.p2_bits ? (uint32_t)(hash64(lo ^ hash64(hi)) >> (64 - q.p2_bits)) : 0u;
}
__global__ __launch_bounds__(kP2Block) void k_part2_hist(Part2Params q) {
  __shared__ uint32_t cnt[kP2Counters];
  const uint32_t n_sub = 1u << q.p2_bits, span = span_of(q);
  for (uint32_t i = threadIdx.x; i < span * n_sub; i += kP2Block) cnt[i] = 0;
  const uint64_t r0 = (uint64_t)blockIdx.x * q.slice_rows;
  const uint64_t r1 = r0 + q.slice_rows < q.n_rows ? r0 + q.slice_rows : q.n_rows;
  // the slice's first coarse bucket (the records are sorted by it)
  const uint32_t c0 = __builtin_amdgcn_readfirstlane(coarse_of(q, q.in_hi[r0]));
  __syncthreads();
  for (uint64_t i = r0 + threadIdx.x; i < r1; i += kP2Block) {
    const uint64_t lo = __builtin_nontemporal_load((P2WordPtr)(uint64_t)(q.in_lo + i));
    const uint64_t hi = __builtin_nontemporal_load((P2WordPtr)(uint64_t)(q.in_hi + i));
    const uint32_t c = coarse_of(q, hi), s = sub_of(q, lo, hi);
    if (c - c0 < span) atomicAdd(&cnt[(c - c0) * n_sub + s], 1u);
    else atomicAdd(&q.fine_count[((uint64_t)c << q.p2_bits) | s], 1u);  // a slice over many small coarse buckets
  }
  __syncthreads();
  const uint32_t n_coarse = 1u << q.p1_bits;
  for (uint32_t i = threadIdx.x; i < span * n_sub; i += kP2Block) {
    const uint32_t c = c0 + i / n_sub;
    if (cnt[i] && c < n_coarse) atomicAdd(&q.fine_count[((uint64_t)c << q.p2_bits) | (i % n_sub)], cnt[i]);
  }
}

// The pairs of two scan tiles a trip, packed from the scan's columns.  NK > 0 (the plan's key
// columns carry no validity bitmaps: part_keys_ahead): the column words of the next trip are
// requested before this trip's are used, so the loads are in flight during the trip's LDS work.
template <int NK, int BLOCK = (int)kP2Block>
struct P2Columns {
  static constexpr bool kAhead = NK > 0;
  static constexpr int NA = NK > 0 ? NK : 1, R = (int)kP2TileRows;
  static constexpr int TPT = BLOCK * R / (int)kTileRows;  // scan tiles a trip
  unsigned long long raw[NA][R];
  uint32_t live_next;
  PartCursor<NA> cur;
  PartSegPos pos;
  __device__ __forceinline__ void init(const AggParams& p, uint64_t T0, uint64_t T1, uint32_t tid) {
    live_next = 0;
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
      for (int r = 0; r < R; ++r) raw[k][r] = 0;
    if (T0 >= T1) return;
    if constexpr (kAhead) {
      part_cursor_init<NA, 0>(p, cur, T0);
      part_issue<NA, 0, R, BLOCK>(p, cur, T0, T1, tid, raw, live_next);
    } else {
      part_seg_init(p, pos, T0);
    }
  }
  __device__ __forceinline__ void next(const AggParams& p, uint64_t T, uint64_t T1, uint32_t tid, uint64_t (&lo)[R], uint64_t (&hi)[R],
                                       uint32_t& live) {
    if constexpr (kAhead) {
      uint64_t rec[2][R];
      part_finish<2, NA, 0, R>(p, raw, rec);
#pragma unroll
      for (int r = 0; r < R; ++r) { lo[r] = rec[0][r]; hi[r] = rec[1][r]; }
      live = live_next;
      if (T + TPT < T1) part_issue<NA, 0, R, BLOCK>(p, cur, T + TPT, T1, tid, raw, live_next);
    } else {
      PartTile tl[TPT];
#pragma unroll
      for (int j = 0; j < TPT; ++j) tl[j] = part_tile_next(p, pos, T + j, T1);
      uint32_t rowc[R];
      part_load_keys<2, R, BLOCK>(p, tl, tid, lo, hi, rowc, live);
    }
  }
};

// The same counts taken from the scan's columns (Part2Params::direct): a workgroup walks
// slice_tiles scan tiles, two at a time.  The LDS counters start at the coarse bucket of the
// slice's first rows (rows in time order move on by at most one bucket within a slice).
template <int NK>
__global__ __launch_bounds__(kP2Block) void k_part2_hist_direct(Part2Params q, AggParams p) {
  __shared__ uint32_t cnt[kP2Counters];
  __shared__ uint32_t c_first, n_stray;
  const uint32_t n_sub = 1u << q.p2_bits, span = span_of(q), tid = threadIdx.x;
  for (uint32_t i = tid; i < span * n_sub; i += kP2Block) cnt[i] = 0;
  if (tid == 0) { c_first = ~0u; n_stray = 0; }
  const uint64_t T0 = (uint64_t)blockIdx.x * q.slice_tiles;
  const uint64_t T1 = T0 + q.slice_tiles < q.n_tiles ? T0 + q.slice_tiles : q.n_tiles;
  P2Columns<NK> in;
  in.init(p, T0, T1, tid);
  __syncthreads();
  uint32_t c0 = 0;
  uint32_t strays = 0;
  for (uint64_t T = T0; T < T1; T += 2) {
    uint64_t lo[kP2TileRows], hi[kP2TileRows];
    uint32_t live;
    in.next(p, T, T1, tid, lo, hi, live);
    if (T == T0) {
      // (the smallest coarse bucket among the first rows)
      uint32_t mine = ~0u;
#pragma unroll
      for (uint32_t r = 0; r < kP2TileRows; ++r)
        if ((live >> r) & 1u) { const uint32_t c = coarse_of(q, hi[r]); mine = c < mine ? c : mine; }
      if (mine != ~0u) atomicMin(&c_first, mine);
      __syncthreads();
      c0 = c_first == ~0u ? 0u : c_first;
    }
#pragma unroll
    for (uint32_t r = 0; r < kP2TileRows; ++r) {
      if (!((live >> r) & 1u)) continue;
      const uint32_t c = coarse_of(q, hi[r]), s = sub_of(q, lo[r], hi[r]);
      if (c - c0 < span) atomicAdd(&cnt[(c - c0) * n_sub + s], 1u);
      else { atomicAdd(&q.fine_count[((uint64_t)c << q.p2_bits) | s], 1u); ++strays; }
    }
  }
  if (strays) atomicAdd(&n_stray, strays);
  __syncthreads();
  const uint32_t n_coarse = 1u << q.p1_bits;
  for (uint32_t i = tid; i < span * n_sub; i += kP2Block) {
    const uint32_t c = c0 + i / n_sub;
    if (cnt[i] && c < n_coarse) atomicAdd(&q.fine_count[((uint64_t)c << q.p2_bits) | (i % n_sub)], cnt[i]);
  }
  if (tid == 0 && n_stray && q.strays) atomicAdd(q.strays, (unsigned long long)n_stray);
}

// counts -> exclusive offsets (and the cursors the scatter pass advances); one workgroup
__global__ __launch_bounds__(1024) void k_part2_scan(Part2Params q, uint32_t n_fine) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_fine; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_fine ? q.fine_count[i] : 0;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off, 64);
      if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wave_tot[w];
    const uint32_t c = carry;
    if (i < n_fine) {
      q.fine_base[i] = c + before + incl - v;
      q.fine_cursor[i] = c + before + incl - v;
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) q.fine_base[n_fine] = carry;
}

// The second level without its histogram: every fine bucket gets a region of fixed size --
// twice its even share of the coarse bucket's rows, plus a margin (one frequent pair can be
// half a fine bucket's rows) -- instead of its exact place.  The bucket numbers are top bits of mixed words, so the fine buckets of a
// coarse bucket are close to even unless few distinct values repeat a lot; a bucket that
// outgrows its region sets flags[1], its surplus rows are dropped, and the caller runs the
// level again with the histogram.  Saves a pass over the records (16 bytes a row).
constexpr uint32_t kP2Margin = 48;
__global__ __launch_bounds__(256) void k_part2_layout(Part2Params q, uint32_t n_fine) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f > n_fine) return;
  const uint32_t n_sub = 1u << q.p2_bits;
  auto coarse_start = [&](uint32_t c) { return 2 * q.coarse_base[c] + (uint64_t)c * kP2Margin * n_sub; };
  if (f == n_fine) { q.fine_base[f] = (uint32_t)coarse_start(1u << q.p1_bits); return; }
  const uint32_t c = f >> q.p2_bits, sub = f & (n_sub - 1);
  const uint64_t rows = q.coarse_base[c + 1] - q.coarse_base[c];
  const uint64_t cap = (((2 * rows) >> q.p2_bits) + kP2Margin) & ~15ull;
  const uint32_t at = (uint32_t)(coarse_start(c) + sub * cap);
  q.fine_base[f] = at;
  q.fine_cursor[f] = at;
  q.fine_limit[f] = (uint32_t)(at + cap);
}

// Regions of one size for the first n_regions fine buckets (Part2Params::spill_rec).
__global__ __launch_bounds__(256) void k_part2_layout_uniform(Part2Params q, uint32_t n_fine) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f > n_fine) return;
  const uint32_t r = f < q.n_regions ? f : q.n_regions;
  const uint32_t at = r * q.region_cap;
  q.fine_base[f] = at;
  if (f == n_fine) return;
  q.fine_cursor[f] = at;
  q.fine_limit[f] = f < q.n_regions ? at + q.region_cap : at;
}

// The spill list sorted by fine bucket: counts, a scan (k_part2_scan), places.  Neighbouring
// records mostly share a bucket (a tile spills its buckets' surplus run by run): a wave takes 64
// neighbouring records, finds the runs with a ballot, and the first lane of a run does the one
// atomic for all of it -- the loads and the stores stay coalesced.
__device__ __forceinline__ uint64_t spill_len(const Part2Params& q) {
  const unsigned long long n = *q.spill_cursor;
  return n < q.spill_capacity ? n : q.spill_capacity;
}
// lane's run among the wave's records: `head` lanes start one; returns the run's first lane and its length
__device__ __forceinline__ void spill_run(bool in, uint32_t f, uint32_t lane, uint32_t& first, uint32_t& len) {
  const uint32_t prev = __shfl_up(f, 1, 64);
  const bool head = in && (lane == 0 || f != prev);
  const uint64_t heads = __ballot(head), live = __ballot(in);
  const uint64_t upto = heads & ((lane == 63 ? 0ull : (2ull << lane)) - 1ull);  // heads at or below this lane
  first = upto ? 63u - (uint32_t)__builtin_clzll(upto) : 0u;
  const uint64_t above = first == 63 ? 0ull : (heads & ~((2ull << first) - 1ull));  // the next run's head
  const uint32_t end = above ? (uint32_t)__builtin_ctzll(above) : (uint32_t)__popcll(live);
  len = end - first;
}
__global__ __launch_bounds__(256) void k_spill_count(Part2Params q) {
  const uint64_t n = spill_len(q);
  const uint32_t lane = threadIdx.x & 63u;
  for (uint64_t i0 = (uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63u); i0 < n; i0 += (uint64_t)gridDim.x * 256) {
    const uint64_t i = i0 + lane;
    const bool in = i < n;
    const uint32_t f = in ? q.spill_fine[i] : ~0u;
    uint32_t first, len;
    spill_run(in, f, lane, first, len);
    if (in && lane == first && f != 0xffffffffu) atomicAdd(&q.spill_count[f], len);  // (all ones: an unused place of a workgroup's chunk)
  }
}
__global__ __launch_bounds__(256) void k_spill_place(Part2Params q, uint64_t out_offset) {
  const uint64_t n = spill_len(q);
  const uint32_t lane = threadIdx.x & 63u;
  for (uint64_t i0 = (uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63u); i0 < n; i0 += (uint64_t)gridDim.x * 256) {
    const uint64_t i = i0 + lane;
    const bool in = i < n;
    const uint32_t f = in ? q.spill_fine[i] : ~0u;
    const uint64_t rec = in ? q.spill_rec[i] : 0ull;
    uint32_t first, len;
    spill_run(in, f, lane, first, len);
    uint32_t at = 0;
    if (in && lane == first && f != 0xffffffffu) at = atomicAdd(&q.spill_cur[f], len);
    at = __shfl(at, (int)first, 64);
    if (in && f != 0xffffffffu) q.out_lo[out_offset + at + (lane - first)] = rec;
  }
}
// A fine bucket as two runs of out_lo: what its region took, and its part of the sorted spill list.
__global__ __launch_bounds__(256) void k_spill_runs(Part2Params q, uint64_t out_offset) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f >= q.n_regions) return;
  const uint32_t b = q.fine_base[f], end = q.fine_cursor[f], lim = q.fine_limit[f];
  q.seg_base_out[f] = b;
  q.seg_cnt_out[f] = (end < lim ? end : lim) - b;
  q.seg_base_out[q.n_regions + f] = (uint32_t)(out_offset + q.spill_base[f]);
  q.seg_cnt_out[q.n_regions + f] = q.spill_count[f];
}

// Exclusive scan of n (<= 2 * BLOCK, even) LDS counters in place, by a workgroup of BLOCK threads; a[n] receives the total.
template <uint32_t BLOCK = kP2Block>
__device__ __forceinline__ void p2_block_scan(uint32_t* a, uint32_t n, uint32_t* wave_tot) {
  const uint32_t t = threadIdx.x, i0 = 2 * t;
  const uint32_t v0 = i0 < n ? a[i0] : 0u, v1 = i0 + 1 < n ? a[i0 + 1] : 0u;
  uint32_t incl = v0 + v1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, 64);
    if ((int)(t & 63) >= off) incl += o;
  }
  if ((t & 63) == 63) wave_tot[t >> 6] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < (t >> 6); ++w) before += wave_tot[w];
  const uint32_t excl = before + incl - (v0 + v1);
  if (i0 < n) a[i0] = excl;
  if (i0 + 1 < n) a[i0 + 1] = excl + v0;
  if (t == BLOCK - 1) a[n] = before + incl;
  __syncthreads();
}

// The tile's records are put in fine-bucket order in LDS first and leave as runs: written
// straight from the registers, 8 bytes at a time to places 32 rows apart, the pass wrote 3.6
// times its payload (partial cache lines).
// (BLOCK: 1024 threads, tiles of 8192 records and 2048 counters -- one workgroup a CU; or, for the
// columns' form, 512 threads, 4096 records and 1024 counters -- two a CU, so that one's barriers and
// waits overlap with the other's work.)
template <bool DIRECT, int NK, uint32_t BLOCK>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_part2_scatter(Part2Params q, AggParams p) {
  constexpr uint32_t NC = 2 * BLOCK;  // LDS counters
  extern __shared__ __align__(16) unsigned long long p2s_lds[];
  constexpr uint32_t kTile = BLOCK * kP2TileRows;
  unsigned long long* s_lo = p2s_lds;
  unsigned long long* s_hi = p2s_lds + kTile;
  uint32_t* cnt = (uint32_t*)(p2s_lds + 2 * kTile);  // [NC]: rows of the tile per counter
  uint32_t* lstart = cnt + NC;                // [NC + 1]: where a counter's run starts in the tile
  uint32_t* base = lstart + NC + 1;           // [NC]: and in the output
  __shared__ uint32_t wave_tot[BLOCK / 64];
  __shared__ uint32_t c_first;
  __shared__ uint32_t n_stray;                 // (DIRECT) rows beyond the tile's LDS counters: rows out of time order
  __shared__ uint32_t spill_n;                 // records of the tile that go to the spill list
  __shared__ unsigned long long spill_at;      // and where the tile's part of the list starts
  const bool spills = DIRECT && q.spill_rec != nullptr;
  const uint32_t n_sub = 1u << q.p2_bits, span = span_of(q, NC), n_cnt = span * n_sub;
  const uint32_t n_coarse = 1u << q.p1_bits;
  // the workgroup's share: records [r0, r1) of the first level, or (DIRECT) scan tiles [T0, T1), two per trip
  uint64_t r0 = 0, r1 = 0, T0 = 0, T1 = 0, trips = 0;
  uint32_t strays = 0, c_slice = 0;
  if (DIRECT && threadIdx.x == 0) { n_stray = 0; c_first = ~0u; }
  P2Columns<DIRECT ? NK : 0, (int)BLOCK> in;
  constexpr uint64_t TPT = kTile / kTileRows;  // scan tiles a trip
  if (DIRECT) {
    T0 = (uint64_t)blockIdx.x * q.slice_tiles;
    T1 = T0 + q.slice_tiles < q.n_tiles ? T0 + q.slice_tiles : q.n_tiles;
    in.init(p, T0, T1, threadIdx.x);
    trips = T1 > T0 ? (T1 - T0 + TPT - 1) / TPT : 0;
  } else {
    r0 = (uint64_t)blockIdx.x * q.slice_rows;
    r1 = r0 + q.slice_rows < q.n_rows ? r0 + q.slice_rows : q.n_rows;
    trips = r1 > r0 ? (r1 - r0 + kTile - 1) / kTile : 0;
  }
  // every thread makes the same number of trips (the barriers are inside)
  for (uint64_t trip = 0; trip < trips; ++trip) {
    const uint64_t t0 = r0 + trip * kTile;
    for (uint32_t i = threadIdx.x; i < NC; i += BLOCK) cnt[i] = 0;
    if (DIRECT && threadIdx.x == 0) spill_n = 0;
    uint32_t c0 = 0;
    if (!DIRECT) c0 = __builtin_amdgcn_readfirstlane(coarse_of(q, q.in_hi[t0]));
    __syncthreads();
    uint64_t lo[kP2TileRows], hi[kP2TileRows];
    uint32_t where[kP2TileRows];  // LDS counter (low 12 bits) and rank in it; ~0: handled directly
    uint32_t live = 0;
    if (DIRECT) {
      in.next(p, T0 + TPT * trip, T1, threadIdx.x, lo, hi, live);
      // The LDS counters start at the smallest coarse bucket among the slice's first rows and stay
      // there: rows in time order move on by at most a bucket or two within a slice, and a barrier
      // per tile to follow them costs more than the few rows that then take the direct way.
      if (trip == 0) {
        uint32_t mine = ~0u;
#pragma unroll
        for (uint32_t r = 0; r < kP2TileRows; ++r)
          if ((live >> r) & 1u) { const uint32_t c = coarse_of(q, hi[r]); mine = c < mine ? c : mine; }
        if (mine != ~0u) atomicMin(&c_first, mine);
        __syncthreads();
        c_slice = c_first == ~0u ? 0u : c_first;
      }
      c0 = c_slice;
    } else {
#pragma unroll
      for (uint32_t r = 0; r < kP2TileRows; ++r) {
        const uint64_t i = t0 + (uint64_t)r * BLOCK + threadIdx.x;
        const bool in = i < r1;
        const uint64_t ic = in ? i : t0;
        lo[r] = __builtin_nontemporal_load((P2WordPtr)(uint64_t)(q.in_lo + ic));
        hi[r] = __builtin_nontemporal_load((P2WordPtr)(uint64_t)(q.in_hi + ic));
        live |= (in ? 1u : 0u) << r;
      }
    }
#pragma unroll
    for (uint32_t r = 0; r < kP2TileRows; ++r) {
      where[r] = ~0u;
      if (!((live >> r) & 1u)) continue;
      uint32_t c, s;
      if (q.compact) {
        // from here on the record is its two mixed words: the copy-out then only shifts and masks
        // (DIRECT: the high word stays as it is, its top bits are the coarse bucket)
        lo[r] = mixed_lo(q, lo[r], hi[r]);
        if (!DIRECT) hi[r] = bij_mix(hi[r], q.hi_bits);
        c = (uint32_t)(hi[r] >> (q.hi_bits - (int)q.p1_bits));
        s = (uint32_t)(lo[r] >> (q.lo_bits - (int)q.p2_bits));
      } else {
        c = coarse_of(q, hi[r]);
        s = sub_of(q, lo[r], hi[r]);
      }
      if (c - c0 < span) {
        const uint32_t li = (c - c0) * n_sub + s;
        where[r] = li | (atomicAdd(&cnt[li], 1u) << 12);
      } else {
        // beyond the coarse buckets the counters cover: a place of its own
        if (DIRECT) ++strays;
        const uint64_t ff = ((uint64_t)c << q.p2_bits) | s;
        const uint32_t g = atomicAdd(&q.fine_cursor[ff], 1u);
        if (q.fine_limit && g >= q.fine_limit[ff]) {
          if (spills) {
            const int lo_rest = q.lo_bits - (int)q.p2_bits, hi_rest = q.hi_bits - (int)q.p1_bits;
            const unsigned long long at = atomicAdd(q.spill_cursor, 1ull);
            if (at < q.spill_capacity) {
              q.spill_rec[at] = ((hi[r] & ((1ull << hi_rest) - 1ull)) << lo_rest) | (lo[r] & ((1ull << lo_rest) - 1ull));
              q.spill_fine[at] = (uint32_t)ff;
            } else {
              q.flags[1] = 1u;
            }
          } else {
            q.flags[1] = 1u;
          }
        } else if (g < q.out_capacity) {
          if (q.compact) {
            const int lo_rest = q.lo_bits - (int)q.p2_bits, hi_rest = q.hi_bits - (int)q.p1_bits;
            q.out_lo[g] = ((hi[r] & ((1ull << hi_rest) - 1ull)) << lo_rest) | (lo[r] & ((1ull << lo_rest) - 1ull));
          } else {
            q.out_lo[g] = lo[r];
            q.out_hi[g] = hi[r];
          }
        }
      }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < NC; i += BLOCK) {
      const uint32_t c = c0 + i / n_sub;
      const uint32_t n_i = i < n_cnt ? cnt[i] : 0u;
      lstart[i] = n_i;
      uint32_t at = 0u, fit = n_i, spill_local = 0u;
      if (n_i && c < n_coarse) {
        const uint64_t ff = ((uint64_t)c << q.p2_bits) | (i % n_sub);
        at = atomicAdd(&q.fine_cursor[ff], n_i);
        if (spills) {
          // what fits the bucket's region goes there, the rest of the run to the spill list
          const uint32_t lim = q.fine_limit[ff];
          fit = at >= lim ? 0u : (lim - at < n_i ? lim - at : n_i);
          if (fit < n_i) spill_local = atomicAdd(&spill_n, n_i - fit);
        } else if (q.fine_limit && (uint64_t)at + n_i > q.fine_limit[ff]) {
          // (regions of fixed size without a spill list: a run that does not fit is dropped as a whole)
          q.flags[1] = 1u;
          at = ~0u;
        }
      }
      base[i] = at;
      if (spills) cnt[i] = fit | (spill_local << 16);  // (the tile's counts have been read: the array carries this from here on)
    }
    __syncthreads();
    if (spills && threadIdx.x == 0 && spill_n) spill_at = atomicAdd(q.spill_cursor, (unsigned long long)spill_n);  // (read behind the scan's barriers)
    p2_block_scan<BLOCK>(lstart, NC, wave_tot);
#pragma unroll
    for (uint32_t r = 0; r < kP2TileRows; ++r) {
      if (where[r] == ~0u) continue;
      const uint32_t pos = lstart[where[r] & 0xfffu] + (where[r] >> 12);
      s_lo[pos] = lo[r];
      s_hi[pos] = hi[r];
    }
    __syncthreads();
    // out: neighbouring lanes hold neighbouring records of one fine bucket
    const uint32_t n_live = lstart[NC];
    if (q.compact) {
      const int lo_rest = q.lo_bits - (int)q.p2_bits, hi_rest = q.hi_bits - (int)q.p1_bits;
      for (uint32_t i = threadIdx.x; i < n_live; i += BLOCK) {
        const uint64_t tl = s_lo[i], th = s_hi[i];  // (staged mixed)
        const uint32_t li = ((uint32_t)(th >> hi_rest) - c0) * n_sub + (uint32_t)(tl >> lo_rest);
        const uint64_t rec = ((th & ((1ull << hi_rest) - 1ull)) << lo_rest) | (tl & ((1ull << lo_rest) - 1ull));
        const uint32_t k = i - lstart[li];
        if (spills && k >= (cnt[li] & 0xffffu)) {
          const unsigned long long at = spill_at + (cnt[li] >> 16) + (k - (cnt[li] & 0xffffu));
          if (at < q.spill_capacity) {
            q.spill_rec[at] = rec;
            q.spill_fine[at] = (uint32_t)((((uint64_t)c0 + li / n_sub) << q.p2_bits) | (li % n_sub));
          } else {
            q.flags[1] = 1u;
          }
          continue;
        }
        const uint32_t g = base[li] + k;
        if (base[li] != ~0u && g < q.out_capacity) q.out_lo[g] = rec;
      }
    } else {
      for (uint32_t i = threadIdx.x; i < n_live; i += BLOCK) {
        const unsigned long long w0 = s_lo[i], w1 = s_hi[i];
        const uint32_t li = (coarse_of(q, w1) - c0) * n_sub + sub_of(q, w0, w1);
        const uint32_t g = base[li] + (i - lstart[li]);
        if (base[li] != ~0u && g < q.out_capacity) {
          q.out_lo[g] = w0;
          q.out_hi[g] = w1;
        }
      }
    }
    __syncthreads();
  }
  if (DIRECT && q.strays) {
    if (strays) atomicAdd(&n_stray, strays);
    __syncthreads();
    if (threadIdx.x == 0 && n_stray) atomicAdd(q.strays, (unsigned long long)n_stray);
  }
}

// ---- the scatter pass by staging lines (Part2Params::lines) ------------------------------------

// The coarse bucket of the first row of scan tile t (key columns without NULLs: the caller checked).
__device__ __forceinline__ uint32_t p2_tile_coarse(const Part2Params& q, const AggParams& p, uint64_t t) {
  const uint32_t seg = find_seg(p.sd, t);
  const uint64_t row0 = (t - p.sd.segs[seg].tile_base) * kTileRows;
  uint64_t hi = 0;
  for (int k = 0; k < p.n_keys; ++k) {
    const KeySpec& ks = p.keys[k];
    if (!ks.word) continue;
    const PartBytePtr base = (PartBytePtr)(uint64_t)p.cols[(uint64_t)ks.col * p.sd.n_segs + seg].values;
    const unsigned long long raw = *(const __attribute__((address_space(1))) part_unaligned_u64*)(base + (row0 << part_width_shift(ks.width)));
    hi |= part_field(ks, raw) << ks.bit_pos;
  }
  return (uint32_t)(hi >> (q.hi_bits - (int)q.p1_bits));
}

// bounds[b] = the first scan tile whose first row lies in coarse bucket b or a later one (rows in
// time order: the coarse bucket is the top of the time key and never falls).  A wave per bound,
// sixty-four probes a step.  Rows that are not in order only cost places in the spill list: a
// piece sends the rows of other coarse buckets there.
__global__ __launch_bounds__(64) void k_part2_bounds(Part2Params q, AggParams p) {
  const uint32_t b = blockIdx.x, lane = threadIdx.x, n_coarse = 1u << q.p1_bits;
  const uint64_t n_tiles = q.n_tiles;
  uint64_t lo = 0, hi = n_tiles;  // the answer lies in [lo, hi]
  if (b == 0) hi = 0;
  if (b >= n_coarse) lo = n_tiles;
  while (hi > lo) {
    const uint64_t step = (hi - lo + 63) / 64;
    const uint64_t t = lo + (uint64_t)lane * step;
    const bool reached = t >= hi ? true : p2_tile_coarse(q, p, t) >= b;
    const uint64_t m = __ballot(reached);
    if (m == 0ull) {  // (no probe reached the bucket: it starts behind the last one)
      lo = lo + 63 * step + 1;
      if (lo > hi) lo = hi;
      continue;
    }
    const uint32_t first = (uint32_t)__builtin_ctzll(m);
    const uint64_t at = lo + (uint64_t)first * step;
    hi = at < hi ? at : hi;
    if (first == 0) hi = lo;
    else lo = lo + (uint64_t)(first - 1) * step + 1;
    if (lo > hi) lo = hi;
  }
  if (lane == 0) q.bounds[b] = (uint32_t)lo;
}

// The pieces: coarse bucket c's stretch of tiles, [bounds[c], bounds[c + 1]), in k = 1 .. piece_max
// parts of about piece_tiles tiles; part j fills records [j * cap, (j + 1) * cap) of its fine
// buckets' regions, cap = region_cap / k rounded down to whole lines.  One workgroup.
__global__ __launch_bounds__(1024) void k_part2_pieces(Part2Params q) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry;
  const uint32_t n_coarse = 1u << q.p1_bits, tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t c0 = 0; c0 < n_coarse; c0 += 1024) {
    const uint32_t c = c0 + tid;
    uint32_t t0 = 0, t1 = 0, k = 0;
    if (c < n_coarse) {
      t0 = q.bounds[c];
      t1 = q.bounds[c + 1];
      if (t1 > t0) {
        k = (t1 - t0 + q.piece_tiles - 1) / q.piece_tiles;
        k = k < 1 ? 1u : (k > q.piece_max ? q.piece_max : k);
      }
    }
    uint32_t incl = k;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off, 64);
      if ((int)(tid & 63) >= off) incl += o;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = carry;
    for (uint32_t w = 0; w < (tid >> 6); ++w) before += wave_tot[w];
    const uint32_t first = before + incl - k;
    const bool has_region = (((uint64_t)c << q.p2_bits) < q.n_regions);
    const uint32_t cap = k && has_region ? (q.region_cap / k) & ~7u : 0u;
    for (uint32_t j = 0; j < k; ++j) {
      if (first + j >= q.max_pieces) { q.flags[1] = 1u; break; }
      P2Piece pc;
      pc.t0 = t0 + (uint32_t)((uint64_t)(t1 - t0) * j / k);
      pc.t1 = t0 + (uint32_t)((uint64_t)(t1 - t0) * (j + 1) / k);
      // (a coarse bucket with far more rows than the others: its pieces would outgrow the workgroup's table of spill chunks)
      if (pc.t1 - pc.t0 > 2 * q.piece_tiles) *q.pieces_too_long = 1u;
      pc.coarse = c;
      pc.src = j;
      pc.seg_first = j * cap;
      pc.seg_cap = cap;
      q.pieces[first + j] = pc;
    }
    __syncthreads();
    if (tid == 1023) carry = before + incl;
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) {
    *q.n_pieces = *q.pieces_too_long ? 0u : (carry < q.max_pieces ? carry : q.max_pieces);
    P2Rare r;
    r.spill_rec = q.spill_rec;
    r.spill_fine = q.spill_fine;
    r.spill_cursor = q.spill_cursor;
    r.spill_capacity = q.spill_capacity;
    r.flags = q.flags;
    r.strays = q.strays;
    r.dropped = q.dropped;
    r.seg_base_out = q.seg_base_out;
    r.seg_cnt_out = q.seg_cnt_out;
    r.piece_chunks = q.piece_chunks;
    r.n_regions = q.n_regions;
    *q.rare = r;
  }
}

constexpr uint32_t kP2Stage = 8;  // records of a staging line (64 bytes)
// The spill list is handed out to the workgroups in chunks of this many records: one addition to
// the list's cursor per chunk (an addition per spilled line -- millions to one address, carried
// out one after the other -- made the pass four times as long).  Inside its chunks a workgroup
// counts in LDS; what it leaves unused of its last chunk is marked (spill_fine = all ones).
constexpr uint32_t kP2SpillChunk = 1024;
constexpr uint32_t kP2NoChunk = 0xffffffffu, kP2BadChunk = 0xfffffffeu;
constexpr uint32_t kP2Recent = 8;
constexpr size_t p2_lines_lds(uint32_t p2_bits, uint32_t chunks) {
  return (((size_t)kP2Stage * 8 + 8 + (size_t)kP2Recent * 8) << p2_bits) + (size_t)chunks * 4;
}
typedef unsigned long long __attribute__((ext_vector_type(2), aligned(8))) p2_u64x2;
typedef uint32_t __attribute__((ext_vector_type(4), aligned(4))) p2_u32x4;

// A workgroup per piece.  Per sub-bucket: `fill` hands out tickets (ticket t: place t % 8 of
// generation t / 8 of the sub-bucket's line), `prog` holds the records of the current generation
// already in place (low half) and the generations copied out so far (high half, modulo 2^16);
// whoever puts a generation's eighth record in place copies the line out and opens the next
// generation.  No tile, no sort, no barrier inside the loop (the protocol of the histogram-free
// GROUP BY's scatter pass, kernels_wc.hip).
template <int NK, bool BIN64>
__global__ __launch_bounds__(kP2Block) __attribute__((amdgpu_waves_per_eu(4))) void k_part2_lines(Part2Params q, AggParams p) {
  constexpr int R = 4;
  extern __shared__ __align__(16) unsigned long long p2l_lds[];
  __shared__ uint32_t n_stray;
  if (blockIdx.x >= *q.n_pieces) return;
  const P2Rare* const rare = q.rare;  // (read where it is used: see P2Rare)
  const uint32_t n_sub = 1u << q.p2_bits, tid = threadIdx.x;
  unsigned long long* stage = p2l_lds;                                  // [n_sub][kP2Stage]
  uint32_t* fill = (uint32_t*)(stage + (size_t)n_sub * kP2Stage);       // [n_sub]
  uint32_t* prog = fill + n_sub;                                        // [n_sub]
  // Records the piece has staged lately, four per sub-bucket: a row that repeats one of them is
  // dropped where it stands -- a pair counts once however often it comes, and the pairs that come
  // tens of thousands of times (a crawler's id in one hour) are what overflows the regions.
  unsigned long long* recent = (unsigned long long*)(prog + n_sub);     // [n_sub][kP2Recent]
  uint32_t* chunk = (uint32_t*)(recent + (size_t)n_sub * kP2Recent);    // [rare->piece_chunks] where the piece's spill chunks start
  __shared__ uint32_t sp_pos;                                           // records the piece has spilled
  const uint32_t t0 = __builtin_amdgcn_readfirstlane(q.pieces[blockIdx.x].t0), t1 = __builtin_amdgcn_readfirstlane(q.pieces[blockIdx.x].t1);
  const uint32_t coarse = __builtin_amdgcn_readfirstlane(q.pieces[blockIdx.x].coarse), src = __builtin_amdgcn_readfirstlane(q.pieces[blockIdx.x].src);
  const uint32_t seg_first = __builtin_amdgcn_readfirstlane(q.pieces[blockIdx.x].seg_first);
  const uint32_t seg_cap = __builtin_amdgcn_readfirstlane(q.pieces[blockIdx.x].seg_cap);
  for (uint32_t i = tid; i < n_sub; i += kP2Block) { fill[i] = 0; prog[i] = 0; }
  for (uint32_t i = tid; i < rare->piece_chunks; i += kP2Block) chunk[i] = kP2NoChunk;
  for (uint32_t i = tid; i < n_sub * kP2Recent; i += kP2Block) recent[i] = kP2Empty;
  if (tid == 0) { n_stray = 0; sp_pos = 0; }
  __syncthreads();
  const int lo_rest = q.lo_bits - (int)q.p2_bits, hi_rest = q.hi_bits - (int)q.p1_bits;
  const uint64_t lo_mask = (1ull << lo_rest) - 1ull, hi_mask = (1ull << hi_rest) - 1ull;
  // record r of sub-bucket s of this piece lies at out_lo[(coarse << p2 | s) * region_cap + seg_first + r]
  const uint64_t piece_base = ((uint64_t)coarse << q.p2_bits) * q.region_cap + seg_first;
  // The place of the piece's spilled record number `pos` in the list (all ones: none).  Whoever
  // holds the first record of a chunk fetches the chunk (sp_chunk) BEFORE anybody of its wave
  // asks for a place in it; lanes of other waves wait here for the chunk's start to appear.
  auto sp_chunk = [&](uint32_t ci) {
    if (ci >= rare->piece_chunks) { rare->flags[1] = 1u; return; }
    const unsigned long long at = atomicAdd(rare->spill_cursor, (unsigned long long)kP2SpillChunk);
    const bool fits = at + kP2SpillChunk <= rare->spill_capacity;
    if (!fits) rare->flags[1] = 1u;
    __hip_atomic_store(&chunk[ci], fits ? (uint32_t)at : kP2BadChunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto sp_place = [&](uint32_t pos) -> uint32_t {
    const uint32_t ci = pos / kP2SpillChunk;
    if (ci >= rare->piece_chunks) return kP2NoChunk;
    uint32_t base;
    do {
      base = __hip_atomic_load(&chunk[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } while (base == kP2NoChunk);
    return base == kP2BadChunk ? kP2NoChunk : base + pos % kP2SpillChunk;
  };
  auto spill_one = [&](uint64_t rec, uint32_t ff) {
    if (ff >= rare->n_regions) { rare->flags[0] = 1u; return; }  // (a bucket the columns' statistics rule out: the result is void)
    const uint32_t pos = atomicAdd(&sp_pos, 1u);
    if (pos % kP2SpillChunk == 0) sp_chunk(pos / kP2SpillChunk);
    const uint32_t at = sp_place(pos);
    if (at != kP2NoChunk) { rare->spill_rec[at] = rec; rare->spill_fine[at] = ff; }
  };
  unsigned long long raw[NK][R];
  uint32_t live = 0, live_next = 0, strays = 0, drops = 0;
  PartCursor<NK> cur;
  if (t0 < t1) {
    part_cursor_init<NK, 0>(p, cur, t0);
    part_issue<NK, 0, R>(p, cur, t0, t1, tid, raw, live_next);
  }
  for (uint64_t t = t0; t < t1; ++t) {
    uint64_t rec2[2][R], key[R];
    part_finish<2, NK, 0, R, BIN64>(p, raw, rec2);
    live = live_next;
    if (t + 1 < t1) part_issue<NK, 0, R>(p, cur, t + 1, t1, tid, raw, live_next);
    uint32_t bkt[R], tk[R], needw = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t hi = rec2[1][r], lo = mixed_lo(q, rec2[0][r], hi);
      const uint32_t c = (uint32_t)(hi >> hi_rest);
      bkt[r] = (uint32_t)(lo >> lo_rest);
      key[r] = ((hi & hi_mask) << lo_rest) | (lo & lo_mask);
      tk[r] = 0;
      if (!((live >> r) & 1u)) continue;
      if (c == coarse) {
        unsigned long long* seen = &recent[bkt[r] * kP2Recent + ((uint32_t)key[r] & (kP2Recent - 1))];
        if (*seen == key[r]) { ++drops; continue; }  // (a record is less than 2^63: never the empty entry)
        *seen = key[r];
        tk[r] = atomicAdd(&fill[bkt[r]], 1u);
        needw |= 1u << r;
      } else {
        // a row of another coarse bucket (the tiles where the bucket changes; rows out of time order)
        ++strays;
        spill_one(key[r], (c << q.p2_bits) | bkt[r]);
      }
    }
    for (;;) {
      uint32_t pr[R], before[R], ok = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) pr[r] = __hip_atomic_load(&prog[bkt[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t gen = tk[r] / kP2Stage, place = tk[r] % kP2Stage;
        if (((needw >> r) & 1u) && (pr[r] >> 16) == (gen & 0xffffu)) {  // (else the line still holds the generation before)
          ok |= 1u << r;
          __hip_atomic_store(&stage[(size_t)bkt[r] * kP2Stage + place], (unsigned long long)key[r], __ATOMIC_RELAXED,
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
      uint32_t fl = 0;
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (((ok >> r) & 1u) && (before[r] & 0xffffu) == kP2Stage - 1) fl |= 1u << r;
      // (the copy-out is issued once per lane and round, for the lane's first full line, and again only while some lane has another)
      while (__ballot(fl != 0u) != 0ull) {
        if (fl != 0u) {
          const int r = __ffs((int)fl) - 1;
          fl &= fl - 1u;
          const uint32_t fb = r == 0 ? bkt[0] : r == 1 ? bkt[1] : r == 2 ? bkt[2] : bkt[3];
          const uint32_t ft = r == 0 ? tk[0] : r == 1 ? tk[1] : r == 2 ? tk[2] : tk[3];
          const ulonglong2* sl = (const ulonglong2*)&stage[(size_t)fb * kP2Stage];
          asm volatile("" ::: "memory");
          const ulonglong2 a = sl[0], b = sl[1], c = sl[2], d = sl[3];
          asm volatile("" ::: "memory");
          // the records are on their way to registers: the next generation may come (-8 records, +1 generation)
          __hip_atomic_fetch_add(&prog[fb], 0x10000u - kP2Stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const uint32_t off = (ft / kP2Stage) * kP2Stage;
          if (off + kP2Stage <= seg_cap) {
            ulonglong2* dst = (ulonglong2*)(q.out_lo + (piece_base + (uint64_t)fb * q.region_cap + off));
            dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
          } else {
            // beyond the piece's part of the region: the line goes to the spill list
            const uint32_t pos = atomicAdd(&sp_pos, kP2Stage);
            const uint32_t c0 = pos / kP2SpillChunk, c1 = (pos + kP2Stage - 1) / kP2SpillChunk;
            if (pos % kP2SpillChunk == 0) sp_chunk(c0);
            if (c1 != c0) sp_chunk(c1);  // (the line holds the first record of the next chunk)
            const uint32_t ff = (coarse << q.p2_bits) | fb;
            if (c1 == c0) {
              const uint32_t at = sp_place(pos);
              if (at != kP2NoChunk) {
                p2_u64x2* dr = (p2_u64x2*)(rare->spill_rec + at);
                dr[0] = p2_u64x2{a.x, a.y}; dr[1] = p2_u64x2{b.x, b.y}; dr[2] = p2_u64x2{c.x, c.y}; dr[3] = p2_u64x2{d.x, d.y};
                p2_u32x4* df = (p2_u32x4*)(rare->spill_fine + at);
                df[0] = p2_u32x4{ff, ff, ff, ff}; df[1] = p2_u32x4{ff, ff, ff, ff};
              }
            } else {
              const unsigned long long w[kP2Stage] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
#pragma unroll
              for (uint32_t j = 0; j < kP2Stage; ++j) {
                const uint32_t at = sp_place(pos + j);
                if (at != kP2NoChunk) { rare->spill_rec[at] = w[j]; rare->spill_fine[at] = ff; }
              }
            }
          }
        }
      }
      if (__ballot(needw != 0u) == 0ull) break;
    }
  }
  __syncthreads();
  // what is left in the lines, and how many records of every fine bucket the piece's part of the region got
  for (uint32_t b = tid; b < n_sub; b += kP2Block) {
    const uint32_t n = fill[b], rem = n % kP2Stage, off = (n / kP2Stage) * kP2Stage;
    const uint32_t ff = (coarse << q.p2_bits) | b;
    uint32_t in_region = off < seg_cap ? off : seg_cap;
    if (rem) {
      if (off + kP2Stage <= seg_cap) {
        unsigned long long* dst = (unsigned long long*)q.out_lo + (piece_base + (uint64_t)b * q.region_cap + off);
        for (uint32_t j = 0; j < rem; ++j) dst[j] = stage[(size_t)b * kP2Stage + j];
        in_region += rem;
      } else {
        for (uint32_t j = 0; j < rem; ++j) spill_one(stage[(size_t)b * kP2Stage + j], ff);
      }
    }
    if (ff < rare->n_regions) {
      rare->seg_base_out[(uint64_t)src * rare->n_regions + ff] = (uint32_t)(piece_base + (uint64_t)b * q.region_cap);
      rare->seg_cnt_out[(uint64_t)src * rare->n_regions + ff] = in_region;
    }
  }
  if (strays) atomicAdd(&n_stray, strays);
  {
    const unsigned long long wave_drops = wave_sum_u64(drops);
    if ((tid & 63u) == 0 && wave_drops && rare->dropped) atomicAdd(rare->dropped, wave_drops);
  }
  __syncthreads();
  if (tid == 0 && n_stray && rare->strays) atomicAdd(rare->strays, (unsigned long long)n_stray);
  // the rest of the piece's last spill chunk holds no records
  const uint32_t spilled = sp_pos;
  if (spilled % kP2SpillChunk != 0 && spilled / kP2SpillChunk < rare->piece_chunks) {
    const uint32_t base = chunk[spilled / kP2SpillChunk];
    if (base != kP2NoChunk && base != kP2BadChunk)
      for (uint32_t i = spilled % kP2SpillChunk + tid; i < kP2SpillChunk; i += kP2Block) rare->spill_fine[base + i] = 0xffffffffu;
  }
}

// (lines) the spill list's part of every fine bucket as its last run
__global__ __launch_bounds__(256) void k_spill_runs_lines(Part2Params q, uint64_t out_offset) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f >= q.n_regions) return;
  q.seg_base_out[(uint64_t)q.piece_max * q.n_regions + f] = (uint32_t)(out_offset + q.spill_base[f]);
  q.seg_cnt_out[(uint64_t)q.piece_max * q.n_regions + f] = q.spill_count[f];
}

// One past the last record of fine bucket f: where the next bucket starts, or (regions of fixed
// size) as far as the scatter pass' cursor got.
__device__ __forceinline__ uint32_t p2_bucket_end(const Part2Params& q, uint32_t f) {
  if (!q.fine_limit) return q.fine_base[f + 1];
  const uint32_t end = q.fine_cursor[f], lim = q.fine_limit[f];
  return end < lim ? end : lim;
}

// Puts the pair into the LDS set unless it is there: 1 = it was new, 0 = it was there, -1 = the
// set is full (which the passes of k_part2_agg are sized to avoid).  The low word claims the entry;
// the claimant publishes the high word right after, until then it reads all ones.
__device__ __forceinline__ int set_insert(unsigned long long* keys_lo, unsigned long long* keys_hi, uint32_t S, uint64_t lo,
                                          uint64_t hi) {
  // (32-bit mixing: the look-up runs once per row, and 64-bit multiplies are several instructions each)
  uint32_t h = (uint32_t)lo ^ (uint32_t)(lo >> 32) * 0x85EBCA6Bu ^ (uint32_t)hi * 0xC2B2AE35u ^ (uint32_t)(hi >> 32);
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  uint32_t idx = __umulhi(h, S);
  for (uint32_t probes = 0; probes < S; ++probes) {
    unsigned long long k = __hip_atomic_load(&keys_lo[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (k == kP2Empty) {
      const unsigned long long old = atomicCAS(&keys_lo[idx], kP2Empty, (unsigned long long)lo);
      if (old == kP2Empty) {
        __hip_atomic_store(&keys_hi[idx], (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return 1;
      }
      k = old;
    }
    if (k == lo) {
      const unsigned long long kh = __hip_atomic_load(&keys_hi[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (kh == hi) return 0;
      if (kh == kP2Empty) { --probes; continue; }  // not published yet: look again
    }
    idx = idx + 1 == S ? 0u : idx + 1;
  }
  return -1;  // every entry holds another pair
}

__global__ __launch_bounds__(kP2AggBlock) void k_part2_agg(Part2Params q, uint32_t n_fine) {
  extern __shared__ __align__(16) unsigned long long p2_lds[];
  __shared__ unsigned long long gkey[kP2Groups];
  __shared__ uint32_t gcnt[2][kP2Groups];
  const uint32_t S = q.lds_entries, tid = threadIdx.x;
  unsigned long long* keys_lo = p2_lds;
  unsigned long long* keys_hi = p2_lds + S;
  for (uint32_t i = tid; i < S; i += kP2AggBlock) { keys_lo[i] = kP2Empty; keys_hi[i] = kP2Empty; }
  for (uint32_t i = tid; i < kP2Groups; i += kP2AggBlock) { gkey[i] = kP2Empty; gcnt[0][i] = 0; gcnt[1][i] = 0; }
  // fine buckets are dealt round robin: they are many and alike, and a counter to hand them
  // out would put a device-scope atomic and a barrier in front of every one of them
  for (uint32_t f = blockIdx.x; f < n_fine; f += gridDim.x) {
    __syncthreads();
    const uint32_t b0 = q.fine_base[f], b1 = p2_bucket_end(q, f);
    if (b1 == b0) continue;
    // A fine bucket of more rows than the set is sure to take (the coarse bucket of a very
    // frequent group) is worked off in several passes over its rows, each taking the pairs of
    // one residue class of another hash: the classes are disjoint, so their counts add up (the
    // group counters go on counting from pass to pass; only the set is emptied in between).
    const uint32_t n = b1 - b0;
    const uint32_t per_pass = S / 5 * 2;
    const uint32_t passes = n > S / 2 ? (n + per_pass - 1) / per_pass : 1u;
    auto take = [&](uint64_t lo, uint64_t hi, uint32_t pass) {
      if (passes > 1 && (uint32_t)(hash64(lo * 0xD6E8FEB86659FD93ull + hi) >> 33) % passes != pass) return;
      const int r = set_insert(keys_lo, keys_hi, S, lo, hi);
      if (r < 0) q.flags[0] = 1u;  // (a class far larger than its share: the result is void)
      if (r != 1) return;
      // a pair seen for the first time counts for its group (walking the set afterwards to
      // count its entries cost three times the instructions of the insertions themselves)
      const uint32_t not_null = (q.x_null_bit >= 0 && ((lo >> q.x_null_bit) & 1ull)) ? 0u : 1u;
      uint32_t g = (((uint32_t)hi ^ (uint32_t)(hi >> 32)) * 0x9E3779B1u) >> 23;  // kP2Groups = 2^9
      bool placed = false;
      for (uint32_t probes = 0; probes < kP2Groups && !placed; ++probes) {
        unsigned long long k = __hip_atomic_load(&gkey[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == kP2Empty) {
          const unsigned long long old = atomicCAS(&gkey[g], kP2Empty, (unsigned long long)hi);
          k = old == kP2Empty ? hi : old;
        }
        if (k == hi) {
          atomicAdd(&gcnt[0][g], 1u);
          if (not_null) atomicAdd(&gcnt[1][g], 1u);
          placed = true;
        } else {
          g = g + 1 == kP2Groups ? 0u : g + 1;
        }
      }
      if (!placed) {  // more groups in one fine bucket than the counters hold: straight to the outer table
        const uint64_t e = global_entry_raw(q.dense, q.capacity, q.g_keys, q.overflow, hi);
        if (e != ~0ull) {
          atomicAdd((unsigned long long*)&q.g_state[e], 1ull);
          if (not_null) atomicAdd((unsigned long long*)&q.g_state[(q.capacity + 1) + e], 1ull);
        }
      }
    };
    // The rows of a bucket are few per thread.  The first four are requested together, so that
    // the thread waits for memory once, not once per row (9.0 -> 7.5 ms at 4e8 rows), and stay
    // in registers for all the passes.
    constexpr uint32_t kBatch = 4;
    uint64_t blo[kBatch], bhi[kBatch];
#pragma unroll
    for (uint32_t j = 0; j < kBatch; ++j) {
      const uint32_t i = b0 + tid + j * kP2AggBlock, ic = i < b1 ? i : b0;
      blo[j] = __builtin_nontemporal_load((P2WordPtr)(uint64_t)(q.out_lo + ic));
      bhi[j] = __builtin_nontemporal_load((P2WordPtr)(uint64_t)(q.out_hi + ic));
    }
    for (uint32_t pass = 0; pass < passes; ++pass) {
      if (pass) {
        __syncthreads();
        for (uint32_t i = tid; i < S; i += kP2AggBlock) { keys_lo[i] = kP2Empty; keys_hi[i] = kP2Empty; }
        __syncthreads();
      }
#pragma unroll
      for (uint32_t j = 0; j < kBatch; ++j)
        if (b0 + tid + j * kP2AggBlock < b1) take(blo[j], bhi[j], pass);
      // (what lies beyond four rows a thread is fetched as it is needed, once per pass)
      for (uint32_t i = b0 + tid + kBatch * kP2AggBlock; i < b1; i += kP2AggBlock)
        take(__builtin_nontemporal_load((P2WordPtr)(uint64_t)(q.out_lo + i)), __builtin_nontemporal_load((P2WordPtr)(uint64_t)(q.out_hi + i)), pass);
    }
    __syncthreads();
    // the set is emptied for the next bucket, the groups' counts go to the outer table
    for (uint32_t i = tid; i < S; i += kP2AggBlock) { keys_lo[i] = kP2Empty; keys_hi[i] = kP2Empty; }
    for (uint32_t i = tid; i < kP2Groups; i += kP2AggBlock) {
      const unsigned long long hi = gkey[i];
      if (hi == kP2Empty) continue;
      const uint64_t e = global_entry_raw(q.dense, q.capacity, q.g_keys, q.overflow, hi);
      if (e != ~0ull) {
        atomicAdd((unsigned long long*)&q.g_state[e], (unsigned long long)gcnt[0][i]);
        if (gcnt[1][i]) atomicAdd((unsigned long long*)&q.g_state[(q.capacity + 1) + e], (unsigned long long)gcnt[1][i]);
      }
      gkey[i] = kP2Empty;
      gcnt[0][i] = 0;
      gcnt[1][i] = 0;
    }
  }
}

// The last pass over one-word records (compact form).  A fine bucket's records share the
// coarse bucket, so the high rest of a record (at most 9 bits) numbers the group inside it:
// the group counters are indexed directly, and the set is a plain one-word table (one CAS to
// claim an entry, nothing to publish).  The groups' keys are rebuilt from (coarse bucket, high
// rest) through the inverse of the bijection when the counts leave for the outer table.
//
// MODE kP2Count: as described.  kP2Emit (the rank's own part of a query sharded over several
// GPUs): a record seen for the first time is written to q.emit_out instead -- the bucket's
// distinct records, at the bucket's place -- and q.fine_count[f] receives their number; nothing
// is counted.  kP2Merge (the owner's part): bucket j of this rank's share of the fine buckets
// is the union of one run of records per sending rank (q.seg_base / q.seg_cnt, in q.out_lo),
// each run distinct in itself; counted like kP2Count.
// kP2EmitRuns: both -- a rank's own part whose buckets lie in several runs (the scatter by staging
// lines): the bucket's distinct records go to q.emit_out at the start of the bucket's region.
enum P2AggMode { kP2Count = 0, kP2Emit = 1, kP2Merge = 2, kP2EmitRuns = 3 };

template <int MODE>
__global__ __launch_bounds__(kP2AggBlock) void k_part2_agg_compact(Part2Params q, uint32_t n_fine) {
  extern __shared__ __align__(16) unsigned long long p2_lds[];
  __shared__ uint32_t gcnt[kP2Groups];
  __shared__ uint32_t crowded;  // a row of the bucket found no room on its probe path
  __shared__ uint32_t emit_n;   // kP2Emit: distinct records of the bucket written so far
  const uint32_t S = q.lds_entries, tid = threadIdx.x, lane = tid & 63u;
  const int lo_rest = q.lo_bits - (int)q.p2_bits, hi_rest = q.hi_bits - (int)q.p1_bits;
  const uint32_t n_groups = 1u << hi_rest;  // <= kP2Groups
  unsigned long long* keys = p2_lds;
  unsigned long long my_pairs = 0;  // distinct pairs this thread handed to the outer table
  for (uint32_t i = tid; i < kP2Groups; i += kP2AggBlock) gcnt[i] = 0;
  if (tid == 0) emit_n = 0;
  constexpr uint32_t kBatch = 8;  // rows a thread requests together
  const uint32_t n_src = (MODE & kP2Merge) != 0 ? q.n_src : 1u;
  // kP2Merge: the bucket's runs (at most 64) are taken as ONE sequence of records -- record i lies
  // in the run r with run_off[r] <= i < run_off[r + 1], at run_base[r] + i - run_off[r] -- so that
  // a batch of loads is full whatever the runs' lengths (run by run, six runs of a thousand records
  // left three quarters of every batch of 8 x 512 empty).  The first wave keeps the runs of the
  // workgroup's next bucket in registers (lane = run), fetched while this bucket is counted.
  __shared__ uint32_t run_off[65], run_base[64];
  uint32_t nx_base = 0, nx_cnt = 0;
  auto fetch_runs = [&](uint32_t f2) {
    nx_base = 0;
    nx_cnt = 0;
    if ((MODE & kP2Merge) != 0 && tid < n_src && f2 < n_fine) {
      nx_base = q.seg_base[(uint64_t)tid * n_fine + f2];
      nx_cnt = q.seg_cnt[(uint64_t)tid * n_fine + f2];
    }
  };
  fetch_runs(blockIdx.x);
  for (uint32_t f = blockIdx.x; f < n_fine; f += gridDim.x) {
    __syncthreads();
    uint32_t b0 = 0, n = 0;  // (other modes: the bucket is records [b0, b0 + n))
    if ((MODE & kP2Merge) != 0) {
      if (tid < 64) {
        uint32_t incl = nx_cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t o = __shfl_up(incl, off, 64);
          if ((int)tid >= off) incl += o;
        }
        if (tid < n_src) { run_base[tid] = nx_base; run_off[tid] = incl - nx_cnt; }
        if (tid == 63) run_off[n_src] = incl;
      }
      fetch_runs(f + gridDim.x);
      __syncthreads();
      n = run_off[n_src];
      if ((MODE & kP2Emit) != 0) b0 = q.fine_base[f];  // (where the bucket's distinct records go)
    } else {
      b0 = q.fine_base[f];
      n = p2_bucket_end(q, f) - b0;
    }
    if (n == 0) {
      if ((MODE & kP2Emit) != 0 && tid == 0) q.fine_count[f] = 0;
      continue;
    }
    // where record i of the bucket lies in q.out_lo (i < n)
    // (a thread asks for rising i: `r`, the run of its last record, is where the search goes on)
    auto at_of = [&](uint32_t i, uint32_t& r) -> uint32_t {
      if ((MODE & kP2Merge) == 0) return b0 + i;
      while (r + 1 < n_src && i >= run_off[r + 1]) ++r;
      return run_base[r] + (i - run_off[r]);
    };
    uint64_t b[kBatch];  // the leading rows stay in registers for all the passes
    uint32_t run_first = 0;
#pragma unroll
    for (uint32_t j = 0; j < kBatch; ++j) {
      const uint32_t i = tid + j * kP2AggBlock;
      b[j] = __builtin_nontemporal_load((P2WordPtr)(uint64_t)(q.out_lo + at_of(i < n ? i : n - 1, run_first)));
    }
    // A bucket of more rows than the set is sure to take (n > S / 2) is first tried in one go
    // all the same: what makes a bucket large is, as a rule, one very frequent pair, and rows
    // that repeat a pair need no entry.  Only when a row finds its probe path crowded is the
    // attempt given up, its counts forgotten, and the bucket taken in several passes over its
    // rows, each for one residue class of another hash of the record.  (Going straight to
    // the passes made the buckets of the most frequent pairs cost a dozen walks over tens of
    // thousands of rows each, and the workgroups they fell to finished long after the rest.)
    const bool large = n > S / 2;
    uint32_t passes = 1;
    for (;;) {
      // the part of the set in use (cleared before every pass): at least twice the rows a
      // pass may bring, so that small buckets do not pay for clearing the whole table
      const uint32_t want = large ? S : n * 5 / 2 + 64;
      const uint32_t Sb = (want < S ? want : S) & ~1u;
      const uint32_t probe_limit = (large && passes == 1) ? 64u : Sb;
      // Buckets of two entries, read with one 16-byte load: a wave's look-up takes as many
      // steps as its slowest lane, and with single entries and linear probing that was five
      // or six steps a row at a third load.  A record sits in the first bucket on its path
      // that had a free entry when it came; entries are never removed, so a bucket with a
      // free entry ends every search, and a free entry is only ever claimed by compare-and-swap.
      // (Reading the home buckets of a thread's eight rows ahead, so that the LDS latencies
      // overlap, made the pass slower, 7.5 -> 10.7 ms: rows that repeat a frequent pair then
      // all see its entry still free and fight over it.  Tried again with the claims of a batch
      // sent together and the rows that did not settle at home walking on together, after the
      // scatter pass had begun to drop the most frequent repeats: 6.9 -> 10.9 ms.)
      const uint32_t nb = Sb >> 1;
      // true: the record was seen for the first time
      auto take = [&](uint64_t rec, uint32_t pass) -> bool {
        uint32_t h = ((uint32_t)rec ^ (uint32_t)(rec >> 32) * 0x85EBCA6Bu);
        h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
        if (passes > 1 && (uint32_t)(hash64(rec * 0xD6E8FEB86659FD93ull) >> 33) % passes != pass) return false;
        uint32_t bkt = __umulhi(h, nb);
        for (uint32_t probes = 0; probes < probe_limit; ++probes) {
          asm volatile("" ::: "memory");  // the bucket is read afresh in every step
          const ulonglong2 c = *(const ulonglong2*)&keys[2 * (size_t)bkt];
          if (c.x == rec || c.y == rec) return false;
          if (c.x == kP2Empty || c.y == kP2Empty) {
            const uint32_t e = 2 * bkt + (c.x == kP2Empty ? 0u : 1u);
            const unsigned long long old = atomicCAS(&keys[e], kP2Empty, (unsigned long long)rec);
            if (old == kP2Empty) {  // seen for the first time: counts for its group
              if ((MODE & kP2Emit) == 0) atomicAdd(&gcnt[(uint32_t)(rec >> lo_rest)], 1u);
              return true;
            }
            if (old == rec) return false;
            continue;  // another record took the entry: look at the bucket again
          }
          bkt = bkt + 1 == nb ? 0u : bkt + 1;
        }
        if (passes == 1) crowded = 1u;
        else q.flags[0] = 1u;  // a class far larger than its share: the result is void
        return false;
      };
      // (every thread of the workgroup makes the same calls: the ballot below needs whole waves)
      auto take_batch = [&](const uint64_t (&r)[kBatch], uint32_t first, uint32_t end, uint32_t pass) {
#pragma unroll
        for (uint32_t j = 0; j < kBatch; ++j) {
          bool fresh = false;
          if (first + tid + j * kP2AggBlock < end) fresh = take(r[j], pass);
          if ((MODE & kP2Emit) != 0) {
            // the wave's new records take consecutive places behind the bucket's earlier ones
            const uint64_t m = __ballot(fresh);
            if (m) {
              uint32_t at = 0;
              if (lane == (uint32_t)(__ffsll((long long)m) - 1)) at = atomicAdd(&emit_n, (uint32_t)__popcll(m));
              at = __shfl(at, __ffsll((long long)m) - 1);
              const uint32_t place = at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
              // (kP2EmitRuns: the records go to the bucket's region, which may be shorter than the bucket with its spilled rows)
              if (fresh && (MODE & kP2Merge) != 0 && place >= q.region_cap) q.flags[1] = 1u;
              else if (fresh) q.emit_out[(uint64_t)b0 + place] = r[j];
            }
          }
        }
      };
      for (uint32_t pass = 0; pass < passes; ++pass) {
        if (pass) __syncthreads();
        for (uint32_t i = tid; i < Sb; i += kP2AggBlock) keys[i] = kP2Empty;
        if (tid == 0) crowded = 0;
        __syncthreads();
        take_batch(b, 0, n, pass);
        // (what lies beyond the first rows is fetched once per pass, again a batch at a time so
        // that a thread waits for memory once per batch)
        uint32_t run_more = run_first;
        for (uint32_t base = kBatch * kP2AggBlock; base < n; base += kBatch * kP2AggBlock) {
          uint64_t more[kBatch];
#pragma unroll
          for (uint32_t j = 0; j < kBatch; ++j) {
            const uint32_t i = base + tid + j * kP2AggBlock;
            more[j] = __builtin_nontemporal_load((P2WordPtr)(uint64_t)(q.out_lo + at_of(i < n ? i : n - 1, run_more)));
          }
          take_batch(more, base, n, pass);
        }
      }
      __syncthreads();
      if (passes > 1 || !crowded) break;
      for (uint32_t i = tid; i < n_groups; i += kP2AggBlock) gcnt[i] = 0;
      if (tid == 0) emit_n = 0;
      const uint32_t per_pass = S / 5 * 2;
      passes = (n + per_pass - 1) / per_pass;
      if (passes < 2) passes = 2;
      __syncthreads();
    }
    if ((MODE & kP2Emit) != 0) {
      if (tid == 0) {
        q.fine_count[f] = emit_n;
        my_pairs += emit_n;
        emit_n = 0;
      }
      continue;
    }
    // (kP2Merge: bucket f of this rank's share is fine bucket (c * world + rank, sub) of the query)
    const uint64_t coarse = (MODE & kP2Merge) != 0 ? (uint64_t)(f >> q.p2_bits) * q.world + q.rank : (uint64_t)(f >> q.p2_bits);
    for (uint32_t i = tid; i < n_groups; i += kP2AggBlock) {
      const uint32_t c = gcnt[i];
      if (c == 0) continue;
      gcnt[i] = 0;
      my_pairs += c;
      const uint64_t hi = q.direct ? ((coarse << hi_rest) | i) : bij_unmix((coarse << hi_rest) | i, q.hi_bits);
      const uint64_t e = global_entry_raw(q.dense, q.capacity, q.g_keys, q.overflow, hi);
      if (e != ~0ull) {
        // (no NULL bit in x in this form: every pair counts towards the distinct values)
        atomicAdd((unsigned long long*)&q.g_state[e], (unsigned long long)c);
        atomicAdd((unsigned long long*)&q.g_state[(q.capacity + 1) + e], (unsigned long long)c);
      }
    }
  }
  // the distinct pairs of the whole query, for the sizing of the next run's fine buckets: one
  // update per workgroup, spread over eight words
  if (q.pairs) {
    __shared__ unsigned long long wg_pairs;
    __syncthreads();
    if (tid == 0) wg_pairs = 0;
    __syncthreads();
    const unsigned long long w = wave_sum_u64(my_pairs);
    if ((tid & 63u) == 0 && w) atomicAdd(&wg_pairs, w);
    __syncthreads();
    if (tid == 0 && wg_pairs) atomicAdd(&q.pairs[blockIdx.x & 7u], wg_pairs);
  }
}

// ---- a COUNT(DISTINCT) sharded over several GPUs: the exchange of the distinct records ----
// Fine bucket (c, sub) of the query belongs to rank c % world.  In a rank's send buffer the
// buckets lie owner by owner, and inside an owner's block in the order of the owner's own
// numbering j = (c / world) * n_sub + sub, so that what an owner receives from a rank is its
// buckets in order.
__device__ __forceinline__ uint32_t p2_owner_major(uint32_t f, uint32_t p1_bits, uint32_t p2_bits, uint32_t world) {
  const uint32_t n_sub = 1u << p2_bits, n_coarse = 1u << p1_bits;
  const uint32_t c = f >> p2_bits, sub = f & (n_sub - 1), owner = c % world;
  // coarse buckets of the owners before this one: owner o has ceil((n_coarse - o) / world)
  const uint32_t before = owner * (n_coarse / world) + (owner < n_coarse % world ? owner : n_coarse % world);
  return (before + c / world) * n_sub + sub;
}

// counts in owner-major order (a 64-bit word each: they travel through the all-to-all of words)
__global__ __launch_bounds__(256) void k_shard_counts(const uint32_t* fine_count, unsigned long long* counts64, uint32_t n_fine,
                                                       uint32_t p1_bits, uint32_t p2_bits, uint32_t world) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f < n_fine) counts64[p2_owner_major(f, p1_bits, p2_bits, world)] = fine_count[f];
}

// exclusive scan of n 64-bit counts (one workgroup); offsets[n] receives the total
__global__ __launch_bounds__(1024) void k_scan_u64(const unsigned long long* counts, unsigned long long* offsets, uint32_t n) {
  __shared__ unsigned long long wave_tot[16];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const unsigned long long v = i < n ? counts[i] : 0ull;
    unsigned long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long o = __shfl_up(incl, off, 64);
      if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wave_tot[w];
    const unsigned long long c = carry;
    if (i < n) offsets[i] = c + before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) offsets[n] = carry;
}

// the distinct records of every fine bucket, copied to their place in the send buffer
__global__ __launch_bounds__(256) void k_shard_gather(const unsigned long long* records, const uint32_t* fine_base, const uint32_t* fine_count,
                                                       const unsigned long long* send_offsets, unsigned long long* send, uint32_t n_fine,
                                                       uint32_t p1_bits, uint32_t p2_bits, uint32_t world) {
  for (uint32_t f = blockIdx.x; f < n_fine; f += gridDim.x) {
    const uint32_t n = fine_count[f];
    const unsigned long long* src = records + fine_base[f];
    unsigned long long* dst = send + send_offsets[p2_owner_major(f, p1_bits, p2_bits, world)];
    for (uint32_t i = threadIdx.x; i < n; i += 256) dst[i] = __builtin_nontemporal_load((P2WordPtr)(uint64_t)(src + i));
  }
}

// the runs an owner received, per sending rank: seg_cnt[src][j] from the counts, seg_base[src][j]
// their places in the receive buffer (rank src's records start at recv_start[src], buckets in order)
__global__ __launch_bounds__(1024) void k_shard_runs(const unsigned long long* recv_counts64, const unsigned long long* recv_start, uint32_t* seg_base,
                                                      uint32_t* seg_cnt, uint32_t n_own) {
  // one workgroup per sending rank: a scan over its n_own counts
  __shared__ unsigned long long wave_tot[16];
  __shared__ unsigned long long carry;
  const uint32_t src = blockIdx.x;
  const unsigned long long* counts = recv_counts64 + (uint64_t)src * n_own;
  if (threadIdx.x == 0) carry = recv_start[src];
  __syncthreads();
  for (uint32_t base = 0; base < n_own; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const unsigned long long v = i < n_own ? counts[i] : 0ull;
    unsigned long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long o = __shfl_up(incl, off, 64);
      if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wave_tot[w];
    const unsigned long long c = carry;
    if (i < n_own) {
      seg_base[(uint64_t)src * n_own + i] = (uint32_t)(c + before + incl - v);
      seg_cnt[(uint64_t)src * n_own + i] = (uint32_t)v;
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + before + incl;
    __syncthreads();
  }
}

}  // namespace

uint64_t part2_fixed_capacity(uint64_t n_rows, uint32_t n_fine) { return 2 * n_rows + (uint64_t)kP2Margin * n_fine + 64; }

uint32_t part2_lds_entries(bool compact) {
  // a workgroup's share of the LDS next to its static arrays (group counters), in 16-byte
  // pairs or (compact form) 8-byte records
  return (uint32_t)((kP2LdsBytes / kP2AggPerCu - kP2Groups * 16 - 256) / (compact ? 8 : 16)) & ~63u;
}

int part2_keys_ahead(const AggParams& p) { return part_keys_ahead(p); }
uint32_t part2_piece_chunks(uint32_t piece_tiles) { return (uint32_t)(2ull * piece_tiles * kTileRows / kP2SpillChunk + 4); }
bool part2_lines_fit(uint32_t p2_bits, uint32_t piece_tiles) {
  return p2_lines_lds(p2_bits, part2_piece_chunks(piece_tiles)) <= kP2LdsBytes;
}

void launch_distinct_second_level(prism_gpu_ctx* ctx, const Part2Params& q, const AggParams* scan) {
  if (q.n_rows == 0) return;
  if (q.direct && (!scan || !q.compact || (q.fine_limit && !q.spill_rec) || q.n_tiles == 0 || q.slice_tiles == 0 || (q.slice_tiles & 1u)))
    fail(PRISM_GPU_EINTERNAL, "bad shape for a second partition level that reads the scan's columns");
  if (q.spill_rec) {
    const uint64_t regions = (uint64_t)q.n_regions * q.region_cap;
    if (!q.direct || !q.fine_limit || !q.spill_fine || !q.spill_cursor || !q.spill_count || !q.spill_base || !q.spill_cur || !q.seg_base_out ||
        !q.seg_cnt_out || q.n_regions == 0 || q.n_regions > (1u << (q.p1_bits + q.p2_bits)) || regions + q.spill_capacity >= (1ull << 32) ||
        q.out_capacity < regions + q.spill_capacity)
      fail(PRISM_GPU_EINTERNAL, "bad shape for fine buckets with a spill list");
  }
  const uint32_t n_fine = 1u << (q.p1_bits + q.p2_bits);
  if (q.p2_bits > 10 || q.p1_bits + q.p2_bits > 21 || q.n_rows >= (1ull << 32) || q.slice_rows == 0)
    fail(PRISM_GPU_EINTERNAL, "bad shape for the second partition level");
  if (q.lds_entries > part2_lds_entries(q.compact != 0)) fail(PRISM_GPU_EINTERNAL, "the pair table exceeds LDS");
  if (q.compact && (q.hi_bits < (int)q.p1_bits || q.hi_bits - (int)q.p1_bits > 9 || q.lo_bits < (int)q.p2_bits || q.lo_bits > 63 ||
                    q.lo_bits - (int)q.p2_bits + q.hi_bits - (int)q.p1_bits > 63 || q.x_null_bit >= 0))
    fail(PRISM_GPU_EINTERNAL, "bad shape for one-word records");
  if (!q.spill_rec && (q.fine_limit ? (q.out_capacity < part2_fixed_capacity(q.n_rows, n_fine) || q.out_capacity >= (1ull << 32)) : q.out_capacity < q.n_rows))
    fail(PRISM_GPU_EINTERNAL, "the second partition level's output array is too short");
  const uint32_t slices = q.direct ? (uint32_t)ceil_div(q.n_tiles, (uint64_t)q.slice_tiles) : (uint32_t)ceil_div(q.n_rows, q.slice_rows);
  AggParams none;
  std::memset(&none, 0, sizeof(none));
  const AggParams& ap = scan ? *scan : none;
  // (key columns without validity bitmaps, at most one more than the key has words: their loads run a trip ahead)
  int ahead = q.direct ? part_keys_ahead(ap) : 0;
  if (ahead != 2 && ahead != 3) ahead = 0;
  if (q.spill_rec) {
    KernelTimer timer(ctx, "part2_layout", 0, true);
    k_part2_layout_uniform<<<dim3((uint32_t)ceil_div(n_fine + 1, 256)), dim3(256), 0, ctx->stream>>>(q, n_fine);
    PRISM_HIP(hipGetLastError());
  } else if (q.fine_limit) {
    KernelTimer timer(ctx, "part2_layout", 0, true);
    k_part2_layout<<<dim3((uint32_t)ceil_div(n_fine + 1, 256)), dim3(256), 0, ctx->stream>>>(q, n_fine);
    PRISM_HIP(hipGetLastError());
  } else {
    KernelTimer timer(ctx, "part2_hist", q.n_rows * 16, true);
    if (q.direct) {
      auto kernel = ahead == 3 ? k_part2_hist_direct<3> : ahead == 2 ? k_part2_hist_direct<2> : k_part2_hist_direct<0>;
      kernel<<<dim3(slices), dim3(kP2Block), 0, ctx->stream>>>(q, ap);
    } else {
      k_part2_hist<<<dim3(slices), dim3(kP2Block), 0, ctx->stream>>>(q);
    }
    PRISM_HIP(hipGetLastError());
    k_part2_scan<<<dim3(1), dim3(1024), 0, ctx->stream>>>(q, n_fine);
    PRISM_HIP(hipGetLastError());
  }
  if (q.lines) {
    if (!q.spill_rec || !q.direct || (ahead != 2 && ahead != 3) || !q.bounds || !q.pieces || !q.n_pieces || !q.pieces_too_long || !q.rare || q.piece_tiles == 0 || q.piece_max == 0 ||
        q.piece_max > 63 || q.max_pieces == 0 || q.n_tiles >= (1ull << 32))
      fail(PRISM_GPU_EINTERNAL, "bad shape for the scatter pass by staging lines");
    {
      KernelTimer timer(ctx, "part2_pieces", 0, true);
      k_part2_bounds<<<dim3((1u << q.p1_bits) + 1), dim3(64), 0, ctx->stream>>>(q, ap);
      k_part2_pieces<<<dim3(1), dim3(1024), 0, ctx->stream>>>(q);
      PRISM_HIP(hipGetLastError());
    }
    KernelTimer timer(ctx, "part2_scatter", q.n_rows * 24, true);
    const size_t lds = p2_lines_lds(q.p2_bits, q.piece_chunks);
    if (q.piece_chunks < 2ull * q.piece_tiles * kTileRows / kP2SpillChunk + 4 || lds > kP2LdsBytes) fail(PRISM_GPU_EINTERNAL, "bad spill chunk table for the scatter pass by staging lines");
    bool bin64 = false;  // a date_bin over a span of 2^32 and more among the keys
    for (int k = 0; k < ap.n_keys; ++k) bin64 = bin64 || (ap.keys[k].date_bin && !ap.keys[k].fast32);
    auto kernel = ahead == 3 ? (bin64 ? k_part2_lines<3, true> : k_part2_lines<3, false>) : (bin64 ? k_part2_lines<2, true> : k_part2_lines<2, false>);
    PRISM_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kernel<<<dim3(q.max_pieces), dim3(kP2Block), lds, ctx->stream>>>(q, ap);
    PRISM_HIP(hipGetLastError());
  } else {
    KernelTimer timer(ctx, "part2_scatter", q.n_rows * (q.compact ? 24 : 32), true);
    // (the columns' form can run two workgroups of 512 threads a CU, option "part2_half_blocks"; measured
    // slower at C5, 13.5-14.4 against 13.1-13.2 ms, and shorter runs spill more: off by default)
    const bool halves = q.direct && ctx->part2_half_blocks.load(std::memory_order_relaxed) != 0;
    const uint32_t block = halves ? 512u : kP2Block;
    const size_t lds = (size_t)block * kP2TileRows * 16 + (size_t)(3 * 2 * block + 1) * 4;
    auto kernel = !q.direct ? k_part2_scatter<false, 0, kP2Block>
                  : halves  ? (ahead == 3 ? k_part2_scatter<true, 3, 512> : ahead == 2 ? k_part2_scatter<true, 2, 512> : k_part2_scatter<true, 0, 512>)
                            : (ahead == 3 ? k_part2_scatter<true, 3, kP2Block> : ahead == 2 ? k_part2_scatter<true, 2, kP2Block>
                                                                                         : k_part2_scatter<true, 0, kP2Block>);
    PRISM_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kernel<<<dim3(slices), dim3(block), lds, ctx->stream>>>(q, ap);
    PRISM_HIP(hipGetLastError());
  }
  if (q.spill_rec) {
    // the spill list sorted by bucket behind the regions; then every bucket as its two runs
    const uint64_t out_offset = (uint64_t)q.n_regions * q.region_cap;
    {
      KernelTimer timer(ctx, "part2_spill", 0, true);
      const uint32_t grid = ctx->n_cus * 16;
      k_spill_count<<<dim3(grid), dim3(256), 0, ctx->stream>>>(q);
      Part2Params sc = q;  // (the scan of the fine buckets' counts, over the spill list's)
      sc.fine_count = q.spill_count;
      sc.fine_base = q.spill_base;
      sc.fine_cursor = q.spill_cur;
      k_part2_scan<<<dim3(1), dim3(1024), 0, ctx->stream>>>(sc, q.n_regions);
      k_spill_place<<<dim3(grid), dim3(256), 0, ctx->stream>>>(q, out_offset);
      if (q.lines) k_spill_runs_lines<<<dim3((uint32_t)ceil_div(q.n_regions, 256)), dim3(256), 0, ctx->stream>>>(q, out_offset);
      else k_spill_runs<<<dim3((uint32_t)ceil_div(q.n_regions, 256)), dim3(256), 0, ctx->stream>>>(q, out_offset);
      PRISM_HIP(hipGetLastError());
    }
    Part2Params m = q;
    m.n_src = q.lines ? q.piece_max + 1 : 2;
    m.world = 1;
    m.rank = 0;
    m.seg_base = q.seg_base_out;
    m.seg_cnt = q.seg_cnt_out;
    KernelTimer timer(ctx, "part2_agg", q.n_rows * 8, true);
    const size_t lds = (size_t)q.lds_entries * 8;
    auto kernel = q.emit_out ? k_part2_agg_compact<kP2EmitRuns> : k_part2_agg_compact<kP2Merge>;
    PRISM_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(q.n_regions, (uint64_t)ctx->n_cus * kP2AggPerCu);
    kernel<<<dim3(grid), dim3(kP2AggBlock), lds, ctx->stream>>>(m, q.n_regions);
    PRISM_HIP(hipGetLastError());
    return;
  }
  {
    KernelTimer timer(ctx, "part2_agg", q.n_rows * (q.compact ? 8 : 16), true);
    const size_t lds = (size_t)q.lds_entries * (q.compact ? 8 : 16);
    if (q.emit_out && !q.compact) fail(PRISM_GPU_EINTERNAL, "distinct records are emitted in the one-word form only");
    auto kernel = q.compact ? (q.emit_out ? k_part2_agg_compact<kP2Emit> : k_part2_agg_compact<kP2Count>) : k_part2_agg;
    PRISM_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(n_fine, (uint64_t)ctx->n_cus * kP2AggPerCu);
    kernel<<<dim3(grid), dim3(kP2AggBlock), lds, ctx->stream>>>(q, n_fine);
    PRISM_HIP(hipGetLastError());
  }
}

// ---- COUNT(DISTINCT) sharded over several GPUs: launch wrappers of the exchange's kernels ----

void launch_shard_counts(prism_gpu_ctx* ctx, const uint32_t* fine_count, unsigned long long* counts64, uint32_t n_fine, uint32_t p1_bits,
                         uint32_t p2_bits, uint32_t world) {
  k_shard_counts<<<dim3((uint32_t)ceil_div(n_fine, 256)), dim3(256), 0, ctx->stream>>>(fine_count, counts64, n_fine, p1_bits, p2_bits, world);
  PRISM_HIP(hipGetLastError());
}

void launch_scan_u64(prism_gpu_ctx* ctx, const unsigned long long* counts, unsigned long long* offsets, uint32_t n) {
  k_scan_u64<<<dim3(1), dim3(1024), 0, ctx->stream>>>(counts, offsets, n);
  PRISM_HIP(hipGetLastError());
}

void launch_shard_gather(prism_gpu_ctx* ctx, const unsigned long long* records, const uint32_t* fine_base, const uint32_t* fine_count,
                         const unsigned long long* send_offsets, unsigned long long* send, uint32_t n_fine, uint32_t p1_bits,
                         uint32_t p2_bits, uint32_t world) {
  KernelTimer timer(ctx, "shard_gather", 0);
  const uint32_t grid = (uint32_t)std::min<uint64_t>(n_fine, (uint64_t)ctx->n_cus * 16);
  k_shard_gather<<<dim3(grid), dim3(256), 0, ctx->stream>>>(records, fine_base, fine_count, send_offsets, send, n_fine, p1_bits, p2_bits, world);
  PRISM_HIP(hipGetLastError());
}

void launch_shard_runs(prism_gpu_ctx* ctx, const unsigned long long* recv_counts64, const unsigned long long* recv_start, uint32_t* seg_base,
                       uint32_t* seg_cnt, uint32_t n_own, uint32_t world) {
  k_shard_runs<<<dim3(world), dim3(1024), 0, ctx->stream>>>(recv_counts64, recv_start, seg_base, seg_cnt, n_own);
  PRISM_HIP(hipGetLastError());
}

void launch_distinct_merge(prism_gpu_ctx* ctx, const Part2Params& q, uint32_t n_own) {
  if (!q.compact || q.n_src == 0 || q.n_src > 64 || !q.seg_base || !q.seg_cnt || q.world == 0 || q.rank >= q.world)
    fail(PRISM_GPU_EINTERNAL, "bad shape for the merge of distinct records");
  if (q.lds_entries > part2_lds_entries(true)) fail(PRISM_GPU_EINTERNAL, "the pair table exceeds LDS");
  if (n_own == 0) return;
  KernelTimer timer(ctx, "part2_merge", 0);
  const size_t lds = (size_t)q.lds_entries * 8;
  PRISM_HIP(hipFuncSetAttribute((const void*)k_part2_agg_compact<kP2Merge>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const uint32_t grid = (uint32_t)std::min<uint64_t>(n_own, (uint64_t)ctx->n_cus * kP2AggPerCu);
  k_part2_agg_compact<kP2Merge><<<dim3(grid), dim3(kP2AggBlock), lds, ctx->stream>>>(q, n_own);
  PRISM_HIP(hipGetLastError());
}

}  // namespace prism
