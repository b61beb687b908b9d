// Device helpers shared by the partitioning kernels (kernels_part.hip, kernels_wc.hip): a
// workgroup's walk over the scan tiles, column loads requested a tile ahead, packed keys and
// records, a block-wide scan.
#pragma once

#include "agg_device.hpp"

namespace prism {

namespace {

constexpr uint32_t kPartBlock = 1024;
constexpr int kPartAggInFlight = 4;

typedef unsigned long long __attribute__((aligned(1))) part_unaligned_u64;
typedef const __attribute__((address_space(1))) unsigned char* PartBytePtr;
typedef const __attribute__((address_space(1))) unsigned long long* PartWordPtr;

__device__ __forceinline__ uint64_t part_uniform(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// A fixed-width value from the 8 bytes loaded at its address (the bytes beyond it belong to
// the next rows or to the allocation's padding).  The width is a plan fact: scalar branches.
__device__ __forceinline__ long long part_widen(unsigned long long raw, int width, int is_unsigned) {
  if (width == 8) return (long long)raw;
  if (width == 4) return is_unsigned ? (long long)(uint32_t)raw : (long long)(int32_t)(uint32_t)raw;
  return is_unsigned ? (long long)(uint16_t)raw : (long long)(int16_t)(uint16_t)raw;
}

// log2 of a column's width in bytes (2, 4 or 8): a row's byte offset is a shift, where a multiply
// by the width -- a run-time value to the compiler -- is a quarter-rate instruction per row and column.
__device__ __forceinline__ uint32_t part_width_shift(int width) { return width == 8 ? 3u : (width == 4 ? 2u : 1u); }

// A key column's field of the packed key, before it is shifted into place: value - bias, or
// for date_bin the bucket number - bias.  Keys whose values span less than 2^32 (KeySpec::
// fast32) take the 32-bit form, (uint32)(value - base) / stride with a 32-bit multiply-high: the
// 64-bit date_bin is some 25 instructions a row, and these kernels run it once per pass.
__device__ __forceinline__ uint64_t part_field(const KeySpec& ks, unsigned long long raw) {
  if (ks.fast32) {
    uint32_t f = (uint32_t)part_widen(raw, ks.width, ks.is_unsigned) - (uint32_t)ks.base;
    if (ks.date_bin) {
      const uint32_t q = __umulhi(ks.magic32, f);
      f = (((f - q) >> ks.pre32) + q) >> ks.shift32;
    }
    return f;
  }
  long long v = part_widen(raw, ks.width, ks.is_unsigned);
  if (ks.date_bin) v = bucket_number(v, ks);
  return (uint64_t)v - (uint64_t)ks.bias;
}

template <int KW>
__device__ __forceinline__ uint64_t part_hash(uint64_t lo, uint64_t hi) {
  return KW == 2 ? hash64(lo ^ hash64(hi)) : hash64(lo);
}

// The hash a row's bucket is taken from (its top bits): for one-word keys 32 bits mixed with
// three 32-bit multiplies (the two 64-bit multiplies of hash64 are eight), for two-word keys
// the whole pair, or (pairs of a COUNT(DISTINCT) that is partitioned twice) the high word
// alone, so that a group's pairs share a bucket -- hashed, or mixed by a bijection of its
// hi_bits bits when the second level stores only what its buckets do not imply.
// (by_hi with one-word keys: the histogram pass of such a partition, run over the high word's
// columns alone, whose packed value then is the one word.)
// (HIST: only the histogram kernel of one-word keys looks at by_hi; the other one-word
// kernels never see such a partition and are spared the branch and its registers.)
template <int KW, bool HIST = false>
__device__ __forceinline__ uint64_t part_bucket_hash(const PartParams& q, uint64_t lo, uint64_t hi) {
  if ((KW == 2 || HIST) && q.by_hi) {
    const uint64_t group = KW == 2 ? hi : lo;
    return q.hi_bits > 0 ? bij_mix(group, q.hi_bits) << (64 - q.hi_bits) : part_hash_hi(group);
  }
  if (KW == 1) return (uint64_t)bucket_hash32(lo) << 32;
  return part_hash<KW>(lo, hi);
}

// One scan tile (kTileRows rows of one segment) as a workgroup sees it; wave uniform.
struct PartTile {
  uint32_t seg;   // segment the tile lies in
  uint32_t n;     // rows of the tile that exist (0: the tile is beyond the workgroup's share)
  uint64_t row0;  // first row of the tile inside its segment
  uint64_t tile;  // the tile's number (selection bitmaps are indexed by it)
};

// A workgroup takes its tiles in rising order: the segment of the last tile is remembered
// (wave uniform), and the table of segments is only read again when a tile lies beyond it.
// (Searching the table for every tile is a chain of dependent scalar loads from device
// memory, microseconds per tile when the memory system is busy.)
struct PartSegPos {
  uint32_t seg;
  uint64_t base, rows, next_base;  // first tile and rows of the segment; first tile of the next one
};

__device__ __forceinline__ void part_seg_load(const AggParams& p, PartSegPos& c, uint32_t seg) {
  c.seg = seg;
  c.base = part_uniform(p.sd.segs[seg].tile_base);
  c.rows = part_uniform(p.sd.segs[seg].rows);
  c.next_base = seg + 1 < p.sd.n_segs ? part_uniform(p.sd.segs[seg + 1].tile_base) : ~0ull;
}

__device__ __forceinline__ void part_seg_init(const AggParams& p, PartSegPos& c, uint64_t tile) {
  const uint64_t last = p.sd.n_tiles ? p.sd.n_tiles - 1 : 0;
  part_seg_load(p, c, __builtin_amdgcn_readfirstlane(find_seg(p.sd, tile < last ? tile : last)));
}

__device__ __forceinline__ PartTile part_tile_next(const AggParams& p, PartSegPos& c, uint64_t tile, uint64_t t1) {
  PartTile r;
  r.seg = c.seg; r.n = 0; r.row0 = 0; r.tile = 0;
  if (tile >= t1) return r;
  while (tile >= c.next_base) part_seg_load(p, c, c.seg + 1);
  r.seg = c.seg;
  r.row0 = (tile - c.base) * kTileRows;
  const uint64_t left = c.rows - r.row0;
  r.n = left > kTileRows ? kTileRows : (uint32_t)left;
  r.tile = tile;
  return r;
}

// Packed keys of the R rows a thread owns in R / 4 consecutive scan tiles (rows tid,
// tid + 1024, ... of each).  Every lane loads, rows that do not exist or are not selected
// load a row that does and are dropped by `live` (bit r).  All loads of one key column are
// issued before the first is used.
// skip: key columns (bit k) that are not loaded; their bits of the packed key stay zero (the
// histogram of a partition by the high word has no use for the columns of the low word).
// (BLOCK threads a workgroup: a thread owns kTileRows / BLOCK rows of a scan tile.)
template <int KW, int R, int BLOCK = (int)kPartBlock>
__device__ __forceinline__ void part_load_keys(const AggParams& p, const PartTile (&tl)[R / (int)(kTileRows / BLOCK)], uint32_t tid,
                                               uint64_t (&lo)[R], uint64_t (&hi)[R], uint32_t (&rowc)[R], uint32_t& live,
                                               uint32_t skip = 0) {
  constexpr int RT = (int)(kTileRows / BLOCK);  // rows of a scan tile per thread
  live = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const PartTile& t = tl[r / RT];
    const uint32_t in_tile = (uint32_t)(r % RT) * BLOCK + tid;
    bool in = in_tile < t.n;
    rowc[r] = (uint32_t)t.row0 + (in ? in_tile : 0u);
    if (p.row_mode == kRowsBitmap) {
      const uint64_t pid = t.tile * kTileRows + in_tile;
      const uint64_t word = in ? p.sel_bits[pid >> 6] : 0ull;
      in = in && ((word >> (pid & 63)) & 1ull);
    }
    live |= (in ? 1u : 0u) << r;
    lo[r] = 0;
    hi[r] = 0;
  }
  // at most 8 loads of a column in flight per lane (16 at a time cost more registers than a
  // 1024-thread workgroup has)
  constexpr int B = R < 8 ? R : 8;
  for (int k = 0; k < p.n_keys; ++k) {
    if ((skip >> k) & 1u) continue;
    const KeySpec& ks = p.keys[k];
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += B) {
      unsigned long long raw[B];
      uint32_t ok = 0xffffffffu;  // bit i: the key of row r0 + i is not NULL
#pragma unroll
      for (int i = 0; i < B; ++i) {
        const int r = r0 + i;
        const ColPtr& c = p.cols[(uint64_t)ks.col * p.sd.n_segs + tl[r / RT].seg];
        const PartBytePtr base = (PartBytePtr)(uint64_t)c.values;
        // (a chunk's column is shorter than 4 GiB: the byte offset fits 32 bits, and the load takes
        // its base from scalar registers)
        raw[i] = __builtin_nontemporal_load((const __attribute__((address_space(1))) part_unaligned_u64*)(base + (rowc[r] << part_width_shift(ks.width))));
      }
      if (ks.null_bit >= 0) {
#pragma unroll
        for (int i = 0; i < B; ++i) {
          const int r = r0 + i;
          const ColPtr& c = p.cols[(uint64_t)ks.col * p.sd.n_segs + tl[r / RT].seg];
          if (!valid_at(c.validity, rowc[r])) ok &= ~(1u << i);
        }
      }
#pragma unroll
      for (int i = 0; i < B; ++i) {
        const int r = r0 + i;
        const uint64_t field = ((ok >> i) & 1u) ? (part_field(ks, raw[i]) << ks.bit_pos) : (1ull << (ks.null_bit >= 0 ? ks.null_bit : 0));
        if (KW == 2 && ks.word) hi[r] |= field; else lo[r] |= field;
      }
    }
  }
}

// Plans whose key columns (at most one per key word) carry no validity bitmap load their keys
// one super-tile ahead: the raw column words are requested here and turned into packed keys
// (part_finish_keys) an iteration later, so the loads are in flight while the tile before is
// ranked, sorted and copied out.
// Where a workgroup stands in the scan when it takes its tiles in order: the segment of the
// last tile and the column bases there, all in scalar registers.  (Looking a tile's segment up
// and fetching the column pointers for it is a chain of eight dependent scalar loads from
// device memory; per tile, that chain was a quarter of the scatter pass.)
template <int NC>
struct PartCursor {
  uint32_t seg;
  uint64_t base, rows, next_base;  // first tile and rows of the segment; first tile of the next one
  PartBytePtr col[NC];             // values of the key columns, then of the value columns
};

template <int NK, int NV>
__device__ __forceinline__ void part_cursor_load(const AggParams& p, PartCursor<NK + NV>& c, uint32_t seg) {
  c.seg = seg;
  c.base = part_uniform(p.sd.segs[seg].tile_base);
  c.rows = part_uniform(p.sd.segs[seg].rows);
  c.next_base = seg + 1 < p.sd.n_segs ? part_uniform(p.sd.segs[seg + 1].tile_base) : ~0ull;
#pragma unroll
  for (int k = 0; k < NK; ++k)
    c.col[k] = (PartBytePtr)part_uniform((uint64_t)p.cols[(uint64_t)p.keys[k].col * p.sd.n_segs + seg].values);
#pragma unroll
  for (int v = 0; v < NV; ++v)
    c.col[NK + v] = (PartBytePtr)part_uniform((uint64_t)p.cols[(uint64_t)p.vals[v].col * p.sd.n_segs + seg].values);
}

template <int NK, int NV>
__device__ __forceinline__ void part_cursor_init(const AggParams& p, PartCursor<NK + NV>& c, uint64_t tile) {
  const uint64_t last = p.sd.n_tiles ? p.sd.n_tiles - 1 : 0;
  part_cursor_load<NK, NV>(p, c, __builtin_amdgcn_readfirstlane(find_seg(p.sd, tile < last ? tile : last)));
}

// The loads of group G of the super-tile that starts at scan tile `tile0`: four rows a thread
// (rows 4 G .. 4 G + 3 of the thread's R), a whole scan tile for workgroups of 1024 threads,
// half of one for 512.  Tiles are asked for in rising order; a tile at or beyond t1 has no
// rows (its lanes load row 0 of the cursor's segment and are dropped by `live`).
template <int NK, int NV, int R, int G, int BLOCK = (int)kPartBlock>
__device__ __forceinline__ void part_issue_group(const AggParams& p, PartCursor<NK + NV>& c, uint64_t tile0, uint64_t t1, uint32_t tid,
                                                 unsigned long long (&raw)[NK + NV][R], uint32_t& live, uint32_t skip = 0) {
  constexpr int GPT = (int)(kTileRows / (4 * BLOCK));  // groups per scan tile
  const uint64_t tile = tile0 + (uint64_t)(G / GPT);
  constexpr uint32_t in_base = (uint32_t)(G % GPT) * 4u * BLOCK;
  uint32_t n = 0, row0 = 0;
  if (tile < t1) {
    while (tile >= c.next_base) part_cursor_load<NK, NV>(p, c, c.seg + 1);
    const uint64_t first = (tile - c.base) * kTileRows, left = c.rows - first;
    row0 = (uint32_t)first;
    n = left > kTileRows ? kTileRows : (uint32_t)left;
  }
  uint32_t rowc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t in_tile = in_base + (uint32_t)i * BLOCK + tid;
    bool in = in_tile < n;
    rowc[i] = row0 + (in ? in_tile : 0u);
    if (p.row_mode == kRowsBitmap) {
      const uint64_t pid = tile * kTileRows + in_tile;
      const uint64_t word = in ? p.sel_bits[pid >> 6] : 0ull;
      in = in && ((word >> (pid & 63)) & 1ull);
    }
    live |= (in ? 1u : 0u) << (4 * G + i);
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    if ((skip >> k) & 1u) continue;  // (wave uniform; the words keep what they held)
    const uint32_t sh = part_width_shift(p.keys[k].width);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      raw[k][4 * G + i] = __builtin_nontemporal_load((const __attribute__((address_space(1))) part_unaligned_u64*)(c.col[k] + (rowc[i] << sh)));
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const uint32_t sh = part_width_shift(p.vals[v].width);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      raw[NK + v][4 * G + i] = __builtin_nontemporal_load((const __attribute__((address_space(1))) part_unaligned_u64*)(c.col[NK + v] + (rowc[i] << sh)));
  }
}

// All the loads of the super-tile that starts at `tile`.
template <int NK, int NV, int R, int BLOCK = (int)kPartBlock>
__device__ __forceinline__ void part_issue(const AggParams& p, PartCursor<NK + NV>& c, uint64_t tile, uint64_t t1, uint32_t tid,
                                           unsigned long long (&raw)[NK + NV][R], uint32_t& live, uint32_t skip = 0) {
  live = 0;
  part_issue_group<NK, NV, R, 0, BLOCK>(p, c, tile, t1, tid, raw, live, skip);
  if constexpr (R >= 8) part_issue_group<NK, NV, R, 1, BLOCK>(p, c, tile, t1, tid, raw, live, skip);
  if constexpr (R >= 16) {
    part_issue_group<NK, NV, R, 2, BLOCK>(p, c, tile, t1, tid, raw, live, skip);
    part_issue_group<NK, NV, R, 3, BLOCK>(p, c, tile, t1, tid, raw, live, skip);
  }
}

// The words requested by part_issue as a record: packed key word(s), then the widened values.
// BIN64 = false: the caller has checked that no key is a date_bin over a span of 2^32 and more (its
// constants -- a 64-bit magic number, origin and stride per key -- then stay out of the registers).
template <int KW, int NK, int NV, int R, bool BIN64 = true>
__device__ __forceinline__ void part_finish(const AggParams& p, const unsigned long long (&raw)[NK + NV][R],
                                            uint64_t (&rec)[KW + NV][R]) {
  uint64_t lo[R], hi[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { lo[r] = 0; hi[r] = 0; }
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const KeySpec& ks = p.keys[k];
    // (the choice between the 32-bit and the 64-bit form is made once per column, outside the
    // rows: inside, both forms' temporaries were live and cost the one-word scatter kernel spills)
    if (ks.fast32) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        uint32_t f = (uint32_t)part_widen(raw[k][r], ks.width, ks.is_unsigned) - (uint32_t)ks.base;
        if (ks.date_bin) {
          const uint32_t q = __umulhi(ks.magic32, f);
          f = (((f - q) >> ks.pre32) + q) >> ks.shift32;
        }
        const uint64_t field = (uint64_t)f << ks.bit_pos;
        if (KW == 2 && ks.word) hi[r] |= field; else lo[r] |= field;
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        long long v = part_widen(raw[k][r], ks.width, ks.is_unsigned);
        if (BIN64 && ks.date_bin) v = bucket_number(v, ks);
        const uint64_t field = ((uint64_t)v - (uint64_t)ks.bias) << ks.bit_pos;
        if (KW == 2 && ks.word) hi[r] |= field; else lo[r] |= field;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const ValSpec& vs = p.vals[v];
#pragma unroll
    for (int r = 0; r < R; ++r) rec[KW + v][r] = (uint64_t)part_widen(raw[NK + v][r], vs.width, vs.is_unsigned);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    rec[0][r] = lo[r];
    if (KW == 2) rec[1][r] = hi[r];
  }
}

// The aggregate inputs of the same rows, widened to 64 bits; validw[r] bit v = value v is not NULL.
template <int NV, bool VALID, int R, int BLOCK = (int)kPartBlock>
__device__ __forceinline__ void part_load_vals(const AggParams& p, const PartTile (&tl)[R / (int)(kTileRows / BLOCK)], const uint32_t (&rowc)[R],
                                               uint64_t (&val)[NV > 0 ? NV : 1][R], uint64_t (&validw)[R]) {
  constexpr int RT = (int)(kTileRows / BLOCK);
#pragma unroll
  for (int r = 0; r < R; ++r) validw[r] = 0;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const ValSpec& vs = p.vals[v];
    unsigned long long raw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const ColPtr& c = p.cols[(uint64_t)vs.col * p.sd.n_segs + tl[r / RT].seg];
      const PartBytePtr base = (PartBytePtr)(uint64_t)c.values;
      raw[r] = __builtin_nontemporal_load((const __attribute__((address_space(1))) part_unaligned_u64*)(base + (rowc[r] << part_width_shift(vs.width))));
      if (VALID) validw[r] |= (valid_at(c.validity, rowc[r]) ? 1ull : 0ull) << v;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) val[v][r] = (uint64_t)part_widen(raw[r], vs.width, vs.is_unsigned);
  }
}

// Exclusive scan of a[0..n) in place (n a power of two between 64 and 4 * kPartBlock), the
// total in a[n].  Block wide; the caller puts a barrier in front and behind.
template <uint32_t BLOCK = kPartBlock>
__device__ __forceinline__ void part_block_scan(uint32_t* a, uint32_t n, uint32_t* wave_tot) {
  const uint32_t tid = threadIdx.x, per = n >= BLOCK ? n / BLOCK : 1u;
  uint32_t v[4], sum = 0;
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) {
    const uint32_t e = tid * per + i;
    v[i] = (i < per && e < n) ? a[e] : 0u;
    sum += v[i];
  }
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, 64);
    if ((int)(tid & 63u) >= off) incl += o;
  }
  if ((tid & 63u) == 63u) wave_tot[tid >> 6] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < (tid >> 6); ++w) before += wave_tot[w];
  uint32_t run = before + incl - sum;
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) {
    const uint32_t e = tid * per + i;
    if (i < per && e < n) { a[e] = run; run += v[i]; }
  }
  if (tid == BLOCK - 1) a[n] = before + incl;
}

// Key columns the scan kernels may request ahead: all of them, when none is nullable and
// they are no more than the key words plus one (0 otherwise).
inline int part_keys_ahead(const AggParams& p) {
  if (p.n_keys < p.key_words || p.n_keys > p.key_words + 1) return 0;
  for (int k = 0; k < p.n_keys; ++k)
    if (p.keys[k].null_bit >= 0) return 0;
  return p.n_keys;
}

}  // namespace

}  // namespace prism
