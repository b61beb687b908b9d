// Device-side data descriptors and the host launch wrappers of every kernel.
//
// Layout recap.  A scan covers a list of segments (one per resident partition
// chunk).  Rows are addressed by a "padded row id": tile * kTileRows + r, where a
// tile never spans two segments.  Selection bitmaps hold one bit per padded row
// id (64 words per tile); selection vectors hold padded row ids.
#pragma once

#include <cstdint>

#include "common.hpp"
#include "context.hpp"

namespace prism {

struct ScanSeg {
  uint64_t rows;       // rows in this segment
  uint64_t tile_base;  // first tile of this segment
};

struct ColPtr {
  const void* values;        // fixed-width values, u32 dictionary codes, or i32 string offsets
  const uint64_t* validity;  // Arrow validity bits, bit 0 = first row of the segment; nullptr = no NULLs
  const uint8_t* bytes;      // string bytes (raw Utf8/Binary columns)
};

struct ScanDesc {
  const ScanSeg* segs;  // device array
  uint32_t n_segs;
  uint64_t n_tiles;
};

// ---- predicates -----------------------------------------------------------

enum CmpMask { kLt = 1, kEq = 2, kGt = 4 };  // which orderings satisfy the comparison

// column <op> literal over a fixed-width column -> selection bits (TRUE rows) and/or
// their count.  `bits` and `count` may each be nullptr.  (cmp_mask2, literal2): a second
// comparison of the same column, AND-ed in; the default accepts every value.
void launch_cmp_leaf(prism_gpu_ctx* ctx, const ScanDesc& sd, const ColPtr* col, int width, bool is_unsigned,
                     int cmp_mask, int64_t literal, uint64_t* bits, unsigned long long* count,
                     int cmp_mask2 = kLt | kEq | kGt, int64_t literal2 = 0);
// bits = rows whose dictionary code c has lut[c] != 0 (and are not NULL).
void launch_lut_leaf(prism_gpu_ctx* ctx, const ScanDesc& sd, const ColPtr* col, const uint8_t* lut, uint32_t lut_len,
                     uint64_t* bits);
// IS NULL / IS NOT NULL.
void launch_null_leaf(prism_gpu_ctx* ctx, const ScanDesc& sd, const ColPtr* col, bool want_null, uint64_t* bits);

// String predicates over raw (offsets + bytes) columns, one row per lane.
enum StrPredKind { kStrCmp = 1, kStrLike = 2 };
struct LikeToken {  // kind: 0 literal byte, 1 any one character, 2 any run
  uint8_t kind;
  uint8_t byte;
};
void launch_str_leaf(prism_gpu_ctx* ctx, const ScanDesc& sd, const ColPtr* col, int kind, int cmp_mask,
                     const uint8_t* literal /*device*/, uint32_t literal_len, const LikeToken* tokens /*device*/,
                     uint32_t n_tokens, bool negated, bool case_insensitive, uint64_t* bits);
// LIKE '%needle%' as a byte-stream search over one segment's string buffer: coalesced
// 16-byte loads, matches mapped back to rows through the offsets.  `bits` must be zeroed.
// '%needle%' over every segment of a raw string column in two passes: the first streams
// the string bytes of all segments in one launch and appends the byte positions where
// the needle occurs to a list; the second maps each position to its row (binary search
// in the segment's offsets; an occurrence that straddles two rows is none) and sets the
// row's bit.  Occurrences are rare next to bytes, so the first pass is a pure streaming
// read.  Returns false when the list overflowed (then nothing was marked and the caller
// falls back to launch_contains_stream per segment).
struct ContainsSeg {
  const int32_t* off;
  const uint8_t* bytes;
  uint64_t rows;
  uint64_t n_bytes;
  uint64_t tile_base;
  uint64_t chunk_base;  // chunks (of contains_chunk_bytes(contains_block(..))) before this segment
};
// Threads of a workgroup of the first pass, and the bytes of a chunk that go with them (four
// 16-byte pieces a thread).  Needles of four bytes and more (the window search): workgroups of
// ONE wave, 32 of them on a CU -- a wave then walks its chunks at its own pace and the loads
// of the CU's waves spread out in time, where four waves in step at a barrier per chunk asked
// for theirs in bursts (C4: 1.30 -> 1.21 ms for 6.5 GB of strings; 128 threads: 1.26).
// Shorter needles may occur all over (a one-letter needle: most chunks fill the list's LDS
// buffer several times) and keep the larger workgroup, whose larger buffer means four times
// fewer additions to the list's one counter (9.9 ms against 16.7 with single waves).
uint32_t contains_block(const prism_gpu_ctx* ctx, uint32_t needle_len);
constexpr uint64_t contains_chunk_bytes(uint32_t block) { return (uint64_t)block * 64; }
bool launch_contains_all(prism_gpu_ctx* ctx, const ContainsSeg* d_segs, uint32_t n_segs, uint64_t n_chunks,
                         const uint8_t* d_needle, uint32_t needle_len, uint64_t* d_matches, uint64_t match_capacity,
                         unsigned long long* d_n_matches, uint64_t* bits);
void launch_contains_stream(prism_gpu_ctx* ctx, const ColPtr& col, uint64_t rows, uint64_t tile_base,
                            const uint8_t* needle /*device*/, uint32_t needle_len, uint64_t* bits);
// bits &= validity (per segment), or bits = validity & ~bits when `negate`.
void launch_mask_validity(prism_gpu_ctx* ctx, const ScanDesc& sd, const ColPtr* col, bool negate, uint64_t* bits);

void launch_bitmap_and(prism_gpu_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n_words);
void launch_bitmap_or(prism_gpu_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n_words);
void launch_bitmap_count(prism_gpu_ctx* ctx, const uint64_t* bits, uint64_t n_words, unsigned long long* count);

// ---- selection-vector compaction -------------------------------------------
// bits -> ascending padded row ids, in two calls: with selvec == nullptr it fills
// tile_offsets (n_tiles + 1 u32: exclusive scan of the per-tile counts) and *total;
// with selvec set it writes the row ids using those offsets.
void launch_compact(prism_gpu_ctx* ctx, const uint64_t* bits, uint64_t n_tiles, uint32_t* tile_offsets,
                    uint32_t* selvec, unsigned long long* total);

// ---- aggregation -------------------------------------------------------------

constexpr int kMaxKeys = 4;
constexpr int kMaxVals = 16;
constexpr int kMaxSlots = 24;

struct KeySpec {
  int32_t col;        // index into AggParams::cols rows
  int32_t width;      // 2, 4, 8
  int32_t is_unsigned;
  int32_t date_bin;   // 1: value = bucket number of the timestamp
  int64_t stride;     // date_bin stride (ms)
  int64_t origin;     // date_bin origin (ms)
  uint64_t magic;     // division by `stride` as multiply-high + shifts
  int32_t magic_shift;
  int32_t bit_pos;    // position of this field in the packed key
  int64_t bias;       // field = value - bias
  int32_t null_bit;   // bit set for a NULL key, -1 when the column has no NULLs
  // 32-bit form of the field, field = (uint32)(value - base) / divisor, usable when the
  // packed key fits 31 bits: plain columns (base = bias, divisor 1) and date_bin when
  // every timestamp of the scan is at or after the origin and (ts - base) and the
  // stride fit in 32 bits, base being the start of the first bucket.
  int32_t fast32;
  int64_t base;
  uint32_t magic32;
  int32_t pre32;
  int32_t shift32;
  int32_t word;       // which 64-bit word of the packed key holds this field (0 or 1)
};

struct ValSpec {
  int32_t col;
  int32_t width;
  int32_t is_unsigned;
  int32_t is_float;  // values are f64 bit patterns (PARTIAL AVG sums)
};

// One leaf of a predicate evaluated inside the aggregation kernel, on a column that is
// loaded as one of the plan's values or keys.  A NULL input never makes a leaf TRUE, except IS NULL.
// kLeafLut: the input is a dictionary code below 64 and the literal a bit set of the codes that pass
enum FusedLeafKind { kLeafCmp = 0, kLeafIsNull = 1, kLeafIsNotNull = 2, kLeafLut = 3 };
struct FusedLeaf {
  // >= 0: index into AggParams::vals.  < 0: the leaf reads GROUP BY key -(val + 1), whose
  // column is loaded anyway (a second, nontemporal load of the same bytes would fetch them
  // from HBM twice): a kernel with 64-bit keys compares the raw value with `lit`, one with
  // 32-bit keys compares (low 32 bits - key_base32), which is the value minus the key's base
  // exactly, with `lit_k32` = lit - base clamped to [-1, 2^32] (kLeafLut: the bit set shifted down by base).
  int32_t val;
  int32_t kind;      // FusedLeafKind
  int32_t cmp_mask;  // CmpMask of a comparison with the literal
  uint32_t key_base32;
  long long lit;
  long long lit_k32;
};
constexpr int kMaxFusedLeaves = 4;

enum SlotOp {
  kSlotAddOne = 0,   // += 1 per row
  kSlotAddValid = 1, // += 1 per non-NULL value
  kSlotAddI64 = 2,   // += value (wrapping)
  kSlotAddF64 = 3,   // += (double)value, or += value when the input is f64
  kSlotMin = 4,
  kSlotMax = 5
};

struct SlotSpec {
  int32_t op;
  int32_t val;  // index into vals, -1 for kSlotAddOne
};

enum RowMode { kRowsAll = 0, kRowsBitmap = 1, kRowsSelvec = 2 };

constexpr uint64_t kEmptyKey = ~0ull;

struct AggParams {
  ScanDesc sd;
  const ColPtr* cols;  // device: cols[c * n_segs + seg]
  int32_t n_keys, n_vals, n_slots, row_mode;
  KeySpec keys[kMaxKeys];
  ValSpec vals[kMaxVals];
  SlotSpec slots[kMaxSlots];
  const uint64_t* sel_bits;
  const uint32_t* selvec;
  uint64_t n_sel;
  // global table: dense (index = packed key) or open addressing on the packed key
  int32_t dense;
  int32_t key_bits;     // bits of the packed key in use
  int32_t lds_slots;    // power of two; 0 = no LDS pre-aggregation
  int32_t stop_if_scattered;  // non-zero: a workgroup whose rows keep missing its LDS table sets *overflow = 3 and the launch winds down
  int32_t window_hint;  // non-zero: time buckets sit in the key's bits from (window_hint - 1) up, so the keys may be narrow at any one time
  uint64_t capacity;    // dense: number of entries; hash: power of two
  uint64_t* g_keys;     // hash only: capacity + 1 entries (last = the all-ones key)
  uint64_t* g_keys_hi;  // second key word of every entry when key_words == 2
  int32_t key_words;    // 1, or 2 for packed keys of 65..127 bits (hash table only, no LDS stage)
  int32_t keys_only;    // two-word keys: only the set of keys is wanted (COUNT(DISTINCT)'s inner level), no accumulators
  // WHERE evaluated in the kernel (streaming kernel only): a row counts when all
  // (pred_or = 0) or any (pred_or = 1) of the leaves are TRUE; the rows that pass are
  // counted into *selected_out
  int32_t n_pred, pred_or;
  FusedLeaf pred[kMaxFusedLeaves];
  unsigned long long* selected_out;
  int32_t selected_sharded;  // selected_out is a sharded count (kCountShards partial sums), not one word
  int64_t* g_state;     // n_slots arrays of (capacity + 1) entries
  uint32_t* overflow;   // set to 1 when the hash table is full
};

// The streaming GROUP BY kernel specialised for p's shape (number of key and value
// columns, 32- or 64-bit packed key), or nullptr when there is none; *lds_bytes
// receives the dynamic LDS size it needs.
using GroupByKernel = void (*)(AggParams);
// The streaming kernel for this plan shape (nullptr: none), with the workgroup size, the
// LDS table size (entries) and the dynamic LDS bytes to launch it with.
GroupByKernel pick_stream_kernel(const AggParams& p, size_t* lds_bytes, uint32_t* block, int* lds_slots);

// Ungrouped aggregates: out[n_slots] must hold the slot identities on entry.
void launch_agg_global(prism_gpu_ctx* ctx, const AggParams& p, int64_t* out);
// GROUP BY: accumulates into p.g_keys / p.g_state.
// Whether launch_group_by would run the plan on the streaming kernel (the only one that
// evaluates AggParams::pred).  Looks at n_keys, keys, key_bits, key_words, n_vals, lds_slots, row_mode.
bool group_by_streams(const AggParams& p);
void launch_group_by(prism_gpu_ctx* ctx, const AggParams& p, uint64_t algorithmic_bytes, const char* stat_name = "group_by");
// ---- partitioned GROUP BY (kernels_part.hip) ----------------------------------------------
// Rows are partitioned by a hash of their packed key into n_buckets buckets in HBM, then
// every bucket is aggregated in LDS and its groups appended to a compact list; the
// AggParams' HBM hash table only takes the groups of buckets too large for one workgroup
// and rows that found the LDS table full.
struct PartItem {
  uint32_t bucket, exclusive;  // exclusive: the item is a whole bucket
  uint64_t lo, hi;             // records [lo, hi)
};
struct PartParams {
  int32_t p_bits;          // log2(n_buckets)
  uint32_t n_buckets;
  uint32_t n_wg;           // workgroups of the histogram and scatter passes (the same rows in both)
  uint32_t tiles_per_wg;   // scan tiles per workgroup, a multiple of 4
  uint32_t* hist;          // [n_wg][n_buckets]
  uint32_t* totals;        // [n_buckets]
  uint64_t* bucket_base;   // [n_buckets + 1]
  uint64_t* rec[4];        // record words, one array per word
  uint64_t rec_capacity;   // records each array holds
  PartItem* items;
  uint32_t* n_items;
  uint32_t max_items;
  uint32_t chunk_rows;     // a bucket with more rows is aggregated in slices
  uint32_t lds_entries;    // entries of the aggregation pass' LDS table
  uint32_t lds_limit;      // entries it may fill
  // the list: entry i of [0, *list_cursor) is a group; arrays are laid out like a hash
  // table's with list_capacity entries (slot s of entry i at s * (list_capacity + 1) + i)
  uint64_t* list_keys;
  uint64_t* list_keys_hi;
  int64_t* list_state;
  uint64_t list_capacity;
  unsigned long long* list_cursor;
  uint32_t* flags;         // [0]: the list overflowed, [1]: rows / groups sent to the HBM table
  uint32_t* queue;         // next item of the aggregation pass (starts at 0)
  int32_t by_hi;           // two-word keys: the bucket follows from the high word alone
  int32_t hi_bits;         // by_hi: > 0 = the bucket is the top p_bits of bij_mix(high word, hi_bits), 0 = of a hash
  int32_t half_blocks;     // the scatter pass runs workgroups of 512 threads, two per CU (n_wg is sized for that)
};
// 8-byte words of a row's record for this plan shape (key words + value columns + one word
// of validity bits when a value column has NULLs); 0: the shape has no partitioned kernels.
int part_record_words(const AggParams& p, bool vals_nullable);
uint32_t part_max_buckets(int words);
uint32_t part_lds_entries(const AggParams& p);
// Whether the scatter pass can run two workgroups of 512 threads per CU for this record width and bucket count.
bool part_scatter_halves(int words, uint32_t n_buckets);
uint32_t part_scan_workgroups(prism_gpu_ctx* ctx, const AggParams& p, uint32_t* tiles_per_wg, bool two_per_cu = false);
// (hist_bytes, scatter_bytes, agg_bytes: algorithmic bytes of the three passes, for the statistics)
// scatter_only: stop once the records lie sorted by bucket (q.rec, q.bucket_base); nothing is aggregated.
void launch_partitioned_group_by(prism_gpu_ctx* ctx, const AggParams& p, const PartParams& q, bool vals_nullable,
                                 uint64_t hist_bytes, uint64_t scatter_bytes, uint64_t agg_bytes, bool scatter_only = false);

// ---- GROUP BY <one-word key> -> COUNT(*) without a histogram pass (kernels_wc.hip) ----------
// Hot keys (found in a sample) are counted in LDS by the scatter pass itself; the other rows go
// through per-bucket staging lines in LDS into fixed segments of the record array, one per
// (workgroup, bucket); a workgroup per bucket then counts the keys in LDS and appends the
// groups to the list.  flags[1] / flags[2] set: a segment or an LDS table overflowed, the
// result is void and the caller takes the path with the histogram.
struct WcParams {
  uint32_t n_wg, tiles_per_wg;  // workgroups of the scatter pass and the scan tiles each takes
  uint32_t cap_shift;           // a segment holds 2^cap_shift records
  uint64_t* rec;                // [buckets][n_wg][2^cap_shift]
  uint32_t* seg_count;          // [buckets][n_wg]
  uint64_t* s_keys;             // [wc_sample_counters()] the sample's table: keys (all ones on entry)
  uint32_t* s_cnt;              // [wc_sample_counters()] and their counts (zero on entry)
  uint32_t* cand;               // [wc_max_candidates()] entries of the sample's table whose count reached the threshold
  uint32_t* n_cand;             // zero on entry
  uint64_t* hot_keys;           // [wc_hot_slots()] the hot table (all ones: free entry)
  unsigned long long* hot_count;  // [wc_hot_slots()] rows per hot key, zero on entry
  unsigned long long* hot_rows;   // their sum, zero on entry
  uint32_t sample_tiles;        // scan tiles the sample takes
  uint32_t hot_threshold;       // sample count from which a key is a candidate
  uint32_t lds_entries;         // entries of the aggregation pass' LDS table (even)
  uint64_t* list_keys;          // the list, laid out as PartParams' (one slot: the count)
  int64_t* list_state;
  uint64_t list_capacity;
  unsigned long long* list_cursor;  // zero on entry
  uint32_t* flags;              // [0] the list overflowed, [1] a segment overflowed, [2] an LDS table overflowed; zero on entry
  // ORDER BY COUNT(*) DESC LIMIT leading (0: the list takes every group).  The hot keys' totals
  // are exact once the scatter pass is done; when at least `leading` of them are in use, the
  // smallest of the `leading` largest is a count that the last group of the answer reaches, and
  // only groups that reach it join the list: *cut is that count (0: every group was listed),
  // *groups_total the groups outside the hot table either way.  Both zero on entry.
  uint32_t leading;
  uint32_t* cut;
  unsigned long long* groups_total;
};
bool wc_plan_fits(const AggParams& p);
uint32_t wc_lds_entries(uint32_t n_wg);
uint32_t wc_buckets();
uint32_t wc_hot_slots();
uint32_t wc_max_candidates();
uint64_t wc_sample_counters();
void launch_wc_group_by(prism_gpu_ctx* ctx, const AggParams& p, const WcParams& w, uint64_t key_bytes);
// ones[0..n_ones) = all ones, zeros[0..n_zeros) = 0 (8-byte words), *flag = 0: what launch_wc_group_by expects on entry
void launch_wc_clear(prism_gpu_ctx* ctx, unsigned long long* ones, uint64_t n_ones, unsigned long long* zeros, uint64_t n_zeros, uint32_t* flag);

// COUNT(DISTINCT x) GROUP BY k over two-word (x, k) pairs whose set is far too large for one
// level of buckets: the records, already sorted into coarse buckets by a hash of k, are
// sorted once more, inside every coarse bucket, by a hash of the whole pair; each fine bucket
// is then deduplicated in LDS and its pairs are counted per group straight into the outer
// table (slot 0: pairs, slot 1: pairs whose x is not NULL) -- the set itself is never stored.
struct P2Piece {
  uint32_t t0, t1;     // scan tiles [t0, t1)
  uint32_t coarse;     // the coarse bucket its rows belong to
  uint32_t src;        // which of the bucket's pieces it is: the run it fills
  uint32_t seg_first;  // its part of a fine bucket's region: first record and records
  uint32_t seg_cap;
};
// What the scatter by staging lines needs on its rare paths only (spilled lines, rows of other
// coarse buckets, the end of a piece): kept in device memory and read where it is used, so that
// the main loop does not hold it in scalar registers (k_part2_pieces fills it in).
struct P2Rare {
  uint64_t* spill_rec;
  uint32_t* spill_fine;
  unsigned long long* spill_cursor;
  uint64_t spill_capacity;
  uint32_t* flags;
  unsigned long long* strays;
  unsigned long long* dropped;
  uint32_t* seg_base_out;
  uint32_t* seg_cnt_out;
  uint32_t piece_chunks, n_regions;
};
struct Part2Params {
  const uint64_t* in_lo;        // records sorted by coarse bucket
  const uint64_t* in_hi;
  uint64_t* out_lo;             // the same records sorted by fine bucket
  uint64_t* out_hi;
  const uint64_t* coarse_base;  // [2^p1_bits + 1]
  uint32_t p1_bits, p2_bits;    // fine bucket = coarse << p2_bits | sub
  uint64_t n_rows;
  uint32_t slice_rows;          // rows per workgroup of the histogram and scatter passes
  uint32_t* fine_count;         // [n_fine], zero on entry
  uint32_t* fine_base;          // [n_fine + 1]
  uint32_t* fine_cursor;        // [n_fine]
  uint32_t* fine_limit;         // [n_fine] or null.  Set: no histogram pass; fine bucket f owns the fixed region
                                // [fine_base[f], fine_limit[f]) (laid out by the library from the coarse buckets' sizes) and holds
                                // the records up to fine_cursor[f]; a bucket that outgrows its region sets flags[1]
  uint64_t out_capacity;        // records out_lo / out_hi hold
  uint32_t lds_entries;         // entries of the pair table in LDS (a fine bucket may hold half as many rows)
  int32_t x_null_bit;           // bit of the low word that marks a NULL x, -1: none
  int32_t dense;                // the outer table
  uint64_t capacity;
  uint64_t* g_keys;
  int64_t* g_state;
  uint32_t* overflow;
  uint32_t* queue;              // next fine bucket of the last pass (zero on entry)
  uint32_t* flags;              // [0]: a fine bucket did not fit LDS (the result is void), [1]: a fixed region overflowed (the same)
  // compact: both bucket numbers are top bits of bijections (coarse: of the high word's
  // hi_bits bits, sub: of the low word's lo_bits bits), and a record sorted into its fine
  // bucket is ONE word, the bits the bucket does not imply: (high rest << (lo_bits - p2_bits))
  // | low rest, below 2^63.  out_hi is unused then.  Needs p1_bits <= hi_bits <= p1_bits + 9,
  // p2_bits <= lo_bits, lo_bits - p2_bits + hi_bits - p1_bits <= 63 and no NULL bit in x.
  int32_t compact, lo_bits, hi_bits;
  unsigned long long* pairs;    // [8] (one-word records): distinct pairs counted, as eight partial sums; may be null
  // A query sharded over several GPUs (one-word records only).  emit_out set: the last pass writes
  // every fine bucket's DISTINCT records to emit_out[fine_base[f] ..] and their number to
  // fine_count[f] instead of counting them per group.  launch_distinct_merge: bucket j of this
  // rank's n_own buckets is the union of n_src runs of out_lo, run s at seg_base[s * n_own + j]
  // with seg_cnt[s * n_own + j] records; it is fine bucket ((j >> p2_bits) * world + rank, j & (2^p2_bits - 1)).
  uint64_t* emit_out;
  uint32_t n_src, world, rank;
  const uint32_t* seg_base;
  const uint32_t* seg_cnt;
  // direct (one-word records only): there is no first level.  The pairs are packed from the
  // scan's columns (the AggParams passed along: tiles [0, n_tiles), slice_tiles per workgroup) and
  // the coarse bucket is the top p1_bits of the high word AS IT IS -- the group key's time
  // bucket, which rows that arrive in time order already are sorted by; in_lo / in_hi /
  // coarse_base are unused.  Rows whose coarse bucket lies outside the LDS counters of their
  // slice (rows out of time order) are counted through global atomics and in *strays.
  int32_t direct;
  uint64_t n_tiles;
  uint32_t slice_tiles;
  unsigned long long* strays;
  // spill_rec set (direct only): no histogram pass either.  The first n_regions fine buckets get
  // regions of region_cap records each (fine_limit set by the library), and what does not fit a
  // region goes to a spill list -- record and bucket number, one reservation in *spill_cursor
  // per tile -- which is then sorted by bucket (spill_count / spill_base / spill_cur: n_regions
  // + 1 words each, zero on entry) behind the regions in out_lo; the last pass takes a bucket
  // as the two runs (seg_base / seg_cnt: 2 x n_regions words).  A spill list that runs over
  // sets flags[1].
  uint64_t* spill_rec;
  uint32_t* spill_fine;
  unsigned long long* spill_cursor;
  uint64_t spill_capacity;
  uint32_t n_regions, region_cap;
  uint32_t *spill_count, *spill_base, *spill_cur;
  uint32_t *seg_base_out, *seg_cnt_out;
  // lines (with spill_rec): the scatter pass without tiles, sorting and barriers.  The scan tiles
  // where the coarse bucket of the first row changes are found by bisection (rows in time order),
  // every coarse bucket's stretch of tiles is cut into at most piece_max pieces of about
  // piece_tiles tiles, and a workgroup takes one piece: its rows belong to ONE coarse bucket, so a
  // staging line of 8 records in LDS per sub-bucket is all it needs, and piece j of a coarse bucket
  // owns the j-th part of each of its fine buckets' regions -- a full line is copied out as 64
  // contiguous, aligned bytes to a place that follows from a counter in LDS.  Lines beyond the part
  // and rows of another coarse bucket go to the spill list.  A fine bucket then is piece_max runs of
  // its region and one of the sorted spill list (seg_base_out / seg_cnt_out: piece_max + 1 runs).
  uint32_t lines, piece_tiles, piece_max, max_pieces;
  uint32_t piece_chunks;  // entries of a workgroup's table of spill chunks (part2_piece_chunks)
  uint32_t* bounds;       // [2^p1_bits + 1] first tile of every coarse bucket's stretch
  P2Piece* pieces;        // [max_pieces]
  uint32_t* n_pieces;
  P2Rare* rare;
  uint32_t* pieces_too_long;    // set when a piece would pass 2 * piece_tiles tiles (no piece is made then: the caller scatters another way)
  unsigned long long* dropped;  // rows dropped because they repeated a recently staged record (statistics)
};
uint32_t part2_lds_entries(bool compact);
// Key columns of the scan whose loads the second level can request a tile ahead (0: some key is nullable, or too many).
int part2_keys_ahead(const AggParams& p);
// Spill chunks a piece of at most 2 * piece_tiles scan tiles can need (all of its rows spilled).
uint32_t part2_piece_chunks(uint32_t piece_tiles);
// Whether the scatter pass by staging lines fits LDS with these shapes.
bool part2_lines_fit(uint32_t p2_bits, uint32_t piece_tiles);
// Records the output arrays must hold when the fine buckets get regions of fixed size.
uint64_t part2_fixed_capacity(uint64_t n_rows, uint32_t n_fine);
void launch_distinct_second_level(prism_gpu_ctx* ctx, const Part2Params& q, const AggParams* scan = nullptr);
// The exchange of a sharded COUNT(DISTINCT): fine bucket (c, sub) belongs to rank c % world.
// counts64: the buckets' distinct-record counts in owner-major order (owner by owner, inside an
// owner's block by j = (c / world) * n_sub + sub), one 64-bit word each; launch_scan_u64 turns them
// into places in the send buffer (offsets[n] = the total); launch_shard_gather copies the records
// there; launch_shard_runs turns what an owner received (counts of n_own buckets from each of
// `world` ranks, rank s's records starting at recv_start[s]) into the runs launch_distinct_merge reads.
void launch_shard_counts(prism_gpu_ctx* ctx, const uint32_t* fine_count, unsigned long long* counts64, uint32_t n_fine, uint32_t p1_bits,
                         uint32_t p2_bits, uint32_t world);
void launch_scan_u64(prism_gpu_ctx* ctx, const unsigned long long* counts, unsigned long long* offsets, uint32_t n);
void launch_shard_gather(prism_gpu_ctx* ctx, const unsigned long long* records, const uint32_t* fine_base, const uint32_t* fine_count,
                         const unsigned long long* send_offsets, unsigned long long* send, uint32_t n_fine, uint32_t p1_bits,
                         uint32_t p2_bits, uint32_t world);
void launch_shard_runs(prism_gpu_ctx* ctx, const unsigned long long* recv_counts64, const unsigned long long* recv_start, uint32_t* seg_base,
                       uint32_t* seg_cnt, uint32_t n_own, uint32_t world);
void launch_distinct_merge(prism_gpu_ctx* ctx, const Part2Params& q, uint32_t n_own);

// Second level of COUNT(DISTINCT x) GROUP BY k: every occupied entry of the
// (k, x) set adds 1 (when its x field is not NULL) to group k of table `q`.
struct DistinctFoldParams {
  const uint64_t* set_keys;      // packed (k, x) keys, kEmptyKey = unused entry
  const uint64_t* set_keys_hi;   // non-null: two-word keys, x in the low word and k in the high word
  uint64_t set_capacity;         // entries to visit
  uint64_t set_special;          // index of the entry that stands for the all-ones key (~0 = none)
  const int64_t* set_rowcount;   // slot 0 of the set table (tells whether the special entry is used)
  int32_t x_bits;            // low bits holding x (including its NULL bit)
  int32_t x_null_bit;        // -1 if x has no NULL bit
  int32_t dense;
  uint64_t capacity;
  uint64_t* g_keys;
  int64_t* g_state;          // slot 0 = existence count, slot 1 = distinct count
  uint32_t* overflow;
  int32_t k_bits;            // bits of the group key (dense outer table: capacity = 2^k_bits)
};
void launch_distinct_fold(prism_gpu_ctx* ctx, const DistinctFoldParams& p);

// ORDER BY <one output column> [DESC] LIMIT k, decided on the device so that only the
// leading groups travel to the host.  Every occupied entry gets a 64-bit rank, larger =
// earlier in the requested order (NULLS FIRST when descending, NULLS LAST when
// ascending, as in sort_and_limit); a radix select finds a threshold rank that at least
// k entries reach, and launch_emit then compacts only those.  Entries that tie with the
// threshold all pass, so the host still sorts the survivors and cuts at k.
enum TopKKind { kOrdI64 = 0, kOrdAvg = 1, kOrdKeyField = 2 };
// A count that many workgroups add to is kept as kCountShards partial sums, each in a
// 128-byte line of its own, and added up by whoever reads it.  Updates of ONE address from
// all over the chip are carried out one after the other, some tens of nanoseconds each: two
// thousand workgroups that finish together and each add to one counter keep a kernel alive
// for 30 to 80 microseconds after its last load (a 50-microsecond kernel twice as long).
constexpr int kCountShards = 64, kCountShardStride = 16;  // (stride in 8-byte words)
constexpr size_t kCountBytes = (size_t)kCountShards * kCountShardStride * 8;
inline unsigned long long count_total(const unsigned long long* shards) {
  unsigned long long t = 0;
  for (int i = 0; i < kCountShards; ++i) t += shards[(size_t)i * kCountShardStride];
  return t;
}

struct TopKOrder {
  int kind;
  int main_slot, count_slot;  // kOrdI64 / kOrdAvg: value slot; count slot (> 0 <=> not NULL, -1: never NULL)
  int is_unsigned;            // kOrdI64: compare as unsigned
  int descending;
  int word, bit_pos, bits, null_bit;  // kOrdKeyField: where the field sits in the packed key
  int value_bits;             // kOrdI64, signed, never NULL: every value lies in [0, 2^value_bits); 0 = unknown
};
constexpr int kTopKBins = 2048;
struct TopKState {
  uint64_t prefix, mask;  // ranks still in play: (rank & mask) == prefix
  uint64_t k_rem;         // how many of those are wanted
  uint32_t done;          // a pass found a bin with few entries: everything from `prefix` up passes
  uint32_t exact;         // all 64 rank bits are resolved: of the entries whose rank equals `prefix`
                          // only k_rem are wanted (ties are broken arbitrarily)
  unsigned long long ties_taken;  // ties let through by the emit pass in progress
  // OR and AND of every occupied entry's rank (gathered by the first pass): where they agree
  // the bit is the same in all ranks, and a pass whose bits are all of that kind is skipped
  unsigned long long or_bits, and_bits;
  uint32_t hist[kTopKBins];
};
// Leaves the threshold in d_state->prefix.  d_state is overwritten.
void launch_topk_threshold(prism_gpu_ctx* ctx, bool dense, uint64_t capacity, const uint64_t* g_keys,
                           const uint64_t* g_keys_hi, const int64_t* g_state, const TopKOrder& order, uint64_t k,
                           TopKState* d_state);

// Result columns assembled on the device: the occupied entries of a table are compacted
// straight into the values the result batch holds (keys unpacked from the packed key,
// date_bin buckets turned back into timestamps, AVG divided, NULL flags worked out), so
// the host copies columns instead of decoding rows.
struct EmitKey {
  int word, bit_pos, bits, null_bit;
  long long bias;
  int date_bin;
  long long stride, origin;
  // dictionary-coded string key: the number of dictionary values (0 = not a code).  A
  // non-NULL code at or above it sets out_count[1], so the host need not check every row.
  unsigned long long n_values;
};
enum EmitAggKind {
  kEmitInt = 0,  // the main slot; NULL when a count slot is given and it is 0
  kEmitAvg = 1,  // Float64: main slot (a double) / count slot; NULL when the count is 0
  kEmitF64 = 2,  // Float64: main slot as is; NULL (0.0) when the count is 0
  kEmitOne = 3   // constant 1
};
struct EmitAgg {
  int kind, main_slot, count_slot;
};
constexpr int kMaxEmitAggs = 32;
struct EmitColumns {
  int n_keys;
  EmitKey keys[kMaxKeys];
  int n_aggs;
  EmitAgg aggs[kMaxEmitAggs];
};
// out_values: [n_keys + n_aggs][stride] 8-byte values, out_valid: the same shape in bytes
// (1 = not NULL).  out_values == nullptr: count only.  order / d_threshold as launch_emit.
// out_count points at two words: [0] the number of groups emitted, [1] set to 1 on a
// dictionary code out of range.  The outputs may be page-locked host memory.
void launch_emit_columns(prism_gpu_ctx* ctx, bool dense, uint64_t capacity, const uint64_t* g_keys, const uint64_t* g_keys_hi,
                         const int64_t* g_state, const EmitColumns& cols, int64_t* out_values, uint8_t* out_valid,
                         uint64_t stride, unsigned long long* out_count, const TopKOrder* order, TopKState* d_threshold);

// Merges groups exported from another table (packed keys + slot-major accumulators with
// stride `n`) into a table: find-or-insert the key, then apply every accumulator with the
// slot's merge rule (all "add" rules add, min and max stay min and max).
struct MergeParams {
  int dense;
  uint64_t capacity;
  uint64_t* g_keys;
  int64_t* g_state;
  uint32_t* overflow;
  int n_slots;
  int ops[kMaxSlots];  // SlotOp of each slot
  const uint64_t* src_keys;   // null: row i has key i (a dense table merged entry by entry)
  const int64_t* src_state;
  uint64_t n;
  uint64_t src_stride;        // values between two slots of the source (0: n)
  // two-word keys (null otherwise): the table's and the source's high words
  uint64_t* g_keys_hi;
  const uint64_t* src_keys_hi;
};
void launch_merge_packed(prism_gpu_ctx* ctx, const MergeParams& p);

// Compact group rows (keys[n], state[n_slots][n]) dealt to owner ranks by a hash of the key:
// counts[world] receives the rows per owner; the scatter writes the rows grouped by owner,
// in the same column layout, with cursor[world] holding each owner's first place on entry
// (and one past its last on exit).
// What a row's owner is derived from: the key (kOwnerKey; both words of a two-word key), or the
// part of it that names the group of a COUNT(DISTINCT) pair, so that all pairs of a group meet
// on one rank: the bits above `shift` of a one-word pair (kOwnerHighBits), the high word of a
// two-word pair (kOwnerHighWord).
enum OwnerOf { kOwnerKey = 0, kOwnerHighBits = 1, kOwnerHighWord = 2 };
struct OwnerParams {
  const uint64_t* keys;
  const uint64_t* keys_hi;  // null: one-word keys
  const int64_t* state;
  uint64_t n;
  int n_slots;
  uint32_t world;
  int owner_of;  // OwnerOf
  int shift;
};
void launch_owner_count(prism_gpu_ctx* ctx, const OwnerParams& p, unsigned long long* counts);
void launch_owner_scatter(prism_gpu_ctx* ctx, const OwnerParams& p, unsigned long long* cursor, uint64_t* out_keys,
                          uint64_t* out_keys_hi, int64_t* out_state);

// Fills n entries with `value`.
void launch_fill_i64(prism_gpu_ctx* ctx, int64_t* dst, uint64_t n, int64_t value);
// keys / keys_hi (either may be null) = all ones, state[s][i] = identity[s] for i < entries, *flag (may be null) = 0
struct TableInit {
  unsigned long long* keys;
  unsigned long long* keys_hi;
  long long* state;
  uint64_t entries;
  int n_slots;
  long long identity[kMaxSlots];
  uint32_t* flag;
};
void launch_table_init(prism_gpu_ctx* ctx, const TableInit& t);
// Compacts the occupied entries of a table: out_keys[i], out_state[s * out_stride + i].
// g_keys_hi / out_keys_hi are non-null for two-word keys.  Entries whose position (counted
// from *out_count's value on entry) is out_limit or more are counted but not written.
void launch_emit(prism_gpu_ctx* ctx, bool dense, uint64_t capacity, const uint64_t* g_keys, const uint64_t* g_keys_hi,
                 const int64_t* g_state, int n_slots, uint64_t* out_keys, uint64_t* out_keys_hi, int64_t* out_state,
                 uint64_t out_stride, unsigned long long* out_count, const TopKOrder* order = nullptr,
                 TopKState* d_threshold = nullptr, uint64_t out_limit = ~0ull);

// ---- loader support ----------------------------------------------------------

struct ColStats {
  long long min;
  long long max;
  unsigned long long null_count;
};
// Folds min/max/null count of `rows` values into *stats (device; initialised by the caller).
// Plain strings -> the codes of a dictionary they are all expected to be in (a table's later chunks
// of a low-cardinality column): a row's bytes are hashed, looked up in an open-addressing index of
// the dictionary (slots hold value number + 1, 0 = free; at most half of them taken) and compared
// with the value's bytes.  A row whose string is not there gets code 0 and raises report[0]
// (page-locked host memory), and the caller codes the chunk the long way.  NULL rows get code 0.
struct DictLookup {
  const int32_t* offsets;
  const uint8_t* bytes;
  const uint64_t* validity;
  uint64_t rows;
  const uint32_t* dict_off;   // n_values + 1
  const uint8_t* dict_bytes;
  const uint32_t* slots;      // mask + 1
  uint32_t mask;
  uint32_t* codes;
  uint32_t* report;
};
inline uint64_t dict_lookup_hash_host(const uint8_t* p, uint32_t len) {
  uint64_t h = 0xcbf29ce484222325ull ^ len;
  for (uint32_t i = 0; i < len; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
  h ^= h >> 32;
  return h * 0x9E3779B97F4A7C15ull >> 20;
}
void launch_dict_lookup(prism_gpu_ctx* ctx, const DictLookup& q, hipStream_t stream);
// Page-locked host memory -> HBM by a kernel that reads the host's memory where it is: for the
// loader's small copies, which as copy-engine commands would queue behind the tens of megabytes
// of column values that the same engine moves.  src and dst must agree modulo 16.
void launch_stage_copy(prism_gpu_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t stream);
void launch_col_stats(prism_gpu_ctx* ctx, const void* values, const uint64_t* validity, int width, bool is_unsigned,
                      uint64_t rows, ColStats* stats, hipStream_t stream);

// A string column that arrives dictionary encoded is expanded on the device: `index` holds, per
// row, the position of the row's value in the batches' dictionaries laid end to end
// (0xffffffff: NULL, an empty string), `dict_off`/`dict_bytes` those dictionaries.  The kernels
// write the column's offsets (rows + 1) and bytes and, when `codes` is set, the table-wide code
// remap[index] of every row (`codes` may be the same buffer as `index`).
struct DictExpand {
  const uint32_t* index;
  uint64_t rows;
  const uint32_t* dict_off;
  const uint8_t* dict_bytes;
  const uint32_t* remap;
  uint32_t* codes;
  int32_t* offsets;
  uint8_t* bytes;
  uint32_t* block_sums;  // dict_expand_blocks(rows) words of scratch
  // launch_dict_lens, optional: two words in page-locked host memory that receive *flag_in (0
  // without one) and the bytes' total, so that a caller who must know them before going on
  // awaits the stream and no copy besides.
  const uint32_t* flag_in = nullptr;
  uint32_t* report = nullptr;
};
// codes[r] (an index into the dictionary of r's batch) becomes remap[seg_base[s] + codes[r]], s the
// segment of row r (seg_row: first rows, ascending, seg_row[0] == 0); NULL rows become 0.
void launch_dict_codes(prism_gpu_ctx* ctx, uint32_t* codes, const uint64_t* validity, uint64_t rows, const uint32_t* seg_row,
                       const uint32_t* seg_base, uint32_t n_segs, const uint32_t* remap, hipStream_t stream);
size_t dict_expand_blocks(uint64_t rows);
// The two halves of launch_dict_expand for a caller that does not know the bytes' total yet: the
// first writes offsets[rows] (the total) and the workgroups' sums, the second, given `bytes`, the rest.
void launch_dict_lens(prism_gpu_ctx* ctx, const DictExpand& q, hipStream_t stream);
void launch_dict_fill(prism_gpu_ctx* ctx, const DictExpand& q, hipStream_t stream);
// One run of a Parquet RLE / bit-packed hybrid, as the library's reader keeps dictionary indices.
struct PackedRun {
  uint32_t first;    // row (of the batch) the run starts at
  uint32_t count;    // rows it covers
  uint32_t kind;     // bit 31: bit-packed (else one repeated value); low bits: the bit width
  uint32_t value;    // the repeated value, or the byte offset of the packed values in `bytes`
};
// out[r] = add + the r-th value of the runs; *flag = 1 when a value is not below dict_len or a row
// lies beyond its run.  8 readable bytes must lie behind the last run's values.
void launch_rle_expand(prism_gpu_ctx* ctx, const PackedRun* runs, uint32_t n_runs, const uint8_t* bytes, uint64_t rows, uint32_t add,
                       uint32_t dict_len, uint32_t* out, uint32_t* flag, hipStream_t stream);
void launch_dict_expand(prism_gpu_ctx* ctx, const DictExpand& q, hipStream_t stream);

// ---- a dictionary built on the device (kernels_strdict.hip) ----
// One chunk's raw string column; rows are numbered over the chunks (base + row).
struct StrSeg {
  const int32_t* off;
  const uint8_t* bytes;
  const uint64_t* validity;
  uint32_t* codes;  // out: the rows' codes
  uint64_t base;
  uint64_t rows;
};
// hashes[row]: 63 bits of the row's bytes (all ones: NULL).
void launch_str_hash(prism_gpu_ctx* ctx, const StrSeg* segs, uint32_t n_segs, uint64_t n_rows, uint64_t seed, uint64_t* hashes);
// keys / rep (capacity entries, a power of two, all ones on entry): the distinct hashes and the smallest row of each.
void launch_str_insert(prism_gpu_ctx* ctx, const uint64_t* hashes, uint64_t n_rows, uint64_t* keys, uint32_t* rep, uint64_t capacity);
// Numbers the occupied entries: code_of_entry, rep_by_code, *n_values; block_sums: str_dict_blocks(capacity) + 1 words.
size_t str_dict_blocks(uint64_t entries);
void launch_str_number(prism_gpu_ctx* ctx, const uint64_t* keys, uint64_t capacity, const uint32_t* rep, uint32_t* block_sums,
                       uint32_t* code_of_entry, uint32_t* rep_by_code, uint32_t* n_values);
// Every row's code into its segment's `codes`; *flag becomes 1 when two different strings share a hash.
void launch_str_codes(prism_gpu_ctx* ctx, const StrSeg* segs, uint32_t n_segs, uint64_t n_rows, const uint64_t* hashes, const uint64_t* keys,
                      const uint32_t* rep, const uint32_t* code_of_entry, uint64_t capacity, uint32_t* flag);
// The strings of the codes' representatives: value_off (n_values + 1) and, when value_bytes is
// set, the bytes (at most bytes_capacity of them); block_sums: str_dict_blocks(n_values) + 1 words.
void launch_str_values(prism_gpu_ctx* ctx, const StrSeg* segs, uint32_t n_segs, const uint32_t* rep_by_code, uint32_t n_values,
                       uint32_t* block_sums, uint32_t* value_off, uint8_t* value_bytes, uint64_t bytes_capacity);

struct SynthColDev {
  void* dst;
  int32_t width;  // 2, 4, 8
  int32_t kind;
  int64_t p[4];
  const uint32_t* cdf;  // device
  uint32_t cdf_len;
  uint64_t column_key;
};
// PRISM_SYNTH_URL: fills the bytes of n_rows strings whose offsets are given.
void launch_synth_url(prism_gpu_ctx* ctx, const int32_t* offsets, uint8_t* bytes, uint64_t n_rows, uint64_t first_row,
                      int64_t p0, int64_t p1, uint64_t column_key);
void launch_synth(prism_gpu_ctx* ctx, const SynthColDev& col, uint64_t first_row, uint64_t n_rows,
                  uint64_t total_rows);

}  // namespace prism
