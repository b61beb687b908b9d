// Aggregation kernels.
//
// GROUP BY keys are packed into one 64-bit word per row (each key column is biased
// by its minimum and given just the bits its range needs, plus a NULL bit; a
// date_bin key contributes its bucket number, computed with a multiply-high
// division).  The packed key either indexes a dense table directly (small key
// spaces) or is inserted into an open-addressing table in HBM with a 64-bit CAS.
// In front of the HBM table every workgroup keeps an LDS-resident partial table:
// rows update it with LDS atomics, and it is folded into the HBM table when it
// fills up and at the end, so hot groups cost HBM one atomic per workgroup rather
// than one per row.  Rows that find the LDS table full go straight to HBM, and a
// workgroup whose keys do not repeat stops using LDS altogether.
//
// Aggregate functions are lowered by the planner to "slots": 8-byte accumulators
// with one of six update rules (SlotOp).  Slot 0 always counts rows, which is also
// how a dense table knows which entries exist.
#include <algorithm>

#include "agg_device.hpp"

namespace prism {

namespace {

__device__ __forceinline__ uint64_t pack_key(const AggParams& p, uint32_t seg, uint64_t row) {
  uint64_t key = 0;
  for (int i = 0; i < p.n_keys; ++i) {
    const KeySpec& k = p.keys[i];
    const ColPtr& c = p.cols[(uint64_t)k.col * p.sd.n_segs + seg];
    if (k.null_bit >= 0 && !valid_at(c.validity, row)) { key |= 1ull << k.null_bit; continue; }
    long long v = load_value(c.values, k.width, k.is_unsigned, row);
    if (k.date_bin) v = bucket_number(v, k);
    key |= ((uint64_t)v - (uint64_t)k.bias) << k.bit_pos;
  }
  return key;
}

__device__ __forceinline__ void load_vals(const AggParams& p, uint32_t seg, uint64_t row, RowVals& rv) {
  rv.valid = 0;
#pragma unroll
  for (int j = 0; j < kMaxVals; ++j) {
    if (j < p.n_vals) {
      const ValSpec& vs = p.vals[j];
      const ColPtr& c = p.cols[(uint64_t)vs.col * p.sd.n_segs + seg];
      rv.v[j] = load_value(c.values, vs.width, vs.is_unsigned, row);
      if (valid_at(c.validity, row)) rv.valid |= 1u << j;
    }
  }
}

struct LdsTable {
  unsigned long long* keys;  // [S]
  long long* state;          // [n_slots][S]
  int S;
};

__device__ __forceinline__ int lds_entry(const LdsTable& t, uint64_t key, int* n_entries) {
  int mask = t.S - 1;
  // one 32-bit multiply: the LDS table only has a few thousand entries to spread over
  int h = (int)((((uint32_t)key ^ (uint32_t)(key >> 32)) * 0x9E3779B1u) >> 12) & mask;
  for (int probe = 0; probe < 16; ++probe) {
    unsigned long long k = t.keys[h];
    if (k == key) return h;
    if (k == kEmptyKey) {
      unsigned long long old = atomicCAS(&t.keys[h], kEmptyKey, (unsigned long long)key);
      if (old == kEmptyKey) { atomicAdd(n_entries, 1); return h; }
      if (old == key) return h;
    }
    h = (h + 1) & mask;
  }
  return -1;
}

template <int N>
__device__ __forceinline__ void apply_row(const AggParams& p, const LdsTable& t, bool use_lds, int* n_entries,
                                          uint64_t key, const RowValsT<N>& rv) {
  int e = (use_lds && key != kEmptyKey) ? lds_entry(t, key, n_entries) : -1;
  if (e < 0) { global_update(p, key, rv); return; }
  for (int s = 0; s < p.n_slots; ++s) {
    long long x;
    if (slot_input(p, s, rv, x)) atomic_apply(&t.state[s * t.S + e], p.slots[s].op, x);
  }
}

__device__ __forceinline__ void process_row(const AggParams& p, const LdsTable& t, bool use_lds, int* n_entries,
                                            uint32_t seg, uint64_t row) {
  uint64_t key = pack_key(p, seg, row);
  RowVals rv;
  load_vals(p, seg, row, rv);
  apply_row(p, t, use_lds, n_entries, key, rv);
}

__device__ __forceinline__ void lds_clear(const AggParams& p, const LdsTable& t) {
  for (int i = threadIdx.x; i < t.S; i += blockDim.x) {
    t.keys[i] = kEmptyKey;
    for (int s = 0; s < p.n_slots; ++s) t.state[s * t.S + i] = slot_identity(p.slots[s].op);
  }
}

// Folds the LDS table into the HBM table and empties it.  Block-wide; the caller
// brackets it with barriers.
__device__ __forceinline__ void lds_flush(const AggParams& p, const LdsTable& t) {
  for (int i = threadIdx.x; i < t.S; i += blockDim.x) {
    unsigned long long key = t.keys[i];
    if (key == kEmptyKey) continue;
    uint64_t e = global_entry(p, key);
    for (int s = 0; s < p.n_slots; ++s) {
      int op = p.slots[s].op;
      long long v = t.state[s * t.S + i];
      if (e != ~0ull && v != slot_identity(op)) atomic_apply(&p.g_state[(uint64_t)s * (p.capacity + 1) + e], op, v);
      t.state[s * t.S + i] = slot_identity(op);
    }
    t.keys[i] = kEmptyKey;
  }
}

__global__ __launch_bounds__(kBlock) void k_group_by(AggParams p) {
  extern __shared__ unsigned long long lds_raw[];
  __shared__ int n_entries;
  __shared__ int bypass;
  LdsTable t;
  t.S = p.lds_slots;
  t.keys = lds_raw;
  t.state = (long long*)(lds_raw + t.S);
  bool have_lds = t.S > 0;
  if (threadIdx.x == 0) { n_entries = 0; bypass = have_lds ? 0 : 1; }
  if (have_lds) lds_clear(p, t);
  __syncthreads();

  if (p.row_mode == kRowsSelvec) {
    uint64_t per = ceil_div(p.n_sel, (uint64_t)gridDim.x);
    per = ceil_div(per, (uint64_t)kBlock) * kBlock;
    uint64_t i0 = per * blockIdx.x, i1 = i0 + per < p.n_sel ? i0 + per : p.n_sel;
    for (uint64_t base = i0; base < i1; base += kBlock) {
      uint64_t i = base + threadIdx.x;
      bool use_lds = bypass == 0;
      if (i < i1) {
        uint64_t pid = p.selvec[i];
        uint64_t tile = pid / kTileRows;
        uint32_t seg = find_seg(p.sd, tile);
        process_row(p, t, use_lds, &n_entries, seg, pid - p.sd.segs[seg].tile_base * kTileRows);
      }
      if (have_lds && ((base - i0) / kBlock) % 16 == 15) {  // block uniform
        __syncthreads();
        int n = n_entries;
        __syncthreads();
        if (n > t.S / 2) {
          lds_flush(p, t);
          if (threadIdx.x == 0) { if ((uint64_t)n * 4 > 16ull * kBlock) bypass = 1; n_entries = 0; }
          __syncthreads();
        }
      }
    }
  } else {
    uint64_t per = ceil_div(p.sd.n_tiles, (uint64_t)gridDim.x);
    uint64_t t0 = per * blockIdx.x, t1 = t0 + per < p.sd.n_tiles ? t0 + per : p.sd.n_tiles;
    uint64_t rows_since_flush = 0;
    for (uint64_t tile = t0; tile < t1; ++tile) {
      uint32_t seg = find_seg(p.sd, tile);
      uint64_t seg_rows = p.sd.segs[seg].rows;
      uint64_t tile_row0 = (tile - p.sd.segs[seg].tile_base) * kTileRows;
      bool use_lds = bypass == 0;
      for (uint32_t it = 0; it < kTileRows / (kBlock * 2); ++it) {
        uint64_t in_tile = (uint64_t)it * (kBlock * 2) + (uint64_t)threadIdx.x * 2;
        uint64_t r = tile_row0 + in_tile;
        uint32_t sel = 3;
        if (p.row_mode == kRowsBitmap) {
          uint64_t pid = tile * kTileRows + in_tile;
          sel = (uint32_t)(p.sel_bits[pid >> 6] >> (pid & 63)) & 3u;
        }
        if ((sel & 1u) && r < seg_rows) process_row(p, t, use_lds, &n_entries, seg, r);
        if ((sel & 2u) && r + 1 < seg_rows) process_row(p, t, use_lds, &n_entries, seg, r + 1);
      }
      rows_since_flush += kTileRows;
      if (have_lds) {
        __syncthreads();
        int n = n_entries;
        __syncthreads();
        if (n > t.S / 2) {
          lds_flush(p, t);
          if (threadIdx.x == 0) {
            // keys that hardly repeat gain nothing from LDS: go straight to HBM from here on
            if ((uint64_t)n * 4 > rows_since_flush) bypass = 1;
            n_entries = 0;
          }
          rows_since_flush = 0;
          __syncthreads();
        }
      }
    }
  }
  if (have_lds) {
    __syncthreads();
    lds_flush(p, t);
  }
}

// ---- GROUP BY with packed keys wider than 64 bits ------------------------------------
//
// Two key words, straight into the HBM table (such keys come from high-cardinality
// columns, where an LDS stage would see no repeats anyway).
__device__ __forceinline__ void pack_key2(const AggParams& p, uint32_t seg, uint64_t row, uint64_t& lo, uint64_t& hi) {
  lo = 0;
  hi = 0;
  for (int i = 0; i < p.n_keys; ++i) {
    const KeySpec& k = p.keys[i];
    const ColPtr& c = p.cols[(uint64_t)k.col * p.sd.n_segs + seg];
    uint64_t field;
    if (k.null_bit >= 0 && !valid_at(c.validity, row)) {
      field = 1ull << k.null_bit;
    } else {
      long long v = load_value(c.values, k.width, k.is_unsigned, row);
      if (k.date_bin) v = bucket_number(v, k);
      field = ((uint64_t)v - (uint64_t)k.bias) << k.bit_pos;
    }
    if (k.word) hi |= field; else lo |= field;
  }
}

__global__ __launch_bounds__(kBlock) void k_group_by_wide(AggParams p) {
  auto row = [&](uint32_t seg, uint64_t r) {
    uint64_t lo, hi;
    pack_key2(p, seg, r, lo, hi);
    RowVals rv;
    load_vals(p, seg, r, rv);
    uint64_t e = global_entry2(p, lo, hi);
    if (e == ~0ull || p.keys_only) return;
    for (int s = 0; s < p.n_slots; ++s) {
      long long x;
      if (slot_input(p, s, rv, x)) atomic_apply(&p.g_state[(uint64_t)s * (p.capacity + 1) + e], p.slots[s].op, x);
    }
  };
  if (p.row_mode == kRowsSelvec) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < p.n_sel; i += (uint64_t)gridDim.x * kBlock) {
      uint64_t pid = p.selvec[i];
      uint32_t seg = find_seg(p.sd, pid / kTileRows);
      row(seg, pid - p.sd.segs[seg].tile_base * kTileRows);
    }
    return;
  }
  for (uint64_t tile = blockIdx.x; tile < p.sd.n_tiles; tile += gridDim.x) {
    if (table_gave_up(p)) return;  // the table is too small: this pass will be re-run
    uint32_t seg = find_seg(p.sd, tile);
    uint64_t seg_rows = p.sd.segs[seg].rows;
    uint64_t tile_row0 = (tile - p.sd.segs[seg].tile_base) * kTileRows;
    for (uint32_t it = 0; it < kTileRows / kBlock; ++it) {
      uint64_t in_tile = (uint64_t)it * kBlock + threadIdx.x;
      uint64_t r = tile_row0 + in_tile;
      bool sel = r < seg_rows;
      if (sel && p.row_mode == kRowsBitmap) {
        uint64_t pid = tile * kTileRows + in_tile;
        sel = (p.sel_bits[pid >> 6] >> (pid & 63)) & 1ull;
      }
      if (sel) row(seg, r);
    }
  }
}

// ---- ungrouped aggregates -------------------------------------------------------

__device__ __forceinline__ long long combine(int op, long long a, long long b) {
  if (op == kSlotMin) return a < b ? a : b;
  if (op == kSlotMax) return a > b ? a : b;
  if (op == kSlotAddF64) return __double_as_longlong(__longlong_as_double(a) + __longlong_as_double(b));
  return (long long)((unsigned long long)a + (unsigned long long)b);
}

__global__ __launch_bounds__(kBlock) void k_agg_global(AggParams p, long long* out) {
  long long acc[kMaxSlots];
#pragma unroll
  for (int s = 0; s < kMaxSlots; ++s) acc[s] = s < p.n_slots ? slot_identity(p.slots[s].op) : 0;

  auto row = [&](uint32_t seg, uint64_t r) {
    RowVals rv;
    load_vals(p, seg, r, rv);
#pragma unroll
    for (int s = 0; s < kMaxSlots; ++s) {
      if (s < p.n_slots) {
        long long x;
        if (slot_input(p, s, rv, x)) acc[s] = combine(p.slots[s].op, acc[s], x);
      }
    }
  };

  if (p.row_mode == kRowsSelvec) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < p.n_sel; i += (uint64_t)gridDim.x * kBlock) {
      uint64_t pid = p.selvec[i];
      uint32_t seg = find_seg(p.sd, pid / kTileRows);
      row(seg, pid - p.sd.segs[seg].tile_base * kTileRows);
    }
  } else {
    for (uint64_t tile = blockIdx.x; tile < p.sd.n_tiles; tile += gridDim.x) {
      uint32_t seg = find_seg(p.sd, tile);
      uint64_t seg_rows = p.sd.segs[seg].rows;
      uint64_t tile_row0 = (tile - p.sd.segs[seg].tile_base) * kTileRows;
      for (uint32_t it = 0; it < kTileRows / (kBlock * 2); ++it) {
        uint64_t in_tile = (uint64_t)it * (kBlock * 2) + (uint64_t)threadIdx.x * 2;
        uint64_t r = tile_row0 + in_tile;
        uint32_t sel = 3;
        if (p.row_mode == kRowsBitmap) {
          uint64_t pid = tile * kTileRows + in_tile;
          sel = (uint32_t)(p.sel_bits[pid >> 6] >> (pid & 63)) & 3u;
        }
        if ((sel & 1u) && r < seg_rows) row(seg, r);
        if ((sel & 2u) && r + 1 < seg_rows) row(seg, r + 1);
      }
    }
  }

  __shared__ long long partial[kBlock / 64][kMaxSlots];
#pragma unroll
  for (int s = 0; s < kMaxSlots; ++s) {
    if (s < p.n_slots) {
      int op = p.slots[s].op;
      long long v = acc[s];
      if (op == kSlotMin) v = wave_min_i64(v);
      else if (op == kSlotMax) v = wave_max_i64(v);
      else if (op == kSlotAddF64) v = __double_as_longlong(wave_sum_f64(__longlong_as_double(v)));
      else v = (long long)wave_sum_u64((unsigned long long)v);
      if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6][s] = v;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < p.n_slots) {
    int s = threadIdx.x, op = p.slots[s].op;
    long long v = partial[0][s];
    for (int w = 1; w < (int)(kBlock / 64); ++w) v = combine(op, v, partial[w][s]);
    if (v != slot_identity(op)) atomic_apply(&out[s], op, v);
  }
}

// ---- COUNT(DISTINCT) second level ----------------------------------------------------

__global__ __launch_bounds__(kBlock) void k_distinct_fold(DistinctFoldParams q) {
  AggParams p;  // only the table fields are used by global_entry
  p.dense = q.dense;
  p.capacity = q.capacity;
  p.g_keys = q.g_keys;
  p.overflow = q.overflow;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < q.set_capacity; i += (uint64_t)gridDim.x * kBlock) {
    if (q.set_keys_hi) {  // two words: x in the low word, the group key in the high word
      uint64_t lo = q.set_keys[i];
      if (lo == kEmptyKey) continue;
      bool x_null = q.x_null_bit >= 0 && ((lo >> q.x_null_bit) & 1ull);
      uint64_t e = global_entry(p, q.set_keys_hi[i]);
      if (e == ~0ull) continue;
      atomicAdd((unsigned long long*)&q.g_state[e], 1ull);
      if (!x_null) atomicAdd((unsigned long long*)&q.g_state[(q.capacity + 1) + e], 1ull);
      continue;
    }
    uint64_t key;
    if (i == q.set_special) {
      if (q.set_rowcount[i] == 0) continue;
      key = kEmptyKey;
    } else {
      key = q.set_keys[i];
      if (key == kEmptyKey) continue;
    }
    uint64_t k = q.x_bits >= 64 ? 0ull : key >> q.x_bits;
    bool x_null = q.x_null_bit >= 0 && ((key >> q.x_null_bit) & 1ull);
    uint64_t e = global_entry(p, k);
    if (e == ~0ull) continue;
    atomicAdd((unsigned long long*)&q.g_state[e], 1ull);
    if (!x_null) atomicAdd((unsigned long long*)&q.g_state[(q.capacity + 1) + e], 1ull);
  }
}

// The same fold for a small dense outer table, without a global atomic per pair: a
// workgroup counts the pairs of kFoldRange consecutive groups in LDS (32-bit counters) and
// adds its non-zero counters to the table at the end; the launcher runs one pass per
// kFoldRange groups.  A pass of a two-word set whose x has no NULL reads the high words
// only: an unused entry's high word is all ones, which no range contains.
constexpr uint32_t kFoldRange = 16384;  // counters per workgroup: 64 KiB of LDS
constexpr uint32_t kFoldBlock = 1024;

__device__ __forceinline__ uint64_t fold_load(const uint64_t* p) {
  return __builtin_nontemporal_load((const unsigned long long*)p);
}

__global__ __launch_bounds__(kFoldBlock) void k_distinct_fold_lds(DistinctFoldParams q, uint64_t k_first, uint32_t range, int with_nulls) {
  extern __shared__ uint32_t fold_counts[];  // [range] pairs, then [range] pairs with a non-NULL x when with_nulls
  for (uint32_t j = threadIdx.x; j < range * (with_nulls ? 2u : 1u); j += kFoldBlock) fold_counts[j] = 0;
  __syncthreads();
  const bool two = q.set_keys_hi != nullptr;
  const uint64_t* words = two ? q.set_keys_hi : q.set_keys;
  const uint64_t per = ceil_div(q.set_capacity, (uint64_t)gridDim.x);
  const uint64_t lo = per * blockIdx.x, hi = lo + per < q.set_capacity ? lo + per : q.set_capacity;
  constexpr int kInFlight = 4;
  for (uint64_t base = lo; base < hi; base += (uint64_t)kInFlight * kFoldBlock) {
    uint64_t w[kInFlight];
#pragma unroll
    for (int u = 0; u < kInFlight; ++u) {
      const uint64_t i = base + (uint64_t)u * kFoldBlock + threadIdx.x;
      w[u] = i < hi ? fold_load(&words[i]) : kEmptyKey;
    }
#pragma unroll
    for (int u = 0; u < kInFlight; ++u) {
      const uint64_t i = base + (uint64_t)u * kFoldBlock + threadIdx.x;
      // all ones: an unused entry (both words start that way), or the slot past the end
      if (w[u] == kEmptyKey) continue;
      const uint64_t k = two ? w[u] : (q.x_bits >= 64 ? 0ull : w[u] >> q.x_bits);
      const uint64_t rel = k - k_first;
      if (rel >= range) continue;
      atomicAdd(&fold_counts[rel], 1u);
      if (with_nulls) {
        const uint64_t x_word = two ? q.set_keys[i] : w[u];
        if (((x_word >> q.x_null_bit) & 1ull) == 0) atomicAdd(&fold_counts[range + rel], 1u);
      }
    }
  }
  // the entry that stands for the all-ones key of a one-word set
  if (!two && q.set_special != ~0ull && blockIdx.x == 0 && threadIdx.x == 0 && q.set_rowcount[q.set_special] != 0) {
    const uint64_t k = q.x_bits >= 64 ? 0ull : kEmptyKey >> q.x_bits;
    const uint64_t rel = k - k_first;
    if (rel < range) {
      atomicAdd(&fold_counts[rel], 1u);
      if (with_nulls && ((kEmptyKey >> q.x_null_bit) & 1ull) == 0) atomicAdd(&fold_counts[range + rel], 1u);
    }
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < range; j += kFoldBlock) {
    const uint32_t pairs = fold_counts[j];
    if (pairs == 0) continue;
    const uint64_t e = k_first + j;
    if (e >= q.capacity) continue;
    atomicAdd((unsigned long long*)&q.g_state[e], (unsigned long long)pairs);
    const uint32_t counted = with_nulls ? fold_counts[range + j] : pairs;
    if (counted) atomicAdd((unsigned long long*)&q.g_state[(q.capacity + 1) + e], (unsigned long long)counted);
  }
}

// ---- table maintenance --------------------------------------------------------------

__global__ __launch_bounds__(kBlock) void k_merge_packed(MergeParams p) {
  const uint64_t stride = p.src_stride ? p.src_stride : p.n;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < p.n; i += (uint64_t)gridDim.x * kBlock) {
    const uint64_t key = p.src_keys ? p.src_keys[i] : i;
    // a dense table is indexed by the key itself: a key beyond it comes from a state that
    // was packed from other statistics, and must not be written anywhere
    if (p.dense && key >= p.capacity) { *p.overflow = 2u; continue; }
    uint64_t e;
    if (p.g_keys_hi) {
      AggParams t{};  // what the two-word look-up reads of its parameters
      t.capacity = p.capacity;
      t.g_keys = p.g_keys;
      t.g_keys_hi = p.g_keys_hi;
      t.overflow = p.overflow;
      e = global_entry2(t, key, p.src_keys_hi[i]);
    } else {
      e = global_entry_raw(p.dense, p.capacity, p.g_keys, p.overflow, key);
    }
    if (e == ~0ull) continue;
    for (int s = 0; s < p.n_slots; ++s) {
      const int op = p.ops[s];
      const long long v = p.src_state[(uint64_t)s * stride + i];
      if (v == slot_identity(op)) continue;
      // counters merge by addition, like the sums
      atomic_apply(&p.g_state[(uint64_t)s * (p.capacity + 1) + e], (op == kSlotAddOne || op == kSlotAddValid) ? kSlotAddI64 : op, v);
    }
  }
}

// A fresh group table in one launch: keys all ones, every slot its identity, the flag word zero.
__global__ __launch_bounds__(kBlock) void k_table_init(TableInit t) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && t.flag) *t.flag = 0u;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < t.entries; i += (uint64_t)gridDim.x * kBlock) {
    if (t.keys) t.keys[i] = ~0ull;
    if (t.keys_hi) t.keys_hi[i] = ~0ull;
    for (int s = 0; s < t.n_slots; ++s) t.state[(uint64_t)s * t.entries + i] = t.identity[s];
  }
}

__global__ __launch_bounds__(kBlock) void k_fill_i64(long long* dst, uint64_t n, long long value) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) dst[i] = value;
}

// Occupied entries: dense -> row count (slot 0) non-zero; hash -> key present, and
// the extra entry for the all-ones key when its row count is non-zero.
__device__ __forceinline__ bool occupied(bool dense, uint64_t capacity, const uint64_t* g_keys, const long long* state0,
                                         uint64_t i) {
  if (dense || i == capacity) return state0[i] != 0;
  return g_keys[i] != kEmptyKey;
}

// Rank of entry i under `o`: larger = earlier in the requested order.
__device__ __forceinline__ uint64_t topk_rank(const TopKOrder& o, bool dense, uint64_t capacity, const uint64_t* g_keys,
                                              const uint64_t* g_keys_hi, const long long* g_state, uint64_t i) {
  bool valid = true;
  uint64_t u;
  if (o.kind == kOrdKeyField) {
    uint64_t key;
    if (o.word == 1) key = i == capacity ? 0ull : g_keys_hi[i];
    else key = dense ? i : (i == capacity ? kEmptyKey : g_keys[i]);
    if (o.null_bit >= 0) valid = ((key >> o.null_bit) & 1ull) == 0;
    u = o.bits ? ((key >> o.bit_pos) & (~0ull >> (64 - o.bits))) : 0ull;
  } else {
    const long long main = g_state[(uint64_t)o.main_slot * (capacity + 1) + i];
    const long long cnt = o.count_slot >= 0 ? g_state[(uint64_t)o.count_slot * (capacity + 1) + i] : 1;
    valid = cnt > 0;
    if (o.kind == kOrdI64) {
      u = o.is_unsigned ? (uint64_t)main : ((uint64_t)main ^ (1ull << 63));
    } else {
      double v = valid ? __longlong_as_double(main) / (double)cnt : 0.0;
      if (v == 0.0) v = 0.0;  // -0.0 ranks with +0.0
      const uint64_t b = (uint64_t)__double_as_longlong(v);
      u = (b >> 63) ? ~b : (b | (1ull << 63));
    }
  }
  if (o.descending) return valid ? u : ~0ull;
  return valid ? ~u : 0ull;
}

// One radix-select pass: histogram of the `bits` rank bits at `shift` over the entries
// whose higher bits match the prefix found so far.
__global__ __launch_bounds__(kBlock) void k_topk_hist(TopKOrder o, int dense, uint64_t capacity, const uint64_t* g_keys,
                                                     const uint64_t* g_keys_hi, const long long* g_state,
                                                     TopKState* st, int shift, int bits, int first_pass) {
  __shared__ uint32_t h[kTopKBins];
  if (st->done) return;
  const bool first = first_pass != 0;
  // (counts and the like use a few low bits: the passes over bits no rank differs in do nothing)
  if (!first && (((st->or_bits ^ st->and_bits) >> shift) & ((1ull << bits) - 1ull)) == 0) return;
  for (int b = threadIdx.x; b < kTopKBins; b += kBlock) h[b] = 0;
  __syncthreads();
  const uint64_t prefix = st->prefix, mask = st->mask;
  const uint64_t n = dense ? capacity : capacity + 1;
  unsigned long long all_or = 0ull, all_and = ~0ull;
  // Eight entries a step, their loads issued before any is looked at: the launch is kept to two
  // workgroups per CU (see launch_topk_threshold), so a thread has many entries to walk.
  constexpr int kAhead = 8;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += kAhead * stride) {
    uint64_t r[kAhead];
    bool occ[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      const uint64_t i = i0 + (uint64_t)j * stride, ic = i < n ? i : i0;
      occ[j] = i < n && g_state[ic] != 0;  // slot 0 counts the rows of the group
      r[j] = topk_rank(o, dense != 0, capacity, g_keys, g_keys_hi, g_state, ic);
    }
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      if (occ[j]) {
        all_or |= r[j];
        all_and &= r[j];
      }
      // The lanes that share the bin of the wave's first entry are counted and added once:
      // counts and the like put nearly every entry into one bin, and the lanes of an LDS
      // atomic that meet on one address are served one after the other.
      bool todo = occ[j] && (r[j] & mask) == prefix;
      const uint32_t bin = (uint32_t)(r[j] >> shift) & ((1u << bits) - 1);
      const uint64_t m_todo = __ballot(todo);
      if (m_todo) {
        const int leader = __ffsll((long long)m_todo) - 1;
        const uint32_t bin0 = __shfl(bin, leader);
        const uint64_t m_same = __ballot(todo && bin == bin0);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&h[bin0], (uint32_t)__popcll(m_same));
        todo = todo && bin != bin0;
      }
      if (todo) atomicAdd(&h[bin], 1u);
    }
  }
  if (first) {
    // one update per workgroup (through LDS): every update of these two words meets all the others
    __shared__ unsigned long long wg_or, wg_and;
    if (threadIdx.x == 0) { wg_or = 0ull; wg_and = ~0ull; }
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      all_or |= __shfl_xor(all_or, off, 64);
      all_and &= __shfl_xor(all_and, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      atomicOr(&wg_or, all_or);
      atomicAnd(&wg_and, all_and);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (wg_or) atomicOr(&st->or_bits, wg_or);
      if (wg_and != ~0ull) atomicAnd(&st->and_bits, wg_and);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kTopKBins; b += kBlock)
    if (h[b]) atomicAdd(&st->hist[b], h[b]);
}

// Picks the bin that holds the k_rem-th largest rank, narrows the prefix to it and
// clears the histogram for the next pass.  One workgroup.
__global__ __launch_bounds__(kBlock) void k_topk_pick(TopKState* st, int shift, int bits, int first_pass) {
  constexpr int kPer = kTopKBins / kBlock;
  __shared__ uint64_t part[kBlock];
  __shared__ uint64_t before[kBlock];  // entries in the bins above this thread's bins
  __shared__ uint64_t total;
  const bool done = st->done != 0;
  const uint64_t k = st->k_rem;
  const int t = threadIdx.x;
  // the pass was skipped: its bits are the same in every rank
  if (!first_pass && !done && (((st->or_bits ^ st->and_bits) >> shift) & ((1ull << bits) - 1ull)) == 0) {
    if (t == 0) {
      const uint64_t field = ((1ull << bits) - 1ull) << shift;
      st->prefix |= st->and_bits & field;
      st->mask |= field;
      if (shift == 0) st->exact = 1u;
    }
    return;
  }
  // thread t owns bins [top - kPer + 1, top], walked from the top down
  const int top = kTopKBins - 1 - t * kPer;
  uint32_t c[kPer];
  uint64_t sum = 0;
  for (int j = 0; j < kPer; ++j) { c[j] = st->hist[top - j]; sum += c[j]; }
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    uint64_t run = 0;
    for (int i = 0; i < kBlock; ++i) { before[i] = run; run += part[i]; }
    total = run;
  }
  __syncthreads();
  if (!done) {
    if (total < k) {
      // fewer entries than wanted: everything passes
      if (t == 0) { st->prefix = 0; st->mask = 0; st->done = 1; }
    } else if (before[t] < k && k <= before[t] + sum) {
      uint64_t acc = before[t];
      for (int j = 0; j < kPer; ++j) {
        if (acc + c[j] >= k) {
          st->prefix |= (uint64_t)(top - j) << shift;
          st->mask |= (uint64_t)((1u << bits) - 1) << shift;
          st->k_rem = k - acc;
          // few entries share the bin: pass them all and let the host sort them out
          if (c[j] <= 1024) st->done = 1;
          break;
        }
        acc += c[j];
      }
    }
  }
  for (int j = 0; j < kPer; ++j) st->hist[top - j] = 0;
  __syncthreads();  // the thread that owns the chosen bin may just have set st->done
  if (t == 0 && shift == 0 && !done) st->exact = st->done ? 0u : 1u;
}

// Output positions for the entries a workgroup takes in one step of its loop: one update of
// the counter per workgroup (updates of one address are carried out one after the other at
// the memory side: a update per wave made the compaction of a large table wait for them).
// `m`: the wave's lanes that take an entry.  Returns where the wave's first entry goes.
// Every thread of the workgroup must call it (two barriers).
__device__ __forceinline__ unsigned long long emit_reserve(uint64_t m, unsigned long long* out_count) {
  __shared__ uint32_t wave_cnt[kBlock / 64];
  __shared__ unsigned long long block_first;
  const uint32_t wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (uint32_t w = 0; w < kBlock / 64; ++w) total += wave_cnt[w];
    block_first = total ? atomicAdd(out_count, (unsigned long long)total) : 0ull;
  }
  __syncthreads();
  unsigned long long first = block_first;
  for (uint32_t w = 0; w < wave; ++w) first += wave_cnt[w];
  __syncthreads();  // wave_cnt is rewritten by the next step
  return first;
}

__global__ __launch_bounds__(kBlock) void k_emit(int dense, uint64_t capacity, const uint64_t* g_keys,
                                                const uint64_t* g_keys_hi, const long long* g_state, int n_slots,
                                                uint64_t* out_keys, uint64_t* out_keys_hi, long long* out_state,
                                                uint64_t out_stride, unsigned long long* out_count, int filtered,
                                                TopKOrder order, TopKState* threshold, uint64_t out_limit) {
  const uint64_t n = dense ? capacity : capacity + 1;
  const uint64_t min_rank = filtered ? threshold->prefix : 0ull;
  const bool exact = filtered && threshold->exact != 0;
  const uint64_t ties_wanted = filtered ? threshold->k_rem : 0ull;
  const int lane = threadIdx.x & 63;
  for (uint64_t base = (uint64_t)blockIdx.x * kBlock; base < n; base += (uint64_t)gridDim.x * kBlock) {
    const uint64_t i = base + threadIdx.x;
    bool take = i < n && occupied(dense != 0, capacity, g_keys, g_state, i);
    if (take && filtered) {
      const uint64_t rank = topk_rank(order, dense != 0, capacity, g_keys, g_keys_hi, g_state, i);
      take = rank >= min_rank;
      // entries that tie with the threshold: only as many as are still wanted get through
      // (once the quota is full a plain read of the counter settles it, without an atomic)
      bool tie = exact && rank == min_rank;
      if (tie && __hip_atomic_load(&threshold->ties_taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ties_wanted) {
        tie = false;
        take = false;
      }
      const uint64_t tm = __ballot(tie);
      if (tm) {
        unsigned long long first_tie = 0;
        if (lane == __ffsll((long long)tm) - 1) first_tie = atomicAdd(&threshold->ties_taken, (unsigned long long)__popcll(tm));
        first_tie = __shfl(first_tie, __ffsll((long long)tm) - 1);
        if (tie) take = first_tie + __popcll(tm & ((1ull << lane) - 1)) < ties_wanted;
      }
    }
    const uint64_t m = __ballot(take);
    const unsigned long long first = emit_reserve(m, out_count);
    if (!take || out_keys == nullptr) continue;  // counting pass
    const unsigned long long pos = first + __popcll(m & ((1ull << lane) - 1));
    if (pos >= out_limit) continue;
    out_keys[pos] = dense ? i : (i == capacity ? kEmptyKey : g_keys[i]);
    if (out_keys_hi) out_keys_hi[pos] = i == capacity ? 0ull : g_keys_hi[i];
    for (int s = 0; s < n_slots; ++s) out_state[(uint64_t)s * out_stride + pos] = g_state[(uint64_t)s * (capacity + 1) + i];
  }
}

__global__ __launch_bounds__(kBlock) void k_emit_columns(int dense, uint64_t capacity, const uint64_t* g_keys,
                                                        const uint64_t* g_keys_hi, const long long* g_state,
                                                        EmitColumns cols, long long* out_values, uint8_t* out_valid,
                                                        uint64_t stride, unsigned long long* out_count, int filtered,
                                                        TopKOrder order, TopKState* threshold) {
  const uint64_t n = dense ? capacity : capacity + 1;
  const uint64_t min_rank = filtered ? threshold->prefix : 0ull;
  const bool exact = filtered && threshold->exact != 0;
  const uint64_t ties_wanted = filtered ? threshold->k_rem : 0ull;
  const int lane = threadIdx.x & 63;
  for (uint64_t base = (uint64_t)blockIdx.x * kBlock; base < n; base += (uint64_t)gridDim.x * kBlock) {
    const uint64_t i = base + threadIdx.x;
    bool take = i < n && occupied(dense != 0, capacity, g_keys, g_state, i);
    if (take && filtered) {
      const uint64_t rank = topk_rank(order, dense != 0, capacity, g_keys, g_keys_hi, g_state, i);
      take = rank >= min_rank;
      bool tie = exact && rank == min_rank;
      if (tie && __hip_atomic_load(&threshold->ties_taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ties_wanted) {
        tie = false;
        take = false;
      }
      const uint64_t tm = __ballot(tie);
      if (tm) {
        unsigned long long first_tie = 0;
        if (lane == __ffsll((long long)tm) - 1) first_tie = atomicAdd(&threshold->ties_taken, (unsigned long long)__popcll(tm));
        first_tie = __shfl(first_tie, __ffsll((long long)tm) - 1);
        if (tie) take = first_tie + __popcll(tm & ((1ull << lane) - 1)) < ties_wanted;
      }
    }
    const uint64_t m = __ballot(take);
    const unsigned long long first = emit_reserve(m, out_count);
    if (!take || out_values == nullptr) continue;
    const unsigned long long pos = first + __popcll(m & ((1ull << lane) - 1));
    if (pos >= stride) continue;  // cannot happen when stride >= occupied entries; never write outside
    const uint64_t key_lo = dense ? i : (i == capacity ? kEmptyKey : g_keys[i]);
    const uint64_t key_hi = g_keys_hi ? (i == capacity ? 0ull : g_keys_hi[i]) : 0ull;
    int c = 0;
    for (int k = 0; k < cols.n_keys; ++k, ++c) {
      const EmitKey& ek = cols.keys[k];
      const uint64_t packed = ek.word ? key_hi : key_lo;
      const bool valid = ek.null_bit < 0 || ((packed >> ek.null_bit) & 1ull) == 0;
      const uint64_t mask = ek.bits == 0 ? 0ull : (~0ull >> (64 - ek.bits));
      long long v = (long long)((uint64_t)ek.bias + ((packed >> ek.bit_pos) & mask));
      if (ek.date_bin) v = (long long)((uint64_t)ek.origin + (uint64_t)v * (uint64_t)ek.stride);
      if (ek.n_values && valid && (unsigned long long)v >= ek.n_values) out_count[1] = 1ull;
      out_values[(uint64_t)c * stride + pos] = valid ? v : 0;
      out_valid[(uint64_t)c * stride + pos] = valid ? 1 : 0;
    }
    for (int a = 0; a < cols.n_aggs; ++a, ++c) {
      const EmitAgg& ea = cols.aggs[a];
      const long long main = ea.main_slot >= 0 ? g_state[(uint64_t)ea.main_slot * (capacity + 1) + i] : 1;
      const long long cnt = ea.count_slot >= 0 ? g_state[(uint64_t)ea.count_slot * (capacity + 1) + i] : 1;
      long long v = main;
      bool valid = true;
      if (ea.kind == kEmitInt) {
        valid = cnt > 0;
      } else if (ea.kind == kEmitAvg) {
        valid = cnt > 0;
        v = __double_as_longlong(valid ? __longlong_as_double(main) / (double)cnt : 0.0);
      } else if (ea.kind == kEmitF64) {
        valid = cnt > 0;
        v = valid ? main : 0;
      } else {
        v = 1;
      }
      out_values[(uint64_t)c * stride + pos] = v;
      out_valid[(uint64_t)c * stride + pos] = valid ? 1 : 0;
    }
  }
}

uint32_t stream_grid(prism_gpu_ctx* ctx, uint64_t items) {
  uint64_t blocks = ceil_div(items, kBlock);
  uint64_t cap = (uint64_t)ctx->n_cus * 8;
  return (uint32_t)std::max<uint64_t>(1, std::min(blocks, cap));
}

}  // namespace

void launch_agg_global(prism_gpu_ctx* ctx, const AggParams& p, int64_t* out) {
  uint64_t work = p.row_mode == kRowsSelvec ? ceil_div(p.n_sel, kBlock) : p.sd.n_tiles;
  if (work == 0) return;
  uint32_t grid = (uint32_t)std::min<uint64_t>(work, (uint64_t)ctx->n_cus * 8);
  k_agg_global<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(p, (long long*)out);
  PRISM_HIP(hipGetLastError());
}

bool group_by_streams(const AggParams& p) {
  if (p.key_words == 2 || p.row_mode == kRowsSelvec || p.lds_slots <= 0) return false;
  size_t lds = 0;
  uint32_t block = 0;
  int slots = 0;
  return pick_stream_kernel(p, &lds, &block, &slots) != nullptr;
}

void launch_group_by(prism_gpu_ctx* ctx, const AggParams& params, uint64_t algorithmic_bytes, const char* stat_name) {
  AggParams p = params;
  if (p.n_pred > 0 && !group_by_streams(p)) fail(PRISM_GPU_EINTERNAL, "in-kernel predicate on a plan the streaming kernel does not run");
  uint64_t work = p.row_mode == kRowsSelvec ? ceil_div(p.n_sel, (uint64_t)kBlock * 16) : p.sd.n_tiles;
  if (work == 0) return;
  size_t lds = (size_t)p.lds_slots * 8 * (1 + p.n_slots);
  uint32_t block = kBlock;
  GroupByKernel kernel = k_group_by;
  if (p.key_words == 2) {
    kernel = k_group_by_wide;
    lds = 0;
  } else if (p.row_mode != kRowsSelvec && p.lds_slots > 0) {
    size_t fast_lds = 0;
    uint32_t fast_block = 0;
    int fast_slots = 0;
    if (GroupByKernel fast = pick_stream_kernel(p, &fast_lds, &fast_block, &fast_slots)) {
      kernel = fast;
      lds = fast_lds;
      block = fast_block;
      p.lds_slots = fast_slots;
      if (lds > 48 * 1024)
        PRISM_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
  }
  // One round of workgroups: exactly as many as are resident at once, each walking a
  // contiguous run of tiles, so that no CU idles while a second, partial round runs and
  // every workgroup's LDS table sees each group many times.
  int per_cu = 0;
  PRISM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)block, lds));
  if (per_cu < 1) per_cu = 1;
  uint32_t grid = (uint32_t)std::min<uint64_t>(work, (uint64_t)ctx->n_cus * (uint64_t)per_cu);
  KernelTimer timer(ctx, stat_name, algorithmic_bytes);
  kernel<<<dim3(grid), dim3(block), lds, ctx->stream>>>(p);
  PRISM_HIP(hipGetLastError());
}

void launch_distinct_fold(prism_gpu_ctx* ctx, const DistinctFoldParams& p) {
  // Small dense outer table: count in LDS, one pass per kFoldRange groups (each pass reads
  // the set again, which is far cheaper than a global atomic per pair).  A workgroup sees
  // fewer than 2^32 entries, so its 32-bit counters cannot wrap.
  const bool with_nulls = p.x_null_bit >= 0;
  const uint32_t range = with_nulls ? kFoldRange / 2 : kFoldRange;
  if (p.dense && p.k_bits <= 18 && p.set_capacity >= (1ull << 16) && p.set_capacity < (1ull << 40)) {
    const uint64_t groups = 1ull << p.k_bits;
    const uint64_t words_per_entry = (with_nulls && p.set_keys_hi) ? 2 : 1;
    const size_t lds = (size_t)kFoldRange * 4;
    PRISM_HIP(hipFuncSetAttribute((const void*)k_distinct_fold_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(p.set_capacity, (uint64_t)kFoldBlock * 4), (uint64_t)ctx->n_cus * 2);
    const uint64_t passes = ceil_div(groups, (uint64_t)range);
    KernelTimer timer(ctx, "distinct_fold", p.set_capacity * 8 * words_per_entry * passes);
    for (uint64_t k_first = 0; k_first < groups; k_first += range) {
      k_distinct_fold_lds<<<dim3(grid), dim3(kFoldBlock), lds, ctx->stream>>>(p, k_first, range, with_nulls ? 1 : 0);
      PRISM_HIP(hipGetLastError());
    }
    return;
  }
  uint32_t grid = stream_grid(ctx, p.set_capacity);
  KernelTimer timer(ctx, "distinct_fold", p.set_capacity * 8);
  k_distinct_fold<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(p);
  PRISM_HIP(hipGetLastError());
}

void launch_merge_packed(prism_gpu_ctx* ctx, const MergeParams& p) {
  if (p.n == 0) return;
  KernelTimer timer(ctx, "merge_packed", p.n * 8 * (1 + (uint64_t)p.n_slots));
  k_merge_packed<<<dim3(stream_grid(ctx, p.n)), dim3(kBlock), 0, ctx->stream>>>(p);
  PRISM_HIP(hipGetLastError());
}

void launch_table_init(prism_gpu_ctx* ctx, const TableInit& t) {
  if (t.n_slots < 0 || t.n_slots > kMaxSlots) fail(PRISM_GPU_EINTERNAL, "bad slot count for a group table");
  k_table_init<<<dim3(stream_grid(ctx, t.entries ? t.entries : 1)), dim3(kBlock), 0, ctx->stream>>>(t);
  PRISM_HIP(hipGetLastError());
}

void launch_fill_i64(prism_gpu_ctx* ctx, int64_t* dst, uint64_t n, int64_t value) {
  if (n == 0) return;
  k_fill_i64<<<dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream>>>((long long*)dst, n, value);
  PRISM_HIP(hipGetLastError());
}

void launch_emit(prism_gpu_ctx* ctx, bool dense, uint64_t capacity, const uint64_t* g_keys, const uint64_t* g_keys_hi,
                 const int64_t* g_state, int n_slots, uint64_t* out_keys, uint64_t* out_keys_hi, int64_t* out_state,
                 uint64_t out_stride, unsigned long long* out_count, const TopKOrder* order,
                 TopKState* d_threshold, uint64_t out_limit) {
  uint64_t n = dense ? capacity : capacity + 1;
  if (n == 0) return;
  TopKOrder o{};
  if (order) {
    o = *order;
    PRISM_HIP(hipMemsetAsync(&d_threshold->ties_taken, 0, sizeof(d_threshold->ties_taken), ctx->stream));
  }
  k_emit<<<dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream>>>(dense ? 1 : 0, capacity, g_keys, g_keys_hi,
                                                                    (const long long*)g_state, n_slots, out_keys,
                                                                    out_keys_hi, (long long*)out_state, out_stride,
                                                                    out_count, order ? 1 : 0, o, d_threshold, out_limit);
  PRISM_HIP(hipGetLastError());
}

void launch_emit_columns(prism_gpu_ctx* ctx, bool dense, uint64_t capacity, const uint64_t* g_keys, const uint64_t* g_keys_hi,
                         const int64_t* g_state, const EmitColumns& cols, int64_t* out_values, uint8_t* out_valid,
                         uint64_t stride, unsigned long long* out_count, const TopKOrder* order, TopKState* d_threshold) {
  uint64_t n = dense ? capacity : capacity + 1;
  if (n == 0) return;
  TopKOrder o{};
  if (order) {
    o = *order;
    PRISM_HIP(hipMemsetAsync(&d_threshold->ties_taken, 0, sizeof(d_threshold->ties_taken), ctx->stream));
  }
  k_emit_columns<<<dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream>>>(
      dense ? 1 : 0, capacity, g_keys, g_keys_hi, (const long long*)g_state, cols, (long long*)out_values, out_valid, stride,
      out_count, order ? 1 : 0, o, d_threshold);
  PRISM_HIP(hipGetLastError());
}

void launch_topk_threshold(prism_gpu_ctx* ctx, bool dense, uint64_t capacity, const uint64_t* g_keys,
                           const uint64_t* g_keys_hi, const int64_t* g_state, const TopKOrder& order, uint64_t k,
                           TopKState* d_state) {
  const uint64_t n = dense ? capacity : capacity + 1;
  PRISM_HIP(hipMemsetAsync(d_state, 0, sizeof(TopKState), ctx->stream));
  PRISM_HIP(hipMemcpyAsync(&d_state->k_rem, &k, 8, hipMemcpyHostToDevice, ctx->stream));
  PRISM_HIP(hipMemsetAsync(&d_state->and_bits, 0xff, 8, ctx->stream));
  // Ranks known to agree in their upper bits (counts: no group has more rows than the scan):
  // the select starts below them, with the common prefix in place.  A valid value v ranks as
  // v ^ 2^63 when descending and as the complement of that when ascending.
  int top = 64;
  if (order.kind == kOrdI64 && !order.is_unsigned && order.count_slot < 0 && order.value_bits > 0 && order.value_bits < 53) {
    top = order.value_bits;
    const uint64_t mask = ~0ull << top;
    const uint64_t preset[2] = {(order.descending ? (1ull << 63) : ~(1ull << 63)) & mask, mask};
    PRISM_HIP(hipMemcpyAsync(&d_state->prefix, preset, 16, hipMemcpyHostToDevice, ctx->stream));
  }
  KernelTimer timer(ctx, "topk_select", n * 16);
  // Few workgroups: each ends with updates of the bins its entries fell into, which for
  // counts and the like are the same few bins for everybody, and updates of one address from
  // all over the chip are carried out one after the other, some tens of nanoseconds each (with
  // eight workgroups per CU a pass over 8 M entries took 80 us, nearly all of it waiting for
  // those updates to drain).
  const uint32_t grid = (uint32_t)std::min<uint64_t>(stream_grid(ctx, n), (uint64_t)ctx->n_cus * 2);
  // the rank bits below `top`, most significant first, 11 at a time (all 64: five passes of 11 bits and one of 9)
  for (int hi = top; hi > 0;) {
    const int bits = hi >= 11 ? 11 : hi;
    const int shift = hi - bits;
    const int first = hi == top ? 1 : 0;
    k_topk_hist<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(order, dense ? 1 : 0, capacity, g_keys, g_keys_hi,
                                                                           (const long long*)g_state, d_state, shift, bits, first);
    k_topk_pick<<<dim3(1), dim3(kBlock), 0, ctx->stream>>>(d_state, shift, bits, first);
    hi = shift;
  }
  PRISM_HIP(hipGetLastError());
}

}  // namespace prism
