// Streaming GROUP BY: the kernel that runs when every row (or a bitmap-selected subset)
// of the scanned segments is aggregated.
//
// Per step a lane handles kU rows (row per lane, so a wave instruction reads one
// contiguous run of each column).  The step is software pipelined:
//   1. convert: wait for the batch in flight and reduce it, in registers, to one packed
//      key per row plus the widened aggregate inputs;
//   2. issue:   start the loads of the next step (every column, kU rows, one unaligned
//      8-byte load per value whatever the column width, so no branch separates them);
//   3. consume: find-or-insert the key in the workgroup's LDS table and apply the slot
//      updates with LDS atomics while the loads of (2) are in flight.
// The kernel is specialised on the number of key columns (NK), of aggregate input
// columns (NV) and on whether the packed key fits in 31 bits (K32: 32-bit key
// arithmetic, 32-bit LDS keys, and date_bin as one 32-bit multiply-high when the
// planner found that the timestamps allow it).
#include <algorithm>
#include <type_traits>

#include "agg_device.hpp"

namespace prism {

namespace {

constexpr int kU = 4;
constexpr uint32_t kStepsPerTile = kTileRows / (kBlock * kU);
constexpr int kLdsProbes = 16;

// Pointers read from descriptor arrays are generic; telling the compiler that they
// point to global memory turns the column reads into global_load instructions (a flat
// load also has to wait for LDS traffic).
template <typename T>
using GlobalPtr = const __attribute__((address_space(1))) T*;
typedef unsigned long long __attribute__((aligned(1))) unaligned_u64;

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

template <bool K32>
using KeyT = std::conditional_t<K32, uint32_t, uint64_t>;

// The workgroup's LDS table.  It is split into 2^rep_log2 replicas of S >> rep_log2
// entries each; a lane uses the replica picked by its lane number.  Rows of a hot
// group that meet in one wave instruction then update different addresses instead of
// serialising on one (LDS atomics to the same address are applied one lane at a
// time), at the price of holding each group up to 2^rep_log2 times.
template <bool K32>
struct Lds {
  long long* state;   // [n_slots][S]
  KeyT<K32>* keys;    // [S]
  int S;
  int rep_log2;
};

template <int NK, int NV>
struct RowBatch {
  static constexpr int NKA = NK > 0 ? NK : 1, NVA = NV > 0 ? NV : 1;
  unsigned long long kraw[NKA][kU];
  unsigned long long vraw[NVA][kU];
  uint32_t kok[NKA], vok[NVA];  // validity masks, bit u = row u
  uint32_t live;                // rows that exist and are selected
};

template <int NK, int NV, bool K32>
struct PackedRows {
  static constexpr int NVA = NV > 0 ? NV : 1;
  KeyT<K32> key[kU];
  long long val[NVA][kU];
  uint32_t vok[NVA];
  uint32_t live;
};

// Wave-uniform description of the segment the workgroup is walking.
template <int NK, int NV>
struct SegCursor {
  static constexpr int NKA = NK > 0 ? NK : 1, NVA = NV > 0 ? NV : 1;
  uint32_t seg = 0xffffffffu;
  uint64_t seg_rows = 0, tile_base = 0, next_tile_base = 0;  // next_tile_base = 0 forces the first lookup
  uint64_t kval[NKA], kvalid[NKA], vval[NVA], vvalid[NVA];   // addresses
};

// Moves the cursor to the segment that holds `tile`.  A workgroup walks its tiles in
// ascending order, so after the first lookup this only compares against the next
// boundary held in registers; descriptor memory is read again only when a segment
// ends (those reads would otherwise drain the load pipeline every step).
template <int NK, int NV>
__device__ __forceinline__ void seek(const AggParams& p, uint64_t tile, SegCursor<NK, NV>& c) {
  if (tile < c.next_tile_base) return;
  uint32_t seg = c.seg == 0xffffffffu ? find_seg(p.sd, tile) : c.seg + 1;
  while (seg + 1 < p.sd.n_segs && p.sd.segs[seg + 1].tile_base <= tile) ++seg;
  seg = __builtin_amdgcn_readfirstlane(seg);
  c.seg = seg;
  c.seg_rows = uniform_u64(p.sd.segs[seg].rows);
  c.tile_base = uniform_u64(p.sd.segs[seg].tile_base);
  c.next_tile_base = seg + 1 < p.sd.n_segs ? uniform_u64(p.sd.segs[seg + 1].tile_base) : ~0ull;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const ColPtr& cp = p.cols[(uint64_t)p.keys[k].col * p.sd.n_segs + seg];
    c.kval[k] = uniform_u64((uint64_t)cp.values);
    c.kvalid[k] = p.keys[k].null_bit >= 0 ? uniform_u64((uint64_t)cp.validity) : 0ull;
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const ColPtr& cp = p.cols[(uint64_t)p.vals[v].col * p.sd.n_segs + seg];
    c.vval[v] = uniform_u64((uint64_t)cp.values);
    c.vvalid[v] = uniform_u64((uint64_t)cp.validity);
  }
}

// kU values of one column: a uniform base (column start + tile start) plus a 32-bit
// per-lane byte offset.  One unaligned 8-byte load per value whatever the width: the
// bytes beyond the value belong to the next rows or to the allocation's padding and
// are dropped by widen().  Unaligned global loads are part of the amdhsa ABI.
__device__ __forceinline__ void load_rows(uint64_t col_base, uint64_t tile_row0, int width, const uint32_t (&rel)[kU],
                                          unsigned long long (&out)[kU]) {
  GlobalPtr<unsigned char> tp = (GlobalPtr<unsigned char>)(col_base + tile_row0 * (uint64_t)width);
#pragma unroll
  for (int u = 0; u < kU; ++u) out[u] = *(const __attribute__((address_space(1))) unaligned_u64*)(tp + rel[u] * (uint32_t)width);
}

// Validity bits of the kU rows (all ones when the column has no bitmap).
__device__ __forceinline__ uint32_t load_valid(uint64_t validity, uint64_t tile_row0, const uint32_t (&rel)[kU]) {
  if (validity == 0) return (1u << kU) - 1u;
  GlobalPtr<unsigned long long> b = (GlobalPtr<unsigned long long>)validity + (tile_row0 >> 6);
  unsigned long long w[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) w[u] = b[rel[u] >> 6];
  uint32_t m = 0;
#pragma unroll
  for (int u = 0; u < kU; ++u) m |= (uint32_t)((w[u] >> (rel[u] & 63)) & 1ull) << u;
  return m;
}

template <int NK, int NV>
__device__ __forceinline__ void issue(const AggParams& p, uint64_t j, SegCursor<NK, NV>& c, RowBatch<NK, NV>& b) {
  const uint64_t tile = j / kStepsPerTile;
  const uint32_t step = (uint32_t)(j % kStepsPerTile);
  seek(p, tile, c);
  const uint64_t tile_row0 = (tile - c.tile_base) * kTileRows;
  const uint64_t left = c.seg_rows - tile_row0;  // rows of the segment from this tile on (>= 1)
  const uint32_t last_rel = left > kTileRows ? kTileRows - 1 : (uint32_t)left - 1;
  const uint32_t rel0 = step * kU * kBlock + threadIdx.x;
  uint32_t rel[kU];
  uint32_t live = 0;
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    uint32_t r = rel0 + u * kBlock;
    live |= (r <= last_rel ? 1u : 0u) << u;
    rel[u] = r < last_rel ? r : last_rel;  // clamped: every lane loads, dead lanes discard
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) load_rows(c.kval[k], tile_row0, p.keys[k].width, rel, b.kraw[k]);
#pragma unroll
  for (int v = 0; v < NV; ++v) load_rows(c.vval[v], tile_row0, p.vals[v].width, rel, b.vraw[v]);
  if (p.row_mode == kRowsBitmap) {
    GlobalPtr<unsigned long long> sb = (GlobalPtr<unsigned long long>)p.sel_bits + tile * kTileWords;
    unsigned long long w[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) w[u] = sb[(rel0 + u * kBlock) >> 6];
    uint32_t m = 0;
#pragma unroll
    for (int u = 0; u < kU; ++u) m |= (uint32_t)((w[u] >> ((rel0 + u * kBlock) & 63)) & 1ull) << u;
    live &= m;
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) b.kok[k] = load_valid(c.kvalid[k], tile_row0, rel);
#pragma unroll
  for (int v = 0; v < NV; ++v) b.vok[v] = load_valid(c.vvalid[v], tile_row0, rel);
  b.live = live;
}

// Low 32 bits of a fixed-width value from the 8 bytes loaded at its address, without
// branching on the width: shift the value to the top of the dword and back down,
// arithmetically for signed types.
__device__ __forceinline__ uint32_t widen_lo(unsigned long long raw, int width, int is_unsigned) {
  const int sh = width >= 4 ? 0 : 16;
  const uint32_t x = (uint32_t)raw << sh;
  return is_unsigned ? (x >> sh) : (uint32_t)((int)x >> sh);
}

// The full value, also branch free.
__device__ __forceinline__ long long widen(unsigned long long raw, int width, int is_unsigned) {
  const uint32_t lo = widen_lo(raw, width, is_unsigned);
  const uint32_t ext = is_unsigned ? 0u : (uint32_t)((int)lo >> 31);
  const uint32_t hi = width == 8 ? (uint32_t)(raw >> 32) : ext;
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// n / d for a divisor described by (magic, pre, shift), including d == 1
// (magic 0, pre 0, shift 0), without a branch.
__device__ __forceinline__ uint32_t fast_div32(uint32_t n, uint32_t magic, int pre, int shift) {
  uint32_t q = __umulhi(magic, n);
  uint32_t t = ((n - q) >> pre) + q;
  return t >> shift;
}

template <int NK, int NV, bool K32>
__device__ __forceinline__ void convert(const AggParams& p, const RowBatch<NK, NV>& b, PackedRows<NK, NV, K32>& out) {
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    KeyT<K32> key = 0;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const KeySpec& ks = p.keys[k];
      // Both alternatives (field, NULL marker) are computed from values in registers and
      // selected by value: selecting between the addresses of two descriptor fields
      // would become a vector load from the kernel arguments and stall the pipeline.
      const int bit_pos = ks.bit_pos, null_bit = ks.null_bit;
      const KeyT<K32> null_field = null_bit >= 0 ? ((KeyT<K32>)1 << null_bit) : (KeyT<K32>)0;
      KeyT<K32> field;
      if (K32) {
        // every key of a K32 plan has the 32-bit form: (value - base) / divisor, with
        // divisor 1 for a plain column and the stride for date_bin
        field = fast_div32(widen_lo(b.kraw[k][u], ks.width, ks.is_unsigned) - (uint32_t)ks.base, ks.magic32, ks.pre32,
                           ks.shift32);
      } else {
        long long v = widen(b.kraw[k][u], ks.width, ks.is_unsigned);
        if (ks.date_bin) v = bucket_number(v, ks);
        field = (KeyT<K32>)((uint64_t)v - (uint64_t)ks.bias);
      }
      field <<= bit_pos;
      key |= ((b.kok[k] >> u) & 1u) ? field : null_field;
    }
    out.key[u] = key;
#pragma unroll
    for (int v = 0; v < NV; ++v) out.val[v][u] = widen(b.vraw[v][u], p.vals[v].width, p.vals[v].is_unsigned);
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) out.vok[v] = b.vok[v];
  out.live = b.live;
}

// Entry of `key` in the LDS table, or -1.  `insert` false: look-up only, a few probes
// (the table is frozen: it keeps the groups it has, everything else goes to HBM).
template <bool K32>
__device__ __forceinline__ int lds_entry(const Lds<K32>& t, KeyT<K32> key, int* n_entries, bool insert) {
  const KeyT<K32> empty = (KeyT<K32>)~(KeyT<K32>)0;
  const int sub = t.S >> t.rep_log2;  // entries per replica
  const int mask = sub - 1;
  // replicas are interleaved (entry = slot * replicas + replica) so that the copies of
  // one group sit in neighbouring banks rather than in the same one
  const int rep = (int)(threadIdx.x & ((1u << t.rep_log2) - 1u));
  uint32_t mixed = K32 ? (uint32_t)key : ((uint32_t)key ^ (uint32_t)((uint64_t)key >> 32));
  int h = (int)((mixed * 0x9E3779B1u) >> 12) & mask;
  const int max_probes = insert ? kLdsProbes : 4;
  for (int probe = 0; probe < max_probes; ++probe) {
    const int e = (h << t.rep_log2) + rep;
    KeyT<K32> k = t.keys[e];
    if (k == key) return e;
    if (k == empty) {
      if (!insert) return -1;
      KeyT<K32> old;
      if (K32) old = (KeyT<K32>)atomicCAS((unsigned int*)&t.keys[e], 0xffffffffu, (unsigned int)key);
      else old = (KeyT<K32>)atomicCAS((unsigned long long*)&t.keys[e], ~0ull, (unsigned long long)key);
      if (old == empty) { atomicAdd(n_entries, 1); return e; }
      if (old == key) return e;
    }
    h = (h + 1) & mask;
  }
  return -1;
}

// Applies the kU rows of a lane.  Slot-major: the LDS entries of all rows are looked
// up first (independent probe chains, so their LDS latencies overlap), then every slot
// is decoded once and applied to all rows, so the per-slot dispatch is paid per step,
// not per row.  Rows that found the LDS table full fall through to the HBM table.
template <int NK, int NV, bool K32>
__device__ __forceinline__ void consume(const AggParams& p, const Lds<K32>& t, bool insert, int* n_entries,
                                        const PackedRows<NK, NV, K32>& b) {
  int e[kU];  // LDS entry, -1 = go to HBM, -2 = row not live
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    e[u] = -2;
    if ((b.live >> u) & 1u) {
      // a 64-bit key equal to the empty marker only lives in the HBM table's extra entry
      const bool lds_ok = K32 || b.key[u] != (KeyT<K32>)kEmptyKey;
      e[u] = lds_ok ? lds_entry<K32>(t, b.key[u], n_entries, insert) : -1;
    }
  }
  for (int s = 0; s < p.n_slots; ++s) {
    const int op = p.slots[s].op, val = p.slots[s].val;
    long long* base = t.state + s * t.S;
    if (op == kSlotAddOne) {
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (e[u] >= 0) atomicAdd((unsigned long long*)&base[e[u]], 1ull);
      continue;
    }
    // the slot's input column, picked by a select chain (a runtime index would put the
    // values in scratch memory)
    uint32_t ok = 0;
    long long x[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) x[u] = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (v == val) {
        ok = b.vok[v];
#pragma unroll
        for (int u = 0; u < kU; ++u) x[u] = b.val[v][u];
      }
    }
    const bool is_float = val >= 0 && p.vals[val].is_float;
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (e[u] < 0 || !((ok >> u) & 1u)) continue;
      long long* addr = &base[e[u]];
      if (op == kSlotAddValid) atomicAdd((unsigned long long*)addr, 1ull);
      else if (op == kSlotAddI64) atomicAdd((unsigned long long*)addr, (unsigned long long)x[u]);
      else if (op == kSlotMax) atomicMax(addr, x[u]);
      else if (op == kSlotMin) atomicMin(addr, x[u]);
      else atomicAdd((double*)addr, is_float ? __longlong_as_double(x[u]) : (double)x[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    if (e[u] != -1) continue;
    RowValsT<NV> rv;
    rv.valid = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      rv.v[v] = b.val[v][u];
      rv.valid |= ((b.vok[v] >> u) & 1u) << v;
    }
    global_update(p, (uint64_t)b.key[u], rv);
  }
}

template <bool K32>
__device__ __forceinline__ void lds_clear(const AggParams& p, const Lds<K32>& t) {
  for (int i = threadIdx.x; i < t.S; i += blockDim.x) {
    t.keys[i] = (KeyT<K32>)~(KeyT<K32>)0;
    for (int s = 0; s < p.n_slots; ++s) t.state[s * t.S + i] = slot_identity(p.slots[s].op);
  }
}

// Folds the LDS table into the HBM table and empties it.  Block-wide; the caller
// brackets it with barriers.
template <bool K32>
__device__ __forceinline__ void lds_flush(const AggParams& p, const Lds<K32>& t) {
  const KeyT<K32> empty = (KeyT<K32>)~(KeyT<K32>)0;
  for (int i = threadIdx.x; i < t.S; i += blockDim.x) {
    KeyT<K32> key = t.keys[i];
    if (key == empty) continue;
    uint64_t e = global_entry(p, (uint64_t)key);
    for (int s = 0; s < p.n_slots; ++s) {
      int op = p.slots[s].op;
      long long v = t.state[s * t.S + i];
      if (e != ~0ull && v != slot_identity(op)) atomic_apply(&p.g_state[(uint64_t)s * (p.capacity + 1) + e], op, v);
      t.state[s * t.S + i] = slot_identity(op);
    }
    t.keys[i] = empty;
  }
}

template <int NK, int NV, bool K32>
__global__ __launch_bounds__(kBlock) void k_group_by_stream(AggParams p) {
  extern __shared__ unsigned long long lds_raw[];
  __shared__ int n_entries;
  constexpr int kInitialRepLog2 = 0;
  Lds<K32> t;
  t.S = p.lds_slots;
  t.rep_log2 = kInitialRepLog2;
  t.state = (long long*)lds_raw;
  t.keys = (KeyT<K32>*)(lds_raw + (size_t)p.n_slots * t.S);
  if (threadIdx.x == 0) n_entries = 0;
  lds_clear<K32>(p, t);
  __syncthreads();

  const uint64_t per = ceil_div(p.sd.n_tiles, (uint64_t)gridDim.x);
  const uint64_t t0 = per * blockIdx.x, t1 = t0 + per < p.sd.n_tiles ? t0 + per : p.sd.n_tiles;
  const uint64_t j0 = t0 * kStepsPerTile, j1 = t1 * kStepsPerTile;
  uint64_t rows_since_flush = 0;
  bool insert = true;  // false once the table is frozen

  SegCursor<NK, NV> cur;
  RowBatch<NK, NV> raw;
  PackedRows<NK, NV, K32> rows;
  if (j0 < j1) issue(p, j0, cur, raw);
  for (uint64_t j = j0; j < j1; ++j) {
    convert<NK, NV, K32>(p, raw, rows);
    // The scheduling barriers keep the next step's loads from being hoisted above the
    // conversion (which would make the conversion wait for them too) and from being
    // sunk into the hashing (which would shorten the time they have to land).
    __builtin_amdgcn_sched_barrier(0);
    if (j + 1 < j1) issue(p, j + 1, cur, raw);
    __builtin_amdgcn_sched_barrier(0);
    consume<NK, NV, K32>(p, t, insert, &n_entries, rows);
    if (insert && (j + 1) % kStepsPerTile == 0) {
      // Block-wide bookkeeping at the end of a tile, when the LDS table is half full.
      // Keys that repeat a lot: fold the table into HBM and start over (with fewer
      // copies per group if replicas were in use).  Keys that hardly repeat (high
      // cardinality): freeze the table instead.  It then keeps the groups it has, which
      // are the frequent ones (they show up first), so the hot groups still cost HBM no
      // per-row atomics on one address, and every other row goes straight to HBM.
      rows_since_flush += kTileRows;
      __syncthreads();
      const int n = n_entries;
      __syncthreads();
      if (n > t.S / 2) {
        if (t.rep_log2 == 0 && (uint64_t)n * 4 > rows_since_flush) {
          insert = false;
        } else {
          lds_flush<K32>(p, t);
          if (t.rep_log2 > 0) t.rep_log2 -= 1;
          if (threadIdx.x == 0) n_entries = 0;
          rows_since_flush = 0;
          __syncthreads();
        }
      }
    }
  }
  __syncthreads();
  lds_flush<K32>(p, t);
}

template <int NK, bool K32>
GroupByKernel pick_nv(int nv) {
  switch (nv) {
    case 0: return k_group_by_stream<NK, 0, K32>;
    case 1: return k_group_by_stream<NK, 1, K32>;
    case 2: return k_group_by_stream<NK, 2, K32>;
    case 3: return k_group_by_stream<NK, 3, K32>;
    case 4: return k_group_by_stream<NK, 4, K32>;
    default: return nullptr;
  }
}

template <bool K32>
GroupByKernel pick_nk(int nk, int nv) {
  switch (nk) {
    case 1: return pick_nv<1, K32>(nv);
    case 2: return pick_nv<2, K32>(nv);
    case 3: return pick_nv<3, K32>(nv);
    case 4: return pick_nv<4, K32>(nv);
    default: return nullptr;
  }
}

}  // namespace

GroupByKernel pick_stream_kernel(const AggParams& p, size_t* lds_bytes) {
  bool k32 = p.key_bits <= 31;
  for (int k = 0; k < p.n_keys; ++k) k32 = k32 && p.keys[k].fast32 != 0;
  *lds_bytes = (size_t)p.lds_slots * (8 * (size_t)p.n_slots + (k32 ? 4 : 8));
  return k32 ? pick_nk<true>(p.n_keys, p.n_vals) : pick_nk<false>(p.n_keys, p.n_vals);
}

}  // namespace prism
