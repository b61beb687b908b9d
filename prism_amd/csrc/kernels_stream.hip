// Streaming GROUP BY: the kernel that runs when every row (or a bitmap-selected subset)
// of the scanned segments is aggregated.
//
// Per step a wave handles kU runs of 64 consecutive rows (row per lane, so a wave
// instruction reads one contiguous run of each column, and the validity / selection
// bits of a run are one 64-bit word that is used directly as a lane mask).  The step is
// software pipelined:
//   1. convert: wait for the batch in flight and reduce it, in registers, to one packed
//      key per row plus the widened aggregate inputs;
//   2. issue:   start the loads of the next step (every column, kU rows, one unaligned
//      8-byte load per value whatever the column width, so no branch separates them);
//   3. consume: find the key in the workgroup's LDS table and apply the slot updates
//      with LDS atomics while the loads of (2) are in flight.
// The instruction count per row is what bounds this kernel once the loads are batched,
// so the steady state avoids per-lane control flow altogether: the LDS table is 4-way
// set associative (one 16-byte read returns the four candidate keys, no probe loop),
// rows that must not touch the table (dead lanes, misses that go to HBM) update a
// per-lane scratch entry instead of being masked off, and NULL inputs are replaced by
// the slot's identity.  Inserting a new key, and the HBM fallback, are out-of-line
// paths entered by a wave-uniform branch.
// The kernel is specialised on the number of key columns (NK), of aggregate input
// columns (NV) and on whether the packed key fits in 31 bits (K32: 32-bit key
// arithmetic, 32-bit LDS keys, and date_bin as one 32-bit multiply-high when the
// planner found that the timestamps allow it).
#include <algorithm>
#include <type_traits>
#include <utility>

#include "agg_device.hpp"

#ifndef PRISM_STREAM_BLOCK
#define PRISM_STREAM_BLOCK 512
#endif
#ifndef PRISM_STREAM_REPLICAS
#define PRISM_STREAM_REPLICAS 16
#endif
#ifndef PRISM_WINDOW_REP_LOG2
#define PRISM_WINDOW_REP_LOG2 2
#endif
#ifndef PRISM_STREAM_LDS_KB
#define PRISM_STREAM_LDS_KB 72
#endif

namespace prism {

namespace {

constexpr int kU = 4;
constexpr uint32_t kWaveRows = 64 * kU;     // rows of one wave per step
// Threads per workgroup.  All waves of a workgroup share one LDS table, so for a given
// amount of LDS per CU a larger workgroup means a larger table, hence more replicas.
constexpr uint32_t kStreamBlock = PRISM_STREAM_BLOCK;
constexpr uint32_t kStepRows = kStreamBlock * kU;  // rows of the workgroup per step
constexpr uint32_t kStepsPerTile = kTileRows / kStepRows;
constexpr int kWays = 4;      // entries per LDS bucket
constexpr int kScratchEntries = 64;  // one per lane, behind the table proper
constexpr int kInitialReplicas = PRISM_STREAM_REPLICAS;
constexpr int kWindowRepLog2 = PRISM_WINDOW_REP_LOG2;    // log2 of the replicas of the window mode  // halved whenever the table fills up
constexpr uint32_t kReplicaSalt = 0x7FEB352Du;

// Pointers read from descriptor arrays are generic; telling the compiler that they
// point to global memory turns the column reads into global_load instructions (a flat
// load also has to wait for LDS traffic).
template <typename T>
using GlobalPtr = const __attribute__((address_space(1))) T*;
typedef unsigned long long __attribute__((aligned(1))) unaligned_u64;
typedef uint32_t __attribute__((aligned(1))) unaligned_u32;

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// Lane predicate from a wave-uniform 64-bit mask (bit = lane).
__device__ __forceinline__ bool lane_bit(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

template <bool K32>
using KeyT = std::conditional_t<K32, uint32_t, uint64_t>;

// The workgroup's LDS table: S entries in buckets of kWays, then kScratchEntries
// entries that nobody reads.  state[slot][stride], keys[S].
template <bool K32>
struct Lds {
  long long* state;
  KeyT<K32>* keys;
  int S;
  int stride;        // S + kScratchEntries
  int bucket_shift;  // 32 - log2(S / kWays)
  // Replicas: a lane adds `salt` (a multiple of its lane number modulo the replica
  // count) to the key's hash, so the rows of one group that meet in one wave
  // instruction land in up to `replicas` different entries.  LDS atomics of a wave are
  // processed 16 lanes at a time and lanes of those 16 that hit the same address are
  // applied one after the other (2 cycles each), which is what bounds this kernel when
  // a few groups are hot.  Replicas of a group are merged when the table is flushed.
  int replicas;
  uint32_t salt;
  // Small key spaces (every packed key times a few replicas fits the table) skip hashing
  // altogether: replica r = lane mod 2^direct_rep_log2 of key k lives at entry
  // (r << key_bits) | ((k + 5 r) mod 2^key_bits).  The rotation by 5 r puts the replicas of
  // one key, and different keys of one replica, in different LDS banks (an entry's bank is
  // given by its low index bits).  -1: hashed.
  int direct_rep_log2;
  int direct_key_bits;
  // Window mode: the same direct indexing for a key space that is too large as a whole but
  // narrow at any one time (a time bucket in the high bits of the key, input in time
  // order): only keys in [win_base, win_base + 2^direct_key_bits) are indexed, the others
  // go to the HBM table, and when too many do the table is flushed and the window moved.
  uint32_t win_base;
  bool windowed;
};

// K32 plans only ever use the low 32 bits of a key value, so they load only those: a
// register that is loaded but never read gets reused by the compiler, and the write
// after the load then has to wait for the load to land.
template <int NK, int NV, bool K32>
struct RowBatch {
  static constexpr int NKA = NK > 0 ? NK : 1, NVA = NV > 0 ? NV : 1;
  std::conditional_t<K32, uint32_t, unsigned long long> kraw[NKA][kU];
  unsigned long long vraw[NVA][kU];
  // validity / selection words of the wave's kU runs (the same value in every lane)
  unsigned long long kokw[NKA][kU], vokw[NVA][kU], selw[kU];
  bool knull[NKA], vnull[NVA];  // the column has a validity bitmap in this segment
  uint64_t tail[kU];            // lanes whose row exists
};

template <int NK, int NV, bool K32>
struct PackedRows {
  static constexpr int NVA = NV > 0 ? NV : 1;
  KeyT<K32> key[kU];
  long long val[NVA][kU];
  uint64_t vok[NVA][kU];  // lane masks
  bool vnull[NVA];
  uint64_t live[kU];      // lanes whose row exists and is selected
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

// kU values of one column: a uniform base (column start + tile start, in scalar
// registers) plus a 32-bit per-lane byte offset.  One unaligned load of sizeof(T) bytes
// per value whatever the column width: the bytes beyond the value belong to the next
// rows or to the allocation's padding and are dropped by widen().  Unaligned global
// loads are part of the amdhsa ABI.  Widths are powers of two, so the offsets are shifts.
// The loads are marked nontemporal: every column byte is read exactly once, and on this
// part a streaming read that does not ask the caches to keep its lines is about 10 %
// faster (scripts/microbench/stream_read.hip).
template <typename T>
__device__ __forceinline__ void load_rows(uint64_t col_base, uint64_t tile_row0, int width, const uint32_t (&rel)[kU],
                                          T (&out)[kU]) {
  typedef T __attribute__((aligned(1))) unaligned_t;
  const int lw = __builtin_ctz((unsigned)width);
  GlobalPtr<unsigned char> tp = (GlobalPtr<unsigned char>)(col_base + (tile_row0 << lw));
#pragma unroll
  for (int u = 0; u < kU; ++u)
    out[u] = __builtin_nontemporal_load((const __attribute__((address_space(1))) unaligned_t*)(tp + (rel[u] << lw)));
}

// The kU consecutive bitmap words that cover the wave's rows of this step.  The address
// is wave-uniform; bitmaps are allocated in whole tiles, so the words exist even where
// the rows do not.
__device__ __forceinline__ void load_words(uint64_t bitmap, uint64_t first_word, unsigned long long (&out)[kU]) {
  GlobalPtr<unsigned long long> w = (GlobalPtr<unsigned long long>)bitmap + first_word;
#pragma unroll
  for (int u = 0; u < kU; ++u) out[u] = w[u];
}

template <int NK, int NV, bool K32>
__device__ __forceinline__ void issue(const AggParams& p, uint64_t j, uint32_t wave, uint32_t lane, SegCursor<NK, NV>& c,
                                      RowBatch<NK, NV, K32>& b) {
  const uint64_t tile = j / kStepsPerTile;
  const uint32_t step = (uint32_t)(j % kStepsPerTile);
  seek(p, tile, c);
  const uint64_t tile_row0 = (tile - c.tile_base) * kTileRows;
  const uint64_t left = c.seg_rows - tile_row0;  // rows of the segment from this tile on (>= 1)
  const uint32_t last_rel = left > kTileRows ? kTileRows - 1 : (uint32_t)left - 1;
  const uint32_t wave_rel0 = step * kStepRows + wave * kWaveRows;  // uniform
  uint32_t rel[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const uint32_t r = wave_rel0 + u * 64 + lane;
    b.tail[u] = __ballot(r <= last_rel);
    rel[u] = r < last_rel ? r : last_rel;  // clamped: every lane loads, dead lanes discard
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) load_rows(c.kval[k], tile_row0, p.keys[k].width, rel, b.kraw[k]);
#pragma unroll
  for (int v = 0; v < NV; ++v) load_rows(c.vval[v], tile_row0, p.vals[v].width, rel, b.vraw[v]);
  const uint64_t word0 = wave_rel0 >> 6;
  if (p.row_mode == kRowsBitmap) load_words((uint64_t)p.sel_bits, tile * kTileWords + word0, b.selw);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    b.knull[k] = c.kvalid[k] != 0;
    if (b.knull[k]) load_words(c.kvalid[k], (tile_row0 >> 6) + word0, b.kokw[k]);
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    b.vnull[v] = c.vvalid[v] != 0;
    if (b.vnull[v]) load_words(c.vvalid[v], (tile_row0 >> 6) + word0, b.vokw[v]);
  }
}

// Low 32 bits of a fixed-width value from the 8 bytes loaded at its address, without
// branching on the width: shift the value to the top of the dword and back down,
// arithmetically for signed types.
__device__ __forceinline__ uint32_t widen_lo(uint32_t raw, int width, int is_unsigned) {
  const int sh = width >= 4 ? 0 : 16;
  const uint32_t x = raw << sh;
  return is_unsigned ? (x >> sh) : (uint32_t)((int)x >> sh);
}

// The full value, also branch free.
__device__ __forceinline__ long long widen(unsigned long long raw, int width, int is_unsigned) {
  const uint32_t lo = widen_lo((uint32_t)raw, width, is_unsigned);
  const uint32_t ext = is_unsigned ? 0u : (uint32_t)((int)lo >> 31);
  const uint32_t hi = width == 8 ? (uint32_t)(raw >> 32) : ext;
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// n / d for a divisor described by (magic, pre, shift).
__device__ __forceinline__ uint32_t fast_div32(uint32_t n, uint32_t magic, int pre, int shift) {
  uint32_t q = __umulhi(magic, n);
  uint32_t t = ((n - q) >> pre) + q;
  return t >> shift;
}

// WHERE inside the kernel: every leaf compares one of the loaded value columns with a
// literal (or asks for its validity); a comparison of 64 rows is one v_cmp whose result is
// the lane mask itself, and masks combine on the scalar unit.  TRUE rows only: a NULL input
// clears the lane, which is what FilterExec keeps under Kleene logic once NOT has been
// pushed into the leaves.
//
// The leaves live in scalar registers for the whole kernel (FusedRegs, filled once and made
// opaque to the compiler): read from the kernel arguments where they are used, every step
// would start with scalar loads and an s_waitcnt lgkmcnt(0), which also waits for the LDS
// atomics of the step before (measured: +25 % on the headline shape).
struct FusedRegs {
  int n, is_or;
  int val[kMaxFusedLeaves], kind[kMaxFusedLeaves], cmp_mask[kMaxFusedLeaves], base32[kMaxFusedLeaves];
  long long lit[kMaxFusedLeaves];
};

__device__ __forceinline__ int pin_scalar(int v) {
  v = __builtin_amdgcn_readfirstlane(v);
  asm volatile("" : "+s"(v));
  return v;
}

template <bool K32>
__device__ __forceinline__ void load_fused_regs(const AggParams& p, FusedRegs& r) {
  r.n = pin_scalar(p.n_pred);
  r.is_or = pin_scalar(p.pred_or);
#pragma unroll
  for (int l = 0; l < kMaxFusedLeaves; ++l) {
    r.val[l] = pin_scalar(p.pred[l].val);
    r.kind[l] = pin_scalar(p.pred[l].kind);
    r.cmp_mask[l] = pin_scalar(p.pred[l].cmp_mask);
    r.base32[l] = pin_scalar((int)p.pred[l].key_base32);
    // a leaf on a key of a 32-bit-key kernel compares in the key's 32-bit domain
    const long long lit = (K32 && p.pred[l].val < 0) ? p.pred[l].lit_k32 : p.pred[l].lit;
    const int lo = pin_scalar((int)(uint32_t)(unsigned long long)lit);
    const int hi = pin_scalar((int)(uint32_t)((unsigned long long)lit >> 32));
    r.lit[l] = (long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
  }
}

template <int L, int NK, int NV, bool K32>
__device__ __forceinline__ void fused_leaf(const FusedRegs& r, const RowBatch<NK, NV, K32>& raw, const PackedRows<NK, NV, K32>& rows,
                                           uint64_t (&mask)[kU]) {
  if (L >= r.n) return;
  const long long lit = r.lit[L];
  // the leaf's input: kU values and the lanes where it is not NULL
  long long x[kU];
  uint64_t ok[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) { x[u] = 0; ok[u] = 0; }
  if (r.val[L] >= 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (v != r.val[L]) continue;
#pragma unroll
      for (int u = 0; u < kU; ++u) { x[u] = rows.val[v][u]; ok[u] = rows.vok[v][u]; }
    }
  } else {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      if (k != -r.val[L] - 1) continue;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (K32) x[u] = (long long)(unsigned long long)(uint32_t)((uint32_t)raw.kraw[k][u] - (uint32_t)r.base32[L]);
        else x[u] = (long long)raw.kraw[k][u];
        ok[u] = raw.knull[k] ? uniform_u64(raw.kokw[k][u]) : ~0ull;
      }
    }
  }
  uint64_t m[kU];
  // one scalar branch per leaf and step picks the comparison; the rows are then one v_cmp each
  if (r.kind[L] == kLeafIsNull) {
#pragma unroll
    for (int u = 0; u < kU; ++u) m[u] = ~ok[u];
  } else if (r.kind[L] == kLeafIsNotNull) {
#pragma unroll
    for (int u = 0; u < kU; ++u) m[u] = ok[u];
  } else if (r.kind[L] == kLeafLut) {
    // a string predicate, decided per dictionary value on the host: bit `code` of the literal
#pragma unroll
    for (int u = 0; u < kU; ++u)
      m[u] = __ballot((unsigned long long)x[u] < 64ull && (((unsigned long long)lit >> ((unsigned)x[u] & 63u)) & 1ull) != 0) & ok[u];
  } else {
#pragma unroll
    for (int u = 0; u < kU; ++u) m[u] = 0;
    switch (r.cmp_mask[L]) {
      case kLt:
#pragma unroll
        for (int u = 0; u < kU; ++u) m[u] = __ballot(x[u] < lit);
        break;
      case kLt | kEq:
#pragma unroll
        for (int u = 0; u < kU; ++u) m[u] = __ballot(x[u] <= lit);
        break;
      case kEq:
#pragma unroll
        for (int u = 0; u < kU; ++u) m[u] = __ballot(x[u] == lit);
        break;
      case kLt | kGt:
#pragma unroll
        for (int u = 0; u < kU; ++u) m[u] = __ballot(x[u] != lit);
        break;
      case kGt | kEq:
#pragma unroll
        for (int u = 0; u < kU; ++u) m[u] = __ballot(x[u] >= lit);
        break;
      case kGt:
#pragma unroll
        for (int u = 0; u < kU; ++u) m[u] = __ballot(x[u] > lit);
        break;
      case kLt | kEq | kGt:
#pragma unroll
        for (int u = 0; u < kU; ++u) m[u] = ~0ull;
        break;
      default: break;  // no ordering satisfies it: never TRUE
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) m[u] &= ok[u];
  }
  if (r.is_or) {
#pragma unroll
    for (int u = 0; u < kU; ++u) mask[u] |= m[u];
  } else {
#pragma unroll
    for (int u = 0; u < kU; ++u) mask[u] &= m[u];
  }
}

template <int NK, int NV, bool K32, int... L>
__device__ __forceinline__ void fused_predicate_leaves(const FusedRegs& r, const RowBatch<NK, NV, K32>& raw,
                                                       const PackedRows<NK, NV, K32>& rows, uint64_t (&mask)[kU],
                                                       std::integer_sequence<int, L...>) {
  (fused_leaf<L, NK, NV, K32>(r, raw, rows, mask), ...);
}

// Clears the rows that fail the predicate from rows.live and returns how many remain.
template <int NK, int NV, bool K32>
__device__ __forceinline__ uint32_t fused_predicate(const FusedRegs& r, const RowBatch<NK, NV, K32>& raw, PackedRows<NK, NV, K32>& rows) {
  uint64_t mask[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) mask[u] = r.is_or ? 0ull : ~0ull;
  fused_predicate_leaves<NK, NV, K32>(r, raw, rows, mask, std::make_integer_sequence<int, kMaxFusedLeaves>{});
  uint32_t n = 0;
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    rows.live[u] &= mask[u];
    n += (uint32_t)__popcll(rows.live[u]);
  }
  return n;
}

// Branches below are on kernel arguments or on per-segment facts, so they are scalar
// branches taken once per step (kU rows), not per row.
template <int NK, int NV, bool K32>
__device__ __forceinline__ void convert(const AggParams& p, const RowBatch<NK, NV, K32>& b, PackedRows<NK, NV, K32>& out) {
#pragma unroll
  for (int u = 0; u < kU; ++u) out.key[u] = 0;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const KeySpec& ks = p.keys[k];
    KeyT<K32> field[kU];
    if (K32) {
      // every key of a K32 plan has the 32-bit form: (value - base) / divisor, with
      // divisor 1 for a plain column and the stride for date_bin
      if (ks.width >= 4) {
#pragma unroll
        for (int u = 0; u < kU; ++u) field[u] = (uint32_t)b.kraw[k][u] - (uint32_t)ks.base;
      } else {
#pragma unroll
        for (int u = 0; u < kU; ++u) field[u] = widen_lo((uint32_t)b.kraw[k][u], ks.width, ks.is_unsigned) - (uint32_t)ks.base;
      }
      if (ks.date_bin) {
#pragma unroll
        for (int u = 0; u < kU; ++u) field[u] = fast_div32((uint32_t)field[u], ks.magic32, ks.pre32, ks.shift32);
      }
    } else {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        long long v = widen(b.kraw[k][u], ks.width, ks.is_unsigned);
        if (ks.date_bin) v = bucket_number(v, ks);
        field[u] = (KeyT<K32>)((uint64_t)v - (uint64_t)ks.bias);
      }
    }
    const int bit_pos = ks.bit_pos;
#pragma unroll
    for (int u = 0; u < kU; ++u) field[u] <<= bit_pos;
    if (b.knull[k]) {
      const KeyT<K32> null_field = (KeyT<K32>)1 << (ks.null_bit >= 0 ? ks.null_bit : 0);
#pragma unroll
      for (int u = 0; u < kU; ++u) field[u] = lane_bit(uniform_u64(b.kokw[k][u])) ? field[u] : null_field;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) out.key[u] |= field[u];
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int width = p.vals[v].width, is_unsigned = p.vals[v].is_unsigned;
    if (width == 8) {
#pragma unroll
      for (int u = 0; u < kU; ++u) out.val[v][u] = (long long)b.vraw[v][u];
    } else {
#pragma unroll
      for (int u = 0; u < kU; ++u) out.val[v][u] = widen(b.vraw[v][u], width, is_unsigned);
    }
    out.vnull[v] = b.vnull[v];
#pragma unroll
    for (int u = 0; u < kU; ++u) out.vok[v][u] = b.vnull[v] ? uniform_u64(b.vokw[v][u]) : ~0ull;
  }
#pragma unroll
  for (int u = 0; u < kU; ++u) out.live[u] = p.row_mode == kRowsBitmap ? (b.tail[u] & uniform_u64(b.selw[u])) : b.tail[u];
}

// Hash of a key (and the lane's replica): the top bits pick the bucket.
template <bool K32>
__device__ __forceinline__ uint32_t lds_hash(const Lds<K32>& t, KeyT<K32> key) {
  const uint32_t mixed = K32 ? (uint32_t)key : ((uint32_t)key ^ (uint32_t)((uint64_t)key >> 32));
  return mixed * 0x9E3779B1u + t.salt;
}

// Out-of-line part of the look-up: finds the key in, or inserts it into, the first
// bucket with room at or after its home bucket (the inline part only looks at the home
// bucket, so a displaced key comes through here every time; few are, since the table is
// flushed when half full).  Returns the entry, or -1 when kProbeBuckets buckets are all
// taken by other keys (the row then goes to HBM).
constexpr int kProbeBuckets = 4;
template <bool K32>
__device__ __noinline__ int lds_insert(KeyT<K32>* keys, int first, int mask, KeyT<K32> key, int* n_entries, bool insert) {
  const KeyT<K32> empty = (KeyT<K32>)~(KeyT<K32>)0;
  // Ways are tried in order.  (Starting at a way derived from the key's hash, to spread the
  // entries over all LDS banks, measured 15-45 % slower.)
  for (int b = 0; b < kProbeBuckets; ++b) {
    for (int w = 0; w < kWays; ++w) {
      const int e = first + w;
      KeyT<K32> k = keys[e];
      if (k == empty) {
        if (!insert) return -1;
        if (K32) k = (KeyT<K32>)atomicCAS((unsigned int*)&keys[e], 0xffffffffu, (unsigned int)key);
        else k = (KeyT<K32>)atomicCAS((unsigned long long*)&keys[e], ~0ull, (unsigned long long)key);
        if (k == empty) { atomicAdd(n_entries, 1); return e; }
      }
      if (k == key) return e;
    }
    first = (first + kWays) & mask;
  }
  return -1;
}

// One slot, kU rows: NULL inputs are replaced by the slot's identity instead of being
// masked off.
__device__ __forceinline__ void apply_slot(int op, char* base, const uint32_t (&off)[kU], const long long (&val)[kU],
                                           const uint64_t (&ok)[kU], bool has_null, bool is_float) {
  long long x[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) x[u] = val[u];
  if (op == kSlotAddValid) {
#pragma unroll
    for (int u = 0; u < kU; ++u) atomicAdd((unsigned long long*)(base + off[u]), lane_bit(ok[u]) ? 1ull : 0ull);
  } else if (op == kSlotAddI64) {
    if (has_null) {
#pragma unroll
      for (int u = 0; u < kU; ++u) x[u] = lane_bit(ok[u]) ? x[u] : 0ll;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) atomicAdd((unsigned long long*)(base + off[u]), (unsigned long long)x[u]);
  } else if (op == kSlotMax) {
    if (has_null) {
#pragma unroll
      for (int u = 0; u < kU; ++u) x[u] = lane_bit(ok[u]) ? x[u] : kI64Min;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) atomicMax((long long*)(base + off[u]), x[u]);
  } else if (op == kSlotMin) {
    if (has_null) {
#pragma unroll
      for (int u = 0; u < kU; ++u) x[u] = lane_bit(ok[u]) ? x[u] : kI64Max;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) atomicMin((long long*)(base + off[u]), x[u]);
  } else {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      double d = is_float ? __longlong_as_double(x[u]) : (double)x[u];
      d = lane_bit(ok[u]) ? d : -0.0;  // x + -0.0 == x for every x, signed zeros included
      atomicAdd((double*)(base + off[u]), d);
    }
  }
}

// Block-shared bookkeeping of the window mode.
struct WindowShared {
  uint32_t misses;     // wave instructions' rows that fell outside the window since the last check
  uint32_t candidate;  // the key of one such row
};

// Applies the kU rows of a lane.
template <int NK, int NV, bool K32>
__device__ __forceinline__ void consume(const AggParams& p, const Lds<K32>& t, bool insert, int* n_entries, WindowShared* win,
                                        uint32_t lane, const PackedRows<NK, NV, K32>& b) {
  const KeyT<K32> empty = (KeyT<K32>)~(KeyT<K32>)0;
  int e[kU];
  bool miss = false;
  if (t.direct_rep_log2 >= 0) {
    // directly indexed: no look-up at all
    const uint32_t rep = lane & ((1u << t.direct_rep_log2) - 1u);
    const uint32_t kmask = (1u << t.direct_key_bits) - 1u, base = rep << t.direct_key_bits, rot = rep * 5u;
    if (!t.windowed) {
      // the whole key space is indexed
#pragma unroll
      for (int u = 0; u < kU; ++u) e[u] = (int)(base | (((uint32_t)b.key[u] + rot) & kmask));
    } else {
      bool outside = false;
      uint32_t outside_key = 0;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t rel = (uint32_t)b.key[u] - t.win_base;
        const bool in = rel <= kmask;
        e[u] = in ? (int)(base | ((rel + rot) & kmask)) : -1;
        if (!in && lane_bit(b.live[u])) { outside = true; outside_key = (uint32_t)b.key[u]; }
      }
      const uint64_t om = __ballot(outside);
      if (om) {  // rare: note how many rows fell outside the window, and where one of them was
        if (lane == (uint32_t)__ffsll((long long)om) - 1u) {
          atomicAdd(&win->misses, (uint32_t)__popcll(om));
          win->candidate = outside_key;
        }
      }
    }
  } else {
  // 1. the four candidate keys of every row's bucket (independent reads, issued together)
  int first[kU];
  uint32_t hash[kU];
  KeyT<K32> cand[kU][kWays];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    hash[u] = lds_hash<K32>(t, b.key[u]);
    first[u] = (int)(hash[u] >> t.bucket_shift) * kWays;
    if (K32) {
      const uint4 c = *(const uint4*)&t.keys[first[u]];
      cand[u][0] = (KeyT<K32>)c.x; cand[u][1] = (KeyT<K32>)c.y; cand[u][2] = (KeyT<K32>)c.z; cand[u][3] = (KeyT<K32>)c.w;
    } else {
      const ulonglong2 c0 = *(const ulonglong2*)&t.keys[first[u]], c1 = *(const ulonglong2*)&t.keys[first[u] + 2];
      cand[u][0] = (KeyT<K32>)c0.x; cand[u][1] = (KeyT<K32>)c0.y; cand[u][2] = (KeyT<K32>)c1.x; cand[u][3] = (KeyT<K32>)c1.y;
    }
  }
  // 2. the matching way, or -1
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    int way = -1;
#pragma unroll
    for (int w = kWays - 1; w >= 0; --w) way = cand[u][w] == b.key[u] ? w : way;
    // a 64-bit key equal to the empty marker only lives in the HBM table's extra entry
    if (!K32 && b.key[u] == empty) way = -1;
    const bool live = lane_bit(b.live[u]);
    e[u] = way >= 0 ? first[u] + way : -1;
    miss |= live && way < 0;
  }
  if (__ballot(miss) != 0ull) {
    // some lane met a key the table does not hold: insert it (or leave it to HBM)
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (lane_bit(b.live[u]) && e[u] < 0 && (K32 || b.key[u] != empty))
        e[u] = lds_insert<K32>(t.keys, first[u], t.S - 1, b.key[u], n_entries, insert);
    }
  }
  }
  // 3. byte offset of the entry to update: the row's, or the lane's scratch entry
  uint32_t off[kU];
  bool to_hbm = false;
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const bool live = lane_bit(b.live[u]);
    const bool hit = live && e[u] >= 0;
    off[u] = (uint32_t)(hit ? e[u] : t.S + (int)lane) * 8u;
    to_hbm |= live && e[u] < 0;
  }
  // 4. slot major: every slot is decoded once and applied to all rows
  char* const state = (char*)t.state;
  for (int s = 0; s < p.n_slots; ++s) {
    const int op = p.slots[s].op, val = p.slots[s].val;
    char* const base = state + (uint32_t)(s * t.stride) * 8u;
    if (op == kSlotAddOne) {
#pragma unroll
      for (int u = 0; u < kU; ++u) atomicAdd((unsigned long long*)(base + off[u]), 1ull);
      continue;
    }
    // the slot's input column is picked by a (scalar) branch per column: a runtime index
    // would put the values in scratch memory, a select chain costs two selects per row
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (v != val) continue;
      apply_slot(op, base, off, b.val[v], b.vok[v], b.vnull[v], p.vals[v].is_float != 0);
    }
  }
  // 5. rows the LDS table could not take
  if (__ballot(to_hbm) != 0ull) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (!(lane_bit(b.live[u]) && e[u] < 0)) continue;
      RowValsT<NV> rv;
      rv.valid = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        rv.v[v] = b.val[v][u];
        rv.valid |= (lane_bit(b.vok[v][u]) ? 1u : 0u) << v;
      }
      global_update(p, (uint64_t)b.key[u], rv);
    }
  }
}

// Ungrouped aggregates (no key columns) with few slots: there is one group, so every lane
// keeps its own accumulators in registers and the LDS table is not touched at all; the
// lanes' accumulators are combined once, when the workgroup is done.
constexpr int kUngroupedSlots = 8;

// One slot of the ungrouped path.  The slot number is a template argument (and the
// caller expands all of them) so that every accumulator stays a named register: left as a
// loop over the slots, the compiler keeps the loop and indexes the accumulators at run time.
template <int S, int NV, bool K32>
__device__ __forceinline__ void consume_ungrouped_slot(const AggParams& p, const PackedRows<0, NV, K32>& b,
                                                       long long (&acc)[kUngroupedSlots]) {
  if (S >= p.n_slots) return;
  const int op = p.slots[S].op, val = p.slots[S].val;
  if (op == kSlotAddOne) {
#pragma unroll
    for (int u = 0; u < kU; ++u) acc[S] += lane_bit(b.live[u]) ? 1 : 0;
    return;
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    if (v != val) continue;
    const bool is_float = p.vals[v].is_float != 0;
    if (op == kSlotAddValid) {
#pragma unroll
      for (int u = 0; u < kU; ++u) acc[S] += lane_bit(b.live[u] & b.vok[v][u]) ? 1 : 0;
    } else if (op == kSlotAddI64) {
#pragma unroll
      for (int u = 0; u < kU; ++u)
        acc[S] = (long long)((unsigned long long)acc[S] + (unsigned long long)(lane_bit(b.live[u] & b.vok[v][u]) ? b.val[v][u] : 0ll));
    } else if (op == kSlotMax) {
#pragma unroll
      for (int u = 0; u < kU; ++u) acc[S] = lane_bit(b.live[u] & b.vok[v][u]) && b.val[v][u] > acc[S] ? b.val[v][u] : acc[S];
    } else if (op == kSlotMin) {
#pragma unroll
      for (int u = 0; u < kU; ++u) acc[S] = lane_bit(b.live[u] & b.vok[v][u]) && b.val[v][u] < acc[S] ? b.val[v][u] : acc[S];
    } else {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const long long x = b.val[v][u];
        const double add = lane_bit(b.live[u] & b.vok[v][u]) ? (is_float ? __longlong_as_double(x) : (double)x) : 0.0;
        acc[S] = __double_as_longlong(__longlong_as_double(acc[S]) + add);
      }
    }
  }
}

template <int NV, bool K32, int... S>
__device__ __forceinline__ void consume_ungrouped_slots(const AggParams& p, const PackedRows<0, NV, K32>& b,
                                                        long long (&acc)[kUngroupedSlots], std::integer_sequence<int, S...>) {
  (consume_ungrouped_slot<S, NV, K32>(p, b, acc), ...);
}

template <int NV, bool K32>
__device__ __forceinline__ void consume_ungrouped(const AggParams& p, const PackedRows<0, NV, K32>& b,
                                                  long long (&acc)[kUngroupedSlots]) {
  consume_ungrouped_slots<NV, K32>(p, b, acc, std::make_integer_sequence<int, kUngroupedSlots>{});
}

// Two partial values of one slot into one.
__device__ __forceinline__ long long merge_partials(int op, long long a, long long b) {
  if (op == kSlotMin) return a < b ? a : b;
  if (op == kSlotMax) return a > b ? a : b;
  if (op == kSlotAddF64) return __double_as_longlong(__longlong_as_double(a) + __longlong_as_double(b));
  return (long long)((unsigned long long)a + (unsigned long long)b);
}

constexpr int kMaxWavesPerBlock = 16;

template <int S>
__device__ __forceinline__ void finish_ungrouped_slot(const AggParams& p, const long long (&acc)[kUngroupedSlots],
                                                      long long (&wave_part)[kUngroupedSlots][kMaxWavesPerBlock]) {
  if (S >= p.n_slots) return;
  const int op = p.slots[S].op;
  long long v = acc[S];
  if (op == kSlotMin) v = wave_min_i64(v);
  else if (op == kSlotMax) v = wave_max_i64(v);
  else if (op == kSlotAddF64) v = __double_as_longlong(wave_sum_f64(__longlong_as_double(v)));
  else v = (long long)wave_sum_u64((unsigned long long)v);
  if ((threadIdx.x & 63u) == 0) wave_part[S][threadIdx.x >> 6] = v;
}

template <int... S>
__device__ __forceinline__ void finish_ungrouped_slots(const AggParams& p, const long long (&acc)[kUngroupedSlots],
                                                       long long (&wave_part)[kUngroupedSlots][kMaxWavesPerBlock],
                                                       std::integer_sequence<int, S...>) {
  (finish_ungrouped_slot<S>(p, acc, wave_part), ...);
}

// The lanes' accumulators become one value per wave, the waves' values one per workgroup (in
// LDS), and the workgroup applies that to the result with one atomic per slot: every one of
// these atomics meets all the others on one address, where they are carried out one after
// the other (a per-wave update was 8 times as many of them: with three slots, tens of
// thousands that kept the kernel alive long after its last load).  Block-wide.
__device__ __forceinline__ void finish_ungrouped(const AggParams& p, const long long (&acc)[kUngroupedSlots]) {
  __shared__ long long wave_part[kUngroupedSlots][kMaxWavesPerBlock];
  finish_ungrouped_slots(p, acc, wave_part, std::make_integer_sequence<int, kUngroupedSlots>{});
  __syncthreads();
  const int s = (int)threadIdx.x;
  if (s >= p.n_slots || s >= kUngroupedSlots) return;
  const int op = p.slots[s].op, n_waves = (int)(blockDim.x >> 6);
  long long v = slot_identity(op);
  for (int w = 0; w < n_waves; ++w) v = merge_partials(op, v, wave_part[s][w]);
  if (v != slot_identity(op))
    atomic_apply(&p.g_state[(uint64_t)s * (p.capacity + 1)], (op == kSlotAddOne || op == kSlotAddValid) ? kSlotAddI64 : op, v);
}

template <bool K32>
__device__ __forceinline__ void lds_clear(const AggParams& p, const Lds<K32>& t) {
  for (int i = threadIdx.x; i < t.stride; i += blockDim.x) {
    if (i < t.S) t.keys[i] = (KeyT<K32>)~(KeyT<K32>)0;
    for (int s = 0; s < p.n_slots; ++s) t.state[s * t.stride + i] = slot_identity(p.slots[s].op);
  }
}

// Folds the LDS table into the HBM table and empties it.  Block-wide; the caller
// brackets it with barriers.
template <bool K32>
__device__ __forceinline__ void lds_flush(const AggParams& p, const Lds<K32>& t) {
  const KeyT<K32> empty = (KeyT<K32>)~(KeyT<K32>)0;
  if (t.direct_rep_log2 >= 0) {
    // Directly indexed: the replicas of a key are merged here, in LDS, so that HBM takes one
    // atomic per key and slot instead of one per replica (a global atomic is a fabric
    // transaction; with 4 replicas they were 4 % of the headline kernel).
    const uint32_t n_keys = 1u << t.direct_key_bits, kmask = n_keys - 1u, reps = 1u << t.direct_rep_log2;
    for (uint32_t k = threadIdx.x; k < n_keys; k += blockDim.x) {
      uint64_t e = ~0ull;
      for (int s = 0; s < p.n_slots; ++s) {
        const int op = p.slots[s].op;
        const long long identity = slot_identity(op);
        long long v = identity;
        for (uint32_t r = 0; r < reps; ++r) {
          const uint32_t i = (r << t.direct_key_bits) | ((k + r * 5u) & kmask);
          v = merge_partials(op, v, t.state[s * t.stride + i]);
          t.state[s * t.stride + i] = identity;
        }
        if (s == 0) {
          if (v == 0) break;  // slot 0 counts the rows: none means the other slots are untouched too
          e = global_entry(p, (uint64_t)(t.win_base + k));
        }
        if (e != ~0ull && v != identity) atomic_apply(&p.g_state[(uint64_t)s * (p.capacity + 1) + e], op, v);
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < t.S; i += blockDim.x) {
    const KeyT<K32> key = t.keys[i];
    if (key == empty) continue;
    uint64_t e = global_entry(p, (uint64_t)key);
    for (int s = 0; s < p.n_slots; ++s) {
      int op = p.slots[s].op;
      long long v = t.state[s * t.stride + i];
      if (e != ~0ull && v != slot_identity(op)) atomic_apply(&p.g_state[(uint64_t)s * (p.capacity + 1) + e], op, v);
      t.state[s * t.stride + i] = slot_identity(op);
    }
    t.keys[i] = empty;
  }
}

#ifdef PRISM_STREAM_WAVES_PER_EU
#define PRISM_STREAM_OCCUPANCY __attribute__((amdgpu_waves_per_eu(PRISM_STREAM_WAVES_PER_EU, PRISM_STREAM_WAVES_PER_EU)))
#else
#define PRISM_STREAM_OCCUPANCY
#endif

// Two 512-thread workgroups share a CU only when a wave needs at most 128 registers.  The
// variants that come out a little above that are told to fit; the wider ones would spill.
constexpr int stream_min_waves(int nk, int nv, bool k32) {
#ifdef PRISM_STREAM_FIT_ALL
  return PRISM_STREAM_FIT_ALL;
#else
  return (nv <= 2 && (k32 || nk + nv <= 2)) ? 4 : 1;
#endif
}

template <int NK, int NV, bool K32, bool PRED>
__global__ __launch_bounds__(kStreamBlock, stream_min_waves(NK, NV, K32)) PRISM_STREAM_OCCUPANCY void k_group_by_stream(AggParams p) {
  extern __shared__ __align__(16) unsigned long long lds_raw[];
  __shared__ int n_entries;
  Lds<K32> t;
  t.S = p.lds_slots;
  t.stride = t.S + kScratchEntries;
  t.bucket_shift = 32 - (31 - __builtin_clz((unsigned)(t.S / kWays)));
  t.state = (long long*)lds_raw;
  t.keys = (KeyT<K32>*)(lds_raw + (size_t)p.n_slots * t.stride);
  t.replicas = kInitialReplicas;
  t.salt = (threadIdx.x & (uint32_t)(t.replicas - 1)) * kReplicaSalt;
  {
    const int log2_s = 31 - __builtin_clz((unsigned)t.S);
    int rep = log2_s - p.key_bits;  // replicas that fit
    if (rep > 4) rep = 4;
    t.direct_rep_log2 = (K32 && NK > 0 && rep >= 2) ? rep : -1;
    t.direct_key_bits = p.key_bits;
    t.win_base = 0;
    t.windowed = false;
    if (K32 && NK > 0 && t.direct_rep_log2 < 0 && p.window_hint && log2_s >= kWindowRepLog2 + 4) {
      t.windowed = true;
      t.direct_rep_log2 = kWindowRepLog2;
      t.direct_key_bits = log2_s - kWindowRepLog2;
    }
  }
  __shared__ WindowShared win;
  __shared__ uint32_t stop_flag;
  if (threadIdx.x == 0) { n_entries = 0; win.misses = 0; win.candidate = 0; stop_flag = 0; }
  lds_clear<K32>(p, t);
  __syncthreads();

  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  // The division runs on the vector ALU (there is no scalar one); without the
  // readfirstlane everything derived from it, the segment cursor included, would be
  // kept in vector registers and every uniform branch below would become a masked one.
  const uint64_t per = uniform_u64(ceil_div(p.sd.n_tiles, (uint64_t)gridDim.x));
  const uint64_t t0 = per * blockIdx.x, t1 = t0 + per < p.sd.n_tiles ? t0 + per : p.sd.n_tiles;
  const uint64_t j0 = t0 * kStepsPerTile, j1 = t1 * kStepsPerTile;
  uint64_t rows_since_flush = 0;
  uint64_t n_selected = 0;  // rows that passed the in-kernel predicate (wave-uniform)
  FusedRegs fused;
  if constexpr (PRED) load_fused_regs<K32>(p, fused);
  bool insert = true;  // false once the table is frozen

  SegCursor<NK, NV> cur;
  RowBatch<NK, NV, K32> raw;
  PackedRows<NK, NV, K32> rows;
  if (j0 < j1) issue<NK, NV, K32>(p, j0, wave, lane, cur, raw);
  if constexpr (NK == 0) {
    if (p.n_slots <= kUngroupedSlots) {
      long long acc[kUngroupedSlots];
#pragma unroll
      for (int s = 0; s < kUngroupedSlots; ++s) acc[s] = s < p.n_slots ? slot_identity(p.slots[s].op) : 0;
      for (uint64_t j = j0; j < j1; ++j) {
        convert<NK, NV, K32>(p, raw, rows);
        if constexpr (PRED) n_selected += fused_predicate<NK, NV, K32>(fused, raw, rows);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < j1) issue<NK, NV, K32>(p, j + 1, wave, lane, cur, raw);
        __builtin_amdgcn_sched_barrier(0);
        consume_ungrouped<NV, K32>(p, rows, acc);
      }
      finish_ungrouped(p, acc);
      if (PRED && lane == 0 && n_selected)
        atomicAdd(p.selected_out + (p.selected_sharded ? (size_t)(blockIdx.x & (kCountShards - 1)) * kCountShardStride : 0), (unsigned long long)n_selected);
      return;
    }
  }
  int window_moves = 0;
  uint64_t tiles_since_move = 0;
  // A window starts at the first key seen, rounded down to a whole time bucket (not to a
  // multiple of the window size: time-ordered rows only move up from there).
  const uint32_t win_align = p.window_hint > 0 ? ~((1u << (p.window_hint - 1)) - 1u) : ~0u;
  for (uint64_t j = j0; j < j1; ++j) {
    convert<NK, NV, K32>(p, raw, rows);
    if constexpr (PRED) n_selected += fused_predicate<NK, NV, K32>(fused, raw, rows);
    if (t.windowed && j == j0) {
      // the first window starts at the workgroup's first key
      if (threadIdx.x == 0) win.candidate = (uint32_t)rows.key[0];
      __syncthreads();
      t.win_base = win.candidate & win_align;
      __syncthreads();
    }
    // The scheduling barriers keep the next step's loads from being hoisted above the
    // conversion (which would make the conversion wait for them too) and from being
    // sunk into the hashing (which would shorten the time they have to land).
    __builtin_amdgcn_sched_barrier(0);
    if (j + 1 < j1) issue<NK, NV, K32>(p, j + 1, wave, lane, cur, raw);
    __builtin_amdgcn_sched_barrier(0);
    consume<NK, NV, K32>(p, t, insert, &n_entries, &win, lane, rows);
    if (t.windowed && (j + 1) % kStepsPerTile == 0) {
      // Window bookkeeping at the end of a tile.  Many rows outside: fold the table into
      // HBM and move the window to where the rows are now.  Moving again and again means
      // the keys are not narrow at any one time: fall back to the hashed table for good.
      tiles_since_move += 1;
      __syncthreads();
      const uint32_t missed = win.misses, cand = win.candidate;
      __syncthreads();
      if (threadIdx.x == 0) win.misses = 0;
      if (missed > kTileRows / 8) {
        lds_flush<K32>(p, t);
        __syncthreads();
        window_moves = tiles_since_move <= 2 ? window_moves + 1 : 0;
        tiles_since_move = 0;
        if (window_moves >= 3) {
          t.windowed = false;
          t.direct_rep_log2 = -1;
          t.win_base = 0;
          lds_clear<K32>(p, t);
        } else {
          t.win_base = cand & win_align;
        }
        __syncthreads();
      }
    }
    // frozen table = nearly every row goes to the HBM table: if that one turned out too
    // small the pass will be re-run with a larger one, so stop early
    if (!insert && (j + 1) % kStepsPerTile == 0 && table_gave_up(p)) break;
    if (insert && t.direct_rep_log2 < 0 && (j + 1) % kStepsPerTile == 0) {
      // Block-wide bookkeeping at the end of a tile, when the LDS table is half full.
      // Keys that repeat a lot: fold the table into HBM and start over, with half as
      // many replicas per group if there was more than one.  Keys that
      // hardly repeat (high cardinality): freeze the table instead.  It then keeps the
      // groups it has, which are the frequent ones (they show up first), so the hot
      // groups still cost HBM no per-row atomics on one address, and every other row
      // goes straight to HBM.
      rows_since_flush += kTileRows;
      if (p.stop_if_scattered && threadIdx.x == 0) stop_flag = table_gave_up(p) ? 1u : 0u;
      __syncthreads();
      const int n = n_entries;
      const bool stop = p.stop_if_scattered && stop_flag != 0u;
      __syncthreads();
      if (stop) break;  // another workgroup found its rows scattered: the caller will partition them instead
      if (n > t.S / 2) {
        if (t.replicas == 1 && (uint64_t)n * 4 > rows_since_flush) {
          if (p.stop_if_scattered) {
            // Most rows would go to HBM one atomic at a time: say so and leave (the caller
            // runs the partitioned GROUP BY instead; this table is dropped).
            if (threadIdx.x == 0) __hip_atomic_store(p.overflow, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          insert = false;
        } else {
          lds_flush<K32>(p, t);
          // fewer replicas only if the table filled quickly: when it took long, it holds
          // groups that are no longer coming (sorted input), not too many copies
          if (t.replicas > 1 && rows_since_flush < 32 * kTileRows) {
            t.replicas >>= 1;
            t.salt = (threadIdx.x & (uint32_t)(t.replicas - 1)) * kReplicaSalt;
          }
          if (threadIdx.x == 0) n_entries = 0;
          rows_since_flush = 0;
          __syncthreads();
        }
      }
    }
  }
  __syncthreads();
  lds_flush<K32>(p, t);
  if (PRED && lane == 0 && n_selected)
        atomicAdd(p.selected_out + (p.selected_sharded ? (size_t)(blockIdx.x & (kCountShards - 1)) * kCountShardStride : 0), (unsigned long long)n_selected);
}

template <int NK, bool K32, bool PRED>
GroupByKernel pick_nv(int nv) {
  switch (nv) {
    case 0:  // a predicate without value columns reads keys only
      if constexpr (PRED && NK == 0) return nullptr;
      else return k_group_by_stream<NK, 0, K32, PRED>;
    case 1: return k_group_by_stream<NK, 1, K32, PRED>;
    case 2: return k_group_by_stream<NK, 2, K32, PRED>;
    case 3: return k_group_by_stream<NK, 3, K32, PRED>;
    case 4: return k_group_by_stream<NK, 4, K32, PRED>;
    default: return nullptr;
  }
}

template <bool K32, bool PRED>
GroupByKernel pick_nk(int nk, int nv) {
  switch (nk) {
    case 0: return K32 ? pick_nv<0, K32, PRED>(nv) : nullptr;  // ungrouped: one group whose key is 0
    case 1: return pick_nv<1, K32, PRED>(nv);
    case 2: return pick_nv<2, K32, PRED>(nv);
    case 3: return pick_nv<3, K32, PRED>(nv);
    case 4: return pick_nv<4, K32, PRED>(nv);
    default: return nullptr;
  }
}

}  // namespace

// The kernels that evaluate a predicate are instantiated in a translation unit of their
// own (kernels_stream_pred.hip includes this file with PRISM_STREAM_PRED_TU defined), so
// that the two halves compile in parallel.
GroupByKernel pick_stream_kernel_pred(bool k32, int n_keys, int n_vals);

#ifdef PRISM_STREAM_PRED_TU
GroupByKernel pick_stream_kernel_pred(bool k32, int n_keys, int n_vals) {
  return k32 ? pick_nk<true, true>(n_keys, n_vals) : pick_nk<false, true>(n_keys, n_vals);
}
#else
GroupByKernel pick_stream_kernel(const AggParams& p, size_t* lds_bytes, uint32_t* block, int* lds_slots) {
  bool k32 = p.key_bits <= 31;
  for (int k = 0; k < p.n_keys; ++k) k32 = k32 && p.keys[k].fast32 != 0;
  // the largest power-of-two table that fits the workgroup's share of the CU's LDS
  const size_t per_entry = 8 * (size_t)p.n_slots + (k32 ? 4 : 8);
  const size_t budget = (size_t)PRISM_STREAM_LDS_KB * 1024 - (size_t)kScratchEntries * 8 * (size_t)p.n_slots - 64;
  int S = 256;
  while ((size_t)S * 2 * per_entry <= budget && S < 8192) S *= 2;
  *lds_slots = S;
  *block = kStreamBlock;
  *lds_bytes = (size_t)(S + kScratchEntries) * 8 * (size_t)p.n_slots + (size_t)S * (k32 ? 4 : 8);
  if (p.n_pred > 0) return pick_stream_kernel_pred(k32, p.n_keys, p.n_vals);
  return k32 ? pick_nk<true, false>(p.n_keys, p.n_vals) : pick_nk<false, false>(p.n_keys, p.n_vals);
}
#endif

}  // namespace prism
