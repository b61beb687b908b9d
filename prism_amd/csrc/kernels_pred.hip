// Predicate evaluation kernels: each leaf of the WHERE tree becomes a bitmap of
// the rows where it is TRUE (NULL rows are never TRUE, which is all FilterExec
// needs once NOTs have been pushed down to the leaves); AND / OR are word-wise.
//
// Fixed-width leaves stream the column with 16-byte loads per lane, so a wave
// instruction moves 1 KiB; result bits are packed to bytes with cross-lane
// shuffles and stored as bytes of the little-endian bitmap words.
#include <algorithm>
#include <cstdint>
#include <type_traits>

#include "device_utils.hpp"

namespace prism {

namespace {

inline uint32_t leaf_grid(prism_gpu_ctx* ctx, uint64_t n_tiles) {
  uint64_t cap = (uint64_t)ctx->n_cus * 8;
  return (uint32_t)(n_tiles < cap ? (n_tiles ? n_tiles : 1) : cap);
}

// Adds each block's count with one atomic, to the block's shard of the count (kCountShards
// partial sums: see kernels.hpp).
__device__ __forceinline__ void block_count_add(unsigned long long local, unsigned long long* count) {
  __shared__ unsigned long long partial[kBlock / 64];
  unsigned long long w = wave_sum_u64(local);
  if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < (int)(kBlock / 64); ++i) t += partial[i];
    if (t) atomicAdd(count + (size_t)(blockIdx.x & (kCountShards - 1)) * kCountShardStride, t);
  }
}

// Packs the V result bits of every lane into bitmap bytes.  `pid` is the padded
// row id of the lane's first row (a multiple of V).  All lanes must call.
template <int V>
__device__ __forceinline__ void store_bits(uint64_t* bits, uint64_t pid, uint32_t m) {
  uint8_t* bytes = (uint8_t*)bits;
  if (V == 8) {
    bytes[pid >> 3] = (uint8_t)m;
  } else if (V == 4) {
    uint32_t o = __shfl_down(m, 1, 64);
    if ((threadIdx.x & 1) == 0) bytes[pid >> 3] = (uint8_t)(m | (o << 4));
  } else {  // V == 2
    uint32_t o1 = __shfl_down(m, 1, 64), o2 = __shfl_down(m, 2, 64), o3 = __shfl_down(m, 3, 64);
    if ((threadIdx.x & 3) == 0) bytes[pid >> 3] = (uint8_t)(m | (o1 << 2) | (o2 << 4) | (o3 << 6));
  }
}

template <typename T>
struct Vec16 {
  static constexpr int V = 16 / sizeof(T);
  T e[V];
};

// 16 bytes that are read once: a nontemporal load (a streaming read that does not ask the
// caches to keep its lines is about 10 % faster on this part).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 stream16(const void* p) {
  const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}

template <typename T>
__device__ __forceinline__ Vec16<T> load16(const T* p) {
  uint4 raw = stream16(p);
  Vec16<T> v;
  __builtin_memcpy(&v, &raw, 16);
  return v;
}

// One tile of a fixed-width leaf.  FULL: the tile lies wholly inside the segment,
// so every load is unconditional and the compiler can batch them.
template <typename T, bool FULL, typename Pred>
__device__ __forceinline__ unsigned long long leaf_tile(const T* vals, const uint64_t* validity, uint64_t seg_rows,
                                                        uint64_t tile_row0, uint64_t tile, Pred pred, uint64_t* bits) {
  constexpr int V = 16 / sizeof(T);
  constexpr int kRowsPerIter = kBlock * V;
  constexpr int kIters = kTileRows / kRowsPerIter;
  unsigned long long local = 0;
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    uint64_t in_tile = (uint64_t)it * kRowsPerIter + (uint64_t)threadIdx.x * V;
    uint64_t r = tile_row0 + in_tile;
    uint32_t m = 0;
    if (FULL || r < seg_rows) {
      Vec16<T> v = load16<T>(vals + r);
#pragma unroll
      for (int j = 0; j < V; ++j) m |= (pred(v.e[j]) ? 1u : 0u) << j;
      if (!FULL) {
        uint64_t left = seg_rows - r;
        if (left < (uint64_t)V) m &= (1u << left) - 1u;
      }
      if (validity) m &= (uint32_t)((validity[r >> 6] >> (r & 63)) & ((1u << V) - 1u));
    }
    local += __popc(m);
    if (bits) store_bits<V>(bits, tile * kTileRows + in_tile, m);
  }
  return local;
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// Every workgroup walks one contiguous share of the tiles, in batches of kLeafLoads
// 16-byte loads per lane (4 / 2 / 1 tiles for 2- / 4- / 8-byte values).  The segment a
// tile belongs to is tracked in scalar registers and looked up again only when a segment
// ends: a binary search over the segment list per tile is a chain of dependent loads that
// costs more than streaming a tile of a narrow column.  A batch that lies wholly inside
// one segment issues all its loads before it touches any of them.
constexpr int kLeafLoads = 8;

template <typename T, typename Pred>
__device__ __forceinline__ void leaf_body(const ScanDesc& sd, const ColPtr* col, Pred pred, uint64_t* bits,
                                          unsigned long long* count) {
  constexpr int V = 16 / sizeof(T);
  constexpr int kRowsPerIter = kBlock * V;
  constexpr int kItersPerTile = kTileRows / kRowsPerIter;
  constexpr uint64_t kTilesPerBatch = kLeafLoads / kItersPerTile;
  static_assert(kLeafLoads % kItersPerTile == 0, "a batch is a whole number of tiles");
  unsigned long long local = 0;
  const uint64_t n_batches = ceil_div(sd.n_tiles, kTilesPerBatch);
  const uint64_t per = uniform_u64(ceil_div(n_batches, (uint64_t)gridDim.x));
  const uint64_t b0 = per * blockIdx.x, b1 = b0 + per < n_batches ? b0 + per : n_batches;
  uint32_t seg = 0xffffffffu;
  uint64_t seg_rows = 0, seg_tile_base = 0, next_tile_base = 0;  // next_tile_base = 0 forces the first lookup
  const T* vals = nullptr;
  const uint64_t* validity = nullptr;
  for (uint64_t b = b0; b < b1; ++b) {
    const uint64_t first = b * kTilesPerBatch;
    const uint64_t last = first + kTilesPerBatch < sd.n_tiles ? first + kTilesPerBatch : sd.n_tiles;
    for (uint64_t tile = first; tile < last;) {
      if (tile >= next_tile_base) {
        uint32_t s = seg == 0xffffffffu ? find_seg(sd, tile) : seg + 1;
        while (s + 1 < sd.n_segs && sd.segs[s + 1].tile_base <= tile) ++s;
        seg = __builtin_amdgcn_readfirstlane(s);
        seg_rows = uniform_u64(sd.segs[seg].rows);
        seg_tile_base = uniform_u64(sd.segs[seg].tile_base);
        next_tile_base = seg + 1 < sd.n_segs ? uniform_u64(sd.segs[seg + 1].tile_base) : ~0ull;
        vals = (const T*)uniform_u64((uint64_t)col[seg].values);
        validity = (const uint64_t*)uniform_u64((uint64_t)col[seg].validity);
      }
      const uint64_t tile_row0 = (tile - seg_tile_base) * kTileRows;
      const bool whole_batch = tile == first && last - first == kTilesPerBatch && last <= next_tile_base &&
                               tile_row0 + kTilesPerBatch * kTileRows <= seg_rows;
      if (whole_batch) {
        Vec16<T> v[kLeafLoads];
#pragma unroll
        for (int it = 0; it < kLeafLoads; ++it) v[it] = load16<T>(vals + tile_row0 + (uint64_t)it * kRowsPerIter + threadIdx.x * V);
#pragma unroll
        for (int it = 0; it < kLeafLoads; ++it) {
          const uint64_t in_batch = (uint64_t)it * kRowsPerIter + (uint64_t)threadIdx.x * V;
          uint32_t m = 0;
#pragma unroll
          for (int j = 0; j < V; ++j) m |= (pred(v[it].e[j]) ? 1u : 0u) << j;
          if (validity) {
            const uint64_t r = tile_row0 + in_batch;
            m &= (uint32_t)((validity[r >> 6] >> (r & 63)) & ((1u << V) - 1u));
          }
          local += __popc(m);
          if (bits) store_bits<V>(bits, tile * kTileRows + in_batch, m);
        }
        tile = last;
      } else {
        if (tile_row0 + kTileRows <= seg_rows)
          local += leaf_tile<T, true>(vals, validity, seg_rows, tile_row0, tile, pred, bits);
        else
          local += leaf_tile<T, false>(vals, validity, seg_rows, tile_row0, tile, pred, bits);
        ++tile;
      }
    }
  }
  if (count) block_count_add(local, count);
}

// A comparison with a literal (and optionally a second one on the same column, AND-ed in: a
// range in one pass) arrives as range tests worked out on the host: the value passes test i
// when ((u - lo_i) <= span_i) != neg_i, with u the value in the unsigned type U of its
// promoted width.  Two instructions per value and test, in 32 bits for columns of up to 4
// bytes; comparing in 64 bits against three flags per ordering cost ten times that and made
// a 2-byte column compute bound.
template <typename T, bool TWO>
__global__ __launch_bounds__(kBlock) void k_cmp_leaf(ScanDesc sd, const ColPtr* col, unsigned long long lo1, unsigned long long span1,
                                                    int neg1, unsigned long long lo2, unsigned long long span2, int neg2,
                                                    uint64_t* bits, unsigned long long* count) {
  using U = typename std::conditional<sizeof(T) == 8, unsigned long long, uint32_t>::type;
  using P = typename std::conditional<sizeof(T) == 8, long long,
                                      typename std::conditional<std::is_signed<T>::value, int32_t, uint32_t>::type>::type;
  const U l1 = (U)lo1, s1 = (U)span1, l2 = (U)lo2, s2 = (U)span2;
  const bool n1 = neg1 != 0, n2 = neg2 != 0;
  leaf_body<T>(
      sd, col,
      [=](T x) {
        const U u = (U)(P)x;
        bool pass = ((U)(u - l1) <= s1) != n1;
        if (TWO) pass = pass && (((U)(u - l2) <= s2) != n2);
        return pass;
      },
      bits, count);
}

__global__ __launch_bounds__(kBlock) void k_lut_leaf(ScanDesc sd, const ColPtr* col, const uint8_t* lut,
                                                    uint32_t lut_len, uint64_t* bits) {
  leaf_body<uint32_t>(
      sd, col, [=](uint32_t code) { return code < lut_len && lut[code] != 0; }, bits, nullptr);
}

// IS [NOT] NULL, and validity masking of a bitmap: word granular, 64 words per tile.
// mode 0: bits = NULL rows; 1: bits = non-NULL rows; 2: bits &= valid; 3: bits = valid & ~bits.
__global__ __launch_bounds__(kBlock) void k_validity(ScanDesc sd, const ColPtr* col, int mode, uint64_t* bits) {
  uint64_t n_words = sd.n_tiles * kTileWords;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t tile = w / kTileWords;
    uint32_t s = find_seg(sd, tile);
    uint64_t seg_rows = sd.segs[s].rows;
    uint64_t lw = w - sd.segs[s].tile_base * kTileWords;  // word index inside the segment
    uint64_t first = lw * 64;
    uint64_t exist = first >= seg_rows ? 0ull : (seg_rows - first >= 64 ? ~0ull : ((1ull << (seg_rows - first)) - 1ull));
    uint64_t valid = exist;
    if (col[s].validity && exist) valid &= col[s].validity[lw];
    uint64_t out;
    if (mode == 0) out = exist & ~valid;
    else if (mode == 1) out = valid;
    else if (mode == 2) out = bits[w] & valid;
    else out = valid & ~bits[w];
    bits[w] = out;
  }
}

// ---- raw string predicates, one row per lane ---------------------------------

__device__ __forceinline__ uint8_t fold_ascii(uint8_t c, bool ci) {
  return (ci && c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c;
}

__device__ __forceinline__ uint32_t utf8_step(const uint8_t* s, uint32_t i, uint32_t n) {
  uint8_t c = s[i];
  uint32_t l = c < 0x80 ? 1u : (c >> 5) == 6 ? 2u : (c >> 4) == 14 ? 3u : (c >> 3) == 30 ? 4u : 1u;
  return l < n - i ? l : n - i;
}

// Glob-style matcher with one backtrack point: the same algorithm SQL LIKE needs
// for '%' (any run) and '_' (one character).
__device__ bool like_match(const LikeToken* t, uint32_t nt, const uint8_t* s, uint32_t n, bool ci) {
  uint32_t si = 0, ti = 0, star_t = 0xffffffffu, star_s = 0;
  while (si < n) {
    if (ti < nt && t[ti].kind == 0 && fold_ascii(t[ti].byte, ci) == fold_ascii(s[si], ci)) { ++si; ++ti; }
    else if (ti < nt && t[ti].kind == 1) { si += utf8_step(s, si, n); ++ti; }
    else if (ti < nt && t[ti].kind == 2) { star_t = ti++; star_s = si; }
    else if (star_t != 0xffffffffu) { ti = star_t + 1; star_s += utf8_step(s, star_s, n); si = star_s; }
    else return false;
  }
  while (ti < nt && t[ti].kind == 2) ++ti;
  return ti == nt;
}

__global__ __launch_bounds__(kBlock) void k_str_leaf(ScanDesc sd, const ColPtr* col, int kind, int cmp_mask,
                                                    const uint8_t* lit, uint32_t lit_len, const LikeToken* tokens,
                                                    uint32_t n_tokens, int negated, int ci, uint64_t* bits) {
  for (uint64_t tile = blockIdx.x; tile < sd.n_tiles; tile += gridDim.x) {
    uint32_t s = find_seg(sd, tile);
    uint64_t seg_rows = sd.segs[s].rows;
    uint64_t tile_row0 = (tile - sd.segs[s].tile_base) * kTileRows;
    const int32_t* off = (const int32_t*)col[s].values;
    const uint8_t* bytes = col[s].bytes;
    const uint64_t* validity = col[s].validity;
    for (uint32_t it = 0; it < kTileRows / kBlock; ++it) {
      uint64_t r = tile_row0 + (uint64_t)it * kBlock + threadIdx.x;
      bool t = false;
      if (r < seg_rows && valid_at(validity, r)) {
        uint32_t b = (uint32_t)off[r], e = (uint32_t)off[r + 1];
        const uint8_t* p = bytes + b;
        uint32_t n = e - b;
        if (kind == kStrCmp) {
          uint32_t m = n < lit_len ? n : lit_len;
          int c = 0;
          for (uint32_t i = 0; i < m && c == 0; ++i) c = (int)p[i] - (int)lit[i];
          if (c == 0) c = n < lit_len ? -1 : (n > lit_len ? 1 : 0);
          t = (c < 0 && (cmp_mask & kLt)) || (c == 0 && (cmp_mask & kEq)) || (c > 0 && (cmp_mask & kGt));
        } else {
          t = like_match(tokens, n_tokens, p, n, ci != 0);
          if (negated) t = !t;
        }
      }
      unsigned long long word = __ballot(t);
      if ((threadIdx.x & 63) == 0) bits[tile * kTileWords + it * (kBlock / 64) + (threadIdx.x >> 6)] = word;
    }
  }
}

// ---- LIKE '%needle%' as a stream search -----------------------------------------
//
// Every lane owns 16 consecutive bytes of the segment's string buffer and tests the
// 16 start positions in it against the first (up to four) needle bytes with
// byte-aligned dword extracts; survivors are verified byte by byte, located in the
// offsets array by binary search, and recorded with an atomic OR.  The buffer is
// read once, fully coalesced, whatever the string lengths are.
__global__ __launch_bounds__(kBlock) void k_contains_stream(const int32_t* off, const uint8_t* bytes, uint64_t rows,
                                                           uint64_t total_bytes, uint64_t tile_base,
                                                           const uint8_t* needle, uint32_t L, uint64_t* bits) {
  const uint32_t k = L < 4 ? L : 4;
  uint32_t first = 0;
  for (uint32_t i = 0; i < k; ++i) first |= (uint32_t)needle[i] << (8 * i);
  const uint32_t nmask = k == 4 ? 0xffffffffu : ((1u << (8 * k)) - 1u);
  const uint64_t n_chunks = (total_bytes + kBlock * 16 - 1) / (kBlock * 16);
  for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    uint64_t p = chunk * (kBlock * 16) + (uint64_t)threadIdx.x * 16;
    if (p >= total_bytes) continue;
    // the buffer is padded by the loader, so these 20 bytes are always readable
    uint4 own = stream16(bytes + p);
    uint32_t next = *(const uint32_t*)(bytes + p + 16);
    uint32_t d[5] = {own.x, own.y, own.z, own.w, next};
    uint32_t cand = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      uint32_t w = __builtin_amdgcn_alignbyte(d[(j >> 2) + 1], d[j >> 2], j & 3);
      cand |= (((w & nmask) == first) ? 1u : 0u) << j;
    }
    while (cand) {
      int j = __ffs(cand) - 1;
      cand &= cand - 1;
      uint64_t pos = p + j;
      if (pos + L > total_bytes) continue;
      bool ok = true;
      for (uint32_t i = 4; i < L && ok; ++i) ok = bytes[pos + i] == needle[i];
      if (!ok) continue;
      // row with off[row] <= pos < off[row + 1]
      uint64_t lo = 0, hi = rows;
      while (hi - lo > 1) {
        uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)(uint32_t)off[mid] <= pos) lo = mid; else hi = mid;
      }
      if (pos + L <= (uint64_t)(uint32_t)off[lo + 1] && (uint64_t)(uint32_t)off[lo] <= pos) {
        uint64_t pid = tile_base * kTileRows + lo;
        atomicOr((unsigned long long*)&bits[pid >> 6], 1ull << (pid & 63));
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_bitmap_binary(const uint64_t* a, const uint64_t* b, uint64_t* out,
                                                         uint64_t n, int is_or) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = is_or ? (a[i] | b[i]) : (a[i] & b[i]);
}

__global__ __launch_bounds__(kBlock) void k_bitmap_count(const uint64_t* bits, uint64_t n, unsigned long long* count) {
  unsigned long long local = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    local += __popcll(bits[i]);
  block_count_add(local, count);
}

}  // namespace

namespace {

// The values of a column of type [tmin, tmax] that satisfy `v <cmp> literal`, as one range
// test: pass iff ((v - lo) <= span) != neg, in unsigned arithmetic of the column's promoted width.
struct RangeTest {
  unsigned long long lo = 0, span = ~0ull;
  int neg = 0;
};

RangeTest range_test(int cmp_mask, int64_t lit, int64_t tmin, int64_t tmax) {
  const bool lt = cmp_mask & kLt, eq = cmp_mask & kEq, gt = cmp_mask & kGt;
  RangeTest never;
  never.neg = 1;  // NOT (everything)
  RangeTest all;
  if (lt && eq && gt) return all;
  if (!lt && !eq && !gt) return never;
  bool negate = false;
  int64_t a, b;  // inclusive
  if (lt && gt) { negate = true; a = b = lit; }           // <>
  else if (lt) { a = INT64_MIN; if (eq) b = lit; else { if (lit == INT64_MIN) return never; b = lit - 1; } }
  else if (gt) { b = INT64_MAX; if (eq) a = lit; else { if (lit == INT64_MAX) return never; a = lit + 1; } }
  else { a = b = lit; }
  a = std::max(a, tmin);
  b = std::min(b, tmax);
  if (a > b) return negate ? all : never;  // no value of the type lies in the range
  RangeTest r;
  r.lo = (unsigned long long)a;
  r.span = (unsigned long long)b - (unsigned long long)a;
  r.neg = negate ? 1 : 0;
  return r;
}

template <typename T>
void launch_cmp_leaf_t(prism_gpu_ctx* ctx, const ScanDesc& sd, const ColPtr* col, const RangeTest& r1, const RangeTest& r2, bool two,
                       uint64_t* bits, unsigned long long* count) {
  dim3 grid(leaf_grid(ctx, sd.n_tiles)), block(kBlock);
  if (two) k_cmp_leaf<T, true><<<grid, block, 0, ctx->stream>>>(sd, col, r1.lo, r1.span, r1.neg, r2.lo, r2.span, r2.neg, bits, count);
  else k_cmp_leaf<T, false><<<grid, block, 0, ctx->stream>>>(sd, col, r1.lo, r1.span, r1.neg, r2.lo, r2.span, r2.neg, bits, count);
}

}  // namespace

void launch_cmp_leaf(prism_gpu_ctx* ctx, const ScanDesc& sd, const ColPtr* col, int width, bool is_unsigned,
                     int cmp_mask, int64_t literal, uint64_t* bits, unsigned long long* count, int cmp_mask2, int64_t literal2) {
  if (sd.n_tiles == 0) return;
  int64_t tmin, tmax;
  if (width == 2) { tmin = is_unsigned ? 0 : INT16_MIN; tmax = is_unsigned ? UINT16_MAX : INT16_MAX; }
  else if (width == 4) { tmin = is_unsigned ? 0 : INT32_MIN; tmax = is_unsigned ? (int64_t)UINT32_MAX : INT32_MAX; }
  else if (width == 8 && !is_unsigned) { tmin = INT64_MIN; tmax = INT64_MAX; }
  else fail(PRISM_GPU_EUNSUPPORTED, "comparison on this column type is not supported");
  const RangeTest r1 = range_test(cmp_mask, literal, tmin, tmax), r2 = range_test(cmp_mask2, literal2, tmin, tmax);
  const bool two = (cmp_mask2 & (kLt | kEq | kGt)) != (kLt | kEq | kGt);
  if (width == 2 && !is_unsigned) launch_cmp_leaf_t<int16_t>(ctx, sd, col, r1, r2, two, bits, count);
  else if (width == 2) launch_cmp_leaf_t<uint16_t>(ctx, sd, col, r1, r2, two, bits, count);
  else if (width == 4 && !is_unsigned) launch_cmp_leaf_t<int32_t>(ctx, sd, col, r1, r2, two, bits, count);
  else if (width == 4) launch_cmp_leaf_t<uint32_t>(ctx, sd, col, r1, r2, two, bits, count);
  else launch_cmp_leaf_t<int64_t>(ctx, sd, col, r1, r2, two, bits, count);
  PRISM_HIP(hipGetLastError());
}

void launch_lut_leaf(prism_gpu_ctx* ctx, const ScanDesc& sd, const ColPtr* col, const uint8_t* lut, uint32_t lut_len,
                     uint64_t* bits) {
  if (sd.n_tiles == 0) return;
  k_lut_leaf<<<dim3(leaf_grid(ctx, sd.n_tiles)), dim3(kBlock), 0, ctx->stream>>>(sd, col, lut, lut_len, bits);
  PRISM_HIP(hipGetLastError());
}

void launch_null_leaf(prism_gpu_ctx* ctx, const ScanDesc& sd, const ColPtr* col, bool want_null, uint64_t* bits) {
  if (sd.n_tiles == 0) return;
  uint64_t blocks = ceil_div(sd.n_tiles * kTileWords, kBlock);
  uint32_t grid = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)ctx->n_cus * 8);
  k_validity<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(sd, col, want_null ? 0 : 1, bits);
  PRISM_HIP(hipGetLastError());
}

void launch_mask_validity(prism_gpu_ctx* ctx, const ScanDesc& sd, const ColPtr* col, bool negate, uint64_t* bits) {
  if (sd.n_tiles == 0) return;
  uint64_t blocks = ceil_div(sd.n_tiles * kTileWords, kBlock);
  uint32_t grid = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)ctx->n_cus * 8);
  k_validity<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(sd, col, negate ? 3 : 2, bits);
  PRISM_HIP(hipGetLastError());
}

void launch_str_leaf(prism_gpu_ctx* ctx, const ScanDesc& sd, const ColPtr* col, int kind, int cmp_mask,
                     const uint8_t* literal, uint32_t literal_len, const LikeToken* tokens, uint32_t n_tokens,
                     bool negated, bool case_insensitive, uint64_t* bits) {
  if (sd.n_tiles == 0) return;
  k_str_leaf<<<dim3(leaf_grid(ctx, sd.n_tiles)), dim3(kBlock), 0, ctx->stream>>>(
      sd, col, kind, cmp_mask, literal, literal_len, tokens, n_tokens, negated ? 1 : 0, case_insensitive ? 1 : 0, bits);
  PRISM_HIP(hipGetLastError());
}

namespace {

// Pass 1: every workgroup walks a contiguous share of chunks (64 bytes a thread) of the concatenated
// segments; a lane tests the 16 byte positions of each of its four 16-byte pieces.  The
// four bytes a lane needs beyond its own piece come from its neighbour's registers; only
// those after the wave's last lane are loaded.  (Measured and dropped: a second register
// buffer to have the next chunk's loads in flight during the search, 40 % slower.)
struct ContainsChunk {
  uint32_t seg;
  const uint8_t* bytes;
  uint64_t total, base;
};

//
// S = 0: the byte-wise prefilter below (needles of up to 3 bytes).  S = 1, 2, 4 (needles of
// at least S + 3 bytes): only the 4-byte windows at positions that are multiples of S are
// looked at, each compared with the S words needle[i .. i + 4), i < S: an occurrence at s
// shows needle[i ..] at the first multiple of S at or after s, i = that position - s < S, and
// at no other window with i < S, so every occurrence is found exactly once.  That is 4 to 7
// vector instructions per word instead of 10, the comparisons leave wave-wide masks in scalar
// registers (nothing is packed per lane), and a window that matches four needle bytes is
// nearly always an occurrence, so the per-candidate loop hardly ever runs.
template <int S, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_contains_find(const ContainsSeg* segs, uint32_t n_segs, uint64_t n_chunks,
                                                         const uint8_t* needle, uint32_t L, uint64_t* matches,
                                                         uint64_t capacity, unsigned long long* n_matches) {
  constexpr int K = 4;  // 16-byte pieces a thread and chunk
  constexpr uint64_t kChunkBytes = contains_chunk_bytes(BLOCK);
  const uint32_t k = L < 4 ? L : 4;
  uint32_t first = 0;
  for (uint32_t i = 0; i < k; ++i) first |= (uint32_t)needle[i] << (8 * i);
  const uint32_t nmask = k == 4 ? 0xffffffffu : ((1u << (8 * k)) - 1u);
  const uint64_t per = uniform_u64(ceil_div(n_chunks, (uint64_t)gridDim.x));
  const uint64_t c0 = per * blockIdx.x, c1 = c0 + per < n_chunks ? c0 + per : n_chunks;
  uint32_t seg_cursor = 0;
  const int lane = threadIdx.x & 63;
  // Occurrences are collected in LDS and moved to the list a few hundred at a time: one
  // atomic on the list's counter per occurrence would serialise on that one address.
  constexpr uint32_t kBuf = BLOCK >= 256 ? 1024 : 256;
  __shared__ uint64_t buf[kBuf];
  __shared__ uint32_t buf_n;
  __shared__ unsigned long long flush_base;
  if (threadIdx.x == 0) buf_n = 0;
  __syncthreads();
  auto flush = [&]() {  // block-wide, between barriers
    const uint32_t n = buf_n < kBuf ? buf_n : kBuf;
    if (threadIdx.x == 0) flush_base = atomicAdd(n_matches, (unsigned long long)n);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += BLOCK)
      if (flush_base + i < capacity) matches[flush_base + i] = buf[i];
    __syncthreads();
    if (threadIdx.x == 0) buf_n = 0;
    __syncthreads();
  };
  // The segment's description stays in scalar registers until the chunks move on to the next
  // segment: read from memory for every chunk, its loads (each waiting for the one before)
  // stood in front of the chunk's data loads.
  ContainsChunk cur_seg{};
  uint64_t seg_chunk_base = 0, next_chunk_base = 0;  // next_chunk_base = 0 forces the first look-up
  auto locate = [&](uint64_t c) {
    if (c >= next_chunk_base) {
      while (seg_cursor + 1 < n_segs && segs[seg_cursor + 1].chunk_base <= c) ++seg_cursor;
      seg_cursor = __builtin_amdgcn_readfirstlane(seg_cursor);
      cur_seg.seg = seg_cursor;
      cur_seg.bytes = (const uint8_t*)uniform_u64((uint64_t)segs[seg_cursor].bytes);
      cur_seg.total = uniform_u64(segs[seg_cursor].n_bytes);
      seg_chunk_base = uniform_u64(segs[seg_cursor].chunk_base);
      next_chunk_base = seg_cursor + 1 < n_segs ? uniform_u64(segs[seg_cursor + 1].chunk_base) : ~0ull;
    }
    ContainsChunk cc = cur_seg;
    cc.base = (c - seg_chunk_base) * kChunkBytes;
    return cc;
  };
  // d[q][0..3]: the lane's 16 bytes of piece q; d[q][4]: the 4 bytes after the wave's last
  // lane (one address per wave).  Every load is unconditional, on the global address space,
  // at an address clamped into the buffer (pieces beyond the end are discarded by the
  // search): behind a per-lane branch the compiler waited for each piece before loading the
  // next, and through a generic pointer the loads were flat_loads.
  auto issue = [&](const ContainsChunk& cc, uint32_t (&d)[K][5]) {
    typedef const __attribute__((address_space(1))) u32x4_t* gvec_t;
    typedef const __attribute__((address_space(1))) uint32_t* gword_t;
    const uint64_t gb = (uint64_t)cc.bytes;
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const uint64_t p = cc.base + (uint64_t)q * (BLOCK * 16) + (uint64_t)threadIdx.x * 16;
      const uint64_t wave_end = cc.base + (uint64_t)q * (BLOCK * 16) + (uint64_t)((threadIdx.x >> 6) + 1) * 1024;
      const u32x4_t own = __builtin_nontemporal_load((gvec_t)(gb + (p < cc.total ? p : 0)));
      d[q][0] = own.x; d[q][1] = own.y; d[q][2] = own.z; d[q][3] = own.w;
      // the buffer is padded by the loader, so the 4 bytes after the last piece are readable
      d[q][4] = *(gword_t)(gb + (wave_end <= cc.total ? wave_end : 0));
    }
  };
  // Prefilter, four byte positions per instruction: a position is a candidate when its byte
  // equals the needle's first byte and the next byte the needle's second.  "Is some byte of
  // x zero" is (x - 0x01010101) & ~x & 0x80808080 (exact for the lowest zero byte, possibly
  // also set above one: a filter may say yes too often).  About 10 vector instructions per
  // 4 positions, against 5 per position for comparing four needle bytes at every position.
  const uint32_t n0 = (first & 0xffu) * 0x01010101u, n1 = ((first >> 8) & 0xffu) * 0x01010101u;
  const bool two = L >= 2;
  auto search = [&](const ContainsChunk& cc, uint32_t (&d)[K][5]) {
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const uint64_t p = cc.base + (uint64_t)q * (BLOCK * 16) + (uint64_t)threadIdx.x * 16;
      const uint32_t from_next = (uint32_t)__shfl_down((int)d[q][0], 1);
      const uint32_t after = lane == 63 ? d[q][4] : from_next;
      uint32_t z[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t lo = d[q][w], hi = w == 3 ? after : d[q][w + 1];
        const uint32_t x0 = lo ^ n0;
        z[w] = (x0 - 0x01010101u) & ~x0 & 0x80808080u;
        if (two) {
          const uint32_t x1 = __builtin_amdgcn_alignbyte(hi, lo, 1) ^ n1;
          z[w] &= (x1 - 0x01010101u) & ~x1 & 0x80808080u;
        }
      }
      const bool inside = p < cc.total;  // (a piece beyond the end holds what a clamped load brought in)
      if (__ballot(inside && (z[0] | z[1] | z[2] | z[3]) != 0u) == 0ull) continue;
      // bit 7 of byte b of z[w] -> bit 4 w + b
      uint32_t cand = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t y = z[w] >> 7;
        cand |= ((y | (y >> 7) | (y >> 14) | (y >> 21)) & 0xfu) << (4 * w);
      }
      if (!inside) cand = 0;
      // per candidate: the needle's first four bytes against the registers, the rest against memory
      while (__ballot(cand != 0) != 0ull) {
        bool hit = false;
        uint64_t pos = 0;
        if (cand) {
          const int j = __ffs(cand) - 1;
          cand &= cand - 1;
          const int wj = j >> 2;
          const uint32_t lo = wj == 0 ? d[q][0] : (wj == 1 ? d[q][1] : (wj == 2 ? d[q][2] : d[q][3]));
          const uint32_t hi = wj == 0 ? d[q][1] : (wj == 1 ? d[q][2] : (wj == 2 ? d[q][3] : after));
          const uint32_t four = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(j & 3));
          pos = p + j;
          hit = (four & nmask) == first && pos + L <= cc.total;
          for (uint32_t i = 4; i < L && hit; ++i) hit = cc.bytes[pos + i] == needle[i];
        }
        const uint64_t m = __ballot(hit);
        if (m) {
          uint32_t at = 0;
          if (lane == __ffsll((long long)m) - 1) at = atomicAdd(&buf_n, (uint32_t)__popcll(m));
          at = __shfl(at, __ffsll((long long)m) - 1);
          if (hit) {
            at += __popcll(m & ((1ull << lane) - 1));
            const uint64_t rec = ((uint64_t)cc.seg << 32) | pos;  // pos < 2^31: offsets are i32
            if (at < kBuf) {
              buf[at] = rec;
            } else {
              // the buffer filled up within one chunk (a needle that occurs all over): append directly
              const unsigned long long g = atomicAdd(n_matches, 1ull);
              if (g < capacity) matches[g] = rec;
            }
          }
        }
      }
    }
    __syncthreads();
    if (buf_n > kBuf / 2) flush();  // uniform: every thread reads the same counter after the barrier
    else __syncthreads();
  };
  // (A barrier per chunk also keeps the waves of a workgroup in step; looking at the buffer
  // only every fourth chunk, so that waves may run ahead, measured 20 % slower.)
  // ---- S > 0: windows at multiples of S against S needle words ----
  uint32_t gram[4] = {0, 0, 0, 0};
  if (S > 0)
    for (int i = 0; i < S; ++i)
      for (int b = 0; b < 4; ++b) gram[i] |= (uint32_t)needle[i + b] << (8 * b);
  auto needle_word = [&](uint32_t at4) {
    return (uint32_t)needle[at4] | ((uint32_t)needle[at4 + 1] << 8) | ((uint32_t)needle[at4 + 2] << 16) | ((uint32_t)needle[at4 + 3] << 24);
  };
  auto search_grams = [&](const ContainsChunk& cc, uint32_t (&d)[K][5]) {
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const uint64_t p = cc.base + (uint64_t)q * (BLOCK * 16) + (uint64_t)threadIdx.x * 16;
      uint32_t after = 0;
      if (S < 4) {
        const uint32_t from_next = (uint32_t)__shfl_down((int)d[q][0], 1);
        after = lane == 63 ? d[q][4] : from_next;
      }
      // window (w, off): the four bytes at offset 4 w + off of the piece
      auto window = [&](int w, int off) {
        const uint32_t lo = d[q][w], hi = w == 3 ? after : d[q][w + 1];
        return off == 0 ? lo : __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)off);
      };
      bool any = false;
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int off = 0; off < 4; off += (S > 0 ? S : 4)) {
          const uint32_t win = window(w, off);
#pragma unroll
          for (int i = 0; i < S; ++i) any = any || win == gram[i];
        }
      const bool inside = p < cc.total;
      if (__ballot(inside && any) == 0ull) continue;
      // rare from here on.  Candidate bit 4 w + off + i: window (w, off) equals needle word i.
      // (Tried and dropped: settling candidates from the neighbours' registers instead of
      // loading their bytes again -- no faster, and the registers cost a resident workgroup;
      // working out on the scalar side which comparisons hit and settling each for the whole
      // wave at once -- 1.60 ms against 1.28.)
      uint32_t cand = 0;
      if (inside) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int off = 0; off < 4; off += (S > 0 ? S : 4)) {
            const uint32_t win = window(w, off);
#pragma unroll
            for (int i = 0; i < S; ++i) cand |= (win == gram[i] ? 1u : 0u) << (4 * w + off + i);
          }
      }
      while (__ballot(cand != 0) != 0ull) {
        bool hit = false;
        uint64_t pos = 0;
        if (cand) {
          const int j = __ffs(cand) - 1;
          cand &= cand - 1;
          const int i = (j & 3) % (S > 0 ? S : 1), at_byte = j - i;  // window offset in the piece, needle word
          hit = p + at_byte >= (uint64_t)i;
          pos = p + at_byte - i;
          hit = hit && pos + L <= cc.total;
          // the window already showed needle[i .. i + 4): what is left are the i bytes before
          // it and the bytes from i + 4 on, four at a time (unaligned loads; the last one may
          // look again at bytes already compared)
          if (hit) {
            typedef const __attribute__((address_space(1), aligned(1))) uint32_t* uword_t;
            const uint64_t gb = (uint64_t)cc.bytes + pos;
            if (i > 0) hit = *(uword_t)gb == first;
            for (uint32_t k2 = (uint32_t)i + 4; k2 < L && hit; k2 += 4) {
              const uint32_t at4 = k2 + 4 <= L ? k2 : L - 4;
              hit = *(uword_t)(gb + at4) == needle_word(at4);
            }
          }
        }
        const uint64_t m = __ballot(hit);
        if (m) {
          uint32_t at = 0;
          if (lane == __ffsll((long long)m) - 1) at = atomicAdd(&buf_n, (uint32_t)__popcll(m));
          at = __shfl(at, __ffsll((long long)m) - 1);
          if (hit) {
            at += __popcll(m & ((1ull << lane) - 1));
            const uint64_t rec = ((uint64_t)cc.seg << 32) | pos;
            if (at < kBuf) {
              buf[at] = rec;
            } else {
              const unsigned long long g = atomicAdd(n_matches, 1ull);
              if (g < capacity) matches[g] = rec;
            }
          }
        }
      }
    }
    __syncthreads();
    if (buf_n > kBuf / 2) flush();
    else __syncthreads();
  };
  uint32_t da[K][5];
  for (uint64_t c = c0; c < c1; ++c) {
    const ContainsChunk ca = locate(c);
    issue(ca, da);
    if (S > 0) search_grams(ca, da);
    else search(ca, da);
  }
  __syncthreads();
  flush();
}

// Pass 2: one thread per occurrence finds its row by binary search over the offsets.  The
// number of occurrences is read on the device, so the host need not wait for the first pass
// before it launches this one.  What bounds the search is the cache lines it brings in that
// nobody else wants: the upper levels of the search are the same few lines for every
// occurrence of a segment.  (Tried and dropped, all slower, 0.23 to 0.40 ms against 0.156:
// cutting the range in sixteen or in four per step with the pivots' loads in flight together;
// four occurrences per thread searched in step; starting from two probes around the row that
// the occurrence's share of the segment's bytes suggests -- fewer loads or more of them in
// flight, but more lines of their own per occurrence.  Also dropped: no second pass at all,
// the first keeping the offsets of the 512 rows around each chunk in LDS and looking the rows
// of a chunk's occurrences up there while the next chunk's loads are in flight -- the window's
// load and hand-over cost 0.10 ms, the look-ups 0.22 ms (they hold up the workgroup's barrier)
// and the registers one resident workgroup per CU, 0.15 ms: 1.77 ms against 1.46 for both passes.)
__global__ __launch_bounds__(kBlock) void k_contains_mark(const ContainsSeg* segs, const uint64_t* matches,
                                                         const unsigned long long* n_matches, uint64_t capacity,
                                                         uint32_t L, uint64_t* bits) {
  const uint64_t found = *n_matches, n = found < capacity ? found : capacity;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const uint64_t m = matches[i];
    const ContainsSeg& s = segs[m >> 32];
    const uint64_t pos = m & 0xffffffffull;
    // row with off[row] <= pos < off[row + 1].  A plain binary search over a segment's offsets is
    // twenty dependent loads from twenty cache lines; the offsets grow almost linearly, so the
    // search starts where the mean string length puts the row, corrects that guess twice by the
    // byte distance still missing (a secant step: one load each), widens a bracket around it by
    // doubling, and only then bisects -- a handful of loads, the last ones from one line.
    const float inv_mean = (float)s.rows / (float)(s.n_bytes ? s.n_bytes : 1);
    int64_t g = (int64_t)((float)pos * inv_mean);
    const int64_t last = (int64_t)s.rows - 1;
    g = g < 0 ? 0 : (g > last ? last : g);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int64_t d = (int64_t)(((float)pos - (float)(uint32_t)s.off[g]) * inv_mean);
      g += d;
      g = g < 0 ? 0 : (g > last ? last : g);
    }
    uint64_t lo, hi;
    if ((uint64_t)(uint32_t)s.off[g] <= pos) {
      lo = (uint64_t)g;
      uint64_t step = 4;
      hi = lo + step;
      while (hi < s.rows && (uint64_t)(uint32_t)s.off[hi] <= pos) { lo = hi; step <<= 1; hi = lo + step; }
      if (hi > s.rows) hi = s.rows;
    } else {
      hi = (uint64_t)g;
      uint64_t step = 4;
      lo = hi > step ? hi - step : 0;
      while (lo > 0 && (uint64_t)(uint32_t)s.off[lo] > pos) { hi = lo; step <<= 1; lo = hi > step ? hi - step : 0; }
    }
    while (hi - lo > 1) {
      const uint64_t mid = (lo + hi) >> 1;
      if ((uint64_t)(uint32_t)s.off[mid] <= pos) lo = mid; else hi = mid;
    }
    if (pos + L <= (uint64_t)(uint32_t)s.off[lo + 1] && (uint64_t)(uint32_t)s.off[lo] <= pos) {
      const uint64_t pid = s.tile_base * kTileRows + lo;
      atomicOr((unsigned long long*)&bits[pid >> 6], 1ull << (pid & 63));
    }
  }
}

}  // namespace

uint32_t contains_block(const prism_gpu_ctx* ctx, uint32_t needle_len) {
  const bool windows = ctx->contains_windows.load(std::memory_order_relaxed) != 0;
  return windows && needle_len >= 4 ? 64u : 256u;
}

namespace {
template <int BLOCK>
void launch_contains_find(prism_gpu_ctx* ctx, int s, uint32_t grid, const ContainsSeg* d_segs, uint32_t n_segs, uint64_t n_chunks,
                          const uint8_t* d_needle, uint32_t needle_len, uint64_t* d_matches, uint64_t match_capacity,
                          unsigned long long* d_n_matches) {
  auto kernel = s == 4 ? k_contains_find<4, BLOCK> : s == 2 ? k_contains_find<2, BLOCK>
                : s == 1 ? k_contains_find<1, BLOCK> : k_contains_find<0, BLOCK>;
  kernel<<<dim3(grid), dim3(BLOCK), 0, ctx->stream>>>(d_segs, n_segs, n_chunks, d_needle, needle_len, d_matches, match_capacity,
                                                     d_n_matches);
}
}  // namespace

bool launch_contains_all(prism_gpu_ctx* ctx, const ContainsSeg* d_segs, uint32_t n_segs, uint64_t n_chunks,
                         const uint8_t* d_needle, uint32_t needle_len, uint64_t* d_matches, uint64_t match_capacity,
                         unsigned long long* d_n_matches, uint64_t* bits) {
  if (n_chunks == 0) return true;
  PRISM_HIP(hipMemsetAsync(d_n_matches, 0, 8, ctx->stream));
  const uint32_t grid = (uint32_t)std::min<uint64_t>(n_chunks, (uint64_t)ctx->n_cus * 8);
  const bool windows = ctx->contains_windows.load(std::memory_order_relaxed) != 0;
  const int s = !windows ? 0 : needle_len >= 7 ? 4 : needle_len >= 5 ? 2 : needle_len == 4 ? 1 : 0;
  // as many workgroups as are resident together (2048 threads a CU)
  const uint32_t block = contains_block(ctx, needle_len);
  const uint32_t find_grid = (uint32_t)std::min<uint64_t>(n_chunks, (uint64_t)ctx->n_cus * (2048 / block));
  if (block == 64) launch_contains_find<64>(ctx, s, find_grid, d_segs, n_segs, n_chunks, d_needle, needle_len, d_matches, match_capacity, d_n_matches);
  else launch_contains_find<256>(ctx, s, find_grid, d_segs, n_segs, n_chunks, d_needle, needle_len, d_matches, match_capacity, d_n_matches);
  PRISM_HIP(hipGetLastError());
  // a list that overflowed leaves bits of real occurrences behind, which the caller's
  // fallback sets again
  k_contains_mark<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(d_segs, d_matches, d_n_matches, match_capacity, needle_len, bits);
  PRISM_HIP(hipGetLastError());
  unsigned long long n = 0;
  PRISM_HIP(hipMemcpyAsync(&n, d_n_matches, 8, hipMemcpyDeviceToHost, ctx->stream));
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  return n <= match_capacity;
}

void launch_contains_stream(prism_gpu_ctx* ctx, const ColPtr& col, uint64_t rows, uint64_t tile_base,
                            const uint8_t* needle, uint32_t needle_len, uint64_t* bits) {
  if (rows == 0) return;
  // total bytes = last offset; the host knows it, but reading it here keeps the call
  // signature small: offsets are rebased to start at 0 by the loader.
  int32_t last = 0;
  PRISM_HIP(hipMemcpyAsync(&last, (const int32_t*)col.values + rows, 4, hipMemcpyDeviceToHost, ctx->stream));
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  uint64_t total = (uint64_t)(uint32_t)last;
  if (total == 0) return;
  uint64_t chunks = ceil_div(total, (uint64_t)kBlock * 16);
  uint32_t grid = (uint32_t)std::min<uint64_t>(chunks, (uint64_t)ctx->n_cus * 16);
  k_contains_stream<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>((const int32_t*)col.values, col.bytes, rows, total,
                                                                 tile_base, needle, needle_len, bits);
  PRISM_HIP(hipGetLastError());
}

void launch_bitmap_and(prism_gpu_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n_words) {
  if (!n_words) return;
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(n_words, kBlock), (uint64_t)ctx->n_cus * 8);
  k_bitmap_binary<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(a, b, out, n_words, 0);
  PRISM_HIP(hipGetLastError());
}

void launch_bitmap_or(prism_gpu_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n_words) {
  if (!n_words) return;
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(n_words, kBlock), (uint64_t)ctx->n_cus * 8);
  k_bitmap_binary<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(a, b, out, n_words, 1);
  PRISM_HIP(hipGetLastError());
}

void launch_bitmap_count(prism_gpu_ctx* ctx, const uint64_t* bits, uint64_t n_words, unsigned long long* count) {
  if (!n_words) return;
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(n_words, kBlock), (uint64_t)ctx->n_cus * 8);
  k_bitmap_count<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(bits, n_words, count);
  PRISM_HIP(hipGetLastError());
}

}  // namespace prism
