// A dictionary for a string column built on the device.
//
// The host-side dictionary of a column (table.cpp) stops at 65 536 values; beyond that the
// column only has its raw form (offsets + bytes) in HBM and could not be a GROUP BY key.  The
// reference groups by any Utf8 column (its columns are plain Arrow strings:
// src/prism-query/src/providers/schema_provider.rs:96-107).  These kernels give every row of
// such a column a code, exactly:
//   1. a 63-bit hash of every row's bytes;
//   2. the hashes go into an open-addressing table; every entry keeps the smallest row that
//      carries its hash (the entry's representative);
//   3. the occupied entries are numbered in entry order (counts per workgroup, a scan, the
//      numbering): the number is the code, and rep_by_code[code] the representative row;
//   4. every row looks its entry up again, compares its bytes with the representative's -- two
//      different strings with one hash raise a flag, and the host starts over with another
//      seed -- and takes the entry's code;
//   5. the representatives' strings are gathered (lengths, a scan, the bytes): the dictionary.
// A row is addressed by its global number over the column's chunks (StrSeg::base + row).
#include <algorithm>

#include "device_utils.hpp"

namespace prism {

namespace {

constexpr int kStrBlock = 256;
constexpr uint64_t kEmptyHash = ~0ull;

__device__ __forceinline__ uint32_t str_seg_of(const StrSeg* segs, uint32_t n_segs, uint64_t row) {
  uint32_t lo = 0, hi = n_segs;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (segs[mid].base <= row) lo = mid; else hi = mid;
  }
  return lo;
}

// 63 bits from the bytes: eight at a time through the 64-bit finaliser, the length mixed in.
__device__ __forceinline__ uint64_t str_hash(const uint8_t* p, uint32_t len, uint64_t seed) {
  uint64_t h = seed ^ ((uint64_t)len * 0x9E3779B97F4A7C15ull);
  uint32_t i = 0;
  for (; i + 8 <= len; i += 8) {
    uint64_t w = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) w |= (uint64_t)p[i + b] << (8 * b);
    h = hash64(h ^ w) + 0xD6E8FEB86659FD93ull;
  }
  uint64_t w = 0;
  for (uint32_t b = 0; i + b < len; ++b) w |= (uint64_t)p[i + b] << (8 * b);
  h = hash64(h ^ w ^ 0xA0761D6478BD642Full);
  return h >> 1;
}

__global__ __launch_bounds__(kStrBlock) void k_str_hash(const StrSeg* segs, uint32_t n_segs, uint64_t n_rows, uint64_t seed, uint64_t* hashes) {
  for (uint64_t r = (uint64_t)blockIdx.x * kStrBlock + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * kStrBlock) {
    const StrSeg& s = segs[str_seg_of(segs, n_segs, r)];
    const uint64_t i = r - s.base;
    if (!valid_at(s.validity, i)) { hashes[r] = kEmptyHash; continue; }
    const uint32_t lo = (uint32_t)s.off[i], hi = (uint32_t)s.off[i + 1];
    hashes[r] = str_hash(s.bytes + lo, hi - lo, seed);
  }
}

__device__ __forceinline__ uint64_t str_home(uint64_t h, uint64_t mask) { return hash64(h) & mask; }

__global__ __launch_bounds__(kStrBlock) void k_str_insert(const uint64_t* hashes, uint64_t n_rows, uint64_t* keys, uint32_t* rep, uint64_t mask) {
  for (uint64_t r = (uint64_t)blockIdx.x * kStrBlock + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * kStrBlock) {
    const uint64_t h = hashes[r];
    if (h == kEmptyHash) continue;
    uint64_t e = str_home(h, mask);
    for (;;) {  // (the table holds at most half as many keys as entries: a free entry is always met)
      const uint64_t seen = keys[e];
      if (seen == h) break;
      if (seen == kEmptyHash) {
        const uint64_t was = atomicCAS((unsigned long long*)&keys[e], (unsigned long long)kEmptyHash, (unsigned long long)h);
        if (was == kEmptyHash || was == h) break;
      }
      e = (e + 1) & mask;
    }
    atomicMin(&rep[e], (uint32_t)r);
  }
}

// Occupied entries per workgroup of kStrBlock * 16 entries.
constexpr int kStrPerThread = 16, kStrSpan = kStrBlock * kStrPerThread;

__device__ __forceinline__ uint32_t str_block_scan(uint32_t v, uint32_t* wave_sums, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kStrBlock / 64; ++w) {
    const uint32_t s = wave_sums[w];
    if (w < wave) before += s;
    all += s;
  }
  *total = all;
  return before + incl - v;
}

__global__ __launch_bounds__(kStrBlock) void k_str_count(const uint64_t* keys, uint64_t capacity, uint32_t* block_sums) {
  __shared__ uint32_t wave_sums[kStrBlock / 64];
  const uint64_t e0 = (uint64_t)blockIdx.x * kStrSpan + (uint64_t)threadIdx.x * kStrPerThread;
  uint32_t n = 0;
  for (int i = 0; i < kStrPerThread; ++i)
    if (e0 + i < capacity && keys[e0 + i] != kEmptyHash) ++n;
  uint32_t total;
  (void)str_block_scan(n, wave_sums, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// One workgroup: block_sums becomes its exclusive prefix sum, *n_values the number of occupied entries.
__global__ __launch_bounds__(kStrBlock) void k_str_scan(uint32_t* block_sums, uint32_t n_blocks, uint32_t* n_values) {
  __shared__ uint32_t wave_sums[kStrBlock / 64];
  const uint32_t per = (n_blocks + kStrBlock - 1) / kStrBlock;
  const uint32_t lo = threadIdx.x * per < n_blocks ? threadIdx.x * per : n_blocks, hi = lo + per < n_blocks ? lo + per : n_blocks;
  uint32_t sum = 0;
  for (uint32_t b = lo; b < hi; ++b) sum += block_sums[b];
  uint32_t total;
  uint32_t at = str_block_scan(sum, wave_sums, &total);
  for (uint32_t b = lo; b < hi; ++b) {
    const uint32_t v = block_sums[b];
    block_sums[b] = at;
    at += v;
  }
  if (threadIdx.x == 0) *n_values = total;
}

__global__ __launch_bounds__(kStrBlock) void k_str_number(const uint64_t* keys, uint64_t capacity, const uint32_t* block_sums, const uint32_t* rep,
                                                         uint32_t* code_of_entry, uint32_t* rep_by_code) {
  __shared__ uint32_t wave_sums[kStrBlock / 64];
  const uint64_t e0 = (uint64_t)blockIdx.x * kStrSpan + (uint64_t)threadIdx.x * kStrPerThread;
  uint32_t n = 0;
  for (int i = 0; i < kStrPerThread; ++i)
    if (e0 + i < capacity && keys[e0 + i] != kEmptyHash) ++n;
  uint32_t total;
  uint32_t code = block_sums[blockIdx.x] + str_block_scan(n, wave_sums, &total);
  for (int i = 0; i < kStrPerThread; ++i) {
    if (e0 + i >= capacity || keys[e0 + i] == kEmptyHash) continue;
    code_of_entry[e0 + i] = code;
    rep_by_code[code] = rep[e0 + i];
    ++code;
  }
}

__global__ __launch_bounds__(kStrBlock) void k_str_codes(const StrSeg* segs, uint32_t n_segs, uint64_t n_rows, const uint64_t* hashes,
                                                        const uint64_t* keys, const uint32_t* rep, const uint32_t* code_of_entry,
                                                        uint64_t mask, uint32_t* flag) {
  for (uint64_t r = (uint64_t)blockIdx.x * kStrBlock + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * kStrBlock) {
    const uint32_t si = str_seg_of(segs, n_segs, r);
    const StrSeg& s = segs[si];
    const uint64_t i = r - s.base;
    const uint64_t h = hashes[r];
    if (h == kEmptyHash) { s.codes[i] = 0; continue; }
    uint64_t e = str_home(h, mask);
    while (keys[e] != h) e = (e + 1) & mask;  // (it was inserted)
    const uint64_t q = rep[e];
    if (q != r) {
      // same hash: the same bytes?
      const StrSeg& t = segs[str_seg_of(segs, n_segs, q)];
      const uint64_t j = q - t.base;
      const uint32_t a0 = (uint32_t)s.off[i], a1 = (uint32_t)s.off[i + 1], b0 = (uint32_t)t.off[j], b1 = (uint32_t)t.off[j + 1];
      bool same = a1 - a0 == b1 - b0;
      for (uint32_t k = 0; same && k < a1 - a0; ++k) same = s.bytes[a0 + k] == t.bytes[b0 + k];
      if (!same) *flag = 1u;
    }
    s.codes[i] = code_of_entry[e];
  }
}

// The dictionary's strings: lengths of the representatives, their scan, the bytes.
__global__ __launch_bounds__(kStrBlock) void k_str_value_lens(const StrSeg* segs, uint32_t n_segs, const uint32_t* rep_by_code, uint32_t n_values,
                                                             uint32_t* block_sums) {
  __shared__ uint32_t wave_sums[kStrBlock / 64];
  const uint64_t c0 = (uint64_t)blockIdx.x * kStrSpan + (uint64_t)threadIdx.x * kStrPerThread;
  uint32_t sum = 0;
  for (int i = 0; i < kStrPerThread; ++i) {
    if (c0 + i >= n_values) break;
    const uint64_t q = rep_by_code[c0 + i];
    const StrSeg& t = segs[str_seg_of(segs, n_segs, q)];
    const uint64_t j = q - t.base;
    sum += (uint32_t)(t.off[j + 1] - t.off[j]);
  }
  uint32_t total;
  (void)str_block_scan(sum, wave_sums, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kStrBlock) void k_str_values(const StrSeg* segs, uint32_t n_segs, const uint32_t* rep_by_code, uint32_t n_values,
                                                         const uint32_t* block_sums, uint32_t* value_off, uint8_t* value_bytes,
                                                         uint64_t bytes_capacity) {
  __shared__ uint32_t wave_sums[kStrBlock / 64];
  const uint64_t c0 = (uint64_t)blockIdx.x * kStrSpan + (uint64_t)threadIdx.x * kStrPerThread;
  uint32_t lens[kStrPerThread];
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < kStrPerThread; ++i) {
    lens[i] = 0;
    if (c0 + i < n_values) {
      const uint64_t q = rep_by_code[c0 + i];
      const StrSeg& t = segs[str_seg_of(segs, n_segs, q)];
      const uint64_t j = q - t.base;
      lens[i] = (uint32_t)(t.off[j + 1] - t.off[j]);
    }
    sum += lens[i];
  }
  uint32_t total;
  uint32_t at = block_sums[blockIdx.x] + str_block_scan(sum, wave_sums, &total);
#pragma unroll
  for (int i = 0; i < kStrPerThread; ++i) {
    if (c0 + i >= n_values) break;
    value_off[c0 + i] = at;
    if (value_bytes && (uint64_t)at + lens[i] <= bytes_capacity) {
      const uint64_t q = rep_by_code[c0 + i];
      const StrSeg& t = segs[str_seg_of(segs, n_segs, q)];
      const uint64_t j = q - t.base;
      const uint8_t* from = t.bytes + (uint32_t)t.off[j];
      for (uint32_t k = 0; k < lens[i]; ++k) value_bytes[at + k] = from[k];
    }
    at += lens[i];
  }
  if (c0 <= (uint64_t)n_values && (uint64_t)n_values < c0 + kStrPerThread) value_off[n_values] = at;  // (the thread that owns the place behind the last value)
}

uint32_t row_grid(const prism_gpu_ctx* ctx, uint64_t rows) {
  return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(rows, (uint64_t)kStrBlock), (uint64_t)ctx->n_cus * 16));
}

}  // namespace

size_t str_dict_blocks(uint64_t entries) { return (size_t)std::max<uint64_t>(1, ceil_div(entries + 1, (uint64_t)kStrSpan)); }

void launch_str_hash(prism_gpu_ctx* ctx, const StrSeg* segs, uint32_t n_segs, uint64_t n_rows, uint64_t seed, uint64_t* hashes) {
  if (n_rows == 0) return;
  k_str_hash<<<dim3(row_grid(ctx, n_rows)), dim3(kStrBlock), 0, ctx->stream>>>(segs, n_segs, n_rows, seed, hashes);
  PRISM_HIP(hipGetLastError());
}

void launch_str_insert(prism_gpu_ctx* ctx, const uint64_t* hashes, uint64_t n_rows, uint64_t* keys, uint32_t* rep, uint64_t capacity) {
  if (n_rows == 0) return;
  k_str_insert<<<dim3(row_grid(ctx, n_rows)), dim3(kStrBlock), 0, ctx->stream>>>(hashes, n_rows, keys, rep, capacity - 1);
  PRISM_HIP(hipGetLastError());
}

void launch_str_number(prism_gpu_ctx* ctx, const uint64_t* keys, uint64_t capacity, const uint32_t* rep, uint32_t* block_sums,
                       uint32_t* code_of_entry, uint32_t* rep_by_code, uint32_t* n_values) {
  const uint32_t n_blocks = (uint32_t)ceil_div(capacity, (uint64_t)kStrSpan);
  k_str_count<<<dim3(n_blocks), dim3(kStrBlock), 0, ctx->stream>>>(keys, capacity, block_sums);
  k_str_scan<<<dim3(1), dim3(kStrBlock), 0, ctx->stream>>>(block_sums, n_blocks, n_values);
  k_str_number<<<dim3(n_blocks), dim3(kStrBlock), 0, ctx->stream>>>(keys, capacity, block_sums, rep, code_of_entry, rep_by_code);
  PRISM_HIP(hipGetLastError());
}

void launch_str_codes(prism_gpu_ctx* ctx, const StrSeg* segs, uint32_t n_segs, uint64_t n_rows, const uint64_t* hashes, const uint64_t* keys,
                      const uint32_t* rep, const uint32_t* code_of_entry, uint64_t capacity, uint32_t* flag) {
  if (n_rows == 0) return;
  k_str_codes<<<dim3(row_grid(ctx, n_rows)), dim3(kStrBlock), 0, ctx->stream>>>(segs, n_segs, n_rows, hashes, keys, rep, code_of_entry,
                                                                               capacity - 1, flag);
  PRISM_HIP(hipGetLastError());
}

void launch_str_values(prism_gpu_ctx* ctx, const StrSeg* segs, uint32_t n_segs, const uint32_t* rep_by_code, uint32_t n_values,
                       uint32_t* block_sums, uint32_t* value_off, uint8_t* value_bytes, uint64_t bytes_capacity) {
  const uint32_t n_blocks = (uint32_t)str_dict_blocks(n_values);
  uint32_t* unused_total = block_sums + n_blocks;  // (one word behind the sums)
  k_str_value_lens<<<dim3(n_blocks), dim3(kStrBlock), 0, ctx->stream>>>(segs, n_segs, rep_by_code, n_values, block_sums);
  k_str_scan<<<dim3(1), dim3(kStrBlock), 0, ctx->stream>>>(block_sums, n_blocks, unused_total);
  k_str_values<<<dim3(n_blocks), dim3(kStrBlock), 0, ctx->stream>>>(segs, n_segs, rep_by_code, n_values, block_sums, value_off, value_bytes,
                                                                  bytes_capacity);
  PRISM_HIP(hipGetLastError());
}

}  // namespace prism
