// Loader-side kernels: column statistics (drive GROUP BY key packing), the expansion of
// dictionary-encoded string columns into offsets + bytes + table-wide codes, and the
// device half of the synthetic generators of prism_synth.h.
#include <algorithm>

#include "../../include/prism_synth.h"
#include "device_utils.hpp"

namespace prism {

namespace {

__global__ __launch_bounds__(kBlock) void k_col_stats(const void* values, const uint64_t* validity, int width,
                                                     int is_unsigned, uint64_t rows, ColStats* stats) {
  long long mn = 0x7fffffffffffffffll, mx = -0x7fffffffffffffffll - 1;
  unsigned long long nulls = 0;
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * kBlock) {
    if (!valid_at(validity, r)) { ++nulls; continue; }
    long long v = load_value(values, width, is_unsigned, r);
    mn = v < mn ? v : mn;
    mx = v > mx ? v : mx;
  }
  // One set of atomics per workgroup: they all go to one address, where the memory side takes
  // them one at a time -- a set per wave made them the whole cost of the kernel.
  __shared__ long long s_mn[kBlock / 64], s_mx[kBlock / 64];
  __shared__ unsigned long long s_nulls[kBlock / 64];
  mn = wave_min_i64(mn);
  mx = wave_max_i64(mx);
  nulls = wave_sum_u64(nulls);
  if ((threadIdx.x & 63) == 0) {
    s_mn[threadIdx.x >> 6] = mn;
    s_mx[threadIdx.x >> 6] = mx;
    s_nulls[threadIdx.x >> 6] = nulls;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) {
      mn = s_mn[w] < mn ? s_mn[w] : mn;
      mx = s_mx[w] > mx ? s_mx[w] : mx;
      nulls += s_nulls[w];
    }
    if (mn <= mx) {
      atomicMin(&stats->min, mn);
      atomicMax(&stats->max, mx);
    }
    if (nulls) atomicAdd(&stats->null_count, nulls);
  }
}

// (the same function as dict_lookup_hash_host)
__device__ __forceinline__ uint64_t dict_lookup_hash(const uint8_t* p, uint32_t len) {
  uint64_t h = 0xcbf29ce484222325ull ^ len;
  for (uint32_t i = 0; i < len; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
  h ^= h >> 32;
  return h * 0x9E3779B97F4A7C15ull >> 20;
}

__global__ __launch_bounds__(kBlock) void k_dict_lookup(DictLookup q) {
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < q.rows; r += (uint64_t)gridDim.x * kBlock) {
    uint32_t code = 0;
    if (valid_at(q.validity, r)) {
      const uint32_t lo = (uint32_t)q.offsets[r], len = (uint32_t)q.offsets[r + 1] - lo;
      const uint8_t* p = q.bytes + lo;
      uint32_t e = (uint32_t)dict_lookup_hash(p, len) & q.mask;
      bool found = false;
      for (uint32_t probes = 0; probes <= q.mask; ++probes, e = (e + 1) & q.mask) {
        const uint32_t v1 = q.slots[e];
        if (v1 == 0) break;
        const uint32_t from = q.dict_off[v1 - 1];
        if (q.dict_off[v1] - from != len) continue;
        uint32_t i = 0;
        while (i < len && q.dict_bytes[from + i] == p[i]) ++i;
        if (i == len) {
          code = v1 - 1;
          found = true;
          break;
        }
      }
      if (!found) q.report[0] = 1u;
    }
    q.codes[r] = code;
  }
}

// dst and src agree modulo 16: bytes up to the first 16-byte boundary, whole vectors, the tail.
__global__ __launch_bounds__(kBlock) void k_stage_copy(uint8_t* dst, const uint8_t* src, uint64_t bytes) {
  const uint64_t head = min((uint64_t)((16 - ((uintptr_t)dst & 15)) & 15), bytes), vecs = (bytes - head) >> 4;
  const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x, stride = (uint64_t)gridDim.x * kBlock;
  const uint4* s = (const uint4*)(src + head);
  uint4* d = (uint4*)(dst + head);
  for (uint64_t i = tid; i < vecs; i += stride) d[i] = s[i];
  if (tid < head) dst[tid] = src[tid];
  const uint64_t tail0 = head + (vecs << 4);
  if (tail0 + tid < bytes && tid < 16) dst[tail0 + tid] = src[tail0 + tid];
}

__global__ __launch_bounds__(kBlock) void k_synth(SynthColDev c, uint64_t first_row, uint64_t n_rows,
                                                 uint64_t total_rows) {
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * kBlock) {
    long long v = prism_synth_value(c.kind, (const int64_t*)c.p, c.cdf, c.cdf_len, c.column_key, first_row + r, total_rows);
    if (c.width == 8) ((long long*)c.dst)[r] = v;
    else if (c.width == 4) ((int*)c.dst)[r] = (int)v;
    else ((short*)c.dst)[r] = (short)v;
  }
}

// One thread per row writes the row's generated string between its offsets.
__global__ __launch_bounds__(kBlock) void k_synth_url(const int32_t* offsets, uint8_t* bytes, uint64_t n_rows,
                                                     uint64_t first_row, long long p0, long long p1, uint64_t column_key) {
  const int64_t p[2] = {p0, p1};
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * kBlock) {
    const uint32_t lo = (uint32_t)offsets[r], hi = (uint32_t)offsets[r + 1];
    prism_synth_url_bytes(bytes + lo, hi - lo, p, column_key, first_row + r);
  }
}

// ---- dictionary-encoded strings -> offsets, bytes, codes --------------------------------------
// A workgroup owns kDictRows consecutive rows, a thread kDictPerThread consecutive rows of
// them, so that a thread's strings are one contiguous run of the output bytes.
constexpr int kDictThreads = 256, kDictPerThread = 16, kDictRows = kDictThreads * kDictPerThread;

__device__ __forceinline__ uint32_t dict_len(const uint32_t* dict_off, uint32_t g) {
  return g == 0xffffffffu ? 0u : dict_off[g + 1] - dict_off[g];
}

// Sum of kDictThreads values; every thread gets the sum of the threads before it, thread 0 of
// the caller may read the total from *total.
__device__ __forceinline__ uint32_t dict_block_scan(uint32_t v, uint32_t* wave_sums, uint32_t* total) {
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
  for (int w = 0; w < kDictThreads / 64; ++w) {
    const uint32_t s = wave_sums[w];
    if (w < wave) before += s;
    all += s;
  }
  *total = all;
  return before + incl - v;
}

__global__ __launch_bounds__(kDictThreads) void k_dict_lens(DictExpand q) {
  __shared__ uint32_t wave_sums[kDictThreads / 64];
  const uint64_t row0 = (uint64_t)blockIdx.x * kDictRows + (uint64_t)threadIdx.x * kDictPerThread;
  uint32_t sum = 0;
  for (int i = 0; i < kDictPerThread; ++i)
    if (row0 + i < q.rows) sum += dict_len(q.dict_off, q.index[row0 + i]);
  uint32_t total;
  (void)dict_block_scan(sum, wave_sums, &total);
  if (threadIdx.x == 0) q.block_sums[blockIdx.x] = total;
}

// One workgroup: block_sums becomes its own exclusive prefix sum; the grand total is the last offset.
__global__ __launch_bounds__(kDictThreads) void k_dict_scan(DictExpand q, uint32_t n_blocks) {
  __shared__ uint32_t wave_sums[kDictThreads / 64];
  const uint32_t per = (n_blocks + kDictThreads - 1) / kDictThreads;
  const uint32_t lo = threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
  uint32_t sum = 0;
  for (uint32_t b = lo; b < hi; ++b) sum += q.block_sums[b];
  uint32_t total;
  uint32_t at = dict_block_scan(sum, wave_sums, &total);
  for (uint32_t b = lo; b < hi; ++b) {
    const uint32_t v = q.block_sums[b];
    q.block_sums[b] = at;
    at += v;
  }
  if (threadIdx.x == 0) {
    q.offsets[q.rows] = (int32_t)total;
    if (q.report) {
      q.report[0] = q.flag_in ? *q.flag_in : 0u;
      q.report[1] = total;
      __threadfence_system();
    }
  }
}

__global__ __launch_bounds__(kDictThreads) void k_dict_expand(DictExpand q) {
  __shared__ uint32_t wave_sums[kDictThreads / 64];
  const uint64_t row0 = (uint64_t)blockIdx.x * kDictRows + (uint64_t)threadIdx.x * kDictPerThread;
  uint32_t g[kDictPerThread];
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < kDictPerThread; ++i) {
    g[i] = row0 + i < q.rows ? q.index[row0 + i] : 0xffffffffu;
    sum += dict_len(q.dict_off, g[i]);
  }
  uint32_t total;
  uint32_t at = q.block_sums[blockIdx.x] + dict_block_scan(sum, wave_sums, &total);
#pragma unroll
  for (int i = 0; i < kDictPerThread; ++i) {
    if (row0 + i >= q.rows) break;
    q.offsets[row0 + i] = (int32_t)at;
    if (g[i] != 0xffffffffu) {
      const uint32_t from = q.dict_off[g[i]], len = q.dict_off[g[i] + 1] - from;
      for (uint32_t k = 0; k < len; ++k) q.bytes[at + k] = q.dict_bytes[from + k];
      at += len;
    }
    // (the index buffer doubles as the codes' buffer: a row's index is read by this thread alone)
    if (q.codes) q.codes[row0 + i] = g[i] != 0xffffffffu ? q.remap[g[i]] : 0u;
  }
}

// Indices into the batches' dictionaries -> table-wide codes, in place.  Rows [seg_row[s],
// seg_row[s + 1]) came with the dictionary whose codes start at remap[seg_base[s]].
__global__ __launch_bounds__(kBlock) void k_dict_codes(uint32_t* codes, const uint64_t* validity, uint64_t rows, const uint32_t* seg_row,
                                                      const uint32_t* seg_base, uint32_t n_segs, const uint32_t* remap) {
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * kBlock) {
    uint32_t lo = 0, hi = n_segs;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (seg_row[mid] <= r) lo = mid; else hi = mid;
    }
    codes[r] = valid_at(validity, r) ? remap[seg_base[lo] + codes[r]] : 0u;
  }
}

// Runs of the RLE / bit-packed hybrid -> one index per row (plus `add`, the place of the batch's
// dictionary among the dictionaries laid end to end).  A row finds its run by bisection over the
// runs' first rows; an index beyond the dictionary raises *flag.
__global__ __launch_bounds__(kBlock) void k_rle_expand(const PackedRun* runs, uint32_t n_runs, const uint8_t* bytes, uint64_t rows,
                                                      uint32_t add, uint32_t dict_len, uint32_t* out, uint32_t* flag) {
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * kBlock) {
    uint32_t lo = 0, hi = n_runs;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (runs[mid].first <= r) lo = mid; else hi = mid;
    }
    const PackedRun run = runs[lo];
    uint32_t v = run.value;
    if (run.kind & 0x80000000u) {
      const uint32_t bw = run.kind & 63u;
      const uint64_t bitpos = (r - run.first) * bw;
      const uint8_t* p = bytes + run.value + (bitpos >> 3);
      uint64_t word = 0;
#pragma unroll
      for (int b = 0; b < 8; ++b) word |= (uint64_t)p[b] << (8 * b);
      v = (uint32_t)((word >> (bitpos & 7)) & ((1ull << bw) - 1ull));
    }
    if (r - run.first >= run.count || v >= dict_len) {
      *flag = 1u;
      v = 0;
    }
    out[r] = add + v;
  }
}

}  // namespace

void launch_rle_expand(prism_gpu_ctx* ctx, const PackedRun* runs, uint32_t n_runs, const uint8_t* bytes, uint64_t rows, uint32_t add,
                       uint32_t dict_len, uint32_t* out, uint32_t* flag, hipStream_t stream) {
  if (rows == 0 || n_runs == 0) return;
  {  // (counted, not timed: the launch is on the caller's stream, among its copies)
    std::lock_guard<std::mutex> g(ctx->mu);
    const int i = stat_index(ctx, "rle_expand");
    if (i >= 0) {
      ctx->stats.kernels[i].launches += 1;
      ctx->stats.kernels[i].algorithmic_bytes += rows * 4;
    }
  }
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(rows, (uint64_t)kBlock * 4), (uint64_t)ctx->n_cus * 8);
  k_rle_expand<<<dim3(std::max(1u, grid)), dim3(kBlock), 0, stream>>>(runs, n_runs, bytes, rows, add, dict_len, out, flag);
  PRISM_HIP(hipGetLastError());
}

void launch_dict_lens(prism_gpu_ctx* ctx, const DictExpand& q, hipStream_t stream) {
  if (q.rows == 0) return;
  const uint32_t n_blocks = (uint32_t)dict_expand_blocks(q.rows);
  k_dict_lens<<<dim3(n_blocks), dim3(kDictThreads), 0, stream>>>(q);
  k_dict_scan<<<dim3(1), dim3(kDictThreads), 0, stream>>>(q, n_blocks);
  PRISM_HIP(hipGetLastError());
}

void launch_dict_fill(prism_gpu_ctx* ctx, const DictExpand& q, hipStream_t stream) {
  if (q.rows == 0) return;
  k_dict_expand<<<dim3((uint32_t)dict_expand_blocks(q.rows)), dim3(kDictThreads), 0, stream>>>(q);
  PRISM_HIP(hipGetLastError());
}

void launch_dict_codes(prism_gpu_ctx* ctx, uint32_t* codes, const uint64_t* validity, uint64_t rows, const uint32_t* seg_row,
                       const uint32_t* seg_base, uint32_t n_segs, const uint32_t* remap, hipStream_t stream) {
  if (rows == 0) return;
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(rows, (uint64_t)kBlock * 4), (uint64_t)ctx->n_cus * 8);
  k_dict_codes<<<dim3(std::max(1u, grid)), dim3(kBlock), 0, stream>>>(codes, validity, rows, seg_row, seg_base, n_segs, remap);
  PRISM_HIP(hipGetLastError());
}

size_t dict_expand_blocks(uint64_t rows) { return (size_t)ceil_div(rows, (uint64_t)kDictRows); }

void launch_dict_expand(prism_gpu_ctx* ctx, const DictExpand& q, hipStream_t stream) {
  if (q.rows == 0) return;
  launch_dict_lens(ctx, q, stream);
  launch_dict_fill(ctx, q, stream);
}

void launch_synth_url(prism_gpu_ctx* ctx, const int32_t* offsets, uint8_t* bytes, uint64_t n_rows, uint64_t first_row,
                      int64_t p0, int64_t p1, uint64_t column_key) {
  if (n_rows == 0) return;
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(n_rows, (uint64_t)kBlock), (uint64_t)ctx->n_cus * 16);
  k_synth_url<<<dim3(std::max(1u, grid)), dim3(kBlock), 0, ctx->stream>>>(offsets, bytes, n_rows, first_row, p0, p1, column_key);
  PRISM_HIP(hipGetLastError());
}

void launch_dict_lookup(prism_gpu_ctx* ctx, const DictLookup& q, hipStream_t stream) {
  if (q.rows == 0) return;
  {  // (counted, not timed: the launch is on the caller's stream, among its copies)
    std::lock_guard<std::mutex> g(ctx->mu);
    const int i = stat_index(ctx, "dict_lookup");
    if (i >= 0) {
      ctx->stats.kernels[i].launches += 1;
      ctx->stats.kernels[i].algorithmic_bytes += q.rows * 4;
    }
  }
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(q.rows, (uint64_t)kBlock * 4), (uint64_t)ctx->n_cus * 8);
  k_dict_lookup<<<dim3(std::max(1u, grid)), dim3(kBlock), 0, stream>>>(q);
  PRISM_HIP(hipGetLastError());
}

void launch_stage_copy(prism_gpu_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return;
  if ((((uintptr_t)dst ^ (uintptr_t)src) & 15) != 0) fail(PRISM_GPU_EINTERNAL, "stage copy between addresses that disagree modulo 16");
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div((uint64_t)bytes, (uint64_t)kBlock * 16 * 4), (uint64_t)ctx->n_cus * 4);
  k_stage_copy<<<dim3(std::max(1u, grid)), dim3(kBlock), 0, stream>>>((uint8_t*)dst, (const uint8_t*)src, (uint64_t)bytes);
  PRISM_HIP(hipGetLastError());
}

void launch_col_stats(prism_gpu_ctx* ctx, const void* values, const uint64_t* validity, int width, bool is_unsigned,
                      uint64_t rows, ColStats* stats, hipStream_t stream) {
  if (rows == 0) return;
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(rows, (uint64_t)kBlock * 8), (uint64_t)ctx->n_cus * 2);
  k_col_stats<<<dim3(std::max(1u, grid)), dim3(kBlock), 0, stream>>>(values, validity, width, is_unsigned ? 1 : 0, rows,
                                                                   stats);
  PRISM_HIP(hipGetLastError());
}

void launch_synth(prism_gpu_ctx* ctx, const SynthColDev& col, uint64_t first_row, uint64_t n_rows,
                  uint64_t total_rows) {
  if (n_rows == 0) return;
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(n_rows, (uint64_t)kBlock * 4), (uint64_t)ctx->n_cus * 16);
  k_synth<<<dim3(std::max(1u, grid)), dim3(kBlock), 0, ctx->stream>>>(col, first_row, n_rows, total_rows);
  PRISM_HIP(hipGetLastError());
}

}  // namespace prism
