// Loader-side kernels: column statistics (drive GROUP BY key packing) and the
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
  mn = wave_min_i64(mn);
  mx = wave_max_i64(mx);
  nulls = wave_sum_u64(nulls);
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&stats->min, mn);
    atomicMax(&stats->max, mx);
    if (nulls) atomicAdd(&stats->null_count, nulls);
  }
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

}  // namespace

void launch_synth_url(prism_gpu_ctx* ctx, const int32_t* offsets, uint8_t* bytes, uint64_t n_rows, uint64_t first_row,
                      int64_t p0, int64_t p1, uint64_t column_key) {
  if (n_rows == 0) return;
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(n_rows, (uint64_t)kBlock), (uint64_t)ctx->n_cus * 16);
  k_synth_url<<<dim3(std::max(1u, grid)), dim3(kBlock), 0, ctx->stream>>>(offsets, bytes, n_rows, first_row, p0, p1, column_key);
  PRISM_HIP(hipGetLastError());
}

void launch_col_stats(prism_gpu_ctx* ctx, const void* values, const uint64_t* validity, int width, bool is_unsigned,
                      uint64_t rows, ColStats* stats, hipStream_t stream) {
  if (rows == 0) return;
  uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(rows, (uint64_t)kBlock * 8), (uint64_t)ctx->n_cus * 8);
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
