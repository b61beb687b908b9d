// Selection-vector compaction: bitmap of selected rows -> ascending padded row ids.
//
// Three small launches: per-tile popcounts, an exclusive scan of the tile counts,
// and the write pass, in which wave-level prefix sums place every set bit.  The
// bitmap is 1 bit per row, so all three together move a few percent of what the
// predicate that produced it had to read.
#include <algorithm>

#include "context.hpp"
#include "device_utils.hpp"

namespace prism {

namespace {

__global__ __launch_bounds__(64) void k_tile_counts(const uint64_t* bits, uint64_t n_tiles, uint32_t* counts) {
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    unsigned long long c = __popcll(bits[tile * kTileWords + threadIdx.x]);
    c = wave_sum_u64(c);
    if (threadIdx.x == 0) counts[tile] = (uint32_t)c;
  }
}

// Single-block exclusive scan; counts[n] receives the total.  Used directly for up to 8192
// tiles, and for the run totals of larger scans.
__global__ __launch_bounds__(1024) void k_scan_counts(uint32_t* counts, uint64_t n, unsigned long long* total) {
  __shared__ uint32_t wave_tot[16];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint64_t base = 0; base < n; base += 1024) {
    uint64_t i = base + threadIdx.x;
    uint32_t v = i < n ? counts[i] : 0;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t o = __shfl_up(incl, off, 64);
      if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wave_tot[w];
    unsigned long long c = carry;
    if (i < n) counts[i] = (uint32_t)(c + before + incl - v);
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counts[n] = (uint32_t)carry;
    *total = carry;
  }
}

// Many tiles: every workgroup scans a run of 1024 counts in place and leaves the run's total;
// the totals are scanned by the single-block kernel above and added back.
__global__ __launch_bounds__(1024) void k_scan_runs(uint32_t* counts, uint64_t n, uint32_t* run_totals) {
  __shared__ uint32_t wave_tot[16];
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  const uint32_t v = i < n ? counts[i] : 0;
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
  if (i < n) counts[i] = before + incl - v;
  if (threadIdx.x == 1023) run_totals[blockIdx.x] = before + incl;
}

__global__ __launch_bounds__(1024) void k_add_run_offsets(uint32_t* counts, uint64_t n, const uint32_t* run_offsets,
                                                          uint64_t n_runs, unsigned long long* total) {
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  if (i < n) counts[i] += run_offsets[blockIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    counts[n] = run_offsets[n_runs];
    *total = run_offsets[n_runs];
  }
}

__global__ __launch_bounds__(kBlock) void k_write_selvec(const uint64_t* bits, uint64_t n_tiles,
                                                        const uint32_t* tile_offsets, uint32_t* selvec) {
  __shared__ uint32_t word_prefix[kTileWords];
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t out_base = tile_offsets[tile];
    if (tile_offsets[tile + 1] == out_base) continue;  // block uniform
    if (threadIdx.x < kTileWords) {
      uint32_t c = (uint32_t)__popcll(bits[tile * kTileWords + threadIdx.x]);
      uint32_t incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(incl, off, 64);
        if ((int)threadIdx.x >= off) incl += o;
      }
      word_prefix[threadIdx.x] = incl - c;
    }
    __syncthreads();
    // four threads per word, 16 bits each
    uint32_t w = threadIdx.x >> 2, q = threadIdx.x & 3;
    uint64_t word = bits[tile * kTileWords + w];
    uint32_t below = (uint32_t)__popcll(word & ((1ull << (16 * q)) - 1ull));
    uint32_t mine = (uint32_t)((word >> (16 * q)) & 0xffffull);
    uint32_t pos = out_base + word_prefix[w] + below;
    uint32_t row0 = (uint32_t)(tile * kTileRows) + w * 64 + q * 16;
    while (mine) {
      int b = __ffs(mine) - 1;
      mine &= mine - 1;
      selvec[pos++] = row0 + b;
    }
    __syncthreads();
  }
}

}  // namespace

void launch_compact(prism_gpu_ctx* ctx, const uint64_t* bits, uint64_t n_tiles, uint32_t* tile_offsets,
                    uint32_t* selvec, unsigned long long* total) {
  if (n_tiles == 0) return;
  uint32_t grid = (uint32_t)std::min<uint64_t>(n_tiles, (uint64_t)ctx->n_cus * 32);
  if (selvec == nullptr) {
    k_tile_counts<<<dim3(grid), dim3(64), 0, ctx->stream>>>(bits, n_tiles, tile_offsets);
    PRISM_HIP(hipGetLastError());
    if (n_tiles <= 8192) {
      k_scan_counts<<<dim3(1), dim3(1024), 0, ctx->stream>>>(tile_offsets, n_tiles, total);
    } else {
      // (the block of run totals goes back to the pool right away: the stream orders its reuse)
      const uint64_t n_runs = ceil_div(n_tiles, 1024);
      Scratch runs(ctx, (n_runs + 1) * 4);
      k_scan_runs<<<dim3((uint32_t)n_runs), dim3(1024), 0, ctx->stream>>>(tile_offsets, n_tiles, runs.as<uint32_t>());
      k_scan_counts<<<dim3(1), dim3(1024), 0, ctx->stream>>>(runs.as<uint32_t>(), n_runs, total);
      k_add_run_offsets<<<dim3((uint32_t)n_runs), dim3(1024), 0, ctx->stream>>>(tile_offsets, n_tiles, runs.as<uint32_t>(), n_runs, total);
    }
    PRISM_HIP(hipGetLastError());
    return;
  }
  uint32_t wgrid = (uint32_t)std::min<uint64_t>(n_tiles, (uint64_t)ctx->n_cus * 8);
  k_write_selvec<<<dim3(wgrid), dim3(kBlock), 0, ctx->stream>>>(bits, n_tiles, tile_offsets, selvec);
  PRISM_HIP(hipGetLastError());
}

}  // namespace prism
