// Hash partitioning of compact group rows by owner rank, ahead of the all-to-all between
// the GPUs of a node (the device-side counterpart of DataFusion's RepartitionExec(Hash)
// between AggregateExec(Partial) and AggregateExec(FinalPartitioned)).
//
// Two passes over the packed keys: per-owner counts (one LDS histogram per workgroup, one
// global atomic per owner and workgroup), then the scatter, in which a workgroup reserves a
// run per owner with one atomic each and its rows take consecutive places inside the runs.
// Rows are 8 * (1 + n_slots) bytes; both passes read the keys, the second moves the rows
// once: algorithmic bytes = n * 8 * (2 + 2 * (1 + n_slots)).
#include "device_utils.hpp"
#include "kernels.hpp"

namespace prism {

namespace {

constexpr int kMaxRanks = 64;

// Not the hash the group tables index with (they use the low bits of hash64): an owner's
// keys still spread over its whole table.
__device__ __forceinline__ uint32_t owner_of(uint64_t key, uint32_t world) {
  return (uint32_t)(((hash64(key ^ 0x5851F42D4C957F2Dull) >> 32) * (uint64_t)world) >> 32);
}

__global__ __launch_bounds__(kBlock) void k_owner_count(const uint64_t* keys, uint64_t n, uint32_t world,
                                                       unsigned long long* counts) {
  __shared__ uint32_t local[kMaxRanks];
  if (threadIdx.x < kMaxRanks) local[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
    atomicAdd(&local[owner_of(keys[i], world)], 1u);
  __syncthreads();
  if (threadIdx.x < world && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)local[threadIdx.x]);
}

__global__ __launch_bounds__(kBlock) void k_owner_scatter(const uint64_t* keys, const long long* state, uint64_t n, int n_slots,
                                                         uint32_t world, unsigned long long* cursor, uint64_t* out_keys,
                                                         long long* out_state) {
  __shared__ uint32_t local[kMaxRanks];
  __shared__ unsigned long long base[kMaxRanks];
  // every thread of a workgroup makes the same number of trips (the barriers are inside)
  const uint64_t trips = ceil_div(n, (uint64_t)gridDim.x * kBlock);
  for (uint64_t t = 0; t < trips; ++t) {
    const uint64_t i = (t * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
    if (threadIdx.x < kMaxRanks) local[threadIdx.x] = 0;
    __syncthreads();
    uint64_t key = 0;
    uint32_t o = 0, place = 0;
    if (i < n) {
      key = keys[i];
      o = owner_of(key, world);
      place = atomicAdd(&local[o], 1u);
    }
    __syncthreads();
    if (threadIdx.x < world && local[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)local[threadIdx.x]);
    __syncthreads();
    if (i < n) {
      const uint64_t dst = base[o] + place;
      out_keys[dst] = key;
      for (int s = 0; s < n_slots; ++s) out_state[(uint64_t)s * n + dst] = state[(uint64_t)s * n + i];
    }
    __syncthreads();
  }
}

}  // namespace

void launch_owner_count(prism_gpu_ctx* ctx, const uint64_t* keys, uint64_t n, uint32_t world, unsigned long long* counts) {
  if (world > (uint32_t)kMaxRanks) fail(PRISM_GPU_EINVAL, "at most 64 ranks");
  PRISM_HIP(hipMemsetAsync(counts, 0, (size_t)world * 8, ctx->stream));
  if (n == 0) return;
  KernelTimer timer(ctx, "owner_count", n * 8);
  const uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(n, kBlock), (uint64_t)ctx->n_cus * 8);
  k_owner_count<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(keys, n, world, counts);
  PRISM_HIP(hipGetLastError());
}

void launch_owner_scatter(prism_gpu_ctx* ctx, const uint64_t* keys, const int64_t* state, uint64_t n, int n_slots, uint32_t world,
                          unsigned long long* cursor, uint64_t* out_keys, int64_t* out_state) {
  if (n == 0) return;
  KernelTimer timer(ctx, "owner_scatter", n * 8 * (1 + 2 * (1 + (uint64_t)n_slots)));
  const uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(n, kBlock), (uint64_t)ctx->n_cus * 8);
  k_owner_scatter<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(keys, (const long long*)state, n, n_slots, world, cursor, out_keys,
                                                               (long long*)out_state);
  PRISM_HIP(hipGetLastError());
}

}  // namespace prism
