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
__device__ __forceinline__ uint32_t owner_of(const OwnerParams& p, uint64_t i) {
  uint64_t k = p.keys[i];
  if (p.owner_of == kOwnerHighBits) k >>= p.shift;
  else if (p.owner_of == kOwnerHighWord) k = p.keys_hi[i];
  else if (p.keys_hi) k ^= hash64(p.keys_hi[i]);
  return (uint32_t)(((hash64(k ^ 0x5851F42D4C957F2Dull) >> 32) * (uint64_t)p.world) >> 32);
}

__global__ __launch_bounds__(kBlock) void k_owner_count(OwnerParams p, unsigned long long* counts) {
  __shared__ uint32_t local[kMaxRanks];
  if (threadIdx.x < kMaxRanks) local[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < p.n; i += (uint64_t)gridDim.x * kBlock)
    atomicAdd(&local[owner_of(p, i)], 1u);
  __syncthreads();
  if (threadIdx.x < p.world && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)local[threadIdx.x]);
}

__global__ __launch_bounds__(kBlock) void k_owner_scatter(OwnerParams p, unsigned long long* cursor, uint64_t* out_keys,
                                                         uint64_t* out_keys_hi, long long* out_state) {
  __shared__ uint32_t local[kMaxRanks];
  __shared__ unsigned long long base[kMaxRanks];
  // every thread of a workgroup makes the same number of trips (the barriers are inside)
  const uint64_t trips = ceil_div(p.n, (uint64_t)gridDim.x * kBlock);
  for (uint64_t t = 0; t < trips; ++t) {
    const uint64_t i = (t * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
    if (threadIdx.x < kMaxRanks) local[threadIdx.x] = 0;
    __syncthreads();
    uint32_t o = 0, place = 0;
    if (i < p.n) {
      o = owner_of(p, i);
      place = atomicAdd(&local[o], 1u);
    }
    __syncthreads();
    if (threadIdx.x < p.world && local[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)local[threadIdx.x]);
    __syncthreads();
    if (i < p.n) {
      const uint64_t dst = base[o] + place;
      out_keys[dst] = p.keys[i];
      if (p.keys_hi) out_keys_hi[dst] = p.keys_hi[i];
      for (int s = 0; s < p.n_slots; ++s) out_state[(uint64_t)s * p.n + dst] = p.state[(uint64_t)s * p.n + i];
    }
    __syncthreads();
  }
}

}  // namespace

void launch_owner_count(prism_gpu_ctx* ctx, const OwnerParams& p, unsigned long long* counts) {
  if (p.world > (uint32_t)kMaxRanks) fail(PRISM_GPU_EINVAL, "at most 64 ranks");
  PRISM_HIP(hipMemsetAsync(counts, 0, (size_t)p.world * 8, ctx->stream));
  if (p.n == 0) return;
  KernelTimer timer(ctx, "owner_count", p.n * 8);
  const uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(p.n, kBlock), (uint64_t)ctx->n_cus * 8);
  k_owner_count<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(p, counts);
  PRISM_HIP(hipGetLastError());
}

void launch_owner_scatter(prism_gpu_ctx* ctx, const OwnerParams& p, unsigned long long* cursor, uint64_t* out_keys,
                          uint64_t* out_keys_hi, int64_t* out_state) {
  if (p.n == 0) return;
  const uint64_t width = 1 + (p.keys_hi ? 1 : 0) + (uint64_t)p.n_slots;
  KernelTimer timer(ctx, "owner_scatter", p.n * 8 * (1 + 2 * width));
  const uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(p.n, kBlock), (uint64_t)ctx->n_cus * 8);
  k_owner_scatter<<<dim3(grid), dim3(kBlock), 0, ctx->stream>>>(p, cursor, out_keys, out_keys_hi, (long long*)out_state);
  PRISM_HIP(hipGetLastError());
}

}  // namespace prism
