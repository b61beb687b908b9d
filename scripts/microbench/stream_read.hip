// Measured streaming-read ceiling of the device: a plain sum over a large buffer with
// 128-bit loads, grid-stride, one partial per workgroup.  The figure every HBM-bound
// kernel of the path is compared with, next to the 8 TB/s datasheet peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_nt(const uint4* p) {
  u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// CONTIG: every workgroup reads one contiguous share of the buffer (as the path's kernels
// do) instead of striding over the whole buffer.
template <int UNROLL, bool NT, bool CONTIG>
__global__ __launch_bounds__(256) void k_sum(const uint4* __restrict__ in, size_t n_vec, unsigned long long* out) {
  unsigned long long acc = 0;
  if (CONTIG) {
    const size_t per = (n_vec + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < n_vec ? lo + per : n_vec;
    size_t i = lo + threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < hi; i += UNROLL * 256) {
      uint4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = NT ? load_nt(&in[i + u * 256]) : in[i + u * 256];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += (unsigned long long)v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < hi; i += 256) { uint4 v = in[i]; acc += (unsigned long long)v.x + v.y + v.z + v.w; }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
    return;
  }
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n_vec; i += UNROLL * stride) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? load_nt(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += (unsigned long long)v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < n_vec; i += stride) { uint4 v = in[i]; acc += (unsigned long long)v.x + v.y + v.z + v.w; }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 20ull) << 30;
  void* buf; unsigned long long* out;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&out, 8));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipMemset(out, 0, 8));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const size_t n_vec = bytes / 16;
  printf("buffer %.1f GiB\n", bytes / 1073741824.0);
  auto run = [&](const char* name, auto kernel, int grid) -> int {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipEventRecord(a));
      kernel<<<grid, 256>>>((const uint4*)buf, n_vec, out);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    printf("%-34s grid %5d x 256: %8.3f ms  %7.1f GB/s  (%.1f %% of 8000)\n", name, grid, best, bytes / best / 1e6,
           bytes / best / 1e6 / 80.0);
    return 0;
  };
  for (int blocks_per_cu : {4, 8, 16}) {
    const int grid = 256 * blocks_per_cu;
    if (run("strided, 4 x 16 B per lane", k_sum<4, false, false>, grid)) return 1;
    if (run("strided, 8 x 16 B per lane", k_sum<8, false, false>, grid)) return 1;
    if (run("strided, 4 x 16 B, nontemporal", k_sum<4, true, false>, grid)) return 1;
    if (run("contiguous share, 4 x 16 B", k_sum<4, false, true>, grid)) return 1;
    if (run("contiguous share, 8 x 16 B", k_sum<8, false, true>, grid)) return 1;
    if (run("contiguous share, 8 x 16 B, nt", k_sum<8, true, true>, grid)) return 1;
  }
  return 0;
}
