// Does the width of the per-lane load matter for a streaming read of three columns
// (8 + 4 + 8 bytes per row, as the headline GROUP BY reads)?  Variant A: row per lane,
// one dwordx2 / dword / dwordx2 load per row and column, 4 rows per lane in flight (what
// the group-by kernel does).  Variant B: the same bytes with 16-byte loads only (a lane
// holds 2 rows of the 8-byte columns and 4 rows of the 4-byte column per load).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 512;
constexpr int kRowsPerStep = kBlock * 4;  // 4 rows per lane

template <bool NT>
__global__ __launch_bounds__(kBlock) void k_narrow(const unsigned long long* a, const unsigned* b, const unsigned long long* c,
                                                    size_t rows, unsigned long long* out) {
  const size_t steps = rows / kRowsPerStep, per = (steps + gridDim.x - 1) / gridDim.x;
  const size_t s0 = per * blockIdx.x, s1 = s0 + per < steps ? s0 + per : steps;
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long acc = 0;
  for (size_t s = s0; s < s1; ++s) {
    unsigned long long va[4], vc[4];
    unsigned vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t r = s * kRowsPerStep + (wave * 4 + u) * 64 + lane;
      if (NT) {
        u32x2 x = __builtin_nontemporal_load((const u32x2*)&a[r]);
        u32x2 z = __builtin_nontemporal_load((const u32x2*)&c[r]);
        va[u] = x.x | ((unsigned long long)x.y << 32);
        vc[u] = z.x | ((unsigned long long)z.y << 32);
        vb[u] = __builtin_nontemporal_load(&b[r]);
      } else {
        va[u] = a[r]; vb[u] = b[r]; vc[u] = c[r];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += va[u] + vb[u] + vc[u];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (lane == 0) atomicAdd(out, acc);
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void k_wide(const unsigned long long* a, const unsigned* b, const unsigned long long* c,
                                                  size_t rows, unsigned long long* out) {
  // per step and workgroup: 2048 rows = 1024 16-B vectors of a, 512 of b, 1024 of c
  const size_t steps = rows / kRowsPerStep, per = (steps + gridDim.x - 1) / gridDim.x;
  const size_t s0 = per * blockIdx.x, s1 = s0 + per < steps ? s0 + per : steps;
  unsigned long long acc = 0;
  for (size_t s = s0; s < s1; ++s) {
    const u32x4* pa = (const u32x4*)(a + s * kRowsPerStep);
    const u32x4* pb = (const u32x4*)(b + s * kRowsPerStep);
    const u32x4* pc = (const u32x4*)(c + s * kRowsPerStep);
    u32x4 v[5];
    if (NT) {
      v[0] = __builtin_nontemporal_load(&pa[threadIdx.x]);
      v[1] = __builtin_nontemporal_load(&pa[threadIdx.x + kBlock]);
      v[2] = __builtin_nontemporal_load(&pb[threadIdx.x]);
      v[3] = __builtin_nontemporal_load(&pc[threadIdx.x]);
      v[4] = __builtin_nontemporal_load(&pc[threadIdx.x + kBlock]);
    } else {
      v[0] = pa[threadIdx.x]; v[1] = pa[threadIdx.x + kBlock]; v[2] = pb[threadIdx.x];
      v[3] = pc[threadIdx.x]; v[4] = pc[threadIdx.x + kBlock];
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) acc += (unsigned long long)v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

int main(int argc, char** argv) {
  const size_t rows = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull) / kRowsPerStep * kRowsPerStep;
  void *a, *b, *c; unsigned long long* out;
  CHECK(hipMalloc(&a, rows * 8)); CHECK(hipMalloc(&b, rows * 4)); CHECK(hipMalloc(&c, rows * 8)); CHECK(hipMalloc(&out, 8));
  CHECK(hipMemset(a, 1, rows * 8)); CHECK(hipMemset(b, 1, rows * 4)); CHECK(hipMemset(c, 1, rows * 8)); CHECK(hipMemset(out, 0, 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double bytes = rows * 20.0;
  auto run = [&](const char* name, auto kernel, int grid) -> int {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipEventRecord(e0));
      kernel<<<grid, kBlock>>>((const unsigned long long*)a, (const unsigned*)b, (const unsigned long long*)c, rows, out);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-44s grid %5d x %d: %8.3f ms  %7.1f GB/s\n", name, grid, kBlock, best, bytes / best / 1e6);
    return 0;
  };
  for (int grid : {512, 1024, 2048}) {
    if (run("row per lane, 8/4/8-byte loads", k_narrow<false>, grid)) return 1;
    if (run("row per lane, 8/4/8-byte loads, nontemporal", k_narrow<true>, grid)) return 1;
    if (run("16-byte loads", k_wide<false>, grid)) return 1;
    if (run("16-byte loads, nontemporal", k_wide<true>, grid)) return 1;
  }
  return 0;
}
