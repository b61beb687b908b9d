// Microbenchmark: cost of LDS atomics on gfx950 as a function of how many lanes of a
// wave hit the same address.  Prints cycles per wave instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: ds_add_u64, 1: ds_add_u32, 2: ds_max_i64, 3: ds_read_b32, 4: ds_read_b128, 5: ds_add_f64
template <int MODE>
__global__ __launch_bounds__(256) void k(const int* idx, int iters, unsigned long long* out) {
  __shared__ __align__(16) unsigned long long tab[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = 0;
  __syncthreads();
  const int e = idx[threadIdx.x & 63];
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) atomicAdd(&tab[e], 1ull);
    else if (MODE == 1) atomicAdd((unsigned int*)&tab[e], 1u);
    else if (MODE == 2) atomicMax((long long*)&tab[e], (long long)it);
    else if (MODE == 3) acc += ((const unsigned int*)tab)[(e * 2 + it * 64) & 8191];
    else if (MODE == 4) { const uint4 v = *(const uint4*)&tab[(e * 2 + it * 4) & 4094]; acc += v.x + v.w; }
    else atomicAdd((double*)&tab[e], 1.0);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tab[0] + acc;
}

int main() {
  int* d_idx; unsigned long long* d_out;
  CHECK(hipMalloc(&d_idx, 64 * 4));
  CHECK(hipMalloc(&d_out, 4096 * 8));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const int iters = 20000, blocks = 256 * 4;
  struct Pat { const char* name; std::vector<int> idx; };
  std::vector<Pat> pats;
  auto mk = [&](const char* n, auto f) { Pat p; p.name = n; for (int l = 0; l < 64; ++l) p.idx.push_back(f(l)); pats.push_back(p); };
  mk("distinct consecutive", [](int l) { return l; });
  mk("all same", [](int) { return 5; });
  mk("16 same + 48 distinct", [](int l) { return l < 16 ? 5 : 100 + l; });
  mk("4 groups of 16", [](int l) { return (l / 16) * 7 + 3; });
  mk("16 groups of 4", [](int l) { return (l / 4) * 7 + 3; });
  mk("32 pairs", [](int l) { return (l / 2) * 3 + 1; });
  mk("random distinct", [](int l) { return (l * 2654435761u >> 7) & 4095; });
  mk("stride 32 (same bank)", [](int l) { return l * 32; });
  mk("stride 16", [](int l) { return l * 16; });
  const char* modes[] = {"ds_add_u64", "ds_add_u32", "ds_max_i64", "ds_read_b32", "ds_read_b128", "ds_add_f64"};
  for (int m = 0; m < 6; ++m) {
    for (auto& p : pats) {
      CHECK(hipMemcpy(d_idx, p.idx.data(), 64 * 4, hipMemcpyHostToDevice));
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a));
        switch (m) {
          case 0: k<0><<<blocks, 256>>>(d_idx, iters, d_out); break;
          case 1: k<1><<<blocks, 256>>>(d_idx, iters, d_out); break;
          case 2: k<2><<<blocks, 256>>>(d_idx, iters, d_out); break;
          case 3: k<3><<<blocks, 256>>>(d_idx, iters, d_out); break;
          case 4: k<4><<<blocks, 256>>>(d_idx, iters, d_out); break;
          case 5: k<5><<<blocks, 256>>>(d_idx, iters, d_out); break;
        }
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      // wave instructions per CU = blocks/256 CUs * 4 waves * iters
      double insts_per_cu = (double)blocks / 256.0 * 4.0 * iters;
      double cycles = best * 1e-3 * 2.4e9 / insts_per_cu;
      printf("%-12s %-24s %8.3f ms  ~%6.1f cycles/wave-inst/CU (at 2.4 GHz)\n", modes[m], p.name, best, cycles);
    }
  }
  return 0;
}
