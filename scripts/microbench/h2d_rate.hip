// Host -> HBM copy rate from pinned memory: one copy at a time and several in flight on
// streams of their own, at the sizes the staging ring uses.  Prints GB/s per configuration.
//   hipcc --offload-arch=gfx950 -O2 -o h2d_rate h2d_rate.hip && ./h2d_rate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const size_t total = 4ull << 30;
  const size_t sizes[] = {1ull << 20, 4ull << 20, 8ull << 20, 16ull << 20, 64ull << 20};
  const int stream_counts[] = {1, 2, 4, 8, 16};
  const int max_streams = 16;
  std::vector<hipStream_t> streams(max_streams);
  std::vector<void*> host(max_streams), dev(max_streams);
  for (int i = 0; i < max_streams; ++i) {
    CK(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));
    CK(hipHostMalloc(&host[i], 64ull << 20, hipHostMallocDefault));
    std::memset(host[i], i + 1, 64ull << 20);
    CK(hipMalloc(&dev[i], 64ull << 20));
  }
  for (size_t sz : sizes) {
    for (int ns : stream_counts) {
      const size_t copies = total / sz;
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (size_t c = 0; c < copies; ++c) {
        const int s = (int)(c % ns);
        CK(hipMemcpyAsync(dev[s], host[s], sz, hipMemcpyHostToDevice, streams[s]));
      }
      CK(hipDeviceSynchronize());
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      printf("copy %3zu MiB x %2d streams: %6.1f GB/s\n", sz >> 20, ns, (double)(copies * sz) / sec / 1e9);
    }
  }
  return 0;
}
