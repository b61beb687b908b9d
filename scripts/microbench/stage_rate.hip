// The staging pattern by itself: T threads, each copying pageable host memory into the two
// halves of a pinned ring (memcpy or streaming stores) and sending a half to HBM on a stream
// of its own while the other half fills.  Prints the aggregate GB/s: the ceiling of
// "pageable -> pinned -> HBM" on this host, with nothing else going on.
//   hipcc --offload-arch=gfx950 -O2 -o stage_rate stage_rate.hip -lpthread && ./stage_rate
#include <hip/hip_runtime.h>
#include <emmintrin.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void copy_nt(uint8_t* d, const uint8_t* s, size_t n) {
  for (size_t i = 0; i + 64 <= n; i += 64) {
    const __m128i a = _mm_loadu_si128((const __m128i*)(s + i)), b = _mm_loadu_si128((const __m128i*)(s + i + 16)),
                  c = _mm_loadu_si128((const __m128i*)(s + i + 32)), e = _mm_loadu_si128((const __m128i*)(s + i + 48));
    _mm_stream_si128((__m128i*)(d + i), a);
    _mm_stream_si128((__m128i*)(d + i + 16), b);
    _mm_stream_si128((__m128i*)(d + i + 32), c);
    _mm_stream_si128((__m128i*)(d + i + 48), e);
  }
  _mm_sfence();
}

int main() {
  const size_t per_thread = 1ull << 30, half = 16ull << 20, piece = 64 << 10;
  for (int mode = 0; mode < 3; ++mode) {       // 0: memcpy, 1: streaming stores, 2: no host copy (DMA only)
    for (int T : {1, 4, 8, 16, 32}) {
      std::vector<std::vector<uint8_t>> src(T);
      for (auto& v : src) v.assign(per_thread / 4, 1);  // (re-read four times: what matters is that it is pageable and larger than cache)
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
          CK(hipSetDevice(0));
          hipStream_t st;
          CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
          void* ring[2];
          void* dev;
          hipEvent_t ev[2];
          bool used[2] = {false, false};
          for (int i = 0; i < 2; ++i) { CK(hipHostMalloc(&ring[i], half, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
          CK(hipMalloc(&dev, half));
          const uint8_t* s = src[t].data();
          const size_t n_src = src[t].size();
          size_t at = 0;
          for (size_t done = 0, h = 0; done < per_thread; done += half, h ^= 1) {
            if (used[h]) CK(hipEventSynchronize(ev[h]));
            if (mode != 2)
              for (size_t o = 0; o < half; o += piece) {
                if (at + piece > n_src) at = 0;
                if (mode == 0) std::memcpy((uint8_t*)ring[h] + o, s + at, piece); else copy_nt((uint8_t*)ring[h] + o, s + at, piece);
                at += piece;
              }
            CK(hipMemcpyAsync(dev, ring[h], half, hipMemcpyHostToDevice, st));
            CK(hipEventRecord(ev[h], st));
            used[h] = true;
          }
          CK(hipStreamSynchronize(st));
          for (int i = 0; i < 2; ++i) { CK(hipHostFree(ring[i])); CK(hipEventDestroy(ev[i])); }
          CK(hipFree(dev));
          CK(hipStreamDestroy(st));
        });
      for (auto& x : th) x.join();
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      printf("%-16s %2d threads: %6.1f GB/s\n", mode == 0 ? "memcpy" : mode == 1 ? "streaming stores" : "copy engine only", T,
             (double)per_thread * T / sec / 1e9);
      fflush(stdout);
    }
  }
  return 0;
}
