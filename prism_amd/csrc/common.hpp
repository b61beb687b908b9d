// Shared host-side helpers: error type, HIP call checking, small integer utilities.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/prism_gpu.h"
#include "../../include/prism_gpu_synth.h"

namespace prism {

// Thrown inside the library and converted to a status code + thread-local message
// at the C boundary (no exception ever crosses it).
struct Error {
  int code;
  std::string msg;
};

[[noreturn]] inline void fail(int code, const std::string& msg) { throw Error{code, msg}; }

#define PRISM_HIP(expr)                                                                      \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      ::prism::fail(_e == hipErrorOutOfMemory ? PRISM_GPU_EOOM : PRISM_GPU_EHIP,             \
                    std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" +       \
                        __FILE__ + ":" + std::to_string(__LINE__) + ")");                    \
    }                                                                                        \
  } while (0)

__host__ __device__ inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
inline uint64_t round_up(uint64_t a, uint64_t b) { return ceil_div(a, b) * b; }
inline uint64_t pow2_ceil(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}
inline int bits_for(uint64_t range) {  // bits needed to hold values 0..range
  int b = 0;
  while (b < 64 && (range >> b) != 0) ++b;
  return b;
}

// Rows per tile: the unit of work assignment and of bitmap addressing.  A tile
// never spans two segments, so a selection bitmap is indexed by the "padded row
// id" tile * kTileRows + row_in_tile, 64 words per tile.
constexpr uint32_t kTileRows = 4096;
constexpr uint32_t kTileWords = kTileRows / 64;
constexpr uint32_t kBlock = 256;

}  // namespace prism
