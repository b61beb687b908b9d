// Device helpers shared by the kernels (wave64, gfx950).
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace prism {

// Segment that owns `tile`: the last segment whose tile_base <= tile.  Wave uniform.
__device__ __forceinline__ uint32_t find_seg(const ScanDesc& sd, uint64_t tile) {
  uint32_t lo = 0, hi = sd.n_segs;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (sd.segs[mid].tile_base <= tile) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

__device__ __forceinline__ long long wave_min_i64(long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { long long o = __shfl_down(v, off, 64); v = o < v ? o : v; }
  return v;
}

__device__ __forceinline__ long long wave_max_i64(long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { long long o = __shfl_down(v, off, 64); v = o > v ? o : v; }
  return v;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// 64-bit finaliser (splitmix64's) used to spread packed keys over hash tables.
__device__ __forceinline__ uint64_t hash64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// 32 mixed bits of a 64-bit key from three 32-bit multiplies; the top bits pick a bucket of
// the partition passes (uniform enough for that; hash64 is the hash of the tables).
__device__ __forceinline__ uint32_t bucket_hash32(uint64_t k) {
  uint32_t x = (uint32_t)k ^ ((uint32_t)(k >> 32) * 0x85EBCA6Bu);
  x = (x ^ (x >> 16)) * 0x7FEB352Du;
  x = (x ^ (x >> 15)) * 0x846CA68Bu;
  return x;
}

// Hash of the high word of a two-word key (the group of a COUNT(DISTINCT) pair).
__device__ __forceinline__ uint64_t part_hash_hi(uint64_t hi) { return hash64(hi ^ 0xA0761D6478BD642Full); }

// A mixing bijection of [0, 2^bits), bits < 64: multiplications by odd constants and shifts
// by at least half the width, all invertible modulo 2^bits.  Buckets taken from its top bits
// are as even as a hash's, and -- unlike with a hash -- a value's bucket and the remaining
// low bits together still identify the value, so a record that lies in a bucket need not
// repeat the bits the bucket stands for.
// (Widths of up to 32 bits -- group keys -- are mixed in 32-bit arithmetic: a 64-bit multiply
// is half a dozen instructions on this part, and these run once or twice per row.)
constexpr uint64_t kBijMul1 = 0x9E3779B97F4A7C15ull, kBijMul2 = 0xD6E8FEB86659FD93ull;
constexpr uint32_t kBijMul1w = 0x9E3779B1u, kBijMul2w = 0x85EBCA6Bu;
constexpr uint64_t odd_inverse64(uint64_t a) {
  uint64_t x = a;  // correct to 3 bits; every step doubles that
  for (int i = 0; i < 5; ++i) x *= 2 - a * x;
  return x;
}
__host__ __device__ __forceinline__ uint64_t bij_mix(uint64_t v, int bits) {
  if (bits <= 0) return 0;
  const int s = (bits + 1) / 2;
  if (bits <= 32) {
    const uint32_t m = bits == 32 ? 0xffffffffu : (1u << bits) - 1u;
    uint32_t x = (uint32_t)v;
    x = (x * kBijMul1w) & m;
    x ^= x >> s;
    x = (x * kBijMul2w) & m;
    x ^= x >> s;
    return x;
  }
  const uint64_t mask = (1ull << bits) - 1ull;
  v = (v * kBijMul1) & mask;
  v ^= v >> s;
  v = (v * kBijMul2) & mask;
  v ^= v >> s;
  return v;
}
__host__ __device__ __forceinline__ uint64_t bij_unmix(uint64_t v, int bits) {
  if (bits <= 0) return 0;
  const int s = (bits + 1) / 2;
  if (bits <= 32) {
    const uint32_t m = bits == 32 ? 0xffffffffu : (1u << bits) - 1u;
    uint32_t x = (uint32_t)v;
    x ^= x >> s;
    x = (x * (uint32_t)odd_inverse64(kBijMul2w)) & m;
    x ^= x >> s;
    x = (x * (uint32_t)odd_inverse64(kBijMul1w)) & m;
    return x;
  }
  const uint64_t mask = (1ull << bits) - 1ull;
  v ^= v >> s;
  v = (v * odd_inverse64(kBijMul2)) & mask;
  v ^= v >> s;
  v = (v * odd_inverse64(kBijMul1)) & mask;
  return v;
}

// One fixed-width value widened to int64.
__device__ __forceinline__ long long load_value(const void* base, int width, int is_unsigned, uint64_t row) {
  if (width == 8) return ((const long long*)base)[row];
  if (width == 4) return is_unsigned ? (long long)((const unsigned int*)base)[row] : (long long)((const int*)base)[row];
  return is_unsigned ? (long long)((const unsigned short*)base)[row] : (long long)((const short*)base)[row];
}

__device__ __forceinline__ bool valid_at(const uint64_t* validity, uint64_t row) {
  return validity == nullptr || ((validity[row >> 6] >> (row & 63)) & 1ull);
}

}  // namespace prism
