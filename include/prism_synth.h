/*
 * Counter-based synthetic column generators, identical on host and device.
 *
 * Every value is a pure function of (seed, column, global row), so a table of
 * any size can be generated in HBM by the GPU, regenerated row-for-row by a CPU
 * checker, and split into partitions or across ranks without coordination.
 * Only integer arithmetic is used: results are bit-identical everywhere.
 *
 * The reference stages its benchmark data from files that are not in its tree
 * (dev/clickbench.sh:14-17 copies ./.scratch/testdir); these generators stand
 * in for that data with the shapes its queries need: an Int16 column that is
 * mostly zero (AdvEngineID), a time-ordered Timestamp(ms) column, a skewed
 * low-cardinality dictionary column (service), a uniform Int64 measure, and a
 * skewed high-cardinality Int64 id (UserID).
 */
#ifndef PRISM_SYNTH_H
#define PRISM_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PRISM_HD __host__ __device__ inline
#else
#define PRISM_HD static inline
#endif

typedef enum prism_synth_kind {
  /* p0 = t0 (ms), p1 = span (ms): t0 + floor(global_row * span / total_rows);
   * monotone in the row number.  Requires total_rows * span < 2^64. */
  PRISM_SYNTH_TS_LINEAR = 1,
  /* cdf[0..len-1] ascending u32 thresholds, cdf[len-1] = 0xFFFFFFFF:
   * code = first i with (rand >> 32) < cdf[i], clamped to len-1. */
  PRISM_SYNTH_CATEGORICAL = 2,
  /* p0 = lo, p1 = range (> 0): lo + floor(rand * range / 2^64). */
  PRISM_SYNTH_UNIFORM = 3,
  /* p0 = P(non-zero) * 2^32, p1 = max non-zero value m:
   * (rand >> 32) < p0 ? 1 + floor((rand & 0xFFFFFFFF) * m / 2^32) : 0. */
  PRISM_SYNTH_SPARSE = 4,
  /* p0 = B (1..62), p1 = salt.  A level k is drawn uniformly from [0, B), then a
   * rank uniformly from [2^k, 2^(k+1)): rank frequencies fall off as 1/rank (a
   * discrete log-uniform law, the integer-only cousin of Zipf s = 1) over
   * 2^B - 1 ranks.  The value is a 63-bit hash of the rank, so ids are spread
   * over the whole non-negative Int64 range like ClickBench UserIDs. */
  PRISM_SYNTH_LOGUNIFORM_ID = 5,
  /* Variable-length ASCII strings for Utf8 columns (URL-like, for LIKE '%needle%').
   * cdf[0..len-1] is a table of byte lengths, the quantiles of the wanted length
   * distribution: a row's length is cdf[floor(u * len)] for a uniform u.  The bytes
   * are drawn from [a-z0-9/.:-], eight per hash.  p0 = P(needle) * 2^32: with that
   * probability "google" overwrites 6 bytes at a uniform offset; p1 = P(near miss)
   * * 2^32: with that probability "googl" or "gogle" (5 bytes) does.  Strings shorter
   * than the pattern are left alone.  See prism_synth_url_length / _bytes. */
  PRISM_SYNTH_URL = 6
} prism_synth_kind;

PRISM_HD uint64_t prism_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

PRISM_HD uint64_t prism_synth_column_key(uint64_t seed, uint32_t column) {
  return prism_mix64(seed ^ (0xD6E8FEB86659FD93ull * (uint64_t)(column + 1u)));
}

PRISM_HD uint64_t prism_synth_rand(uint64_t column_key, uint64_t row) {
  return prism_mix64(column_key + row);
}

PRISM_HD uint64_t prism_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

/* One generated value, widened to int64; the caller narrows it to the column's
 * storage type (Int16/Int32/Int64/Timestamp, or a u32 dictionary code). */
PRISM_HD int64_t prism_synth_value(int32_t kind, const int64_t* p,
                                   const uint32_t* cdf, uint32_t cdf_len,
                                   uint64_t column_key, uint64_t global_row,
                                   uint64_t total_rows) {
  switch (kind) {
    case PRISM_SYNTH_TS_LINEAR: {
      uint64_t span = (uint64_t)p[1];
      return p[0] + (int64_t)((global_row * span) / total_rows);
    }
    case PRISM_SYNTH_CATEGORICAL: {
      uint32_t r = (uint32_t)(prism_synth_rand(column_key, global_row) >> 32);
      uint32_t code = 0;
      while (code + 1u < cdf_len && r >= cdf[code]) ++code;
      return (int64_t)code;
    }
    case PRISM_SYNTH_UNIFORM: {
      uint64_t r = prism_synth_rand(column_key, global_row);
      return p[0] + (int64_t)prism_mulhi64(r, (uint64_t)p[1]);
    }
    case PRISM_SYNTH_SPARSE: {
      uint64_t r = prism_synth_rand(column_key, global_row);
      if ((r >> 32) < (uint64_t)p[0])
        return 1 + (int64_t)(((r & 0xFFFFFFFFull) * (uint64_t)p[1]) >> 32);
      return 0;
    }
    case PRISM_SYNTH_LOGUNIFORM_ID: {
      uint64_t r1 = prism_synth_rand(column_key, global_row);
      uint64_t r2 = prism_mix64(r1);
      uint64_t k = prism_mulhi64(r1, (uint64_t)p[0]);
      uint64_t rank = (1ull << k) + (r2 & ((1ull << k) - 1ull));
      return (int64_t)(prism_mix64(rank ^ (uint64_t)p[1]) >> 1);
    }
    default:
      return 0;
  }
}

/* ---- PRISM_SYNTH_URL ---- */

PRISM_HD uint32_t prism_synth_url_length(const uint32_t* lens, uint32_t n_lens,
                                         uint64_t column_key, uint64_t global_row) {
  uint64_t r = prism_synth_rand(column_key, global_row);
  return lens[(uint32_t)(((r >> 32) * (uint64_t)n_lens) >> 32)];
}

PRISM_HD uint8_t prism_synth_url_char(uint32_t byte) {
  uint32_t i = (byte * 40u) >> 8; /* 0..39 */
  if (i < 26u) return (uint8_t)('a' + i);
  if (i < 36u) return (uint8_t)('0' + (i - 26u));
  return i == 36u ? (uint8_t)'/' : i == 37u ? (uint8_t)'.' : i == 38u ? (uint8_t)':' : (uint8_t)'-';
}

/* Writes the `len` bytes of the row's string to dst. */
PRISM_HD void prism_synth_url_bytes(uint8_t* dst, uint32_t len, const int64_t* p,
                                    uint64_t column_key, uint64_t global_row) {
  const uint64_t row_key = prism_mix64(prism_synth_rand(column_key, global_row) ^ 0xA24BAED4963EE407ull);
  for (uint32_t w = 0; w * 8u < len; ++w) {
    uint64_t h = prism_mix64(row_key + w);
    for (uint32_t j = 0; j < 8u && w * 8u + j < len; ++j)
      dst[w * 8u + j] = prism_synth_url_char((uint32_t)(h >> (8u * j)) & 0xffu);
  }
  const uint64_t r = prism_mix64(row_key ^ 0x9FB21C651E98DF25ull);
  const uint64_t hi = r >> 32, lo = r & 0xFFFFFFFFull;
  const uint8_t needle[6] = {'g', 'o', 'o', 'g', 'l', 'e'};
  const uint8_t miss_a[5] = {'g', 'o', 'o', 'g', 'l'};
  const uint8_t miss_b[5] = {'g', 'o', 'g', 'l', 'e'};
  if (hi < (uint64_t)p[0]) {
    if (len >= 6u) {
      uint32_t off = (uint32_t)((lo * (uint64_t)(len - 5u)) >> 32);
      for (uint32_t j = 0; j < 6u; ++j) dst[off + j] = needle[j];
    }
  } else if (hi < (uint64_t)p[0] + (uint64_t)p[1]) {
    if (len >= 5u) {
      uint32_t off = (uint32_t)((lo * (uint64_t)(len - 4u)) >> 32);
      const uint8_t* m = ((r >> 7) & 1ull) ? miss_a : miss_b;
      for (uint32_t j = 0; j < 5u; ++j) dst[off + j] = m[j];
    }
  }
}

#endif /* PRISM_SYNTH_H */
