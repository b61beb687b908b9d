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
  PRISM_SYNTH_LOGUNIFORM_ID = 5
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

#endif /* PRISM_SYNTH_H */
