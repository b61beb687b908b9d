/* Synthetic tables for tests and benchmarks of libprism_gpu.so: partitions generated
 * directly in HBM, so that a billion-row benchmark table does not have to cross PCIe.
 * Not part of the drop-in boundary (include/prism_gpu.h); the symbols live in the same
 * library. */
#ifndef PRISM_GPU_SYNTH_H
#define PRISM_GPU_SYNTH_H

#include "prism_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Sets the dictionary of a Utf8 column whose rows are generated on the device
 * as dictionary codes (prism_gpu_table_synth). */
int prism_gpu_table_set_dictionary(prism_gpu_table* table, int32_t column,
                                   const char* const* values,
                                   const uint64_t* lengths, uint32_t n);

/* Synthetic partitions generated directly in HBM by the generators of
 * prism_synth.h (same functions on host and device, so a CPU checker can
 * regenerate identical rows).  No counterpart in the reference: it stands in
 * for the ClickBench / event data the reference's dev/clickbench.sh stages. */
typedef struct prism_gpu_synth_col {
  int32_t column; /* index in the table schema */
  int32_t kind;   /* prism_synth_kind (prism_synth.h) */
  int64_t p[4];   /* generator parameters */
  const uint32_t* cdf; /* for PRISM_SYNTH_CATEGORICAL: ascending thresholds */
  uint32_t cdf_len;
} prism_gpu_synth_col;

int prism_gpu_table_synth(prism_gpu_table* table, uint32_t partition_id,
                          uint64_t seed, uint64_t first_row, uint64_t n_rows,
                          uint64_t total_rows, const prism_gpu_synth_col* cols,
                          uint32_t n_cols);

#ifdef __cplusplus
}
#endif

#endif /* PRISM_GPU_SYNTH_H */
