/*
 * Arrow C Data Interface structure definitions.
 *
 * These two structs are the stable ABI published by the Apache Arrow project
 * ("The Arrow C data interface").  arrow-rs (arrow::ffi::FFI_ArrowSchema /
 * FFI_ArrowArray), Arrow C++ and pyarrow (RecordBatch._export_to_c /
 * _import_from_c) all exchange exactly this layout, which is why it is the data
 * currency of the prism_gpu boundary: a Rust ExecutionPlan can hand its
 * RecordBatch to this library without a copy and without linking Arrow here.
 *
 * Ownership rule of the interface: the producer fills the struct and sets
 * `release`; the consumer calls `release` exactly once when done; a struct whose
 * `release` is NULL has been released (or moved from).
 */
#ifndef PRISM_ARROW_C_DATA_H
#define PRISM_ARROW_C_DATA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE

#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4

struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};

struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};

#endif /* ARROW_C_DATA_INTERFACE */

#ifdef __cplusplus
}
#endif

#endif /* PRISM_ARROW_C_DATA_H */
