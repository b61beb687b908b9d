/*
 * prism_gpu — C ABI of the MI355X scan -> filter -> aggregate path that stands
 * in for the DataFusion operators prism-query plans today.
 *
 * The reference (swgillespie/prism) has no FFI of its own: prism-query builds a
 * DataFusion SessionContext (src/prism-query/src/main.rs:82), its TableProvider
 * returns a ParquetExec over one file group per ingest partition
 * (src/prism-query/src/providers/table_provider.rs:94-131), and DataFusion then
 * runs FilterExec / AggregateExec(Partial) / RepartitionExec /
 * AggregateExec(Final) over Arrow RecordBatches.  Each entry point below names
 * the reference interface it replaces.  Data crosses the boundary as Arrow C
 * Data Interface structs (arrow_c_data.h), which arrow-rs exports/imports with
 * arrow::ffi, so a Rust `ExecutionPlan` shim needs nothing but these
 * declarations.
 *
 * Conventions
 *  - every function returns 0 on success or a negative prism_gpu_status; the
 *    message is available from prism_gpu_last_error() (thread local);
 *  - nothing throws or aborts across the boundary;
 *  - an ArrowArray passed as "moved" is released by the library; arrays and
 *    schemas written to `out` parameters are owned by the caller, who calls
 *    their `release`; a result array's buffers may lie in page-locked host
 *    blocks of up to 64 MiB that go back to a small pool on release (they stay
 *    valid after the ctx is destroyed), so copy results that are kept for long;
 *  - a ctx, table or plan may be used from several threads once created; a
 *    prism_gpu_part is used by one thread at a time.
 */
#ifndef PRISM_GPU_H
#define PRISM_GPU_H

#include <stddef.h>
#include <stdint.h>

#include "arrow_c_data.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PRISM_GPU_ABI_VERSION 1u

typedef struct prism_gpu_ctx prism_gpu_ctx;
typedef struct prism_gpu_table prism_gpu_table;
typedef struct prism_gpu_plan prism_gpu_plan;
typedef struct prism_gpu_part prism_gpu_part;
typedef struct prism_gpu_state prism_gpu_state;

/* Error convention: stands in for datafusion::common::Result<T> on this path
 * (src/prism-query/src/providers/table_provider.rs:85,93 map failures to
 * DataFusionError::Internal / ::Execution).  EUNSUPPORTED from plan_create lets
 * the caller keep DataFusion's own CPU operator for that plan. */
typedef enum prism_gpu_status {
  PRISM_GPU_OK = 0,
  PRISM_GPU_EINVAL = -1,
  PRISM_GPU_EUNSUPPORTED = -2,
  PRISM_GPU_EOOM = -3,
  PRISM_GPU_EHIP = -4,
  PRISM_GPU_ERCCL = -5,
  PRISM_GPU_EINTERNAL = -6
} prism_gpu_status;

/* Column types: the closed set of proto ColumnType
 * (proto/common/prism/common/v1/partition.proto:26-38), same numbering, mapped
 * to Arrow types exactly as column_type_to_datafusion does
 * (src/prism-query/src/providers/schema_provider.rs:96-107). */
typedef enum prism_gpu_col_type {
  PRISM_COL_UNSPECIFIED = 0,
  PRISM_COL_INT64 = 1,
  PRISM_COL_UTF8 = 2,
  PRISM_COL_TIMESTAMP = 3, /* Timestamp(Millisecond, None) */
  PRISM_COL_INT16 = 4,
  PRISM_COL_INT32 = 5,
  PRISM_COL_UINT16 = 6,
  PRISM_COL_BINARY = 7
} prism_gpu_col_type;

/* ---- plan description ---------------------------------------------------
 * A plan is the part of DataFusion's physical plan between ParquetExec and the
 * final RecordBatches: FilterExec predicate, AggregateExec group expressions and
 * aggregate expressions, the aggregate mode, and an optional
 * SortExec(fetch)/GlobalLimitExec tail.  It is passed as plain C structs. */

typedef enum prism_gpu_expr_kind {
  PRISM_EXPR_COLUMN = 1,   /* column = index into the input schema            */
  PRISM_EXPR_LIT_I64 = 2,  /* i64 = literal, already coerced to the column type
                              by DataFusion's type coercion                     */
  PRISM_EXPR_LIT_STR = 3,  /* str/str_len = UTF-8 literal (not NUL terminated)  */
  PRISM_EXPR_CMP = 4,      /* op = prism_gpu_cmp_op; lhs column, rhs literal    */
  PRISM_EXPR_AND = 5,      /* Kleene three-valued logic, like arrow's and_kleene */
  PRISM_EXPR_OR = 6,
  PRISM_EXPR_NOT = 7,      /* lhs only                                          */
  PRISM_EXPR_LIKE = 8,     /* lhs Utf8 column, rhs LIT_STR pattern; flags bit 0
                              = NOT LIKE, bit 1 = ILIKE (ASCII case folding)    */
  PRISM_EXPR_IS_NULL = 9,  /* lhs column                                        */
  PRISM_EXPR_IS_NOT_NULL = 10
} prism_gpu_expr_kind;

typedef enum prism_gpu_cmp_op {
  PRISM_CMP_EQ = 1,
  PRISM_CMP_NEQ = 2,
  PRISM_CMP_LT = 3,
  PRISM_CMP_LTEQ = 4,
  PRISM_CMP_GT = 5,
  PRISM_CMP_GTEQ = 6
} prism_gpu_cmp_op;

#define PRISM_LIKE_NEGATED 1
#define PRISM_LIKE_CASE_INSENSITIVE 2

typedef struct prism_gpu_expr {
  int32_t kind;   /* prism_gpu_expr_kind */
  int32_t op;     /* prism_gpu_cmp_op for CMP */
  int32_t lhs;    /* index of the left child in the expr array, -1 if none */
  int32_t rhs;    /* index of the right child, -1 if none */
  int32_t column; /* for COLUMN */
  int32_t flags;
  int64_t i64;    /* for LIT_I64 */
  const char* str;
  uint64_t str_len;
} prism_gpu_expr;

typedef enum prism_gpu_key_kind {
  PRISM_KEY_COLUMN = 1,
  /* date_bin(INTERVAL stride, column, origin) over Timestamp(ms); stride and
   * origin in milliseconds.  DataFusion has no `time_bucket`; this is its
   * spelling of a fixed-width time bucket. */
  PRISM_KEY_DATE_BIN = 2
} prism_gpu_key_kind;

typedef struct prism_gpu_key {
  int32_t kind;
  int32_t column;
  int64_t stride_ms;
  int64_t origin_ms;
  const char* name; /* output field name; NULL = the column's own name */
} prism_gpu_key;

typedef enum prism_gpu_agg_func {
  PRISM_AGG_COUNT_STAR = 1, /* COUNT(*) == COUNT(UInt8(1)): counts rows       */
  PRISM_AGG_COUNT = 2,      /* counts non-NULL values                          */
  PRISM_AGG_SUM = 3,        /* Int* -> Int64, UInt16 -> UInt64, wrapping       */
  PRISM_AGG_MIN = 4,        /* type preserving, NULLs ignored; Utf8 / Binary:
                               byte-lexicographic (columns GROUP BY accepts); its
                               PARTIAL state is the string so far, "MIN(s)[min]" */
  PRISM_AGG_MAX = 5,
  PRISM_AGG_AVG = 6,        /* Float64 = f64 sum of values cast to f64 / count */
  PRISM_AGG_COUNT_DISTINCT = 7 /* Next to other aggregates, or over more than one column: one pass
                                  per kind of aggregate, joined on the group keys; its PARTIAL
                                  state is then a List<T> column "COUNT(DISTINCT x)[count distinct]"
                                  holding the group's distinct values (DataFusion's state of a
                                  distinct count), which prism_gpu_merge and the streaming finish
                                  accept; prism_gpu_execute_sharded does not take such plans */
} prism_gpu_agg_func;

typedef struct prism_gpu_agg {
  int32_t func;
  int32_t column;   /* -1 for COUNT(*) */
  const char* name; /* output field name; NULL = DataFusion's default, e.g.
                       "COUNT(*)", "SUM(value)" */
} prism_gpu_agg;

/* AggregateMode of DataFusion's AggregateExec.  PARTIAL output = group columns
 * followed by state columns named like DataFusion's ("COUNT(*)[count]" Int64,
 * "SUM(x)[sum]", "MAX(x)[max]", "AVG(x)[count]" UInt64 + "AVG(x)[sum]"
 * Float64), so an unmodified AggregateExec(Final) can consume it. */
typedef enum prism_gpu_agg_mode {
  PRISM_MODE_SINGLE = 0,
  PRISM_MODE_PARTIAL = 1
} prism_gpu_agg_mode;

typedef struct prism_gpu_plan_desc {
  uint32_t abi_version; /* PRISM_GPU_ABI_VERSION */
  int32_t mode;         /* prism_gpu_agg_mode */
  const prism_gpu_expr* exprs;
  uint32_t n_exprs;
  int32_t predicate;    /* root of the WHERE tree in exprs, -1 = no filter */
  const prism_gpu_key* keys;
  uint32_t n_keys;
  const prism_gpu_agg* aggs;
  uint32_t n_aggs;
  /* ORDER BY <output column> [DESC] LIMIT k over the final rows (SortExec with
   * fetch + GlobalLimitExec).  order_by indexes the OUTPUT columns (keys first,
   * then aggregates); -1 = none.  Ties are broken arbitrarily, as in the
   * reference.  Ignored in PARTIAL mode. */
  int32_t order_by;
  int32_t descending;
  int64_t limit;        /* -1 = no limit */
  /* Expected number of groups, 0 = unknown (the library then sizes its hash
   * table from the scanned row count). */
  uint64_t groups_hint;
} prism_gpu_plan_desc;

/* ---- context ------------------------------------------------------------ */

/* Replaces SessionContext::new() (src/prism-query/src/main.rs:82) for this
 * path: binds the calling process to one GPU and creates its streams, pinned
 * staging ring and scratch pool. */
int prism_gpu_init(int device_ordinal, prism_gpu_ctx** out);
void prism_gpu_ctx_free(prism_gpu_ctx* ctx);

/* Execution knobs of a context, the counterpart of DataFusion's SessionConfig options
 * (the reference sets none: SessionContext::new() at src/prism-query/src/main.rs:82 runs
 * with the defaults).  Names:
 *   "part_min_rows"  a GROUP BY over fewer selected rows is never partitioned by key
 *                    before it is aggregated (default 4194304)
 *   "part_force"     1: partition whenever the plan's shape allows it, without waiting for
 *                    the streaming kernel to report that its rows scatter (default 0)
 *   "part_half_blocks" 0: the scatter pass of a partitioned GROUP BY always runs one workgroup of
 *                    1024 threads per CU; 1 (default): two of 512 threads where the records allow it
 *   "part_wc"        0: GROUP BY <one-word key> with COUNT(*) alone is always partitioned with a
 *                    histogram pass; 1 (default): without one (hot keys counted in LDS, fixed
 *                    segments per workgroup and bucket), falling back when a segment overflows
 *   "part_wc_min_rows" fewer selected rows never take that path (default 8388608)
 *   "part_wc_leading" 0: that path lists every group; 1 (default): under ORDER BY COUNT(*) DESC LIMIT n
 *                    (n <= 1024) only the groups whose count reaches the n-th largest total among the
 *                    keys counted in LDS -- exact totals, known before the aggregation pass -- are
 *                    listed, and the selection of the first n runs over those few
 *   "part2_fixed"    0: the second partition level of COUNT(DISTINCT) always counts the rows of its
 *                    fine buckets first; 1 (default): fine buckets get regions of fixed size, and the counting
 *                    pass only runs when one overflows
 *   "part2_direct"   0: COUNT(DISTINCT) partitioned twice always sorts its pairs into coarse buckets by the
 *                    group key first; 1 (default): when the group key starts with a time key, the rows are
 *                    taken to arrive in time order -- sorted by those bits already -- and the second level
 *                    reads the columns itself (a plan whose rows turn out not to goes back to two levels)
 *   "part2_lines"    0: that one-level form sorts every tile of rows by fine bucket in LDS and writes the runs;
 *                    1 (default): a workgroup takes a stretch of rows of ONE time bucket (found by bisection),
 *                    keeps a staging line of 8 records per fine bucket in LDS and copies full lines out as 64
 *                    aligned bytes into its own part of the bucket's region (no barrier, no partial cache lines),
 *                    when a time bucket runs through 64 scan tiles and more and there are 1.5 such stretches per CU;
 *                    2: whenever the plan's shape allows it (tests)
 *   "part2_half_blocks" 1: the scatter pass of that one-level form runs two workgroups of 512 threads per CU
 *                    (default 0: one of 1024, which measured faster)
 *   "part2_compact"  0: COUNT(DISTINCT) partitioned twice always keeps two-word records at the
 *                    second level; 1 (default): one-word records when the pairs allow it
 *   "contains_windows" 0: LIKE '%needle%' over a column without a dictionary always uses the
 *                    byte-wise prefilter; 1 (default): needles of 4 bytes and more are found by
 *                    comparing 4-byte windows with the needle's leading words
 *   "dict_expand_device" 0: a string column that arrives dictionary encoded (an Arrow dictionary
 *                    array; the dictionary-encoded chunks of a Parquet file) is expanded into
 *                    offsets and bytes on the host; 1 (default): its indices and dictionaries are
 *                    staged and kernels write the offsets, the bytes and the table-wide codes
 *   "dict_lookup_min_rows" a chunk of at least so many rows (default 65536; negative: never) whose
 *                    string column arrives as plain Utf8 and whose table-wide dictionary exists
 *                    already gets its codes from a kernel that looks every row up in the
 *                    dictionary (a table's first chunk seeds it from a sample of its rows); a chunk
 *                    with a string the dictionary lacks is coded on the host as before, and after
 *                    misses in a row the column is left to the host for 2, 4, then 8 appends
 *   "stream_chunk_rows" rows a streaming partition gathers before it stages them as one chunk
 *                    (default 1048576; 1..2^28)
 *   "stream_agg_chunks" staged chunks a streaming partition that keeps its groups in HBM lets gather
 *                    before it aggregates them in one pass (default 8; 1..256)
 *   "stream_state"   0: a streaming partition (prism_gpu_partition_open) aggregates every chunk of about
 *                    2^20 rows into a PARTIAL batch on the host and merges the batches in finish;
 *                    1 (default): plans with a GROUP BY and no COUNT(DISTINCT) keep one group table
 *                    per partition in HBM, chunk results are folded into it on the device, and a
 *                    PARTIAL batch is only made when a chunk's keys fall outside the table's layout
 *   "parquet_threads" threads that decode the column chunks of one Parquet row group in
 *                    prism_gpu_table_append_parquet (default 4; 1..64)
 *   "shard_distinct" 0: prism_gpu_execute_sharded always builds a COUNT(DISTINCT)'s set of pairs as a
 *                    hash table in HBM before the exchange; 1 (default): over 2^24 rows and more the
 *                    pairs are partitioned twice, as on one GPU, and the buckets' distinct records travel
 *   "sharded_fail_phase" tests only: this context's rank fails on purpose in the given local
 *                    phase (1..5) of prism_gpu_execute_sharded (default 0: never)
 * Unknown names are PRISM_GPU_EINVAL. */
int prism_gpu_set_option(prism_gpu_ctx* ctx, const char* name, int64_t value);

/* Thread-local message for the last failing call on this thread. */
const char* prism_gpu_last_error(void);

/* ---- columnar chunk loader ----------------------------------------------
 * Replaces what ParquetExec hands downstream (table_provider.rs:110-131):
 * decoded Arrow columns of one partition, staged through pinned host memory
 * into HBM and kept resident.  `schema` is the table schema the reference
 * builds in PrismSchemaProvider::table (schema_provider.rs:73-79). */
int prism_gpu_table_create(prism_gpu_ctx* ctx, const struct ArrowSchema* schema,
                           prism_gpu_table** out);
/* Appends one RecordBatch (a struct-typed ArrowArray matching the table schema;
 * Utf8 / Binary columns may arrive dictionary encoded, with Int32 indices -- the only index
 * width arrow-rs produces for DataFusion's Dictionary(Int32, Utf8) -- and a dictionary
 * without NULL entries; NULL rows are told by the indices' validity) to partition
 * `partition_id`.
 * The array is moved: the library releases it once the bytes are staged. */
int prism_gpu_table_append(prism_gpu_table* table, uint32_t partition_id,
                           struct ArrowArray* batch);
int prism_gpu_table_num_rows(const prism_gpu_table* table, uint64_t* out);
int prism_gpu_table_num_partitions(const prism_gpu_table* table, uint32_t* out);
void prism_gpu_table_free(prism_gpu_table* table);

/* Parquet files decoded by the library itself: what ParquetExec does for the reference
 * (table_provider.rs:110-131) on files shaped like prism-ingest's (ingest.rs:98-101: data
 * page v2, dictionary pages with PLAIN fallback, Snappy).  `file` is the whole object in
 * host memory (what one object_store GET returns).  Covered: flat schemas; INT32 / INT64 /
 * BYTE_ARRAY columns (Int16 / UInt16 / Int32 / Int64 / Timestamp / Utf8 / Binary), required
 * or optional; data pages v1 and v2; PLAIN and dictionary encodings; UNCOMPRESSED and SNAPPY;
 * anything else is PRISM_GPU_EUNSUPPORTED and names the column.
 *
 * prism_gpu_table_append_parquet appends every row group as a chunk of `partition_id`;
 * columns are matched to the table's by name (a table column the file lacks reads as NULLs;
 * Binary / Utf8 and Int64 / Timestamp stand in for each other).
 * prism_gpu_parquet_read only decodes, into one host RecordBatch with the file's own schema
 * (no GPU involved; out_schema may be NULL). */
int prism_gpu_table_append_parquet(prism_gpu_table* table, uint32_t partition_id,
                                   const void* file, size_t file_len);
int prism_gpu_parquet_read(const void* file, size_t file_len, struct ArrowSchema* out_schema,
                           struct ArrowArray* out);

/* ---- plans ---------------------------------------------------------------
 * Replaces the physical planning DataFusion does after TableProvider::scan
 * (table_provider.rs:73-132): FilterExec + AggregateExec (+ SortExec/Limit). */
int prism_gpu_plan_create(prism_gpu_ctx* ctx,
                          const struct ArrowSchema* input_schema,
                          const prism_gpu_plan_desc* desc, prism_gpu_plan** out);
/* ExecutionPlan::schema(): the schema of what execute/finish will return. */
int prism_gpu_plan_output_schema(const prism_gpu_plan* plan,
                                 struct ArrowSchema* out);
/* The range of `column` (an integer or Timestamp(ms) input column) outside which no row can
 * pass the plan's WHERE, inclusive on both sides; INT64_MIN / INT64_MAX for an open side,
 * start > end when no value can pass.  It is what the reference's TableProvider::scan would
 * put into GetTablePartitionsRequest.time_range (always None today,
 * src/prism-query/src/providers/table_provider.rs:86-93) so that the catalogue only lists
 * partitions whose Partition.time_range (proto/common/prism/common/v1/partition.proto:5-14)
 * overlaps it.  `ctx` may be NULL in prism_gpu_plan_create for a plan that is only inspected
 * (schemas, this range) and never run. */
int prism_gpu_plan_time_range(const prism_gpu_plan* plan, int32_t column,
                              int64_t* start_ms, int64_t* end_ms);
void prism_gpu_plan_free(prism_gpu_plan* plan);

/* ExecutionPlan::execute + collect over HBM-resident partitions.
 * partition_ids == NULL or n_partitions == 0 scans every partition, as the
 * reference does (time_range: None, table_provider.rs:90).  The result is one
 * struct-typed ArrowArray (a record batch) in host memory. */
int prism_gpu_execute(prism_gpu_plan* plan, prism_gpu_table* table,
                      const uint32_t* partition_ids, size_t n_partitions,
                      struct ArrowArray* out);

/* prism_gpu_execute for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in
 * the reference.  The call allocates nothing on the device (no hipMalloc / hipFree, no pool):
 * the group table, the scan descriptors and the counters live in a fixed 9 MiB device array
 * inside the library, which every GPU the library runs on holds until the process ends, and
 * the result goes to buffers of the caller instead of an ArrowArray.
 *
 * Limits (PRISM_GPU_EUNSUPPORTED beyond them, with a message that says which): a GROUP BY
 * plan in SINGLE mode, no COUNT(DISTINCT), no ORDER BY / LIMIT; a WHERE only of the kind the
 * aggregation kernel evaluates itself (up to four comparisons / IS NULL tests joined by one
 * of AND / OR); a packed key of at most 22 bits over the scanned partitions, and
 * (2^key_bits + 1) x accumulators <= 1048576 (8 MiB); at most ~2700 partition chunks x columns.
 *
 * Output: out_values is [n_columns][out_stride] 8-byte values and out_valid the same shape in
 * bytes (1 = not NULL), columns in the order of prism_gpu_plan_final_schema; Int64 / Timestamp
 * / counts as int64, AVG as the bit pattern of a double, string keys as the code of the value
 * in the table's dictionary.  Both may be device memory or page-locked host memory.
 * *n_groups (host memory) receives the number of rows; if it exceeds out_stride the call fails
 * with PRISM_GPU_EINVAL and nothing was written past the buffers.
 *
 * The array belongs to the process (it is part of the library's loaded code object) and to
 * the device: every process has its own copy per GPU, and the calls of one process on one GPU
 * are serialised by the library.  Different processes never share it.  The array keeps the
 * last call's contents; each call clears what it needs. */
int prism_gpu_execute_noalloc(prism_gpu_plan* plan, prism_gpu_table* table,
                              const uint32_t* partition_ids, size_t n_partitions,
                              int64_t* out_values, uint8_t* out_valid, uint64_t out_stride,
                              uint64_t* n_groups);

/* Streaming form, one prism_gpu_part per DataFusion partition
 * (ExecutionPlan::execute(partition, ..) polled on its own thread):
 * push_batch moves one input RecordBatch in, finish returns the partition's
 * output (PARTIAL state or SINGLE result).
 *
 * push_batch takes the ArrowArray by the Arrow move rule: the struct is copied and
 * batch->release is set to NULL at once; the library calls the release callback when the
 * rows have been copied to the device, which happens once about 2^20 rows have gathered (or
 * at finish), not inside the push.  What the struct points at (children, buffers,
 * private_data) therefore has to stay valid until that release, as for any Arrow consumer.
 * Each gathered chunk is copied through the partition's own page-locked ring on its own
 * stream, aggregated while the next chunk gathers, and the chunk states are merged at
 * finish.  Different parts may be driven from different threads at the same time; one part
 * is driven by one thread at a time. */
int prism_gpu_partition_open(prism_gpu_plan* plan, uint32_t partition,
                             prism_gpu_part** out);
int prism_gpu_push_batch(prism_gpu_part* part, struct ArrowArray* batch);
int prism_gpu_finish(prism_gpu_part* part, struct ArrowArray* out);
void prism_gpu_part_free(prism_gpu_part* part);

/* AggregateExec(Final): merges PARTIAL outputs of `plan` (from any number of
 * partitions, GPUs or processes) into final rows, applying the plan's
 * ORDER BY / LIMIT.  The state arrays are borrowed, not released. */
int prism_gpu_merge(prism_gpu_plan* plan, struct ArrowArray* const* states,
                    size_t n_states, struct ArrowArray* out);
/* Schema of prism_gpu_merge's output (== the SINGLE-mode output schema). */
int prism_gpu_plan_final_schema(const prism_gpu_plan* plan,
                                struct ArrowSchema* out);

/* ---- device-resident aggregate state: the exchange step between GPUs ------
 * The reference makes every partition an independent unit that produces a Partial
 * aggregate (one file group per partition, table_provider.rs:94-109) and lets
 * DataFusion's RepartitionExec(Hash) + AggregateExec(FinalPartitioned) merge them.
 * Between GPUs of one node the same merge is done on the group tables themselves,
 * in HBM, with a collective in between (RCCL all-reduce for small, densely indexed
 * group spaces; all-gather / all-to-all of (key, state) rows otherwise), instead of
 * exporting Arrow state batches through the host:
 *
 *   prism_gpu_scan_stats     per-column min / max / NULL count of this rank's scan;
 *                            the caller reduces them over the ranks (MIN, MAX, SUM)
 *   prism_gpu_aggregate      scan -> filter -> GROUP BY with the keys packed from
 *                            the agreed statistics, so equal groups have equal packed
 *                            keys (and equal dense indices) on every rank; the group
 *                            table stays in HBM behind the returned handle
 *   prism_gpu_state_describe what the table looks like and how each 8-byte
 *                            accumulator ("slot") merges
 *   prism_gpu_state_export   dense table: the slot arrays, [n_slots][entries] int64;
 *                            hash table: the occupied groups, [entries] packed keys
 *                            followed by [n_slots][entries] accumulators;
 *                            copied into a caller-provided DEVICE buffer
 *   prism_gpu_state_import   dense table: replaces the slot arrays (after an
 *                            all-reduce of the exported buffer)
 *   prism_gpu_state_merge    hash table: merges groups exported by another rank
 *   prism_gpu_state_finish   final rows (ORDER BY / LIMIT applied), as prism_gpu_merge
 *
 * Supported for plans with at least one GROUP BY key (otherwise PRISM_GPU_EUNSUPPORTED,
 * and the caller exchanges PARTIAL batches through prism_gpu_merge instead).  States of
 * COUNT(DISTINCT) plans (the set of (value, group) pairs) and of keys wider than 64 bits can
 * be finished but have no export layout: describe / export / import / merge return
 * PRISM_GPU_EUNSUPPORTED for them; prism_gpu_execute_sharded exchanges them itself.  String keys travel
 * as dictionary codes: the ranks must hold the same dictionary for the column
 * (compare prism_gpu_table_dictionary_digest).  The table must outlive the state.
 * Device buffers are read and written on the library's own stream, and the calls return
 * once that work is done: the caller makes sure that nothing it queued on other streams
 * (an initialising kernel, a collective) is still pending on the buffer when it calls. */
typedef struct prism_gpu_col_stats {
  int64_t min, max; /* min > max: no non-NULL value */
  uint64_t nulls;
} prism_gpu_col_stats;

enum prism_gpu_slot_merge {
  PRISM_SLOT_SUM_I64 = 0, /* wrapping */
  PRISM_SLOT_SUM_F64 = 1,
  PRISM_SLOT_MIN_I64 = 2,
  PRISM_SLOT_MAX_I64 = 3
};

#define PRISM_GPU_MAX_SLOTS 24
typedef struct prism_gpu_state_info {
  int32_t dense;     /* 1: entries = size of the key space, every rank has the same */
  int32_t n_slots;
  uint64_t entries;  /* dense: entries per slot array; hash: occupied groups */
  uint64_t export_bytes;
  int32_t slot_merge[PRISM_GPU_MAX_SLOTS]; /* prism_gpu_slot_merge */
} prism_gpu_state_info;

/* `out` has one entry per column of the plan's input schema. */
int prism_gpu_scan_stats(prism_gpu_plan* plan, prism_gpu_table* table,
                         const uint32_t* partition_ids, size_t n_partitions,
                         prism_gpu_col_stats* out);
/* `stats` (one per input column) may be NULL: the local statistics are used. */
int prism_gpu_aggregate(prism_gpu_plan* plan, prism_gpu_table* table,
                        const uint32_t* partition_ids, size_t n_partitions,
                        const prism_gpu_col_stats* stats, prism_gpu_state** out);
int prism_gpu_state_describe(prism_gpu_state* state, prism_gpu_state_info* out);
int prism_gpu_state_export(prism_gpu_state* state, void* device_dst, size_t bytes);
int prism_gpu_state_import(prism_gpu_state* state, const void* device_src, size_t bytes);
int prism_gpu_state_merge(prism_gpu_state* state, const void* device_src, uint64_t n_groups);
int prism_gpu_state_finish(prism_gpu_state* state, struct ArrowArray* out);
void prism_gpu_state_free(prism_gpu_state* state);
/* FNV-1a over the dictionary values of a string column, in code order. */
int prism_gpu_table_dictionary_digest(prism_gpu_table* table, int32_t column,
                                      uint64_t* digest, uint32_t* n_values);

/* ---- the exchange inside the library: one call per rank -------------------
 * The same merge as above, with the collectives issued by the library: the ranks (one
 * process per GPU) join a communicator, and prism_gpu_execute_sharded runs
 *
 *   scan statistics of the rank's partitions  -> one small all-gather (the union's
 *                                                 min / max / NULL counts, dictionary digests)
 *   scan -> filter -> GROUP BY of the partitions (keys packed from the agreed statistics)
 *   small key space (dense table):  reduce of the accumulator arrays onto `root`, in place
 *                                   in HBM, one collective per run of slots that merge alike
 *   large key space (hash table):   rows dealt to owner ranks by a hash of the packed key
 *                                   (HIP kernels: count per owner, scatter), all-to-all,
 *                                   merge at the owner, the owner's groups -- cut to the
 *                                   leading ones when the plan has ORDER BY .. LIMIT -- sent
 *                                   to `root`, which merges the disjoint slices
 *   final rows on `root`            (ORDER BY / LIMIT applied)
 *
 * which is RepartitionExec(Hash) + AggregateExec(FinalPartitioned) + the coalescing of the
 * output partitions in DataFusion's plan for the same query.  COUNT(DISTINCT x) plans deal
 * the (x, group) pairs to the owner of the *group*, which merges them into its set, counts
 * them per group and passes (group, count) rows on; keys wider than 64 bits travel as two
 * words.  Every rank must make the same sequence of calls with the same plan.  Plans
 * without GROUP BY return PRISM_GPU_EUNSUPPORTED on every rank before any collective.
 *
 * Two transports.  prism_gpu_comm_init joins an RCCL communicator (xGMI between the GPUs of
 * a node; librccl is loaded on first use): rank 0 calls prism_gpu_comm_unique_id and hands
 * the 128 bytes to the other ranks out of band.  prism_gpu_comm_init_callbacks runs the
 * same exchange over three functions of the caller, on page-locked host copies of the
 * buffers: for hosts with a fabric of their own, and for tests in which several ranks
 * share one GPU (which RCCL does not allow).  Failures of the transport are reported as
 * PRISM_GPU_ERCCL; an RCCL communicator whose collective failed is aborted (which releases
 * peers still inside it) and refuses further calls.
 *
 * A rank's own failure (out of memory, a table overflow, a bad table) does not strand the
 * others: every small all-gather of the exchange carries a status word per rank, and one
 * is placed in front of each bulk collective, so when one rank fails every rank returns an
 * error at the same point -- the failing rank its own, the others that rank's status code
 * with a message naming it.  String columns that travel as dictionary codes (GROUP BY keys
 * and the column of a COUNT(DISTINCT)) must have equal dictionaries on all ranks, else
 * PRISM_GPU_EINVAL on every rank. */
typedef struct prism_gpu_comm prism_gpu_comm;
#define PRISM_GPU_COMM_ID_BYTES 128

enum prism_gpu_reduce_op { PRISM_REDUCE_SUM = 0, PRISM_REDUCE_MIN = 1, PRISM_REDUCE_MAX = 2 };
enum prism_gpu_reduce_type { PRISM_REDUCE_I64 = 0, PRISM_REDUCE_F64 = 1 };

typedef struct prism_gpu_transport {
  void* user;
  /* Every rank contributes `bytes` bytes; `all` receives world * bytes, in rank order. */
  int (*allgather)(void* user, const void* mine, void* all, size_t bytes);
  /* Element-wise reduction of n 8-byte values (prism_gpu_reduce_type) with
   * prism_gpu_reduce_op, in place; the result is needed on `root` only. */
  int (*reduce)(void* user, void* buf, size_t n, int type, int op, int root);
  /* 8-byte words: rank r receives send_counts[r] words of this rank's `send` (runs in rank
   * order) and this rank receives recv_counts[s] words from rank s into `recv` (runs in
   * rank order). */
  int (*alltoallv)(void* user, const uint64_t* send, const uint64_t* send_counts,
                   uint64_t* recv, const uint64_t* recv_counts);
  /* Non-zero return values fail the call in progress with PRISM_GPU_ERCCL. */
} prism_gpu_transport;

typedef struct prism_gpu_comm_stats {
  uint64_t collectives;   /* issued by this rank */
  uint64_t bytes_sent;    /* payload this rank handed to the transport */
  double exchange_ms;     /* host wall clock spent between the aggregate and the final rows */
} prism_gpu_comm_stats;

int prism_gpu_comm_unique_id(void* id_out /* PRISM_GPU_COMM_ID_BYTES */);
int prism_gpu_comm_init(prism_gpu_ctx* ctx, const void* unique_id, int rank, int world,
                        prism_gpu_comm** out);
int prism_gpu_comm_init_callbacks(prism_gpu_ctx* ctx, const prism_gpu_transport* transport,
                                  int rank, int world, prism_gpu_comm** out);
void prism_gpu_comm_free(prism_gpu_comm* comm);
/* *has_result = 1 on `root` (out is a final RecordBatch the caller releases), 0 elsewhere. */
int prism_gpu_execute_sharded(prism_gpu_comm* comm, prism_gpu_plan* plan, prism_gpu_table* table,
                              const uint32_t* partition_ids, size_t n_partitions, int root,
                              struct ArrowArray* out, int* has_result);
int prism_gpu_comm_get_stats(prism_gpu_comm* comm, prism_gpu_comm_stats* out);

/* ---- measurement ---------------------------------------------------------
 * Per-kernel device time measured with HIP events on the stream the kernels run
 * on, accumulated since the last reset, for roofline reporting.  Profiling is
 * off by default (events cost a little). */
#define PRISM_GPU_MAX_KERNEL_STATS 24
typedef struct prism_gpu_kernel_stat {
  char name[48];
  uint64_t launches;
  double total_ms;
  uint64_t algorithmic_bytes; /* bytes the launches had to move at minimum */
} prism_gpu_kernel_stat;

typedef struct prism_gpu_stats {
  uint64_t rows_scanned;
  uint64_t rows_selected;
  uint64_t groups_out;
  uint64_t bytes_staged; /* host -> HBM */
  double staging_ms;
  uint32_t n_kernels;
  prism_gpu_kernel_stat kernels[PRISM_GPU_MAX_KERNEL_STATS];
  uint64_t chunks_pruned; /* chunks skipped because their min/max rule the predicate out */
} prism_gpu_stats;

/* 0: launches are counted, not timed.  1: every timed launch is bracketed by events on the library's
 * stream (total_ms).  2: the same, except for the separate passes of a partitioned GROUP BY /
 * COUNT(DISTINCT) -- the launch as a whole is still timed under "group_by"; an event between two
 * passes delays the second by some microseconds. */
int prism_gpu_set_profiling(prism_gpu_ctx* ctx, int enabled);
int prism_gpu_get_stats(prism_gpu_ctx* ctx, prism_gpu_stats* out);
int prism_gpu_reset_stats(prism_gpu_ctx* ctx);

#ifdef __cplusplus
}
#endif

#endif /* PRISM_GPU_H */
