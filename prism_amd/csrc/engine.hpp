// Host-side engine objects behind the opaque handles of prism_gpu.h.
#pragma once

#include <atomic>
#include <deque>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "arrow_io.hpp"
#include "context.hpp"
#include "kernels.hpp"

namespace prism {

// ---- resident tables --------------------------------------------------------

// One column of one resident chunk (all pointers are HBM addresses).
struct DeviceColumn {
  void* values = nullptr;        // fixed-width values, or i32 offsets (rebased to 0) of a raw string column
  uint64_t* validity = nullptr;  // nullptr when the chunk has no NULLs in this column
  uint8_t* bytes = nullptr;      // raw string bytes
  uint32_t* codes = nullptr;     // table-wide dictionary codes of a string column
  uint64_t n_bytes = 0;
};

// A run of rows of one partition, appended in one call.
struct Chunk {
  uint32_t partition_id = 0;
  uint64_t rows = 0;
  std::vector<DeviceColumn> cols;
  ColStats* d_stats = nullptr;      // device, one per column
  std::vector<ColStats> stats;      // host copy, valid once stats_ready
  bool stats_ready = false;
};

// Table-wide dictionary of a string column.  It stays alive while the column has
// few distinct values; once it dies the column only has its raw representation.
struct ColumnDict {
  bool alive = true;
  // Not alive, but every chunk has codes all the same: kernels numbered the column's distinct
  // strings (table_device_dictionary) and `values` holds them.  Only GROUP BY keys read such
  // codes; predicates keep to the raw strings.  An append ends it (the next query rebuilds).
  bool device_built = false;
  std::unordered_map<std::string, uint32_t> ids;
  std::vector<std::string> values;
  static constexpr int kFrontBits = 12;
  std::vector<uint32_t> front;  // recent answers of id_of: index into `values` + 1, 0 = none
  int lookup_pause = 0;         // appends for which plain strings are not looked up on the device (after misses in a row)
  int lookup_misses = 0;        // such look-ups that missed, in a row
  uint32_t id_of(const char* s, size_t n);
};

constexpr size_t kMaxDictValues = 1u << 16;

}  // namespace prism

struct prism_gpu_table {
  prism_gpu_ctx* ctx = nullptr;
  bool pooled = false;  // column memory comes from the context's scratch pool (short-lived chunk tables)
  std::vector<prism::FieldType> fields;
  // (a deque: scans keep pointers to chunks, and appending must not move the ones they hold)
  std::deque<prism::Chunk> chunks;
  std::vector<prism::ColumnDict> dicts;  // one per column (used by string columns)
  std::mutex mu;  // appends hold it throughout; scans while they collect their chunks
  // Column buffers of middling size are cut from slabs: obtaining 8-16 MiB from the runtime
  // costs ~0.12 ms a time (and ~0.15 ms to return), a few hundred MiB a third of that, and a
  // loader appends a few such buffers per row group.  A slab goes back once every buffer cut
  // from it has been freed (the chunks of a dropped partition, mostly neighbours in a slab).
  struct Slab {
    uint8_t* base = nullptr;
    size_t size = 0, used = 0;
    uint32_t live = 0;  // buffers cut from it and not yet freed
  };
  std::vector<Slab> slabs;  // (the last one is the one being filled)
  std::mutex slab_mu;
  ~prism_gpu_table();
};

namespace prism {

// Appends one RecordBatch.  `take` = release the array when done (moved), else borrow.
// Bytes of page-locked buffers leave them by copies that may still be under way when the rows are
// in the table's books: with `pending` null the call waits for them before it returns (and before
// it releases a batch it took); otherwise, when such copies were queued, *pending receives an event
// behind them -- the caller keeps the batch (take must be false), awaits and destroys the event, and
// only then releases the batch.  A loader decodes its next row group meanwhile.
void table_append(prism_gpu_table* t, uint32_t partition_id, ArrowArray* batch, bool take, hipEvent_t* pending = nullptr);
// The same for several batches that become one chunk.
void table_append_batches(prism_gpu_table* t, uint32_t partition_id, ArrowArray* const* batches, size_t n_batches, bool take,
                          hipEvent_t* pending = nullptr);
void free_chunk_memory(prism_gpu_table* t, Chunk& ch);
void table_synth(prism_gpu_table* t, uint32_t partition_id, uint64_t seed, uint64_t first_row, uint64_t n_rows,
                 uint64_t total_rows, const prism_gpu_synth_col* cols, uint32_t n_cols);
void table_sync_stats(prism_gpu_table* t);
// Codes for a string column whose host dictionary gave up (more than kMaxDictValues values):
// built on the device over all chunks, exact (no two different strings share a code).  No-op
// when the column has a live dictionary or the device-built one is complete.
void table_device_dictionary(prism_gpu_table* t, int column);
// Frees the chunks of one partition (the caller knows that no kernel still reads them).
void table_drop_partition(prism_gpu_table* t, uint32_t partition_id);

// ---- plans --------------------------------------------------------------------

// Predicate tree after NOT push-down: only leaves, AND and OR remain.
struct PredNode {
  enum Kind { CMP, LIKE, IS_NULL, IS_NOT_NULL, AND, OR } kind;
  int column = -1;
  int cmp_mask = 0;        // CMP: kLt | kEq | kGt
  int64_t lit_i64 = 0;
  std::string lit_str;     // CMP on strings, LIKE pattern
  bool negated = false;    // NOT LIKE
  bool case_insensitive = false;
  int lhs = -1, rhs = -1;
};

struct KeyPlan {
  int kind;     // prism_gpu_key_kind
  int column;
  int64_t stride_ms = 0, origin_ms = 0;
  std::string name;
};

struct AggPlan {
  int func;
  int column;  // -1 for COUNT(*)
  std::string name;
};

}  // namespace prism

struct prism_gpu_plan {
  prism_gpu_ctx* ctx = nullptr;
  std::vector<prism::FieldType> fields;  // input schema
  int mode = PRISM_MODE_SINGLE;
  std::vector<prism::PredNode> pred;
  int pred_root = -1;
  std::vector<prism::KeyPlan> keys;
  std::vector<prism::AggPlan> aggs;
  int order_by = -1;
  bool descending = false;
  int64_t limit = -1;
  uint64_t groups_hint = 0;
  bool distinct_plan = false;
  int distinct_col = -1;
  // COUNT(DISTINCT) next to other aggregates, or over several columns (SINGLE mode): the plan
  // runs as several plans over the same rows -- the other aggregates, and the COUNT(DISTINCT)s
  // of each column -- whose results are joined on the group keys.
  // mixed_map[i] = (sub-plan, aggregate index there) of the plan's i-th aggregate.
  // MIN / MAX over a string column run the same way: a sub-plan that groups by the keys *and*
  // the column (sub_string_col[k] >= 0; its result holds every (group, value) pair once), from
  // which the byte-lexicographic extremes per group are picked on the host.  Sub-plan 0
  // always has one row per group (COUNT(*) alone when the plan has no plain aggregate).
  bool mixed = false;
  std::vector<std::unique_ptr<prism_gpu_plan>> subs;
  std::vector<int> sub_string_col;
  std::vector<std::pair<int, int>> mixed_map;  // (-1, _) never occurs: every aggregate maps to a sub-plan
  std::vector<prism::FieldType> partial_fields, final_fields;
  // capacity the GROUP BY hash table ended up with the last time this plan ran: the next
  // run starts there instead of growing into it again (0: no run yet)
  std::atomic<uint64_t> learned_capacity{0};
  // What earlier runs of this plan found out about its groups.  part_mode 1: the rows mostly
  // missed the workgroups' LDS tables (the streaming kernel said so and stopped), so GROUP BY
  // partitions the rows by key first; part_groups: the number of groups the last partitioned
  // run produced (sizes the buckets and the result list of the next).
  std::atomic<int> part_mode{0};
  std::atomic<uint64_t> part_groups{0};
  std::atomic<uint64_t> part_table_capacity{0};  // capacity that the HBM table behind the LDS tables needed last time
  // The same two for runs over PARTIAL state rows (a merge): a plan that aggregates partitions and
  // then merges their states sees a partition's groups in the one and all groups in the other, and
  // sized by the wrong one its list of groups is too short for the merge every time.
  std::atomic<uint64_t> merge_part_groups{0};
  std::atomic<uint64_t> merge_part_table_capacity{0};
  std::atomic<uint64_t>& learnt_groups(bool merge_input) { return merge_input ? merge_part_groups : part_groups; }
  std::atomic<uint64_t>& learnt_table_capacity(bool merge_input) { return merge_input ? merge_part_table_capacity : part_table_capacity; }
  std::atomic<int> wc_mode{0};  // -1: the histogram-free GROUP BY overflowed a segment or an LDS table on this plan's data
  // COUNT(DISTINCT) partitioned twice: distinct pairs per 1024 rows of the last run (0: not run yet);
  // rows that repeat a pair need no entry of the LDS sets, so fine buckets may hold more rows
  std::atomic<uint32_t> distinct_pairs_per_1024{0};
  std::atomic<int> part2_fixed{0};  // -1: a fine bucket of this plan's data outgrew its fixed region: the second level keeps its histogram
  std::atomic<int> part2_lines{0};  // -1: a time bucket of this plan's data held far more rows than the others: the one-level scatter goes tile by tile
  std::atomic<int> part2_direct{0};  // -1: this plan's rows did not arrive in time order: COUNT(DISTINCT) partitions them by the group key first
};

// One streaming partition (ExecutionPlan::execute(partition) on a thread of its own).  It
// works on a child context (streams, pinned ring, scratch pool of its own), gathers the
// incoming batches into chunks of about 2^20 rows (option "stream_chunk_rows"), stages a chunk while the one
// before it is aggregated into PARTIAL state, and merges the states at the end.
struct prism_gpu_part {
  prism_gpu_plan* plan = nullptr;
  uint32_t partition = 0;
  prism_gpu_ctx* ctx = nullptr;             // child context, returned to the parent when the part is freed
  std::vector<ArrowArray> pending;          // batches moved in, not yet staged
  uint64_t pending_rows = 0, rows_total = 0, chunks_done = 0;
  std::unique_ptr<prism_gpu_table> staged;    // the chunk in flight: staged, not yet aggregated
  std::unique_ptr<prism_gpu_table> partials;  // PARTIAL state of the chunks aggregated so far
  std::vector<ArrowArray> mixed_partials;     // the same for plans whose PARTIAL batches hold List columns (kept on the host)
  // Plain GROUP BY plans keep the partition's groups in HBM instead (part.cpp):
  bool keeps_state = false;
  std::unique_ptr<prism_gpu_table> table;     // the chunks in flight, and the dictionaries of the whole partition
  uint32_t next_id = 0;                       // chunks are told apart by a number of their own
  std::vector<uint32_t> staged_ids;           // staged, not yet aggregated (the last one's copies may be under way)
  prism_gpu_state* state = nullptr;           // the groups of the chunks aggregated so far
  std::vector<prism_gpu_col_stats> layout_stats;  // the column statistics its keys are packed from
  int widen = 16;                             // how far beyond a chunk's statistics the next layout reaches
  uint64_t state_chunks = 0;
  ~prism_gpu_part();
};

namespace prism {

std::unique_ptr<prism_gpu_plan> plan_create(prism_gpu_ctx* ctx, const ArrowSchema* schema,
                                            const prism_gpu_plan_desc* desc);

// [lo, hi] (inclusive; lo > hi: no value) outside which `column` cannot satisfy the plan's WHERE.
void plan_value_range(const prism_gpu_plan* plan, int column, int64_t* lo, int64_t* hi);

// Runs the plan over resident chunks.  merge_input: the table holds PARTIAL state
// batches of this plan and the result is the final rows.
void execute_plan(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                  bool merge_input, ArrowArray* out);

// prism_gpu_merge for plans whose aggregates are of several kinds (COUNT(DISTINCT) next to
// others or over several columns, MIN / MAX over strings): host batches in the plan's PARTIAL schema.
void merge_mixed(prism_gpu_plan* plan, ArrowArray* const* states, size_t n_states, bool emit_partial, ArrowArray* out,
                 prism_gpu_ctx* ctx_override = nullptr);

// The same on a given (child) context, over all chunks of the table; emit_partial: PARTIAL
// state columns whatever the plan's mode (merge_input + emit_partial merges states into states).
void execute_plan_on(prism_gpu_ctx* ctx, prism_gpu_plan* plan, prism_gpu_table* table, bool merge_input, bool emit_partial,
                     ArrowArray* out);

// Streaming partitions (part.cpp).
prism_gpu_ctx* ctx_acquire_child(prism_gpu_ctx* parent);
void ctx_release_child(prism_gpu_ctx* child);
void ctx_destroy(prism_gpu_ctx* ctx);
std::unique_ptr<prism_gpu_part> part_open(prism_gpu_plan* plan, uint32_t partition);
void part_push(prism_gpu_part* part, ArrowArray* batch);
void part_finish(prism_gpu_part* part, ArrowArray* out);

// COUNT(DISTINCT x) GROUP BY k over two-word pairs, sharded over several GPUs: a rank's own
// part ends with the DISTINCT records of every fine bucket (one word each: the bits the bucket
// does not imply), which are dealt to the buckets' owners.  The numbers of buckets follow from
// `sizing_rows`, which the ranks agree on, so that every rank cuts the same buckets.
struct DistinctShardRequest {
  uint64_t sizing_rows;
};
struct DistinctShard {
  bool direct = false;  // the coarse bucket is the top p1 bits of the group word as it is (no first level ran)
  int p1 = 0, p2 = 0, lo_bits = 0, hi_bits = 0;
  uint32_t n_fine = 0, lds_entries = 0;
  Scratch records;  // fine bucket f: records[fine_base[f] .. + fine_count[f])
  Scratch fine;     // the block fine_count / fine_base point into
  uint32_t* fine_count = nullptr;
  uint32_t* fine_base = nullptr;
};

// Device-resident aggregate state (prism_gpu_aggregate and friends).
void scan_stats(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                prism_gpu_col_stats* out);
prism_gpu_state* aggregate_state(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids,
                                 size_t n_partitions, const prism_gpu_col_stats* stats, const DistinctShardRequest* shard = nullptr,
                                 prism_gpu_ctx* ctx_override = nullptr);
// The groups of `src` folded into `dst` (states of one plan built on one context from the same
// column statistics).  False, and nothing done, when the two tables differ in kind (a dense
// and a hashed one, dense tables of different sizes) or hold COUNT(DISTINCT) pairs.
bool state_merge_state(prism_gpu_state* dst, prism_gpu_state* src);
// The state's groups as a RecordBatch: the plan's result rows, or (partial) its PARTIAL state columns.
void state_emit(prism_gpu_state* st, bool partial, ArrowArray* out);
// The shard a state holds (null: the state is a table).
DistinctShard* state_shard(prism_gpu_state* st);
// The owner's part: `records` holds, for each of `world` ranks, the distinct records of this
// rank's n_own fine buckets in order (rank s's start at recv_start[s], host array), counts64 their
// numbers ([world][n_own], device).  Returns a state that counts the pairs per group.
prism_gpu_state* distinct_shard_merge(prism_gpu_state* local, const uint64_t* records, uint64_t n_records, const uint64_t* counts64,
                                      const uint64_t* recv_start, uint32_t n_own, int world, int rank);
void state_describe(prism_gpu_state* st, prism_gpu_state_info* out);
void state_export(prism_gpu_state* st, void* device_dst, size_t bytes);
void state_import(prism_gpu_state* st, const void* device_src, size_t bytes);
void state_merge(prism_gpu_state* st, const void* device_src, uint64_t n_groups);
void state_finish(prism_gpu_state* st, ArrowArray* out);
void state_free(prism_gpu_state* st);

// What a collective needs to see of a state (comm.cpp).  Dense: slot s occupies
// dense_state[s * dense_stride .. + entries].  Hash: the occupied groups in compact form,
// keys[entries] and state[n_slots][entries]; valid until the state is merged into or freed.
struct StateView {
  prism_gpu_plan* plan = nullptr;
  bool dense = false;
  int n_slots = 0;
  uint64_t entries = 0;
  int64_t* dense_state = nullptr;
  uint64_t dense_stride = 0;
  const uint64_t* keys = nullptr;
  const uint64_t* keys_hi = nullptr;  // two-word keys
  const int64_t* state = nullptr;
  std::vector<int> slot_ops;  // SlotOp per slot
  int key_words = 1;
  int distinct = 0;    // 1: the set of (x, group) pairs of a COUNT(DISTINCT) plan, 2: its per-group counts
  int pair_shift = 0;  // distinct == 1, one-word pairs: the group key sits above this bit
};
StateView state_view(prism_gpu_state* st);
// Rows in compact form merged into a state (high words for two-word keys, else null).
void state_merge_rows(prism_gpu_state* st, const uint64_t* src_keys, const uint64_t* src_keys_hi, const int64_t* src_state,
                      uint64_t n_groups);
// COUNT(DISTINCT): the pair set folded into a new state that counts the pairs per group.
prism_gpu_state* state_distinct_outer(prism_gpu_state* inner);
// The groups of a hash state that can still appear in the plan's result: all of them, or,
// with ORDER BY .. LIMIT over a column the device can rank, the leading ones (ties at the
// cut included).  Compact form in fresh blocks: keys[n], state[n_slots][n]; returns n.
uint64_t state_leading_rows(prism_gpu_state* st, struct Scratch& out_keys, struct Scratch& out_keys_hi, struct Scratch& out_state);

// Parquet (parquet.cpp): decode on the host; append row groups as chunks.
void parquet_read(const void* file, size_t len, ArrowSchema* out_schema, ArrowArray* out);
void table_append_parquet(prism_gpu_table* table, uint32_t partition_id, const void* file, size_t len);
// Dictionary indices as a Parquet file holds them: the runs of the RLE / bit-packed hybrid, not
// yet expanded.  The library's own reader hands such a column to table_append (inside a
// dictionary-encoded ArrowArray without an index buffer), and a kernel expands the runs in HBM.
// (PackedRun: kernels.hpp)
struct PackedIndices {
  const uint8_t* bytes;  // (8 bytes of padding behind the last run's values)
  size_t n_bytes;
  const PackedRun* runs;
  size_t n_runs;
};
// Null unless `a` is such an array.
const PackedIndices* parquet_packed_indices(const ArrowArray* a);
// The same expansion on the host (n indices).
void packed_indices_expand(const PackedIndices& pk, int64_t n, std::vector<int32_t>& out);

// FNV-1a over the dictionary values of a string column, in code order.
void table_dictionary_digest(prism_gpu_table* table, int column, uint64_t* digest, uint32_t* n_values);

// ---- collectives between the GPUs of a node (comm.cpp) ----
prism_gpu_comm* comm_create_rccl(prism_gpu_ctx* ctx, const void* unique_id, int rank, int world);
prism_gpu_comm* comm_create_callbacks(prism_gpu_ctx* ctx, const prism_gpu_transport* t, int rank, int world);
void comm_unique_id(void* out);
void comm_destroy(prism_gpu_comm* c);
// 1 when `out` was filled (this rank is `root`), 0 otherwise.
int execute_sharded(prism_gpu_comm* c, prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids,
                    size_t n_partitions, int root, ArrowArray* out);
void comm_stats(prism_gpu_comm* c, prism_gpu_comm_stats* out);

// The entry point that allocates nothing (prism_gpu_execute_noalloc): its device memory is
// a fixed array of the library (noalloc.hip), laid out as descriptors | counters | accumulators.
constexpr size_t kNoAllocDescriptorBytes = 1ull << 20;
constexpr uint64_t kNoAllocMaxAccumulators = 1ull << 20;  // 8-byte accumulators: entries x slots
constexpr size_t kNoAllocBytes = kNoAllocDescriptorBytes + kNoAllocMaxAccumulators * 8;
// Address of the array on `device` (found once per device, at prism_gpu_init) and the lock
// that keeps two calls of one process from using it at once.
void noalloc_init(int device);
void* noalloc_scratch(int device);
std::mutex& noalloc_mutex(int device);
void execute_noalloc(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                     int64_t* out_values, uint8_t* out_valid, uint64_t out_stride, uint64_t* n_groups);

// Host implementation of SQL LIKE, used on dictionary values.
std::vector<LikeToken> like_compile(const std::string& pattern);
bool like_match_host(const std::vector<LikeToken>& tokens, const std::string& s, bool case_insensitive);

}  // namespace prism
