// extern "C" entry points of include/prism_gpu.h: argument checking, conversion of
// internal errors to status codes, and nothing else.
#include <cstring>

#include "engine.hpp"

namespace {

thread_local std::string g_last_error;

template <class F>
int guarded(F&& f) {
  try {
    f();
    return PRISM_GPU_OK;
  } catch (const prism::Error& e) {
    g_last_error = e.msg;
    return e.code;
  } catch (const std::bad_alloc&) {
    g_last_error = "host memory exhausted";
    return PRISM_GPU_EOOM;
  } catch (const std::exception& e) {
    g_last_error = std::string("internal error: ") + e.what();
    return PRISM_GPU_EINTERNAL;
  } catch (...) {
    g_last_error = "unknown internal error";
    return PRISM_GPU_EINTERNAL;
  }
}

void require(bool ok, const char* what) {
  if (!ok) prism::fail(PRISM_GPU_EINVAL, what);
}

std::unique_ptr<prism_gpu_table> new_table(prism_gpu_ctx* ctx, const std::vector<prism::FieldType>& fields) {
  auto t = std::make_unique<prism_gpu_table>();
  t->ctx = ctx;
  t->fields = fields;
  t->dicts.resize(fields.size());
  return t;
}

}  // namespace

extern "C" {

const char* prism_gpu_last_error(void) { return g_last_error.c_str(); }

int prism_gpu_init(int device_ordinal, prism_gpu_ctx** out) {
  return guarded([&] {
    require(out != nullptr, "out is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) prism::fail(PRISM_GPU_EHIP, "no HIP device is available; this library has no CPU fallback");
    require(device_ordinal >= 0 && device_ordinal < n, "device ordinal out of range");
    auto ctx = std::make_unique<prism_gpu_ctx>();
    ctx->device = device_ordinal;
    PRISM_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    PRISM_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    prism::noalloc_init(device_ordinal);
    PRISM_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    PRISM_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    PRISM_HIP(hipStreamCreateWithFlags(&ctx->direct_stream, hipStreamNonBlocking));
    PRISM_HIP(hipEventCreateWithFlags(&ctx->stream_join, hipEventDisableTiming));
    *out = ctx.release();
  });
}

void prism_gpu_ctx_free(prism_gpu_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  prism::ctx_destroy(ctx);
}

int prism_gpu_table_create(prism_gpu_ctx* ctx, const struct ArrowSchema* schema, prism_gpu_table** out) {
  return guarded([&] {
    require(ctx && schema && out, "null argument");
    *out = new_table(ctx, prism::parse_schema(schema)).release();
  });
}

int prism_gpu_table_append(prism_gpu_table* table, uint32_t partition_id, struct ArrowArray* batch) {
  return guarded([&] {
    if (!table) {
      if (batch && batch->release) batch->release(batch);
      prism::fail(PRISM_GPU_EINVAL, "table is null");
    }
    prism::table_append(table, partition_id, batch, true);
  });
}

int prism_gpu_table_num_rows(const prism_gpu_table* table, uint64_t* out) {
  return guarded([&] {
    require(table && out, "null argument");
    uint64_t n = 0;
    for (auto& c : table->chunks) n += c.rows;
    *out = n;
  });
}

int prism_gpu_table_num_partitions(const prism_gpu_table* table, uint32_t* out) {
  return guarded([&] {
    require(table && out, "null argument");
    std::vector<uint32_t> ids;
    for (auto& c : table->chunks)
      if (std::find(ids.begin(), ids.end(), c.partition_id) == ids.end()) ids.push_back(c.partition_id);
    *out = (uint32_t)ids.size();
  });
}

void prism_gpu_table_free(prism_gpu_table* table) { delete table; }

int prism_gpu_table_set_dictionary(prism_gpu_table* table, int32_t column, const char* const* values,
                                   const uint64_t* lengths, uint32_t n) {
  return guarded([&] {
    require(table && values && lengths, "null argument");
    require(column >= 0 && (size_t)column < table->fields.size(), "column out of range");
    require(table->fields[column].is_string, "only string columns have a dictionary");
    std::lock_guard<std::mutex> g(table->mu);
    prism::ColumnDict& d = table->dicts[column];
    require(d.values.empty(), "the column already has dictionary values");
    for (uint32_t i = 0; i < n; ++i) {
      uint32_t id = d.id_of(values[i], (size_t)lengths[i]);
      if (id != i) prism::fail(PRISM_GPU_EINVAL, "dictionary values must be distinct");
    }
  });
}

int prism_gpu_table_synth(prism_gpu_table* table, uint32_t partition_id, uint64_t seed, uint64_t first_row,
                          uint64_t n_rows, uint64_t total_rows, const prism_gpu_synth_col* cols, uint32_t n_cols) {
  return guarded([&] {
    require(table && cols, "null argument");
    prism::table_synth(table, partition_id, seed, first_row, n_rows, total_rows, cols, n_cols);
  });
}

int prism_gpu_plan_create(prism_gpu_ctx* ctx, const struct ArrowSchema* input_schema, const prism_gpu_plan_desc* desc,
                          prism_gpu_plan** out) {
  return guarded([&] {
    require(input_schema && desc && out, "null argument");
    *out = prism::plan_create(ctx, input_schema, desc).release();
  });
}

int prism_gpu_plan_output_schema(const prism_gpu_plan* plan, struct ArrowSchema* out) {
  return guarded([&] {
    require(plan && out, "null argument");
    prism::export_schema(plan->mode == PRISM_MODE_PARTIAL ? plan->partial_fields : plan->final_fields, out);
  });
}

int prism_gpu_plan_final_schema(const prism_gpu_plan* plan, struct ArrowSchema* out) {
  return guarded([&] {
    require(plan && out, "null argument");
    prism::export_schema(plan->final_fields, out);
  });
}

int prism_gpu_plan_time_range(const prism_gpu_plan* plan, int32_t column, int64_t* start_ms, int64_t* end_ms) {
  return guarded([&] {
    require(plan && start_ms && end_ms, "null argument");
    prism::plan_value_range(plan, column, start_ms, end_ms);
  });
}

void prism_gpu_plan_free(prism_gpu_plan* plan) { delete plan; }

int prism_gpu_execute(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                      struct ArrowArray* out) {
  return guarded([&] {
    require(plan && table && out, "null argument");
    require(plan->ctx != nullptr, "the plan was created without a context: it can be inspected, not run");
    prism::execute_plan(plan, table, partition_ids, n_partitions, false, out);
  });
}

int prism_gpu_execute_noalloc(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                              int64_t* out_values, uint8_t* out_valid, uint64_t out_stride, uint64_t* n_groups) {
  return guarded([&] {
    require(plan && table && n_groups && (out_stride == 0 || (out_values && out_valid)), "null argument");
    require(plan->ctx != nullptr, "the plan was created without a context: it can be inspected, not run");
    prism::execute_noalloc(plan, table, partition_ids, n_partitions, out_values, out_valid, out_stride, n_groups);
  });
}

int prism_gpu_partition_open(prism_gpu_plan* plan, uint32_t partition, prism_gpu_part** out) {
  return guarded([&] {
    require(plan && out, "null argument");
    require(plan->ctx != nullptr, "the plan was created without a context: it can be inspected, not run");
    *out = prism::part_open(plan, partition).release();
  });
}

int prism_gpu_push_batch(prism_gpu_part* part, struct ArrowArray* batch) {
  return guarded([&] {
    if (!part) {
      if (batch && batch->release) batch->release(batch);
      prism::fail(PRISM_GPU_EINVAL, "part is null");
    }
    prism::part_push(part, batch);
  });
}

int prism_gpu_finish(prism_gpu_part* part, struct ArrowArray* out) {
  return guarded([&] {
    require(part && out, "null argument");
    prism::part_finish(part, out);
  });
}

void prism_gpu_part_free(prism_gpu_part* part) { delete part; }

int prism_gpu_merge(prism_gpu_plan* plan, struct ArrowArray* const* states, size_t n_states, struct ArrowArray* out) {
  return guarded([&] {
    require(plan && out && (states || n_states == 0), "null argument");
    require(plan->ctx != nullptr, "the plan was created without a context: it can be inspected, not run");
    if (plan->mixed) {
      prism::merge_mixed(plan, states, n_states, false, out);
      return;
    }
    auto table = new_table(plan->ctx, plan->partial_fields);
    // Page-locked states (the ones this library hands out are) are read by the copy engine where
    // they lie: the appends only queue those copies, one behind the other, and the events behind
    // them are awaited once, before the states go back to the caller (the execution below has
    // waited for the copies anyway).
    struct Pending {
      std::vector<hipEvent_t> events;
      ~Pending() {
        for (hipEvent_t e : events) {
          (void)hipEventSynchronize(e);
          (void)hipEventDestroy(e);
        }
      }
    } pending;
    for (size_t i = 0; i < n_states; ++i) {
      require(states[i] != nullptr, "null state array");
      hipEvent_t e = nullptr;
      prism::table_append(table.get(), (uint32_t)i, states[i], false, &e);
      if (e) pending.events.push_back(e);
    }
    prism::execute_plan(plan, table.get(), nullptr, 0, true, out);
  });
}

int prism_gpu_scan_stats(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                         prism_gpu_col_stats* out) {
  return guarded([&] {
    require(plan && table && out, "null argument");
    require(plan->ctx != nullptr, "the plan was created without a context: it can be inspected, not run");
    prism::scan_stats(plan, table, partition_ids, n_partitions, out);
  });
}

int prism_gpu_aggregate(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                        const prism_gpu_col_stats* stats, prism_gpu_state** out) {
  return guarded([&] {
    require(plan && table && out, "null argument");
    require(plan->ctx != nullptr, "the plan was created without a context: it can be inspected, not run");
    *out = prism::aggregate_state(plan, table, partition_ids, n_partitions, stats);
  });
}

int prism_gpu_state_describe(prism_gpu_state* state, prism_gpu_state_info* out) {
  return guarded([&] {
    require(state && out, "null argument");
    prism::state_describe(state, out);
  });
}

int prism_gpu_state_export(prism_gpu_state* state, void* device_dst, size_t bytes) {
  return guarded([&] {
    require(state && (device_dst || bytes == 0), "null argument");
    prism::state_export(state, device_dst, bytes);
  });
}

int prism_gpu_state_import(prism_gpu_state* state, const void* device_src, size_t bytes) {
  return guarded([&] {
    require(state && device_src, "null argument");
    prism::state_import(state, device_src, bytes);
  });
}

int prism_gpu_state_merge(prism_gpu_state* state, const void* device_src, uint64_t n_groups) {
  return guarded([&] {
    require(state && (device_src || n_groups == 0), "null argument");
    prism::state_merge(state, device_src, n_groups);
  });
}

int prism_gpu_state_finish(prism_gpu_state* state, struct ArrowArray* out) {
  return guarded([&] {
    require(state && out, "null argument");
    prism::state_finish(state, out);
  });
}

void prism_gpu_state_free(prism_gpu_state* state) { prism::state_free(state); }

int prism_gpu_table_dictionary_digest(prism_gpu_table* table, int32_t column, uint64_t* digest, uint32_t* n_values) {
  return guarded([&] {
    require(table && digest && n_values, "null argument");
    require(column >= 0 && (size_t)column < table->fields.size(), "column out of range");
    prism::table_dictionary_digest(table, column, digest, n_values);
  });
}

int prism_gpu_table_append_parquet(prism_gpu_table* table, uint32_t partition_id, const void* file, size_t file_len) {
  return guarded([&] {
    require(table && file, "null argument");
    prism::table_append_parquet(table, partition_id, file, file_len);
  });
}

int prism_gpu_parquet_read(const void* file, size_t file_len, struct ArrowSchema* out_schema, struct ArrowArray* out) {
  return guarded([&] {
    require(file && out, "null argument");
    prism::parquet_read(file, file_len, out_schema, out);
  });
}

// ---- collectives ----

int prism_gpu_comm_unique_id(void* id_out) {
  return guarded([&] {
    require(id_out != nullptr, "null argument");
    prism::comm_unique_id(id_out);
  });
}

int prism_gpu_comm_init(prism_gpu_ctx* ctx, const void* unique_id, int rank, int world, prism_gpu_comm** out) {
  return guarded([&] {
    require(ctx && unique_id && out, "null argument");
    *out = prism::comm_create_rccl(ctx, unique_id, rank, world);
  });
}

int prism_gpu_comm_init_callbacks(prism_gpu_ctx* ctx, const prism_gpu_transport* transport, int rank, int world,
                                  prism_gpu_comm** out) {
  return guarded([&] {
    require(ctx && transport && out, "null argument");
    *out = prism::comm_create_callbacks(ctx, transport, rank, world);
  });
}

void prism_gpu_comm_free(prism_gpu_comm* comm) { prism::comm_destroy(comm); }

int prism_gpu_execute_sharded(prism_gpu_comm* comm, prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids,
                              size_t n_partitions, int root, struct ArrowArray* out, int* has_result) {
  return guarded([&] {
    require(comm && plan && table && out && has_result, "null argument");
    require(plan->ctx != nullptr, "the plan was created without a context: it can be inspected, not run");
    *has_result = 0;
    *has_result = prism::execute_sharded(comm, plan, table, partition_ids, n_partitions, root, out);
  });
}

int prism_gpu_comm_get_stats(prism_gpu_comm* comm, prism_gpu_comm_stats* out) {
  return guarded([&] {
    require(comm && out, "null argument");
    prism::comm_stats(comm, out);
  });
}

int prism_gpu_set_option(prism_gpu_ctx* ctx, const char* name, int64_t value) {
  return guarded([&] {
    require(ctx && name, "null argument");
    const std::string n(name);
    if (n == "part_min_rows") { require(value >= 0, "part_min_rows must not be negative"); ctx->part_min_rows.store((uint64_t)value); }
    else if (n == "part_force") ctx->part_force.store(value != 0 ? 1 : 0);
    else if (n == "part_wc") ctx->part_wc.store(value != 0 ? 1 : 0);
    else if (n == "part2_lines") { require(value >= 0 && value <= 2, "part2_lines must be 0, 1 or 2"); ctx->part2_lines.store((int)value); }
    else if (n == "part_wc_leading") ctx->part_wc_leading.store(value != 0 ? 1 : 0);
    else if (n == "part_wc_min_rows") { require(value >= 0, "part_wc_min_rows must not be negative"); ctx->part_wc_min_rows.store((uint64_t)value); }
    else if (n == "part2_fixed") ctx->part2_fixed.store(value != 0 ? 1 : 0);
    else if (n == "part2_half_blocks") ctx->part2_half_blocks.store(value != 0 ? 1 : 0);
    else if (n == "part2_direct") ctx->part2_direct.store(value != 0 ? 1 : 0);
    else if (n == "part2_compact") ctx->part2_compact.store(value != 0 ? 1 : 0);
    else if (n == "part_half_blocks") ctx->part_half_blocks.store(value != 0 ? 1 : 0);
    else if (n == "contains_windows") ctx->contains_windows.store(value != 0 ? 1 : 0);
    else if (n == "dict_expand_device") ctx->dict_expand_device.store(value != 0);
    else if (n == "dict_lookup_min_rows") ctx->dict_lookup_min_rows.store(value < 0 ? INT64_MAX : value);
    else if (n == "stream_chunk_rows") { require(value >= 1 && value <= (1ll << 28), "stream_chunk_rows must be 1..2^28"); ctx->stream_chunk_rows.store((uint64_t)value); }
    else if (n == "stream_agg_chunks") { require(value >= 1 && value <= 256, "stream_agg_chunks must be 1..256"); ctx->stream_agg_chunks.store((int)value); }
    else if (n == "stream_state") ctx->stream_state.store(value != 0 ? 1 : 0);
    else if (n == "parquet_threads") { require(value >= 1 && value <= 64, "parquet_threads must be 1..64"); ctx->parquet_threads.store((int)value); }
    else if (n == "shard_distinct") ctx->shard_distinct.store(value != 0 ? 1 : 0);
    else if (n == "sharded_fail_phase") ctx->sharded_fail_phase.store((int)value);
    else prism::fail(PRISM_GPU_EINVAL, "unknown option `" + n + "`");
  });
}

int prism_gpu_set_profiling(prism_gpu_ctx* ctx, int enabled) {
  return guarded([&] {
    require(ctx != nullptr, "ctx is null");
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->profiling = enabled == 2 ? 2 : enabled != 0 ? 1 : 0;
  });
}

int prism_gpu_get_stats(prism_gpu_ctx* ctx, prism_gpu_stats* out) {
  return guarded([&] {
    require(ctx && out, "null argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    *out = ctx->stats;
  });
}

int prism_gpu_reset_stats(prism_gpu_ctx* ctx) {
  return guarded([&] {
    require(ctx != nullptr, "ctx is null");
    std::lock_guard<std::mutex> g(ctx->mu);
    std::memset(&ctx->stats, 0, sizeof(ctx->stats));
  });
}

}  // extern "C"
