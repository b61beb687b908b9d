// Streaming partitions: the form of the operator that mirrors ExecutionPlan::execute(partition)
// (DataFusion polls every partition's stream on a thread of its own,
// src/prism-query/src/main.rs:45; ParquetExec hands over RecordBatches of 8192 rows,
// src/prism-query/src/providers/table_provider.rs:110-131).
//
// A part never holds more than two chunks of its partition in HBM: the batches that arrive
// are kept (they are moved in) until about a million rows are there, then they are gathered
// column by column into the pinned ring and leave for HBM as one chunk; while those copies
// run, the chunk staged before is aggregated into PARTIAL state, which is all that is kept
// of it.  finish() aggregates the last chunk and merges the states (or, for a partition that
// was one chunk, returns that chunk's result as it is).
#include <cstring>

#include "engine.hpp"

namespace prism {

// ---- child contexts ------------------------------------------------------------------

namespace {
// a child context runs with its parent's settings as they are when it is taken
void copy_options(prism_gpu_ctx* c, prism_gpu_ctx* parent) {
  c->profiling = parent->profiling;
  c->part_min_rows.store(parent->part_min_rows.load());
  c->part_force.store(parent->part_force.load());
  c->part2_compact.store(parent->part2_compact.load());
  c->part_half_blocks.store(parent->part_half_blocks.load());
  c->part_wc.store(parent->part_wc.load());
  c->part_wc_leading.store(parent->part_wc_leading.load());
  c->part2_lines.store(parent->part2_lines.load());
  c->part2_fixed.store(parent->part2_fixed.load());
  c->part2_direct.store(parent->part2_direct.load());
  c->part2_half_blocks.store(parent->part2_half_blocks.load());
  c->part_wc_min_rows.store(parent->part_wc_min_rows.load());
  c->dict_lookup_min_rows.store(parent->dict_lookup_min_rows.load());
  c->contains_windows.store(parent->contains_windows.load());
  c->dict_expand_device.store(parent->dict_expand_device.load());
}
}  // namespace

prism_gpu_ctx* ctx_acquire_child(prism_gpu_ctx* parent) {
  {
    std::lock_guard<std::mutex> g(parent->mu);
    if (!parent->idle_children.empty()) {
      prism_gpu_ctx* c = parent->idle_children.back();
      parent->idle_children.pop_back();
      copy_options(c, parent);
      return c;
    }
  }
  bind(parent);
  auto c = std::make_unique<prism_gpu_ctx>();
  c->device = parent->device;
  c->n_cus = parent->n_cus;
  c->lds_per_block_max = parent->lds_per_block_max;
  c->parent = parent;
  c->stage_bytes = prism_gpu_ctx::kChildStageBytes;
  copy_options(c.get(), parent);
  PRISM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  PRISM_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  PRISM_HIP(hipStreamCreateWithFlags(&c->direct_stream, hipStreamNonBlocking));
  PRISM_HIP(hipEventCreateWithFlags(&c->stream_join, hipEventDisableTiming));
  return c.release();
}

void ctx_release_child(prism_gpu_ctx* child) {
  if (!child) return;
  prism_gpu_ctx* parent = child->parent;
  std::lock_guard<std::mutex> g(parent->mu);
  {
    // what the child measured becomes part of the parent's statistics
    std::lock_guard<std::mutex> gc(child->mu);
    prism_gpu_stats& a = parent->stats;
    const prism_gpu_stats& b = child->stats;
    a.rows_scanned += b.rows_scanned;
    a.rows_selected += b.rows_selected;
    a.groups_out += b.groups_out;
    a.bytes_staged += b.bytes_staged;
    a.staging_ms += b.staging_ms;
    a.chunks_pruned += b.chunks_pruned;
    for (uint32_t i = 0; i < b.n_kernels; ++i) {
      const int k = stat_index(parent, b.kernels[i].name);
      if (k < 0) continue;
      a.kernels[k].launches += b.kernels[i].launches;
      a.kernels[k].total_ms += b.kernels[i].total_ms;
      a.kernels[k].algorithmic_bytes += b.kernels[i].algorithmic_bytes;
    }
    std::memset(&child->stats, 0, sizeof(child->stats));
  }
  parent->idle_children.push_back(child);
}

void ctx_destroy(prism_gpu_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (prism_gpu_ctx* c : ctx->idle_children) ctx_destroy(c);
  ctx->idle_children.clear();
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  (void)sync_copy_streams(ctx);
  for (auto& kv : ctx->free_blocks) (void)hipFree(kv.second);
  for (auto& kv : ctx->live_blocks) (void)hipFree(kv.first);
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  for (auto& p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (int i = 0; i < 2; ++i) {
    if (ctx->stage[i]) (void)hipHostFree(ctx->stage[i]);
    if (ctx->stage_done[i]) (void)hipEventDestroy(ctx->stage_done[i]);
  }
  if (ctx->load_tmp) (void)hipFree(ctx->load_tmp);
  if (ctx->load_landing) (void)hipHostFree(ctx->load_landing);
  if (ctx->result_pinned) (void)hipHostFree(ctx->result_pinned);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->direct_stream) (void)hipStreamDestroy(ctx->direct_stream);
  if (ctx->stream_join) (void)hipEventDestroy(ctx->stream_join);
  delete ctx;
}

// ---- parts ------------------------------------------------------------------------------

namespace {

std::unique_ptr<prism_gpu_table> chunk_table(prism_gpu_ctx* ctx, const std::vector<FieldType>& fields) {
  auto t = std::make_unique<prism_gpu_table>();
  t->ctx = ctx;
  t->fields = fields;
  t->dicts.resize(fields.size());
  t->pooled = true;
  return t;
}

// ---- device-resident state ----------------------------------------------------------------
// A part whose plan is a plain GROUP BY keeps ONE group table in HBM for the whole partition:
// every staged chunk is aggregated on the device (a table of its own, packed from the part's
// key layout) and folded into the part's table; nothing leaves the device before `finish`.
// The layout is fixed from the first chunk's column statistics, widened: a later chunk whose
// keys or NULLs fall outside it makes the part set its table aside as a PARTIAL batch (the
// form chunks always took before) and start a new one, widened further each time.

bool plan_keeps_state(const prism_gpu_plan* plan) {
  return !plan->keys.empty() && !plan->distinct_plan && !plan->mixed && plan->ctx->stream_state.load(std::memory_order_relaxed) != 0;
}

// The statistics the part's keys are packed from: those of the chunks `seen`, widened by `factor`.
std::vector<prism_gpu_col_stats> widened_stats(const prism_gpu_part* part, const std::vector<const Chunk*>& seen, int factor) {
  const prism_gpu_plan* plan = part->plan;
  const prism_gpu_table* t = part->table.get();
  std::vector<prism_gpu_col_stats> out(t->fields.size());
  for (size_t c = 0; c < t->fields.size(); ++c) {
    out[c].min = 0x7fffffffffffffffll;
    out[c].max = -0x7fffffffffffffffll - 1;
    out[c].nulls = 0;
    for (const Chunk* ch : seen) {
      out[c].min = std::min<long long>(out[c].min, ch->stats[c].min);
      out[c].max = std::max<long long>(out[c].max, ch->stats[c].max);
      out[c].nulls += ch->stats[c].null_count;
    }
  }
  std::vector<bool> is_key(t->fields.size(), false), is_time(t->fields.size(), false);
  for (const KeyPlan& k : plan->keys) {
    is_key[(size_t)k.column] = true;
    if (k.kind == PRISM_KEY_DATE_BIN || t->fields[(size_t)k.column].logical == PRISM_COL_TIMESTAMP) is_time[(size_t)k.column] = true;
  }
  for (size_t c = 0; c < t->fields.size(); ++c) {
    const FieldType& f = t->fields[c];
    prism_gpu_col_stats& o = out[c];
    if (!is_key[c]) {
      // values: only whether the column has NULLs shapes the accumulators; any value is covered
      o.min = -0x7fffffffffffffffll - 1;
      o.max = 0x7fffffffffffffffll;
      continue;
    }
    if (o.min > o.max) { o.min = 0; o.max = 0; }  // (a chunk of NULLs only)
    if (f.is_string) {
      // dictionary codes are handed out from 0 in order of appearance
      o.min = 0;
      o.max = std::max<long long>(255, o.max * 4 + 3);
      continue;
    }
    // the type's own limits
    long long lo = -0x7fffffffffffffffll - 1, hi = 0x7fffffffffffffffll;
    if (f.width == 2) { lo = f.is_unsigned ? 0 : -32768; hi = f.is_unsigned ? 65535 : 32767; }
    else if (f.width == 4) { lo = f.is_unsigned ? 0 : -2147483648ll; hi = f.is_unsigned ? 4294967295ll : 2147483647ll; }
    else if (f.is_unsigned) lo = 0;
    const __int128 range = (__int128)o.max - (__int128)o.min + 1;
    // time mostly moves forward within a partition; other keys are as likely to grow either way
    const __int128 below = is_time[c] ? range : range * (factor / 2), above = is_time[c] ? range * factor : range * (factor / 2);
    const __int128 nlo = (__int128)o.min - below, nhi = (__int128)o.max + above;
    o.min = nlo < (__int128)lo ? lo : (long long)nlo;
    o.max = nhi > (__int128)hi ? hi : (long long)nhi;
  }
  return out;
}

bool stats_cover(const prism_gpu_part* part, const Chunk& ch) {
  const prism_gpu_plan* plan = part->plan;
  std::vector<int> used;
  for (const KeyPlan& k : plan->keys) used.push_back(k.column);
  for (const AggPlan& a : plan->aggs) if (a.column >= 0) used.push_back(a.column);
  for (int c : used) {
    const ColStats& cs = ch.stats[(size_t)c];
    const prism_gpu_col_stats& o = part->layout_stats[(size_t)c];
    if (!((cs.min > cs.max || (o.min <= cs.min && cs.max <= o.max)) && (cs.null_count == 0 || o.nulls > 0))) return false;
  }
  return true;
}

void partial_append(prism_gpu_part* part, ArrowArray* state) {
  if (part->plan->mixed) {
    // (List columns do not go into a table: the batches wait on the host for merge_mixed)
    part->mixed_partials.push_back(*state);
    state->release = nullptr;
    part->chunks_done += 1;
    return;
  }
  if (!part->partials) part->partials = chunk_table(part->ctx, part->plan->partial_fields);
  table_append(part->partials.get(), 0, state, true);
  part->chunks_done += 1;
}

// The part's table leaves the device as one PARTIAL batch (a chunk did not fit its layout).
void set_state_aside(prism_gpu_part* part) {
  if (!part->state) return;
  HostTimer timer(part->ctx, "host:state_aside");
  ArrowArray batch;
  std::memset(&batch, 0, sizeof(batch));
  state_emit(part->state, true, &batch);
  state_free(part->state);
  part->state = nullptr;
  partial_append(part, &batch);
  part->widen = std::min(part->widen * 4, 1 << 20);
}

// The key columns that are strings without a host dictionary: their codes are numbered by the
// device anew whenever chunks have arrived, so a group table that holds such codes cannot wait
// for the next chunks -- it is set aside (as strings) right after it was built.
std::vector<int> undictionaried_keys(const prism_gpu_part* part) {
  std::vector<int> out;
  if (!part->table) return out;
  for (const KeyPlan& k : part->plan->keys)
    if (part->table->fields[(size_t)k.column].is_string && !part->table->dicts[(size_t)k.column].alive) out.push_back(k.column);
  return out;
}

// The staged chunks `ids` -> one group table, folded into the part's.
void aggregate_into_state(prism_gpu_part* part, const std::vector<uint32_t>& ids) {
  if (ids.empty()) return;
  prism_gpu_table* t = part->table.get();
  const std::vector<int> renumbered = undictionaried_keys(part);
  if (!renumbered.empty()) {
    set_state_aside(part);  // (it holds codes of the dictionary as it was)
    for (int c : renumbered) table_device_dictionary(t, c);
  }
  table_sync_stats(t);
  std::vector<const Chunk*> chunks;
  for (const Chunk& c : t->chunks)
    for (uint32_t id : ids) if (c.partition_id == id) chunks.push_back(&c);
  if (chunks.size() != ids.size()) fail(PRISM_GPU_EINTERNAL, "a staged chunk is gone");
  bool covered = part->state != nullptr;
  for (const Chunk* ch : chunks) covered = covered && stats_cover(part, *ch);
  if (part->state && !covered) set_state_aside(part);
  if (!part->state) part->layout_stats = widened_stats(part, chunks, part->widen);
  prism_gpu_state* st = aggregate_state(part->plan, t, ids.data(), ids.size(), part->layout_stats.data(), nullptr, part->ctx);
  if (!part->state) {
    part->state = st;
  } else {
    bool merged = false;
    try {
      merged = state_merge_state(part->state, st);
    } catch (...) {
      state_free(st);
      throw;
    }
    if (!merged) {
      // tables of different kinds (the chunks' came out dense, the part's hashed, or the like): both go the PARTIAL way
      set_state_aside(part);
      part->state = st;
    } else {
      state_free(st);
    }
  }
  if (!renumbered.empty()) set_state_aside(part);
  for (uint32_t id : ids) table_drop_partition(t, id);
  part->state_chunks += ids.size();
}

// Aggregates what is staged: every chunk (`all`), or every chunk but the newest, whose copies
// are still under way.
void aggregate_staged(prism_gpu_part* part, bool all) {
  if (part->keeps_state) {
    std::vector<uint32_t> ids = part->staged_ids;
    if (!all && !ids.empty()) ids.pop_back();
    aggregate_into_state(part, ids);
    part->staged_ids.erase(part->staged_ids.begin(), part->staged_ids.begin() + (ptrdiff_t)ids.size());
    return;
  }
  // The staged chunk -> PARTIAL state, kept in part->partials; the chunk's memory goes back to the pool.
  if (!part->staged) return;
  ArrowArray state;
  std::memset(&state, 0, sizeof(state));
  execute_plan_on(part->ctx, part->plan, part->staged.get(), false, true, &state);
  part->staged.reset();
  partial_append(part, &state);
}

// The pending batches -> one chunk on its way to HBM; the chunks before it are aggregated meanwhile.
void stage_pending(prism_gpu_part* part) {
  if (part->pending.empty()) return;
  // (moved: released by the append, also when it fails)
  std::vector<ArrowArray> batches;
  batches.swap(part->pending);
  part->pending_rows = 0;
  std::vector<ArrowArray*> ptrs;
  for (ArrowArray& a : batches) ptrs.push_back(&a);
  if (part->keeps_state) {
    // One table for the whole partition (its dictionaries give a string the same code in every
    // chunk); a chunk is known in it by a number of its own.  Chunks wait in HBM until a few
    // have gathered (option "stream_agg_chunks"): one aggregation over several million rows
    // costs far less than one per chunk, and takes the paths that need that many rows.
    if (!part->table) part->table = chunk_table(part->ctx, part->plan->fields);
    const uint32_t id = part->next_id++;
    const bool had_dictionaries = undictionaried_keys(part).empty();
    table_append_batches(part->table.get(), id, ptrs.data(), ptrs.size(), true);
    part->staged_ids.push_back(id);
    // a key's dictionary gave up during this append: the table built so far is read out through the values it had
    if (had_dictionaries && !undictionaried_keys(part).empty()) set_state_aside(part);
    if (part->staged_ids.size() > (size_t)std::max(1, part->plan->ctx->stream_agg_chunks.load(std::memory_order_relaxed))) aggregate_staged(part, false);
    return;
  }
  auto next = chunk_table(part->ctx, part->plan->fields);
  table_append_batches(next.get(), part->partition, ptrs.data(), ptrs.size(), true);
  aggregate_staged(part, true);
  part->staged = std::move(next);
}

}  // namespace

std::unique_ptr<prism_gpu_part> part_open(prism_gpu_plan* plan, uint32_t partition) {
  auto part = std::make_unique<prism_gpu_part>();
  part->plan = plan;
  part->partition = partition;
  part->ctx = ctx_acquire_child(plan->ctx);
  part->keeps_state = plan_keeps_state(plan);
  return part;
}

void part_push(prism_gpu_part* part, ArrowArray* batch) {
  if (!batch) fail(PRISM_GPU_EINVAL, "batch is null");
  // moved: the part owns the array from here on
  ArrowArray held = *batch;
  batch->release = nullptr;
  if (held.n_children != (int64_t)part->plan->fields.size() || held.offset != 0) {
    const bool offset = held.offset != 0;
    const long long n = (long long)held.n_children;
    if (held.release) held.release(&held);
    if (offset) fail(PRISM_GPU_EUNSUPPORTED, "struct arrays with a non-zero offset are not supported");
    fail(PRISM_GPU_EINVAL, "batch has " + std::to_string(n) + " columns, the plan's input schema has " +
                               std::to_string(part->plan->fields.size()));
  }
  if (held.length == 0) {
    if (held.release) held.release(&held);
    return;
  }
  part->pending_rows += (uint64_t)held.length;
  part->rows_total += (uint64_t)held.length;
  part->pending.push_back(held);
  if (part->pending_rows >= part->plan->ctx->stream_chunk_rows.load(std::memory_order_relaxed)) stage_pending(part);
}

void part_finish(prism_gpu_part* part, ArrowArray* out) {
  bind(part->ctx);
  stage_pending(part);
  const bool partial_out = part->plan->mode == PRISM_MODE_PARTIAL;
  if (part->keeps_state) {
    aggregate_staged(part, true);
    if (part->state && part->chunks_done == 0) {
      // the partition's groups never left the device: they are read once, in the form asked for
      state_emit(part->state, partial_out, out);
      state_free(part->state);
      part->state = nullptr;
      part->state_chunks = 0;
      return;
    }
    set_state_aside(part);
    part->widen = 16;
    part->state_chunks = 0;
  } else {
    if (part->staged && part->chunks_done == 0) {
      // the whole partition is this one chunk: its result is the partition's
      execute_plan_on(part->ctx, part->plan, part->staged.get(), false, false, out);
      part->staged.reset();
      return;
    }
    aggregate_staged(part, true);
  }
  if (part->plan->mixed) {
    std::vector<ArrowArray*> ptrs;
    for (ArrowArray& a : part->mixed_partials) ptrs.push_back(&a);
    merge_mixed(part->plan, ptrs.data(), ptrs.size(), partial_out, out, part->ctx);
    for (ArrowArray& a : part->mixed_partials)
      if (a.release) a.release(&a);
    part->mixed_partials.clear();
    part->chunks_done = 0;
    return;
  }
  if (!part->partials) part->partials = chunk_table(part->ctx, part->plan->partial_fields);
  execute_plan_on(part->ctx, part->plan, part->partials.get(), true, partial_out, out);
  part->partials.reset();
  part->chunks_done = 0;
}

}  // namespace prism

prism_gpu_part::~prism_gpu_part() {
  for (ArrowArray& a : pending)
    if (a.release) a.release(&a);
  pending.clear();
  staged.reset();
  partials.reset();
  for (ArrowArray& a : mixed_partials)
    if (a.release) a.release(&a);
  mixed_partials.clear();
  if (state) prism::state_free(state);
  state = nullptr;
  table.reset();
  if (ctx) prism::ctx_release_child(ctx);
}
