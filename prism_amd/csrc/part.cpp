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
  c->part2_fixed.store(parent->part2_fixed.load());
  c->part_wc_min_rows.store(parent->part_wc_min_rows.load());
  c->contains_windows.store(parent->contains_windows.load());
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
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  for (auto& kv : ctx->free_blocks) (void)hipFree(kv.second);
  for (auto& kv : ctx->live_blocks) (void)hipFree(kv.first);
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  for (auto& p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (int i = 0; i < 2; ++i) {
    if (ctx->stage[i]) (void)hipHostFree(ctx->stage[i]);
    if (ctx->stage_done[i]) (void)hipEventDestroy(ctx->stage_done[i]);
  }
  if (ctx->load_tmp) (void)hipFree(ctx->load_tmp);
  if (ctx->result_pinned) (void)hipHostFree(ctx->result_pinned);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
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

// The staged chunk -> PARTIAL state, kept in part->partials; the chunk's memory goes back to the pool.
void aggregate_staged(prism_gpu_part* part) {
  if (!part->staged) return;
  ArrowArray state;
  std::memset(&state, 0, sizeof(state));
  execute_plan_on(part->ctx, part->plan, part->staged.get(), false, true, &state);
  part->staged.reset();
  if (!part->partials) part->partials = chunk_table(part->ctx, part->plan->partial_fields);
  table_append(part->partials.get(), 0, &state, true);
  part->chunks_done += 1;
}

// The pending batches -> one chunk on its way to HBM; the chunk before it is aggregated meanwhile.
void stage_pending(prism_gpu_part* part) {
  if (part->pending.empty()) return;
  auto next = chunk_table(part->ctx, part->plan->fields);
  std::vector<ArrowArray*> ptrs;
  for (ArrowArray& a : part->pending) ptrs.push_back(&a);
  // (moved: released by the append, also when it fails)
  std::vector<ArrowArray> batches;
  batches.swap(part->pending);
  part->pending_rows = 0;
  ptrs.clear();
  for (ArrowArray& a : batches) ptrs.push_back(&a);
  table_append_batches(next.get(), part->partition, ptrs.data(), ptrs.size(), true);
  aggregate_staged(part);
  part->staged = std::move(next);
}

}  // namespace

std::unique_ptr<prism_gpu_part> part_open(prism_gpu_plan* plan, uint32_t partition) {
  if (plan->mixed)
    fail(PRISM_GPU_EUNSUPPORTED, "COUNT(DISTINCT) next to other aggregates has no PARTIAL form: it runs over resident partitions (prism_gpu_execute)");
  auto part = std::make_unique<prism_gpu_part>();
  part->plan = plan;
  part->partition = partition;
  part->ctx = ctx_acquire_child(plan->ctx);
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
  if (part->pending_rows >= kPartChunkRows) stage_pending(part);
}

void part_finish(prism_gpu_part* part, ArrowArray* out) {
  bind(part->ctx);
  stage_pending(part);
  if (part->staged && part->chunks_done == 0) {
    // the whole partition is this one chunk: its result is the partition's
    execute_plan_on(part->ctx, part->plan, part->staged.get(), false, false, out);
    part->staged.reset();
    return;
  }
  aggregate_staged(part);
  if (!part->partials) part->partials = chunk_table(part->ctx, part->plan->partial_fields);
  execute_plan_on(part->ctx, part->plan, part->partials.get(), true, part->plan->mode == PRISM_MODE_PARTIAL, out);
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
  if (ctx) prism::ctx_release_child(ctx);
}
