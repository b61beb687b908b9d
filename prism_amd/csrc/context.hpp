// Per-process GPU context: device binding, streams, a size-bucketed scratch pool
// in HBM, a pinned host staging ring, and HIP-event kernel timing.
#pragma once

#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

namespace prism {

// Pinned host blocks that result batches are built in.  A block is leased to one result,
// whose Arrow buffers point into it, and comes back when the caller releases the batch
// (possibly after the context is gone: the pool lives as long as any lease).
struct PinnedPool {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;
  ~PinnedPool() {
    for (auto& kv : free_blocks) (void)hipHostFree(kv.second);
  }
};

}  // namespace prism

struct prism_gpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;       // compute
  hipStream_t copy_stream = nullptr;  // host <-> HBM staging: the pinned ring's copies and the loader's kernels
  // Large copies that read the caller's own page-locked buffers, and the statistics kernels over
  // what they bring: apart, so that nothing waiting for a small copy waits for them.  Work on it
  // may wait for copy_stream (through stream_join), never the other way round within an append.
  hipStream_t direct_stream = nullptr;
  hipEvent_t stream_join = nullptr;  // (under stage_mu)
  bool direct_seen = false;          // this context has sent such copies (under stage_mu)
  int n_cus = 256;
  size_t lds_per_block_max = 65536;

  std::mutex mu;  // guards the pool, the staging ring and the stats

  // ---- scratch pool: freed blocks are kept and reused by size class ----
  std::multimap<size_t, void*> free_blocks;
  std::map<void*, size_t> live_blocks;
  size_t pool_bytes = 0;

  // ---- pinned staging ring (two halves, double buffered against the copy) ----
  // Allocated on first use.  A half is refilled only after the event recorded behind the
  // copy that read it has completed; stage_mu makes staging one thread at a time per context
  // (threads that stream partitions concurrently have a context each: child contexts).
  static constexpr size_t kStageBytes = 64ull << 20, kChildStageBytes = 32ull << 20;
  std::mutex stage_mu;
  size_t stage_bytes = kStageBytes;
  void* stage[2] = {nullptr, nullptr};
  hipEvent_t stage_done[2] = {nullptr, nullptr};
  bool stage_used[2] = {false, false};
  int stage_cur = -1;     // the half being filled
  size_t stage_fill = 0;  // ... and how far
  // device scratch of the loader's kernels (under stage_mu; grown on demand, reused by the next
  // append: every use is ordered on copy_stream)
  void* load_tmp = nullptr;
  size_t load_tmp_bytes = 0;
  // where the loader reads a few bytes back (page-locked: a copy into ordinary memory takes a
  // detour through the runtime's own staging; under stage_mu, allocated with the ring)
  void* load_landing = nullptr;

  // ---- child contexts: same device, streams / ring / pool / statistics of their own ----
  prism_gpu_ctx* parent = nullptr;
  std::vector<prism_gpu_ctx*> idle_children;  // (parent only, under mu) kept warm between partitions

  // ---- pinned landing buffer for results read back from HBM ----
  std::mutex result_mu;
  void* result_pinned = nullptr;
  size_t result_pinned_bytes = 0;
  std::shared_ptr<prism::PinnedPool> result_pool = std::make_shared<prism::PinnedPool>();

  // ---- tuning (prism_gpu_set_option) ----
  std::atomic<uint64_t> part_min_rows{1ull << 22};  // fewer selected rows: never partition a GROUP BY
  std::atomic<int> part_force{0};                    // 1: partition whenever the plan shape allows it
  std::atomic<int> part_half_blocks{1};              // 0: the scatter pass of a partitioned GROUP BY always runs one 1024-thread workgroup per CU
  std::atomic<int> part_wc{1};                       // 0: GROUP BY key -> COUNT(*) never takes the histogram-free path
  std::atomic<int> part2_lines{1};                   // 0: the one-level COUNT(DISTINCT) scatters tile by tile (sort in LDS, runs out)
  std::atomic<int> part_wc_leading{1};               // 0: that path lists every group, whatever the plan's ORDER BY .. LIMIT
  std::atomic<uint64_t> part_wc_min_rows{1ull << 23};  // fewer selected rows: the path with the histogram
  std::atomic<int> part2_fixed{1};                   // 0: the second partition level of COUNT(DISTINCT) always runs its histogram pass
  std::atomic<int> part2_half_blocks{0};             // 1: the one-level COUNT(DISTINCT) scatter runs two workgroups of 512 threads per CU
  std::atomic<int> part2_direct{1};                  // 0: COUNT(DISTINCT) partitioned twice always sorts its pairs by the group key first, also when the rows arrive in time order
  std::atomic<int> part2_compact{1};                 // 0: the second partition level of COUNT(DISTINCT) always keeps two-word records
  std::atomic<int> contains_windows{1};              // 0: substring search always by the byte-wise prefilter
  std::atomic<int> parquet_threads{4};               // threads that decode the column chunks of a Parquet row group
  std::atomic<int64_t> dict_lookup_min_rows{65536};  // plain strings of a chunk of at least so many rows are coded on the device when the dictionary has them all
  std::atomic<int> dict_expand_device{1};            // 0: dictionary-encoded string input is expanded into offsets + bytes on the host
  std::atomic<uint64_t> stream_chunk_rows{1ull << 20};  // rows a streaming partition gathers before it stages a chunk
  std::atomic<int> stream_agg_chunks{8};             // staged chunks a streaming partition lets gather before it aggregates them in one pass
  std::atomic<int> stream_state{1};                  // 0: a streaming partition turns every chunk into a PARTIAL batch instead of keeping one group table in HBM
  std::atomic<int> shard_distinct{1};                // 0: a sharded COUNT(DISTINCT) always exchanges the set of pairs it built as a hash table in HBM
  std::atomic<int> sharded_fail_phase{0};            // tests: prism_gpu_execute_sharded fails on purpose in this local phase (1..5) on this context's rank

  // ---- profiling ----
  int profiling = 0;  // 0: off; 1: every timed launch; 2: not the passes inside a composite launch (KernelTimer::detail)
  struct Pending {
    int stat;
    hipEvent_t a, b;
  };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
  prism_gpu_stats stats;

  prism_gpu_ctx() { std::memset(&stats, 0, sizeof(stats)); }
};

namespace prism {

inline void bind(prism_gpu_ctx* ctx) { PRISM_HIP(hipSetDevice(ctx->device)); }

// (under ctx->stage_mu)
inline void ensure_stage_ring(prism_gpu_ctx* ctx) {
  for (int i = 0; i < 2; ++i) {
    if (!ctx->stage[i]) PRISM_HIP(hipHostMalloc(&ctx->stage[i], ctx->stage_bytes, hipHostMallocDefault));
    if (!ctx->stage_done[i]) PRISM_HIP(hipEventCreateWithFlags(&ctx->stage_done[i], hipEventDisableTiming));
    if (!ctx->load_landing) PRISM_HIP(hipHostMalloc(&ctx->load_landing, 64, hipHostMallocDefault));
  }
}

// Scratch memory for one call; returned to the pool by free_scratch.
inline void* alloc_scratch(prism_gpu_ctx* ctx, size_t bytes) {
  size_t cls = pow2_ceil(bytes < 256 ? 256 : bytes);
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    auto it = ctx->free_blocks.find(cls);
    if (it != ctx->free_blocks.end()) {
      void* p = it->second;
      ctx->free_blocks.erase(it);
      ctx->live_blocks[p] = cls;
      return p;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, cls);
  if (e == hipErrorOutOfMemory) {
    // give cached blocks back to the driver and retry once
    std::vector<void*> drop;
    {
      std::lock_guard<std::mutex> g(ctx->mu);
      for (auto& kv : ctx->free_blocks) drop.push_back(kv.second);
      ctx->free_blocks.clear();
    }
    for (void* d : drop) (void)hipFree(d);
    e = hipMalloc(&p, cls);
  }
  if (e != hipSuccess)
    fail(e == hipErrorOutOfMemory ? PRISM_GPU_EOOM : PRISM_GPU_EHIP,
         std::string("hipMalloc(") + std::to_string(cls) + ") failed: " + hipGetErrorString(e));
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->live_blocks[p] = cls;
  ctx->pool_bytes += cls;
  return p;
}

inline void free_scratch(prism_gpu_ctx* ctx, void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> g(ctx->mu);
  auto it = ctx->live_blocks.find(p);
  if (it == ctx->live_blocks.end()) return;
  ctx->free_blocks.emplace(it->second, p);
  ctx->live_blocks.erase(it);
}

// RAII scratch allocation.
struct Scratch {
  prism_gpu_ctx* ctx = nullptr;
  void* p = nullptr;
  Scratch() = default;
  Scratch(prism_gpu_ctx* c, size_t bytes) : ctx(c), p(alloc_scratch(c, bytes)) {}
  Scratch(const Scratch&) = delete;
  Scratch& operator=(const Scratch&) = delete;
  Scratch(Scratch&& o) noexcept : ctx(o.ctx), p(o.p) { o.p = nullptr; }
  Scratch& operator=(Scratch&& o) noexcept {
    if (this != &o) { reset(); ctx = o.ctx; p = o.p; o.p = nullptr; }
    return *this;
  }
  ~Scratch() { reset(); }
  void reset() { if (p) free_scratch(ctx, p); p = nullptr; }
  template <class T> T* as() const { return (T*)p; }
};

// ---- kernel timing -------------------------------------------------------

inline int stat_index(prism_gpu_ctx* ctx, const char* name) {
  for (uint32_t i = 0; i < ctx->stats.n_kernels; ++i)
    if (std::strncmp(ctx->stats.kernels[i].name, name, sizeof(ctx->stats.kernels[i].name)) == 0) return (int)i;
  if (ctx->stats.n_kernels >= PRISM_GPU_MAX_KERNEL_STATS) return -1;
  int i = (int)ctx->stats.n_kernels++;
  std::memset(&ctx->stats.kernels[i], 0, sizeof(ctx->stats.kernels[i]));
  std::strncpy(ctx->stats.kernels[i].name, name, sizeof(ctx->stats.kernels[i].name) - 1);
  return i;
}

inline hipEvent_t take_event(prism_gpu_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  PRISM_HIP(hipEventCreate(&e));
  return e;
}

// Brackets one kernel launch with events on the compute stream when profiling
// is on; always counts the launch and its algorithmic bytes.
struct KernelTimer {
  prism_gpu_ctx* ctx;
  int stat;
  hipEvent_t a = nullptr;
  // detail: one pass of a launch that a timer of its own brackets as a whole (the passes of a
  // partitioned GROUP BY inside "group_by").  An event between two kernels of a stream keeps
  // the second from starting for some microseconds; profiling level 2 leaves those out.
  KernelTimer(prism_gpu_ctx* c, const char* name, uint64_t algorithmic_bytes, bool detail = false) : ctx(c) {
    std::lock_guard<std::mutex> g(ctx->mu);
    stat = stat_index(ctx, name);
    if (stat >= 0) {
      ctx->stats.kernels[stat].launches += 1;
      ctx->stats.kernels[stat].algorithmic_bytes += algorithmic_bytes;
    }
    if (ctx->profiling && !(detail && ctx->profiling == 2) && stat >= 0) {
      a = take_event(ctx);
      PRISM_HIP(hipEventRecord(a, ctx->stream));
    }
  }
  ~KernelTimer() {
    if (!a) return;
    std::lock_guard<std::mutex> g(ctx->mu);
    hipEvent_t b = take_event(ctx);
    (void)hipEventRecord(b, ctx->stream);
    ctx->pending.push_back({stat, a, b});
  }
};

// Wall-clock time of a host-side phase, reported beside the kernels under a "host:"
// name (always measured: it costs two clock reads).
struct HostTimer {
  prism_gpu_ctx* ctx;
  const char* name;
  std::chrono::steady_clock::time_point t0;
  HostTimer(prism_gpu_ctx* c, const char* n) : ctx(c), name(n), t0(std::chrono::steady_clock::now()) {}
  ~HostTimer() {
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> g(ctx->mu);
    int i = stat_index(ctx, name);
    if (i >= 0) { ctx->stats.kernels[i].launches += 1; ctx->stats.kernels[i].total_ms += ms; }
  }
};

// Everything the loader has queued: the small copies and its kernels, then the large direct copies.
inline hipError_t sync_copy_streams(prism_gpu_ctx* ctx) {
  hipError_t a = ctx->copy_stream ? hipStreamSynchronize(ctx->copy_stream) : hipSuccess;
  hipError_t b = ctx->direct_stream ? hipStreamSynchronize(ctx->direct_stream) : hipSuccess;
  return a != hipSuccess ? a : b;
}

// Call after the stream has been synchronised: folds event pairs into the stats.
inline void collect_timings(prism_gpu_ctx* ctx) {
  std::lock_guard<std::mutex> g(ctx->mu);
  for (auto& p : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) ctx->stats.kernels[p.stat].total_ms += ms;
    ctx->event_pool.push_back(p.a);
    ctx->event_pool.push_back(p.b);
  }
  ctx->pending.clear();
}

}  // namespace prism
