// End-to-end feeder for the streaming form of the operator: host Arrow batches of 8192 rows
// (what ParquetExec hands downstream, src/prism-query/src/providers/table_provider.rs:110-131),
// one prism_gpu_part per partition on a thread of its own (src/prism-query/src/main.rs:45),
// PARTIAL state per partition, prism_gpu_merge at the end.  Workload: the log-events query
// (date_bin(1 minute, timestamp), service -> COUNT(*), SUM(value), MAX(value)); `service`
// arrives dictionary encoded.  Prints one JSON object: rows/s from the first push to the
// merged result, and the staging rate against the PCIe bound.
// With `c3` as the workload: one Int64 id column (log-uniform ranks over 2^24 hashed to 63 bits,
// as the C3 generator of prism_synth.h draws them) grouped by the id with COUNT(*): most rows of
// a chunk are groups of their own, every partition meets nearly every id.
//
//   stream_bench <rows> <threads> <steps> [batch_rows] [c2|c3] [state|nostate] [option=value ...]
// (nostate: option "stream_state" 0, a PARTIAL batch per chunk instead of one table per partition in HBM)
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/arrow_c_data.h"
#include "../../include/prism_gpu.h"
#include "../../include/prism_synth.h"

static void noop_release_schema(ArrowSchema* s) { s->release = nullptr; }
static void noop_release_array(ArrowArray* a) { a->release = nullptr; }

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, prism_gpu_last_error()); exit(1); } } while (0)

static uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct Partition {
  std::vector<int64_t> ts, value;
  std::vector<int32_t> svc;
  // "c2plain": the service column as plain Utf8 (what an Arrow scan yields unless asked for dictionaries)
  std::vector<int32_t> svc_off;
  std::string svc_bytes;
};

// One batch = rows [lo, hi) of a partition: struct array whose children point into the
// partition's buffers through their offsets (nothing is copied to build it).
struct Batch {
  ArrowArray top, cols[3], dict;
  ArrowArray* children[3];
  const void* top_buf[1];
  const void* ts_buf[2];
  const void* svc_buf[3];
  const void* val_buf[2];
};

int main(int argc, char** argv) {
  const uint64_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 200000000ull;
  const int threads = argc > 2 ? atoi(argv[2]) : 16;
  const int steps = argc > 3 ? atoi(argv[3]) : 3;
  const int64_t batch_rows = argc > 4 ? atoll(argv[4]) : 8192;
  const bool c3 = argc > 5 && strcmp(argv[5], "c3") == 0;
  const bool plain = argc > 5 && strcmp(argv[5], "c2plain") == 0;
  const int n_services = 64;
  const int64_t t0 = 1700000000000ll, day = 86400000ll;

  prism_gpu_ctx* ctx = nullptr;
  if (prism_gpu_init(0, &ctx) != 0) { fprintf(stderr, "NO GPU: %s\n", prism_gpu_last_error()); return 3; }

  // dictionary of service names
  std::string names;
  std::vector<int32_t> name_off(1, 0);
  for (int i = 0; i < n_services; ++i) {
    char buf[16];
    snprintf(buf, sizeof(buf), "svc-%02d", i);
    names += buf;
    name_off.push_back((int32_t)names.size());
  }
  const void* dict_buf[3] = {nullptr, name_off.data(), names.data()};

  // schema {timestamp: Timestamp(ms), service: Dictionary(Int32, Utf8), value: Int64}
  ArrowSchema f[3], dict_schema, top, *fp[3] = {&f[0], &f[1], &f[2]};
  memset(f, 0, sizeof(f)); memset(&dict_schema, 0, sizeof(dict_schema)); memset(&top, 0, sizeof(top));
  f[0].format = "tsm:"; f[0].name = "timestamp";
  f[1].format = "i"; f[1].name = "service"; f[1].dictionary = &dict_schema;
  f[2].format = "l"; f[2].name = "value";
  dict_schema.format = "u"; dict_schema.name = ""; dict_schema.release = noop_release_schema;
  if (plain) { f[1].format = "u"; f[1].dictionary = nullptr; }
  for (auto& x : f) { x.flags = ARROW_FLAG_NULLABLE; x.release = noop_release_schema; }
  top.format = "+s"; top.name = ""; top.n_children = 3; top.children = fp; top.release = noop_release_schema;

  if (argc > 6 && strcmp(argv[6], "nostate") == 0) CHECK(prism_gpu_set_option(ctx, "stream_state", 0));
  for (int a = 7; a < argc; ++a) {  // further arguments: option=value (prism_gpu_set_option)
    const char* eq = strchr(argv[a], '=');
    if (!eq) { fprintf(stderr, "expected option=value, got %s\n", argv[a]); return 2; }
    CHECK(prism_gpu_set_option(ctx, std::string(argv[a], eq - argv[a]).c_str(), atoll(eq + 1)));
  }
  prism_gpu_key keys[2] = {{PRISM_KEY_DATE_BIN, 0, 60000, 0, nullptr}, {PRISM_KEY_COLUMN, 1, 0, 0, nullptr}};
  prism_gpu_agg aggs[3] = {{PRISM_AGG_COUNT_STAR, -1, nullptr}, {PRISM_AGG_SUM, 2, nullptr}, {PRISM_AGG_MAX, 2, nullptr}};
  prism_gpu_plan_desc desc;
  memset(&desc, 0, sizeof(desc));
  desc.abi_version = PRISM_GPU_ABI_VERSION;
  desc.mode = PRISM_MODE_PARTIAL;
  desc.predicate = -1;
  desc.keys = keys; desc.n_keys = 2;
  desc.aggs = aggs; desc.n_aggs = 3;
  desc.order_by = -1; desc.limit = -1;
  if (c3) {  // {UserID: Int64}, GROUP BY UserID -> COUNT(*)
    f[0].format = "l"; f[0].name = "UserID";
    top.n_children = 1;
    keys[0] = {PRISM_KEY_COLUMN, 0, 0, 0, nullptr};
    desc.n_keys = 1;
    desc.n_aggs = 1;
    // ORDER BY COUNT(*) DESC LIMIT 10, applied by the merge (a PARTIAL state carries every group)
    desc.order_by = 1; desc.descending = 1; desc.limit = 10;
  }
  const int n_cols = c3 ? 1 : 3;
  prism_gpu_plan* plan = nullptr;
  CHECK(prism_gpu_plan_create(ctx, &top, &desc, &plan));

  // partitions: contiguous row ranges, time ordered, generated by the threads that will feed them
  const int n_parts = threads;
  std::vector<Partition> parts(n_parts);
  {
    std::vector<std::thread> gen;
    for (int p = 0; p < n_parts; ++p)
      gen.emplace_back([&, p] {
        const uint64_t lo = rows * p / n_parts, hi = rows * (p + 1) / n_parts, n = hi - lo;
        Partition& P = parts[p];
        if (c3) {
          const int64_t prm[2] = {24, 0x5EED};
          const uint64_t key = prism_synth_column_key(3, 0);
          P.ts.resize(n);  // (the id column lives in `ts`)
          for (uint64_t i = 0; i < n; ++i) P.ts[i] = prism_synth_value(PRISM_SYNTH_LOGUNIFORM_ID, prm, nullptr, 0, key, lo + i, rows);
          return;
        }
        P.ts.resize(n); P.value.resize(n); P.svc.resize(n);
        for (uint64_t i = 0; i < n; ++i) {
          const uint64_t r = lo + i, h = mix64(r);
          P.ts[i] = t0 + (int64_t)((unsigned __int128)r * (uint64_t)day / rows);
          P.svc[i] = (int32_t)((h >> 40) % (uint64_t)n_services);
          P.value[i] = (int64_t)((h & 0xffffffffull) * 50000ull >> 32);
        }
        if (plain) {
          P.svc_off.resize(n + 1);
          P.svc_off[0] = 0;
          P.svc_bytes.reserve(n * 6);
          for (uint64_t i = 0; i < n; ++i) {
            P.svc_bytes.append(names, (size_t)name_off[P.svc[i]], (size_t)(name_off[P.svc[i] + 1] - name_off[P.svc[i]]));
            P.svc_off[i + 1] = (int32_t)P.svc_bytes.size();
          }
        }
      });
    for (auto& t : gen) t.join();
  }
  const uint64_t bytes_per_row = c3 ? 8 : plain ? 8 + 4 + 6 + 8 : 8 + 4 + 8;

  double best = 1e30, total = 0;
  uint64_t groups = 0;
  for (int step = 0; step < steps + 1; ++step) {  // step 0 warms the pools up
    CHECK(prism_gpu_reset_stats(ctx));
    // (the verbose last step is timed kernel by kernel: slower, and not part of the average's best)
    if (getenv("STREAM_BENCH_VERBOSE")) CHECK(prism_gpu_set_profiling(ctx, step == steps ? 1 : 0));
    std::vector<ArrowArray> states(n_parts);
    auto t_start = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int p = 0; p < n_parts; ++p)
      th.emplace_back([&, p] {
        const Partition& P = parts[p];
        prism_gpu_part* part = nullptr;
        CHECK(prism_gpu_partition_open(plan, (uint32_t)p, &part));
        const int64_t n = (int64_t)P.ts.size();
        for (int64_t lo = 0; lo < n; lo += batch_rows) {
          const int64_t len = lo + batch_rows < n ? batch_rows : n - lo;
          Batch b;
          memset(&b, 0, sizeof(b));
          b.ts_buf[1] = P.ts.data(); b.svc_buf[1] = P.svc.data(); b.val_buf[1] = P.value.data();
          if (plain) { b.svc_buf[1] = P.svc_off.data(); b.svc_buf[2] = P.svc_bytes.data(); }
          const void** bufs[3] = {b.ts_buf, b.svc_buf, b.val_buf};
          for (int c = 0; c < n_cols; ++c) {
            b.cols[c].length = len; b.cols[c].offset = lo; b.cols[c].n_buffers = 2;
            b.cols[c].buffers = bufs[c]; b.cols[c].release = noop_release_array;
            b.children[c] = &b.cols[c];
          }
          b.dict.length = n_services; b.dict.n_buffers = 3; b.dict.buffers = dict_buf; b.dict.release = noop_release_array;
          if (!c3 && !plain) b.cols[1].dictionary = &b.dict;
          if (plain) b.cols[1].n_buffers = 3;
          b.top.length = len; b.top.n_buffers = 1; b.top.buffers = b.top_buf; b.top.n_children = n_cols;
          b.top.children = b.children; b.top.release = noop_release_array;
          // (the part keeps the ArrowArray struct by value; the children live in `b`... so a
          // batch must outlive the push until it is staged: allocate it)
          Batch* hb = new Batch(b);
          for (int c = 0; c < n_cols; ++c) { hb->cols[c].buffers = c == 0 ? hb->ts_buf : c == 1 ? hb->svc_buf : hb->val_buf; hb->children[c] = &hb->cols[c]; }
          if (!c3 && !plain) hb->cols[1].dictionary = &hb->dict;
          hb->top.buffers = hb->top_buf; hb->top.children = hb->children;
          hb->top.private_data = hb;
          hb->top.release = [](ArrowArray* a) { Batch* x = (Batch*)a->private_data; a->release = nullptr; delete x; };
          CHECK(prism_gpu_push_batch(part, &hb->top));
        }
        CHECK(prism_gpu_finish(part, &states[p]));
        prism_gpu_part_free(part);
      });
    for (auto& t : th) t.join();
    const double parts_sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    std::vector<ArrowArray*> ptrs;
    for (auto& s : states) ptrs.push_back(&s);
    ArrowArray out;
    if (step == steps && getenv("STREAM_BENCH_VERBOSE") && atoi(getenv("STREAM_BENCH_VERBOSE")) > 1) CHECK(prism_gpu_reset_stats(ctx));  // (2: the merge's own statistics)
    CHECK(prism_gpu_merge(plan, ptrs.data(), ptrs.size(), &out));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    const int64_t* cnt = (const int64_t*)out.children[c3 ? 1 : 2]->buffers[1];
    if (c3) {
      // the ten most frequent ids, in order; the ids a partition saw are counted in its state
      if (out.length != 10) { fprintf(stderr, "%lld rows, not the 10 leading ones\n", (long long)out.length); return 1; }
      for (int64_t i = 1; i < out.length; ++i)
        if (cnt[i] > cnt[i - 1]) { fprintf(stderr, "result is not ordered by COUNT(*)\n"); return 1; }
      groups = 0;
      for (auto& s : states) groups += (uint64_t)s.length;
    } else {
      // every row lands in exactly one group
      uint64_t sum = 0;
      for (int64_t i = 0; i < out.length; ++i) sum += (uint64_t)cnt[i];
      if (sum != rows) { fprintf(stderr, "COUNT(*) adds up to %" PRIu64 ", not %" PRIu64 "\n", sum, rows); return 1; }
      groups = (uint64_t)out.length;
    }
    out.release(&out);
    for (auto& s : states) s.release(&s);
    if (step > 0) { total += sec; if (sec < best) best = sec; }
    if (step == steps && getenv("STREAM_BENCH_VERBOSE")) {
      prism_gpu_stats vs;
      CHECK(prism_gpu_get_stats(ctx, &vs));
      fprintf(stderr, "last step: partitions done after %.1f ms, merged result after %.1f ms\n", parts_sec * 1e3, sec * 1e3);
      for (uint32_t k = 0; k < vs.n_kernels; ++k)
        fprintf(stderr, "  %-24s %8llu x  %10.2f ms in all\n", vs.kernels[k].name, (unsigned long long)vs.kernels[k].launches, vs.kernels[k].total_ms);
    }
  }
  prism_gpu_stats st;
  CHECK(prism_gpu_get_stats(ctx, &st));
  const double avg = total / steps;
  printf("{\"workload\": \"%s\", \"rows\": %" PRIu64 ", \"threads\": %d, \"batch_rows\": %lld, \"steps\": %d, \"groups\": %" PRIu64
         ", \"seconds_avg\": %.6f, \"seconds_best\": %.6f, \"rows_per_s\": %.1f, \"host_bytes_per_row\": %" PRIu64
         ", \"host_to_hbm_GBps\": %.3f, \"bytes_staged_last_step\": %" PRIu64 ", \"staging_thread_ms_last_step\": %.1f}\n",
         c3 ? "c3" : plain ? "c2plain" : "c2", rows, threads, (long long)batch_rows, steps, groups, avg, best, rows / avg, bytes_per_row,
         rows * bytes_per_row / avg / 1e9, (uint64_t)st.bytes_staged, st.staging_ms);
  prism_gpu_plan_free(plan);
  prism_gpu_ctx_free(ctx);
  return 0;
}
