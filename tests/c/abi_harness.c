/*
 * C harness for libprism_gpu.so: drives the C ABI of include/prism_gpu.h from plain C, with
 * hand-built Arrow C Data Interface structs, the way a host written in a compiled language
 * would (the reference's host is Rust: src/prism-query/src/main.rs:82-90).  No Python, no
 * Arrow library.  Every expected value is computed here with scalar loops.
 *
 *   abi_harness          runs the checks on GPU 0, prints "OK", exits 0
 *   exit code 3          no GPU: prism_gpu_init refused loudly (there is no CPU fallback)
 *   exit code 1          a check failed (message on stderr)
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/arrow_c_data.h"
#include "../../include/prism_gpu.h"

#define N_ROWS 5000
#define N_KEYS 37

static void release_noop_schema(struct ArrowSchema* s) { s->release = NULL; }
static void release_array(struct ArrowArray* a) {
  /* buffers are owned by the caller's static arrays; only mark released */
  a->release = NULL;
}

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
  fprintf(stderr, __VA_ARGS__); fprintf(stderr, " (last error: %s)\n", prism_gpu_last_error()); exit(1); } } while (0)

static int64_t g_k[N_ROWS], g_v[N_ROWS];
static uint8_t g_v_valid[(N_ROWS + 7) / 8];

/* A struct array {k: int64, v: int64 nullable} over rows [lo, hi) of the static data. */
typedef struct Batch {
  struct ArrowArray top, cols[2];
  struct ArrowArray* children[2];
  const void* top_buffers[1];
  const void* k_buffers[2];
  const void* v_buffers[2];
} Batch;

static void make_batch(Batch* b, int64_t lo, int64_t hi) {
  memset(b, 0, sizeof(*b));
  b->k_buffers[0] = NULL;
  b->k_buffers[1] = g_k;
  b->v_buffers[0] = g_v_valid;
  b->v_buffers[1] = g_v;
  int64_t nulls = 0;
  for (int64_t i = lo; i < hi; ++i) nulls += !((g_v_valid[i >> 3] >> (i & 7)) & 1);
  for (int c = 0; c < 2; ++c) {
    b->cols[c].length = hi - lo;
    b->cols[c].offset = lo; /* a sliced array: offset applies to values and validity */
    b->cols[c].n_buffers = 2;
    b->cols[c].release = release_array;
    b->children[c] = &b->cols[c];
  }
  b->cols[0].buffers = b->k_buffers;
  b->cols[1].buffers = b->v_buffers;
  b->cols[1].null_count = nulls;
  b->top.length = hi - lo;
  b->top.n_buffers = 1;
  b->top.buffers = b->top_buffers;
  b->top.n_children = 2;
  b->top.children = b->children;
  b->top.release = release_array;
}

typedef struct Expect { int64_t count, sum, min; int any; } Expect;

/* out: struct array {k, COUNT(*), SUM(v), MIN(v)} */
static void check_result(const struct ArrowArray* out, const Expect* want, int n_want, const char* what) {
  CHECK(out->n_children == 4, "%s: %lld columns", what, (long long)out->n_children);
  CHECK(out->length == n_want, "%s: %lld groups, expected %d", what, (long long)out->length, n_want);
  const int64_t* k = (const int64_t*)out->children[0]->buffers[1];
  const int64_t* cnt = (const int64_t*)out->children[1]->buffers[1];
  const int64_t* sum = (const int64_t*)out->children[2]->buffers[1];
  const int64_t* mn = (const int64_t*)out->children[3]->buffers[1];
  const uint8_t* sum_valid = (const uint8_t*)out->children[2]->buffers[0];
  int seen[N_KEYS] = {0};
  for (int64_t i = 0; i < out->length; ++i) {
    CHECK(k[i] >= 0 && k[i] < N_KEYS && !seen[k[i]], "%s: bad or repeated key %lld", what, (long long)k[i]);
    seen[k[i]] = 1;
    const Expect* e = &want[k[i]];
    CHECK(e->count > 0, "%s: group %lld should not exist", what, (long long)k[i]);
    CHECK(cnt[i] == e->count, "%s: COUNT(*) of %lld = %lld, expected %lld", what, (long long)k[i], (long long)cnt[i], (long long)e->count);
    int valid = !sum_valid || ((sum_valid[i >> 3] >> (i & 7)) & 1);
    CHECK(valid == e->any, "%s: SUM validity of %lld", what, (long long)k[i]);
    if (e->any) {
      CHECK(sum[i] == e->sum, "%s: SUM of %lld = %lld, expected %lld", what, (long long)k[i], (long long)sum[i], (long long)e->sum);
      CHECK(mn[i] == e->min, "%s: MIN of %lld", what, (long long)k[i]);
    }
  }
}

/* ---- concurrent streaming partitions ------------------------------------------------
 * DataFusion polls execute(partition) of every partition on its own worker thread
 * (src/prism-query/src/main.rs:45) and hands over 8192-row batches.  Four threads, one
 * prism_gpu_part each on a shared plan; every batch owns its buffers and frees them when the
 * library releases it (the library keeps batches until a chunk of them is staged). */
#define T_THREADS 4
#define T_ROWS_PER_THREAD 2300000 /* more than two chunks of 2^20 rows */
#define T_BATCH 8192

static int64_t big_k(int64_t i) { return (int64_t)(((uint64_t)i * 2654435761ull >> 7) % N_KEYS); }
static int64_t big_v(int64_t i) { return (int64_t)(((uint64_t)i * 40503ull + 17) % 2000) - 1000; }
static int big_valid(int64_t i) { return i % 13 != 5; }

typedef struct HeapBatch {
  struct ArrowArray top, cols[2];
  struct ArrowArray* children[2];
  const void* top_buffers[1];
  const void* k_buffers[2];
  const void* v_buffers[2];
  int64_t* k;
  int64_t* v;
  uint8_t* valid;
} HeapBatch;

static void release_child(struct ArrowArray* a) { a->release = NULL; }
static void release_heap_batch(struct ArrowArray* a) {
  HeapBatch* b = (HeapBatch*)a->private_data;
  a->release = NULL;
  free(b->k); free(b->v); free(b->valid); free(b);
}

/* rows [lo, hi) of the generated table as a batch the consumer owns */
static struct ArrowArray* heap_batch(int64_t lo, int64_t hi) {
  HeapBatch* b = (HeapBatch*)calloc(1, sizeof(HeapBatch));
  int64_t n = hi - lo, nulls = 0;
  b->k = (int64_t*)malloc((size_t)n * 8);
  b->v = (int64_t*)malloc((size_t)n * 8);
  b->valid = (uint8_t*)calloc((size_t)(n + 7) / 8, 1);
  for (int64_t i = 0; i < n; ++i) {
    b->k[i] = big_k(lo + i);
    b->v[i] = big_v(lo + i);
    if (big_valid(lo + i)) b->valid[i >> 3] |= (uint8_t)(1u << (i & 7)); else ++nulls;
  }
  b->k_buffers[1] = b->k;
  b->v_buffers[0] = b->valid;
  b->v_buffers[1] = b->v;
  for (int c = 0; c < 2; ++c) {
    b->cols[c].length = n;
    b->cols[c].n_buffers = 2;
    b->cols[c].release = release_child;
    b->children[c] = &b->cols[c];
  }
  b->cols[0].buffers = b->k_buffers;
  b->cols[1].buffers = b->v_buffers;
  b->cols[1].null_count = nulls;
  b->top.length = n;
  b->top.n_buffers = 1;
  b->top.buffers = b->top_buffers;
  b->top.n_children = 2;
  b->top.children = b->children;
  b->top.release = release_heap_batch;
  b->top.private_data = b;
  return &b->top;
}

typedef struct ThreadArg {
  prism_gpu_plan* plan;
  uint32_t partition;
  int64_t lo, hi;
  struct ArrowArray state;
  int rc;
  char err[256];
} ThreadArg;

static void* stream_partition(void* p) {
  ThreadArg* a = (ThreadArg*)p;
  prism_gpu_part* part = NULL;
  a->rc = prism_gpu_partition_open(a->plan, a->partition, &part);
  for (int64_t r = a->lo; a->rc == 0 && r < a->hi; r += T_BATCH) {
    int64_t z = r + T_BATCH < a->hi ? r + T_BATCH : a->hi;
    a->rc = prism_gpu_push_batch(part, heap_batch(r, z));
  }
  if (a->rc == 0) a->rc = prism_gpu_finish(part, &a->state);
  if (a->rc != 0) snprintf(a->err, sizeof(a->err), "%s", prism_gpu_last_error()); /* the message is per thread */
  prism_gpu_part_free(part);
  return NULL;
}

static void check_concurrent_streams(prism_gpu_plan* pplan) {
  static Expect want[N_KEYS];
  memset(want, 0, sizeof(want));
  const int64_t total = (int64_t)T_THREADS * T_ROWS_PER_THREAD;
  for (int64_t i = 0; i < total; ++i) {
    int valid = big_valid(i);
    int64_t v = big_v(i);
    if (valid && v < -500) continue;
    Expect* e = &want[big_k(i)];
    e->count += 1;
    if (valid) { e->sum += v; e->min = (!e->any || v < e->min) ? v : e->min; e->any = 1; }
  }
  int n_want = 0;
  for (int k = 0; k < N_KEYS; ++k) n_want += want[k].count > 0;

  pthread_t th[T_THREADS];
  static ThreadArg args[T_THREADS];
  struct ArrowArray* ptrs[T_THREADS + 1];
  for (int t = 0; t < T_THREADS; ++t) {
    memset(&args[t], 0, sizeof(args[t]));
    args[t].plan = pplan;
    args[t].partition = (uint32_t)t;
    args[t].lo = (int64_t)t * T_ROWS_PER_THREAD;
    args[t].hi = (int64_t)(t + 1) * T_ROWS_PER_THREAD;
    CHECK(pthread_create(&th[t], NULL, stream_partition, &args[t]) == 0, "pthread_create");
  }
  for (int t = 0; t < T_THREADS; ++t) {
    pthread_join(th[t], NULL);
    CHECK(args[t].rc == 0, "streaming thread %d: rc %d: %s", t, args[t].rc, args[t].err);
    ptrs[t] = &args[t].state;
  }
  struct ArrowArray out;
  CHECK(prism_gpu_merge(pplan, ptrs, T_THREADS, &out) == 0, "merge of concurrent streams");
  check_result(&out, want, n_want, "4 concurrent streaming partitions");
  out.release(&out);
  /* the same rows through one part on this thread give the same answer */
  ThreadArg one;
  memset(&one, 0, sizeof(one));
  one.plan = pplan; one.partition = 0; one.lo = 0; one.hi = total;
  stream_partition(&one);
  CHECK(one.rc == 0, "single stream: rc %d: %s", one.rc, one.err);
  ptrs[0] = &one.state;
  CHECK(prism_gpu_merge(pplan, ptrs, 1, &out) == 0, "merge of the single stream");
  check_result(&out, want, n_want, "one streaming partition over all rows");
  out.release(&out);
  one.state.release(&one.state);
  for (int t = 0; t < T_THREADS; ++t) args[t].state.release(&args[t].state);
}

int main(void) {
  /* data: k cycles unevenly over N_KEYS values, v in [-1000, 1000), every 11th v is NULL */
  uint64_t x = 0x9E3779B97F4A7C15ull;
  memset(g_v_valid, 0, sizeof(g_v_valid));
  for (int i = 0; i < N_ROWS; ++i) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    g_k[i] = (int64_t)((x >> 33) % N_KEYS);
    g_v[i] = (int64_t)((x >> 13) % 2000) - 1000;
    if (i % 11 != 0) g_v_valid[i >> 3] |= (uint8_t)(1u << (i & 7));
  }

  prism_gpu_ctx* ctx = NULL;
  int rc = prism_gpu_init(0, &ctx);
  if (rc != PRISM_GPU_OK) {
    fprintf(stderr, "NO GPU: prism_gpu_init -> %d: %s\n", rc, prism_gpu_last_error());
    return rc == PRISM_GPU_EHIP ? 3 : 1;
  }

  /* schema {k: int64, v: int64} */
  struct ArrowSchema fields[2], top, *fptr[2] = {&fields[0], &fields[1]};
  memset(fields, 0, sizeof(fields));
  memset(&top, 0, sizeof(top));
  fields[0].format = "l"; fields[0].name = "k"; fields[0].flags = ARROW_FLAG_NULLABLE; fields[0].release = release_noop_schema;
  fields[1].format = "l"; fields[1].name = "v"; fields[1].flags = ARROW_FLAG_NULLABLE; fields[1].release = release_noop_schema;
  top.format = "+s"; top.name = ""; top.n_children = 2; top.children = fptr; top.release = release_noop_schema;

  /* plan: SELECT k, COUNT(*), SUM(v), MIN(v) WHERE v >= -500 OR v IS NULL GROUP BY k */
  prism_gpu_expr ex[6];
  memset(ex, 0, sizeof(ex));
  ex[0].kind = PRISM_EXPR_COLUMN; ex[0].column = 1; ex[0].lhs = ex[0].rhs = -1;
  ex[1].kind = PRISM_EXPR_LIT_I64; ex[1].i64 = -500; ex[1].lhs = ex[1].rhs = -1;
  ex[2].kind = PRISM_EXPR_CMP; ex[2].op = PRISM_CMP_GTEQ; ex[2].lhs = 0; ex[2].rhs = 1;
  ex[3].kind = PRISM_EXPR_IS_NULL; ex[3].lhs = 0; ex[3].rhs = -1;
  ex[4].kind = PRISM_EXPR_OR; ex[4].lhs = 2; ex[4].rhs = 3;
  prism_gpu_key key = {PRISM_KEY_COLUMN, 0, 0, 0, NULL};
  prism_gpu_agg aggs[3] = {{PRISM_AGG_COUNT_STAR, -1, NULL}, {PRISM_AGG_SUM, 1, NULL}, {PRISM_AGG_MIN, 1, NULL}};
  prism_gpu_plan_desc desc;
  memset(&desc, 0, sizeof(desc));
  desc.abi_version = PRISM_GPU_ABI_VERSION;
  desc.mode = PRISM_MODE_SINGLE;
  desc.exprs = ex; desc.n_exprs = 5; desc.predicate = 4;
  desc.keys = &key; desc.n_keys = 1;
  desc.aggs = aggs; desc.n_aggs = 3;
  desc.order_by = -1; desc.limit = -1;

  Expect want[N_KEYS], want_p0[N_KEYS];
  memset(want, 0, sizeof(want));
  memset(want_p0, 0, sizeof(want_p0));
  const int split = 1777; /* partition 0 = rows [0, split), partition 1 = the rest */
  for (int i = 0; i < N_ROWS; ++i) {
    int valid = (g_v_valid[i >> 3] >> (i & 7)) & 1;
    if (valid && g_v[i] < -500) continue;
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1 && i >= split) break;
      Expect* e = pass ? &want_p0[g_k[i]] : &want[g_k[i]];
      e->count += 1;
      if (valid) {
        e->sum += g_v[i];
        e->min = (!e->any || g_v[i] < e->min) ? g_v[i] : e->min;
        e->any = 1;
      }
    }
  }
  int n_want = 0, n_want_p0 = 0;
  for (int k = 0; k < N_KEYS; ++k) { n_want += want[k].count > 0; n_want_p0 += want_p0[k].count > 0; }

  /* 1. resident table, two partitions from sliced arrays */
  prism_gpu_table* table = NULL;
  CHECK(prism_gpu_table_create(ctx, &top, &table) == 0, "table_create");
  Batch b0, b1;
  make_batch(&b0, 0, split);
  make_batch(&b1, split, N_ROWS);
  CHECK(prism_gpu_table_append(table, 0, &b0.top) == 0, "append 0");
  CHECK(b0.top.release == NULL, "append must release the moved array");
  CHECK(prism_gpu_table_append(table, 1, &b1.top) == 0, "append 1");
  uint64_t rows = 0; uint32_t parts = 0;
  CHECK(prism_gpu_table_num_rows(table, &rows) == 0 && rows == N_ROWS, "num_rows");
  CHECK(prism_gpu_table_num_partitions(table, &parts) == 0 && parts == 2, "num_partitions");

  prism_gpu_plan* plan = NULL;
  CHECK(prism_gpu_plan_create(ctx, &top, &desc, &plan) == 0, "plan_create");
  struct ArrowSchema out_schema;
  CHECK(prism_gpu_plan_output_schema(plan, &out_schema) == 0, "output_schema");
  CHECK(out_schema.n_children == 4 && strcmp(out_schema.children[1]->name, "COUNT(*)") == 0 &&
            strcmp(out_schema.children[2]->name, "SUM(v)") == 0 && strcmp(out_schema.children[2]->format, "l") == 0,
        "output schema names/types");
  out_schema.release(&out_schema);

  struct ArrowArray out;
  CHECK(prism_gpu_execute(plan, table, NULL, 0, &out) == 0, "execute all");
  check_result(&out, want, n_want, "execute all");
  out.release(&out);
  uint32_t only0 = 0;
  CHECK(prism_gpu_execute(plan, table, &only0, 1, &out) == 0, "execute partition 0");
  check_result(&out, want_p0, n_want_p0, "execute partition 0");
  out.release(&out);

  /* 2. streaming form + PARTIAL -> merge */
  desc.mode = PRISM_MODE_PARTIAL;
  prism_gpu_plan* pplan = NULL;
  CHECK(prism_gpu_plan_create(ctx, &top, &desc, &pplan) == 0, "plan_create partial");
  struct ArrowArray states[2], *state_ptrs[2] = {&states[0], &states[1]};
  for (uint32_t p = 0; p < 2; ++p) {
    prism_gpu_part* part = NULL;
    CHECK(prism_gpu_partition_open(pplan, p, &part) == 0, "partition_open");
    /* feed the partition in three pushes */
    int64_t lo = p == 0 ? 0 : split, hi = p == 0 ? split : N_ROWS;
    Batch b[3]; /* a moved array keeps pointing at its children: they live until finish */
    for (int s = 0; s < 3; ++s) {
      int64_t a = lo + (hi - lo) * s / 3, z = lo + (hi - lo) * (s + 1) / 3;
      make_batch(&b[s], a, z);
      CHECK(prism_gpu_push_batch(part, &b[s].top) == 0, "push_batch");
    }
    CHECK(prism_gpu_finish(part, &states[p]) == 0, "finish");
    prism_gpu_part_free(part);
  }
  CHECK(prism_gpu_merge(pplan, state_ptrs, 2, &out) == 0, "merge");
  check_result(&out, want, n_want, "partial -> merge");
  out.release(&out);
  CHECK(states[0].release != NULL, "merge borrows the states");
  states[0].release(&states[0]);
  states[1].release(&states[1]);

  /* 3. device-resident state: statistics, aggregate, describe, finish */
  {
    prism_gpu_col_stats stats[2];
    CHECK(prism_gpu_scan_stats(plan, table, NULL, 0, stats) == 0, "scan_stats");
    CHECK(stats[0].min == 0 && stats[0].max == N_KEYS - 1, "scan_stats of k: %lld..%lld", (long long)stats[0].min, (long long)stats[0].max);
    uint32_t p0 = 0, p1 = 1;
    prism_gpu_state *s0 = NULL, *s1 = NULL;
    CHECK(prism_gpu_aggregate(plan, table, &p0, 1, stats, &s0) == 0, "aggregate 0");
    CHECK(prism_gpu_aggregate(plan, table, &p1, 1, stats, &s1) == 0, "aggregate 1");
    prism_gpu_state_info info;
    CHECK(prism_gpu_state_describe(s1, &info) == 0 && info.dense == 1 && info.n_slots >= 3, "describe");
    /* (export / import / merge take device buffers and are exercised from the Python tests,
       which have a device allocator at hand; this harness links no HIP runtime itself) */
    CHECK(prism_gpu_state_finish(s0, &out) == 0, "state_finish");
    check_result(&out, want_p0, n_want_p0, "state of partition 0");
    out.release(&out);
    prism_gpu_state_free(s0);
    prism_gpu_state_free(s1);
  }

  /* 4. four threads streaming a partition each, 8192 rows at a time */
  check_concurrent_streams(pplan);

  /* 5. errors come back as codes with a message, never as aborts */
  prism_gpu_plan* bad = NULL;
  ex[0].column = 7; /* no such column */
  desc.mode = PRISM_MODE_SINGLE;
  rc = prism_gpu_plan_create(ctx, &top, &desc, &bad);
  CHECK(rc == PRISM_GPU_EINVAL && bad == NULL && strlen(prism_gpu_last_error()) > 0, "unknown column: rc %d", rc);
  CHECK(prism_gpu_execute(NULL, table, NULL, 0, &out) == PRISM_GPU_EINVAL, "null plan");

  prism_gpu_plan_free(pplan);
  prism_gpu_plan_free(plan);
  prism_gpu_table_free(table);
  prism_gpu_ctx_free(ctx);
  printf("OK\n");
  return 0;
}
