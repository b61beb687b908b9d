// The exchange between the GPUs of a node, inside the library: a communicator per rank
// (RCCL over xGMI, or three callbacks of the caller), and the sharded form of a GROUP BY
// query on top of it (prism_gpu_execute_sharded).
//
// The reference has no multi-GPU code; what it has is DataFusion's plan shape for a
// partitioned aggregate: AggregateExec(Partial) per partition, RepartitionExec(Hash) on the
// group keys, AggregateExec(FinalPartitioned) per hash partition, then the output
// partitions coalesced (every partition of src/prism-query/src/providers/
// table_provider.rs:94-109 is such a unit).  Here a rank's partitions are aggregated into
// one group table in HBM, and the tables meet without leaving device memory.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

#include "context.hpp"
#include "engine.hpp"
#include "kernels.hpp"

namespace prism {

namespace {

// librccl is loaded when the first communicator is made: the library itself has no link
// dependency on it, and a process that already holds a copy (PyTorch ships one under the
// same soname) shares that one.
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

Rccl& rccl() {
  static Rccl r;
  static std::mutex mu;
  std::lock_guard<std::mutex> g(mu);
  if (r.handle) return r;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  std::string why;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
    const char* e = dlerror();
    why += std::string(why.empty() ? "" : "; ") + (e ? e : n);
  }
  if (!h) fail(PRISM_GPU_ERCCL, "librccl could not be loaded: " + why);
  auto sym = [&](const char* name) {
    void* p = dlsym(h, name);
    if (!p) fail(PRISM_GPU_ERCCL, std::string("librccl has no ") + name);
    return p;
  };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
  r.Reduce = (decltype(r.Reduce))sym("ncclReduce");
  r.Send = (decltype(r.Send))sym("ncclSend");
  r.Recv = (decltype(r.Recv))sym("ncclRecv");
  r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  r.handle = h;
  return r;
}

#define PRISM_RCCL(expr)                                                                                         \
  do {                                                                                                           \
    ncclResult_t _r = (expr);                                                                                    \
    if (_r != ncclSuccess)                                                                                       \
      ::prism::fail(PRISM_GPU_ERCCL, std::string(#expr) + " failed: " + rccl().GetErrorString(_r) + " (" + __FILE__ + ":" + \
                                         std::to_string(__LINE__) + ")");                                       \
  } while (0)

static_assert(sizeof(ncclUniqueId) == PRISM_GPU_COMM_ID_BYTES, "the unique id is handed around as 128 bytes");

}  // namespace

}  // namespace prism

struct prism_gpu_comm {
  prism_gpu_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  bool callbacks = false;
  bool aborted = false;  // a collective failed: the communicator was torn down and takes no more calls
  prism_gpu_transport cb{};
  std::mutex mu;
  prism_gpu_comm_stats stats{};
};

namespace prism {

namespace {

// Page-locked host block for the callbacks transport (device buffers travel as host copies).
struct Pinned {
  void* p = nullptr;
  explicit Pinned(size_t bytes) { PRISM_HIP(hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault)); }
  Pinned(const Pinned&) = delete;
  Pinned& operator=(const Pinned&) = delete;
  ~Pinned() { (void)hipHostFree(p); }
  template <class T> T* as() const { return (T*)p; }
};

void account(prism_gpu_comm* c, uint64_t collectives, uint64_t bytes) {
  std::lock_guard<std::mutex> g(c->mu);
  c->stats.collectives += collectives;
  c->stats.bytes_sent += bytes;
}

void cb_check(int rc, const char* what) {
  if (rc != 0) fail(PRISM_GPU_ERCCL, std::string("the transport's ") + what + " returned " + std::to_string(rc));
}

// Small vectors of 8-byte values, one per rank, in host memory.
void allgather_small(prism_gpu_comm* c, const int64_t* mine, size_t n, int64_t* all) {
  account(c, 1, n * 8);
  if (c->world == 1) {
    std::memcpy(all, mine, n * 8);
    return;
  }
  if (c->callbacks) {
    cb_check(c->cb.allgather(c->cb.user, mine, all, n * 8), "allgather");
    return;
  }
  prism_gpu_ctx* ctx = c->ctx;
  Scratch d(ctx, n * 8 * (size_t)c->world);
  PRISM_HIP(hipMemcpyAsync(d.as<int64_t>() + n * (size_t)c->rank, mine, n * 8, hipMemcpyHostToDevice, ctx->stream));
  PRISM_RCCL(rccl().AllGather(d.as<int64_t>() + n * (size_t)c->rank, d.p, n, ncclInt64, c->nccl, ctx->stream));
  PRISM_HIP(hipMemcpyAsync(all, d.p, n * 8 * (size_t)c->world, hipMemcpyDeviceToHost, ctx->stream));
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
}

// In-place reduction of a device array onto `root`.
void reduce_device(prism_gpu_comm* c, void* dev, size_t n, int type, int op, int root) {
  if (n == 0 || c->world == 1) return;
  account(c, 1, n * 8);
  prism_gpu_ctx* ctx = c->ctx;
  if (c->callbacks) {
    Pinned host(n * 8);
    PRISM_HIP(hipMemcpyAsync(host.p, dev, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    cb_check(c->cb.reduce(c->cb.user, host.p, n, type, op, root), "reduce");
    if (c->rank == root) {
      PRISM_HIP(hipMemcpyAsync(dev, host.p, n * 8, hipMemcpyHostToDevice, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return;
  }
  const ncclRedOp_t rop = op == PRISM_REDUCE_MIN ? ncclMin : op == PRISM_REDUCE_MAX ? ncclMax : ncclSum;
  PRISM_RCCL(rccl().Reduce(dev, dev, n, type == PRISM_REDUCE_F64 ? ncclFloat64 : ncclInt64, rop, root, c->nccl, ctx->stream));
}

// Columns of 8-byte words between all ranks.  Column k of the send block starts at
// send + k * send_stride and holds, in rank order, send_counts[r] words for rank r; column
// k of the receive block starts at recv + k * recv_stride and receives recv_counts[s] words
// from rank s, in rank order.
void alltoallv_columns(prism_gpu_comm* c, int n_cols, const uint64_t* send, uint64_t send_stride,
                       const std::vector<uint64_t>& send_counts, uint64_t* recv, uint64_t recv_stride,
                       const std::vector<uint64_t>& recv_counts) {
  prism_gpu_ctx* ctx = c->ctx;
  const int world = c->world;
  uint64_t send_total = 0, recv_total = 0;
  for (int r = 0; r < world; ++r) { send_total += send_counts[r]; recv_total += recv_counts[r]; }
  account(c, (uint64_t)n_cols, (send_total - send_counts[c->rank]) * 8 * (uint64_t)n_cols);
  if (c->callbacks) {
    Pinned hs(send_total * 8), hr(recv_total * 8);
    for (int k = 0; k < n_cols; ++k) {
      if (send_total) PRISM_HIP(hipMemcpyAsync(hs.p, send + (uint64_t)k * send_stride, send_total * 8, hipMemcpyDeviceToHost, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      cb_check(c->cb.alltoallv(c->cb.user, hs.as<uint64_t>(), send_counts.data(), hr.as<uint64_t>(), recv_counts.data()), "alltoallv");
      if (recv_total) PRISM_HIP(hipMemcpyAsync(recv + (uint64_t)k * recv_stride, hr.p, recv_total * 8, hipMemcpyHostToDevice, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return;
  }
  // one group: every (column, peer) pair is a send and a receive; the rank's own run is a copy
  std::vector<uint64_t> soff(world + 1, 0), roff(world + 1, 0);
  for (int r = 0; r < world; ++r) { soff[r + 1] = soff[r] + send_counts[r]; roff[r + 1] = roff[r] + recv_counts[r]; }
  PRISM_RCCL(rccl().GroupStart());
  for (int k = 0; k < n_cols; ++k) {
    for (int r = 0; r < world; ++r) {
      if (r == c->rank) continue;
      if (send_counts[r])
        PRISM_RCCL(rccl().Send(send + (uint64_t)k * send_stride + soff[r], send_counts[r], ncclUint64, r, c->nccl, ctx->stream));
      if (recv_counts[r])
        PRISM_RCCL(rccl().Recv(recv + (uint64_t)k * recv_stride + roff[r], recv_counts[r], ncclUint64, r, c->nccl, ctx->stream));
    }
  }
  PRISM_RCCL(rccl().GroupEnd());
  if (send_counts[c->rank])
    for (int k = 0; k < n_cols; ++k)
      PRISM_HIP(hipMemcpyAsync(recv + (uint64_t)k * recv_stride + roff[c->rank], send + (uint64_t)k * send_stride + soff[c->rank],
                               send_counts[c->rank] * 8, hipMemcpyDeviceToDevice, ctx->stream));
}

struct StateGuard {
  prism_gpu_state* st = nullptr;
  ~StateGuard() { state_free(st); }
};

int reduce_of(int slot_op) { return slot_op == kSlotMin ? PRISM_REDUCE_MIN : slot_op == kSlotMax ? PRISM_REDUCE_MAX : PRISM_REDUCE_SUM; }

}  // namespace

void comm_unique_id(void* out) {
  ncclUniqueId id;
  PRISM_RCCL(rccl().GetUniqueId(&id));
  std::memcpy(out, &id, sizeof(id));
}

prism_gpu_comm* comm_create_rccl(prism_gpu_ctx* ctx, const void* unique_id, int rank, int world) {
  if (world < 1 || world > 64 || rank < 0 || rank >= world) fail(PRISM_GPU_EINVAL, "rank / world out of range (at most 64 ranks)");
  bind(ctx);
  auto c = std::make_unique<prism_gpu_comm>();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  PRISM_RCCL(rccl().CommInitRank(&c->nccl, world, id, rank));
  return c.release();
}

prism_gpu_comm* comm_create_callbacks(prism_gpu_ctx* ctx, const prism_gpu_transport* t, int rank, int world) {
  if (world < 1 || world > 64 || rank < 0 || rank >= world) fail(PRISM_GPU_EINVAL, "rank / world out of range (at most 64 ranks)");
  if (!t || !t->allgather || !t->reduce || !t->alltoallv) fail(PRISM_GPU_EINVAL, "the transport needs allgather, reduce and alltoallv");
  auto c = std::make_unique<prism_gpu_comm>();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->callbacks = true;
  c->cb = *t;
  return c.release();
}

void comm_destroy(prism_gpu_comm* c) {
  if (!c) return;
  if (c->nccl) {
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    (void)rccl().CommDestroy(c->nccl);
  }
  delete c;
}

void comm_stats(prism_gpu_comm* c, prism_gpu_comm_stats* out) {
  std::lock_guard<std::mutex> g(c->mu);
  *out = c->stats;
}

namespace {

// A rank's own failure between two collectives.  It is kept until the next small all-gather,
// which carries a status word of every rank: a failure on one rank then becomes an error on
// all of them at the same point of the exchange, instead of one rank unwinding while the
// others wait in the next collective for good.
struct LocalStatus {
  int code = 0;
  std::string msg;
  template <class F>
  void run(F&& f) {
    if (code) return;
    try {
      f();
    } catch (const Error& e) {
      code = e.code ? e.code : PRISM_GPU_EINTERNAL;
      msg = e.msg;
    } catch (const std::bad_alloc&) {
      code = PRISM_GPU_EOOM;
      msg = "host memory exhausted";
    } catch (const std::exception& e) {
      code = PRISM_GPU_EINTERNAL;
      msg = std::string("internal error: ") + e.what();
    }
  }
};

// All-gather of n payload words per rank with the rank's status in front.  Returns the
// gathered block ((n + 1) words per rank); throws on every rank when any rank reports a
// failure: the rank's own error where it happened, "rank r failed" elsewhere.
std::vector<int64_t> agree(prism_gpu_comm* c, const LocalStatus& st, const int64_t* mine, size_t n, const char* where) {
  std::vector<int64_t> vec(n + 1, 0), all((n + 1) * (size_t)c->world);
  vec[0] = st.code;
  if (n && !st.code) std::memcpy(vec.data() + 1, mine, n * 8);
  allgather_small(c, vec.data(), n + 1, all.data());
  if (st.code) fail(st.code, st.msg);
  for (int r = 0; r < c->world; ++r) {
    const int64_t code = all[(size_t)r * (n + 1)];
    if (code != 0)
      fail((int)code, "rank " + std::to_string(r) + " of the communicator failed " + where + " (status " + std::to_string(code) +
                          "); the query was given up on every rank");
  }
  return all;
}

// Test hook (prism_gpu_set_option "sharded_fail_phase"): the rank fails on purpose in the
// given local phase of execute_sharded.
void maybe_fail(prism_gpu_ctx* ctx, int phase) {
  if (ctx->sharded_fail_phase.load(std::memory_order_relaxed) == phase)
    fail(PRISM_GPU_EINTERNAL, "failure injected in phase " + std::to_string(phase) + " of the sharded execution");
}

// The exchange of a sharded COUNT(DISTINCT) (every rank holds the distinct records of its
// fine buckets: state_shard): fine bucket (c, sub) belongs to rank c % world; the records are
// laid out owner by owner, the buckets' counts and the records travel in one all-to-all each,
// and the owner counts the union of the runs it received per group.  Returns the owner's state
// (pairs counted per group), or null when this rank failed on the way (`status` says how; the
// caller's next agreement makes that everybody's error).
prism_gpu_state* exchange_distinct_shard(prism_gpu_comm* c, LocalStatus& status, prism_gpu_state* local, prism_gpu_ctx* ctx) {
  const int world = c->world, rank = c->rank;
  DistinctShard* sh = state_shard(local);
  if (!sh) fail(PRISM_GPU_EINTERNAL, "no distinct records to exchange");
  const uint32_t n_fine = sh->n_fine, n_sub = 1u << sh->p2, n_coarse = 1u << sh->p1;
  auto n_own_of = [&](int o) { return (uint64_t)(n_coarse / (uint32_t)world + ((uint32_t)o < n_coarse % (uint32_t)world ? 1u : 0u)) * n_sub; };
  std::vector<uint64_t> block_start(world + 1, 0), cnt_send(world), cnt_recv(world);
  for (int o = 0; o < world; ++o) block_start[o + 1] = block_start[o] + n_own_of(o);
  const uint64_t n_own = n_own_of(rank);
  for (int o = 0; o < world; ++o) { cnt_send[o] = n_own_of(o); cnt_recv[o] = n_own; }
  Scratch counts64, offsets, send;
  std::vector<uint64_t> send_counts(world, 0);
  uint64_t send_total = 0;
  status.run([&] {
    counts64 = Scratch(ctx, (size_t)n_fine * 8 + 8);
    offsets = Scratch(ctx, ((size_t)n_fine + 1) * 8);
    launch_shard_counts(ctx, sh->fine_count, counts64.as<unsigned long long>(), n_fine, (uint32_t)sh->p1, (uint32_t)sh->p2, (uint32_t)world);
    launch_scan_u64(ctx, counts64.as<unsigned long long>(), offsets.as<unsigned long long>(), n_fine);
    std::vector<uint64_t> edge(world + 1, 0);
    for (int o = 0; o <= world; ++o)
      PRISM_HIP(hipMemcpyAsync(&edge[o], offsets.as<uint64_t>() + block_start[o], 8, hipMemcpyDeviceToHost, ctx->stream));
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    for (int o = 0; o < world; ++o) send_counts[o] = edge[o + 1] - edge[o];
    send_total = edge[world];
    send = Scratch(ctx, std::max<uint64_t>(1, send_total) * 8);
    launch_shard_gather(ctx, sh->records.as<unsigned long long>(), sh->fine_base, sh->fine_count, offsets.as<unsigned long long>(),
                        send.as<unsigned long long>(), n_fine, (uint32_t)sh->p1, (uint32_t)sh->p2, (uint32_t)world);
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    collect_timings(ctx);
  });
  // who sends how many records to whom
  std::vector<uint64_t> recv_counts(world, 0), roff(world + 1, 0);
  {
    const std::vector<int64_t> all = agree(c, status, (const int64_t*)send_counts.data(), (size_t)world, "while it dealt its distinct records to their owners");
    for (int s = 0; s < world; ++s) recv_counts[s] = (uint64_t)all[(size_t)s * (world + 1) + 1 + rank];
  }
  for (int s = 0; s < world; ++s) roff[s + 1] = roff[s] + recv_counts[s];
  const uint64_t recv_total = roff[world];
  Scratch recv_cnt64, recv;
  status.run([&] {
    if (recv_total >= (1ull << 32)) fail(PRISM_GPU_EUNSUPPORTED, "more than 2^32 distinct records meet on one rank");
    recv_cnt64 = Scratch(ctx, (size_t)world * std::max<uint64_t>(1, n_own) * 8);
    recv = Scratch(ctx, std::max<uint64_t>(1, recv_total) * 8);
    maybe_fail(ctx, 3);
  });
  agree(c, status, nullptr, 0, "before the all-to-all of its distinct records");
  alltoallv_columns(c, 1, counts64.as<uint64_t>(), n_fine, cnt_send, recv_cnt64.as<uint64_t>(), (uint64_t)world * n_own, cnt_recv);
  alltoallv_columns(c, 1, send.as<uint64_t>(), send_total, send_counts, recv.as<uint64_t>(), recv_total, recv_counts);
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  send.reset();
  prism_gpu_state* merged = nullptr;
  status.run([&] {
    merged = distinct_shard_merge(local, recv.as<uint64_t>(), recv_total, recv_cnt64.as<uint64_t>(), roff.data(), (uint32_t)n_own, world, rank);
  });
  return merged;
}

int execute_sharded_impl(prism_gpu_comm* c, prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids,
                         size_t n_partitions, int root, ArrowArray* out) {
  prism_gpu_ctx* ctx = c->ctx;
  const int world = c->world, rank = c->rank;
  const size_t n_cols = plan->fields.size();

  // ---- 1. the statistics the keys are packed from, agreed with one all-gather ----
  // String columns that travel as dictionary codes -- GROUP BY keys, and the column of a
  // COUNT(DISTINCT), which is packed like a key -- must have the same dictionary on every rank.
  std::vector<int> coded;
  for (const KeyPlan& k : plan->keys)
    if (plan->fields[k.column].is_string) coded.push_back(k.column);
  if (plan->distinct_plan && plan->distinct_col >= 0 && plan->fields[plan->distinct_col].is_string) coded.push_back(plan->distinct_col);
  const size_t vec_n = n_cols * 3 + coded.size() + 1;  // (the last word: the rows this rank scans)
  std::vector<prism_gpu_col_stats> mine(n_cols), agreed(n_cols);
  std::vector<int64_t> vec(vec_n, 0);
  LocalStatus status;
  status.run([&] {
    scan_stats(plan, table, partition_ids, n_partitions, mine.data());
    for (size_t i = 0; i < n_cols; ++i) {
      vec[i * 3] = mine[i].min;
      vec[i * 3 + 1] = mine[i].max;
      vec[i * 3 + 2] = (int64_t)mine[i].nulls;
    }
    for (size_t i = 0; i < coded.size(); ++i) {
      uint64_t digest = 0;
      uint32_t n_values = 0;
      table_dictionary_digest(table, coded[i], &digest, &n_values);
      vec[n_cols * 3 + i] = (int64_t)digest;
    }
    {
      std::lock_guard<std::mutex> g(table->mu);
      uint64_t rows = 0;
      for (const Chunk& ch : table->chunks) {
        bool in = partition_ids == nullptr || n_partitions == 0;
        for (size_t k = 0; k < n_partitions && !in; ++k) in = partition_ids[k] == ch.partition_id;
        if (in) rows += ch.rows;
      }
      vec[vec_n - 1] = (int64_t)rows;
    }
    maybe_fail(ctx, 1);
  });
  uint64_t max_rows = 0;
  {
    const std::vector<int64_t> all = agree(c, status, vec.data(), vec_n, "while it collected its column statistics");
    const size_t stride = vec_n + 1;
    for (size_t i = 0; i < n_cols; ++i) {
      agreed[i] = {INT64_MAX, INT64_MIN, 0};
      for (int r = 0; r < world; ++r) {
        const int64_t* v = all.data() + (size_t)r * stride + 1 + i * 3;
        if (v[0] > v[1]) { agreed[i].nulls += (uint64_t)v[2]; continue; }  // no value on that rank
        agreed[i].min = std::min(agreed[i].min, v[0]);
        agreed[i].max = std::max(agreed[i].max, v[1]);
        agreed[i].nulls += (uint64_t)v[2];
      }
    }
    for (int r = 0; r < world; ++r) max_rows = std::max<uint64_t>(max_rows, (uint64_t)all[(size_t)r * stride + vec_n]);
    // (every rank sees the same digests, so every rank takes this exit or none does)
    for (size_t i = 0; i < coded.size(); ++i)
      for (int r = 1; r < world; ++r)
        if (all[(size_t)r * stride + 1 + n_cols * 3 + i] != all[1 + n_cols * 3 + i])
          fail(PRISM_GPU_EINVAL, "string column `" + plan->fields[coded[i]].name +
                                     "` has different dictionaries on different ranks (its values travel as dictionary codes)");
  }

  // ---- 2. this rank's share ----
  auto t0 = std::chrono::steady_clock::now();
  auto done = [&](int has) {
    std::lock_guard<std::mutex> g(c->mu);
    c->stats.exchange_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return has;
  };
  if (world == 1 && plan->distinct_plan) {
    // one rank: nothing to exchange, and the pairs need no set in HBM
    execute_plan(plan, table, partition_ids, n_partitions, false, out);
    t0 = std::chrono::steady_clock::now();
    return done(1);
  }
  StateGuard local;
  // COUNT(DISTINCT) over many rows: the pairs are partitioned twice on every rank, as on one GPU,
  // and what travels are the DISTINCT records of the fine buckets, each bucket to the rank that
  // owns it; the owner counts the union of what it received.  Whether a rank can do its part
  // depends on the plan, the agreed statistics and the largest scan of any rank -- facts all
  // ranks share -- but it is only known by trying, so the ranks say how it went and take the
  // other way together (the set of pairs as a hash table in HBM) if any of them could not.
  StateGuard shard_work;
  bool shard_done = false;
  if (plan->distinct_plan && ctx->shard_distinct.load(std::memory_order_relaxed) != 0 &&
      max_rows >= (ctx->part_force.load(std::memory_order_relaxed) != 0 ? 1ull : (1ull << 24))) {
    LocalStatus tried;
    DistinctShardRequest req{max_rows};
    tried.run([&] { local.st = aggregate_state(plan, table, partition_ids, n_partitions, agreed.data(), &req); });
    const int64_t ok = tried.code == 0 ? 1 : 0;
    const std::vector<int64_t> all = agree(c, status, &ok, 1, "while it collected its column statistics");
    bool all_ok = true;
    for (int r = 0; r < world; ++r) all_ok = all_ok && all[(size_t)r * 2 + 1] == 1;
    if (all_ok) {
      t0 = std::chrono::steady_clock::now();
      status.run([&] { maybe_fail(ctx, 2); });
      shard_work.st = exchange_distinct_shard(c, status, local.st, ctx);
      shard_done = true;
    } else {
      state_free(local.st);
      local.st = nullptr;
    }
  }
  if (!shard_done) {
    status.run([&] {
      local.st = aggregate_state(plan, table, partition_ids, n_partitions, agreed.data());
      maybe_fail(ctx, 2);
    });
    t0 = std::chrono::steady_clock::now();
  }
  StateView v;
  if (!shard_done) status.run([&] { v = state_view(local.st); });
  if (world == 1) {
    if (status.code) fail(status.code, status.msg);
    state_finish(local.st, out);
    return done(1);
  }

  // ---- 3a. small key space: the accumulator arrays are reduced in place ----
  // (dense or not follows from the agreed statistics: the same on every rank that got this far)
  OwnerParams op{};
  Scratch d_counts, send;
  std::vector<uint64_t> send_counts(world, 0);
  const bool two = !status.code && v.key_words == 2;
  const int width = status.code ? 0 : 1 + (two ? 1 : 0) + v.n_slots;
  const uint64_t n = status.code ? 0 : v.entries;
  uint64_t* send_keys = nullptr;
  if (!shard_done && !status.code && !v.dense) {
    // ---- 3b. large key space: rows to their owners, merge there, the owners' groups to root ----
    // Rows travel in compact column form: keys, high words (two-word keys), one column per slot.
    status.run([&] {
      op.keys = v.keys;
      op.keys_hi = v.keys_hi;
      op.state = v.state;
      op.n = n;
      op.n_slots = v.n_slots;
      op.world = (uint32_t)world;
      // all pairs of a COUNT(DISTINCT) group must meet on one rank: the owner follows from the group alone
      op.owner_of = v.distinct == 1 ? (two ? kOwnerHighWord : kOwnerHighBits) : kOwnerKey;
      op.shift = v.pair_shift;
      d_counts = Scratch(ctx, 8 * 64);
      send = Scratch(ctx, std::max<uint64_t>(1, n) * 8 * (uint64_t)width);
      launch_owner_count(ctx, op, d_counts.as<unsigned long long>());
      std::vector<uint64_t> cursor(world, 0);
      PRISM_HIP(hipMemcpyAsync(send_counts.data(), d_counts.p, 8 * (size_t)world, hipMemcpyDeviceToHost, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      for (int r = 1; r < world; ++r) cursor[r] = cursor[r - 1] + send_counts[r - 1];
      PRISM_HIP(hipMemcpyAsync(d_counts.p, cursor.data(), 8 * (size_t)world, hipMemcpyHostToDevice, ctx->stream));
      send_keys = send.as<uint64_t>();
      uint64_t* const send_hi = two ? send_keys + n : nullptr;
      int64_t* const send_state = (int64_t*)(send_keys + (two ? 2 : 1) * n);
      launch_owner_scatter(ctx, op, d_counts.as<unsigned long long>(), send_keys, send_hi, send_state);
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
    });
  }
  StateGuard owned, outer;
  const uint32_t no_partition = 0xFFFFFFFFu;
  prism_gpu_state* work = nullptr;
  StateView w;
  const bool pairs = shard_done || v.distinct == 1;  // what reaches the root are (group, count) rows of a COUNT(DISTINCT)
  if (shard_done) {
    work = shard_work.st;
    status.run([&] {
      w = state_view(work);
      maybe_fail(ctx, 4);
    });
  } else {
    // who sends how much to whom (all zero from a rank with a dense state); the status of phase 2
    std::vector<int64_t> counts_all;
    {
      const std::vector<int64_t> all = agree(c, status, (const int64_t*)send_counts.data(), (size_t)world, "while it aggregated its partitions");
      counts_all.resize((size_t)world * world);
      for (int s = 0; s < world; ++s)
        std::memcpy(counts_all.data() + (size_t)s * world, all.data() + (size_t)s * (world + 1) + 1, 8 * (size_t)world);
    }
    if (v.dense) {
      for (int s = 0; s < v.n_slots;) {
        int e = s + 1;
        const bool f64 = v.slot_ops[s] == kSlotAddF64;
        while (e < v.n_slots && reduce_of(v.slot_ops[e]) == reduce_of(v.slot_ops[s]) && (v.slot_ops[e] == kSlotAddF64) == f64) ++e;
        // slot arrays lie dense_stride apart, one spare entry between two: a run of slots is one array
        const size_t cnt = (size_t)(e - s) * v.dense_stride - (v.dense_stride - v.entries);
        reduce_device(c, v.dense_state + (uint64_t)s * v.dense_stride, cnt, f64 ? PRISM_REDUCE_F64 : PRISM_REDUCE_I64, reduce_of(v.slot_ops[s]), root);
        s = e;
      }
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      if (rank != root) return done(0);
      state_finish(local.st, out);
      return done(1);
    }

    std::vector<uint64_t> recv_counts(world);
    uint64_t recv_total = 0;
    for (int s = 0; s < world; ++s) recv_total += recv_counts[s] = (uint64_t)counts_all[(size_t)s * world + rank];
    // (the receive block is allocated before the collective: a rank that cannot have it must not
    // leave the others inside the all-to-all, so the allocation's outcome is agreed first)
    Scratch recv;
    status.run([&] {
      recv = Scratch(ctx, std::max<uint64_t>(1, recv_total) * 8 * (uint64_t)width);
      maybe_fail(ctx, 3);
    });
    agree(c, status, nullptr, 0, "before the all-to-all of its group rows");
    alltoallv_columns(c, width, send_keys, n, send_counts, recv.as<uint64_t>(), recv_total, recv_counts);
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    collect_timings(ctx);
    send.reset();

    // the owner's table: same key packing, no rows of its own
    status.run([&] {
      owned.st = aggregate_state(plan, table, &no_partition, 1, agreed.data());
      const uint64_t* rk = recv.as<uint64_t>();
      state_merge_rows(owned.st, rk, two ? rk + recv_total : nullptr, (const int64_t*)(rk + (two ? 2 : 1) * recv_total), recv_total);
      recv.reset();
      // COUNT(DISTINCT): the owner holds every pair of its groups, so it can count them; what goes
      // on to the root are (group, count) rows of groups no other rank has
      work = owned.st;
      if (v.distinct == 1) {
        outer.st = state_distinct_outer(owned.st);
        work = outer.st;
      }
      w = state_view(work);
      maybe_fail(ctx, 4);
    });
  }
  // the owner's groups that can still make the result
  Scratch lead_keys, lead_hi, lead_state;
  int64_t n_lead_i = 0;
  status.run([&] {
    if (!w.dense) n_lead_i = (int64_t)state_leading_rows(work, lead_keys, lead_hi, lead_state);
  });
  std::vector<int64_t> lead_all(world);
  {
    const std::vector<int64_t> all = agree(c, status, &n_lead_i, 1, "while it merged the group rows it owns");
    for (int s = 0; s < world; ++s) lead_all[s] = all[(size_t)s * 2 + 1];
  }
  if (w.dense) {
    for (int s = 0; s < w.n_slots; ++s)
      reduce_device(c, w.dense_state + (uint64_t)s * w.dense_stride, w.entries, PRISM_REDUCE_I64, reduce_of(w.slot_ops[s]), root);
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    if (rank != root) return done(0);
    state_finish(work, out);
    return done(1);
  }

  const bool wtwo = w.key_words == 2;
  const int wwidth = 1 + (wtwo ? 1 : 0) + w.n_slots;
  const uint64_t n_lead = (uint64_t)n_lead_i;
  std::vector<uint64_t> to_root(world, 0), from_all(world, 0);
  to_root[root] = n_lead;
  uint64_t root_total = 0;
  if (rank == root)
    for (int s = 0; s < world; ++s) root_total += from_all[s] = (uint64_t)lead_all[s];
  // keys, high words and slot columns sit in separate blocks: sent as separate column sets
  Scratch gathered;
  status.run([&] {
    gathered = Scratch(ctx, std::max<uint64_t>(1, root_total) * 8 * (uint64_t)wwidth);
    maybe_fail(ctx, 5);
  });
  agree(c, status, nullptr, 0, "before its leading groups went to the root");
  uint64_t* const gk = gathered.as<uint64_t>();
  alltoallv_columns(c, 1, lead_keys.as<uint64_t>(), n_lead, to_root, gk, root_total, from_all);
  if (wtwo) alltoallv_columns(c, 1, lead_hi.as<uint64_t>(), n_lead, to_root, gk + root_total, root_total, from_all);
  alltoallv_columns(c, w.n_slots, lead_state.as<uint64_t>(), n_lead, to_root, gk + (wtwo ? 2 : 1) * root_total, root_total, from_all);
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  if (rank != root) return done(0);
  // (no collective follows: what fails from here on fails on the root alone, and the others are done)
  StateGuard empty_inner, final_state;
  if (pairs) {
    empty_inner.st = aggregate_state(plan, table, &no_partition, 1, agreed.data());
    final_state.st = state_distinct_outer(empty_inner.st);
  } else {
    final_state.st = aggregate_state(plan, table, &no_partition, 1, agreed.data());
  }
  state_merge_rows(final_state.st, gk, wtwo ? gk + root_total : nullptr, (const int64_t*)(gk + (wtwo ? 2 : 1) * root_total), root_total);
  state_finish(final_state.st, out);
  return done(1);
}

}  // namespace

int execute_sharded(prism_gpu_comm* c, prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids,
                    size_t n_partitions, int root, ArrowArray* out) {
  prism_gpu_ctx* ctx = c->ctx;
  if (plan->ctx != ctx || table->ctx != ctx) fail(PRISM_GPU_EINVAL, "communicator, plan and table belong to different contexts");
  if (root < 0 || root >= c->world) fail(PRISM_GPU_EINVAL, "root is not a rank of the communicator");
  if (c->aborted) fail(PRISM_GPU_ERCCL, "the communicator was torn down after a collective failed; create a new one");
  // what the device-resident state does not cover is refused before any collective, and for
  // reasons every rank sees alike (the plan; the key width follows from the agreed statistics)
  if (plan->keys.empty()) fail(PRISM_GPU_EUNSUPPORTED, "the sharded form needs a GROUP BY (ungrouped aggregates exchange PARTIAL batches)");
  if (plan->mixed) fail(PRISM_GPU_EUNSUPPORTED, "COUNT(DISTINCT) next to other aggregates runs on one GPU only");
  if (plan->mode != PRISM_MODE_SINGLE) fail(PRISM_GPU_EINVAL, "the sharded form runs plans in SINGLE mode: it does the merge itself");
  bind(ctx);
  try {
    return execute_sharded_impl(c, plan, table, partition_ids, n_partitions, root, out);
  } catch (const Error& e) {
    // A collective that failed leaves the ranks out of step (some may be inside it still): the
    // RCCL communicator is aborted, which also releases peers blocked in it, and takes no more calls.
    if (e.code == PRISM_GPU_ERCCL && c->nccl && c->world > 1) {
      (void)rccl().CommAbort(c->nccl);
      c->nccl = nullptr;
      c->aborted = true;
    }
    throw;
  }
}

}  // namespace prism
