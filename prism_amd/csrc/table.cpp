// The columnar chunk loader: decoded Arrow columns of a partition are staged
// through a double-buffered pinned host ring into HBM and kept resident.
//
// It takes the place of what ParquetExec hands to FilterExec in the reference
// (src/prism-query/src/providers/table_provider.rs:110-131): per partition, the
// projected columns as Arrow arrays.  On the way in, string columns also get
// table-wide dictionary codes (so GROUP BY / = / LIKE on low-cardinality strings
// run on 4-byte codes), validity bitmaps are re-aligned to the chunk's first row,
// and a statistics kernel records min / max / NULL count per column, which is
// what lets GROUP BY keys be packed into a few bits.
#if defined(__x86_64__)
#include <emmintrin.h>
#endif
#include <chrono>
#include <cstring>

#include "../../include/prism_synth.h"
#include "engine.hpp"

namespace prism {

uint32_t ColumnDict::id_of(const char* s, size_t n) {
  // In front of the map, a direct-mapped table of recent answers, indexed by a cheap hash of the
  // length and the first and last eight bytes and confirmed by comparing the bytes: a plain Utf8
  // column is looked up row by row (the PARTIAL state of a plan with a string key has one row per
  // group, or per distinct pair), and a string built and hashed per row was 20 ns of each.
  uint64_t a = 0, b = 0;
  if (n >= 8) {
    std::memcpy(&a, s, 8);
    std::memcpy(&b, s + n - 8, 8);
  } else {
    std::memcpy(&a, s, n);
  }
  const size_t slot = (size_t)(((a * 0x9E3779B97F4A7C15ull) ^ ((b + n) * 0xC2B2AE3D27D4EB4Full)) >> (64 - kFrontBits));
  if (front.empty()) front.assign((size_t)1 << kFrontBits, 0u);
  if (const uint32_t f = front[slot]) {
    const std::string& v = values[f - 1];
    if (v.size() == n && std::memcmp(v.data(), s, n) == 0) return f - 1;
  }
  std::string key(s, n);
  auto it = ids.find(key);
  uint32_t id;
  if (it != ids.end()) {
    id = it->second;
  } else {
    id = (uint32_t)values.size();
    values.push_back(key);
    ids.emplace(std::move(key), id);
  }
  front[slot] = id + 1;
  return id;
}

namespace {

// Device memory of a table's columns: a plain allocation, or (tables that come and go,
// like the chunks of a streaming partition) a block of the context's scratch pool.
// Requests below kSlabMin are cheap as they are (the runtime cuts them from blocks of its own),
// those of kSlabMax and above cost one runtime call each whatever their size.
constexpr size_t kSlabMin = 256u << 10, kSlabMax = 64u << 20, kSlabFirst = 32u << 20, kSlabLargest = 512u << 20;

void* dev_alloc(prism_gpu_table* t, size_t bytes) {
  // slack: kernels read whole 16-byte vectors (plus one dword) at the tail
  const size_t padded = round_up(bytes, 256) + 256;
  if (t->pooled) return alloc_scratch(t->ctx, padded);
  void* p = nullptr;
  if (padded < kSlabMin || padded >= kSlabMax) {
    PRISM_HIP(hipMalloc(&p, padded));
    return p;
  }
  std::lock_guard<std::mutex> g(t->slab_mu);
  if (t->slabs.empty() || t->slabs.back().size - t->slabs.back().used < padded) {
    // the slab being filled has no room (what is left of it stays unused): the next one is twice as large
    if (!t->slabs.empty() && t->slabs.back().live == 0) {
      (void)hipFree(t->slabs.back().base);
      t->slabs.pop_back();
    }
    size_t size = kSlabFirst;
    for (const auto& sl : t->slabs) size = std::max(size, std::min(kSlabLargest, sl.size * 2));
    size = std::max(size, padded);
    PRISM_HIP(hipMalloc(&p, size));
    prism_gpu_table::Slab sl;
    sl.base = (uint8_t*)p;
    sl.size = size;
    t->slabs.push_back(sl);
  }
  prism_gpu_table::Slab& sl = t->slabs.back();
  p = sl.base + sl.used;
  sl.used += padded;
  sl.live += 1;
  return p;
}

void dev_free(prism_gpu_table* t, void* p) {
  if (!p) return;
  if (t->pooled) { free_scratch(t->ctx, p); return; }
  {
    std::lock_guard<std::mutex> g(t->slab_mu);
    for (size_t i = 0; i < t->slabs.size(); ++i) {
      prism_gpu_table::Slab& sl = t->slabs[i];
      if ((uint8_t*)p < sl.base || (uint8_t*)p >= sl.base + sl.size) continue;
      if (--sl.live != 0) return;
      if (i + 1 == t->slabs.size()) {
        sl.used = 0;  // the slab being filled starts over
      } else {
        (void)hipFree(sl.base);
        t->slabs.erase(t->slabs.begin() + (ptrdiff_t)i);
      }
      return;
    }
  }
  (void)hipFree(p);
}

// Bytes into the pinned ring.  The host never reads them back (the copy engine does), so they
// are written with streaming stores: no line of the ring is fetched before it is overwritten,
// and the caller's working set stays in cache.
inline void copy_into_ring(uint8_t* d, const uint8_t* s, size_t n) {
#if defined(__x86_64__)
  if (n >= 1024) {
    const size_t head = (size_t)(-(uintptr_t)d) & 63;
    std::memcpy(d, s, head);
    d += head; s += head; n -= head;
    const size_t lines = n >> 6;
    for (size_t i = 0; i < lines; ++i, d += 64, s += 64) {
      const __m128i a = _mm_loadu_si128((const __m128i*)s), b = _mm_loadu_si128((const __m128i*)(s + 16)),
                    c = _mm_loadu_si128((const __m128i*)(s + 32)), e = _mm_loadu_si128((const __m128i*)(s + 48));
      _mm_stream_si128((__m128i*)d, a);
      _mm_stream_si128((__m128i*)(d + 16), b);
      _mm_stream_si128((__m128i*)(d + 32), c);
      _mm_stream_si128((__m128i*)(d + 48), e);
    }
    _mm_sfence();
    n &= 63;
  }
#endif
  std::memcpy(d, s, n);
}

// Host -> HBM through the pinned ring: bytes are gathered into a pinned half (they may come
// from many small Arrow buffers) and leave with one copy per destination, or per half, while
// the gathering goes on.  A half is reused only after the copies that read it have completed (its event); the
// ring belongs to one context, and a context stages for one thread at a time (ctx->stage_mu).
struct StageWriter {
  prism_gpu_ctx* ctx;
  uint8_t* dst = nullptr;  // device address the next flushed byte goes to
  int half = -1;             // the ring half being filled (-1: none taken yet, or since the last flush)
  size_t start = 0, fill = 0;  // bytes [start, fill) of it are gathered and not yet on their way
  uint64_t staged = 0;  // bytes handed to the copy stream; the owner adds them to the statistics under ctx->mu
  bool direct = false;  // some bytes travel straight from the caller's (pinned) buffers: await the stream before those may go
  // Those bytes travel on a stream of their own (ctx->direct_stream): tens of megabytes at a
  // time, and whatever waits for a small copy on copy_stream (the reuse of a ring half, a
  // total read back in the middle of an append) would otherwise wait for them as well.
  bool direct_pending = false;  // ... and some of them since new_column(): see join() and reader_stream()
  explicit StageWriter(prism_gpu_ctx* c) : ctx(c) {}
  void new_column() { flush(); direct_pending = false; }
  // In front of kernels on copy_stream that read what was written since new_column().
  void join() {
    flush();
    if (!direct_pending) return;
    PRISM_HIP(hipEventRecord(ctx->stream_join, ctx->direct_stream));
    PRISM_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->stream_join, 0));
    direct_pending = false;
  }
  // The stream for a kernel that reads what was written since new_column() and whose results
  // nothing on copy_stream waits for: the direct stream, once it has seen the small copies.
  hipStream_t reader_stream() {
    flush();
    if (!direct_pending) return ctx->copy_stream;
    PRISM_HIP(hipEventRecord(ctx->stream_join, ctx->copy_stream));
    PRISM_HIP(hipStreamWaitEvent(ctx->direct_stream, ctx->stream_join, 0));
    return ctx->direct_stream;
  }
  // Large buffers that already are page-locked memory of this process (the PARTIAL batches this
  // library hands out are; so is whatever a caller allocated with hipHostMalloc) need no second
  // host copy: the copy engine reads them where they are.
  static constexpr size_t kDirectMinBytes = 1u << 20, kKernelCopyMaxBytes = 8u << 20, kKernelCopyAlwaysBytes = 1u << 20;
  static bool page_locked(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
      (void)hipGetLastError();  // (ordinary host memory is reported as an error: not one of ours)
      return false;
    }
    return at.type == hipMemoryTypeHost;
  }
  void begin(void* device_dst) { flush(); dst = (uint8_t*)device_dst; }
  // The half being filled is filled on, flush after flush, until it is full: only then is its
  // event recorded and the other half taken (whose event, by then, is long past -- waiting for
  // an event that has not completed costs tens of microseconds however short the wait).
  void open() {
    if (ctx->stage_cur < 0) ctx->stage_cur = 0;
    half = ctx->stage_cur;
    start = fill = ctx->stage_fill;
    if (fill == ctx->stage_bytes) next_half();
  }
  void next_half() {
    PRISM_HIP(hipEventRecord(ctx->stage_done[half], ctx->copy_stream));
    ctx->stage_used[half] = true;
    half ^= 1;
    if (ctx->stage_used[half]) PRISM_HIP(hipEventSynchronize(ctx->stage_done[half]));
    ctx->stage_cur = half;
    ctx->stage_fill = start = fill = 0;
  }
  void write(const void* src, size_t bytes) {
    const uint8_t* s = (const uint8_t*)src;
    if (bytes >= kDirectMinBytes && page_locked(src)) {
      flush();
      PRISM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->direct_stream));
      ctx->direct_seen = true;
      direct_pending = true;
      staged += bytes;
      dst += bytes;
      direct = true;
      return;
    }
    while (bytes) {
      if (half < 0) open();
      const size_t room = ctx->stage_bytes - fill, n = bytes < room ? bytes : room;
      copy_into_ring((uint8_t*)ctx->stage[half] + fill, s, n);
      fill += n;
      s += n;
      bytes -= n;
      if (fill == ctx->stage_bytes) {
        send();
        next_half();
      }
    }
  }
  // queues the copy of what has been gathered since the last one
  void send() {
    if (fill == start) return;
    const uint8_t* from = (const uint8_t*)ctx->stage[half] + start;
    // Small amounts are fetched by a kernel: as commands of the copy engine they would wait
    // their turn behind the direct copies, and so would whoever awaits copy_stream.
    // (Megabytes at a time are the copy engine's to move when nothing else occupies it: the
    // chunks of a streaming feed, all of which come through the ring.)
    const size_t by_kernel = ctx->direct_seen ? kKernelCopyMaxBytes : kKernelCopyAlwaysBytes;
    if (fill - start <= by_kernel && (((uintptr_t)dst ^ (uintptr_t)from) & 15) == 0)
      launch_stage_copy(ctx, dst, from, fill - start, ctx->copy_stream);
    else
      PRISM_HIP(hipMemcpyAsync(dst, from, fill - start, hipMemcpyHostToDevice, ctx->copy_stream));
    staged += fill - start;
    dst += fill - start;
    start = fill;
  }
  void flush() {
    if (half < 0) return;
    send();
    // (the next gather starts on a line of its own)
    ctx->stage_fill = std::min(ctx->stage_bytes, round_up(fill, 256));
    half = -1;
  }
};

void stage_copy(prism_gpu_ctx* ctx, void* dst, const void* src, size_t bytes) {
  std::lock_guard<std::mutex> g(ctx->stage_mu);
  ensure_stage_ring(ctx);
  StageWriter w(ctx);
  w.begin(dst);
  w.write(src, bytes);
  w.join();  // (callers order their kernels on copy_stream)
  std::lock_guard<std::mutex> gs(ctx->mu);
  ctx->stats.bytes_staged += w.staged;
}

// Device scratch of the loader's kernels, at least `need` bytes (under ctx->stage_mu).  One
// block per context, grown on demand: a column's use follows the previous one's on copy_stream,
// and what the host writes into it travels on that stream too.
void* load_scratch(prism_gpu_ctx* ctx, size_t need) {
  if (ctx->load_tmp_bytes < need) {
    // (hipFree waits for the device: the kernels of earlier appends no longer read the old block)
    if (ctx->load_tmp) PRISM_HIP(hipFree(ctx->load_tmp));
    ctx->load_tmp = nullptr;
    ctx->load_tmp_bytes = 0;
    const size_t grown = pow2_ceil(need < 4096 ? 4096 : need);
    hipError_t e = hipMalloc(&ctx->load_tmp, grown);
    if (e != hipSuccess) fail(e == hipErrorOutOfMemory ? PRISM_GPU_EOOM : PRISM_GPU_EHIP, std::string("hipMalloc of the loader's scratch failed: ") + hipGetErrorString(e));
    ctx->load_tmp_bytes = grown;
  }
  return ctx->load_tmp;
}

bool bit_at(const uint8_t* bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }

// Validity bits [offset, offset + n) re-aligned to bit 0, as 64-bit words.
std::vector<uint64_t> realign_validity(const uint8_t* bits, int64_t offset, int64_t n) {
  std::vector<uint64_t> out((size_t)ceil_div((uint64_t)n, kTileRows) * kTileWords, 0);  // whole tiles: kernels read a tile's words even past the last row
  if ((offset & 7) == 0) {
    std::memcpy(out.data(), bits + (offset >> 3), (size_t)ceil_div((uint64_t)n, 8));
    if (n & 63) out[(size_t)(n >> 6)] &= (1ull << (n & 63)) - 1ull;
  } else {
    for (int64_t i = 0; i < n; ++i)
      if (bit_at(bits, offset + i)) out[(size_t)(i >> 6)] |= 1ull << (i & 63);
  }
  return out;
}

struct StrView {
  const char* p;
  size_t n;
};

}  // namespace

}  // namespace prism

namespace prism {

void free_chunk_memory(prism_gpu_table* t, Chunk& ch) {
  for (auto& c : ch.cols) {
    dev_free(t, c.values);
    dev_free(t, c.validity);
    dev_free(t, c.bytes);
    dev_free(t, c.codes);
    c = DeviceColumn();
  }
  dev_free(t, ch.d_stats);
  ch.d_stats = nullptr;
}

}  // namespace prism

prism_gpu_table::~prism_gpu_table() {
  if (ctx) (void)hipSetDevice(ctx->device);
  // copies and statistics kernels of recent appends may still be reading or writing the buffers
  if (ctx) (void)prism::sync_copy_streams(ctx);
  for (auto& ch : chunks) prism::free_chunk_memory(this, ch);
  for (auto& sl : slabs) (void)hipFree(sl.base);
}

namespace prism {

namespace {

void init_stats(prism_gpu_table* t, Chunk& ch, size_t n_cols) {
  ch.stats.assign(n_cols, ColStats{0x7fffffffffffffffll, -0x7fffffffffffffffll - 1, 0});
  ch.d_stats = (ColStats*)dev_alloc(t, n_cols * sizeof(ColStats));
  stage_copy(t->ctx, ch.d_stats, ch.stats.data(), n_cols * sizeof(ColStats));
}

}  // namespace

namespace {

// Appends bits [offset, offset + n) of `bits` (nullptr: all ones) at bit position `at` of `out`.
void append_bits(std::vector<uint64_t>& out, uint64_t at, const uint8_t* bits, int64_t offset, int64_t n) {
  if (n <= 0) return;
  if (!bits) {
    for (int64_t i = 0; i < n;) {
      const uint64_t pos = at + (uint64_t)i, room = 64 - (pos & 63), take = (uint64_t)(n - i) < room ? (uint64_t)(n - i) : room;
      const uint64_t mask = take == 64 ? ~0ull : (((1ull << take) - 1ull) << (pos & 63));
      out[(size_t)(pos >> 6)] |= mask;
      i += (int64_t)take;
    }
    return;
  }
  if ((at & 7) == 0 && (offset & 7) == 0) {
    uint8_t* o = (uint8_t*)out.data() + (at >> 3);
    const uint8_t* in = bits + (offset >> 3);
    const int64_t whole = n >> 3;
    std::memcpy(o, in, (size_t)whole);
    for (int64_t i = whole * 8; i < n; ++i)
      if (bit_at(bits, offset + i)) out[(size_t)((at + (uint64_t)i) >> 6)] |= 1ull << ((at + (uint64_t)i) & 63);
    return;
  }
  for (int64_t i = 0; i < n; ++i)
    if (bit_at(bits, offset + i)) out[(size_t)((at + (uint64_t)i) >> 6)] |= 1ull << ((at + (uint64_t)i) & 63);
}

}  // namespace

// One chunk from one or more RecordBatches (the batches of a streaming partition arrive
// 8192 rows at a time; a chunk of them is staged and scanned as one).
void table_append_batches(prism_gpu_table* t, uint32_t partition_id, ArrowArray* const* batches, size_t n_batches, bool take,
                          hipEvent_t* pending) {
  prism_gpu_ctx* ctx = t->ctx;
  bind(ctx);
  if (pending) {
    *pending = nullptr;
    if (take) fail(PRISM_GPU_EINTERNAL, "an append that hands its copies' event over cannot take the batches");
  }
  struct Releaser {
    ArrowArray* const* b;
    size_t n;
    bool take;
    ~Releaser() {
      if (!take) return;
      for (size_t i = 0; i < n; ++i)
        if (b[i] && b[i]->release) b[i]->release(b[i]);
    }
  } releaser{batches, n_batches, take};

  int64_t n = 0;
  std::vector<std::vector<ArrowArray>> views(n_batches);
  for (size_t b = 0; b < n_batches; ++b) {
    const ArrowArray* batch = batches[b];
    if (!batch) fail(PRISM_GPU_EINVAL, "batch is null");
    if (batch->n_children != (int64_t)t->fields.size())
      fail(PRISM_GPU_EINVAL, "batch has " + std::to_string(batch->n_children) + " columns, table schema has " +
                                 std::to_string(t->fields.size()));
    if (batch->offset < 0 || batch->length < 0) fail(PRISM_GPU_EINVAL, "batch has a negative offset or length");
    for (size_t ci = 0; ci < t->fields.size(); ++ci)
      if (batch->children[ci]->length < batch->offset + batch->length)
        fail(PRISM_GPU_EINVAL, "column " + t->fields[ci].name + " is shorter than the batch");
    // A struct array's offset and length select rows [offset, offset + length) of every child
    // (on top of the child's own offset): the columns are read through adjusted views.
    views[b].resize(t->fields.size());
    for (size_t ci = 0; ci < t->fields.size(); ++ci) {
      views[b][ci] = *batch->children[ci];
      views[b][ci].offset += batch->offset;
      views[b][ci].length = batch->length;
    }
    n += batch->length;
  }
  if (n == 0) return;
  // (prism_gpu_table_append cuts larger batches up itself; the streaming path gathers about 2^20 rows per chunk)
  if (n > 0x1fffffff) fail(PRISM_GPU_EUNSUPPORTED, "a chunk may hold at most 2^29-1 rows");

  auto t0 = std::chrono::steady_clock::now();
  // Bytes that travel straight from the batches' own (page-locked) buffers must have left them before
  // the batches go: an event behind this append's work, waited for once the locks are released --
  // callers that append from several threads then queue their copies while this one's are under way.
  struct DirectDone {
    hipEvent_t ev = nullptr;
    hipEvent_t* hand_over = nullptr;  // (set once the append has succeeded)
    ~DirectDone() {
      if (!ev) return;
      if (hand_over) {
        *hand_over = ev;
        return;
      }
      (void)hipEventSynchronize(ev);
      (void)hipEventDestroy(ev);
    }
  } direct_done;  // (declared in front of the locks: its wait runs after they are released, and before the batches' release)
  std::unique_lock<std::mutex> g(t->mu);
  Chunk ch;
  ch.partition_id = partition_id;
  ch.rows = (uint64_t)n;
  ch.cols.resize(t->fields.size());
  // the chunk owns its device memory from here on: on failure free it
  struct ChunkGuard {
    prism_gpu_table* t;
    Chunk* ch;
    bool armed = true;
    ~ChunkGuard() {
      if (!armed) return;
      (void)sync_copy_streams(t->ctx);  // copies into the chunk may be under way
      free_chunk_memory(t, *ch);
    }
  } guard{t, &ch};
  init_stats(t, ch, t->fields.size());

  std::unique_lock<std::mutex> stage_lock(ctx->stage_mu);
  HostTimer locked_timer(ctx, "host:append_locked");  // (the part of an append that callers on other threads wait for)
  ensure_stage_ring(ctx);
  StageWriter w(ctx);
  // String columns first: theirs are the small copies (and, for packed indices, a wait for a
  // total from the device) that should not stand behind this chunk's own blocks of values.
  for (size_t turn = 0; turn < 2 * t->fields.size(); ++turn) {
    const size_t ci = turn % t->fields.size();
    const FieldType& f = t->fields[ci];
    if (f.is_string != (turn < t->fields.size())) continue;
    DeviceColumn& dc = ch.cols[ci];
    w.new_column();
    // ---- validity: one bitmap for the chunk, present only if some row is NULL ----
    bool any_null = false;
    for (size_t b = 0; b < n_batches && !any_null; ++b) {
      const ArrowArray* a = &views[b][ci];
      any_null = a->null_count != 0 && a->n_buffers > 0 && a->buffers[0] != nullptr && a->length > 0;
    }
    if (any_null) {
      std::vector<uint64_t> words((size_t)ceil_div((uint64_t)n, kTileRows) * kTileWords, 0);  // whole tiles: kernels read a tile's words even past the last row
      uint64_t at = 0;
      for (size_t b = 0; b < n_batches; ++b) {
        const ArrowArray* a = &views[b][ci];
        const uint8_t* vbits = (a->null_count != 0 && a->n_buffers > 0) ? (const uint8_t*)a->buffers[0] : nullptr;
        append_bits(words, at, vbits, a->offset, a->length);
        at += (uint64_t)a->length;
      }
      any_null = false;
      for (int64_t i = 0; i < n && !any_null; i += 64) {
        const uint64_t want = n - i >= 64 ? ~0ull : ((1ull << (n - i)) - 1ull);
        any_null = (words[(size_t)(i >> 6)] & want) != want;
      }
      if (any_null) {
        dc.validity = (uint64_t*)dev_alloc(t, words.size() * 8);
        w.begin(dc.validity);
        w.write(words.data(), words.size() * 8);
      }
    }
    if (!f.is_string) {
      dc.values = dev_alloc(t, (size_t)n * f.width);
      w.begin(dc.values);
      for (size_t b = 0; b < n_batches; ++b) {
        const ArrowArray* a = &views[b][ci];
        if (a->dictionary) fail(PRISM_GPU_EUNSUPPORTED, "dictionary encoding is only supported for string columns");
        if (a->length == 0) continue;
        if (a->n_buffers < 2 || !a->buffers[1]) fail(PRISM_GPU_EINVAL, "column " + f.name + " has no value buffer");
        w.write((const uint8_t*)a->buffers[1] + (size_t)a->offset * f.width, (size_t)a->length * f.width);
      }
      launch_col_stats(ctx, dc.values, dc.validity, f.width, f.is_unsigned, (uint64_t)n, &ch.d_stats[ci], w.reader_stream());
      continue;
    }
    // ---- string column: raw representation + table-wide dictionary codes ----
    ColumnDict& dict = t->dicts[ci];
    std::vector<uint32_t> codes;
    // Chunks of a streaming partition whose every batch is dictionary encoded keep only the
    // codes (the table-wide dictionary of such a table stays small: it is the union of the
    // batches' dictionaries).
    bool codes_only = t->pooled && dict.alive;
    for (const Chunk& old : t->chunks) codes_only = codes_only && old.cols[ci].values == nullptr;  // (the table of a part holds the chunks in flight)
    for (size_t b = 0; b < n_batches && codes_only; ++b) {
      const ArrowArray* a = &views[b][ci];
      codes_only = a->length == 0 || (a->dictionary != nullptr && a->dictionary->length <= 4096);
    }
    // Input that is dictionary encoded throughout is not expanded here: its indices and
    // dictionaries are staged, and kernels write the offsets, the bytes and the codes.
    bool device_expand = !codes_only && ctx->dict_expand_device.load(std::memory_order_relaxed) != 0;
    for (size_t b = 0; b < n_batches && device_expand; ++b) device_expand = views[b][ci].length == 0 || views[b][ci].dictionary != nullptr;
    std::vector<uint32_t> index, cat_off, remap_all;  // (device_expand) rows' positions in the dictionaries laid end to end
    // A codes-only chunk takes its codes on the device as well: the batches' index buffers are
    // staged as they are, and a kernel maps them through the dictionaries' codes.
    const bool device_codes = codes_only && ctx->dict_expand_device.load(std::memory_order_relaxed) != 0;
    std::vector<uint32_t> seg_row, seg_base;
    // Batches of the library's Parquet reader whose indices are still the file's RLE / bit-packed
    // runs (device_expand): runs and packed bytes are staged, and a kernel writes the indices.
    struct PackedBatch { const PackedIndices* pk; uint64_t row0, rows; uint32_t base, dict_len; };
    std::vector<PackedBatch> packed_batches;
    uint64_t packed_bound = 0;  // no more bytes than this come out of the packed batches
    const ArrowArray* seg_dict = nullptr;  // the dictionary of the last segment: batches that share it extend the segment
    // Plain strings of a column whose dictionary exists already: the codes are looked up on the
    // device first (a later chunk of a low-cardinality column holds nothing new), and only a chunk
    // with a string the dictionary lacks is coded here, row by row.  A column that keeps bringing
    // new strings is left alone for a while after each miss.
    bool dev_lookup = dict.alive && !codes_only && !device_expand && dict.values.size() <= 16384 &&
                      n >= ctx->dict_lookup_min_rows.load(std::memory_order_relaxed);
    for (size_t b = 0; b < n_batches && dev_lookup; ++b) dev_lookup = views[b][ci].dictionary == nullptr;
    if (dev_lookup && dict.lookup_pause > 0) {
      --dict.lookup_pause;
      dev_lookup = false;
    }
    if (dev_lookup && dict.values.empty()) {
      // The first chunk of a table: a sample of its rows seeds the dictionary, and the kernel finds
      // out whether their strings were all there is.
      const int64_t step = std::max<int64_t>(1, n / 8192);  // (spread over the chunk: rows of one value often lie together)
      for (size_t b = 0; b < n_batches; ++b) {
        const ArrowArray* a = &views[b][ci];
        if (a->length == 0) continue;
        if (a->n_buffers < 3 || !a->buffers[1]) fail(PRISM_GPU_EINVAL, "string column " + f.name + " has no offsets");
        const uint8_t* vbits = (a->null_count != 0 && a->n_buffers > 0) ? (const uint8_t*)a->buffers[0] : nullptr;
        const int32_t* so = (const int32_t*)a->buffers[1] + a->offset;
        const uint8_t* sb = (const uint8_t*)a->buffers[2];
        for (int64_t i = 0; i < a->length; i += step)
          if (!vbits || bit_at(vbits, a->offset + i)) (void)dict.id_of((const char*)sb + so[i], (size_t)(so[i + 1] - so[i]));
      }
      dev_lookup = !dict.values.empty();
    }
    if (dict.alive && !device_expand && !device_codes && !dev_lookup) codes.resize((size_t)n);
    std::vector<int32_t> offs(codes_only || device_expand ? 1 : (size_t)n + 1);
    offs[0] = 0;
    // bytes of the chunk: total first (it sizes the device buffer), then the pieces
    uint64_t total = 0;
    struct Piece { const uint8_t* p; size_t n; };
    std::vector<Piece> pieces;
    std::vector<std::vector<uint8_t>> expanded;  // raw bytes of dictionary-encoded input
    int64_t row0 = 0;
    for (size_t b = 0; b < n_batches; ++b) {
      const ArrowArray* a = &views[b][ci];
      const int64_t nb = a->length;
      if (nb == 0) continue;
      const uint8_t* vbits = (a->null_count != 0 && a->n_buffers > 0) ? (const uint8_t*)a->buffers[0] : nullptr;
      if (a->dictionary) {
        // Int32 indices into a Utf8 / Binary dictionary (the schema declares the index type
        // when the column is declared dictionary encoded; other index widths are not accepted)
        const ArrowArray* d = a->dictionary;
        const PackedIndices* pk = parquet_packed_indices(a);
        if (a->n_buffers < 2 || (!a->buffers[1] && !pk) || d->n_buffers < 3 || !d->buffers[1])
          fail(PRISM_GPU_EINVAL, "dictionary-encoded column " + f.name + " lacks a buffer");
        const int32_t* idx = pk ? nullptr : (const int32_t*)a->buffers[1] + a->offset;
        std::vector<int32_t> unpacked;
        auto need_idx = [&]() {  // (packed indices that no kernel will expand)
          if (idx) return;
          packed_indices_expand(*pk, a->offset + nb, unpacked);
          idx = unpacked.data() + a->offset;
        };
        const int32_t* doff = (const int32_t*)d->buffers[1] + d->offset;
        const char* dbytes = (const char*)d->buffers[2];
        const uint8_t* dvalid = (d->null_count != 0 && d->n_buffers > 0) ? (const uint8_t*)d->buffers[0] : nullptr;
        if (dvalid) fail(PRISM_GPU_EUNSUPPORTED, "NULL entries inside a dictionary are not supported (column " + f.name + ")");
        if (device_codes) {
          need_idx();
          const bool same = seg_dict && seg_dict->length == d->length && seg_dict->offset == d->offset &&
                            seg_dict->buffers[1] == d->buffers[1] && seg_dict->buffers[2] == d->buffers[2];
          if (!same) {
            seg_row.push_back((uint32_t)row0);
            seg_base.push_back((uint32_t)remap_all.size());
            for (int64_t k = 0; k < d->length; ++k) remap_all.push_back(dict.id_of(dbytes + doff[k], (size_t)(doff[k + 1] - doff[k])));
            seg_dict = d;
          }
          // every index a row uses must lie inside the dictionary (rows without NULLs: one pass for the extremes)
          if (!vbits) {
            int32_t lo = 0, hi = -1;
            for (int64_t i = 0; i < nb; ++i) { lo = idx[i] < lo ? idx[i] : lo; hi = idx[i] > hi ? idx[i] : hi; }
            if (lo < 0 || hi >= d->length) fail(PRISM_GPU_EINVAL, "dictionary index out of range in " + f.name);
          } else {
            for (int64_t i = 0; i < nb; ++i)
              if (bit_at(vbits, a->offset + i) && (idx[i] < 0 || idx[i] >= d->length)) fail(PRISM_GPU_EINVAL, "dictionary index out of range in " + f.name);
          }
          pieces.push_back({(const uint8_t*)idx, (size_t)nb * 4});
          row0 += nb;
          continue;
        }
        std::vector<uint32_t> remap;
        if (dict.alive) {
          remap.resize((size_t)d->length);
          for (int64_t k = 0; k < d->length; ++k) remap[(size_t)k] = dict.id_of(dbytes + doff[k], (size_t)(doff[k + 1] - doff[k]));
        }
        if (device_expand) {
          if (cat_off.empty()) cat_off.push_back(0);
          const uint64_t base = cat_off.size() - 1;
          if (base + (uint64_t)d->length >= 0xffffffffull) fail(PRISM_GPU_EUNSUPPORTED, "the dictionaries of one chunk of " + f.name + " hold 2^32 entries or more");
          uint64_t longest = 0;
          for (int64_t k = 0; k < d->length; ++k) {
            if (doff[k + 1] < doff[k]) fail(PRISM_GPU_EINVAL, "dictionary offsets of " + f.name + " decrease");
            longest = std::max<uint64_t>(longest, (uint64_t)(doff[k + 1] - doff[k]));
            const uint64_t end = (uint64_t)cat_off.back() + (uint64_t)(doff[k + 1] - doff[k]);
            if (end > 0x7fffffffull) fail(PRISM_GPU_EUNSUPPORTED, "the dictionaries of one chunk of " + f.name + " exceed 2 GiB");
            cat_off.push_back((uint32_t)end);
          }
          if (d->length > 0 && doff[d->length] > doff[0]) pieces.push_back({(const uint8_t*)dbytes + doff[0], (size_t)(doff[d->length] - doff[0])});
          remap_all.insert(remap_all.end(), remap.begin(), remap.end());
          // (the kernels sum the lengths in 32 bits: the packed batches stay packed while their
          // bytes cannot pass the chunk's limit whatever the indices are)
          if (pk && !vbits && a->offset == 0 && total + packed_bound + (uint64_t)nb * longest <= 0x7fffffffull) {
            packed_bound += (uint64_t)nb * longest;
            packed_batches.push_back({pk, (uint64_t)row0, (uint64_t)nb, (uint32_t)base, (uint32_t)d->length});
            row0 += nb;
            continue;
          }
          need_idx();
          if (index.empty()) index.resize((size_t)n);  // (rows of packed batches: written on the device, after this has been copied)
          const uint32_t* len_at = cat_off.data() + base;
          for (int64_t i = 0; i < nb; ++i) {
            const bool valid = !vbits || bit_at(vbits, a->offset + i);
            if (!valid) { index[(size_t)(row0 + i)] = 0xffffffffu; continue; }
            const int32_t k = idx[i];
            if (k < 0 || k >= d->length) fail(PRISM_GPU_EINVAL, "dictionary index out of range in " + f.name);
            index[(size_t)(row0 + i)] = (uint32_t)(base + (uint64_t)k);
            total += len_at[k + 1] - len_at[k];
          }
          if (total + packed_bound > 0x7fffffffull) fail(PRISM_GPU_EUNSUPPORTED, "string data of one chunk exceeds 2 GiB");
          row0 += nb;
          continue;
        }
        need_idx();
        if (codes_only) {
          // a streaming chunk whose input is dictionary encoded: the codes are all the kernels
          // need (no per-row expansion of the strings on the host)
          for (int64_t i = 0; i < nb; ++i) {
            const bool valid = !vbits || bit_at(vbits, a->offset + i);
            const int32_t k = valid ? idx[i] : 0;
            if (valid && (k < 0 || k >= d->length)) fail(PRISM_GPU_EINVAL, "dictionary index out of range in " + f.name);
            codes[(size_t)(row0 + i)] = valid ? remap[(size_t)k] : 0u;
          }
          row0 += nb;
          continue;
        }
        expanded.emplace_back();
        std::vector<uint8_t>& ex = expanded.back();
        for (int64_t i = 0; i < nb; ++i) {
          const bool valid = !vbits || bit_at(vbits, a->offset + i);
          if (valid) {
            const int32_t k = idx[i];
            if (k < 0 || k >= d->length) fail(PRISM_GPU_EINVAL, "dictionary index out of range in " + f.name);
            ex.insert(ex.end(), dbytes + doff[k], dbytes + doff[k + 1]);
            if (dict.alive) codes[(size_t)(row0 + i)] = remap[(size_t)k];
          } else if (dict.alive) {
            codes[(size_t)(row0 + i)] = 0;
          }
          if (total + ex.size() > 0x7fffffffull) fail(PRISM_GPU_EUNSUPPORTED, "string data of one chunk exceeds 2 GiB");
          offs[(size_t)(row0 + i) + 1] = (int32_t)(total + ex.size());
        }
        total += ex.size();
        pieces.push_back({ex.data(), ex.size()});
      } else {
        if (a->n_buffers < 3 || !a->buffers[1]) fail(PRISM_GPU_EINVAL, "string column " + f.name + " has no offsets");
        const int32_t* so = (const int32_t*)a->buffers[1] + a->offset;
        const uint8_t* sb = (const uint8_t*)a->buffers[2];
        const int32_t base = so[0];
        const uint64_t len = (uint64_t)(so[nb] - base);
        if (total + len > 0x7fffffffull) fail(PRISM_GPU_EUNSUPPORTED, "string data of one chunk exceeds 2 GiB");
        for (int64_t i = 1; i <= nb; ++i) offs[(size_t)(row0 + i)] = (int32_t)(total + (uint64_t)(so[i] - base));
        if (dict.alive && !dev_lookup) {
          for (int64_t i = 0; i < nb; ++i) {
            const bool valid = !vbits || bit_at(vbits, a->offset + i);
            codes[(size_t)(row0 + i)] = valid ? dict.id_of((const char*)sb + so[i], (size_t)(so[i + 1] - so[i])) : 0;
            if (dict.values.size() > kMaxDictValues) break;
          }
        }
        if (len) pieces.push_back({sb + base, (size_t)len});
        total += len;
      }
      row0 += nb;
    }
    auto retire_full_dictionary = [&]() {
      if (!dict.alive || dict.values.size() <= kMaxDictValues) return;
      // too many distinct values: this column keeps only its raw form from now on
      // (the values stay: group tables built so far hold their codes and are still read out through them)
      dict.alive = false;
      dict.device_built = false;
      dict.ids.clear();
      dict.front.clear();
      for (auto& old : t->chunks) {
        dev_free(t, old.cols[ci].codes);
        old.cols[ci].codes = nullptr;
      }
    };
    retire_full_dictionary();
    if (codes_only && !dict.alive) fail(PRISM_GPU_EUNSUPPORTED, "dictionary-encoded column " + f.name + " has more than " + std::to_string(kMaxDictValues) + " distinct values");
    if (device_codes) {
      // (pieces holds the batches' index buffers)
      auto pad = [](size_t v) { return (v + 15) & ~(size_t)15; };
      const size_t n_segs = seg_row.size(), base_at = pad(n_segs * 4), remap_at = base_at + pad(n_segs * 4),
                   need = remap_at + pad(remap_all.size() * 4);
      uint8_t* tmp = (uint8_t*)load_scratch(ctx, need);
      dc.codes = (uint32_t*)dev_alloc(t, (size_t)n * 4);
      w.begin(tmp);
      w.write(seg_row.data(), n_segs * 4);
      w.begin(tmp + base_at);
      w.write(seg_base.data(), n_segs * 4);
      w.begin(tmp + remap_at);
      w.write(remap_all.data(), remap_all.size() * 4);
      w.begin(dc.codes);
      for (const Piece& pc : pieces) w.write(pc.p, pc.n);
      w.join();
      launch_dict_codes(ctx, dc.codes, dc.validity, (uint64_t)n, (const uint32_t*)tmp, (const uint32_t*)(tmp + base_at), (uint32_t)n_segs,
                        (const uint32_t*)(tmp + remap_at), ctx->copy_stream);
      launch_col_stats(ctx, dc.codes, dc.validity, 4, true, (uint64_t)n, &ch.d_stats[ci], ctx->copy_stream);
      continue;
    }
    if (device_expand) {
      // (pieces holds the dictionaries' bytes here, total the bytes of the expanded column)
      const size_t n_ent = cat_off.size() - 1, n_blocks = dict_expand_blocks((uint64_t)n);
      auto pad = [](size_t v) { return (v + 15) & ~(size_t)15; };
      size_t packed_need = 16;  // (a flag and the bytes' total; then every packed batch's runs and bytes)
      uint64_t packed_rows = 0;
      for (const PackedBatch& pb : packed_batches) {
        packed_need += pad(pb.pk->n_runs * sizeof(PackedRun)) + pad(pb.pk->n_bytes);
        packed_rows += pb.rows;
      }
      // What the host provides lies end to end at the front of the scratch and leaves with one
      // copy: the dictionaries' offsets, their codes, their bytes, a flag and a spare word (zero),
      // every packed batch's runs and bytes.  What only kernels write follows.
      const size_t remap_bytes = dict.alive ? n_ent * 4 : 0;
      const size_t off_at = 0, remap_at = off_at + pad(cat_off.size() * 4), dbytes_at = remap_at + pad(remap_bytes),
                   packed_at = dbytes_at + pad((size_t)cat_off.back()), sums_at = packed_at + packed_need,
                   index_at = sums_at + pad(n_blocks * 4), need = index_at + pad(dict.alive ? 0 : (size_t)n * 4);
      uint8_t* tmp = (uint8_t*)load_scratch(ctx, need);
      dc.values = dev_alloc(t, ((size_t)n + 1) * 4);
      if (packed_batches.empty()) {
        dc.bytes = (uint8_t*)dev_alloc(t, (size_t)total);
        dc.n_bytes = total;
      }
      if (dict.alive) dc.codes = (uint32_t*)dev_alloc(t, (size_t)n * 4);
      static const uint8_t zeros[16] = {0};
      size_t pos = 0;
      auto put = [&](const void* p, size_t bytes) {  // ... and zeros up to the next multiple of 16
        if (bytes) w.write(p, bytes);
        pos += bytes;
        if (pad(pos) != pos) {
          w.write(zeros, pad(pos) - pos);
          pos = pad(pos);
        }
      };
      w.begin(tmp);
      put(cat_off.data(), cat_off.size() * 4);
      if (dict.alive) put(remap_all.data(), n_ent * 4);
      for (const Piece& pc : pieces) {
        w.write(pc.p, pc.n);
        pos += pc.n;
      }
      put(nullptr, 0);
      put(zeros, 16);
      std::vector<size_t> runs_at;
      for (const PackedBatch& pb : packed_batches) {
        runs_at.push_back(pos);
        put(pb.pk->runs, pb.pk->n_runs * sizeof(PackedRun));
        put(pb.pk->bytes, pb.pk->n_bytes);
      }
      if (pos != sums_at) fail(PRISM_GPU_EINTERNAL, "the loader's scratch is laid out otherwise than it was sized");
      uint32_t* d_index = dict.alive ? dc.codes : (uint32_t*)(tmp + index_at);
      if (packed_rows < (uint64_t)n) {
        w.begin(d_index);
        w.write(index.data(), (size_t)n * 4);
      }
      w.join();
      uint32_t* d_flag = (uint32_t*)(tmp + packed_at);
      for (size_t k = 0; k < packed_batches.size(); ++k) {
        const PackedBatch& pb = packed_batches[k];
        const uint8_t* d_runs = tmp + runs_at[k];
        launch_rle_expand(ctx, (const PackedRun*)d_runs, (uint32_t)pb.pk->n_runs, d_runs + pad(pb.pk->n_runs * sizeof(PackedRun)), pb.rows,
                          pb.base, pb.dict_len, d_index + pb.row0, d_flag, ctx->copy_stream);
      }
      DictExpand q;
      q.index = d_index;
      q.rows = (uint64_t)n;
      q.dict_off = (const uint32_t*)(tmp + off_at);
      q.dict_bytes = tmp + dbytes_at;
      q.remap = (const uint32_t*)(tmp + remap_at);
      q.codes = dict.alive ? dc.codes : nullptr;
      q.offsets = (int32_t*)dc.values;
      q.bytes = dc.bytes;
      q.block_sums = (uint32_t*)(tmp + sums_at);
      if (packed_batches.empty()) {
        launch_dict_expand(ctx, q, ctx->copy_stream);
      } else {
        // the bytes' total is only known on the device: it sizes the column's buffer (the scan
        // kernel leaves it, and the flag, in page-locked memory)
        volatile uint32_t* back = (volatile uint32_t*)ctx->load_landing;
        q.flag_in = d_flag;
        q.report = (uint32_t*)ctx->load_landing;
        launch_dict_lens(ctx, q, ctx->copy_stream);
        PRISM_HIP(hipStreamSynchronize(ctx->copy_stream));
        if (back[0]) fail(PRISM_GPU_EINVAL, "dictionary index out of range in " + f.name);
        total = back[1];
        dc.bytes = (uint8_t*)dev_alloc(t, (size_t)total);
        dc.n_bytes = total;
        q.bytes = dc.bytes;
        launch_dict_fill(ctx, q, ctx->copy_stream);
      }
      launch_col_stats(ctx, dict.alive ? (const void*)dc.codes : dc.values, dc.validity, 4, true, (uint64_t)n, &ch.d_stats[ci], ctx->copy_stream);
      continue;
    }
    if (!codes_only) {
      dc.values = dev_alloc(t, offs.size() * 4);
      w.begin(dc.values);
      w.write(offs.data(), offs.size() * 4);
      dc.bytes = (uint8_t*)dev_alloc(t, (size_t)total);
      dc.n_bytes = total;
      w.begin(dc.bytes);
      for (const Piece& pc : pieces) w.write(pc.p, pc.n);
    }
    bool coded = false;
    if (dev_lookup) {
      // the dictionary as the kernel wants it: offsets, bytes, and an index with at most half its slots taken
      const size_t n_values = dict.values.size();
      std::vector<uint32_t> d_off(n_values + 1, 0);
      for (size_t v = 0; v < n_values; ++v) d_off[v + 1] = d_off[v] + (uint32_t)dict.values[v].size();
      std::vector<uint8_t> d_bytes(d_off.back());
      for (size_t v = 0; v < n_values; ++v) std::memcpy(d_bytes.data() + d_off[v], dict.values[v].data(), dict.values[v].size());
      const uint32_t n_slots = (uint32_t)pow2_ceil(std::max<size_t>(64, 2 * n_values));
      std::vector<uint32_t> slots(n_slots, 0);
      for (size_t v = 0; v < n_values; ++v) {
        uint32_t e = (uint32_t)dict_lookup_hash_host((const uint8_t*)dict.values[v].data(), (uint32_t)dict.values[v].size()) & (n_slots - 1);
        while (slots[e]) e = (e + 1) & (n_slots - 1);
        slots[e] = (uint32_t)v + 1;
      }
      auto pad = [](size_t v) { return (v + 15) & ~(size_t)15; };
      const size_t off_at = 0, slots_at = pad(d_off.size() * 4), bytes_at = slots_at + pad((size_t)n_slots * 4), need = bytes_at + pad(d_bytes.size()) + 16;
      uint8_t* tmp = (uint8_t*)load_scratch(ctx, need);
      w.begin(tmp + off_at);
      w.write(d_off.data(), d_off.size() * 4);
      w.begin(tmp + slots_at);
      w.write(slots.data(), (size_t)n_slots * 4);
      if (!d_bytes.empty()) {
        w.begin(tmp + bytes_at);
        w.write(d_bytes.data(), d_bytes.size());
      }
      w.join();
      dc.codes = (uint32_t*)dev_alloc(t, (size_t)n * 4);
      volatile uint32_t* missed = (volatile uint32_t*)ctx->load_landing;
      missed[0] = 0;
      DictLookup q;
      q.offsets = (const int32_t*)dc.values;
      q.bytes = dc.bytes;
      q.validity = dc.validity;
      q.rows = (uint64_t)n;
      q.dict_off = (const uint32_t*)(tmp + off_at);
      q.dict_bytes = tmp + bytes_at;
      q.slots = (const uint32_t*)(tmp + slots_at);
      q.mask = n_slots - 1;
      q.codes = dc.codes;
      q.report = (uint32_t*)ctx->load_landing;
      launch_dict_lookup(ctx, q, ctx->copy_stream);
      PRISM_HIP(hipStreamSynchronize(ctx->copy_stream));
      coded = missed[0] == 0;
      if (coded) dict.lookup_misses = 0;
      if (!coded) {
        // a string the dictionary lacks: the chunk is coded row by row after all (which adds it)
        // (one miss may be the chunk that completes the dictionary; misses in a row are a column that keeps growing)
        dict.lookup_pause = dict.lookup_misses == 0 ? 0 : std::min(8, 1 << dict.lookup_misses);
        dict.lookup_misses = std::min(dict.lookup_misses + 1, 8);
        codes.resize((size_t)n);
        int64_t r0 = 0;
        for (size_t b = 0; b < n_batches && dict.values.size() <= kMaxDictValues; ++b) {
          const ArrowArray* a = &views[b][ci];
          const int64_t nb = a->length;
          if (nb == 0) continue;
          const uint8_t* vbits = (a->null_count != 0 && a->n_buffers > 0) ? (const uint8_t*)a->buffers[0] : nullptr;
          const int32_t* so = (const int32_t*)a->buffers[1] + a->offset;
          const uint8_t* sb = (const uint8_t*)a->buffers[2];
          for (int64_t i = 0; i < nb; ++i) {
            const bool valid = !vbits || bit_at(vbits, a->offset + i);
            codes[(size_t)(r0 + i)] = valid ? dict.id_of((const char*)sb + so[i], (size_t)(so[i + 1] - so[i])) : 0;
            if (dict.values.size() > kMaxDictValues) break;
          }
          r0 += nb;
        }
        retire_full_dictionary();
        if (!dict.alive) {
          dev_free(t, dc.codes);
          dc.codes = nullptr;
        }
      }
    }
    if (dict.alive && coded) {
      launch_col_stats(ctx, dc.codes, dc.validity, 4, true, (uint64_t)n, &ch.d_stats[ci], ctx->copy_stream);
    } else if (dict.alive) {
      if (!dc.codes) dc.codes = (uint32_t*)dev_alloc(t, (size_t)n * 4);
      w.begin(dc.codes);
      w.write(codes.data(), (size_t)n * 4);
      w.join();
      launch_col_stats(ctx, dc.codes, dc.validity, 4, true, (uint64_t)n, &ch.d_stats[ci], ctx->copy_stream);
    } else {
      // no codes: the offsets stand in as values so that the NULL count is still recorded
      w.join();
      launch_col_stats(ctx, dc.values, dc.validity, 4, true, (uint64_t)n, &ch.d_stats[ci], ctx->copy_stream);
    }
  }
  w.flush();
  // The bytes are in pinned memory or on their way: the batches can go.  The copies are
  // awaited where the chunk's statistics are read (table_sync_stats), before any scan.
  // (Bytes that travel straight from the batches' own buffers must have left them first.)
  if (w.direct) {
    PRISM_HIP(hipEventCreateWithFlags(&direct_done.ev, hipEventDisableTiming));
    PRISM_HIP(hipEventRecord(direct_done.ev, ctx->direct_stream));
  }
  {
    std::lock_guard<std::mutex> gs(ctx->mu);
    ctx->stats.staging_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->stats.bytes_staged += w.staged;
  }
  guard.armed = false;
  t->chunks.push_back(std::move(ch));
  direct_done.hand_over = pending;
  stage_lock.unlock();
  g.unlock();
  // (direct_done waits here, on the way out)
}

void table_append(prism_gpu_table* t, uint32_t partition_id, ArrowArray* batch, bool take, hipEvent_t* pending) {
  ArrowArray* one[1] = {batch};
  constexpr int64_t kMaxChunkRows = 1ll << 28;
  if (!batch || batch->length <= kMaxChunkRows || batch->offset < 0) {
    table_append_batches(t, partition_id, one, 1, take, pending);
    return;
  }
  if (pending) *pending = nullptr;  // (several chunks: each waits for its own copies)
  // A batch of more rows than a chunk holds (kernels address a chunk's column with 32-bit byte
  // offsets) becomes several chunks of the partition: row ranges of the same buffers, borrowed.
  struct Releaser {
    ArrowArray* b;
    bool take;
    ~Releaser() { if (take && b->release) b->release(b); }
  } releaser{batch, take};
  for (int64_t first = 0; first < batch->length; first += kMaxChunkRows) {
    ArrowArray slice = *batch;
    slice.offset = batch->offset + first;
    slice.length = std::min<int64_t>(kMaxChunkRows, batch->length - first);
    slice.release = nullptr;
    ArrowArray* part[1] = {&slice};
    table_append_batches(t, partition_id, part, 1, false);
  }
}

void table_synth(prism_gpu_table* t, uint32_t partition_id, uint64_t seed, uint64_t first_row, uint64_t n_rows,
                 uint64_t total_rows, const prism_gpu_synth_col* cols, uint32_t n_cols) {
  prism_gpu_ctx* ctx = t->ctx;
  bind(ctx);
  if (n_rows == 0) return;
  if (n_cols != t->fields.size()) fail(PRISM_GPU_EINVAL, "one generator per table column is required");
  if (total_rows == 0 || first_row + n_rows > total_rows) fail(PRISM_GPU_EINVAL, "row range exceeds total_rows");
  std::lock_guard<std::mutex> g(t->mu);
  Chunk ch;
  ch.partition_id = partition_id;
  ch.rows = n_rows;
  ch.cols.resize(t->fields.size());
  struct ChunkGuard {
    prism_gpu_table* t;
    Chunk* ch;
    bool armed = true;
    ~ChunkGuard() { if (armed) free_chunk_memory(t, *ch); }
  } guard{t, &ch};
  init_stats(t, ch, t->fields.size());
  PRISM_HIP(hipStreamSynchronize(ctx->copy_stream));
  std::vector<void*> cdfs;
  for (uint32_t i = 0; i < n_cols; ++i) {
    const prism_gpu_synth_col& sc = cols[i];
    if (sc.column < 0 || (size_t)sc.column >= t->fields.size()) fail(PRISM_GPU_EINVAL, "generator column out of range");
    const FieldType& f = t->fields[sc.column];
    DeviceColumn& dc = ch.cols[sc.column];
    if (dc.values || dc.codes) fail(PRISM_GPU_EINVAL, "two generators for one column");
    if (sc.kind == PRISM_SYNTH_URL) {
      // raw strings: lengths (hence offsets) on the host, bytes on the device
      if (!f.is_string) fail(PRISM_GPU_EINVAL, "the URL generator needs a string column");
      if (!sc.cdf || sc.cdf_len == 0) fail(PRISM_GPU_EINVAL, "the URL generator needs a table of lengths");
      ColumnDict& dict = t->dicts[sc.column];
      if (!dict.values.empty()) fail(PRISM_GPU_EINVAL, "the column already has a dictionary");
      dict.alive = false;  // high cardinality by construction: raw representation only
      const uint64_t key = prism_synth_column_key(seed, (uint32_t)sc.column);
      std::vector<int32_t> offs((size_t)n_rows + 1);
      uint64_t pos = 0;
      offs[0] = 0;
      for (uint64_t r = 0; r < n_rows; ++r) {
        pos += prism_synth_url_length(sc.cdf, sc.cdf_len, key, first_row + r);
        if (pos > 0x7fffffffull) fail(PRISM_GPU_EUNSUPPORTED, "string data of one partition exceeds 2 GiB; use more partitions");
        offs[(size_t)r + 1] = (int32_t)pos;
      }
      dc.values = dev_alloc(t, offs.size() * 4);
      stage_copy(ctx, dc.values, offs.data(), offs.size() * 4);
      PRISM_HIP(hipStreamSynchronize(ctx->copy_stream));
      dc.bytes = (uint8_t*)dev_alloc(t, pos);
      dc.n_bytes = pos;
      launch_synth_url(ctx, (const int32_t*)dc.values, dc.bytes, n_rows, first_row, sc.p[0], sc.p[1], key);
      launch_col_stats(ctx, dc.values, nullptr, 4, true, n_rows, &ch.d_stats[sc.column], ctx->stream);
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      continue;
    }
    SynthColDev d{};
    d.width = f.is_string ? 4 : f.width;
    d.kind = sc.kind;
    for (int k = 0; k < 4; ++k) d.p[k] = sc.p[k];
    d.cdf = nullptr;
    d.cdf_len = sc.cdf_len;
    if (sc.kind == PRISM_SYNTH_TS_LINEAR) {
      unsigned __int128 prod = (unsigned __int128)total_rows * (unsigned __int128)(uint64_t)sc.p[1];
      if (sc.p[1] < 0 || (prod >> 64) != 0) fail(PRISM_GPU_EINVAL, "TS_LINEAR needs total_rows * span < 2^64");
    }
    if (sc.kind == PRISM_SYNTH_CATEGORICAL) {
      if (!sc.cdf || sc.cdf_len == 0) fail(PRISM_GPU_EINVAL, "CATEGORICAL generator needs a cdf table");
      void* dcdf = dev_alloc(t, sc.cdf_len * 4);
      cdfs.push_back(dcdf);
      PRISM_HIP(hipMemcpy(dcdf, sc.cdf, sc.cdf_len * 4, hipMemcpyHostToDevice));
      d.cdf = (const uint32_t*)dcdf;
    }
    if (sc.kind == PRISM_SYNTH_UNIFORM && sc.p[1] <= 0) fail(PRISM_GPU_EINVAL, "UNIFORM generator needs range > 0");
    if (sc.kind == PRISM_SYNTH_LOGUNIFORM_ID && (sc.p[0] < 1 || sc.p[0] > 62))
      fail(PRISM_GPU_EINVAL, "LOGUNIFORM_ID needs 1 <= B <= 62");
    d.column_key = prism_synth_column_key(seed, (uint32_t)sc.column);
    void* dst = dev_alloc(t, (size_t)n_rows * d.width);
    d.dst = dst;
    if (f.is_string) dc.codes = (uint32_t*)dst; else dc.values = dst;
    launch_synth(ctx, d, first_row, n_rows, total_rows);
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    launch_col_stats(ctx, dst, nullptr, d.width, f.is_string || f.is_unsigned, n_rows, &ch.d_stats[sc.column], ctx->stream);
  }
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  for (void* p : cdfs) dev_free(t, p);
  for (size_t ci = 0; ci < t->fields.size(); ++ci)
    if (!ch.cols[ci].values && !ch.cols[ci].codes) fail(PRISM_GPU_EINVAL, "column " + t->fields[ci].name + " has no generator");
  guard.armed = false;
  t->chunks.push_back(std::move(ch));
}

void table_drop_partition(prism_gpu_table* t, uint32_t partition_id) {
  std::lock_guard<std::mutex> g(t->mu);
  for (size_t i = 0; i < t->chunks.size();) {
    if (t->chunks[i].partition_id != partition_id) { ++i; continue; }
    free_chunk_memory(t, t->chunks[i]);
    t->chunks.erase(t->chunks.begin() + (ptrdiff_t)i);
  }
}

void table_device_dictionary(prism_gpu_table* t, int ci) {
  prism_gpu_ctx* ctx = t->ctx;
  bind(ctx);
  std::lock_guard<std::mutex> g(t->mu);
  ColumnDict& dict = t->dicts[(size_t)ci];
  if (dict.alive || !t->fields[(size_t)ci].is_string) return;
  bool complete = dict.device_built;
  for (const Chunk& ch : t->chunks) complete = complete && (ch.rows == 0 || ch.cols[(size_t)ci].codes != nullptr);
  if (complete) return;
  HostTimer timer(ctx, "host:device_dictionary");
  // the chunks' copies run on the staging streams; the kernels below on the compute stream
  PRISM_HIP(sync_copy_streams(ctx));
  std::vector<StrSeg> segs;
  std::vector<Chunk*> seg_chunks;
  uint64_t n_rows = 0, col_bytes = 0;
  for (Chunk& ch : t->chunks) {
    DeviceColumn& dc = ch.cols[(size_t)ci];
    dev_free(t, dc.codes);
    dc.codes = nullptr;
    if (ch.rows == 0) continue;
    if (!dc.values) fail(PRISM_GPU_EINTERNAL, "string column " + t->fields[(size_t)ci].name + " has neither a dictionary nor its raw form");
    dc.codes = (uint32_t*)dev_alloc(t, (size_t)ch.rows * 4);
    segs.push_back({(const int32_t*)dc.values, dc.bytes, dc.validity, dc.codes, n_rows, ch.rows});
    seg_chunks.push_back(&ch);
    n_rows += ch.rows;
    col_bytes += dc.n_bytes;
  }
  dict.values.clear();
  dict.front.clear();
  dict.device_built = false;
  if (n_rows >= 0xfffffff0ull) fail(PRISM_GPU_EUNSUPPORTED, "GROUP BY on string column " + t->fields[(size_t)ci].name + ": 2^32 rows or more");
  if (col_bytes >= 0xfffffff0ull) fail(PRISM_GPU_EUNSUPPORTED, "GROUP BY on string column " + t->fields[(size_t)ci].name + ": 4 GiB of string data or more");
  if (segs.empty()) { dict.device_built = true; return; }
  const uint64_t capacity = pow2_ceil(std::max<uint64_t>(1024, 2 * n_rows));
  const size_t count_blocks = str_dict_blocks(capacity) + 1;
  Scratch d_segs(ctx, segs.size() * sizeof(StrSeg)), hashes(ctx, n_rows * 8), keys(ctx, capacity * 8), rep(ctx, capacity * 4),
      code_of_entry(ctx, capacity * 4), rep_by_code(ctx, n_rows * 4), sums(ctx, count_blocks * 4), small(ctx, 256);
  PRISM_HIP(hipMemcpyAsync(d_segs.p, segs.data(), segs.size() * sizeof(StrSeg), hipMemcpyHostToDevice, ctx->stream));
  uint32_t* const d_flag = small.as<uint32_t>();
  uint32_t* const d_n_values = d_flag + 1;
  uint32_t h_small[2] = {1, 0};
  for (uint64_t attempt = 0; attempt < 4 && h_small[0] != 0; ++attempt) {
    // (a new seed only after two different strings met in one hash: with 63 bits, practically never)
    PRISM_HIP(hipMemsetAsync(keys.p, 0xff, capacity * 8, ctx->stream));
    PRISM_HIP(hipMemsetAsync(rep.p, 0xff, capacity * 4, ctx->stream));
    PRISM_HIP(hipMemsetAsync(small.p, 0, 8, ctx->stream));
    launch_str_hash(ctx, d_segs.as<StrSeg>(), (uint32_t)segs.size(), n_rows, 0x5bd1e9955bd1e995ull + attempt * 0x9E3779B97F4A7C15ull, hashes.as<uint64_t>());
    launch_str_insert(ctx, hashes.as<uint64_t>(), n_rows, keys.as<uint64_t>(), rep.as<uint32_t>(), capacity);
    launch_str_number(ctx, keys.as<uint64_t>(), capacity, rep.as<uint32_t>(), sums.as<uint32_t>(), code_of_entry.as<uint32_t>(),
                      rep_by_code.as<uint32_t>(), d_n_values);
    launch_str_codes(ctx, d_segs.as<StrSeg>(), (uint32_t)segs.size(), n_rows, hashes.as<uint64_t>(), keys.as<uint64_t>(), rep.as<uint32_t>(),
                     code_of_entry.as<uint32_t>(), capacity, d_flag);
    PRISM_HIP(hipMemcpyAsync(h_small, small.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
  }
  if (h_small[0] != 0) fail(PRISM_GPU_EINTERNAL, "different strings of column " + t->fields[(size_t)ci].name + " kept meeting in one hash");
  const uint32_t n_values = h_small[1];
  // the distinct strings, in code order: offsets first (they tell the size), then the bytes
  const size_t value_blocks = str_dict_blocks(n_values) + 1;
  Scratch vsums(ctx, value_blocks * 4), value_off(ctx, ((size_t)n_values + 1) * 4);
  launch_str_values(ctx, d_segs.as<StrSeg>(), (uint32_t)segs.size(), rep_by_code.as<uint32_t>(), n_values, vsums.as<uint32_t>(),
                    value_off.as<uint32_t>(), nullptr, 0);
  std::vector<uint32_t> off((size_t)n_values + 1);
  PRISM_HIP(hipMemcpyAsync(off.data(), value_off.p, off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  const uint64_t total = off[n_values];
  std::vector<uint8_t> bytes((size_t)total);
  if (total) {
    Scratch value_bytes(ctx, (size_t)total);
    launch_str_values(ctx, d_segs.as<StrSeg>(), (uint32_t)segs.size(), rep_by_code.as<uint32_t>(), n_values, vsums.as<uint32_t>(),
                      value_off.as<uint32_t>(), value_bytes.as<uint8_t>(), total);
    PRISM_HIP(hipMemcpyAsync(bytes.data(), value_bytes.p, (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
  }
  dict.values.reserve(n_values);
  for (uint32_t v = 0; v < n_values; ++v) dict.values.emplace_back((const char*)bytes.data() + off[v], (size_t)(off[v + 1] - off[v]));
  // the column's statistics are those of its codes from now on (they size the key's bits)
  const ColStats identity{0x7fffffffffffffffll, -0x7fffffffffffffffll - 1, 0};
  for (Chunk* ch : seg_chunks) {
    PRISM_HIP(hipMemcpyAsync(&ch->d_stats[ci], &identity, sizeof(identity), hipMemcpyHostToDevice, ctx->stream));
    launch_col_stats(ctx, ch->cols[(size_t)ci].codes, ch->cols[(size_t)ci].validity, 4, true, ch->rows, &ch->d_stats[ci], ctx->stream);
    ch->stats_ready = false;
  }
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  collect_timings(ctx);
  dict.device_built = true;
}

void table_sync_stats(prism_gpu_table* t) {
  std::lock_guard<std::mutex> g(t->mu);
  bool waited = false;
  for (auto& ch : t->chunks) {
    if (ch.stats_ready) continue;
    // the chunk's copies and statistics kernels run on the staging streams
    if (!waited) { PRISM_HIP(sync_copy_streams(t->ctx)); waited = true; }
    PRISM_HIP(hipMemcpy(ch.stats.data(), ch.d_stats, ch.stats.size() * sizeof(ColStats), hipMemcpyDeviceToHost));
    ch.stats_ready = true;
  }
}

void table_dictionary_digest(prism_gpu_table* table, int column, uint64_t* digest, uint32_t* n_values) {
  std::lock_guard<std::mutex> g(table->mu);
  const ColumnDict& d = table->dicts[column];
  uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a
  auto mix = [&](uint8_t b) { h = (h ^ b) * 0x100000001b3ull; };
  for (const std::string& v : d.values) {
    for (unsigned char c : v) mix(c);
    mix(0xff);  // separator: no UTF-8 byte
  }
  if (!d.alive) h = ~h;
  *digest = h;
  *n_values = (uint32_t)d.values.size();
}

}  // namespace prism
