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
#include <chrono>
#include <cstring>

#include "../../include/prism_synth.h"
#include "engine.hpp"

namespace prism {

uint32_t ColumnDict::id_of(const char* s, size_t n) {
  std::string key(s, n);
  auto it = ids.find(key);
  if (it != ids.end()) return it->second;
  uint32_t id = (uint32_t)values.size();
  values.push_back(key);
  ids.emplace(std::move(key), id);
  return id;
}

namespace {

void* dev_alloc(size_t bytes) {
  void* p = nullptr;
  // slack: kernels read whole 16-byte vectors (plus one dword) at the tail
  PRISM_HIP(hipMalloc(&p, round_up(bytes, 256) + 256));
  return p;
}

// Host -> HBM through the pinned ring.  The source may be pageable memory owned by
// the caller (Arrow buffers): it is copied into a pinned half while the other half
// is still in flight on the copy stream.
void stage_copy(prism_gpu_ctx* ctx, void* dst, const void* src, size_t bytes) {
  std::lock_guard<std::mutex> g(ctx->mu);
  const size_t step = prism_gpu_ctx::kStageBytes;
  for (size_t off = 0; off < bytes; off += step) {
    size_t n = bytes - off < step ? bytes - off : step;
    int i = ctx->stage_next;
    ctx->stage_next ^= 1;
    if (ctx->stage_busy[i]) PRISM_HIP(hipEventSynchronize(ctx->stage_done[i]));
    std::memcpy(ctx->stage[i], (const uint8_t*)src + off, n);
    PRISM_HIP(hipMemcpyAsync((uint8_t*)dst + off, ctx->stage[i], n, hipMemcpyHostToDevice, ctx->copy_stream));
    PRISM_HIP(hipEventRecord(ctx->stage_done[i], ctx->copy_stream));
    ctx->stage_busy[i] = true;
  }
  ctx->stats.bytes_staged += bytes;
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

prism_gpu_table::~prism_gpu_table() {
  if (ctx) (void)hipSetDevice(ctx->device);
  for (auto& ch : chunks) {
    for (auto& c : ch.cols) {
      if (c.values) (void)hipFree(c.values);
      if (c.validity) (void)hipFree(c.validity);
      if (c.bytes) (void)hipFree(c.bytes);
      if (c.codes) (void)hipFree(c.codes);
    }
    if (ch.d_stats) (void)hipFree(ch.d_stats);
  }
}

namespace prism {

namespace {

void init_stats(prism_gpu_ctx* ctx, Chunk& ch, size_t n_cols) {
  ch.stats.assign(n_cols, ColStats{0x7fffffffffffffffll, -0x7fffffffffffffffll - 1, 0});
  ch.d_stats = (ColStats*)dev_alloc(n_cols * sizeof(ColStats));
  stage_copy(ctx, ch.d_stats, ch.stats.data(), n_cols * sizeof(ColStats));
}

}  // namespace

void table_append(prism_gpu_table* t, uint32_t partition_id, ArrowArray* batch, bool take) {
  prism_gpu_ctx* ctx = t->ctx;
  bind(ctx);
  struct Releaser {
    ArrowArray* a;
    bool take;
    ~Releaser() { if (take && a && a->release) a->release(a); }
  } releaser{batch, take};

  if (!batch) fail(PRISM_GPU_EINVAL, "batch is null");
  if (batch->n_children != (int64_t)t->fields.size())
    fail(PRISM_GPU_EINVAL, "batch has " + std::to_string(batch->n_children) + " columns, table schema has " +
                               std::to_string(t->fields.size()));
  if (batch->offset != 0) fail(PRISM_GPU_EUNSUPPORTED, "struct arrays with a non-zero offset are not supported");
  const int64_t n = batch->length;
  if (n == 0) return;
  if (n > 0x7fffffff) fail(PRISM_GPU_EUNSUPPORTED, "a batch may hold at most 2^31-1 rows");

  auto t0 = std::chrono::steady_clock::now();
  std::lock_guard<std::mutex> g(t->mu);
  Chunk ch;
  ch.partition_id = partition_id;
  ch.rows = (uint64_t)n;
  ch.cols.resize(t->fields.size());
  // the chunk owns its device memory from here on: on failure free it
  struct ChunkGuard {
    Chunk* ch;
    bool armed = true;
    ~ChunkGuard() {
      if (!armed) return;
      for (auto& c : ch->cols) {
        if (c.values) (void)hipFree(c.values);
        if (c.validity) (void)hipFree(c.validity);
        if (c.bytes) (void)hipFree(c.bytes);
        if (c.codes) (void)hipFree(c.codes);
      }
      if (ch->d_stats) (void)hipFree(ch->d_stats);
    }
  } guard{&ch};
  init_stats(ctx, ch, t->fields.size());

  for (size_t ci = 0; ci < t->fields.size(); ++ci) {
    const FieldType& f = t->fields[ci];
    const ArrowArray* a = batch->children[ci];
    DeviceColumn& dc = ch.cols[ci];
    if (a->length != n) fail(PRISM_GPU_EINVAL, "column " + f.name + " has a different length than the batch");
    const uint8_t* vbits = (a->null_count != 0 && a->n_buffers > 0) ? (const uint8_t*)a->buffers[0] : nullptr;
    if (vbits) {
      std::vector<uint64_t> words = realign_validity(vbits, a->offset, n);
      bool any_null = false;
      for (int64_t i = 0; i < n && !any_null; i += 64) {
        uint64_t want = n - i >= 64 ? ~0ull : ((1ull << (n - i)) - 1ull);
        any_null = (words[(size_t)(i >> 6)] & want) != want;
      }
      if (any_null) {
        dc.validity = (uint64_t*)dev_alloc(words.size() * 8);
        stage_copy(ctx, dc.validity, words.data(), words.size() * 8);
      }
    }
    if (!f.is_string) {
      if (a->dictionary) fail(PRISM_GPU_EUNSUPPORTED, "dictionary encoding is only supported for string columns");
      if (a->n_buffers < 2 || !a->buffers[1]) fail(PRISM_GPU_EINVAL, "column " + f.name + " has no value buffer");
      const uint8_t* src = (const uint8_t*)a->buffers[1] + (size_t)a->offset * f.width;
      dc.values = dev_alloc((size_t)n * f.width);
      stage_copy(ctx, dc.values, src, (size_t)n * f.width);
      launch_col_stats(ctx, dc.values, dc.validity, f.width, f.is_unsigned, (uint64_t)n, &ch.d_stats[ci],
                       ctx->copy_stream);
      continue;
    }
    // ---- string column: raw representation + table-wide dictionary codes ----
    ColumnDict& dict = t->dicts[ci];
    std::vector<uint32_t> codes;
    if (dict.alive) codes.resize((size_t)n);
    std::vector<int32_t> offs((size_t)n + 1);
    std::vector<uint8_t> expanded;  // raw bytes when the input is dictionary encoded
    const uint8_t* raw_src = nullptr;
    size_t raw_len = 0;
    if (a->dictionary) {
      const ArrowArray* d = a->dictionary;
      const int32_t* idx = (const int32_t*)a->buffers[1] + a->offset;
      const int32_t* doff = (const int32_t*)d->buffers[1] + d->offset;
      const char* dbytes = (const char*)d->buffers[2];
      std::vector<uint32_t> remap;
      if (dict.alive) {
        remap.resize((size_t)d->length);
        for (int64_t k = 0; k < d->length; ++k) remap[(size_t)k] = dict.id_of(dbytes + doff[k], (size_t)(doff[k + 1] - doff[k]));
      }
      size_t total = 0;
      for (int64_t i = 0; i < n; ++i) {
        bool valid = !vbits || bit_at(vbits, a->offset + i);
        int32_t k = valid ? idx[i] : 0;
        if (valid && (k < 0 || k >= d->length)) fail(PRISM_GPU_EINVAL, "dictionary index out of range in " + f.name);
        total += valid ? (size_t)(doff[k + 1] - doff[k]) : 0;
      }
      if (total > 0x7fffffffull) fail(PRISM_GPU_EUNSUPPORTED, "string data of one batch exceeds 2 GiB");
      expanded.resize(total);
      size_t pos = 0;
      offs[0] = 0;
      for (int64_t i = 0; i < n; ++i) {
        bool valid = !vbits || bit_at(vbits, a->offset + i);
        if (valid) {
          int32_t k = idx[i];
          size_t len = (size_t)(doff[k + 1] - doff[k]);
          std::memcpy(expanded.data() + pos, dbytes + doff[k], len);
          pos += len;
          if (dict.alive) codes[(size_t)i] = remap[(size_t)k];
        } else if (dict.alive) {
          codes[(size_t)i] = 0;
        }
        offs[(size_t)i + 1] = (int32_t)pos;
      }
      raw_src = expanded.data();
      raw_len = total;
    } else {
      if (a->n_buffers < 3 || !a->buffers[1]) fail(PRISM_GPU_EINVAL, "string column " + f.name + " has no offsets");
      const int32_t* so = (const int32_t*)a->buffers[1] + a->offset;
      const uint8_t* sb = (const uint8_t*)a->buffers[2];
      int32_t base = so[0];
      for (int64_t i = 0; i <= n; ++i) offs[(size_t)i] = so[i] - base;
      raw_src = sb ? sb + base : nullptr;
      raw_len = (size_t)(so[n] - base);
      if (dict.alive) {
        for (int64_t i = 0; i < n; ++i) {
          bool valid = !vbits || bit_at(vbits, a->offset + i);
          codes[(size_t)i] = valid ? dict.id_of((const char*)sb + so[i], (size_t)(so[i + 1] - so[i])) : 0;
          if (dict.values.size() > kMaxDictValues) break;
        }
      }
    }
    if (dict.alive && dict.values.size() > kMaxDictValues) {
      // too many distinct values: this column keeps only its raw form from now on
      dict.alive = false;
      dict.ids.clear();
      dict.values.clear();
      for (auto& old : t->chunks) {
        if (old.cols[ci].codes) { (void)hipFree(old.cols[ci].codes); old.cols[ci].codes = nullptr; }
      }
    }
    dc.values = dev_alloc(offs.size() * 4);
    stage_copy(ctx, dc.values, offs.data(), offs.size() * 4);
    dc.bytes = (uint8_t*)dev_alloc(raw_len);
    dc.n_bytes = raw_len;
    if (raw_len) stage_copy(ctx, dc.bytes, raw_src, raw_len);
    if (dict.alive) {
      dc.codes = (uint32_t*)dev_alloc((size_t)n * 4);
      stage_copy(ctx, dc.codes, codes.data(), (size_t)n * 4);
      launch_col_stats(ctx, dc.codes, dc.validity, 4, true, (uint64_t)n, &ch.d_stats[ci], ctx->copy_stream);
    } else {
      // no codes: the offsets stand in as values so that the NULL count is still recorded
      launch_col_stats(ctx, dc.values, dc.validity, 4, true, (uint64_t)n, &ch.d_stats[ci], ctx->copy_stream);
    }
  }
  PRISM_HIP(hipStreamSynchronize(ctx->copy_stream));
  {
    std::lock_guard<std::mutex> gs(ctx->mu);
    ctx->stage_busy[0] = ctx->stage_busy[1] = false;
    ctx->stats.staging_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  guard.armed = false;
  t->chunks.push_back(std::move(ch));
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
  init_stats(ctx, ch, t->fields.size());
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
      dc.values = dev_alloc(offs.size() * 4);
      stage_copy(ctx, dc.values, offs.data(), offs.size() * 4);
      PRISM_HIP(hipStreamSynchronize(ctx->copy_stream));
      {
        std::lock_guard<std::mutex> gs(ctx->mu);
        ctx->stage_busy[0] = ctx->stage_busy[1] = false;
      }
      dc.bytes = (uint8_t*)dev_alloc(pos);
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
      void* dcdf = dev_alloc(sc.cdf_len * 4);
      cdfs.push_back(dcdf);
      PRISM_HIP(hipMemcpy(dcdf, sc.cdf, sc.cdf_len * 4, hipMemcpyHostToDevice));
      d.cdf = (const uint32_t*)dcdf;
    }
    if (sc.kind == PRISM_SYNTH_UNIFORM && sc.p[1] <= 0) fail(PRISM_GPU_EINVAL, "UNIFORM generator needs range > 0");
    if (sc.kind == PRISM_SYNTH_LOGUNIFORM_ID && (sc.p[0] < 1 || sc.p[0] > 62))
      fail(PRISM_GPU_EINVAL, "LOGUNIFORM_ID needs 1 <= B <= 62");
    d.column_key = prism_synth_column_key(seed, (uint32_t)sc.column);
    void* dst = dev_alloc((size_t)n_rows * d.width);
    d.dst = dst;
    if (f.is_string) dc.codes = (uint32_t*)dst; else dc.values = dst;
    launch_synth(ctx, d, first_row, n_rows, total_rows);
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    launch_col_stats(ctx, dst, nullptr, d.width, f.is_string || f.is_unsigned, n_rows, &ch.d_stats[sc.column], ctx->stream);
  }
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  for (void* p : cdfs) (void)hipFree(p);
  for (size_t ci = 0; ci < t->fields.size(); ++ci)
    if (!ch.cols[ci].values && !ch.cols[ci].codes) fail(PRISM_GPU_EINVAL, "column " + t->fields[ci].name + " has no generator");
  t->chunks.push_back(std::move(ch));
}

void table_sync_stats(prism_gpu_table* t) {
  std::lock_guard<std::mutex> g(t->mu);
  for (auto& ch : t->chunks) {
    if (ch.stats_ready) continue;
    PRISM_HIP(hipMemcpy(ch.stats.data(), ch.d_stats, ch.stats.size() * sizeof(ColStats), hipMemcpyDeviceToHost));
    ch.stats_ready = true;
  }
}

}  // namespace prism
