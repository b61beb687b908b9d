#include "arrow_io.hpp"

#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <type_traits>
#include <utility>

namespace prism {

FieldType field_from_format(const char* format) {
  FieldType t;
  std::string f(format ? format : "");
  t.format = f;
  if (f == "l") { t.logical = PRISM_COL_INT64; t.width = 8; }
  else if (f == "i") { t.logical = PRISM_COL_INT32; t.width = 4; }
  else if (f == "s") { t.logical = PRISM_COL_INT16; t.width = 2; }
  else if (f == "S") { t.logical = PRISM_COL_UINT16; t.width = 2; t.is_unsigned = true; }
  else if (f == "L") { t.logical = kTypeUInt64; t.width = 8; t.is_unsigned = true; }
  else if (f == "g") { t.logical = kTypeFloat64; t.width = 8; t.is_float = true; }
  else if (f == "u") { t.logical = PRISM_COL_UTF8; t.is_string = true; }
  else if (f == "z") { t.logical = PRISM_COL_BINARY; t.is_string = true; }
  else if (f.rfind("tsm:", 0) == 0) { t.logical = PRISM_COL_TIMESTAMP; t.width = 8; }
  else fail(PRISM_GPU_EUNSUPPORTED, "unsupported Arrow type '" + f + "' (supported: Int16/Int32/Int64/UInt16/Timestamp(ms)/Utf8/Binary)");
  return t;
}

FieldType list_field(const std::string& name, const std::string& item_format) {
  (void)field_from_format(item_format.c_str());  // (a supported value type)
  FieldType t;
  t.logical = kTypeList;
  t.format = "+l";
  t.name = name;
  t.nullable = true;
  t.list_item = item_format;
  return t;
}

std::vector<FieldType> parse_schema(const ArrowSchema* schema) {
  if (!schema || !schema->format || std::string(schema->format) != "+s")
    fail(PRISM_GPU_EINVAL, "schema must be a struct (record batch) schema");
  std::vector<FieldType> out;
  for (int64_t i = 0; i < schema->n_children; ++i) {
    const ArrowSchema* c = schema->children[i];
    if (c->format && std::string(c->format) == "+l") {
      if (c->n_children != 1 || !c->children[0] || c->children[0]->dictionary) fail(PRISM_GPU_EUNSUPPORTED, "a List column must have one plain value type");
      FieldType t = list_field(c->name ? c->name : "", c->children[0]->format ? c->children[0]->format : "");
      t.nullable = (c->flags & ARROW_FLAG_NULLABLE) != 0;
      out.push_back(t);
      continue;
    }
    FieldType t = field_from_format(c->dictionary ? c->dictionary->format : c->format);
    if (c->dictionary) {
      if (!t.is_string) fail(PRISM_GPU_EUNSUPPORTED, "dictionary-encoded columns must have string values");
      if (std::string(c->format) != "i") fail(PRISM_GPU_EUNSUPPORTED, "dictionary indices must be Int32");
    }
    t.name = c->name ? c->name : "";
    t.nullable = (c->flags & ARROW_FLAG_NULLABLE) != 0;
    out.push_back(t);
  }
  return out;
}

void OutColumn::borrow(std::shared_ptr<void> owner, const int64_t* vals, const uint8_t* valid_bytes, size_t n) {
  if (rows() != 0) fail(PRISM_GPU_EINTERNAL, "only an empty output column can borrow rows");
  ext_owner = std::move(owner);
  ext_vals = vals;
  ext_valid = valid_bytes;
  ext_rows = n;
}

void OutColumn::materialise() {
  if (!borrowed()) return;
  const size_t n = ext_rows;
  valid.assign(ext_valid, ext_valid + n);
  if (type.is_string) {
    codes.resize(n);
    for (size_t i = 0; i < n; ++i) codes[i] = (uint32_t)ext_vals[i];
  } else if (type.is_float) {
    const double* f = (const double*)ext_vals;
    f64.assign(f, f + n);
  } else {
    i64.assign(ext_vals, ext_vals + n);
  }
  ext_vals = nullptr;
  ext_valid = nullptr;
  ext_rows = 0;
  ext_owner.reset();
}

void OutColumn::gather(const std::vector<size_t>& idx) {
  materialise();
  auto take = [&](auto& v) {
    if (v.empty()) return;
    typename std::remove_reference<decltype(v)>::type out;
    out.reserve(idx.size());
    for (size_t i : idx) out.push_back(std::move(v[i]));
    v.swap(out);
  };
  take(i64);
  take(f64);
  take(str);
  take(codes);
  take(valid);
}

void OutColumn::push_null() {
  materialise();
  valid.push_back(0);
  if (type.is_string && dict) codes.push_back(0);
  else if (type.is_string) str.emplace_back();
  else if (type.is_float) f64.push_back(0.0);
  else i64.push_back(0);
}
void OutColumn::push_int(int64_t v) { materialise(); valid.push_back(1); i64.push_back(v); }
void OutColumn::push_float(double v) { materialise(); valid.push_back(1); f64.push_back(v); }
void OutColumn::push_str(const std::string& s) { materialise(); valid.push_back(1); str.push_back(s); }

OutColumn make_out_column(const std::string& name, const std::string& format, bool nullable) {
  OutColumn c;
  if (format.rfind("+l:", 0) == 0) {  // (a List of the format behind the colon: see make_columns' callers)
    c.type = list_field(name, format.substr(3));
    c.items = std::make_shared<OutColumn>(make_out_column("item", format.substr(3), true));
    c.list_off.assign(1, 0);
    return c;
  }
  c.type = field_from_format(format.c_str());
  c.type.name = name;
  c.type.nullable = nullable;
  return c;
}

namespace {

// Host blocks of exported batches are recycled: a fresh allocation of this size comes
// from mmap and costs a page fault per 4 KiB when it is first written, which is most of
// the time an export of ~10^5 rows takes.  Bounded: at most kKeepPerClass blocks per
// power-of-two size class, none above kMaxPooledBytes.
class HostBlockPool {
 public:
  static constexpr size_t kMinBytes = 4096, kMaxPooledBytes = 16ull << 20, kKeepPerClass = 8;
  static HostBlockPool& get() {
    static HostBlockPool* pool = new HostBlockPool();  // never destroyed: batches may outlive static destruction
    return *pool;
  }
  uint8_t* take(size_t bytes, size_t* cls_out) {
    size_t cls = kMinBytes;
    while (cls < bytes) cls <<= 1;
    *cls_out = cls;
    if (cls <= kMaxPooledBytes) {
      std::lock_guard<std::mutex> g(mu_);
      auto& v = free_[cls];
      if (!v.empty()) {
        uint8_t* p = v.back();
        v.pop_back();
        return p;
      }
    }
    return static_cast<uint8_t*>(std::malloc(cls));
  }
  void give(uint8_t* p, size_t cls) {
    if (cls <= kMaxPooledBytes) {
      std::lock_guard<std::mutex> g(mu_);
      auto& v = free_[cls];
      if (v.size() < kKeepPerClass) {
        v.push_back(p);
        return;
      }
    }
    std::free(p);
  }

 private:
  std::mutex mu_;
  std::map<size_t, std::vector<uint8_t*>> free_;
};

struct BatchOwner {
  std::vector<std::pair<uint8_t*, size_t>> blocks;
  // 8-byte columns are not copied: their vectors move in here and the Arrow buffers
  // point at them
  std::vector<std::vector<int64_t>> kept_i64;
  std::vector<std::vector<double>> kept_f64;
  // blocks that borrowed columns point into
  std::vector<std::shared_ptr<void>> leases;
  std::deque<std::vector<const void*>> nested_buffer_lists;  // (deques: the arrays point at their elements)
  std::deque<ArrowArray> nested;                            // the item arrays of list columns
  std::deque<std::vector<ArrowArray*>> nested_ptrs;
  std::vector<ArrowArray> children;
  std::vector<ArrowArray*> child_ptrs;
  const void* struct_buffers[1] = {nullptr};

  BatchOwner() = default;
  BatchOwner(const BatchOwner&) = delete;
  BatchOwner& operator=(const BatchOwner&) = delete;
  ~BatchOwner() {
    for (auto& b : blocks) HostBlockPool::get().give(b.first, b.second);
  }

  // `bytes` + 64 spare bytes
  uint8_t* block(size_t bytes, bool zero = true) {
    size_t cls = 0;
    uint8_t* p = HostBlockPool::get().take(bytes + 64, &cls);
    if (!p) fail(PRISM_GPU_EOOM, "out of host memory while exporting a result batch");
    blocks.emplace_back(p, cls);
    if (zero) std::memset(p, 0, bytes + 64);
    return p;
  }
};

void release_batch(ArrowArray* a) {
  delete static_cast<BatchOwner*>(a->private_data);
  a->release = nullptr;
}
void release_noop_array(ArrowArray* a) { a->release = nullptr; }

struct SchemaOwner {
  std::vector<std::string> names, formats;
  std::vector<ArrowSchema> children;
  std::vector<ArrowSchema*> child_ptrs;
  std::deque<std::string> item_formats;  // the value types of List fields
  std::deque<ArrowSchema> items;
  std::deque<std::vector<ArrowSchema*>> item_ptrs;
};
void release_schema_owner(ArrowSchema* s) {
  delete static_cast<SchemaOwner*>(s->private_data);
  s->release = nullptr;
}
void release_noop_schema(ArrowSchema* s) { s->release = nullptr; }

}  // namespace

namespace {

// One column -> one ArrowArray whose buffers the owner keeps.
void fill_array(BatchOwner* owner, OutColumn& col, ArrowArray& a) {
    const size_t n = col.rows();
    std::memset(&a, 0, sizeof(a));
    int64_t nulls = 0;
    uint8_t* bits = nullptr;
    const uint8_t* valid = col.valid_data();
    if (n && std::memchr(valid, 0, n) != nullptr) {  // bitmap only when there is a NULL
      bits = owner->block((n + 7) / 8);
      for (size_t i = 0; i < n; ++i) {
        if (valid[i]) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
        else ++nulls;
      }
    }
    owner->nested_buffer_lists.emplace_back();
    auto& bl = owner->nested_buffer_lists.back();
    bl.push_back(nulls ? bits : nullptr);
    if (col.type.logical == kTypeList) {
      if (!col.items || col.list_off.size() != n + 1) fail(PRISM_GPU_EINTERNAL, "list column without its items");
      int32_t* off = (int32_t*)owner->block((n + 1) * 4, false);
      std::memcpy(off, col.list_off.data(), (n + 1) * 4);
      bl.push_back(off);
      owner->nested.emplace_back();
      ArrowArray& child = owner->nested.back();
      fill_array(owner, *col.items, child);
      owner->nested_ptrs.emplace_back(1, &child);
      a.n_children = 1;
      a.children = owner->nested_ptrs.back().data();
    } else if (col.type.is_string) {
      int32_t* off = (int32_t*)owner->block((n + 1) * 4, false);
      off[0] = 0;
      uint8_t* data = nullptr;
      if (col.dict) {
        // dictionary-coded: the lengths come from a flat table, not one std::string per row
        const std::vector<std::string>& d = *col.dict;
        std::vector<uint32_t> len(d.size());
        for (size_t v = 0; v < d.size(); ++v) len[v] = (uint32_t)d[v].size();
        uint32_t max_len = 0;
        for (uint32_t l : len) max_len = l > max_len ? l : max_len;
        if (max_len <= 32 && !d.empty() && d.size() * 8 <= n && (uint64_t)n * max_len <= 0x7fffffffull) {
          // Few short values, many rows: one pass that copies a fixed 16 or 32 bytes per row
          // from a flat copy of the dictionary (two or four moves instead of a call to
          // memcpy with a run-time length) into a block sized for the longest value per row.
          const size_t w = max_len <= 16 ? 16 : 32;
          std::vector<uint8_t> flat(d.size() * w, 0);
          for (size_t v = 0; v < d.size(); ++v) std::memcpy(&flat[v * w], d[v].data(), len[v]);
          data = owner->block((size_t)n * max_len + w, false);
          uint32_t pos = 0;
          bool uniform = nulls == 0;
          for (uint32_t l : len) uniform = uniform && l == max_len;
          if (uniform) {
            // Values of one length and no NULL: a row's place is its number times the length, so the
            // copies do not wait for one another (with running positions every row's store address
            // hangs on the row before: code -> length -> sum, some five cycles a row).
            if (w == 16) {
              for (size_t i = 0; i < n; ++i) std::memcpy(data + i * max_len, &flat[(size_t)col.code_at(i) * 16], 16);
            } else {
              for (size_t i = 0; i < n; ++i) std::memcpy(data + i * max_len, &flat[(size_t)col.code_at(i) * 32], 32);
            }
            for (size_t i = 0; i < n; ++i) off[i + 1] = (int32_t)((i + 1) * max_len);
          } else if (w == 16) {
            for (size_t i = 0; i < n; ++i) {
              const uint32_t code = valid[i] ? col.code_at(i) : 0u;
              std::memcpy(data + pos, &flat[(size_t)code * 16], 16);
              pos += valid[i] ? len[code] : 0u;
              off[i + 1] = (int32_t)pos;
            }
          } else {
            for (size_t i = 0; i < n; ++i) {
              const uint32_t code = valid[i] ? col.code_at(i) : 0u;
              std::memcpy(data + pos, &flat[(size_t)code * 32], 32);
              pos += valid[i] ? len[code] : 0u;
              off[i + 1] = (int32_t)pos;
            }
          }
        } else {
          uint64_t pos = 0;
          for (size_t i = 0; i < n; ++i) {
            pos += valid[i] ? len[col.code_at(i)] : 0u;
            off[i + 1] = (int32_t)pos;
            if (pos > 0x7fffffffull) fail(PRISM_GPU_EUNSUPPORTED, "string output exceeds 2 GiB");
          }
          data = owner->block((size_t)pos, false);
          for (size_t i = 0; i < n; ++i)
            if (valid[i]) std::memcpy(data + off[i], d[col.code_at(i)].data(), (size_t)(off[i + 1] - off[i]));
        }
      } else {
        size_t total = 0;
        for (size_t i = 0; i < n; ++i) total += valid[i] ? col.str[i].size() : 0;
        if (total > 0x7fffffffull) fail(PRISM_GPU_EUNSUPPORTED, "string output exceeds 2 GiB");
        data = owner->block(total, false);
        size_t pos = 0;
        for (size_t i = 0; i < n; ++i) {
          if (valid[i]) {
            const std::string& v = col.str[i];
            std::memcpy(data + pos, v.data(), v.size());
            pos += v.size();
          }
          off[i + 1] = (int32_t)pos;
        }
      }
      bl.push_back(off);
      bl.push_back(data);
    } else if (col.type.width == 8 && col.borrowed()) {
      owner->leases.push_back(col.ext_owner);
      bl.push_back(col.ext_vals);
    } else if (col.type.is_float) {
      owner->kept_f64.push_back(std::move(col.f64));
      bl.push_back(owner->kept_f64.back().data());
    } else if (col.type.width == 8) {
      owner->kept_i64.push_back(std::move(col.i64));
      bl.push_back(owner->kept_i64.back().data());
    } else if (col.type.width == 4) {
      const int64_t* src = col.i64_data();
      int32_t* v = (int32_t*)owner->block(n * 4);
      for (size_t i = 0; i < n; ++i) v[i] = (int32_t)src[i];
      bl.push_back(v);
    } else {
      const int64_t* src = col.i64_data();
      int16_t* v = (int16_t*)owner->block(n * 2);
      for (size_t i = 0; i < n; ++i) v[i] = (int16_t)src[i];
      bl.push_back(v);
    }
    a.length = (int64_t)n;
    a.null_count = nulls;
    a.n_buffers = (int64_t)bl.size();
    a.buffers = bl.data();
    a.release = release_noop_array;
  }

}  // namespace

void export_record_batch(std::vector<OutColumn>& cols, ArrowArray* out) {
  auto owner = std::make_unique<BatchOwner>();
  const size_t n = cols.empty() ? 0 : cols[0].rows();
  owner->children.resize(cols.size());
  for (size_t c = 0; c < cols.size(); ++c) {
    if (cols[c].rows() != n) fail(PRISM_GPU_EINTERNAL, "output columns have different lengths");
    fill_array(owner.get(), cols[c], owner->children[c]);
  }
  for (auto& a : owner->children) owner->child_ptrs.push_back(&a);
  std::memset(out, 0, sizeof(*out));
  out->length = (int64_t)n;
  out->n_buffers = 1;
  out->buffers = owner->struct_buffers;
  out->n_children = (int64_t)cols.size();
  out->children = owner->child_ptrs.data();
  out->release = release_batch;
  out->private_data = owner.release();
}

void export_schema(const std::vector<FieldType>& fields, ArrowSchema* out) {
  auto owner = std::make_unique<SchemaOwner>();
  for (auto& f : fields) { owner->names.push_back(f.name); owner->formats.push_back(f.format); }
  owner->children.resize(fields.size());
  for (size_t i = 0; i < fields.size(); ++i) {
    ArrowSchema& s = owner->children[i];
    std::memset(&s, 0, sizeof(s));
    s.format = owner->formats[i].c_str();
    s.name = owner->names[i].c_str();
    s.flags = fields[i].nullable ? ARROW_FLAG_NULLABLE : 0;
    s.release = release_noop_schema;
    if (fields[i].logical == kTypeList) {
      owner->item_formats.push_back(fields[i].list_item);
      owner->items.emplace_back();
      ArrowSchema& item = owner->items.back();
      std::memset(&item, 0, sizeof(item));
      item.format = owner->item_formats.back().c_str();
      item.name = "item";
      item.flags = ARROW_FLAG_NULLABLE;
      item.release = release_noop_schema;
      owner->item_ptrs.emplace_back(1, &item);
      s.n_children = 1;
      s.children = owner->item_ptrs.back().data();
    }
  }
  for (auto& s : owner->children) owner->child_ptrs.push_back(&s);
  std::memset(out, 0, sizeof(*out));
  out->format = "+s";
  out->name = "";
  out->n_children = (int64_t)fields.size();
  out->children = owner->child_ptrs.data();
  out->release = release_schema_owner;
  out->private_data = owner.release();
}

}  // namespace prism
