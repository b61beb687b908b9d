#include "arrow_io.hpp"

#include <cstdlib>
#include <cstring>
#include <memory>
#include <type_traits>

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

std::vector<FieldType> parse_schema(const ArrowSchema* schema) {
  if (!schema || !schema->format || std::string(schema->format) != "+s")
    fail(PRISM_GPU_EINVAL, "schema must be a struct (record batch) schema");
  std::vector<FieldType> out;
  for (int64_t i = 0; i < schema->n_children; ++i) {
    const ArrowSchema* c = schema->children[i];
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

void OutColumn::gather(const std::vector<size_t>& idx) {
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
  valid.push_back(0);
  if (type.is_string && dict) codes.push_back(0);
  else if (type.is_string) str.emplace_back();
  else if (type.is_float) f64.push_back(0.0);
  else i64.push_back(0);
}
void OutColumn::push_int(int64_t v) { valid.push_back(1); i64.push_back(v); }
void OutColumn::push_float(double v) { valid.push_back(1); f64.push_back(v); }
void OutColumn::push_str(const std::string& s) { valid.push_back(1); str.push_back(s); }

OutColumn make_out_column(const std::string& name, const std::string& format, bool nullable) {
  OutColumn c;
  c.type = field_from_format(format.c_str());
  c.type.name = name;
  c.type.nullable = nullable;
  return c;
}

namespace {

struct BatchOwner {
  std::vector<std::unique_ptr<uint8_t[]>> blocks;
  // 8-byte columns are not copied: their vectors move in here and the Arrow buffers
  // point at them
  std::vector<std::vector<int64_t>> kept_i64;
  std::vector<std::vector<double>> kept_f64;
  std::vector<std::vector<const void*>> buffer_lists;
  std::vector<ArrowArray> children;
  std::vector<ArrowArray*> child_ptrs;
  const void* struct_buffers[1] = {nullptr};

  uint8_t* block(size_t bytes, bool zero = true) {
    blocks.emplace_back(new uint8_t[bytes + 64]);
    if (zero) std::memset(blocks.back().get(), 0, bytes + 64);
    return blocks.back().get();
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
};
void release_schema_owner(ArrowSchema* s) {
  delete static_cast<SchemaOwner*>(s->private_data);
  s->release = nullptr;
}
void release_noop_schema(ArrowSchema* s) { s->release = nullptr; }

}  // namespace

void export_record_batch(std::vector<OutColumn>& cols, ArrowArray* out) {
  auto owner = std::make_unique<BatchOwner>();
  const size_t n = cols.empty() ? 0 : cols[0].rows();
  owner->children.resize(cols.size());
  owner->buffer_lists.resize(cols.size());
  for (size_t c = 0; c < cols.size(); ++c) {
    OutColumn& col = cols[c];
    if (col.rows() != n) fail(PRISM_GPU_EINTERNAL, "output columns have different lengths");
    ArrowArray& a = owner->children[c];
    std::memset(&a, 0, sizeof(a));
    int64_t nulls = 0;
    uint8_t* bits = nullptr;
    if (n && std::memchr(col.valid.data(), 0, n) != nullptr) {  // bitmap only when there is a NULL
      bits = owner->block((n + 7) / 8);
      for (size_t i = 0; i < n; ++i) {
        if (col.valid[i]) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
        else ++nulls;
      }
    }
    auto& bl = owner->buffer_lists[c];
    bl.push_back(nulls ? bits : nullptr);
    if (col.type.is_string) {
      size_t total = 0;
      for (size_t i = 0; i < n; ++i) total += col.valid[i] ? col.str_at(i).size() : 0;
      if (total > 0x7fffffffull) fail(PRISM_GPU_EUNSUPPORTED, "string output exceeds 2 GiB");
      int32_t* off = (int32_t*)owner->block((n + 1) * 4, false);
      off[0] = 0;
      uint8_t* data = owner->block(total, false);
      size_t pos = 0;
      for (size_t i = 0; i < n; ++i) {
        if (col.valid[i]) {
          const std::string& v = col.str_at(i);
          std::memcpy(data + pos, v.data(), v.size());
          pos += v.size();
        }
        off[i + 1] = (int32_t)pos;
      }
      bl.push_back(off);
      bl.push_back(data);
    } else if (col.type.is_float) {
      owner->kept_f64.push_back(std::move(col.f64));
      bl.push_back(owner->kept_f64.back().data());
    } else if (col.type.width == 8) {
      owner->kept_i64.push_back(std::move(col.i64));
      bl.push_back(owner->kept_i64.back().data());
    } else if (col.type.width == 4) {
      int32_t* v = (int32_t*)owner->block(n * 4);
      for (size_t i = 0; i < n; ++i) v[i] = (int32_t)col.i64[i];
      bl.push_back(v);
    } else {
      int16_t* v = (int16_t*)owner->block(n * 2);
      for (size_t i = 0; i < n; ++i) v[i] = (int16_t)col.i64[i];
      bl.push_back(v);
    }
    a.length = (int64_t)n;
    a.null_count = nulls;
    a.n_buffers = (int64_t)bl.size();
    a.buffers = bl.data();
    a.release = release_noop_array;
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
