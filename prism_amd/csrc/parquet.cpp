// Parquet -> Arrow columns on the host, for files shaped like the ones prism-ingest writes
// (src/prism-ingest/src/ingest.rs:98-101: writer version 2.0, i.e. data page v2; dictionary
// pages with PLAIN fallback; Snappy) and the ones ParquetExec reads for the reference
// (src/prism-query/src/providers/table_provider.rs:110-131).  It replaces, for this path, the
// parquet 47.0.0 crate's reader (an un-vendored dependency, Cargo.lock:2114-2115): the
// published format is restated here -- Thrift compact protocol for the footer and the page
// headers, the RLE / bit-packed hybrid for definition levels and dictionary indices, PLAIN
// values, raw Snappy blocks.
//
// Covered: flat schemas; INT32 / INT64 / BYTE_ARRAY columns, required or optional; data pages
// v1 and v2; PLAIN, PLAIN_DICTIONARY / RLE_DICTIONARY; UNCOMPRESSED and SNAPPY.  Anything else
// (nested columns, other physical types, encodings or codecs) is PRISM_GPU_EUNSUPPORTED and
// names the column.
#include <atomic>
#include <deque>
#include <cstring>
#include <mutex>
#include <thread>
#include <memory>
#include <string>
#include <vector>

#include <unordered_map>

#include "engine.hpp"

namespace prism {

namespace {

[[noreturn]] void bad(const std::string& what) { fail(PRISM_GPU_EINVAL, "parquet: " + what); }
[[noreturn]] void unsupported(const std::string& what) { fail(PRISM_GPU_EUNSUPPORTED, "parquet: " + what); }

// ---- Thrift compact protocol ----------------------------------------------------------------
struct Thrift {
  const uint8_t* p;
  const uint8_t* end;
  uint8_t byte() {
    if (p >= end) bad("truncated metadata");
    return *p++;
  }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      const uint8_t b = byte();
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    bad("varint too long");
  }
  int64_t zigzag() {
    const uint64_t v = varint();
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  std::string binary() {
    const uint64_t n = varint();
    if (n > (uint64_t)(end - p)) bad("truncated string");
    std::string s((const char*)p, (size_t)n);
    p += n;
    return s;
  }
  // Next field of a struct: false at the stop byte.  `type` is the compact type code.
  bool field(int16_t& id, int& type) {
    const uint8_t b = byte();
    if (b == 0) return false;
    type = b & 0x0f;
    const int delta = b >> 4;
    id = delta ? (int16_t)(id + delta) : (int16_t)zigzag();
    return true;
  }
  void list_header(int& elem_type, uint64_t& n) {
    const uint8_t b = byte();
    elem_type = b & 0x0f;
    n = b >> 4;
    if (n == 15) n = varint();
  }
  void skip(int type) {
    switch (type) {
      case 1: case 2: return;  // booleans live in the field header
      case 3: byte(); return;
      case 4: case 5: case 6: varint(); return;
      case 7: for (int i = 0; i < 8; ++i) byte(); return;
      case 8: binary(); return;
      case 9: case 10: {
        int et; uint64_t n;
        list_header(et, n);
        for (uint64_t i = 0; i < n; ++i) {
          if (et == 1 || et == 2) byte();  // booleans in a list take a byte each
          else skip(et);
        }
        return;
      }
      case 11: {
        const uint64_t n = varint();
        if (n == 0) return;
        const uint8_t kv = byte();
        for (uint64_t i = 0; i < n; ++i) { skip(kv >> 4); skip(kv & 0x0f); }
        return;
      }
      case 12: {
        int16_t id = 0; int t;
        while (field(id, t)) skip(t);
        return;
      }
      default: bad("unknown thrift type");
    }
  }
};

// ---- file metadata --------------------------------------------------------------------------
enum Physical { kBoolean = 0, kInt32 = 1, kInt64 = 2, kInt96 = 3, kFloat = 4, kDouble = 5, kByteArray = 6, kFixed = 7 };

struct Leaf {
  std::string name;
  int physical = -1;
  bool optional = true;
  int converted = -1;
  // logicalType union
  bool is_string = false;
  int ts_unit = 0;       // 1 ms, 2 us, 3 ns
  int int_bits = 0;      // INTEGER(bitWidth, isSigned)
  bool int_signed = true;
};

struct ChunkMeta {
  int physical = -1;
  int codec = 0;
  int64_t num_values = 0, total_compressed = 0, data_page_offset = 0, dictionary_page_offset = 0;
};

struct RowGroupMeta {
  int64_t num_rows = 0;
  std::vector<ChunkMeta> chunks;
};

struct FileMeta {
  std::vector<Leaf> leaves;
  std::vector<RowGroupMeta> groups;
  int64_t num_rows = 0;
};

void read_logical_type(Thrift& t, Leaf& leaf) {
  int16_t id = 0; int ty;
  while (t.field(id, ty)) {
    if (id == 1 && ty == 12) { leaf.is_string = true; t.skip(ty); }
    else if (id == 8 && ty == 12) {  // TimestampType {1: isAdjustedToUTC, 2: unit}
      int16_t id2 = 0; int ty2;
      while (t.field(id2, ty2)) {
        if (id2 == 2 && ty2 == 12) {
          int16_t id3 = 0; int ty3;
          while (t.field(id3, ty3)) { leaf.ts_unit = id3; t.skip(ty3); }
        } else t.skip(ty2);
      }
    } else if (id == 10 && ty == 12) {  // IntType {1: bitWidth (byte), 2: isSigned}
      int16_t id2 = 0; int ty2;
      while (t.field(id2, ty2)) {
        if (id2 == 1 && ty2 == 3) leaf.int_bits = (int8_t)t.byte();
        else if (id2 == 2 && (ty2 == 1 || ty2 == 2)) leaf.int_signed = ty2 == 1;
        else t.skip(ty2);
      }
    } else t.skip(ty);
  }
}

FileMeta read_footer(const uint8_t* file, size_t len) {
  if (len < 12 || std::memcmp(file, "PAR1", 4) != 0 || std::memcmp(file + len - 4, "PAR1", 4) != 0) bad("not a Parquet file (magic bytes)");
  uint32_t meta_len;
  std::memcpy(&meta_len, file + len - 8, 4);
  if ((uint64_t)meta_len + 12 > len) bad("footer length beyond the file");
  Thrift t{file + len - 8 - meta_len, file + len - 8};
  FileMeta fm;
  int root_children = -1;
  int16_t id = 0; int ty;
  while (t.field(id, ty)) {
    if (id == 2 && ty == 9) {  // schema
      int et; uint64_t n;
      t.list_header(et, n);
      for (uint64_t i = 0; i < n; ++i) {
        Leaf leaf;
        int num_children = 0, repetition = 0;
        int16_t id2 = 0; int ty2;
        while (t.field(id2, ty2)) {
          if (id2 == 1) leaf.physical = (int)t.zigzag();
          else if (id2 == 3) repetition = (int)t.zigzag();
          else if (id2 == 4) leaf.name = t.binary();
          else if (id2 == 5) num_children = (int)t.zigzag();
          else if (id2 == 6) leaf.converted = (int)t.zigzag();
          else if (id2 == 10 && ty2 == 12) read_logical_type(t, leaf);
          else t.skip(ty2);
        }
        if (i == 0) { root_children = num_children; continue; }
        if (num_children > 0) unsupported("nested column `" + leaf.name + "`");
        if (repetition == 2) unsupported("repeated column `" + leaf.name + "`");
        leaf.optional = repetition == 1;
        fm.leaves.push_back(leaf);
      }
    } else if (id == 3) {
      fm.num_rows = t.zigzag();
    } else if (id == 4 && ty == 9) {  // row groups
      int et; uint64_t n;
      t.list_header(et, n);
      for (uint64_t g = 0; g < n; ++g) {
        RowGroupMeta rg;
        int16_t id2 = 0; int ty2;
        while (t.field(id2, ty2)) {
          if (id2 == 1 && ty2 == 9) {
            int et2; uint64_t nc;
            t.list_header(et2, nc);
            for (uint64_t c = 0; c < nc; ++c) {
              ChunkMeta cm;
              int16_t id3 = 0; int ty3;
              while (t.field(id3, ty3)) {
                if (id3 == 3 && ty3 == 12) {  // ColumnMetaData
                  int16_t id4 = 0; int ty4;
                  while (t.field(id4, ty4)) {
                    if (id4 == 1) cm.physical = (int)t.zigzag();
                    else if (id4 == 4) cm.codec = (int)t.zigzag();
                    else if (id4 == 5) cm.num_values = t.zigzag();
                    else if (id4 == 7) cm.total_compressed = t.zigzag();
                    else if (id4 == 9) cm.data_page_offset = t.zigzag();
                    else if (id4 == 11) cm.dictionary_page_offset = t.zigzag();
                    else t.skip(ty4);
                  }
                } else t.skip(ty3);
              }
              rg.chunks.push_back(cm);
            }
          } else if (id2 == 3) rg.num_rows = t.zigzag();
          else t.skip(ty2);
        }
        fm.groups.push_back(rg);
      }
    } else t.skip(ty);
  }
  if (root_children < 0 || (size_t)root_children != fm.leaves.size()) bad("schema is not a flat list of columns");
  for (const RowGroupMeta& rg : fm.groups)
    if (rg.chunks.size() != fm.leaves.size()) bad("a row group does not hold every column");
  return fm;
}

// The Arrow type a leaf reads as (what arrow-rs gives ParquetExec for it).
FieldType leaf_field(const Leaf& l) {
  const char* fmt = nullptr;
  if (l.physical == kInt64) {
    const bool ts = l.ts_unit != 0 || l.converted == 9 || l.converted == 10;
    fmt = ts ? "tsm:" : "l";
    if (!ts && (l.converted == 14 || (l.int_bits == 64 && !l.int_signed))) unsupported("UInt64 column `" + l.name + "`");
  } else if (l.physical == kInt32) {
    int bits = l.int_bits, sign = l.int_signed;
    if (!bits) {
      if (l.converted == 15) { bits = 8; sign = true; } else if (l.converted == 16) { bits = 16; sign = true; }
      else if (l.converted == 11) { bits = 8; sign = false; } else if (l.converted == 12) { bits = 16; sign = false; }
      else if (l.converted == 13) { bits = 32; sign = false; } else { bits = 32; sign = true; }
    }
    if (bits == 32 && !sign) unsupported("UInt32 column `" + l.name + "`");
    fmt = bits <= 16 ? (sign ? "s" : "S") : "i";
  } else if (l.physical == kByteArray) {
    fmt = (l.is_string || l.converted == 0) ? "u" : "z";
  } else {
    unsupported("column `" + l.name + "` has a physical type other than INT32 / INT64 / BYTE_ARRAY");
  }
  FieldType f = field_from_format(fmt);
  f.name = l.name;
  f.nullable = true;
  return f;
}

// ---- Snappy (raw block format) --------------------------------------------------------------
// Into dst, where `expect` bytes and kSnappySlack more may be written.
constexpr size_t kSnappySlack = 16;
void snappy_decompress_to(const uint8_t* src, size_t n, uint8_t* dst, size_t expect);

void snappy_decompress(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
  out.resize(expect + kSnappySlack);
  snappy_decompress_to(src, n, out.data(), expect);
  out.resize(expect);
}

void snappy_decompress_to(const uint8_t* src, size_t n, uint8_t* const dst, size_t expect) {
  Thrift v{src, src + n};
  const uint64_t total = v.varint();
  if (total != expect) bad("Snappy block announces another size than the page header");
  // Sixteen bytes of slack behind the block: short literals and copies are moved as two 8-byte words
  // whatever their length (what they write beyond it is overwritten by the next element, or cut off
  // at the end).  Sorted timestamps compress into a copy of five or six bytes and a literal of two
  // or three per value: with a library call per element, millions of them per column chunk, this
  // loop was most of the loader's time.
  const size_t cap = (size_t)total;
  const uint8_t* p = v.p;
  const uint8_t* end = src + n;
  size_t o = 0;
  while (p < end) {
    const uint8_t tag = *p++;
    if ((tag & 3) == 0) {
      size_t l = tag >> 2;
      if (l >= 60) {
        const int extra = (int)l - 59;
        if (end - p < extra) bad("truncated Snappy literal");
        l = 0;
        for (int i = 0; i < extra; ++i) l |= (size_t)p[i] << (8 * i);
        p += extra;
      }
      l += 1;
      if ((size_t)(end - p) < l || o + l > cap) bad("Snappy literal overruns");
      if (l <= 16 && (size_t)(end - p) >= 16) {
        std::memcpy(dst + o, p, 8);
        std::memcpy(dst + o + 8, p + 8, 8);
      } else {
        std::memcpy(dst + o, p, l);
      }
      p += l;
      o += l;
    } else {
      size_t l, off;
      if ((tag & 3) == 1) {
        if (end - p < 1) bad("truncated Snappy copy");
        l = ((tag >> 2) & 7) + 4;
        off = ((size_t)(tag >> 5) << 8) | *p++;
      } else if ((tag & 3) == 2) {
        if (end - p < 2) bad("truncated Snappy copy");
        l = (tag >> 2) + 1;
        off = (size_t)p[0] | ((size_t)p[1] << 8);
        p += 2;
      } else {
        if (end - p < 4) bad("truncated Snappy copy");
        l = (tag >> 2) + 1;
        off = (size_t)p[0] | ((size_t)p[1] << 8) | ((size_t)p[2] << 16) | ((size_t)p[3] << 24);
        p += 4;
      }
      if (off == 0 || off > o || o + l > cap) bad("Snappy copy out of range");
      uint8_t* d = dst + o;
      const uint8_t* s = d - off;
      if (off >= 8) {
        // word by word, front to back: a word read behind the ones just written sees them, as a copy that overlaps itself must
        // (l <= 64: at most 7 bytes beyond it are written, inside the slack)
        for (size_t i = 0; i < l; i += 8) std::memcpy(d + i, s + i, 8);
      } else {
        for (size_t i = 0; i < l; ++i) d[i] = s[i];  // a pattern shorter than a word: byte by byte
      }
      o += l;
    }
  }
  if (o != cap) bad("Snappy block shorter than announced");
}

// ---- RLE / bit-packed hybrid ------------------------------------------------------------------
// Decodes `count` values of `bit_width` bits from [p, end): emit(i, value) for i = 0 .. count - 1, in order.
template <typename Emit>
void rle_decode_each(const uint8_t* p, const uint8_t* end, int bit_width, size_t count, Emit&& emit) {
  if (bit_width == 0) {
    for (size_t i = 0; i < count; ++i) emit(i, 0u);
    return;
  }
  if (bit_width > 32) bad("bit width above 32");
  size_t n = 0;
  Thrift v{p, end};
  const int value_bytes = (bit_width + 7) / 8;
  while (n < count) {
    const uint64_t h = v.varint();
    if (h & 1) {  // bit-packed: groups of 8 values
      const uint64_t groups = h >> 1;
      // (checked before multiplying: a huge group count would wrap the byte count round to a small one)
      if (groups > (uint64_t)(v.end - v.p) / (uint64_t)bit_width) bad("bit-packed run beyond the page");
      const uint64_t bytes = groups * (uint64_t)bit_width;
      const uint8_t* b = v.p;
      uint64_t bitpos = 0;
      const uint64_t values = groups * 8;
      const uint64_t mask = (1ull << bit_width) - 1ull;
      const uint64_t take = std::min<uint64_t>(values, count - n);
      uint64_t i = 0;
      // (whole 8-byte loads while they stay inside the run, then byte-exact ones)
      for (; i < take && (bitpos >> 3) + 8 <= bytes; ++i, bitpos += bit_width) {
        uint64_t word;
        std::memcpy(&word, b + (bitpos >> 3), 8);
        emit(n++, (uint32_t)((word >> (bitpos & 7)) & mask));
      }
      for (; i < take; ++i, bitpos += bit_width) {
        uint64_t word = 0;
        const uint64_t byte0 = bitpos >> 3;
        std::memcpy(&word, b + byte0, (size_t)std::min<uint64_t>(8, bytes - byte0));
        emit(n++, (uint32_t)((word >> (bitpos & 7)) & mask));
      }
      v.p += bytes;
    } else {
      const uint64_t run = h >> 1;
      if (v.end - v.p < value_bytes) bad("RLE run beyond the page");
      uint32_t value = 0;
      std::memcpy(&value, v.p, value_bytes);
      v.p += value_bytes;
      for (uint64_t i = 0; i < run && n < count; ++i) emit(n++, value);
      if (run == 0) bad("empty RLE run");
    }
  }
}

void rle_decode(const uint8_t* p, const uint8_t* end, int bit_width, size_t count, std::vector<uint32_t>& out) {
  out.resize(count);
  uint32_t* const o = out.data();
  rle_decode_each(p, end, bit_width, count, [o](size_t i, uint32_t value) { o[i] = value; });
}

// The runs of `count` values at [p, end) without expanding them: appended to `runs` (rows from
// first_row on), the bit-packed values' bytes to `packed`.  A repeated value is checked against the
// dictionary here; bit-packed ones are by whoever expands them.
void rle_runs(const uint8_t* p, const uint8_t* end, int bit_width, size_t count, uint64_t first_row, size_t dict_size,
              std::vector<PackedRun>& runs, std::vector<uint8_t>& packed) {
  if (bit_width > 32) bad("bit width above 32");
  if (first_row + count > 0xffffffffull) unsupported("more than 2^32 rows in one row group");
  if (bit_width == 0) {
    if (count && dict_size == 0) bad("dictionary index out of range");
    if (count) runs.push_back({(uint32_t)first_row, (uint32_t)count, 0u, 0u});
    return;
  }
  size_t n = 0;
  Thrift v{p, end};
  const int value_bytes = (bit_width + 7) / 8;
  while (n < count) {
    const uint64_t h = v.varint();
    if (h & 1) {
      const uint64_t groups = h >> 1;
      if (groups == 0) bad("empty bit-packed run");
      if (groups > (uint64_t)(v.end - v.p) / (uint64_t)bit_width) bad("bit-packed run beyond the page");
      const uint64_t bytes = groups * (uint64_t)bit_width, values = std::min<uint64_t>(groups * 8, count - n);
      if (packed.size() + bytes > 0xfffffff0ull) unsupported("more than 4 GiB of packed indices in one column chunk");
      runs.push_back({(uint32_t)(first_row + n), (uint32_t)values, 0x80000000u | (uint32_t)bit_width, (uint32_t)packed.size()});
      packed.insert(packed.end(), v.p, v.p + bytes);
      v.p += bytes;
      n += (size_t)values;
    } else {
      const uint64_t run = h >> 1;
      if (run == 0) bad("empty RLE run");
      if (v.end - v.p < value_bytes) bad("RLE run beyond the page");
      uint32_t value = 0;
      std::memcpy(&value, v.p, value_bytes);
      v.p += value_bytes;
      if (value >= dict_size) bad("dictionary index out of range");
      const uint64_t take = std::min<uint64_t>(run, count - n);
      runs.push_back({(uint32_t)(first_row + n), (uint32_t)take, (uint32_t)bit_width, value});
      n += (size_t)take;
    }
  }
}

// ---- page-locked blocks for the decoded values of large column chunks ---------------------------
// The loader (table_append_parquet) decodes a fixed-width column chunk straight into page-locked
// memory: table_append then hands such a buffer to the copy engine as it is, where a pageable one
// goes through the context's staging ring first -- one thread at a time copying at memcpy speed,
// which for sixteen files appended at once was as long as their decode.  The blocks are kept
// for the next chunk (at most kMaxRetained bytes of them); the host-only reader never asks.
struct PinnedBlocks {
  static constexpr size_t kMaxRetained = 2ull << 30;
  static constexpr size_t kMinBytes = 1u << 20;
  std::mutex mu;
  std::unordered_map<size_t, std::vector<void*>> free_blocks;  // by size class
  std::unordered_map<void*, size_t> out;                       // handed out: block -> size class
  size_t retained = 0;
  static size_t size_class(size_t bytes) {
    size_t c = kMinBytes;
    while (c < bytes) c <<= 1;
    return c;
  }
  void* take(size_t bytes) {
    const size_t cls = size_class(bytes);
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = free_blocks.find(cls);
      if (it != free_blocks.end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        retained -= cls;
        out.emplace(p, cls);
        return p;
      }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, cls, hipHostMallocPortable) != hipSuccess || !p) {
      (void)hipGetLastError();  // (no page-locked memory to be had: the caller takes ordinary memory)
      return nullptr;
    }
    std::lock_guard<std::mutex> g(mu);
    out.emplace(p, cls);
    return p;
  }
  // false: not one of ours
  bool give(void* p) {
    size_t cls;
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = out.find(p);
      if (it == out.end()) return false;
      cls = it->second;
      out.erase(it);
      if (retained + cls <= kMaxRetained) {
        free_blocks[cls].push_back(p);
        retained += cls;
        return true;
      }
    }
    (void)hipHostFree(p);
    return true;
  }
};
PinnedBlocks& pinned_blocks() {
  static PinnedBlocks* pool = new PinnedBlocks();  // (never destroyed: no HIP call from a static destructor)
  return *pool;
}
thread_local bool tls_pin_values = false;  // set by the loader's decode threads

template <typename T>
struct ValuesAlloc {
  using value_type = T;
  ValuesAlloc() = default;
  template <typename U>
  ValuesAlloc(const ValuesAlloc<U>&) {}
  T* allocate(size_t n) {
    const size_t bytes = n * sizeof(T);
    if (tls_pin_values && bytes >= PinnedBlocks::kMinBytes)
      if (void* p = pinned_blocks().take(bytes)) return (T*)p;
    void* p = std::malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return (T*)p;
  }
  void deallocate(T* p, size_t) {
    if (!pinned_blocks().give(p)) std::free(p);
  }
  // resize() leaves new elements uninitialised (the decoder writes every row it reports);
  // assign(n, 0) and the like still store what they are given
  template <typename U>
  void construct(U* p) noexcept { ::new ((void*)p) U; }
  template <typename U, typename... A>
  void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
  template <typename U>
  bool operator==(const ValuesAlloc<U>&) const { return true; }
  template <typename U>
  bool operator!=(const ValuesAlloc<U>&) const { return false; }
};

// ---- one column of one row group ----------------------------------------------------------------
struct PqColumn {
  FieldType type;
  std::vector<uint8_t, ValuesAlloc<uint8_t>> values;  // fixed width: type.width bytes per row (the loader's: page-locked when large)
  std::vector<int32_t> offsets;   // strings: rows + 1
  std::vector<uint8_t> bytes;
  std::vector<uint8_t> validity;  // bit per row, all rows present
  int64_t nulls = 0;
  int64_t rows = 0;
  // A string column whose every data page is dictionary encoded may stay in that form (for the
  // table loader, which expands it on the device): one index per row (0 where the row is NULL)
  // and the chunk's dictionary, instead of offsets and bytes.
  bool dict_form = false;
  // ... and, for the loader that expands them on the device, the indices may stay the runs of
  // the file's RLE / bit-packed hybrid (chunks without NULLs): `index` is empty then.
  bool packed_form = false;
  std::vector<uint8_t> packed;
  std::vector<PackedRun> runs;
  PackedIndices packed_view{};
  std::vector<int32_t> index;
  std::vector<int32_t> dict_offsets;  // entries + 1
  std::vector<uint8_t> dict_bytes;
};

struct PageHeader {
  int type = -1;
  int32_t uncompressed = 0, compressed = 0;
  int32_t num_values = 0, encoding = 0;
  // v2
  int32_t def_len = 0, rep_len = 0;
  bool is_compressed = true;
};

PageHeader read_page_header(Thrift& t) {
  PageHeader h;
  int16_t id = 0; int ty;
  while (t.field(id, ty)) {
    if (id == 1) h.type = (int)t.zigzag();
    else if (id == 2) h.uncompressed = (int32_t)t.zigzag();
    else if (id == 3) h.compressed = (int32_t)t.zigzag();
    else if ((id == 5 || id == 7) && ty == 12) {  // DataPageHeader / DictionaryPageHeader
      int16_t id2 = 0; int ty2;
      while (t.field(id2, ty2)) {
        if (id2 == 1) h.num_values = (int32_t)t.zigzag();
        else if (id2 == 2) h.encoding = (int)t.zigzag();
        else t.skip(ty2);
      }
    } else if (id == 8 && ty == 12) {  // DataPageHeaderV2
      int16_t id2 = 0; int ty2;
      while (t.field(id2, ty2)) {
        if (id2 == 1) h.num_values = (int32_t)t.zigzag();
        else if (id2 == 4) h.encoding = (int)t.zigzag();
        else if (id2 == 5) h.def_len = (int32_t)t.zigzag();
        else if (id2 == 6) h.rep_len = (int32_t)t.zigzag();
        else if (id2 == 7 && (ty2 == 1 || ty2 == 2)) h.is_compressed = ty2 == 1;
        else t.skip(ty2);
      }
    } else t.skip(ty);
  }
  return h;
}

struct Dictionary {
  std::vector<int64_t> ints;                          // INT32 / INT64
  std::vector<std::pair<uint32_t, uint32_t>> spans;   // BYTE_ARRAY: offset, length into `bytes`
  std::vector<uint8_t> bytes;
  size_t size() const { return ints.empty() ? spans.size() : ints.size(); }
};

// Divisor that brings a timestamp of the leaf's unit to milliseconds (as arrow's cast does).
int64_t timestamp_divisor(const Leaf& leaf) {
  if (leaf.ts_unit == 2 || (leaf.ts_unit == 0 && leaf.converted == 10)) return 1000;  // us
  if (leaf.ts_unit == 3) return 1000000;                                               // ns
  return 1;
}

void set_bits(std::vector<uint8_t>& bits, int64_t from, int64_t n) {
  int64_t i = from;
  const int64_t end = from + n;
  for (; i < end && (i & 7); ++i) bits[(size_t)i >> 3] |= (uint8_t)(1u << (i & 7));
  if (end - i >= 8) {
    std::memset(&bits[(size_t)i >> 3], 0xff, (size_t)((end - i) >> 3));
    i += (end - i) & ~(int64_t)7;
  }
  for (; i < end; ++i) bits[(size_t)i >> 3] |= (uint8_t)(1u << (i & 7));
}

void decode_chunk(const uint8_t* file, size_t len, const Leaf& leaf, const ChunkMeta& cm, int64_t rows, PqColumn& col, bool keep_dict = false,
                  bool keep_packed = false) {
  if (cm.codec != 0 && cm.codec != 1) unsupported("column `" + leaf.name + "` uses a codec other than UNCOMPRESSED / SNAPPY");
  if (cm.physical != leaf.physical) bad("column chunk and schema disagree on the type of `" + leaf.name + "`");
  const bool strings = leaf.physical == kByteArray;
  int64_t pos = cm.data_page_offset;
  if (cm.dictionary_page_offset > 0 && cm.dictionary_page_offset < pos) pos = cm.dictionary_page_offset;
  if (pos < 4 || pos > (int64_t)len || cm.total_compressed < 0 || cm.total_compressed > (int64_t)len - pos)
    bad("column chunk of `" + leaf.name + "` lies outside the file");
  if (rows < 0 || cm.num_values < 0) bad("negative row or value count in `" + leaf.name + "`");
  const int64_t chunk_end = pos + cm.total_compressed;
  Dictionary dict;
  std::vector<uint8_t> page, levels_buf;
  std::vector<uint32_t> def, idx;
  int64_t seen = 0;
  bool as_index = strings && keep_dict;  // until a data page turns out not to be dictionary encoded
  bool as_packed = as_index && keep_packed;  // ... and, while no page holds a NULL, not even expanded
  int64_t seen_rows = 0;
  // (a page that cannot stay packed: what was kept as runs becomes indices, and the chunk goes on that way)
  auto unpack = [&]() {
    if (!as_packed) return;
    as_packed = false;
    col.packed.insert(col.packed.end(), 8, (uint8_t)0);
    const PackedIndices view{col.packed.data(), col.packed.size(), col.runs.data(), col.runs.size()};
    packed_indices_expand(view, seen_rows, col.index);
    col.packed.clear();
    col.runs.clear();
  };
  const int w = col.type.width;
  const int64_t divisor = strings ? 1 : timestamp_divisor(leaf);
  if (strings) {
    if (as_index) col.index.reserve((size_t)rows);
    else { col.offsets.reserve((size_t)rows + 1); col.offsets.push_back(0); }
  } else {
    col.values.resize((size_t)rows * (size_t)w);  // (uninitialised: every page writes its rows, zeros where they are NULL)
  }
  col.validity.assign((size_t)(rows + 7) / 8, 0);
  auto append_string = [&](const uint8_t* p, uint32_t l) {
    if (col.bytes.size() + l > 0x7fffffffull) unsupported("more than 2 GiB of string data in one row group of `" + leaf.name + "`");
    col.bytes.insert(col.bytes.end(), p, p + l);
    col.offsets.push_back((int32_t)col.bytes.size());
  };
  while (seen < cm.num_values) {
    if (pos >= chunk_end) bad("column chunk of `" + leaf.name + "` ends before its values do");
    Thrift t{file + pos, file + chunk_end};
    const PageHeader h = read_page_header(t);
    const uint8_t* body = t.p;
    if (h.compressed < 0 || h.uncompressed < 0 || (int64_t)h.compressed > (file + chunk_end) - body) bad("page of `" + leaf.name + "` lies outside its chunk");
    if (h.num_values < 0 || h.def_len < 0 || h.rep_len < 0) bad("negative count or length in a page header of `" + leaf.name + "`");
    pos = (int64_t)(body - file) + h.compressed;
    if (h.type != 0 && h.type != 2 && h.type != 3) continue;  // index pages and the like

    // the page's value bytes, decompressed
    const uint8_t* vals = nullptr;
    const uint8_t* vals_end = nullptr;
    const uint8_t* def_p = nullptr;
    const uint8_t* def_end = nullptr;
    bool def_ready = false;  // the page's definition levels are in `def` already
    if (h.type == 3) {
      if (h.rep_len != 0) unsupported("repetition levels in `" + leaf.name + "`");
      const int64_t lv = (int64_t)h.def_len + h.rep_len;
      if (lv > h.compressed || lv > h.uncompressed) bad("levels longer than the page");
      def_p = body;
      def_end = body + h.def_len;
      const uint8_t* data = body + lv;
      const size_t data_comp = (size_t)(h.compressed - lv), data_raw = (size_t)(h.uncompressed - lv);
      // A page of PLAIN values of the column's own width with no row missing is decompressed
      // where its values belong (the levels of a v2 page lie in front of the compressed block, so
      // that is known by now); the decoder's slack falls on the rows of the pages to come.
      const size_t nv = (size_t)h.num_values;
      bool in_place = cm.codec == 1 && h.is_compressed && !strings && h.encoding == 0 && divisor == 1 &&
                      w == (leaf.physical == kInt32 ? 4 : 8) && data_raw > 0 && data_raw == nv * (size_t)w &&
                      seen + (int64_t)nv <= rows && ((size_t)seen + nv) * (size_t)w + kSnappySlack <= col.values.size();
      if (in_place && leaf.optional) {
        rle_decode(def_p, def_end, 1, nv, def);
        def_ready = true;
        for (size_t i = 0; i < nv && in_place; ++i) in_place = def[i] != 0;
      }
      if (in_place) {
        uint8_t* out = col.values.data() + (size_t)seen * (size_t)w;
        snappy_decompress_to(data, data_comp, out, data_raw);
        vals = out;
        vals_end = out + data_raw;
      } else if (cm.codec == 1 && h.is_compressed && data_raw > 0) {
        snappy_decompress(data, data_comp, page, data_raw);
        vals = page.data();
        vals_end = vals + page.size();
      } else {
        vals = data;
        vals_end = data + data_comp;
      }
    } else {
      if (cm.codec == 1) {
        snappy_decompress(body, (size_t)h.compressed, page, (size_t)h.uncompressed);
        vals = page.data();
        vals_end = vals + page.size();
      } else {
        vals = body;
        vals_end = body + h.compressed;
      }
      if (h.type == 0 && leaf.optional) {  // v1: 4-byte length, then the hybrid
        if (vals_end - vals < 4) bad("page too short for its definition levels");
        uint32_t l;
        std::memcpy(&l, vals, 4);
        if ((uint64_t)l + 4 > (uint64_t)(vals_end - vals)) bad("definition levels longer than the page");
        def_p = vals + 4;
        def_end = def_p + l;
        vals = def_end;
      }
    }

    if (h.type == 2) {  // dictionary page: PLAIN values
      if (h.encoding != 0 && h.encoding != 2) unsupported("dictionary page encoding of `" + leaf.name + "`");
      const uint8_t* p = vals;
      for (int32_t i = 0; i < h.num_values; ++i) {
        if (strings) {
          if (vals_end - p < 4) bad("dictionary page too short");
          uint32_t l;
          std::memcpy(&l, p, 4);
          p += 4;
          if ((uint64_t)l > (uint64_t)(vals_end - p)) bad("dictionary value beyond the page");
          dict.spans.emplace_back((uint32_t)dict.bytes.size(), l);
          dict.bytes.insert(dict.bytes.end(), p, p + l);
          p += l;
        } else {
          const int w = leaf.physical == kInt32 ? 4 : 8;
          if (vals_end - p < w) bad("dictionary page too short");
          int64_t v = 0;
          if (w == 4) { int32_t x; std::memcpy(&x, p, 4); v = x; } else std::memcpy(&v, p, 8);
          dict.ints.push_back(v);
          p += w;
        }
      }
      continue;
    }

    // ---- a data page ----
    const size_t n = (size_t)h.num_values;
    if (seen + (int64_t)n > rows) bad("more values than rows in `" + leaf.name + "`");
    size_t present = n;
    if (leaf.optional) {
      if (!def_ready) rle_decode(def_p, def_end, 1, n, def);
      present = 0;
      for (size_t i = 0; i < n; ++i) present += def[i];
    }
    const bool by_dict = h.encoding == 2 || h.encoding == 8;
    bool gather_page = false;
    if (!by_dict && h.encoding != 0) unsupported("encoding " + std::to_string(h.encoding) + " in column `" + leaf.name + "`");
    seen_rows = seen;
    const bool pack_page = as_packed && by_dict && present == n;
    if (!pack_page) unpack();
    if (by_dict) {
      if (vals_end - vals < 1) bad("dictionary-encoded page without a bit width");
      // (integers with no row missing: the indices are decoded and looked up in one pass, further down)
      gather_page = !strings && present == n && divisor == 1 && (w == 8 || w == 4);
      if (pack_page) {
        rle_runs(vals + 1, vals_end, vals[0], present, (uint64_t)seen, dict.size(), col.runs, col.packed);
      } else if (!gather_page) {
        rle_decode(vals + 1, vals_end, vals[0], present, idx);
        uint32_t largest = 0;
        for (uint32_t k : idx) largest = k > largest ? k : largest;
        if (!idx.empty() && largest >= dict.size()) bad("dictionary index out of range in `" + leaf.name + "`");
      }
    }
    if (present == n) set_bits(col.validity, seen, (int64_t)n);
    else {
      for (size_t i = 0; i < n; ++i)
        if (def[i]) { const int64_t row = seen + (int64_t)i; col.validity[(size_t)row >> 3] |= (uint8_t)(1u << (row & 7)); }
      col.nulls += (int64_t)(n - present);
    }
    const bool dense = present == n;
    const uint8_t* p = vals;
    size_t k = 0;  // position among the present values
    if (strings) {
      if (as_index && !by_dict) {
        // a page that fell back to PLAIN: the rows so far are expanded, and the chunk is decoded to offsets and bytes
        col.offsets.reserve((size_t)rows + 1);
        col.offsets.push_back(0);
        for (int64_t r = 0; r < seen; ++r) {
          const bool ok = col.validity[(size_t)r >> 3] >> (r & 7) & 1;
          if (ok) append_string(dict.bytes.data() + dict.spans[(size_t)col.index[(size_t)r]].first, dict.spans[(size_t)col.index[(size_t)r]].second);
          else col.offsets.push_back((int32_t)col.bytes.size());
        }
        col.index.clear();
        col.index.shrink_to_fit();
        as_index = false;
      }
      if (pack_page) {
        // (nothing to do: the page's runs were kept)
      } else if (as_index) {
        const size_t at = col.index.size();
        col.index.resize(at + n);
        int32_t* out = col.index.data() + at;
        if (dense) for (size_t i = 0; i < n; ++i) out[i] = (int32_t)idx[i];
        else for (size_t i = 0; i < n; ++i) out[i] = def[i] ? (int32_t)idx[k++] : 0;
      } else {
        for (size_t i = 0; i < n; ++i) {
          if (!dense && !def[i]) { col.offsets.push_back((int32_t)col.bytes.size()); continue; }
          if (by_dict) {
            append_string(dict.bytes.data() + dict.spans[idx[k]].first, dict.spans[idx[k]].second);
          } else {
            if (vals_end - p < 4) bad("page of `" + leaf.name + "` ends inside a value");
            uint32_t l;
            std::memcpy(&l, p, 4);
            p += 4;
            if ((uint64_t)l > (uint64_t)(vals_end - p)) bad("value beyond the page in `" + leaf.name + "`");
            append_string(p, l);
            p += l;
          }
          ++k;
        }
      }
    } else {
      const int pw = leaf.physical == kInt32 ? 4 : 8;
      uint8_t* out = col.values.data() + (size_t)seen * (size_t)w;
      if (!by_dict && (uint64_t)(vals_end - p) / (uint64_t)pw < (uint64_t)present) bad("page of `" + leaf.name + "` ends inside a value");
      if (!by_dict && dense && divisor == 1 && pw == w) {
        if (p != out) std::memcpy(out, p, n * (size_t)w);  // PLAIN values of the column's own width, none missing (decompressed in place where that could be)
      } else if (gather_page) {
        // dictionary indices, none missing: decoded and looked up in one pass (through a vector of
        // indices, and with the general loop's branches per row before that, this took as long as
        // everything else of the column chunk together).  An index beyond the dictionary reads
        // entry 0 and is reported once the page is through.
        if (dict.ints.empty()) bad("dictionary index out of range in `" + leaf.name + "`");
        const int64_t* d = dict.ints.data();
        const uint32_t dn = (uint32_t)std::min<size_t>(dict.ints.size(), 0xffffffffu);
        uint32_t largest = 0;
        if (w == 8) {
          rle_decode_each(vals + 1, vals_end, vals[0], n, [&](size_t i, uint32_t k) {
            largest = k > largest ? k : largest;
            const int64_t v = d[k < dn ? k : 0];
            std::memcpy(out + i * 8, &v, 8);  // (the column's buffer is a byte vector: 8-byte stores through memcpy)
          });
        } else {
          rle_decode_each(vals + 1, vals_end, vals[0], n, [&](size_t i, uint32_t k) {
            largest = k > largest ? k : largest;
            const int32_t v = (int32_t)d[k < dn ? k : 0];
            std::memcpy(out + i * 4, &v, 4);
          });
        }
        if (largest >= dn) bad("dictionary index out of range in `" + leaf.name + "`");
      } else {
        for (size_t i = 0; i < n; ++i) {
          if (!dense && !def[i]) {
            std::memset(out + i * (size_t)w, 0, (size_t)w);
            continue;
          }
          int64_t v;
          if (by_dict) v = dict.ints[idx[k]];
          else if (pw == 4) { int32_t x; std::memcpy(&x, p + k * 4, 4); v = x; }
          else std::memcpy(&v, p + k * 8, 8);
          ++k;
          if (divisor != 1) v /= divisor;
          std::memcpy(out + i * (size_t)w, &v, (size_t)w);  // little endian: the low bytes are the narrow value
        }
      }
    }
    seen += (int64_t)n;
  }
  if (seen != rows) bad("column `" + leaf.name + "` holds " + std::to_string(seen) + " values for " + std::to_string(rows) + " rows");
  col.rows = rows;
  if (as_packed) {
    col.packed_form = true;
    col.packed.insert(col.packed.end(), 8, (uint8_t)0);  // (whoever expands the runs reads 8 bytes at a time)
  }
  if (as_index) {
    col.dict_form = true;
    col.dict_offsets.reserve(dict.spans.size() + 1);
    for (const auto& sp : dict.spans) col.dict_offsets.push_back((int32_t)sp.first);  // (the spans lie in dict.bytes in order, end to end)
    col.dict_offsets.push_back((int32_t)dict.bytes.size());
    col.dict_bytes = std::move(dict.bytes);
  }
}

// ---- Arrow export of decoded row groups ---------------------------------------------------------
struct PqOwner {
  std::vector<PqColumn> cols;
  std::vector<ArrowArray> children;
  std::vector<ArrowArray> dictionaries;  // of the children in dictionary form
  std::vector<ArrowArray*> child_ptrs;
  std::vector<std::vector<const void*>> buffers;
  const void* top_buffers[1] = {nullptr};
};

void release_pq(ArrowArray* a) {
  delete (PqOwner*)a->private_data;
  a->release = nullptr;
}
void release_pq_child(ArrowArray* a) { a->release = nullptr; }
// (the release function of a child whose indices are still packed: what parquet_packed_indices tells such a child by)
void release_pq_packed(ArrowArray* a) { a->release = nullptr; }

void export_columns(std::vector<PqColumn>&& cols, int64_t rows, ArrowArray* out) {
  auto owner = std::make_unique<PqOwner>();
  owner->cols = std::move(cols);
  const size_t n = owner->cols.size();
  owner->children.resize(n);
  owner->child_ptrs.resize(n);
  owner->buffers.resize(2 * n);
  owner->dictionaries.resize(n);
  for (size_t c = 0; c < n; ++c) {
    PqColumn& col = owner->cols[c];
    ArrowArray& a = owner->children[c];
    std::memset(&a, 0, sizeof(a));
    auto& b = owner->buffers[c];
    b.push_back(col.nulls ? (const void*)col.validity.data() : nullptr);
    if (col.dict_form) {
      // Int32 indices; the values are the dictionary child (a Utf8 / Binary array without NULLs)
      b.push_back(col.packed_form ? nullptr : col.index.empty() ? (const void*)"" : (const void*)col.index.data());
      if (col.packed_form) col.packed_view = PackedIndices{col.packed.data(), col.packed.size(), col.runs.data(), col.runs.size()};
      ArrowArray& d = owner->dictionaries[c];
      std::memset(&d, 0, sizeof(d));
      auto& db = owner->buffers[n + c];
      db.push_back(nullptr);
      db.push_back(col.dict_offsets.data());
      db.push_back(col.dict_bytes.empty() ? (const void*)"" : (const void*)col.dict_bytes.data());
      d.length = (int64_t)col.dict_offsets.size() - 1;
      d.n_buffers = 3;
      d.buffers = db.data();
      d.release = release_pq_child;
      a.dictionary = &d;
    } else if (col.type.is_string) {
      b.push_back(col.offsets.data());
      b.push_back(col.bytes.empty() ? (const void*)"" : (const void*)col.bytes.data());
    } else {
      b.push_back(col.values.empty() ? (const void*)"" : (const void*)col.values.data());
    }
    a.length = rows;
    a.null_count = col.nulls;
    a.n_buffers = (int64_t)b.size();
    a.buffers = b.data();
    a.release = release_pq_child;
    if (col.packed_form) {
      a.release = release_pq_packed;
      a.private_data = &col.packed_view;
    }
    owner->child_ptrs[c] = &a;
  }
  std::memset(out, 0, sizeof(*out));
  out->length = rows;
  out->n_buffers = 1;
  out->buffers = owner->top_buffers;
  out->n_children = (int64_t)n;
  out->children = owner->child_ptrs.data();
  out->release = release_pq;
  out->private_data = owner.release();
}

void append_column(PqColumn& dst, PqColumn&& src) {
  if (dst.rows == 0 && dst.values.empty() && dst.offsets.empty()) {
    FieldType t = dst.type;
    dst = std::move(src);
    dst.type = t;
    return;
  }
  // validity: re-align the bits behind dst.rows
  dst.validity.resize((size_t)(dst.rows + src.rows + 7) / 8, 0);
  for (int64_t i = 0; i < src.rows; ++i)
    if (src.validity[(size_t)i >> 3] >> (i & 7) & 1) dst.validity[(size_t)(dst.rows + i) >> 3] |= (uint8_t)(1u << ((dst.rows + i) & 7));
  if (dst.type.is_string) {
    const int32_t base = dst.offsets.back();
    if ((uint64_t)base + src.bytes.size() > 0x7fffffffull) unsupported("more than 2 GiB of string data in one column; read the file per row group");
    for (size_t i = 1; i < src.offsets.size(); ++i) dst.offsets.push_back(base + src.offsets[i]);
    dst.bytes.insert(dst.bytes.end(), src.bytes.begin(), src.bytes.end());
  } else {
    dst.values.insert(dst.values.end(), src.values.begin(), src.values.end());
  }
  dst.nulls += src.nulls;
  dst.rows += src.rows;
}

}  // namespace

const PackedIndices* parquet_packed_indices(const ArrowArray* a) {
  return a && a->release == release_pq_packed ? (const PackedIndices*)a->private_data : nullptr;
}

void packed_indices_expand(const PackedIndices& pk, int64_t n, std::vector<int32_t>& out) {
  out.assign((size_t)n, 0);
  for (size_t r = 0; r < pk.n_runs; ++r) {
    const PackedRun& run = pk.runs[r];
    if ((uint64_t)run.first + run.count > (uint64_t)n) fail(PRISM_GPU_EINTERNAL, "packed index runs beyond their rows");
    int32_t* dst = out.data() + run.first;
    if (!(run.kind & 0x80000000u)) {
      std::fill(dst, dst + run.count, (int32_t)run.value);
      continue;
    }
    const uint32_t bw = run.kind & 63u;
    const uint64_t mask = (1ull << bw) - 1ull;
    const uint8_t* b = pk.bytes + run.value;
    uint64_t bitpos = 0;
    for (uint32_t i = 0; i < run.count; ++i, bitpos += bw) {
      uint64_t word;
      std::memcpy(&word, b + (bitpos >> 3), 8);  // (the buffer is padded by 8 bytes)
      dst[i] = (int32_t)((word >> (bitpos & 7)) & mask);
    }
  }
}

void parquet_read(const void* file_, size_t len, ArrowSchema* out_schema, ArrowArray* out) {
  const uint8_t* file = (const uint8_t*)file_;
  const FileMeta fm = read_footer(file, len);
  std::vector<FieldType> fields;
  for (const Leaf& l : fm.leaves) fields.push_back(leaf_field(l));
  std::vector<PqColumn> all(fields.size());
  for (size_t c = 0; c < fields.size(); ++c) {
    all[c].type = fields[c];
    if (fields[c].is_string) all[c].offsets.push_back(0);
  }
  int64_t rows = 0;
  for (const RowGroupMeta& rg : fm.groups) {
    if (rg.num_rows == 0) continue;
    for (size_t c = 0; c < fields.size(); ++c) {
      PqColumn col;
      col.type = fields[c];
      decode_chunk(file, len, fm.leaves[c], rg.chunks[c], rg.num_rows, col);
      if (rows == 0) { col.type = fields[c]; all[c] = std::move(col); }
      else append_column(all[c], std::move(col));
    }
    rows += rg.num_rows;
  }
  for (size_t c = 0; c < fields.size(); ++c) all[c].rows = rows;
  if (out_schema) export_schema(fields, out_schema);
  export_columns(std::move(all), rows, out);
}

void table_append_parquet(prism_gpu_table* table, uint32_t partition_id, const void* file_, size_t len) {
  const uint8_t* file = (const uint8_t*)file_;
  const FileMeta fm = read_footer(file, len);
  // The file's columns are matched to the table's by name, as ParquetExec's schema adapter
  // does; a table column the file lacks reads as NULLs.  Types must agree, except that Binary
  // and Utf8 stand in for each other, and Int64 for Timestamp(ms).
  std::vector<int> leaf_of(table->fields.size(), -1);
  for (size_t f = 0; f < table->fields.size(); ++f) {
    for (size_t l = 0; l < fm.leaves.size(); ++l)
      if (fm.leaves[l].name == table->fields[f].name) leaf_of[f] = (int)l;
    if (leaf_of[f] < 0) continue;
    const FieldType have = leaf_field(fm.leaves[leaf_of[f]]);
    const FieldType& want = table->fields[f];
    const bool same = have.format == want.format || (have.is_string && want.is_string) ||
                      (have.width == 8 && want.width == 8 && !have.is_float && !want.is_float && !have.is_string && !want.is_string);
    if (!same)
      unsupported("column `" + want.name + "` is " + have.format + " in the file and " + want.format + " in the table");
  }
  constexpr size_t kRowGroupsInFlight = 2;
  struct InFlight {
    struct Item {
      ArrowArray batch{};
      hipEvent_t copied = nullptr;
    };
    std::deque<Item> q;
    void settle_oldest() {
      Item& it = q.front();
      if (it.copied) {
        (void)hipEventSynchronize(it.copied);
        (void)hipEventDestroy(it.copied);
      }
      if (it.batch.release) it.batch.release(&it.batch);
      q.pop_front();
    }
    ~InFlight() {
      while (!q.empty()) settle_oldest();
    }
  } in_flight;
  for (const RowGroupMeta& rg : fm.groups) {
    if (rg.num_rows == 0) continue;
    std::vector<PqColumn> cols(table->fields.size());
    // The column chunks of a row group are independent byte ranges: they are decoded by a few
    // threads at once (option "parquet_threads"; callers that append several files from threads
    // of their own multiply that).  The first failure is kept and thrown once all have stopped.
    std::atomic<size_t> next{0};
    std::mutex err_mu;
    std::unique_ptr<Error> err;
    auto decode_columns = [&]() {
      // (the thread's page-locked blocks belong with the table's GPU: helper threads start on device 0)
      if (hipSetDevice(table->ctx->device) != hipSuccess) (void)hipGetLastError();
      tls_pin_values = true;
      struct Unpin { ~Unpin() { tls_pin_values = false; } } unpin;
      for (;;) {
        const size_t f = next.fetch_add(1);
        if (f >= table->fields.size()) return;
        try {
          PqColumn& col = cols[f];
          col.type = table->fields[f];
          if (leaf_of[f] >= 0) {
            // (string chunks stay dictionary encoded where the file has them so: table_append maps
            // a dictionary's values to the table's codes once, and the rows are expanded on the device)
            decode_chunk(file, len, fm.leaves[leaf_of[f]], rg.chunks[leaf_of[f]], rg.num_rows, col, /*keep_dict=*/true,
                         /*keep_packed=*/table->ctx->dict_expand_device.load(std::memory_order_relaxed) != 0);
          } else {
            col.rows = rg.num_rows;
            col.nulls = rg.num_rows;
            col.validity.assign((size_t)(rg.num_rows + 7) / 8, 0);
            if (col.type.is_string) col.offsets.assign((size_t)rg.num_rows + 1, 0);
            else col.values.assign((size_t)rg.num_rows * col.type.width, 0);
          }
        } catch (const Error& e) {
          std::lock_guard<std::mutex> g(err_mu);
          if (!err) err.reset(new Error(e));
        } catch (const std::bad_alloc&) {
          std::lock_guard<std::mutex> g(err_mu);
          if (!err) err.reset(new Error{PRISM_GPU_EOOM, "host memory exhausted while decoding a Parquet column chunk"});
        } catch (const std::exception& e) {
          std::lock_guard<std::mutex> g(err_mu);
          if (!err) err.reset(new Error{PRISM_GPU_EINTERNAL, std::string("internal error: ") + e.what()});
        }
      }
    };
    const size_t n_threads = std::min<size_t>(table->fields.size(), (size_t)std::max<int>(1, table->ctx->parquet_threads.load(std::memory_order_relaxed)));
    std::vector<std::thread> helpers;
    for (size_t i = 1; i < n_threads; ++i) helpers.emplace_back(decode_columns);
    decode_columns();
    for (std::thread& t : helpers) t.join();
    if (err) throw *err;
    // The row group's page-locked blocks are read by the copy engine after the append has
    // returned: the batch is kept until the event behind those copies has passed, and the next
    // row group is decoded meanwhile (waiting here left the thread idle behind every other
    // caller's copies, the engine's queue being one).
    in_flight.q.emplace_back();
    export_columns(std::move(cols), rg.num_rows, &in_flight.q.back().batch);
    table_append(table, partition_id, &in_flight.q.back().batch, false, &in_flight.q.back().copied);
    while (in_flight.q.size() > kRowGroupsInFlight) in_flight.settle_oldest();
  }
}

}  // namespace prism
