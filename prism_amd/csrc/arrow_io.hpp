// Arrow C Data Interface plumbing: parsing input schemas/arrays and building the
// struct-typed arrays the library returns.  Host only.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"

namespace prism {

// Logical column types beyond prism_gpu_col_type that appear only in PARTIAL
// state batches consumed by prism_gpu_merge.
constexpr int kTypeUInt64 = 100;
constexpr int kTypeFloat64 = 101;
// List<T>: the state of a COUNT(DISTINCT) that stands next to other aggregates (the distinct
// values a partition saw per group).  Only result batches and prism_gpu_merge's input carry it.
constexpr int kTypeList = 102;

struct FieldType {
  int logical = 0;      // prism_gpu_col_type or kTypeUInt64 / kTypeFloat64
  int width = 0;        // bytes per value on the device (strings: 0)
  bool is_unsigned = false;
  bool is_string = false;
  bool is_float = false;
  bool nullable = true;
  std::string format;   // Arrow format string of the value type
  std::string name;
  std::string list_item;  // kTypeList: the format of the list's values (format is "+l")
};
FieldType list_field(const std::string& name, const std::string& item_format);

FieldType field_from_format(const char* format);
// Parses a struct-typed ArrowSchema into fields (dictionary-encoded Utf8 fields
// are reported as Utf8).
std::vector<FieldType> parse_schema(const ArrowSchema* schema);

// ---- export --------------------------------------------------------------

// One output column under construction (host memory).
struct OutColumn {
  FieldType type;
  std::vector<int64_t> i64;        // integer / timestamp values (any width)
  std::vector<double> f64;
  std::vector<std::string> str;
  // dictionary-coded strings: when `dict` is set the values are (*dict)[codes[i]] and
  // `str` is unused (large GROUP BY results never materialise one std::string per row)
  const std::vector<std::string>* dict = nullptr;
  std::vector<uint32_t> codes;
  std::vector<uint8_t> valid;      // one byte per row
  // kTypeList: row i holds items [list_off[i], list_off[i + 1]) of `items` (list_off starts with one 0)
  std::shared_ptr<OutColumn> items;
  std::vector<int32_t> list_off;
  // Borrowed rows: `ext_rows` 8-byte values (integers, doubles, or dictionary codes, by
  // `type` and `dict`) and validity bytes in a block that `ext_owner` keeps alive (a pinned
  // landing block the device wrote whole result columns into).  A column is either
  // borrowed or held in the vectors above, never both; materialise() copies it over, and
  // everything that edits rows does that first.  Export hands 8-byte borrowed values to
  // Arrow as they are.
  const int64_t* ext_vals = nullptr;
  const uint8_t* ext_valid = nullptr;
  size_t ext_rows = 0;
  std::shared_ptr<void> ext_owner;
  bool borrowed() const { return ext_vals != nullptr; }
  void borrow(std::shared_ptr<void> owner, const int64_t* vals, const uint8_t* valid_bytes, size_t n);
  void materialise();
  size_t rows() const { return borrowed() ? ext_rows : valid.size(); }
  const uint8_t* valid_data() const { return borrowed() ? ext_valid : valid.data(); }
  const int64_t* i64_data() const { return borrowed() ? ext_vals : i64.data(); }
  const double* f64_data() const { return borrowed() ? (const double*)ext_vals : f64.data(); }
  uint32_t code_at(size_t i) const { return borrowed() ? (uint32_t)ext_vals[i] : codes[i]; }
  void push_null();
  void push_int(int64_t v);
  void push_float(double v);
  void push_str(const std::string& s);
  const std::string& str_at(size_t i) const { return dict ? (*dict)[code_at(i)] : str[i]; }
  // Keeps rows idx[0], idx[1], ... in that order.
  void gather(const std::vector<size_t>& idx);
};

OutColumn make_out_column(const std::string& name, const std::string& format, bool nullable);

// Moves the columns into a freshly allocated struct ArrowArray (caller-owned).
void export_record_batch(std::vector<OutColumn>& cols, ArrowArray* out);
void export_schema(const std::vector<FieldType>& fields, ArrowSchema* out);

}  // namespace prism
