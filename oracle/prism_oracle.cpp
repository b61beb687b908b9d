// CPU oracle for the scan -> filter -> aggregate path.  TEST INFRASTRUCTURE ONLY.
//
// This file restates, in plain scalar C++, the behaviour of the DataFusion 32.0.0
// / arrow 47.0.0 operators that prism-query plans for a query
// (ParquetExec -> FilterExec -> AggregateExec(Partial) -> AggregateExec(Final)
// -> SortExec(fetch)).  prism itself only wires those operators together
// (src/prism-query/src/main.rs:82-90,
//  src/prism-query/src/providers/table_provider.rs:110-131,
//  src/prism-ingest/src/ingest.rs:67-77 for the ungrouped MIN/MAX aggregate,
//  src/prism-ql/src/lower.rs:83-95 for COUNT(*) [BY col]); the arithmetic lives
// in the crates pinned by Cargo.lock:156-355 (arrow 47.0.0) and Cargo.lock:886-1073
// (datafusion 32.0.0), which are not present offline.  What is restated here is
// their published behaviour:
//   * FilterExec keeps a row only when the predicate is TRUE (NULL and FALSE are
//     both dropped); AND/OR/NOT follow Kleene three-valued logic; comparisons and
//     LIKE return NULL for a NULL operand;
//   * LIKE is case sensitive and byte-wise, '%' matches any run, '_' one
//     character, '\' escapes;
//   * COUNT(*) counts rows, COUNT(x) counts non-NULL x, both Int64;
//   * SUM(Int16/32/64) is Int64 and SUM(UInt16) is UInt64, with wrapping
//     overflow; SUM/MIN/MAX/AVG over no non-NULL value is NULL;
//   * AVG casts every value to Float64, sums in Float64 and divides by the count;
//   * MIN/MAX keep the input type;
//   * NULL is a group key like any other value;
//   * date_bin(stride, ts, origin) = origin + d - d % stride with d = ts - origin
//     (truncating %), minus one more stride when d < 0 and stride > 1;
//   * COUNT(DISTINCT x) ignores NULLs;
//   * ORDER BY puts NULLs last when ascending and first when descending; ties
//     are in arbitrary order.
//
// PINNING.  The only result of this path that the reference's own tests fix is
// ungrouped MIN/MAX over Timestamp(ms) on src/prism-ingest/src/testdata/demo.log
// (src/prism-ingest/src/ingest.rs:224-226); tests/test_oracle_golden.py checks
// that vector.  Everything else (filters, COUNT/SUM/AVG, hash GROUP BY, date_bin,
// LIKE, COUNT DISTINCT, ORDER BY/LIMIT, partial/final merge) is PARITY UNPINNED
// by the reference; it is cross-checked against fixtures computed independently
// with pyarrow and numpy (tests/golden/, made by scripts/make_golden.py).
//
// Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may call into this file.  It is a checker, never the product path.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../include/prism_gpu.h"
#include "../include/prism_gpu_synth.h"
#include "../include/prism_synth.h"

namespace {

thread_local std::string g_err;

struct OracleError {
  int code;
  std::string msg;
};

[[noreturn]] void fail(int code, const std::string& m) { throw OracleError{code, m}; }

// ---------------------------------------------------------------- columns ---

enum class Phys { I16, U16, I32, I64, U64, F64, STR };

struct ColType {
  int logical = 0;  // prism_gpu_col_type, or 100 = UInt64, 101 = Float64 (state columns)
  Phys phys = Phys::I64;
  std::string format;
};

ColType type_from_format(const char* f) {
  std::string s(f);
  ColType t;
  t.format = s;
  if (s == "l") { t.logical = PRISM_COL_INT64; t.phys = Phys::I64; }
  else if (s == "i") { t.logical = PRISM_COL_INT32; t.phys = Phys::I32; }
  else if (s == "s") { t.logical = PRISM_COL_INT16; t.phys = Phys::I16; }
  else if (s == "S") { t.logical = PRISM_COL_UINT16; t.phys = Phys::U16; }
  else if (s == "L") { t.logical = 100; t.phys = Phys::U64; }
  else if (s == "g") { t.logical = 101; t.phys = Phys::F64; }
  else if (s == "u") { t.logical = PRISM_COL_UTF8; t.phys = Phys::STR; }
  else if (s == "z") { t.logical = PRISM_COL_BINARY; t.phys = Phys::STR; }
  else if (s.rfind("tsm:", 0) == 0) { t.logical = PRISM_COL_TIMESTAMP; t.phys = Phys::I64; }
  else fail(PRISM_GPU_EUNSUPPORTED, "oracle: unsupported Arrow format '" + s + "'");
  return t;
}

// A read-only view of one column of one batch.
struct ColView {
  ColType type;
  int64_t n = 0;
  int64_t offset = 0;
  const uint8_t* validity = nullptr;
  const void* values = nullptr;     // fixed width values, or i32 offsets for STR
  const uint8_t* data = nullptr;    // STR bytes
  // dictionary-encoded Utf8: values = i32 indices, dict_* describe the dictionary
  bool dict = false;
  const int32_t* dict_offsets = nullptr;
  const uint8_t* dict_data = nullptr;
  int64_t dict_offset = 0;

  bool is_null(int64_t i) const {
    if (!validity) return false;
    int64_t j = i + offset;
    return !((validity[j >> 3] >> (j & 7)) & 1);
  }
  int64_t i64(int64_t i) const {
    int64_t j = i + offset;
    switch (type.phys) {
      case Phys::I16: return ((const int16_t*)values)[j];
      case Phys::U16: return ((const uint16_t*)values)[j];
      case Phys::I32: return ((const int32_t*)values)[j];
      case Phys::I64: return ((const int64_t*)values)[j];
      case Phys::U64: return (int64_t)((const uint64_t*)values)[j];
      default: return 0;
    }
  }
  double f64(int64_t i) const { return ((const double*)values)[i + offset]; }
  std::string_view str(int64_t i) const {
    int64_t j = i + offset;
    if (dict) {
      int64_t k = ((const int32_t*)values)[j] + dict_offset;
      return std::string_view((const char*)dict_data + dict_offsets[k],
                              (size_t)(dict_offsets[k + 1] - dict_offsets[k]));
    }
    const int32_t* o = (const int32_t*)values;
    return std::string_view((const char*)data + o[j], (size_t)(o[j + 1] - o[j]));
  }
};

struct Batch {
  int64_t n = 0;
  std::vector<ColView> cols;
  // owned storage for synthetic batches
  std::vector<std::vector<uint8_t>> owned;
};

ColView view_of(const ArrowSchema* fs, const ArrowArray* fa) {
  ColView v;
  v.n = fa->length;
  v.offset = fa->offset;
  if (fs->dictionary) {
    ColType vt = type_from_format(fs->dictionary->format);
    if (vt.phys != Phys::STR) fail(PRISM_GPU_EUNSUPPORTED, "oracle: dictionary of non-string");
    if (std::string(fs->format) != "i")
      fail(PRISM_GPU_EUNSUPPORTED, "oracle: dictionary index type must be int32");
    v.type = vt;
    v.dict = true;
    v.validity = (const uint8_t*)fa->buffers[0];
    v.values = fa->buffers[1];
    const ArrowArray* d = fa->dictionary;
    v.dict_offsets = (const int32_t*)d->buffers[1];
    v.dict_data = (const uint8_t*)d->buffers[2];
    v.dict_offset = d->offset;
    return v;
  }
  v.type = type_from_format(fs->format);
  v.validity = (fa->null_count == 0) ? nullptr : (const uint8_t*)fa->buffers[0];
  v.values = fa->buffers[1];
  if (v.type.phys == Phys::STR) v.data = (const uint8_t*)fa->buffers[2];
  return v;
}

Batch batch_of(const ArrowSchema* schema, const ArrowArray* arr) {
  if (std::string(schema->format) != "+s") fail(PRISM_GPU_EINVAL, "oracle: schema is not a struct");
  if (arr->n_children != schema->n_children) fail(PRISM_GPU_EINVAL, "oracle: batch/schema mismatch");
  if (arr->offset != 0) fail(PRISM_GPU_EUNSUPPORTED, "oracle: struct offset");
  Batch b;
  b.n = arr->length;
  for (int64_t c = 0; c < schema->n_children; ++c)
    b.cols.push_back(view_of(schema->children[c], arr->children[c]));
  return b;
}

// ------------------------------------------------------------------- LIKE ---

inline size_t utf8_len_at(const unsigned char* s, size_t i, size_t n) {
  unsigned char c = s[i];
  size_t l = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1;
  return std::min(l, n - i);
}

struct LikeTok {
  enum Kind { LIT, ANY_ONE, ANY_RUN } kind;
  unsigned char c;
};

std::vector<LikeTok> like_compile(std::string_view pat) {
  std::vector<LikeTok> t;
  for (size_t i = 0; i < pat.size(); ++i) {
    unsigned char c = (unsigned char)pat[i];
    if (c == '\\' && i + 1 < pat.size()) { t.push_back({LikeTok::LIT, (unsigned char)pat[++i]}); }
    else if (c == '%') { if (t.empty() || t.back().kind != LikeTok::ANY_RUN) t.push_back({LikeTok::ANY_RUN, 0}); }
    else if (c == '_') t.push_back({LikeTok::ANY_ONE, 0});
    else t.push_back({LikeTok::LIT, c});
  }
  return t;
}

inline unsigned char fold(unsigned char c, bool ci) {
  return (ci && c >= 'A' && c <= 'Z') ? (unsigned char)(c + 32) : c;
}

// Iterative wildcard match with single backtrack point (classic glob algorithm).
bool like_match(const std::vector<LikeTok>& t, std::string_view sv, bool ci) {
  const unsigned char* s = (const unsigned char*)sv.data();
  size_t n = sv.size(), si = 0, ti = 0, star_t = (size_t)-1, star_s = 0;
  while (si < n) {
    if (ti < t.size() && t[ti].kind == LikeTok::LIT && fold(t[ti].c, ci) == fold(s[si], ci)) { ++si; ++ti; }
    else if (ti < t.size() && t[ti].kind == LikeTok::ANY_ONE) { si += utf8_len_at(s, si, n); ++ti; }
    else if (ti < t.size() && t[ti].kind == LikeTok::ANY_RUN) { star_t = ti++; star_s = si; }
    else if (star_t != (size_t)-1) { ti = star_t + 1; star_s += utf8_len_at(s, star_s, n); si = star_s; }
    else return false;
  }
  while (ti < t.size() && t[ti].kind == LikeTok::ANY_RUN) ++ti;
  return ti == t.size();
}

// ------------------------------------------------------------- predicates ---

struct Pred {
  const prism_gpu_plan_desc* d;
  std::vector<std::vector<LikeTok>> like;  // per expr index
};

// returns 1 TRUE, 0 FALSE, -1 NULL
int eval(const Pred& P, int e, const Batch& b, int64_t row) {
  const prism_gpu_expr& x = P.d->exprs[e];
  switch (x.kind) {
    case PRISM_EXPR_CMP: {
      const prism_gpu_expr& l = P.d->exprs[x.lhs];
      const prism_gpu_expr& r = P.d->exprs[x.rhs];
      const ColView& c = b.cols[l.column];
      if (c.is_null(row)) return -1;
      int cmp;
      if (c.type.phys == Phys::STR) {
        std::string_view a = c.str(row), lit(r.str, (size_t)r.str_len);
        int k = std::memcmp(a.data(), lit.data(), std::min(a.size(), lit.size()));
        cmp = k != 0 ? (k < 0 ? -1 : 1) : (a.size() < lit.size() ? -1 : a.size() > lit.size() ? 1 : 0);
      } else {
        int64_t a = c.i64(row);
        cmp = a < r.i64 ? -1 : a > r.i64 ? 1 : 0;
      }
      switch (x.op) {
        case PRISM_CMP_EQ: return cmp == 0;
        case PRISM_CMP_NEQ: return cmp != 0;
        case PRISM_CMP_LT: return cmp < 0;
        case PRISM_CMP_LTEQ: return cmp <= 0;
        case PRISM_CMP_GT: return cmp > 0;
        case PRISM_CMP_GTEQ: return cmp >= 0;
      }
      fail(PRISM_GPU_EINVAL, "oracle: bad cmp op");
    }
    case PRISM_EXPR_AND: {
      int a = eval(P, x.lhs, b, row), c = eval(P, x.rhs, b, row);
      if (a == 0 || c == 0) return 0;
      if (a == -1 || c == -1) return -1;
      return 1;
    }
    case PRISM_EXPR_OR: {
      int a = eval(P, x.lhs, b, row), c = eval(P, x.rhs, b, row);
      if (a == 1 || c == 1) return 1;
      if (a == -1 || c == -1) return -1;
      return 0;
    }
    case PRISM_EXPR_NOT: {
      int a = eval(P, x.lhs, b, row);
      return a == -1 ? -1 : !a;
    }
    case PRISM_EXPR_LIKE: {
      const ColView& c = b.cols[P.d->exprs[x.lhs].column];
      if (c.is_null(row)) return -1;
      bool m = like_match(P.like[e], c.str(row), (x.flags & PRISM_LIKE_CASE_INSENSITIVE) != 0);
      return (x.flags & PRISM_LIKE_NEGATED) ? !m : m;
    }
    case PRISM_EXPR_IS_NULL: return b.cols[P.d->exprs[x.lhs].column].is_null(row) ? 1 : 0;
    case PRISM_EXPR_IS_NOT_NULL: return b.cols[P.d->exprs[x.lhs].column].is_null(row) ? 0 : 1;
    default: fail(PRISM_GPU_EINVAL, "oracle: expression kind cannot be a predicate");
  }
}

void validate_expr(const prism_gpu_plan_desc* d, int e, const std::vector<ColType>& in) {
  if (e < 0 || (uint32_t)e >= d->n_exprs) fail(PRISM_GPU_EINVAL, "oracle: expr index out of range");
  const prism_gpu_expr& x = d->exprs[e];
  auto col_of = [&](int idx) -> const ColType& {
    if (idx < 0 || (uint32_t)idx >= d->n_exprs || d->exprs[idx].kind != PRISM_EXPR_COLUMN)
      fail(PRISM_GPU_EUNSUPPORTED, "oracle: operand must be a column");
    int c = d->exprs[idx].column;
    if (c < 0 || (size_t)c >= in.size()) fail(PRISM_GPU_EINVAL, "oracle: column out of range");
    return in[c];
  };
  switch (x.kind) {
    case PRISM_EXPR_CMP: {
      const ColType& t = col_of(x.lhs);
      if (x.rhs < 0 || (uint32_t)x.rhs >= d->n_exprs) fail(PRISM_GPU_EINVAL, "oracle: bad rhs");
      int rk = d->exprs[x.rhs].kind;
      if (t.phys == Phys::STR ? rk != PRISM_EXPR_LIT_STR : rk != PRISM_EXPR_LIT_I64)
        fail(PRISM_GPU_EUNSUPPORTED, "oracle: literal type does not match column");
      if (t.phys == Phys::F64) fail(PRISM_GPU_EUNSUPPORTED, "oracle: float compare");
      break;
    }
    case PRISM_EXPR_AND: case PRISM_EXPR_OR:
      validate_expr(d, x.lhs, in); validate_expr(d, x.rhs, in); break;
    case PRISM_EXPR_NOT: validate_expr(d, x.lhs, in); break;
    case PRISM_EXPR_LIKE: {
      const ColType& t = col_of(x.lhs);
      if (t.logical != PRISM_COL_UTF8) fail(PRISM_GPU_EUNSUPPORTED, "oracle: LIKE needs a Utf8 column");
      if (x.rhs < 0 || d->exprs[x.rhs].kind != PRISM_EXPR_LIT_STR) fail(PRISM_GPU_EINVAL, "oracle: LIKE pattern");
      break;
    }
    case PRISM_EXPR_IS_NULL: case PRISM_EXPR_IS_NOT_NULL: col_of(x.lhs); break;
    default: fail(PRISM_GPU_EINVAL, "oracle: not a predicate");
  }
}

// ---------------------------------------------------------------- grouping ---

constexpr int MAXK = 4;

struct GKey {
  int64_t v[MAXK];
  uint32_t nulls;
  bool operator==(const GKey& o) const {
    return nulls == o.nulls && v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3];
  }
};
struct GKeyHash {
  size_t operator()(const GKey& k) const {
    uint64_t h = prism_mix64(k.v[0] ^ k.nulls);
    h = prism_mix64(h ^ k.v[1]);
    h = prism_mix64(h ^ k.v[2]);
    h = prism_mix64(h ^ k.v[3]);
    return (size_t)h;
  }
};

// Strings used as group keys or distinct values are interned: id <-> bytes.
struct Interner {
  std::unordered_map<std::string, int64_t> ids;
  std::vector<std::string> strs;
  int64_t id(std::string_view s) {
    auto it = ids.find(std::string(s));
    if (it != ids.end()) return it->second;
    int64_t i = (int64_t)strs.size();
    strs.emplace_back(s);
    ids.emplace(strs.back(), i);
    return i;
  }
};

struct AggState {
  int64_t count = 0;   // COUNT / non-null count
  uint64_t sum = 0;    // wrapping integer sum
  double fsum = 0.0;   // AVG sum
  int64_t mn = 0, mx = 0;
  std::unordered_set<int64_t>* distinct = nullptr;
};

int64_t date_bin_ms(int64_t ts, int64_t stride, int64_t origin) {
  // date_bin_nanos_interval + compute_distance of datafusion-physical-expr 32,
  // applied in milliseconds (the ms -> ns scaling and back cancel for strides
  // that are whole milliseconds).
  int64_t d = (int64_t)((uint64_t)ts - (uint64_t)origin);
  int64_t delta = d - d % stride;
  if (d < 0 && stride > 1) delta -= stride;
  return (int64_t)((uint64_t)origin + (uint64_t)delta);
}

struct Plan {
  const prism_gpu_plan_desc* d;
  std::vector<ColType> in;         // input schema types
  std::vector<std::string> in_names;
  Pred pred;
  bool merge = false;              // input is PARTIAL state of this plan
  bool distinct_plan = false;      // every aggregate is COUNT(DISTINCT same column)
  int distinct_col = -1;
};

// Key -> group number: a flat open-addressing table (linear probing, at most half full) of
// (upper hash bits, group number + 1) words, the shape of the hashbrown RawTable<(hash,
// group_idx)> behind DataFusion 32's GroupValues (a node-based map made the CPU figures of
// high-cardinality GROUP BYs look worse than the reference's design would be).
struct GroupIndex {
  std::vector<uint64_t> slot;  // 0 = free
  size_t used = 0;
  void grow(const std::vector<GKey>& keys) {
    std::vector<uint64_t> next(slot.empty() ? 1024 : slot.size() * 2, 0);
    const size_t mask = next.size() - 1;
    for (uint64_t w : slot) {
      if (!w) continue;
      size_t i = (size_t)GKeyHash()(keys[(uint32_t)w - 1]) & mask;
      while (next[i]) i = (i + 1) & mask;
      next[i] = w;
    }
    slot.swap(next);
  }
  // the group of `k`, or ~0u after reserving room for it to be added as group keys.size()
  uint32_t find_or_add(const GKey& k, std::vector<GKey>& keys) {
    if ((used + 1) * 2 > slot.size()) grow(keys);
    const uint64_t h = GKeyHash()(k), tag = h & 0xFFFFFFFF00000000ull;
    const size_t mask = slot.size() - 1;
    size_t i = (size_t)h & mask;
    for (; slot[i]; i = (i + 1) & mask) {
      if ((slot[i] & 0xFFFFFFFF00000000ull) != tag) continue;
      const uint32_t g = (uint32_t)slot[i] - 1;
      if (keys[g] == k) return g;
    }
    slot[i] = tag | (uint64_t)(keys.size() + 1);
    ++used;
    return ~0u;
  }
};

struct GroupTable {
  GroupIndex index;
  std::vector<GKey> keys;
  std::vector<AggState> st;        // n_groups * n_aggs
  std::vector<Interner> key_strs;  // per key column
  Interner distinct_strs;
  uint32_t n_aggs = 0;
  int64_t rows_in = 0, rows_selected = 0;

  ~GroupTable() { for (auto& s : st) delete s.distinct; }
  GroupTable() = default;
  GroupTable(const GroupTable&) = delete;
  GroupTable& operator=(const GroupTable&) = delete;

  uint32_t group_of(const GKey& k) {
    uint32_t g = index.find_or_add(k, keys);
    if (g != ~0u) return g;
    g = (uint32_t)keys.size();
    keys.push_back(k);
    st.resize(st.size() + n_aggs);
    return g;
  }
};

bool is_unsigned_sum(const ColType& t) { return t.phys == Phys::U16 || t.phys == Phys::U64; }

// Layout of PARTIAL state columns after the key columns.
struct StateLayout {
  std::vector<int> first_col;  // per aggregate: index of its first state column
  int n_cols = 0;
};

StateLayout state_layout(const Plan& p) {
  StateLayout L;
  int c = (int)p.d->n_keys;
  if (p.distinct_plan) {  // keys..., distinct value
    L.first_col.assign(p.d->n_aggs, c);
    L.n_cols = c + 1;
    return L;
  }
  for (uint32_t a = 0; a < p.d->n_aggs; ++a) {
    L.first_col.push_back(c);
    c += (p.d->aggs[a].func == PRISM_AGG_AVG) ? 2 : 1;
  }
  L.n_cols = c;
  return L;
}

void update_raw(const Plan& p, GroupTable& T, const Batch& b) {
  const prism_gpu_plan_desc* d = p.d;
  T.rows_in += b.n;
  for (int64_t r = 0; r < b.n; ++r) {
    if (d->predicate >= 0 && eval(p.pred, d->predicate, b, r) != 1) continue;
    ++T.rows_selected;
    GKey k{};
    for (uint32_t i = 0; i < d->n_keys; ++i) {
      const ColView& c = b.cols[d->keys[i].column];
      if (c.is_null(r)) { k.nulls |= 1u << i; continue; }
      if (c.type.phys == Phys::STR) k.v[i] = T.key_strs[i].id(c.str(r));
      else if (d->keys[i].kind == PRISM_KEY_DATE_BIN)
        k.v[i] = date_bin_ms(c.i64(r), d->keys[i].stride_ms, d->keys[i].origin_ms);
      else k.v[i] = c.i64(r);
    }
    uint32_t g = T.group_of(k);
    AggState* S = &T.st[(size_t)g * T.n_aggs];
    for (uint32_t a = 0; a < d->n_aggs; ++a) {
      const prism_gpu_agg& ag = d->aggs[a];
      if (ag.func == PRISM_AGG_COUNT_STAR) { ++S[a].count; continue; }
      const ColView& c = b.cols[ag.column];
      if (c.is_null(r)) continue;
      switch (ag.func) {
        case PRISM_AGG_COUNT: ++S[a].count; break;
        case PRISM_AGG_SUM: S[a].sum += (uint64_t)c.i64(r); ++S[a].count; break;
        case PRISM_AGG_AVG: S[a].fsum += (double)c.i64(r); ++S[a].count; break;
        case PRISM_AGG_MIN: { int64_t v = c.i64(r); if (!S[a].count || v < S[a].mn) S[a].mn = v; ++S[a].count; break; }
        case PRISM_AGG_MAX: { int64_t v = c.i64(r); if (!S[a].count || v > S[a].mx) S[a].mx = v; ++S[a].count; break; }
        case PRISM_AGG_COUNT_DISTINCT: {
          if (!S[a].distinct) S[a].distinct = new std::unordered_set<int64_t>();
          S[a].distinct->insert(c.type.phys == Phys::STR ? T.distinct_strs.id(c.str(r)) : c.i64(r));
          break;
        }
      }
    }
  }
}

// Merge one PARTIAL-state batch (keys..., state columns...) into T.
void update_merge(const Plan& p, GroupTable& T, const Batch& b) {
  const prism_gpu_plan_desc* d = p.d;
  StateLayout L = state_layout(p);
  if ((int)b.cols.size() != L.n_cols) fail(PRISM_GPU_EINVAL, "oracle: partial state has wrong column count");
  for (int64_t r = 0; r < b.n; ++r) {
    GKey k{};
    for (uint32_t i = 0; i < d->n_keys; ++i) {
      const ColView& c = b.cols[i];
      if (c.is_null(r)) { k.nulls |= 1u << i; continue; }
      k.v[i] = c.type.phys == Phys::STR ? T.key_strs[i].id(c.str(r)) : c.i64(r);
    }
    uint32_t g = T.group_of(k);
    AggState* S = &T.st[(size_t)g * T.n_aggs];
    for (uint32_t a = 0; a < d->n_aggs; ++a) {
      const ColView& c = b.cols[L.first_col[a]];
      switch (d->aggs[a].func) {
        case PRISM_AGG_COUNT_STAR: case PRISM_AGG_COUNT: S[a].count += c.i64(r); break;
        case PRISM_AGG_SUM: if (!c.is_null(r)) { S[a].sum += (uint64_t)c.i64(r); ++S[a].count; } break;
        case PRISM_AGG_MIN: if (!c.is_null(r)) { int64_t v = c.i64(r); if (!S[a].count || v < S[a].mn) S[a].mn = v; ++S[a].count; } break;
        case PRISM_AGG_MAX: if (!c.is_null(r)) { int64_t v = c.i64(r); if (!S[a].count || v > S[a].mx) S[a].mx = v; ++S[a].count; } break;
        case PRISM_AGG_AVG: {
          const ColView& s = b.cols[L.first_col[a] + 1];
          S[a].count += c.i64(r);
          if (!s.is_null(r)) S[a].fsum += s.f64(r);
          break;
        }
        case PRISM_AGG_COUNT_DISTINCT: {
          if (c.is_null(r)) break;
          if (!S[a].distinct) S[a].distinct = new std::unordered_set<int64_t>();
          S[a].distinct->insert(c.type.phys == Phys::STR ? T.distinct_strs.id(c.str(r)) : c.i64(r));
          break;
        }
      }
    }
  }
}

// Fold table B into A (used by the multi-threaded path).
// Merges B's groups into A; with n_slices > 1 only the groups whose key hash falls into
// `slice` (the share of one FinalPartitioned task behind a hash repartition).
void fold(const Plan& p, GroupTable& A, GroupTable& B, size_t n_slices = 1, size_t slice = 0) {
  const prism_gpu_plan_desc* d = p.d;
  if (slice == 0) {
    A.rows_in += B.rows_in;
    A.rows_selected += B.rows_selected;
  }
  for (size_t g = 0; g < B.keys.size(); ++g) {
    if (n_slices > 1 && (GKeyHash()(B.keys[g]) >> 40) % n_slices != slice) continue;
    GKey k = B.keys[g];
    for (uint32_t i = 0; i < d->n_keys; ++i)
      if (!B.key_strs[i].strs.empty() && !(k.nulls & (1u << i)))
        k.v[i] = A.key_strs[i].id(B.key_strs[i].strs[(size_t)k.v[i]]);
    uint32_t ga = A.group_of(k);
    AggState* SA = &A.st[(size_t)ga * A.n_aggs];
    AggState* SB = &B.st[g * B.n_aggs];
    for (uint32_t a = 0; a < d->n_aggs; ++a) {
      switch (d->aggs[a].func) {
        case PRISM_AGG_MIN: if (SB[a].count) { if (!SA[a].count || SB[a].mn < SA[a].mn) SA[a].mn = SB[a].mn; } break;
        case PRISM_AGG_MAX: if (SB[a].count) { if (!SA[a].count || SB[a].mx > SA[a].mx) SA[a].mx = SB[a].mx; } break;
        case PRISM_AGG_COUNT_DISTINCT:
          if (SB[a].distinct) {
            if (!SA[a].distinct) SA[a].distinct = new std::unordered_set<int64_t>();
            bool strs = !B.distinct_strs.strs.empty();
            for (int64_t v : *SB[a].distinct)
              SA[a].distinct->insert(strs ? A.distinct_strs.id(B.distinct_strs.strs[(size_t)v]) : v);
          }
          break;
        default: break;
      }
      SA[a].count += SB[a].count;
      SA[a].sum += SB[a].sum;
      SA[a].fsum += SB[a].fsum;
    }
  }
}

// --------------------------------------------------------- Arrow building ---

struct OutCol {
  std::string name, format;
  bool nullable = true;
  bool is_str = false, is_f64 = false;
  int width = 8;
  std::vector<int64_t> iv;
  std::vector<double> fv;
  std::vector<std::string> sv;
  std::vector<uint8_t> valid;  // one byte per row
  void push_null() { valid.push_back(0); if (is_str) sv.emplace_back(); else if (is_f64) fv.push_back(0); else iv.push_back(0); }
  void push_i(int64_t v) { valid.push_back(1); iv.push_back(v); }
  void push_f(double v) { valid.push_back(1); fv.push_back(v); }
  void push_s(std::string_view v) { valid.push_back(1); sv.emplace_back(v); }
  size_t rows() const { return valid.size(); }
};

OutCol make_col(const std::string& name, const std::string& format, bool nullable) {
  OutCol c;
  c.name = name; c.format = format; c.nullable = nullable;
  ColType t = type_from_format(format.c_str());
  c.is_str = t.phys == Phys::STR;
  c.is_f64 = t.phys == Phys::F64;
  c.width = (t.phys == Phys::I16 || t.phys == Phys::U16) ? 2 : t.phys == Phys::I32 ? 4 : 8;
  return c;
}

struct Holder {  // owns everything an exported ArrowArray/ArrowSchema points to
  std::vector<std::vector<uint8_t>> bufs;
  std::vector<const void*> ptrs[64];
  std::vector<ArrowArray> kids;
  std::vector<ArrowArray*> kid_ptrs;
  const void* top_bufs[1] = {nullptr};
};

void release_array(ArrowArray* a) {
  delete (Holder*)a->private_data;
  a->release = nullptr;
}
void release_child(ArrowArray* a) { a->release = nullptr; }

void export_cols(std::vector<OutCol>& cols, ArrowArray* out) {
  if (cols.size() > 64) fail(PRISM_GPU_EINTERNAL, "oracle: too many output columns");
  auto* h = new Holder();
  size_t n = cols.empty() ? 0 : cols[0].rows();
  h->kids.resize(cols.size());
  for (size_t c = 0; c < cols.size(); ++c) {
    OutCol& oc = cols[c];
    ArrowArray& k = h->kids[c];
    std::memset(&k, 0, sizeof(k));
    int64_t nulls = 0;
    std::vector<uint8_t> bits((n + 7) / 8 + 8, 0);
    for (size_t i = 0; i < n; ++i) { if (oc.valid[i]) bits[i >> 3] |= (uint8_t)(1u << (i & 7)); else ++nulls; }
    h->bufs.push_back(std::move(bits));
    const void* vbits = nulls ? h->bufs.back().data() : nullptr;
    h->ptrs[c].push_back(vbits);
    if (oc.is_str) {
      std::vector<uint8_t> off((n + 1) * 4 + 8), data;
      int32_t* o = (int32_t*)off.data();
      o[0] = 0;
      for (size_t i = 0; i < n; ++i) { data.insert(data.end(), oc.sv[i].begin(), oc.sv[i].end()); o[i + 1] = (int32_t)data.size(); }
      data.resize(data.size() + 8);
      h->bufs.push_back(std::move(off)); h->ptrs[c].push_back(h->bufs.back().data());
      h->bufs.push_back(std::move(data)); h->ptrs[c].push_back(h->bufs.back().data());
      k.n_buffers = 3;
    } else {
      std::vector<uint8_t> vals(n * 8 + 8);
      for (size_t i = 0; i < n; ++i) {
        if (oc.is_f64) std::memcpy(vals.data() + i * 8, &oc.fv[i], 8);
        else if (oc.width == 8) std::memcpy(vals.data() + i * 8, &oc.iv[i], 8);
        else if (oc.width == 4) { int32_t v = (int32_t)oc.iv[i]; std::memcpy(vals.data() + i * 4, &v, 4); }
        else { int16_t v = (int16_t)oc.iv[i]; std::memcpy(vals.data() + i * 2, &v, 2); }
      }
      h->bufs.push_back(std::move(vals)); h->ptrs[c].push_back(h->bufs.back().data());
      k.n_buffers = 2;
    }
    k.length = (int64_t)n;
    k.null_count = nulls;
    k.buffers = h->ptrs[c].data();
    k.release = release_child;
  }
  for (auto& k : h->kids) h->kid_ptrs.push_back(&k);
  std::memset(out, 0, sizeof(*out));
  out->length = (int64_t)n;
  out->n_buffers = 1;
  out->buffers = h->top_bufs;
  out->n_children = (int64_t)cols.size();
  out->children = h->kid_ptrs.data();
  out->release = release_array;
  out->private_data = h;
}

struct SchemaHolder {
  std::vector<std::string> names, formats;
  std::vector<ArrowSchema> kids;
  std::vector<ArrowSchema*> kid_ptrs;
};
void release_schema(ArrowSchema* s) { delete (SchemaHolder*)s->private_data; s->release = nullptr; }
void release_schema_child(ArrowSchema* s) { s->release = nullptr; }

void export_schema(const std::vector<OutCol>& cols, ArrowSchema* out) {
  auto* h = new SchemaHolder();
  for (auto& c : cols) { h->names.push_back(c.name); h->formats.push_back(c.format); }
  h->kids.resize(cols.size());
  for (size_t i = 0; i < cols.size(); ++i) {
    ArrowSchema& k = h->kids[i];
    std::memset(&k, 0, sizeof(k));
    k.format = h->formats[i].c_str();
    k.name = h->names[i].c_str();
    k.flags = cols[i].nullable ? ARROW_FLAG_NULLABLE : 0;
    k.release = release_schema_child;
  }
  for (auto& k : h->kids) h->kid_ptrs.push_back(&k);
  std::memset(out, 0, sizeof(*out));
  out->format = "+s";
  out->name = "";
  out->n_children = (int64_t)cols.size();
  out->children = h->kid_ptrs.data();
  out->release = release_schema;
  out->private_data = h;
}

// ------------------------------------------------------------ plan set-up ---

std::string agg_default_name(const Plan& p, const prism_gpu_agg& a) {
  if (a.name) return a.name;
  std::string col = a.column >= 0 ? p.in_names[a.column] : "*";
  switch (a.func) {
    case PRISM_AGG_COUNT_STAR: return "COUNT(*)";
    case PRISM_AGG_COUNT: return "COUNT(" + col + ")";
    case PRISM_AGG_SUM: return "SUM(" + col + ")";
    case PRISM_AGG_MIN: return "MIN(" + col + ")";
    case PRISM_AGG_MAX: return "MAX(" + col + ")";
    case PRISM_AGG_AVG: return "AVG(" + col + ")";
    case PRISM_AGG_COUNT_DISTINCT: return "COUNT(DISTINCT " + col + ")";
  }
  return "?";
}

Plan make_plan(const ArrowSchema* schema, const prism_gpu_plan_desc* d) {
  if (!d || d->abi_version != PRISM_GPU_ABI_VERSION) fail(PRISM_GPU_EINVAL, "oracle: bad plan version");
  if (std::string(schema->format) != "+s") fail(PRISM_GPU_EINVAL, "oracle: schema is not a struct");
  Plan p;
  p.d = d;
  for (int64_t c = 0; c < schema->n_children; ++c) {
    const ArrowSchema* f = schema->children[c];
    p.in.push_back(type_from_format(f->dictionary ? f->dictionary->format : f->format));
    p.in_names.push_back(f->name ? f->name : "");
  }
  if (d->n_keys > MAXK) fail(PRISM_GPU_EUNSUPPORTED, "oracle: too many group keys");
  if (d->predicate >= 0) validate_expr(d, d->predicate, p.in);
  p.pred.d = d;
  p.pred.like.resize(d->n_exprs);
  for (uint32_t e = 0; e < d->n_exprs; ++e)
    if (d->exprs[e].kind == PRISM_EXPR_LIKE) {
      const prism_gpu_expr& r = d->exprs[d->exprs[e].rhs];
      p.pred.like[e] = like_compile(std::string_view(r.str, (size_t)r.str_len));
    }
  for (uint32_t i = 0; i < d->n_keys; ++i) {
    const prism_gpu_key& k = d->keys[i];
    if (k.column < 0 || (size_t)k.column >= p.in.size()) fail(PRISM_GPU_EINVAL, "oracle: key column out of range");
    if (k.kind == PRISM_KEY_DATE_BIN) {
      if (p.in[k.column].logical != PRISM_COL_TIMESTAMP) fail(PRISM_GPU_EUNSUPPORTED, "oracle: date_bin needs a timestamp");
      if (k.stride_ms <= 0) fail(PRISM_GPU_EINVAL, "oracle: date_bin stride must be positive");
    }
  }
  uint32_t n_distinct = 0;
  for (uint32_t a = 0; a < d->n_aggs; ++a) {
    const prism_gpu_agg& ag = d->aggs[a];
    if (ag.func == PRISM_AGG_COUNT_STAR) continue;
    if (ag.column < 0 || (size_t)ag.column >= p.in.size()) fail(PRISM_GPU_EINVAL, "oracle: aggregate column out of range");
    const ColType& t = p.in[ag.column];
    if (ag.func == PRISM_AGG_COUNT_DISTINCT) {
      ++n_distinct;
      if (p.distinct_col >= 0 && p.distinct_col != ag.column) fail(PRISM_GPU_EUNSUPPORTED, "oracle: several DISTINCT columns");
      p.distinct_col = ag.column;
      continue;
    }
    if (ag.func == PRISM_AGG_COUNT) continue;
    if (t.phys == Phys::STR) fail(PRISM_GPU_EUNSUPPORTED, "oracle: aggregate over a string column");
    if ((ag.func == PRISM_AGG_SUM || ag.func == PRISM_AGG_AVG) && t.logical == PRISM_COL_TIMESTAMP)
      fail(PRISM_GPU_EUNSUPPORTED, "oracle: SUM/AVG over a timestamp");
  }
  if (n_distinct) {
    if (n_distinct != d->n_aggs) fail(PRISM_GPU_EUNSUPPORTED, "oracle: DISTINCT mixed with other aggregates");
    p.distinct_plan = true;
  }
  return p;
}

void init_table(const Plan& p, GroupTable& T) {
  T.n_aggs = p.d->n_aggs;
  T.key_strs.resize(p.d->n_keys);
}

std::string key_format(const Plan& p, uint32_t i) {
  const ColType& t = p.in[p.d->keys[i].column];
  return t.format;
}
std::string key_name(const Plan& p, uint32_t i) {
  return p.d->keys[i].name ? p.d->keys[i].name : p.in_names[p.d->keys[i].column];
}

std::vector<OutCol> final_columns(const Plan& p) {
  std::vector<OutCol> cols;
  for (uint32_t i = 0; i < p.d->n_keys; ++i) cols.push_back(make_col(key_name(p, i), key_format(p, i), true));
  for (uint32_t a = 0; a < p.d->n_aggs; ++a) {
    const prism_gpu_agg& ag = p.d->aggs[a];
    std::string name = agg_default_name(p, ag);
    switch (ag.func) {
      case PRISM_AGG_COUNT_STAR: case PRISM_AGG_COUNT: case PRISM_AGG_COUNT_DISTINCT:
        cols.push_back(make_col(name, "l", false)); break;
      case PRISM_AGG_SUM: cols.push_back(make_col(name, is_unsigned_sum(p.in[ag.column]) ? "L" : "l", true)); break;
      case PRISM_AGG_MIN: case PRISM_AGG_MAX: cols.push_back(make_col(name, p.in[ag.column].format, true)); break;
      case PRISM_AGG_AVG: cols.push_back(make_col(name, "g", true)); break;
    }
  }
  return cols;
}

std::vector<OutCol> partial_columns(const Plan& p) {
  std::vector<OutCol> cols;
  for (uint32_t i = 0; i < p.d->n_keys; ++i) cols.push_back(make_col(key_name(p, i), key_format(p, i), true));
  if (p.distinct_plan) {
    cols.push_back(make_col("alias1", p.in[p.distinct_col].format, true));
    return cols;
  }
  for (uint32_t a = 0; a < p.d->n_aggs; ++a) {
    const prism_gpu_agg& ag = p.d->aggs[a];
    std::string name = agg_default_name(p, ag);
    switch (ag.func) {
      case PRISM_AGG_COUNT_STAR: case PRISM_AGG_COUNT: cols.push_back(make_col(name + "[count]", "l", false)); break;
      case PRISM_AGG_SUM: cols.push_back(make_col(name + "[sum]", is_unsigned_sum(p.in[ag.column]) ? "L" : "l", true)); break;
      case PRISM_AGG_MIN: cols.push_back(make_col(name + "[min]", p.in[ag.column].format, true)); break;
      case PRISM_AGG_MAX: cols.push_back(make_col(name + "[max]", p.in[ag.column].format, true)); break;
      case PRISM_AGG_AVG:
        cols.push_back(make_col(name + "[count]", "L", false));
        cols.push_back(make_col(name + "[sum]", "g", true));
        break;
    }
  }
  return cols;
}

void push_key(const Plan& p, const GroupTable& T, std::vector<OutCol>& cols, const GKey& k) {
  for (uint32_t i = 0; i < p.d->n_keys; ++i) {
    if (k.nulls & (1u << i)) cols[i].push_null();
    else if (cols[i].is_str) cols[i].push_s(T.key_strs[i].strs[(size_t)k.v[i]]);
    else cols[i].push_i(k.v[i]);
  }
}

void emit(const Plan& p, GroupTable& T, bool partial, ArrowArray* out) {
  const prism_gpu_plan_desc* d = p.d;
  // An ungrouped aggregate always produces one row, even over no input.
  if (d->n_keys == 0 && T.keys.empty()) T.group_of(GKey{});
  std::vector<OutCol> cols = partial ? partial_columns(p) : final_columns(p);
  if (partial && p.distinct_plan) {
    // inner GROUP BY (keys, x): one row per distinct pair, plus a NULL-x row for
    // groups that only saw NULLs so that the group itself survives the merge.
    for (size_t g = 0; g < T.keys.size(); ++g) {
      AggState& S = T.st[g * T.n_aggs];
      OutCol& xc = cols[d->n_keys];
      if (!S.distinct || S.distinct->empty()) { push_key(p, T, cols, T.keys[g]); xc.push_null(); continue; }
      for (int64_t v : *S.distinct) {
        push_key(p, T, cols, T.keys[g]);
        if (xc.is_str) xc.push_s(T.distinct_strs.strs[(size_t)v]); else xc.push_i(v);
      }
    }
    export_cols(cols, out);
    return;
  }
  std::vector<size_t> order(T.keys.size());
  for (size_t g = 0; g < order.size(); ++g) order[g] = g;
  for (size_t g : order) {
    push_key(p, T, cols, T.keys[g]);
    AggState* S = &T.st[g * T.n_aggs];
    size_t c = d->n_keys;
    for (uint32_t a = 0; a < d->n_aggs; ++a) {
      switch (d->aggs[a].func) {
        case PRISM_AGG_COUNT_STAR: case PRISM_AGG_COUNT: cols[c++].push_i(S[a].count); break;
        case PRISM_AGG_COUNT_DISTINCT: cols[c++].push_i(S[a].distinct ? (int64_t)S[a].distinct->size() : 0); break;
        case PRISM_AGG_SUM: if (S[a].count) cols[c].push_i((int64_t)S[a].sum); else cols[c].push_null(); ++c; break;
        case PRISM_AGG_MIN: if (S[a].count) cols[c].push_i(S[a].mn); else cols[c].push_null(); ++c; break;
        case PRISM_AGG_MAX: if (S[a].count) cols[c].push_i(S[a].mx); else cols[c].push_null(); ++c; break;
        case PRISM_AGG_AVG:
          if (partial) {
            cols[c++].push_i(S[a].count);
            if (S[a].count) cols[c].push_f(S[a].fsum); else cols[c].push_null();
            ++c;
          } else {
            if (S[a].count) cols[c].push_f(S[a].fsum / (double)S[a].count); else cols[c].push_null();
            ++c;
          }
          break;
      }
    }
  }
  if (!partial && (d->order_by >= 0 || d->limit >= 0)) {
    size_t n = cols.empty() ? 0 : cols[0].rows();
    std::vector<size_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = i;
    if (d->order_by >= 0) {
      if ((size_t)d->order_by >= cols.size()) fail(PRISM_GPU_EINVAL, "oracle: order_by out of range");
      const OutCol& oc = cols[d->order_by];
      bool desc = d->descending != 0;
      bool uns = oc.format == "L";
      std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
        bool na = !oc.valid[a], nb = !oc.valid[b];
        if (na || nb) return desc ? (na && !nb) : (!na && nb);  // NULLS FIRST when DESC, LAST when ASC
        int cmp;
        if (oc.is_str) cmp = oc.sv[a].compare(oc.sv[b]);
        else if (oc.is_f64) cmp = oc.fv[a] < oc.fv[b] ? -1 : oc.fv[a] > oc.fv[b] ? 1 : 0;
        else if (uns) cmp = (uint64_t)oc.iv[a] < (uint64_t)oc.iv[b] ? -1 : (uint64_t)oc.iv[a] > (uint64_t)oc.iv[b] ? 1 : 0;
        else cmp = oc.iv[a] < oc.iv[b] ? -1 : oc.iv[a] > oc.iv[b] ? 1 : 0;
        return desc ? cmp > 0 : cmp < 0;
      });
    }
    if (d->limit >= 0 && (size_t)d->limit < idx.size()) idx.resize((size_t)d->limit);
    std::vector<OutCol> sorted;
    for (auto& c : cols) {
      OutCol s = make_col(c.name, c.format, c.nullable);
      for (size_t i : idx) {
        if (!c.valid[i]) s.push_null();
        else if (c.is_str) s.push_s(c.sv[i]);
        else if (c.is_f64) s.push_f(c.fv[i]);
        else s.push_i(c.iv[i]);
      }
      sorted.push_back(std::move(s));
    }
    cols.swap(sorted);
  }
  export_cols(cols, out);
}

// One thread per slice of the batch list, then a sequential fold: the shape of
// DataFusion's Partial -> Final pipeline with one partition per thread.
void run_threads(const Plan& p, const std::vector<const Batch*>& batches, bool merge, int threads, GroupTable& T) {
  init_table(p, T);
  if (threads <= 1 || batches.size() <= 1) {
    for (const Batch* b : batches) { if (merge) update_merge(p, T, *b); else update_raw(p, T, *b); }
    return;
  }
  int nt = std::min<int>(threads, (int)batches.size());
  std::vector<std::unique_ptr<GroupTable>> parts;
  for (int t = 0; t < nt; ++t) { parts.emplace_back(new GroupTable()); init_table(p, *parts.back()); }
  std::vector<std::thread> th;
  std::atomic<size_t> next{0};
  std::vector<OracleError> errs(nt, OracleError{0, ""});
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t]() {
      try {
        for (;;) {
          size_t i = next.fetch_add(1);
          if (i >= batches.size()) break;
          if (merge) update_merge(p, *parts[t], *batches[i]); else update_raw(p, *parts[t], *batches[i]);
        }
      } catch (OracleError& e) { errs[t] = e; }
    });
  for (auto& x : th) x.join();
  for (auto& e : errs) if (e.code) throw e;
  // The Final stage.  Many groups and keys that are plain integers: the partial tables are
  // repartitioned by key hash, one slice per thread (DataFusion's RepartitionExec(Hash) +
  // AggregateExec(FinalPartitioned)), and the disjoint results are strung together.  Else
  // (few groups, or string keys, whose ids are per table) one sequential fold.
  size_t partial_groups = 0;
  for (auto& part : parts) partial_groups += part->keys.size();
  bool ints_only = parts[0]->distinct_strs.strs.empty();
  for (auto& part : parts) {
    for (auto& ks : part->key_strs) ints_only = ints_only && ks.strs.empty();
    ints_only = ints_only && part->distinct_strs.strs.empty();
  }
  if (partial_groups < 200000 || !ints_only) {
    for (int t = 0; t < nt; ++t) fold(p, T, *parts[t]);
    return;
  }
  std::vector<std::unique_ptr<GroupTable>> finals;
  for (int t = 0; t < nt; ++t) { finals.emplace_back(new GroupTable()); init_table(p, *finals.back()); }
  th.clear();
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t]() {
      try {
        for (auto& part : parts) fold(p, *finals[t], *part, (size_t)nt, (size_t)t);
      } catch (OracleError& e) { errs[t] = e; }
    });
  for (auto& x : th) x.join();
  for (auto& e : errs) if (e.code) throw e;
  for (auto& f : finals) {
    T.rows_in += f->rows_in;
    T.rows_selected += f->rows_selected;
    T.keys.insert(T.keys.end(), f->keys.begin(), f->keys.end());
    T.st.insert(T.st.end(), f->st.begin(), f->st.end());
    for (auto& st : f->st) st.distinct = nullptr;  // the sets moved with their states
  }
}

template <class F>
int guarded(F&& f) {
  try { f(); return 0; }
  catch (OracleError& e) { g_err = e.msg; return e.code; }
  catch (std::exception& e) { g_err = std::string("oracle: ") + e.what(); return PRISM_GPU_EINTERNAL; }
}

// ------------------------------------------------------- synthetic tables ---

struct SynthTable {
  std::vector<ColType> types;
  std::vector<std::unique_ptr<Batch>> batches;
  uint64_t rows = 0;
};

}  // namespace

extern "C" {

const char* prism_oracle_last_error(void) { return g_err.c_str(); }

// Runs `plan` (mode SINGLE or PARTIAL) over borrowed RecordBatches.
int prism_oracle_execute(const ArrowSchema* schema, ArrowArray* const* batches, size_t n_batches,
                         const prism_gpu_plan_desc* desc, int threads, ArrowArray* out) {
  return guarded([&] {
    Plan p = make_plan(schema, desc);
    std::vector<Batch> bs;
    for (size_t i = 0; i < n_batches; ++i) bs.push_back(batch_of(schema, batches[i]));
    std::vector<const Batch*> ptrs;
    for (auto& b : bs) ptrs.push_back(&b);
    GroupTable T;
    run_threads(p, ptrs, false, threads, T);
    emit(p, T, desc->mode == PRISM_MODE_PARTIAL, out);
  });
}

// AggregateExec(Final): merges PARTIAL outputs of `plan`.  `input_schema` is the
// schema the plan was written against; `state_schema` describes the state arrays.
int prism_oracle_merge(const ArrowSchema* input_schema, const ArrowSchema* state_schema,
                       ArrowArray* const* states, size_t n_states, const prism_gpu_plan_desc* desc,
                       ArrowArray* out) {
  return guarded([&] {
    Plan p = make_plan(input_schema, desc);
    std::vector<Batch> bs;
    for (size_t i = 0; i < n_states; ++i) bs.push_back(batch_of(state_schema, states[i]));
    std::vector<const Batch*> ptrs;
    for (auto& b : bs) ptrs.push_back(&b);
    GroupTable T;
    run_threads(p, ptrs, true, 1, T);
    emit(p, T, false, out);
  });
}

int prism_oracle_output_schema(const ArrowSchema* input_schema, const prism_gpu_plan_desc* desc, int partial,
                               ArrowSchema* out) {
  return guarded([&] {
    Plan p = make_plan(input_schema, desc);
    std::vector<OutCol> cols = partial ? partial_columns(p) : final_columns(p);
    export_schema(cols, out);
  });
}

// Synthetic table: `n_parts` partitions generated with prism_synth.h, rows
// [first_row, first_row + n_rows) of a `total_rows` table.  Utf8 columns are
// generated as dictionary codes and carry `dict_values` as their dictionary.
int prism_oracle_synth_create(const ArrowSchema* schema, uint64_t seed, uint64_t first_row, uint64_t n_rows,
                              uint64_t total_rows, uint32_t n_parts, const prism_gpu_synth_col* cols,
                              uint32_t n_cols, int threads, void** out) {
  return guarded([&] {
    auto* T = new SynthTable();
    std::unique_ptr<SynthTable> guard(T);
    for (int64_t c = 0; c < schema->n_children; ++c) T->types.push_back(type_from_format(schema->children[c]->format));
    if (n_cols != T->types.size()) fail(PRISM_GPU_EINVAL, "oracle: one generator per column is required");
    if (n_parts == 0) n_parts = 1;
    T->batches.resize(n_parts);
    T->rows = n_rows;
    auto gen_part = [&](uint32_t pi) {
      uint64_t lo = n_rows * pi / n_parts, hi = n_rows * (pi + 1) / n_parts;
      auto b = std::make_unique<Batch>();
      b->n = (int64_t)(hi - lo);
      b->owned.resize(2 * (size_t)n_cols);  // [n_cols + ci]: the bytes of a generated string column
      for (uint32_t ci = 0; ci < n_cols; ++ci) {
        const prism_gpu_synth_col& sc = cols[ci];
        const ColType& t = T->types[sc.column];
        if (sc.kind == PRISM_SYNTH_URL) {
          if (t.phys != Phys::STR) fail(PRISM_GPU_EINVAL, "oracle: the URL generator needs a string column");
          uint64_t key = prism_synth_column_key(seed, (uint32_t)sc.column);
          std::vector<uint8_t>& offs = b->owned[ci];
          offs.resize((size_t)(hi - lo + 1) * 4);
          int32_t* o = (int32_t*)offs.data();
          uint64_t pos = 0;
          o[0] = 0;
          for (uint64_t r = lo; r < hi; ++r) {
            pos += prism_synth_url_length(sc.cdf, sc.cdf_len, key, first_row + r);
            if (pos > 0x7fffffffull) fail(PRISM_GPU_EUNSUPPORTED, "oracle: string data of one partition exceeds 2 GiB");
            o[r - lo + 1] = (int32_t)pos;
          }
          std::vector<uint8_t>& data = b->owned[n_cols + ci];
          data.resize((size_t)pos + 8);
          for (uint64_t r = lo; r < hi; ++r)
            prism_synth_url_bytes(data.data() + o[r - lo], (uint32_t)(o[r - lo + 1] - o[r - lo]), sc.p, key, first_row + r);
          continue;
        }
        int w = t.phys == Phys::STR ? 4 : (t.phys == Phys::I16 || t.phys == Phys::U16) ? 2 : t.phys == Phys::I32 ? 4 : 8;
        b->owned[ci].resize((size_t)(hi - lo) * w + 8);
        uint8_t* dst = b->owned[ci].data();
        uint64_t key = prism_synth_column_key(seed, (uint32_t)sc.column);
        for (uint64_t r = lo; r < hi; ++r) {
          int64_t v = prism_synth_value(sc.kind, sc.p, sc.cdf, sc.cdf_len, key, first_row + r, total_rows);
          size_t i = (size_t)(r - lo);
          if (w == 8) std::memcpy(dst + i * 8, &v, 8);
          else if (w == 4) { int32_t x = (int32_t)v; std::memcpy(dst + i * 4, &x, 4); }
          else { int16_t x = (int16_t)v; std::memcpy(dst + i * 2, &x, 2); }
        }
      }
      b->cols.resize(T->types.size());
      for (uint32_t ci = 0; ci < n_cols; ++ci) {
        ColView v;
        v.type = T->types[cols[ci].column];
        if (cols[ci].kind == PRISM_SYNTH_URL) v.data = b->owned[n_cols + ci].data();  // real strings
        else if (v.type.phys == Phys::STR) { v.type.phys = Phys::I32; }  // dictionary codes stand for the strings
        v.n = b->n;
        v.values = b->owned[ci].data();
        b->cols[cols[ci].column] = v;
      }
      T->batches[pi] = std::move(b);
    };
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> th;
    int nt = std::max(1, std::min<int>(threads, (int)n_parts));
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&]() { for (;;) { uint32_t i = next.fetch_add(1); if (i >= n_parts) break; gen_part(i); } });
    for (auto& x : th) x.join();
    *out = guard.release();
  });
}

void prism_oracle_synth_free(void* t) { delete (SynthTable*)t; }

// Executes over a synthetic table; *elapsed_ms receives the wall time of the
// scan -> filter -> aggregate -> merge work alone (generation excluded).
// Group keys over generated Utf8 columns come out as Int32 dictionary codes.
int prism_oracle_synth_execute(void* table, const ArrowSchema* schema, const prism_gpu_plan_desc* desc,
                               int threads, ArrowArray* out, double* elapsed_ms) {
  return guarded([&] {
    auto* T = (SynthTable*)table;
    Plan p = make_plan(schema, desc);
    for (auto& t : p.in) if (t.phys == Phys::STR) { t.phys = Phys::I32; t.format = "i"; t.logical = PRISM_COL_INT32; }
    std::vector<const Batch*> ptrs;
    for (auto& b : T->batches) ptrs.push_back(b.get());
    auto t0 = std::chrono::steady_clock::now();
    GroupTable G;
    run_threads(p, ptrs, false, threads, G);
    auto t1 = std::chrono::steady_clock::now();
    if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    emit(p, G, desc->mode == PRISM_MODE_PARTIAL, out);
  });
}

}  // extern "C"
