// Plan validation and normalisation: what DataFusion's physical planner does
// between TableProvider::scan (src/prism-query/src/providers/table_provider.rs:73-132)
// and the first poll of the aggregate stream.
#include <algorithm>
#include <cstring>

#include "engine.hpp"

namespace prism {

std::vector<LikeToken> like_compile(const std::string& pattern) {
  std::vector<LikeToken> t;
  for (size_t i = 0; i < pattern.size(); ++i) {
    uint8_t c = (uint8_t)pattern[i];
    if (c == '\\' && i + 1 < pattern.size()) t.push_back({0, (uint8_t)pattern[++i]});
    else if (c == '%') { if (t.empty() || t.back().kind != 2) t.push_back({2, 0}); }
    else if (c == '_') t.push_back({1, 0});
    else t.push_back({0, c});
  }
  return t;
}

namespace {
inline uint8_t fold_ascii(uint8_t c, bool ci) { return (ci && c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }
inline size_t utf8_step(const uint8_t* s, size_t i, size_t n) {
  uint8_t c = s[i];
  size_t l = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1;
  return l < n - i ? l : n - i;
}
}  // namespace

bool like_match_host(const std::vector<LikeToken>& t, const std::string& str, bool ci) {
  const uint8_t* s = (const uint8_t*)str.data();
  size_t n = str.size(), si = 0, ti = 0, star_t = (size_t)-1, star_s = 0, nt = t.size();
  while (si < n) {
    if (ti < nt && t[ti].kind == 0 && fold_ascii(t[ti].byte, ci) == fold_ascii(s[si], ci)) { ++si; ++ti; }
    else if (ti < nt && t[ti].kind == 1) { si += utf8_step(s, si, n); ++ti; }
    else if (ti < nt && t[ti].kind == 2) { star_t = ti++; star_s = si; }
    else if (star_t != (size_t)-1) { ti = star_t + 1; star_s += utf8_step(s, star_s, n); si = star_s; }
    else return false;
  }
  while (ti < nt && t[ti].kind == 2) ++ti;
  return ti == nt;
}

namespace {

int cmp_mask_of(int op) {
  switch (op) {
    case PRISM_CMP_EQ: return kEq;
    case PRISM_CMP_NEQ: return kLt | kGt;
    case PRISM_CMP_LT: return kLt;
    case PRISM_CMP_LTEQ: return kLt | kEq;
    case PRISM_CMP_GT: return kGt;
    case PRISM_CMP_GTEQ: return kGt | kEq;
  }
  fail(PRISM_GPU_EINVAL, "unknown comparison operator " + std::to_string(op));
}

struct Builder {
  const prism_gpu_plan_desc* d;
  prism_gpu_plan* plan;

  const prism_gpu_expr& at(int i) const {
    if (i < 0 || (uint32_t)i >= d->n_exprs) fail(PRISM_GPU_EINVAL, "expression index " + std::to_string(i) + " out of range");
    return d->exprs[i];
  }
  int column_of(int i) const {
    const prism_gpu_expr& e = at(i);
    if (e.kind != PRISM_EXPR_COLUMN) fail(PRISM_GPU_EUNSUPPORTED, "only <column> <op> <literal> predicates are supported");
    if (e.column < 0 || (size_t)e.column >= plan->fields.size()) fail(PRISM_GPU_EINVAL, "predicate column out of range");
    return e.column;
  }
  int add(const PredNode& n) { plan->pred.push_back(n); return (int)plan->pred.size() - 1; }

  // Lowers expression i; `neg` = an odd number of NOTs above it.  De Morgan and the
  // complement of each leaf hold in three-valued logic (NULL stays NULL), so after
  // this pass the tree has no NOT and a selection bitmap of TRUE rows is enough.
  int lower(int i, bool neg, int depth = 0) {
    if (depth > 64) fail(PRISM_GPU_EINVAL, "predicate nesting too deep");
    const prism_gpu_expr& e = at(i);
    switch (e.kind) {
      case PRISM_EXPR_NOT: return lower(e.lhs, !neg, depth + 1);
      case PRISM_EXPR_AND:
      case PRISM_EXPR_OR: {
        PredNode n;
        bool is_and = (e.kind == PRISM_EXPR_AND) != neg;
        n.kind = is_and ? PredNode::AND : PredNode::OR;
        n.lhs = lower(e.lhs, neg, depth + 1);
        n.rhs = lower(e.rhs, neg, depth + 1);
        return add(n);
      }
      case PRISM_EXPR_CMP: {
        PredNode n;
        n.kind = PredNode::CMP;
        n.column = column_of(e.lhs);
        n.cmp_mask = cmp_mask_of(e.op);
        if (neg) n.cmp_mask ^= (kLt | kEq | kGt);
        const prism_gpu_expr& r = at(e.rhs);
        const FieldType& f = plan->fields[n.column];
        if (f.is_float) fail(PRISM_GPU_EUNSUPPORTED, "comparisons on Float64 are not supported");
        if (f.is_string) {
          if (r.kind != PRISM_EXPR_LIT_STR) fail(PRISM_GPU_EUNSUPPORTED, "string column compared with a non-string literal");
          n.lit_str.assign(r.str ? r.str : "", (size_t)r.str_len);
        } else {
          if (r.kind != PRISM_EXPR_LIT_I64) fail(PRISM_GPU_EUNSUPPORTED, "integer column compared with a non-integer literal");
          n.lit_i64 = r.i64;
        }
        return add(n);
      }
      case PRISM_EXPR_LIKE: {
        PredNode n;
        n.kind = PredNode::LIKE;
        n.column = column_of(e.lhs);
        if (plan->fields[n.column].logical != PRISM_COL_UTF8) fail(PRISM_GPU_EUNSUPPORTED, "LIKE needs a Utf8 column");
        const prism_gpu_expr& r = at(e.rhs);
        if (r.kind != PRISM_EXPR_LIT_STR) fail(PRISM_GPU_EUNSUPPORTED, "LIKE pattern must be a string literal");
        n.lit_str.assign(r.str ? r.str : "", (size_t)r.str_len);
        n.negated = ((e.flags & PRISM_LIKE_NEGATED) != 0) != neg;
        n.case_insensitive = (e.flags & PRISM_LIKE_CASE_INSENSITIVE) != 0;
        return add(n);
      }
      case PRISM_EXPR_IS_NULL:
      case PRISM_EXPR_IS_NOT_NULL: {
        PredNode n;
        bool is_null = (e.kind == PRISM_EXPR_IS_NULL) != neg;
        n.kind = is_null ? PredNode::IS_NULL : PredNode::IS_NOT_NULL;
        n.column = column_of(e.lhs);
        return add(n);
      }
      default: fail(PRISM_GPU_EINVAL, "expression kind " + std::to_string(e.kind) + " is not a predicate");
    }
  }
};

std::string default_agg_name(const prism_gpu_plan& p, const AggPlan& a) {
  std::string col = a.column >= 0 ? p.fields[a.column].name : "*";
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

FieldType named(const std::string& name, const std::string& format, bool nullable) {
  FieldType f = field_from_format(format.c_str());
  f.name = name;
  f.nullable = nullable;
  return f;
}

}  // namespace

std::unique_ptr<prism_gpu_plan> plan_create(prism_gpu_ctx* ctx, const ArrowSchema* schema,
                                            const prism_gpu_plan_desc* d) {
  if (!d) fail(PRISM_GPU_EINVAL, "plan description is null");
  if (d->abi_version != PRISM_GPU_ABI_VERSION)
    fail(PRISM_GPU_EINVAL, "plan description has ABI version " + std::to_string(d->abi_version) + ", library has " +
                               std::to_string(PRISM_GPU_ABI_VERSION));
  auto plan = std::make_unique<prism_gpu_plan>();
  plan->ctx = ctx;
  plan->fields = parse_schema(schema);
  if (d->mode != PRISM_MODE_SINGLE && d->mode != PRISM_MODE_PARTIAL) fail(PRISM_GPU_EINVAL, "unknown aggregate mode");
  plan->mode = d->mode;
  if (d->n_aggs == 0) fail(PRISM_GPU_EUNSUPPORTED, "a plan needs at least one aggregate");
  if (d->n_keys > (uint32_t)kMaxKeys) fail(PRISM_GPU_EUNSUPPORTED, "at most " + std::to_string(kMaxKeys) + " GROUP BY keys");
  if ((d->n_exprs && !d->exprs) || (d->n_keys && !d->keys) || !d->aggs) fail(PRISM_GPU_EINVAL, "null array in plan description");

  if (d->predicate >= 0) {
    Builder b{d, plan.get()};
    plan->pred_root = b.lower(d->predicate, false);
  }
  for (uint32_t i = 0; i < d->n_keys; ++i) {
    const prism_gpu_key& k = d->keys[i];
    if (k.column < 0 || (size_t)k.column >= plan->fields.size()) fail(PRISM_GPU_EINVAL, "GROUP BY column out of range");
    const FieldType& f = plan->fields[k.column];
    KeyPlan kp;
    kp.kind = k.kind;
    kp.column = k.column;
    if (k.kind == PRISM_KEY_DATE_BIN) {
      if (f.logical != PRISM_COL_TIMESTAMP) fail(PRISM_GPU_EUNSUPPORTED, "date_bin needs a Timestamp(ms) column");
      if (k.stride_ms <= 0) fail(PRISM_GPU_EINVAL, "date_bin stride must be a positive number of milliseconds");
      kp.stride_ms = k.stride_ms;
      kp.origin_ms = k.origin_ms;
    } else if (k.kind != PRISM_KEY_COLUMN) {
      fail(PRISM_GPU_EINVAL, "unknown GROUP BY key kind");
    }
    if (f.is_float) fail(PRISM_GPU_EUNSUPPORTED, "GROUP BY on Float64 is not supported");
    kp.name = k.name ? k.name : f.name;
    plan->keys.push_back(kp);
  }
  uint32_t n_distinct = 0;
  bool several_distinct_columns = false, string_min_max = false;
  for (uint32_t i = 0; i < d->n_aggs; ++i) {
    const prism_gpu_agg& a = d->aggs[i];
    AggPlan ap;
    ap.func = a.func;
    ap.column = a.column;
    if (a.func < PRISM_AGG_COUNT_STAR || a.func > PRISM_AGG_COUNT_DISTINCT) fail(PRISM_GPU_EINVAL, "unknown aggregate function");
    if (a.func == PRISM_AGG_COUNT_STAR) {
      ap.column = -1;
    } else {
      if (a.column < 0 || (size_t)a.column >= plan->fields.size()) fail(PRISM_GPU_EINVAL, "aggregate column out of range");
      const FieldType& f = plan->fields[a.column];
      if (a.func == PRISM_AGG_COUNT_DISTINCT) {
        ++n_distinct;
        if (plan->distinct_col >= 0 && plan->distinct_col != a.column) several_distinct_columns = true;
        plan->distinct_col = a.column;
      } else if (a.func != PRISM_AGG_COUNT) {
        if (f.is_string && (a.func == PRISM_AGG_MIN || a.func == PRISM_AGG_MAX)) string_min_max = true;
        else if (f.is_string) fail(PRISM_GPU_EUNSUPPORTED, "SUM / AVG over string columns");
        if (f.is_float) fail(PRISM_GPU_EUNSUPPORTED, "aggregates over Float64 columns are not supported");
        if ((a.func == PRISM_AGG_SUM || a.func == PRISM_AGG_AVG) && f.logical == PRISM_COL_TIMESTAMP)
          fail(PRISM_GPU_EUNSUPPORTED, "SUM/AVG over a timestamp");
      }
    }
    ap.name = a.name ? a.name : default_agg_name(*plan, ap);
    plan->aggs.push_back(ap);
  }
  if ((n_distinct && (n_distinct != d->n_aggs || several_distinct_columns)) || string_min_max) {
    // COUNT(DISTINCT) next to other aggregates or over several columns, MIN / MAX over strings:
    // one plan per kind over the same rows, joined on the keys
    // kind 0: the plain aggregates; then one kind per distinct column (kind id = column) and
    // one per string column under MIN / MAX (kind id = -2 - column)
    std::vector<std::vector<prism_gpu_agg>> kinds(1);
    std::vector<int> kind_id(1, -1);
    for (uint32_t i = 0; i < d->n_aggs; ++i) {
      const prism_gpu_agg& a = d->aggs[i];
      const bool str_mm = (a.func == PRISM_AGG_MIN || a.func == PRISM_AGG_MAX) && plan->fields[a.column].is_string;
      const int id = a.func == PRISM_AGG_COUNT_DISTINCT ? a.column : str_mm ? -2 - a.column : -1;
      size_t k = 0;
      while (k < kinds.size() && kind_id[k] != id) ++k;
      if (k == kinds.size()) { kinds.emplace_back(); kind_id.push_back(id); }
      plan->mixed_map.emplace_back((int)k, (int)kinds[k].size());
      kinds[k].push_back(a);
    }
    const prism_gpu_agg count_rows{PRISM_AGG_COUNT_STAR, -1, nullptr};
    for (size_t k = 0; k < kinds.size(); ++k) {
      prism_gpu_plan_desc sub = *d;
      sub.mode = PRISM_MODE_SINGLE;  // (the plan asks a sub-plan for state rows where it wants them: exec.cpp, execute_mixed)
      sub.order_by = -1;
      sub.limit = -1;
      std::vector<prism_gpu_key> keys(d->keys, d->keys + d->n_keys);
      std::vector<prism_gpu_agg> aggs = kinds[k];
      int string_col = -1;
      if (kind_id[k] <= -2) {  // GROUP BY keys, column: every (group, value) pair once
        string_col = -2 - kind_id[k];
        keys.push_back(prism_gpu_key{PRISM_KEY_COLUMN, string_col, 0, 0, "__value"});
        if (keys.size() > (size_t)kMaxKeys) fail(PRISM_GPU_EUNSUPPORTED, "too many GROUP BY keys for MIN / MAX over a string column");
        aggs.assign(1, count_rows);
      } else if (aggs.empty()) {
        aggs.assign(1, count_rows);  // no plain aggregate: sub-plan 0 only lists the groups
      }
      sub.keys = keys.data();
      sub.n_keys = (uint32_t)keys.size();
      sub.aggs = aggs.data();
      sub.n_aggs = (uint32_t)aggs.size();
      plan->subs.push_back(plan_create(ctx, schema, &sub));
      plan->sub_string_col.push_back(string_col);
    }
    plan->mixed = true;
    plan->distinct_col = -1;
  } else if (n_distinct) {
    // the shape DataFusion rewrites with SingleDistinctToGroupBy: two chained GROUP BYs
    plan->distinct_plan = true;
  }
  plan->order_by = d->order_by;
  plan->descending = d->descending != 0;
  plan->limit = d->limit;
  plan->groups_hint = d->groups_hint;

  // ---- output schemas ----
  auto key_field = [&](const KeyPlan& k) { return named(k.name, plan->fields[k.column].format, true); };
  for (auto& k : plan->keys) { plan->final_fields.push_back(key_field(k)); plan->partial_fields.push_back(key_field(k)); }
  if (plan->distinct_plan) plan->partial_fields.push_back(named("alias1", plan->fields[plan->distinct_col].format, true));
  for (auto& a : plan->aggs) {
    const FieldType* f = a.column >= 0 ? &plan->fields[a.column] : nullptr;
    switch (a.func) {
      case PRISM_AGG_COUNT_STAR:
      case PRISM_AGG_COUNT:
        plan->final_fields.push_back(named(a.name, "l", false));
        plan->partial_fields.push_back(named(a.name + "[count]", "l", false));
        break;
      case PRISM_AGG_COUNT_DISTINCT:
        plan->final_fields.push_back(named(a.name, "l", false));
        // next to other aggregates the state is the list of the group's distinct values (what
        // DataFusion's distinct-count accumulator keeps); alone, the rows of the inner GROUP BY
        if (plan->mixed) plan->partial_fields.push_back(list_field(a.name + "[count distinct]", f->format));
        break;
      case PRISM_AGG_SUM: {
        const char* fmt = f->is_unsigned ? "L" : "l";
        plan->final_fields.push_back(named(a.name, fmt, true));
        plan->partial_fields.push_back(named(a.name + "[sum]", fmt, true));
        break;
      }
      case PRISM_AGG_MIN:
      case PRISM_AGG_MAX:
        plan->final_fields.push_back(named(a.name, f->format, true));
        plan->partial_fields.push_back(named(a.name + (a.func == PRISM_AGG_MIN ? "[min]" : "[max]"), f->format, true));
        break;
      case PRISM_AGG_AVG:
        plan->final_fields.push_back(named(a.name, "g", true));
        plan->partial_fields.push_back(named(a.name + "[count]", "L", false));
        plan->partial_fields.push_back(named(a.name + "[sum]", "g", true));
        break;
    }
  }
  if (plan->order_by >= (int)plan->final_fields.size()) fail(PRISM_GPU_EINVAL, "order_by column out of range");
  return plan;
}

namespace {

struct Interval {
  int64_t lo, hi;  // inclusive; lo > hi: empty
};
constexpr int64_t kI64Lo = -0x7fffffffffffffffll - 1, kI64Hi = 0x7fffffffffffffffll;

// Values of `column` for which node `i` can be TRUE.
Interval range_of(const prism_gpu_plan* p, int i, int column) {
  const PredNode& n = p->pred[i];
  switch (n.kind) {
    case PredNode::AND: {
      const Interval a = range_of(p, n.lhs, column), b = range_of(p, n.rhs, column);
      return {std::max(a.lo, b.lo), std::min(a.hi, b.hi)};
    }
    case PredNode::OR: {
      const Interval a = range_of(p, n.lhs, column), b = range_of(p, n.rhs, column);
      if (a.lo > a.hi) return b;
      if (b.lo > b.hi) return a;
      return {std::min(a.lo, b.lo), std::max(a.hi, b.hi)};
    }
    case PredNode::CMP: {
      if (n.column != column || p->fields[column].is_string || p->fields[column].is_float) return {kI64Lo, kI64Hi};
      const int64_t v = n.lit_i64;
      switch (n.cmp_mask) {
        case kEq: return {v, v};
        case kLt: return v == kI64Lo ? Interval{0, -1} : Interval{kI64Lo, v - 1};
        case kLt | kEq: return {kI64Lo, v};
        case kGt: return v == kI64Hi ? Interval{0, -1} : Interval{v + 1, kI64Hi};
        case kGt | kEq: return {v, kI64Hi};
        case 0: return {0, -1};
        default: return {kI64Lo, kI64Hi};  // <>, or always true
      }
    }
    case PredNode::IS_NULL:
      // a NULL has no value inside any range: partitions are not pruned for it
      return {kI64Lo, kI64Hi};
    default: return {kI64Lo, kI64Hi};
  }
}

}  // namespace

void plan_value_range(const prism_gpu_plan* plan, int column, int64_t* lo, int64_t* hi) {
  if (column < 0 || (size_t)column >= plan->fields.size()) fail(PRISM_GPU_EINVAL, "column out of range");
  const FieldType& f = plan->fields[column];
  if (f.is_string || f.is_float) fail(PRISM_GPU_EINVAL, "a value range exists for integer and timestamp columns only");
  Interval r{kI64Lo, kI64Hi};
  if (plan->pred_root >= 0) r = range_of(plan, plan->pred_root, column);
  *lo = r.lo;
  *hi = r.hi;
}

}  // namespace prism
