// The executor: predicate -> selection -> aggregation -> result batch, over the
// resident chunks of a table.  One call corresponds to what DataFusion does for
// FilterExec + AggregateExec(Partial) [+ AggregateExec(Final) + SortExec(fetch)]
// after prism's TableProvider::scan
// (src/prism-query/src/providers/table_provider.rs:73-132) has produced the scan.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "engine.hpp"

namespace prism {

namespace {

// ---- scan set ------------------------------------------------------------------

struct ColUse {
  int column;
  bool codes;  // dictionary codes of a string column rather than its raw form
};

std::shared_ptr<void> lease_pinned(prism_gpu_ctx* ctx, size_t bytes);

struct ScanSet {
  prism_gpu_ctx* ctx = nullptr;
  std::vector<const Chunk*> chunks;
  std::vector<ScanSeg> segs;
  uint64_t n_tiles = 0, rows = 0;
  std::vector<ColUse> uses;
  const prism_gpu_col_stats* stats_override = nullptr;  // agreed between ranks: replaces the chunks' statistics
  Scratch d_segs, d_cols;
  ScanDesc desc{};
  // The descriptors leave for the device from a page-locked block, in one copy that nobody waits
  // for (from the vectors it was two copies and a wait for them: 40 us in front of every query's
  // first kernel).  The block must outlive the copy: whoever drops the scan set early -- an error
  // unwinding -- waits for the stream first.
  std::shared_ptr<void> staged;
  ~ScanSet() {
    if (staged && ctx) (void)hipStreamSynchronize(ctx->stream);
  }

  bool uploaded = false;
  int use(int column, bool codes) {
    for (size_t i = 0; i < uses.size(); ++i)
      if (uses[i].column == column && uses[i].codes == codes) return (int)i;
    if (uploaded) fail(PRISM_GPU_EINTERNAL, "column requested after the scan descriptor was uploaded");
    uses.push_back({column, codes});
    return (int)uses.size() - 1;
  }
  ColPtr* cols_dev = nullptr;
  const ColPtr* col(int slot) const { return cols_dev + (size_t)slot * segs.size(); }

  // `arena` (with `arena_bytes` of room): the descriptors go there instead of into scratch
  // blocks of the pool (the entry point that may not allocate); returns the bytes used.
  size_t upload(void* arena = nullptr, size_t arena_bytes = 0) {
    uploaded = true;
    if (segs.empty()) return 0;
    const size_t seg_bytes = round_up(segs.size() * sizeof(ScanSeg), 256), col_bytes = uses.size() * segs.size() * sizeof(ColPtr);
    void* segs_dst;
    if (arena) {
      if (seg_bytes + col_bytes > arena_bytes) fail(PRISM_GPU_EUNSUPPORTED, "too many partitions for the allocation-free entry point");
      segs_dst = arena;
    } else {
      d_segs = Scratch(ctx, seg_bytes + col_bytes + 256);
      segs_dst = d_segs.p;
    }
    std::vector<ColPtr> cp(uses.size() * segs.size());
    for (size_t u = 0; u < uses.size(); ++u)
      for (size_t s = 0; s < segs.size(); ++s) {
        const DeviceColumn& dc = chunks[s]->cols[uses[u].column];
        ColPtr& c = cp[u * segs.size() + s];
        c.validity = dc.validity;
        if (uses[u].codes) { c.values = dc.codes; c.bytes = nullptr; }
        else { c.values = dc.values; c.bytes = dc.bytes; }
        if (!c.values) fail(PRISM_GPU_EINTERNAL, "column representation missing on the device");
      }
    cols_dev = cp.empty() ? nullptr : (ColPtr*)((uint8_t*)segs_dst + seg_bytes);
    if (arena) {
      // (the entry point that may not allocate: straight from the vectors, and waited for)
      PRISM_HIP(hipMemcpyAsync(segs_dst, segs.data(), segs.size() * sizeof(ScanSeg), hipMemcpyHostToDevice, ctx->stream));
      if (!cp.empty()) PRISM_HIP(hipMemcpyAsync(cols_dev, cp.data(), col_bytes, hipMemcpyHostToDevice, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));  // segs' and cp's host memory may go away
    } else {
      staged = lease_pinned(ctx, seg_bytes + col_bytes);
      std::memcpy(staged.get(), segs.data(), segs.size() * sizeof(ScanSeg));
      if (!cp.empty()) std::memcpy((uint8_t*)staged.get() + seg_bytes, cp.data(), col_bytes);
      PRISM_HIP(hipMemcpyAsync(segs_dst, staged.get(), seg_bytes + col_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    desc.segs = (const ScanSeg*)segs_dst;
    desc.n_segs = (uint32_t)segs.size();
    desc.n_tiles = n_tiles;
    return seg_bytes + round_up(col_bytes, 256);
  }
};

struct ColInfo {  // statistics of one column over the scan set
  long long min = 0x7fffffffffffffffll, max = -0x7fffffffffffffffll - 1;
  unsigned long long nulls = 0;
};

ColInfo col_info(const ScanSet& ss, int column) {
  ColInfo ci;
  if (ss.stats_override) {
    ci.min = ss.stats_override[column].min;
    ci.max = ss.stats_override[column].max;
    ci.nulls = ss.stats_override[column].nulls;
    return ci;
  }
  for (const Chunk* ch : ss.chunks) {
    const ColStats& s = ch->stats[column];
    ci.min = std::min<long long>(ci.min, s.min);
    ci.max = std::max<long long>(ci.max, s.max);
    ci.nulls += s.null_count;
  }
  return ci;
}

// ---- date_bin on the host (same arithmetic as the kernel) ---------------------------

void div_magic(uint64_t d, uint64_t& magic, int& shift) {
  if (d == 1) { magic = 0; shift = -1; return; }
  int fl = 63 - __builtin_clzll(d);
  if ((d & (d - 1)) == 0) { magic = 0; shift = fl - 1; return; }
  unsigned __int128 num = (unsigned __int128)1 << (64 + fl);
  uint64_t proposed = (uint64_t)(num / d), rem = (uint64_t)(num % d);
  proposed += proposed;
  uint64_t twice = rem + rem;
  if (twice >= d || twice < rem) proposed += 1;
  magic = proposed + 1;
  shift = fl;
}

// n / d == (((n - q) >> pre) + q) >> shift with q = mulhi(magic, n), for every 32-bit n.
void div_magic32(uint32_t d, uint32_t& magic, int& pre, int& shift) {
  pre = 1;
  if (d == 1) { magic = 0; pre = 0; shift = 0; return; }
  int fl = 31 - __builtin_clz(d);
  if ((d & (d - 1)) == 0) { magic = 0; shift = fl - 1; return; }
  uint64_t num = 1ull << (32 + fl);
  uint32_t proposed = (uint32_t)(num / d), rem = (uint32_t)(num % d);
  proposed += proposed;
  uint32_t twice = rem + rem;
  if (twice >= d || twice < rem) proposed += 1;
  magic = proposed + 1;
  shift = fl;
}

long long bucket_number_host(long long ts, long long stride, long long origin) {
  long long d = (long long)((uint64_t)ts - (uint64_t)origin);
  uint64_t a = d < 0 ? (0ull - (uint64_t)d) : (uint64_t)d;
  long long q = (long long)(a / (uint64_t)stride);
  if (d < 0) q = -q - (stride > 1 ? 1 : 0);
  return q;
}

// ---- lowering of keys and aggregates ---------------------------------------------------

struct KeySrc {
  int column;
  bool date_bin;
  int64_t stride, origin;
};

struct KeyField {  // how to unpack one key from the packed key
  int word = 0, bit_pos = 0, bits = 0, null_bit = -1;
  long long bias = 0;
};

struct KeyLayout {
  std::vector<KeySpec> specs;
  std::vector<KeyField> fields;
  int total_bits = 0;   // bits in use when the key is one word
  int words = 1;        // 1 or 2
  int word_bits[2] = {0, 0};
};

// Where the next key field goes: the packed key is one or two 64-bit words and a field
// never straddles them.
struct KeyCursor {
  int bit[2] = {0, 0};   // bits used in each word
  bool closed0 = false;  // nothing more may go into word 0
};

// Packs the given key sources from `cur` on.
void layout_keys(ScanSet& ss, const prism_gpu_table* t, const std::vector<KeySrc>& keys, KeyCursor& cur, KeyLayout& L) {
  // Bits are handed out to the other keys first and to time keys (date_bin, timestamps)
  // last, so that time ends up in the high bits of the packed key: over time-ordered input
  // the packed keys of any short stretch of rows then fall in a narrow range, which the
  // streaming kernel's window mode indexes directly.  specs / fields stay in key order.
  auto is_time = [&](const KeySrc& k) { return k.date_bin || t->fields[k.column].logical == PRISM_COL_TIMESTAMP; };
  std::vector<size_t> order;
  for (size_t i = 0; i < keys.size(); ++i) if (!is_time(keys[i])) order.push_back(i);
  for (size_t i = 0; i < keys.size(); ++i) if (is_time(keys[i])) order.push_back(i);
  const size_t first_out = L.specs.size();
  L.specs.resize(first_out + keys.size());
  L.fields.resize(first_out + keys.size());
  for (size_t ki : order) {
    const KeySrc& k = keys[ki];
    const FieldType& f = t->fields[k.column];
    if (f.is_string && !t->dicts[k.column].alive && !t->dicts[k.column].device_built)
      fail(PRISM_GPU_EUNSUPPORTED, "GROUP BY / DISTINCT on string column `" + f.name + "` with more than " +
                                       std::to_string(kMaxDictValues) + " distinct values is not supported");
    ColInfo ci = col_info(ss, k.column);
    long long vmin = 0, vmax = 0;
    if (ci.min <= ci.max) {
      vmin = ci.min;
      vmax = ci.max;
      if (k.date_bin) { vmin = bucket_number_host(ci.min, k.stride, k.origin); vmax = bucket_number_host(ci.max, k.stride, k.origin); }
    }
    KeySpec s{};
    s.col = ss.use(k.column, f.is_string);
    s.width = f.is_string ? 4 : f.width;
    s.is_unsigned = (f.is_string || f.is_unsigned) ? 1 : 0;
    s.date_bin = k.date_bin ? 1 : 0;
    s.stride = k.stride;
    s.origin = k.origin;
    if (k.date_bin) div_magic((uint64_t)k.stride, s.magic, s.magic_shift);
    if (k.date_bin && ci.min <= ci.max && ci.min >= k.origin && (uint64_t)k.stride < (1ull << 32)) {
      // every timestamp is at or after the origin: bucket numbers are plain floors, and
      // relative to the first bucket's start they may fit 32-bit arithmetic
      long long base = (long long)((uint64_t)k.origin + (uint64_t)vmin * (uint64_t)k.stride);
      if ((uint64_t)ci.max - (uint64_t)base < (1ull << 32)) {
        s.fast32 = 1;
        s.base = base;
        div_magic32((uint32_t)k.stride, s.magic32, s.pre32, s.shift32);
      }
    }
    if (!k.date_bin && (uint64_t)vmax - (uint64_t)vmin < (1ull << 31)) {
      s.fast32 = 1;  // plain column: divisor 1
      s.base = vmin;
      div_magic32(1, s.magic32, s.pre32, s.shift32);
    }
    KeyField kf;
    kf.bits = bits_for((uint64_t)vmax - (uint64_t)vmin);
    kf.bias = vmin;
    const int need = kf.bits + (ci.nulls ? 1 : 0);
    // first fit: word 0 while the field fits there (64 bits when it is the only word in
    // use, 63 once a second word exists), else word 1
    int w;
    if (!cur.closed0 && cur.bit[0] + need <= (cur.bit[1] ? 63 : 64)) w = 0;
    else if (cur.bit[1] + need <= 63 && cur.bit[0] <= 63) w = 1;
    else fail(PRISM_GPU_EUNSUPPORTED, "GROUP BY key does not fit two 63-bit words");
    kf.word = w;
    kf.bit_pos = kf.bits ? cur.bit[w] : 0;
    cur.bit[w] += kf.bits;
    if (ci.nulls) kf.null_bit = cur.bit[w]++;
    s.bias = kf.bias;
    s.bit_pos = kf.bit_pos;
    s.null_bit = kf.null_bit;
    s.word = kf.word;
    L.specs[first_out + ki] = s;
    L.fields[first_out + ki] = kf;
  }
  L.word_bits[0] = cur.bit[0];
  L.word_bits[1] = cur.bit[1];
  L.words = (cur.bit[1] > 0 || cur.closed0) ? 2 : 1;
  L.total_bits = L.words == 1 ? cur.bit[0] : 128;
}

// Bits the keys would take in one word (fields plus NULL bits).
int key_bits_needed(ScanSet& ss, const prism_gpu_table* t, const std::vector<KeySrc>& keys) {
  int bits = 0;
  for (const KeySrc& k : keys) {
    ColInfo ci = col_info(ss, k.column);
    long long vmin = 0, vmax = 0;
    if (ci.min <= ci.max) {
      vmin = ci.min;
      vmax = ci.max;
      if (k.date_bin) { vmin = bucket_number_host(ci.min, k.stride, k.origin); vmax = bucket_number_host(ci.max, k.stride, k.origin); }
    }
    bits += bits_for((uint64_t)vmax - (uint64_t)vmin) + (ci.nulls ? 1 : 0);
  }
  return bits;
}

struct SlotSrc {
  int op;
  int column;  // table column, -1 for kSlotAddOne
};

struct AggRule {  // how an output aggregate is read from the slots
  int func;
  int main_slot = -1;   // value
  int count_slot = -1;  // > 0 <=> value is not NULL (-1: never NULL); AVG: the divisor
};

struct Lowered {
  std::vector<SlotSrc> slots;
  std::vector<AggRule> rules;
  int slot(int op, int column) {
    for (size_t i = 0; i < slots.size(); ++i)
      if (slots[i].op == op && slots[i].column == column) return (int)i;
    slots.push_back({op, column});
    return (int)slots.size() - 1;
  }
};

// Raw rows -> slots.
Lowered lower_raw(const prism_gpu_plan* p, ScanSet& ss) {
  Lowered L;
  L.slot(kSlotAddOne, -1);  // slot 0: rows per group
  for (const AggPlan& a : p->aggs) {
    AggRule r;
    r.func = a.func;
    auto valid_count = [&](int column) {
      // a column without NULLs in the scanned chunks needs no counter of its own
      return col_info(ss, column).nulls ? L.slot(kSlotAddValid, column) : 0;
    };
    switch (a.func) {
      case PRISM_AGG_COUNT_STAR: r.main_slot = 0; break;
      case PRISM_AGG_COUNT: r.main_slot = valid_count(a.column); break;
      case PRISM_AGG_SUM: r.main_slot = L.slot(kSlotAddI64, a.column); r.count_slot = valid_count(a.column); break;
      case PRISM_AGG_MIN: r.main_slot = L.slot(kSlotMin, a.column); r.count_slot = valid_count(a.column); break;
      case PRISM_AGG_MAX: r.main_slot = L.slot(kSlotMax, a.column); r.count_slot = valid_count(a.column); break;
      case PRISM_AGG_AVG: r.main_slot = L.slot(kSlotAddF64, a.column); r.count_slot = valid_count(a.column); break;
      default: fail(PRISM_GPU_EINTERNAL, "aggregate cannot be lowered");
    }
    L.rules.push_back(r);
  }
  return L;
}

// PARTIAL state columns -> slots (AggregateExec(Final)'s merge_batch).
Lowered lower_merge(const prism_gpu_plan* p) {
  Lowered L;
  L.slot(kSlotAddOne, -1);
  int c = (int)p->keys.size();
  for (const AggPlan& a : p->aggs) {
    AggRule r;
    r.func = a.func;
    switch (a.func) {
      case PRISM_AGG_COUNT_STAR:
      case PRISM_AGG_COUNT: r.main_slot = L.slot(kSlotAddI64, c); c += 1; break;
      case PRISM_AGG_SUM: r.main_slot = L.slot(kSlotAddI64, c); r.count_slot = L.slot(kSlotAddValid, c); c += 1; break;
      case PRISM_AGG_MIN: r.main_slot = L.slot(kSlotMin, c); r.count_slot = L.slot(kSlotAddValid, c); c += 1; break;
      case PRISM_AGG_MAX: r.main_slot = L.slot(kSlotMax, c); r.count_slot = L.slot(kSlotAddValid, c); c += 1; break;
      case PRISM_AGG_AVG: r.count_slot = L.slot(kSlotAddI64, c); r.main_slot = L.slot(kSlotAddF64, c + 1); c += 2; break;
      default: fail(PRISM_GPU_EINTERNAL, "aggregate cannot be merged");
    }
    L.rules.push_back(r);
  }
  return L;
}

void fill_slots(const prism_gpu_table* t, ScanSet& ss, const Lowered& L, AggParams& ap) {
  if (L.slots.size() > (size_t)kMaxSlots) fail(PRISM_GPU_EUNSUPPORTED, "too many aggregates in one plan");
  std::vector<int> val_cols;
  ap.n_slots = (int)L.slots.size();
  for (size_t i = 0; i < L.slots.size(); ++i) {
    ap.slots[i].op = L.slots[i].op;
    ap.slots[i].val = -1;
    if (L.slots[i].column < 0) continue;
    auto it = std::find(val_cols.begin(), val_cols.end(), L.slots[i].column);
    if (it == val_cols.end()) { val_cols.push_back(L.slots[i].column); it = val_cols.end() - 1; }
    ap.slots[i].val = (int)(it - val_cols.begin());
  }
  if (val_cols.size() > (size_t)kMaxVals) fail(PRISM_GPU_EUNSUPPORTED, "too many aggregate input columns in one plan");
  ap.n_vals = (int)val_cols.size();
  for (size_t j = 0; j < val_cols.size(); ++j) {
    const FieldType& f = t->fields[val_cols[j]];
    ValSpec& v = ap.vals[j];
    // COUNT(x) over a string column only needs its validity: either representation will do
    v.col = ss.use(val_cols[j], f.is_string && t->dicts[val_cols[j]].alive);
    v.width = f.is_string ? 4 : f.width;
    v.is_unsigned = (f.is_string || f.is_unsigned) ? 1 : 0;
    v.is_float = f.is_float ? 1 : 0;
  }
}

long long slot_identity_host(int op) {
  return op == kSlotMin ? 0x7fffffffffffffffll : (op == kSlotMax ? -0x7fffffffffffffffll - 1 : 0ll);
}

// ---- predicate evaluation -------------------------------------------------------------

// Is node `i` a comparison of a fixed-width column, or the AND of two such comparisons of
// one column (a range)?  Those are one pass of the comparison kernel.
bool one_column_range(const prism_gpu_plan* p, const prism_gpu_table* t, int i, const PredNode** a, const PredNode** b) {
  const PredNode& n = p->pred[i];
  auto plain_cmp = [&](const PredNode& x) {
    return x.kind == PredNode::CMP && !t->fields[x.column].is_string && !t->fields[x.column].is_float;
  };
  if (plain_cmp(n)) { *a = &n; *b = nullptr; return true; }
  if (n.kind != PredNode::AND) return false;
  const PredNode &l = p->pred[n.lhs], &r = p->pred[n.rhs];
  if (!plain_cmp(l) || !plain_cmp(r) || l.column != r.column) return false;
  *a = &l;
  *b = &r;
  return true;
}

// Which representation of its column a predicate leaf reads: dictionary codes or
// the raw values / offsets + bytes.
bool leaf_uses_codes(const prism_gpu_table* t, const ScanSet& ss, const PredNode& n) {
  const FieldType& f = t->fields[n.column];
  if (!f.is_string) return false;
  if (n.kind == PredNode::IS_NULL || n.kind == PredNode::IS_NOT_NULL) return ss.chunks[0]->cols[n.column].values == nullptr;
  return t->dicts[n.column].alive;
}

// A string predicate (comparison or LIKE) decided for every value of the column's
// dictionary: lut[code] = 1 when a row with that code passes (one spare entry at the end).
std::vector<uint8_t> dict_lut(const PredNode& n, const ColumnDict& dict) {
  std::vector<uint8_t> lut(dict.values.size() + 1, 0);
  std::vector<LikeToken> toks;
  if (n.kind == PredNode::LIKE) toks = like_compile(n.lit_str);
  for (size_t i = 0; i < dict.values.size(); ++i) {
    const std::string& v = dict.values[i];
    bool t;
    if (n.kind == PredNode::LIKE) {
      t = like_match_host(toks, v, n.case_insensitive) != n.negated;
    } else {
      int c = std::memcmp(v.data(), n.lit_str.data(), std::min(v.size(), n.lit_str.size()));
      if (c == 0) c = v.size() < n.lit_str.size() ? -1 : (v.size() > n.lit_str.size() ? 1 : 0);
      t = (c < 0 && (n.cmp_mask & kLt)) || (c == 0 && (n.cmp_mask & kEq)) || (c > 0 && (n.cmp_mask & kGt));
    }
    lut[i] = t ? 1 : 0;
  }
  return lut;
}

struct Selection {
  bool all = true;      // no predicate: every row
  Scratch bits;         // TRUE rows, one bit per padded row id
  uint64_t count = 0;
};

struct PredEval {
  prism_gpu_plan* plan;
  prism_gpu_table* table;
  ScanSet& ss;
  prism_gpu_ctx* ctx;
  uint64_t n_words;
  std::vector<Scratch> keep;  // literals etc. that must outlive the launches

  Scratch new_bits(bool zero) {
    Scratch b(ctx, n_words * 8);
    if (zero) PRISM_HIP(hipMemsetAsync(b.p, 0, n_words * 8, ctx->stream));
    return b;
  }
  const uint8_t* to_device(const void* src, size_t n) {
    keep.emplace_back(ctx, n + 16);
    PRISM_HIP(hipMemcpyAsync(keep.back().p, src, n, hipMemcpyHostToDevice, ctx->stream));
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    return keep.back().as<uint8_t>();
  }

  // String predicate through the column's dictionary: evaluated once per distinct
  // value on the host, then a code -> bit lookup on the device.
  Scratch lut_leaf(const PredNode& n, const ColumnDict& dict) {
    std::vector<uint8_t> lut = dict_lut(n, dict);
    const uint8_t* d_lut = to_device(lut.data(), lut.size());
    Scratch bits = new_bits(false);
    KernelTimer timer(ctx, "pred_lut", ss.rows * 4);
    launch_lut_leaf(ctx, ss.desc, ss.col(ss.use(n.column, true)), d_lut, (uint32_t)dict.values.size(), bits.as<uint64_t>());
    return bits;
  }

  Scratch eval(int i) {
    const PredNode& n = plan->pred[i];
    switch (n.kind) {
      case PredNode::AND:
      case PredNode::OR: {
        const PredNode *ra = nullptr, *rb = nullptr;
        if (n.kind == PredNode::AND && one_column_range(plan, table, i, &ra, &rb) && rb) {
          // lo <= x AND x < hi: both comparisons in one pass over the column
          const FieldType& f = table->fields[ra->column];
          Scratch bits = new_bits(false);
          KernelTimer timer(ctx, "pred_cmp", ss.rows * f.width);
          launch_cmp_leaf(ctx, ss.desc, ss.col(ss.use(ra->column, false)), f.width, f.is_unsigned, ra->cmp_mask, ra->lit_i64,
                          bits.as<uint64_t>(), nullptr, rb->cmp_mask, rb->lit_i64);
          return bits;
        }
        Scratch a = eval(n.lhs), b = eval(n.rhs);
        if (n.kind == PredNode::AND) launch_bitmap_and(ctx, a.as<uint64_t>(), b.as<uint64_t>(), a.as<uint64_t>(), n_words);
        else launch_bitmap_or(ctx, a.as<uint64_t>(), b.as<uint64_t>(), a.as<uint64_t>(), n_words);
        // b returns to the pool here; the stream orders its reuse after this launch
        return a;
      }
      case PredNode::IS_NULL:
      case PredNode::IS_NOT_NULL: {
        Scratch bits = new_bits(false);
        launch_null_leaf(ctx, ss.desc, ss.col(ss.use(n.column, leaf_uses_codes(table, ss, n))), n.kind == PredNode::IS_NULL,
                         bits.as<uint64_t>());
        return bits;
      }
      case PredNode::CMP: {
        const FieldType& f = table->fields[n.column];
        if (!f.is_string) {
          Scratch bits = new_bits(false);
          KernelTimer timer(ctx, "pred_cmp", ss.rows * f.width);
          launch_cmp_leaf(ctx, ss.desc, ss.col(ss.use(n.column, false)), f.width, f.is_unsigned, n.cmp_mask, n.lit_i64,
                          bits.as<uint64_t>(), nullptr);
          return bits;
        }
        const ColumnDict& dict = table->dicts[n.column];
        if (dict.alive) return lut_leaf(n, dict);
        const uint8_t* lit = to_device(n.lit_str.data(), n.lit_str.size());
        Scratch bits = new_bits(false);
        KernelTimer timer(ctx, "pred_str", 0);
        launch_str_leaf(ctx, ss.desc, ss.col(ss.use(n.column, false)), kStrCmp, n.cmp_mask, lit, (uint32_t)n.lit_str.size(),
                        nullptr, 0, false, false, bits.as<uint64_t>());
        return bits;
      }
      case PredNode::LIKE: {
        const ColumnDict& dict = table->dicts[n.column];
        if (dict.alive) return lut_leaf(n, dict);
        if (ss.chunks[0]->cols[n.column].values == nullptr)
          fail(PRISM_GPU_EUNSUPPORTED, "LIKE on a generated dictionary column without a dictionary");
        std::vector<LikeToken> toks = like_compile(n.lit_str);
        int col = ss.use(n.column, false);
        // '%needle%': any run, literal bytes, any run
        bool contains = toks.size() >= 3 && toks.front().kind == 2 && toks.back().kind == 2 && !n.case_insensitive;
        for (size_t k = 1; contains && k + 1 < toks.size(); ++k) contains = toks[k].kind == 0;
        uint64_t str_bytes = 0;
        for (const Chunk* ch : ss.chunks) str_bytes += ch->cols[n.column].n_bytes;
        if (contains) {
          std::string needle;
          for (size_t k = 1; k + 1 < toks.size(); ++k) needle.push_back((char)toks[k].byte);
          const uint8_t* d_needle = to_device(needle.data(), needle.size());
          Scratch bits = new_bits(true);
          // all segments in one streaming launch; should the needle occur more often than
          // there are rows (a one-letter needle, say), segment by segment with the search
          // for the row done in place
          std::vector<ContainsSeg> csegs(ss.segs.size());
          const uint64_t chunk_bytes = contains_chunk_bytes(contains_block(ctx, (uint32_t)needle.size()));
          uint64_t n_chunks = 0;
          for (size_t s = 0; s < ss.segs.size(); ++s) {
            const DeviceColumn& dc = ss.chunks[s]->cols[n.column];
            csegs[s] = {(const int32_t*)dc.values, dc.bytes, ss.segs[s].rows, dc.n_bytes, ss.segs[s].tile_base, n_chunks};
            n_chunks += ceil_div(dc.n_bytes, chunk_bytes);
          }
          const ContainsSeg* d_csegs = (const ContainsSeg*)to_device(csegs.data(), csegs.size() * sizeof(ContainsSeg));
          const uint64_t match_capacity = std::max<uint64_t>(ss.rows, 1u << 16);
          Scratch d_matches(ctx, match_capacity * 8), d_n_matches(ctx, 256);
          bool done;
          {
            KernelTimer timer(ctx, "like_contains", str_bytes + ss.rows * 4);
            done = launch_contains_all(ctx, d_csegs, (uint32_t)csegs.size(), n_chunks, d_needle, (uint32_t)needle.size(),
                                       d_matches.as<uint64_t>(), match_capacity, d_n_matches.as<unsigned long long>(),
                                       bits.as<uint64_t>());
          }
          if (!done) {
            KernelTimer timer(ctx, "like_contains_dense", str_bytes + ss.rows * 4);
            for (size_t s = 0; s < ss.segs.size(); ++s) {
              const DeviceColumn& dc = ss.chunks[s]->cols[n.column];
              ColPtr cp{dc.values, dc.validity, dc.bytes};
              launch_contains_stream(ctx, cp, ss.segs[s].rows, ss.segs[s].tile_base, d_needle, (uint32_t)needle.size(),
                                     bits.as<uint64_t>());
            }
          }
          // NULL rows are not TRUE, and NOT LIKE flips the rest: nothing to do for a plain LIKE
          // over chunks without a validity bitmap
          bool any_validity = false;
          for (const Chunk* ch : ss.chunks) any_validity = any_validity || ch->cols[n.column].validity != nullptr;
          if (n.negated || any_validity) launch_mask_validity(ctx, ss.desc, ss.col(col), n.negated, bits.as<uint64_t>());
          return bits;
        }
        const uint8_t* d_toks = to_device(toks.data(), toks.size() * sizeof(LikeToken));
        Scratch bits = new_bits(false);
        KernelTimer timer(ctx, "pred_str", str_bytes + ss.rows * 4);
        launch_str_leaf(ctx, ss.desc, ss.col(col), kStrLike, 0, nullptr, 0, (const LikeToken*)d_toks, (uint32_t)toks.size(),
                        n.negated, n.case_insensitive, bits.as<uint64_t>());
        return bits;
      }
    }
    fail(PRISM_GPU_EINTERNAL, "bad predicate node");
  }
};

// ---- result assembly ---------------------------------------------------------------------

struct GroupRows {
  std::vector<uint64_t> keys;   // packed (low word)
  std::vector<uint64_t> keys_hi;  // high word of two-word keys, else empty
  std::vector<int64_t> state;   // slot-major, n groups each
  uint64_t n = 0;
};

int64_t state_at(const GroupRows& g, int slot, uint64_t i) { return g.state[(size_t)slot * g.n + i]; }

// Appends the key columns of n groups, column by column (tight loops, no per-row
// allocation: string keys stay dictionary codes until the Arrow buffers are written).
void append_key_columns(const prism_gpu_table* t, const std::vector<KeySrc>& keys, const KeyLayout& L, size_t first_key,
                        const GroupRows& g, std::vector<OutColumn>& cols, size_t first_col) {
  const size_t n = (size_t)g.n;
  for (size_t i = 0; i < keys.size(); ++i) {
    const KeyField& kf = L.fields[first_key + i];
    const uint64_t* packed = kf.word ? g.keys_hi.data() : g.keys.data();
    const KeySrc& k = keys[i];
    const FieldType& f = t->fields[k.column];
    OutColumn& oc = cols[first_col + i];
    const size_t base = oc.rows();
    const uint64_t mask = kf.bits == 0 ? 0ull : (kf.bits == 64 ? ~0ull : ((1ull << kf.bits) - 1ull));
    oc.valid.resize(base + n, 1);
    if (kf.null_bit >= 0)
      for (size_t r = 0; r < n; ++r) oc.valid[base + r] = ((packed[r] >> kf.null_bit) & 1ull) ? 0 : 1;
    if (f.is_string) {
      const ColumnDict& d = t->dicts[k.column];
      if (base == 0) oc.dict = &d.values;
      if (oc.dict != &d.values) fail(PRISM_GPU_EINTERNAL, "string key column built from two dictionaries");
      oc.codes.resize(base + n);
      for (size_t r = 0; r < n; ++r) {
        uint64_t code = (uint64_t)kf.bias + ((packed[r] >> kf.bit_pos) & mask);
        if (!oc.valid[base + r]) code = 0;
        if (code >= d.values.size() && oc.valid[base + r]) fail(PRISM_GPU_EINTERNAL, "dictionary code out of range in result");
        oc.codes[base + r] = (uint32_t)code;
      }
    } else {
      oc.i64.resize(base + n);
      for (size_t r = 0; r < n; ++r) {
        long long v = (long long)((uint64_t)kf.bias + ((packed[r] >> kf.bit_pos) & mask));
        if (k.date_bin) v = (long long)((uint64_t)k.origin + (uint64_t)v * (uint64_t)k.stride);
        oc.i64[base + r] = oc.valid[base + r] ? v : 0;
      }
    }
  }
}

std::vector<OutColumn> make_columns(const std::vector<FieldType>& fields) {
  std::vector<OutColumn> cols;
  for (auto& f : fields) cols.push_back(make_out_column(f.name, f.logical == kTypeList ? "+l:" + f.list_item : f.format, f.nullable));
  return cols;
}

// Appends the aggregate columns of all groups of g, column by column.
void append_agg_columns(const Lowered& L, const GroupRows& g, bool partial, std::vector<OutColumn>& cols, size_t c) {
  const size_t n = (size_t)g.n;
  for (const AggRule& r : L.rules) {
    const int64_t* main = g.state.data() + (size_t)r.main_slot * n;
    const int64_t* cnt = r.count_slot >= 0 ? g.state.data() + (size_t)r.count_slot * n : nullptr;
    auto append_ints = [&](OutColumn& oc, const int64_t* v, bool nullable) {
      const size_t base = oc.rows();
      oc.i64.insert(oc.i64.end(), v, v + n);
      oc.valid.resize(base + n, 1);
      if (nullable && cnt)
        for (size_t i = 0; i < n; ++i) oc.valid[base + i] = cnt[i] > 0;
    };
    auto append_floats = [&](OutColumn& oc, bool divide) {
      const size_t base = oc.rows();
      oc.f64.resize(base + n);
      oc.valid.resize(base + n, 1);
      for (size_t i = 0; i < n; ++i) {
        double sum;
        std::memcpy(&sum, &main[i], 8);
        const int64_t k = cnt ? cnt[i] : 1;
        oc.valid[base + i] = k > 0;
        oc.f64[base + i] = k > 0 ? (divide ? sum / (double)k : sum) : 0.0;
      }
    };
    switch (r.func) {
      case PRISM_AGG_COUNT_STAR:
      case PRISM_AGG_COUNT: append_ints(cols[c++], main, false); break;
      case PRISM_AGG_SUM:
      case PRISM_AGG_MIN:
      case PRISM_AGG_MAX: append_ints(cols[c++], main, true); break;
      case PRISM_AGG_AVG:
        if (partial) {
          if (cnt) append_ints(cols[c++], cnt, false);
          else { OutColumn& oc = cols[c++]; oc.i64.resize(oc.i64.size() + n, 1); oc.valid.resize(oc.valid.size() + n, 1); }
          append_floats(cols[c++], false);
        } else {
          append_floats(cols[c++], true);
        }
        break;
    }
  }
}

void sort_and_limit(const prism_gpu_plan* p, std::vector<OutColumn>& cols) {
  if (p->order_by < 0 && p->limit < 0) return;
  for (auto& c : cols) c.materialise();
  size_t n = cols.empty() ? 0 : cols[0].rows();
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  if (p->order_by >= 0) {
    const OutColumn& oc = cols[p->order_by];
    const bool desc = p->descending, uns = oc.type.is_unsigned && !oc.type.is_string;
    auto less = [&](size_t a, size_t b) {
      bool na = !oc.valid[a], nb = !oc.valid[b];
      if (na || nb) return desc ? (na && !nb) : (!na && nb);  // NULLS FIRST for DESC, LAST for ASC
      int c;
      if (oc.type.is_string) c = oc.str_at(a).compare(oc.str_at(b));
      else if (oc.type.is_float) c = oc.f64[a] < oc.f64[b] ? -1 : (oc.f64[a] > oc.f64[b] ? 1 : 0);
      else if (uns) c = (uint64_t)oc.i64[a] < (uint64_t)oc.i64[b] ? -1 : ((uint64_t)oc.i64[a] > (uint64_t)oc.i64[b] ? 1 : 0);
      else c = oc.i64[a] < oc.i64[b] ? -1 : (oc.i64[a] > oc.i64[b] ? 1 : 0);
      return desc ? c > 0 : c < 0;
    };
    if (p->limit >= 0 && (size_t)p->limit < n) std::partial_sort(idx.begin(), idx.begin() + p->limit, idx.end(), less);
    else std::stable_sort(idx.begin(), idx.end(), less);
  }
  if (p->limit >= 0 && (size_t)p->limit < idx.size()) idx.resize((size_t)p->limit);
  for (auto& c : cols) c.gather(idx);
}

// ---- HBM group tables --------------------------------------------------------------------------

constexpr int kDenseMaxBits = 22;
constexpr int kDenseMaxBitsLarge = 28;

struct GroupTable {
  bool dense = false;
  // A list rather than a hash table: the arrays are laid out like a hash table's, but the
  // groups sit in the leading entries in no particular order (the partitioned GROUP BY's
  // output).  Everything that walks the entries works as is; nothing may be inserted by hashing.
  bool list = false;
  // no group of the table has seen more rows than this (0: unknown); lets ORDER BY COUNT(..)
  // start its radix select below the bits no count can have
  uint64_t rows_bound = 0;
  uint64_t capacity = 0;
  int n_slots = 0;
  int key_words = 1;
  Scratch keys, keys_hi, state, overflow;
};

void table_alloc(prism_gpu_ctx* ctx, GroupTable& g, bool dense, uint64_t capacity, const std::vector<int>& slot_ops,
                 int key_words = 1) {
  g.dense = dense;
  g.key_words = key_words;
  g.keys_hi = Scratch();
  if (key_words == 2) g.keys_hi = Scratch(ctx, (capacity + 1) * 8);
  g.capacity = capacity;
  g.n_slots = (int)slot_ops.size();
  g.state = Scratch(ctx, (capacity + 1) * 8 * slot_ops.size());
  g.overflow = Scratch(ctx, 256);
  if (!dense) g.keys = Scratch(ctx, (capacity + 1) * 8);
  // (one launch for the whole table: a fill per array was five launches in front of every query's kernel)
  TableInit t{};
  t.keys = dense ? nullptr : g.keys.as<unsigned long long>();
  t.keys_hi = key_words == 2 ? g.keys_hi.as<unsigned long long>() : nullptr;
  t.state = g.state.as<long long>();
  t.entries = capacity + 1;
  t.n_slots = (int)slot_ops.size();
  if (slot_ops.size() > (size_t)kMaxSlots) fail(PRISM_GPU_EINTERNAL, "too many accumulators in one group table");
  for (size_t s = 0; s < slot_ops.size(); ++s) t.identity[s] = slot_identity_host(slot_ops[s]);
  t.flag = g.overflow.as<uint32_t>();
  launch_table_init(ctx, t);
}

// The table's flag word: 0, 1 (an insertion found the table full), 2 (a merged key fell
// outside a dense table) or 3 (the streaming kernel stopped because its rows scatter).
// (Words read back from the device land in a page-locked block of the pool: a copy into pageable
// memory -- a variable on the stack -- goes through the runtime's own staging first, some tens
// of microseconds that a query of a few million rows notices.)
uint32_t table_flag(prism_gpu_ctx* ctx, const GroupTable& g) {
  std::shared_ptr<void> landing = lease_pinned(ctx, 64);
  PRISM_HIP(hipMemcpyAsync(landing.get(), g.overflow.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  return *(const volatile uint32_t*)landing.get();
}

bool table_overflowed(prism_gpu_ctx* ctx, const GroupTable& g) { return table_flag(ctx, g) != 0; }

// Reads the occupied entries of a table; with `order`, only those whose rank reaches the
// threshold launch_topk_threshold left in d_threshold.
GroupRows table_read(prism_gpu_ctx* ctx, const GroupTable& g, const TopKOrder* order = nullptr,
                     TopKState* d_threshold = nullptr) {
  GroupRows rows;
  Scratch d_count(ctx, 256);
  PRISM_HIP(hipMemsetAsync(d_count.p, 0, 8, ctx->stream));
  launch_emit(ctx, g.dense, g.capacity, g.keys.as<uint64_t>(), g.keys_hi.as<uint64_t>(), g.state.as<int64_t>(),
              g.n_slots, nullptr, nullptr, nullptr, 0, d_count.as<unsigned long long>(), order, d_threshold);
  std::shared_ptr<void> n_landing = lease_pinned(ctx, 64);  // (page-locked: see table_flag)
  PRISM_HIP(hipMemcpyAsync(n_landing.get(), d_count.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  const unsigned long long n = *(const volatile unsigned long long*)n_landing.get();
  rows.n = n;
  if (n == 0) return rows;
  Scratch d_keys(ctx, n * 8), d_state(ctx, n * 8 * g.n_slots), d_keys_hi;
  if (g.key_words == 2) d_keys_hi = Scratch(ctx, n * 8);
  PRISM_HIP(hipMemsetAsync(d_count.p, 0, 8, ctx->stream));
  launch_emit(ctx, g.dense, g.capacity, g.keys.as<uint64_t>(), g.keys_hi.as<uint64_t>(), g.state.as<int64_t>(),
              g.n_slots, d_keys.as<uint64_t>(), d_keys_hi.as<uint64_t>(), d_state.as<int64_t>(), n,
              d_count.as<unsigned long long>(), order, d_threshold);
  if (g.key_words == 2) {
    rows.keys_hi.resize(n);
    PRISM_HIP(hipMemcpyAsync(rows.keys_hi.data(), d_keys_hi.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  rows.keys.resize(n);
  rows.state.resize((size_t)n * g.n_slots);
  // HBM -> pinned host memory at full PCIe rate, then a plain copy into the vectors
  const size_t key_bytes = n * 8, state_bytes = n * 8 * (size_t)g.n_slots;
  if (key_bytes + state_bytes <= (256ull << 20)) {
    std::lock_guard<std::mutex> lock(ctx->result_mu);
    if (ctx->result_pinned_bytes < key_bytes + state_bytes) {
      if (ctx->result_pinned) (void)hipHostFree(ctx->result_pinned);
      ctx->result_pinned = nullptr;
      ctx->result_pinned_bytes = 0;
      size_t want = pow2_ceil(key_bytes + state_bytes);
      PRISM_HIP(hipHostMalloc(&ctx->result_pinned, want, hipHostMallocDefault));
      ctx->result_pinned_bytes = want;
    }
    uint8_t* pin = (uint8_t*)ctx->result_pinned;
    PRISM_HIP(hipMemcpyAsync(pin, d_keys.p, key_bytes, hipMemcpyDeviceToHost, ctx->stream));
    PRISM_HIP(hipMemcpyAsync(pin + key_bytes, d_state.p, state_bytes, hipMemcpyDeviceToHost, ctx->stream));
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(rows.keys.data(), pin, key_bytes);
    std::memcpy(rows.state.data(), pin + key_bytes, state_bytes);
  } else {
    PRISM_HIP(hipMemcpyAsync(rows.keys.data(), d_keys.p, key_bytes, hipMemcpyDeviceToHost, ctx->stream));
    PRISM_HIP(hipMemcpyAsync(rows.state.data(), d_state.p, state_bytes, hipMemcpyDeviceToHost, ctx->stream));
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
  }
  return rows;
}

// Tables with fewer entries than this are read whole; sorting them on the host is cheap.
constexpr uint64_t kTopKMinEntries = 4096;

// Whether ORDER BY ... LIMIT of this plan can be decided on the device, and how.
bool topk_order(const prism_gpu_plan* p, const prism_gpu_table* t, const std::vector<KeySrc>& keys, const KeyLayout& layout,
                const Lowered& low, const std::vector<OutColumn>& cols, TopKOrder& o, uint64_t rows_bound = 0) {
  if (p->order_by < 0 || p->limit < 0 || p->limit > (1 << 20)) return false;
  std::memset(&o, 0, sizeof(o));
  o.descending = p->descending ? 1 : 0;
  const size_t c = (size_t)p->order_by;
  if (c < keys.size()) {
    if (t->fields[keys[c].column].is_string) return false;  // dictionary codes do not sort like their strings
    const KeyField& f = layout.fields[c];
    o.kind = kOrdKeyField;
    o.word = f.word;
    o.bit_pos = f.bit_pos;
    o.bits = f.bits;
    o.null_bit = f.null_bit;
    return true;
  }
  if (c - keys.size() >= low.rules.size()) return false;
  const AggRule& r = low.rules[c - keys.size()];
  o.main_slot = r.main_slot;
  o.count_slot = r.count_slot;
  o.is_unsigned = cols[c].type.is_unsigned ? 1 : 0;
  if (r.func == PRISM_AGG_AVG) {
    if (r.count_slot < 0) return false;
    o.kind = kOrdAvg;
  } else {
    o.kind = kOrdI64;
    // a count is never NULL and lies in [0, rows]: the bits above are the same in every rank
    if ((r.func == PRISM_AGG_COUNT_STAR || r.func == PRISM_AGG_COUNT) && r.count_slot < 0 && !o.is_unsigned && rows_bound) {
      int b = 1;
      while (b < 63 && (rows_bound >> b)) ++b;
      o.value_bits = b;
    }
  }
  return true;
}

// Zone-map pruning: can the predicate be TRUE for any row of the chunk, judging by the
// chunk's per-column min / max / NULL count?  The reference scans every partition of a
// table (time_range: None, src/prism-query/src/providers/table_provider.rs:90); a WHERE on
// the timestamp therefore reads everything there, while here chunks outside the range
// are dropped before any kernel runs.  Unknown shapes answer "maybe".
bool chunk_may_match(const prism_gpu_plan* p, const prism_gpu_table* t, int node, const Chunk& ch) {
  const PredNode& n = p->pred[node];
  switch (n.kind) {
    case PredNode::AND: return chunk_may_match(p, t, n.lhs, ch) && chunk_may_match(p, t, n.rhs, ch);
    case PredNode::OR: return chunk_may_match(p, t, n.lhs, ch) || chunk_may_match(p, t, n.rhs, ch);
    case PredNode::IS_NULL: return ch.stats[n.column].null_count > 0;
    case PredNode::IS_NOT_NULL: return ch.stats[n.column].null_count < ch.rows;
    case PredNode::CMP: {
      const FieldType& f = t->fields[n.column];
      if (f.is_string || f.is_float) return true;
      const ColStats& s = ch.stats[n.column];
      if (s.min > s.max) return false;  // only NULLs: a comparison is never TRUE
      bool any = false;
      if (n.cmp_mask & kLt) any = any || s.min < n.lit_i64;
      if (n.cmp_mask & kEq) any = any || (s.min <= n.lit_i64 && n.lit_i64 <= s.max);
      if (n.cmp_mask & kGt) any = any || s.max > n.lit_i64;
      return any;
    }
    default: return true;
  }
}

int lds_slots_for(int n_slots) {
  // about 32 KiB of LDS per workgroup, so that several workgroups share a CU
  size_t per_entry = 8 * (1 + (size_t)n_slots);
  int s = 256;
  while ((size_t)s * 2 * per_entry <= 32768) s *= 2;
  return s;
}


// Results up to this size land in a pinned block of their own that the exported batch
// keeps; larger ones are copied out of the context's landing buffer.
constexpr size_t kLeasedResultMaxBytes = 64ull << 20;

// A pinned block of at least `bytes` from the context's pool; goes back when the last
// reference is dropped.
std::shared_ptr<void> lease_pinned(prism_gpu_ctx* ctx, size_t bytes) {
  const size_t cls = pow2_ceil(bytes < 4096 ? 4096 : bytes);
  std::shared_ptr<PinnedPool> pool = ctx->result_pool;
  void* p = nullptr;
  {
    std::lock_guard<std::mutex> g(pool->mu);
    auto it = pool->free_blocks.find(cls);
    if (it != pool->free_blocks.end()) {
      p = it->second;
      pool->free_blocks.erase(it);
    }
  }
  if (!p) PRISM_HIP(hipHostMalloc(&p, cls, hipHostMallocDefault));
  return std::shared_ptr<void>(p, [pool, cls](void* q) {
    {
      std::lock_guard<std::mutex> g(pool->mu);
      if (pool->free_blocks.size() < 8) {
        pool->free_blocks.emplace(cls, q);
        return;
      }
    }
    (void)hipHostFree(q);
  });
}

// How the result columns are built from a table's entries (launch_emit_columns).
EmitColumns emit_columns_of(const prism_gpu_table* table, const std::vector<KeySrc>& keys, const KeyLayout& layout,
                            const Lowered& low, bool partial) {
  EmitColumns ec{};
  ec.n_keys = (int)keys.size();
  for (size_t i = 0; i < keys.size(); ++i) {
    const KeyField& kf = layout.fields[i];
    const bool is_dict = table->fields[keys[i].column].is_string;
    ec.keys[i] = {kf.word, kf.bit_pos, kf.bits, kf.null_bit, kf.bias, keys[i].date_bin ? 1 : 0, keys[i].stride, keys[i].origin,
                  is_dict ? (unsigned long long)table->dicts[keys[i].column].values.size() : 0ull};
  }
  auto add = [&](int kind, int main_slot, int count_slot) {
    if (ec.n_aggs >= kMaxEmitAggs) fail(PRISM_GPU_EUNSUPPORTED, "too many aggregate output columns in one plan");
    ec.aggs[ec.n_aggs++] = {kind, main_slot, count_slot};
  };
  for (const AggRule& r : low.rules) {
    switch (r.func) {
      case PRISM_AGG_COUNT_STAR:
      case PRISM_AGG_COUNT: add(kEmitInt, r.main_slot, -1); break;
      case PRISM_AGG_SUM:
      case PRISM_AGG_MIN:
      case PRISM_AGG_MAX: add(kEmitInt, r.main_slot, r.count_slot); break;
      case PRISM_AGG_AVG:
        if (partial) {
          if (r.count_slot >= 0) add(kEmitInt, r.count_slot, -1); else add(kEmitOne, -1, -1);
          add(kEmitF64, r.main_slot, r.count_slot);
        } else {
          add(kEmitAvg, r.main_slot, r.count_slot);
        }
        break;
      default: fail(PRISM_GPU_EINTERNAL, "aggregate cannot be emitted");
    }
  }
  return ec;
}

// Reads the groups of a finished table and appends them to `cols`; with ORDER BY ...
// LIMIT only the leading groups leave the device.  The columns are assembled on the device
// (launch_emit_columns) and arrive through pinned memory as whole columns.  Returns the
// number of groups read.
uint64_t emit_groups(prism_gpu_ctx* ctx, const prism_gpu_plan* plan, const prism_gpu_table* table,
                     const std::vector<KeySrc>& keys, const KeyLayout& layout, const Lowered& low, const GroupTable& gt,
                     bool partial, std::vector<OutColumn>& cols) {
  const EmitColumns ec = emit_columns_of(table, keys, layout, low, partial);
  const size_t n_cols = (size_t)ec.n_keys + (size_t)ec.n_aggs;
  if (n_cols != cols.size()) fail(PRISM_GPU_EINTERNAL, "result columns do not match the output schema");

  uint64_t n = 0;
  const int64_t* h_vals = nullptr;
  const uint8_t* h_valid = nullptr;
  uint64_t h_stride = 0;  // rows between two columns in the host block
  std::unique_lock<std::mutex> pinned_lock(ctx->result_mu, std::defer_lock);
  std::shared_ptr<void> lease;
  {
    HostTimer ht(ctx, "host:read_groups");
    TopKOrder order;
    Scratch d_topk;
    const TopKOrder* use_order = nullptr;
    const uint64_t entries = gt.dense ? gt.capacity : gt.capacity + 1;
    if (!partial && entries >= kTopKMinEntries && topk_order(plan, table, keys, layout, low, cols, order, gt.rows_bound)) {
      d_topk = Scratch(ctx, sizeof(TopKState));
      launch_topk_threshold(ctx, gt.dense, gt.capacity, gt.keys.as<uint64_t>(), gt.keys_hi.as<uint64_t>(),
                            gt.state.as<int64_t>(), order, (uint64_t)std::max<int64_t>(1, plan->limit), d_topk.as<TopKState>());
      use_order = &order;
    }
    auto emit = [&](int64_t* vals, uint8_t* valid, uint64_t stride, unsigned long long* count) {
      PRISM_HIP(hipMemsetAsync(count, 0, 16, ctx->stream));
      launch_emit_columns(ctx, gt.dense, gt.capacity, gt.keys.as<uint64_t>(), gt.keys_hi.as<uint64_t>(), gt.state.as<int64_t>(),
                          ec, vals, valid, stride, count, use_order, d_topk.as<TopKState>());
    };
    Scratch d_count(ctx, 256), d_vals, d_valid;
    std::shared_ptr<void> counts_landing = lease_pinned(ctx, 64);  // (page-locked: see table_flag)
    unsigned long long* const count = (unsigned long long*)counts_landing.get();  // groups emitted, "a dictionary code was out of range"
    unsigned long long* const recount = count + 2;  // the same from the second pass over a large table
    count[0] = count[1] = recount[0] = recount[1] = 0;
    uint64_t stride = 0;
    bool fresh = true;
    for (const OutColumn& oc : cols) fresh = fresh && oc.rows() == 0;
    // A table with few entries is compacted in one pass into buffers sized for all of them,
    // and so are the leading groups of ORDER BY ... LIMIT (at most the limit plus the entries
    // of the last radix bin pass the device's threshold); any other large table is counted first.
    const uint64_t topk_room = use_order ? (uint64_t)std::max<int64_t>(1, plan->limit) + 4096 : 0;
    bool one_pass = use_order ? topk_room < entries : entries * n_cols * 9 <= (48ull << 20);
    if (one_pass) {
      stride = use_order ? topk_room : entries;
      if (fresh) {
        // straight into a page-locked host block that the exported batch will point into:
        // the kernel's stores cross PCIe while it is still scanning the table
        lease = lease_pinned(ctx, (size_t)stride * n_cols * 9);
        int64_t* vals = (int64_t*)lease.get();
        uint8_t* valid = (uint8_t*)lease.get() + (size_t)stride * n_cols * 8;
        emit(vals, valid, stride, d_count.as<unsigned long long>());
        PRISM_HIP(hipMemcpyAsync(count, d_count.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        PRISM_HIP(hipStreamSynchronize(ctx->stream));
        h_vals = vals;
        h_valid = valid;
        h_stride = stride;
      } else {
        d_vals = Scratch(ctx, stride * n_cols * 8);
        d_valid = Scratch(ctx, stride * n_cols);
        emit(d_vals.as<int64_t>(), d_valid.as<uint8_t>(), stride, d_count.as<unsigned long long>());
        PRISM_HIP(hipMemcpyAsync(count, d_count.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        PRISM_HIP(hipStreamSynchronize(ctx->stream));
      }
      if (count[0] > stride) {
        // more groups passed than there was room for (the kernel wrote none of the extra ones): count known, do it again
        one_pass = false;
        h_vals = nullptr;
        h_valid = nullptr;
        lease.reset();
      }
    }
    if (!one_pass) {
      if (count[0] == 0) {
        emit(nullptr, nullptr, 0, d_count.as<unsigned long long>());
        PRISM_HIP(hipMemcpyAsync(count, d_count.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        PRISM_HIP(hipStreamSynchronize(ctx->stream));
      }
      if (count[0]) {
        stride = count[0];
        d_vals = Scratch(ctx, stride * n_cols * 8);
        d_valid = Scratch(ctx, stride * n_cols);
        emit(d_vals.as<int64_t>(), d_valid.as<uint8_t>(), stride, d_count.as<unsigned long long>());
        PRISM_HIP(hipMemcpyAsync(recount, d_count.p, 16, hipMemcpyDeviceToHost, ctx->stream));  // lands with the copies below
      }
    }
    n = count[0];
    if (n && !h_vals) {
      // HBM -> pinned host memory, compacted to n entries per column on the way
      const size_t need = (size_t)n * n_cols * 9;
      fresh = fresh && need <= kLeasedResultMaxBytes;
      uint8_t* pin = nullptr;
      if (fresh) {
        lease = lease_pinned(ctx, need);
        pin = (uint8_t*)lease.get();
      } else {
        pinned_lock.lock();
        if (ctx->result_pinned_bytes < need) {
          if (ctx->result_pinned) (void)hipHostFree(ctx->result_pinned);
          ctx->result_pinned = nullptr;
          ctx->result_pinned_bytes = 0;
          const size_t want = pow2_ceil(need);
          PRISM_HIP(hipHostMalloc(&ctx->result_pinned, want, hipHostMallocDefault));
          ctx->result_pinned_bytes = want;
        }
        pin = (uint8_t*)ctx->result_pinned;
      }
      PRISM_HIP(hipMemcpy2DAsync(pin, n * 8, d_vals.p, stride * 8, n * 8, n_cols, hipMemcpyDeviceToHost, ctx->stream));
      PRISM_HIP(hipMemcpy2DAsync(pin + (size_t)n * n_cols * 8, n, d_valid.p, stride, n, n_cols, hipMemcpyDeviceToHost, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      h_vals = (const int64_t*)pin;
      h_valid = pin + (size_t)n * n_cols * 8;
      h_stride = n;
    }
    if (!h_vals) lease.reset();
    if (n == 0) lease.reset();
    if (count[1] || recount[1]) fail(PRISM_GPU_EINTERNAL, "dictionary code out of range in result");
  }
  HostTimer ht(ctx, "host:assemble");
  for (size_t c = 0; c < n_cols && n; ++c) {
    OutColumn& oc = cols[c];
    const int64_t* v = h_vals + c * h_stride;
    const uint8_t* ok = h_valid + c * h_stride;
    const bool is_dict = c < keys.size() && table->fields[keys[c].column].is_string;
    uint64_t n_values = 0;
    if (is_dict) {
      const ColumnDict& d = table->dicts[keys[c].column];
      if (oc.rows() == 0) oc.dict = &d.values;
      if (oc.dict != &d.values) fail(PRISM_GPU_EINTERNAL, "string key column built from two dictionaries");
      n_values = d.values.size();
      // (the device has checked the codes against the dictionary size, unless there is no dictionary at all)
      if (n_values == 0 && std::memchr(ok, 1, n) != nullptr) fail(PRISM_GPU_EINTERNAL, "dictionary code out of range in result");
    }
    if (lease) {
      // the exported batch points into the block: no copy
      oc.borrow(lease, v, ok, n);
      continue;
    }
    oc.materialise();
    oc.valid.insert(oc.valid.end(), ok, ok + n);
    if (is_dict) {
      const size_t base = oc.codes.size();
      oc.codes.resize(base + n);
      bool bad = false;
      for (size_t r = 0; r < n; ++r) {
        bad |= (ok[r] != 0) & ((uint64_t)v[r] >= n_values);
        oc.codes[base + r] = (uint32_t)v[r];
      }
      if (bad) fail(PRISM_GPU_EINTERNAL, "dictionary code out of range in result");
    } else if (oc.type.is_float) {
      const double* f = (const double*)v;
      oc.f64.insert(oc.f64.end(), f, f + n);
    } else {
      oc.i64.insert(oc.i64.end(), v, v + n);
    }
  }
  return n;
}


// ---- partitioned GROUP BY ------------------------------------------------------------------
// (kernels_part.hip: rows partitioned by key hash in HBM, every bucket aggregated in LDS.)

// Scans of fewer selected rows than ctx->part_min_rows stay on the streaming kernels: below
// that the partitioned path's fixed cost (five launches, a histogram per workgroup) is not
// repaid.  ctx->part_force partitions every plan shape that can be partitioned without
// waiting for the streaming kernel's verdict (tests and experiments; prism_gpu_set_option).
uint64_t part_min_rows(const prism_gpu_ctx* ctx) { return ctx->part_min_rows; }
bool part_forced(const prism_gpu_ctx* ctx) { return ctx->part_force != 0; }

// GROUP BY <one-word key> -> COUNT(*) without a histogram pass (kernels_wc.hip).  True: `gt`
// is the list of the groups and *groups_out their number.  False: a segment or an LDS table
// overflowed (nothing usable was produced; the plan remembers, and the caller partitions with
// the histogram).
bool wc_group_by(prism_gpu_ctx* ctx, prism_gpu_plan* plan, const AggParams& ap, uint64_t selected, uint64_t key_bytes,
                 const std::vector<int>& slot_ops, GroupTable& gt, uint64_t* groups_out, uint32_t leading, GroupRows* lead, bool merge_input) {
  // (with `leading`: the few listed groups are fetched along with the counters -- `lead`, and no table in `gt`)
  constexpr uint64_t kLeadFetch = 2048;
  WcParams w;
  std::memset(&w, 0, sizeof(w));
  w.leading = leading;
  const uint64_t per = std::max<uint64_t>(1, ceil_div(ap.sd.n_tiles, (uint64_t)ctx->n_cus));
  w.tiles_per_wg = (uint32_t)per;
  w.n_wg = (uint32_t)std::max<uint64_t>(1, ceil_div(ap.sd.n_tiles, per));
  const uint32_t P = wc_buckets();
  // a segment: the rows a (workgroup, bucket) pair expects, half as many again, and a margin
  // for small scans whose counts vary more.  (Sizing the segments for the rows the hot keys left
  // on the plan's last run -- half the size on skewed data -- was tried: keys just below the hot
  // table's reach then overflow their segments, and every run paid for a second attempt.)
  const uint64_t mean = ceil_div(selected, (uint64_t)w.n_wg * P);
  const uint64_t cap = pow2_ceil(std::max<uint64_t>(64, mean + mean / 2 + 64));
  while ((1ull << w.cap_shift) < cap) ++w.cap_shift;
  const uint32_t full_shift = w.cap_shift;
  w.lds_entries = wc_lds_entries(w.n_wg);
  w.sample_tiles = (uint32_t)std::min<uint64_t>(32, ap.sd.n_tiles);
  // a candidate: a key seen this often in the sample (1/32768 of it, at least 4 times)
  w.hot_threshold = (uint32_t)std::max<uint64_t>(4, ((uint64_t)w.sample_tiles * kTileRows) >> 15);
  const uint64_t n_cm = wc_sample_counters();
  Scratch rec(ctx, ((uint64_t)P * w.n_wg << full_shift) * 8 + 256), seg_count(ctx, (uint64_t)P * w.n_wg * 4),
      cm(ctx, n_cm * 12 + (uint64_t)wc_hot_slots() * 8 + 256), cand(ctx, (uint64_t)wc_max_candidates() * 4), hot(ctx, (uint64_t)wc_hot_slots() * 8);
  w.rec = rec.as<uint64_t>();
  w.seg_count = seg_count.as<uint32_t>();
  w.s_keys = cm.as<uint64_t>();
  w.s_cnt = (uint32_t*)(cm.as<uint64_t>() + n_cm);
  w.cand = cand.as<uint32_t>();
  w.hot_keys = hot.as<uint64_t>();
  // what starts at zero lies in one block behind the sample's keys and is cleared in one go: the
  // sample's counts, the hot keys' rows, and a small block -- [0] the list cursor (8 bytes),
  // [8] the candidates, [16] four flags, [32] the hot keys' rows in total, [40] the groups outside the
  // hot table, [48] the count a listed group reaches (ORDER BY COUNT(*) DESC LIMIT)
  w.hot_count = (unsigned long long*)(w.s_cnt + n_cm);
  uint8_t* const d_small = (uint8_t*)(w.hot_count + wc_hot_slots());
  w.list_cursor = (unsigned long long*)d_small;
  w.n_cand = (uint32_t*)(d_small + 8);
  w.flags = (uint32_t*)(d_small + 16);
  w.hot_rows = (unsigned long long*)(d_small + 32);
  w.groups_total = (unsigned long long*)(d_small + 40);
  w.cut = (uint32_t*)(d_small + 48);
  if (w.lds_entries < wc_hot_slots() || leading > wc_hot_slots()) w.leading = 0;
  uint64_t groups_est = plan->learnt_groups(merge_input).load(std::memory_order_relaxed);
  if (!groups_est) groups_est = plan->groups_hint;
  uint64_t list_cap = groups_est ? groups_est + groups_est / 16 + 1024 : std::min<uint64_t>(selected, 1ull << 25);
  list_cap = std::max<uint64_t>(1, std::min(list_cap, selected + wc_hot_slots()));
  for (;;) {
    GroupTable list;
    list.list = true;
    list.capacity = list_cap;
    list.n_slots = 1;
    list.key_words = 1;
    list.keys = Scratch(ctx, (list_cap + 1) * 8);
    list.state = Scratch(ctx, (list_cap + 1) * 8);
    list.overflow = Scratch(ctx, 256);
    // (one launch instead of three fills)
    launch_wc_clear(ctx, (unsigned long long*)w.s_keys, n_cm, (unsigned long long*)w.s_cnt, (n_cm * 4 + (size_t)wc_hot_slots() * 8 + 64) / 8,
                    list.overflow.as<uint32_t>());
    w.list_keys = list.keys.as<uint64_t>();
    w.list_state = list.state.as<int64_t>();
    w.list_capacity = list_cap;
    {
      KernelTimer timer(ctx, "group_by", selected * key_bytes);
      launch_wc_group_by(ctx, ap, w, key_bytes);
    }
    struct { unsigned long long cursor; uint32_t n_cand, pad; uint32_t flags[4]; unsigned long long hot_rows, groups_total; uint32_t cut, pad2; } h{};
    HostTimer ht_wait(ctx, "host:part_wait");
    std::shared_ptr<void> h_landing = lease_pinned(ctx, sizeof(h));  // (page-locked: see table_flag)
    PRISM_HIP(hipMemcpyAsync(h_landing.get(), d_small, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    std::shared_ptr<void> lead_pin;
    const uint64_t lead_fetch = std::min<uint64_t>(kLeadFetch, list_cap);
    if (w.leading && lead) {
      lead_pin = lease_pinned(ctx, (size_t)kLeadFetch * 16);
      PRISM_HIP(hipMemcpyAsync(lead_pin.get(), list.keys.p, lead_fetch * 8, hipMemcpyDeviceToHost, ctx->stream));
      PRISM_HIP(hipMemcpyAsync((uint64_t*)lead_pin.get() + kLeadFetch, list.state.p, lead_fetch * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(&h, h_landing.get(), sizeof(h));
    if (std::getenv("PRISM_WC_DEBUG"))
      std::fprintf(stderr, "[wc] n_wg %u tiles/wg %u cap 2^%u entries %u sample %u thr %u: groups %llu cand %u flags %u %u %u\n", w.n_wg,
                   w.tiles_per_wg, w.cap_shift, w.lds_entries, w.sample_tiles, w.hot_threshold, h.cursor, h.n_cand, h.flags[0],
                   h.flags[1], h.flags[2]);
    if (std::getenv("PRISM_WC_DEBUG")) std::fprintf(stderr, "[wc] hot rows %llu of %llu\n", h.hot_rows, (unsigned long long)selected);
    if (h.flags[1] || h.flags[2]) {
      plan->wc_mode.store(-1, std::memory_order_relaxed);
      return false;
    }
    const uint64_t groups = h.cursor;
    if (groups > list_cap || h.flags[0]) {
      // the list was too short (the count is exact now): once more
      list_cap = groups + 1024;
      continue;
    }
    if (h.cut && lead_pin && groups <= lead_fetch) {
      // the groups that can lead the order are on the host already
      const uint64_t* k = (const uint64_t*)lead_pin.get();
      lead->keys.assign(k, k + groups);
      lead->state.assign((const int64_t*)(k + kLeadFetch), (const int64_t*)(k + kLeadFetch) + groups);
      lead->n = groups;
      *groups_out = groups;
      plan->learnt_groups(merge_input).store(std::max<uint64_t>(1, h.groups_total + wc_hot_slots()), std::memory_order_relaxed);
      return true;
    }
    // (only the groups that can lead the order were listed: the list ends behind them)
    if (h.cut) list.capacity = list_cap = groups;
    const uint64_t tail = list_cap + 1 - groups;  // the entries past the groups, and the extra entry
    PRISM_HIP(hipMemsetAsync(list.keys.as<uint64_t>() + groups, 0xff, tail * 8, ctx->stream));
    PRISM_HIP(hipMemsetAsync(list.state.as<int64_t>() + groups, 0, tail * 8, ctx->stream));
    gt = std::move(list);
    *groups_out = groups;
    plan->learnt_groups(merge_input).store(std::max<uint64_t>(1, h.cut ? h.groups_total + wc_hot_slots() : groups), std::memory_order_relaxed);
    return true;
  }
}

// GROUP BY of the scan described by `ap` (keys, values, slots, row mode; no in-kernel
// predicate) through partitioning.  `gt` becomes a list of the groups; returns their number.
uint64_t partitioned_group_by(prism_gpu_ctx* ctx, prism_gpu_plan* plan, AggParams ap, uint64_t selected, uint64_t scan_rows,
                              uint64_t key_bytes, uint64_t value_bytes, bool vals_nullable, const std::vector<int>& slot_ops,
                              GroupTable& gt, uint32_t leading = 0, GroupRows* lead = nullptr, bool merge_input = false) {
  const int words = part_record_words(ap, vals_nullable);
  if (words <= 0) fail(PRISM_GPU_EINTERNAL, "plan shape cannot be partitioned");
  // one-word keys and COUNT(*) alone: the path without a histogram, unless it overflowed before
  if (ctx->part_wc.load(std::memory_order_relaxed) != 0 && plan->wc_mode.load(std::memory_order_relaxed) >= 0 && wc_plan_fits(ap) &&
      selected >= ctx->part_wc_min_rows.load(std::memory_order_relaxed) && selected < (1ull << 32) - wc_hot_slots()) {
    uint64_t n = 0;
    if (wc_group_by(ctx, plan, ap, selected, key_bytes, slot_ops, gt, &n, leading, lead, merge_input)) return n;
  }
  const uint32_t p_max = part_max_buckets(words);
  const uint32_t lds_entries = part_lds_entries(ap);
  const uint32_t lds_limit = lds_entries / 4 * 3;
  uint64_t groups_est = plan->learnt_groups(merge_input).load(std::memory_order_relaxed);
  if (!groups_est) groups_est = plan->groups_hint;
  // a bucket should hold at most two thirds of what the LDS table takes; unknown: as many buckets as there are
  uint64_t want = groups_est ? pow2_ceil(ceil_div(groups_est * 3, (uint64_t)lds_limit * 2)) : p_max;
  // many rows: at least two buckets per CU, whatever the number of groups
  if (selected >= (1ull << 22)) want = std::max<uint64_t>(want, 512);
  const uint32_t P = (uint32_t)std::min<uint64_t>(p_max, std::max<uint64_t>(64, want));
  int p_bits = 0;
  while ((1u << p_bits) < P) ++p_bits;

  PartParams q;
  std::memset(&q, 0, sizeof(q));
  q.p_bits = p_bits;
  q.n_buckets = P;
  q.half_blocks = (ctx->part_half_blocks.load(std::memory_order_relaxed) != 0 && part_scatter_halves(words, P)) ? 1 : 0;
  q.n_wg = part_scan_workgroups(ctx, ap, &q.tiles_per_wg, q.half_blocks != 0);
  // A bucket of more than 4 times the average holds a heavy hitter and is aggregated in
  // slices, by several workgroups, that meet in the HBM table.  (Slices are kept large: every
  // slice sends each of the bucket's ordinary keys it has seen to the HBM table.)
  q.chunk_rows = (uint32_t)std::min<uint64_t>(1u << 30, std::max<uint64_t>(65536, 4 * ceil_div(selected, P)));
  q.max_items = (uint32_t)(P + selected / q.chunk_rows + 2);
  q.lds_entries = lds_entries;
  q.lds_limit = lds_limit;
  q.rec_capacity = std::max<uint64_t>(1, selected);

  Scratch rec[4], d_hist(ctx, (size_t)q.n_wg * P * 4), d_totals(ctx, (size_t)P * 4), d_base(ctx, (size_t)(P + 1) * 8),
      d_items(ctx, (size_t)q.max_items * sizeof(PartItem)), d_small(ctx, 256);
  for (int w = 0; w < words; ++w) {
    rec[w] = Scratch(ctx, q.rec_capacity * 8 + 256);
    q.rec[w] = rec[w].as<uint64_t>();
  }
  q.hist = d_hist.as<uint32_t>();
  q.totals = d_totals.as<uint32_t>();
  q.bucket_base = d_base.as<uint64_t>();
  q.items = d_items.as<PartItem>();
  // one small block: [0] the list cursor (8 bytes), [8] the item count, [16] the two flags
  q.list_cursor = d_small.as<unsigned long long>();
  q.n_items = (uint32_t*)(d_small.as<uint8_t>() + 8);
  q.flags = (uint32_t*)(d_small.as<uint8_t>() + 16);
  q.queue = (uint32_t*)(d_small.as<uint8_t>() + 24);

  uint64_t list_cap = groups_est ? groups_est + groups_est / 16 + 1024 : std::min<uint64_t>(selected, 1ull << 25);
  list_cap = std::max<uint64_t>(1, std::min(list_cap, selected));
  // the HBM table behind the LDS tables: groups of buckets that were aggregated in slices, and stragglers
  uint64_t fb_cap = std::max<uint64_t>(1ull << 16, plan->learnt_table_capacity(merge_input).load(std::memory_order_relaxed));
  const uint64_t fb_max = pow2_ceil(std::max<uint64_t>(1024, 2 * selected));
  fb_cap = std::min(fb_cap, fb_max);
  const int n_slots = (int)slot_ops.size();
  const uint64_t rec_bytes = selected * 8 * (uint64_t)words;
  uint64_t groups = 0;
  for (;;) {
    std::unique_ptr<HostTimer> ht_alloc(new HostTimer(ctx, "host:part_alloc"));
    GroupTable fb;
    table_alloc(ctx, fb, false, fb_cap, slot_ops, ap.key_words);
    GroupTable list;
    list.list = true;
    list.capacity = list_cap;
    list.n_slots = n_slots;
    list.key_words = ap.key_words;
    // (the entries past the last group -- empty keys, and a zero in slot 0, which counts a group's
    // rows -- are written once the number of groups is known: clearing the whole list up front
    // was two passes over tens of megabytes per query)
    list.keys = Scratch(ctx, (list_cap + 1) * 8);
    if (ap.key_words == 2) list.keys_hi = Scratch(ctx, (list_cap + 1) * 8);
    list.state = Scratch(ctx, (list_cap + 1) * 8 * (size_t)n_slots);
    list.overflow = Scratch(ctx, 256);
    PRISM_HIP(hipMemsetAsync(list.overflow.p, 0, 4, ctx->stream));
    PRISM_HIP(hipMemsetAsync(d_small.p, 0, 64, ctx->stream));
    q.list_keys = list.keys.as<uint64_t>();
    q.list_keys_hi = list.keys_hi.as<uint64_t>();
    q.list_state = list.state.as<int64_t>();
    q.list_capacity = list_cap;
    ap.dense = 0;
    ap.capacity = fb_cap;
    ap.g_keys = fb.keys.as<uint64_t>();
    ap.g_keys_hi = fb.keys_hi.as<uint64_t>();
    ap.g_state = fb.state.as<int64_t>();
    ap.overflow = fb.overflow.as<uint32_t>();
    ht_alloc.reset();
    {
      // the whole path under the name the one-kernel GROUP BY reports, so that callers price it the same way
      KernelTimer timer(ctx, "group_by", scan_rows * value_bytes);
      launch_partitioned_group_by(ctx, ap, q, vals_nullable, scan_rows * key_bytes, scan_rows * value_bytes + rec_bytes, rec_bytes);
    }
    struct { unsigned long long cursor; uint32_t n_items, pad; uint32_t flags[2]; } h{};
    uint32_t fb_flag = 0;
    HostTimer ht_wait(ctx, "host:part_wait");
    PRISM_HIP(hipMemcpyAsync(&h, d_small.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    PRISM_HIP(hipMemcpyAsync(&fb_flag, fb.overflow.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    if (fb_flag) {
      if (fb_cap >= fb_max) fail(PRISM_GPU_EINTERNAL, "group table overflow at maximum capacity");
      fb_cap = std::min(fb_max, fb_cap * 4);
      continue;
    }
    groups = h.cursor;
    if (h.flags[1]) {
      // groups that went through the HBM table join the list
      launch_emit(ctx, false, fb.capacity, fb.keys.as<uint64_t>(), fb.keys_hi.as<uint64_t>(), fb.state.as<int64_t>(), n_slots,
                  list.keys.as<uint64_t>(), list.keys_hi.as<uint64_t>(), list.state.as<int64_t>(), list_cap + 1, q.list_cursor,
                  nullptr, nullptr, list_cap);
      PRISM_HIP(hipMemcpyAsync(&h, d_small.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      groups = h.cursor;
    }
    if (groups > list_cap) {
      // the list was too short (the count is exact now): once more
      list_cap = groups + 1024;
      continue;
    }
    {
      const uint64_t tail = list_cap + 1 - groups;  // the entries past the groups, and the extra entry
      PRISM_HIP(hipMemsetAsync(list.keys.as<uint64_t>() + groups, 0xff, tail * 8, ctx->stream));
      if (ap.key_words == 2) PRISM_HIP(hipMemsetAsync(list.keys_hi.as<uint64_t>() + groups, 0xff, tail * 8, ctx->stream));
      PRISM_HIP(hipMemsetAsync(list.state.as<int64_t>() + groups, 0, tail * 8, ctx->stream));
    }
    if (h.flags[1]) plan->learnt_table_capacity(merge_input).store(fb_cap, std::memory_order_relaxed);
    // More groups than the largest number of buckets can hold in LDS: most of them went
    // through the HBM table after all, on top of the partitioning.  The plan goes back to the
    // streaming kernels.
    if (P == p_max && (uint64_t)h.flags[1] > groups / 2 && groups > (uint64_t)p_max * lds_limit)
      plan->part_mode.store(-1, std::memory_order_relaxed);
    gt = std::move(list);
    break;
  }
  plan->learnt_groups(merge_input).store(std::max<uint64_t>(1, groups), std::memory_order_relaxed);
  return groups;
}

// ---- COUNT(DISTINCT x) GROUP BY keys: the outer level over the set of (x, keys) pairs ----
// The pair is packed with x in the low bits of a one-word key (the group key above it), or
// x alone in the low word of a two-word key.
struct DistinctShape {
  bool two_words;
  int x_bits, k_bits, x_null_bit;
};

DistinctShape distinct_shape(const KeyLayout& layout) {
  const KeyField& xf = layout.fields[0];
  DistinctShape d;
  d.two_words = layout.words == 2;
  d.x_bits = d.two_words ? 0 : xf.bits + (xf.null_bit >= 0 ? 1 : 0);
  d.k_bits = d.two_words ? layout.word_bits[1] : layout.total_bits - d.x_bits;
  d.x_null_bit = xf.null_bit;
  return d;
}

// Allocates the outer table (group key -> {pairs seen, pairs with a non-NULL x}) for pairs of this layout.
void distinct_outer_alloc(prism_gpu_ctx* ctx, const KeyLayout& layout, uint64_t selected, GroupTable& q) {
  const DistinctShape d = distinct_shape(layout);
  const bool qdense = d.k_bits <= kDenseMaxBits;
  const uint64_t qcap = qdense ? (1ull << d.k_bits)
                               : pow2_ceil(std::max<uint64_t>(1024, 2 * std::min<uint64_t>(selected, d.k_bits < 63 ? (1ull << d.k_bits) : selected)));
  table_alloc(ctx, q, qdense, qcap, {kSlotAddOne, kSlotAddOne});
}

// Counts the pairs of the set `gt` per group into a fresh outer table `q`.
void distinct_fold_outer(prism_gpu_ctx* ctx, const KeyLayout& layout, uint64_t selected, const GroupTable& gt, GroupTable& q) {
  const DistinctShape d = distinct_shape(layout);
  distinct_outer_alloc(ctx, layout, selected, q);
  DistinctFoldParams fp{};
  fp.set_keys = gt.dense ? nullptr : gt.keys.as<uint64_t>();
  fp.set_keys_hi = d.two_words ? gt.keys_hi.as<uint64_t>() : nullptr;
  fp.set_capacity = gt.capacity + 1;
  fp.set_special = gt.capacity;
  fp.set_rowcount = gt.state.as<int64_t>();
  fp.x_bits = d.x_bits;
  fp.x_null_bit = d.x_null_bit;
  fp.dense = q.dense ? 1 : 0;
  fp.capacity = q.capacity;
  fp.g_keys = q.keys.as<uint64_t>();
  fp.g_state = q.state.as<int64_t>();
  fp.overflow = q.overflow.as<uint32_t>();
  fp.k_bits = d.k_bits;
  if (gt.dense) {
    // a dense (k, x) set has no key array; materialise its occupied entries first
    GroupRows pairs = table_read(ctx, gt);
    Scratch d_pairs(ctx, std::max<uint64_t>(1, pairs.n) * 8);
    if (pairs.n) PRISM_HIP(hipMemcpyAsync(d_pairs.p, pairs.keys.data(), pairs.n * 8, hipMemcpyHostToDevice, ctx->stream));
    fp.set_keys = d_pairs.as<uint64_t>();
    fp.set_capacity = pairs.n;
    fp.set_special = ~0ull;
    if (pairs.n) launch_distinct_fold(ctx, fp);
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
  } else {
    launch_distinct_fold(ctx, fp);
  }
  if (table_overflowed(ctx, q)) fail(PRISM_GPU_EINTERNAL, "distinct group table overflow");
}

// Reads the outer table and appends the result rows (group keys, then the distinct count for
// every aggregate of the plan).  Returns the number of groups.
uint64_t distinct_emit_outer(prism_gpu_ctx* ctx, const prism_gpu_plan* plan, const prism_gpu_table* table,
                             const std::vector<KeySrc>& keys, const KeyLayout& layout, const GroupTable& q, std::vector<OutColumn>& cols) {
  const DistinctShape d = distinct_shape(layout);
  GroupRows g = table_read(ctx, q);
  // unpack keys from the high bits: shift the layout down by x_bits
  KeyLayout kl;
  for (size_t i = 1; i < layout.fields.size(); ++i) {
    KeyField f = layout.fields[i];
    f.word = 0;  // the outer table's key is the group key alone, in one word
    if (f.bits) f.bit_pos -= d.x_bits;
    if (f.null_bit >= 0) f.null_bit -= d.x_bits;
    kl.fields.push_back(f);
  }
  append_key_columns(table, keys, kl, 0, g, cols, 0);
  for (size_t a = 0; a < plan->aggs.size(); ++a) {
    OutColumn& oc = cols[keys.size() + a];
    const int64_t* v = g.state.data() + (size_t)g.n;  // slot 1 = distinct count
    oc.i64.insert(oc.i64.end(), v, v + g.n);
    oc.valid.resize(oc.valid.size() + g.n, 1);
  }
  return g.n;
}

// COUNT(DISTINCT x) GROUP BY keys over two-word pairs, partitioned twice (kernels_part2.hip):
// the rows' pairs are sorted into coarse buckets by a hash of the group key, every coarse
// bucket into fine buckets by a hash of the pair, and the fine buckets are deduplicated in
// LDS and counted per group into `outer`.  False: a fine bucket did not fit (very uneven
// data) or the outer table overflowed; nothing usable was produced.
// With `shard` (a query sharded over several GPUs) the last pass writes every fine bucket's
// distinct records instead of counting them, the bucket numbers follow from req->sizing_rows
// (the same on every rank) and from nothing this rank has learnt, and `outer` stays empty.
// `time_bits` / `time_buckets`: when the group key's topmost field is a time key without NULLs,
// its bits and the number of time buckets the scanned rows span (else 0).  Rows that arrive in
// time order are then sorted by the top bits of the group key already, which is all the first
// level would do for them: the second level reads the columns itself (Part2Params::direct) --
// 56 bytes a row moved instead of 96.  Rows out of order still count right (through global
// atomics); when more than an eighth of them were, the plan goes back to two levels.
bool partitioned_distinct(prism_gpu_ctx* ctx, prism_gpu_plan* plan, const AggParams& ap, uint64_t selected, uint64_t key_bytes,
                          const KeyLayout& layout, GroupTable& outer, const DistinctShardRequest* req = nullptr,
                          DistinctShard* shard = nullptr, int time_bits = 0, uint64_t time_buckets = 0) {
  const int words = part_record_words(ap, false);
  if (words != 2 || selected >= (1ull << 32) || (selected == 0 && !shard)) return false;
  const uint64_t sizing = shard ? std::max<uint64_t>(1, req->sizing_rows) : selected;
  if (sizing >= (1ull << 32) || selected > sizing) return false;
  // coarse buckets of about half a million rows, fine buckets of about a quarter of the LDS set
  // (at most 1024 of them while the scatter pass can run two half-size workgroups per CU: the
  // counters of 2048 buckets next to two tiles do not fit the LDS)
  const int p1_most = ctx->part_half_blocks.load(std::memory_order_relaxed) != 0 && part_scatter_halves(words, 1024) ? 10 : 11;
  int p1 = 6;
  while (p1 < p1_most && (sizing >> p1) > (1ull << 19)) ++p1;
  if ((1u << p1) > part_max_buckets(words)) return false;
  // (a fine bucket of up to half the set's entries is taken in one go; larger ones, in the
  // coarse buckets of very frequent groups, in several passes)
  // Rows that repeat a pair need no entry: when the plan's last run found that only a share
  // of the rows were distinct pairs, a fine bucket may hold as many more rows as keeps its
  // expected distinct pairs where they would be if every row were one (at most four times as
  // many rows; the last pass falls back to several passes per bucket should a bucket's pairs
  // crowd its set all the same, and the share is learnt anew on every run).  Fewer, larger
  // fine buckets mean longer runs out of the second scatter pass and less work per bucket.
  const uint32_t seen = shard ? 0u : plan->distinct_pairs_per_1024.load(std::memory_order_relaxed);
  const uint64_t rows_per_entry_x8 = seen ? std::min<uint64_t>(32, std::max<uint64_t>(8, 8 * 1024 / (uint64_t)seen)) : 8;
  auto fine_bits = [&](int coarse_bits, uint32_t entries) {
    const uint64_t fine_rows = (uint64_t)entries * 3 / 8 * rows_per_entry_x8 / 8;
    int b = 0;
    while (b < 10 && (sizing >> (coarse_bits + b)) > fine_rows) ++b;
    return (sizing >> (coarse_bits + b)) > fine_rows ? -1 : b;
  };
  // One-word records at the second level: when both bucket numbers are taken from bijections
  // of the words, a record in its fine bucket needs only the bits the bucket does not imply;
  // they must fit 63 bits, the groups of a coarse bucket the LDS counters (9 bits), and x must
  // have no NULL bit (it would be mixed away).  Half the bytes written and read, and an LDS
  // set of twice the entries.  Buckets a little finer than the rows alone ask for are taken
  // when that is what makes the records fit, as long as a fine bucket keeps 128 rows.
  const int lo_bits = layout.word_bits[0], hi_bits = layout.word_bits[1];
  bool compact = ctx->part2_compact.load(std::memory_order_relaxed) != 0 && layout.fields[0].null_bit < 0 && lo_bits <= 63;
  int p2 = -1;
  if (compact) {
    int c1 = p1, c2 = fine_bits(c1, part2_lds_entries(true));
    int most = 0;  // log2 of the coarse buckets the first level can take
    while ((2u << most) <= part_max_buckets(words)) ++most;
    const int top1 = std::min<int>({p1_most, hi_bits, most});
    auto excess = [&]() { return lo_bits + hi_bits - 63 - c1 - c2; };
    while (c2 >= 0 && c1 < top1 && (excess() > 0 || hi_bits - c1 > 9)) {
      ++c1;
      c2 = fine_bits(c1, part2_lds_entries(true));
    }
    if (c2 >= 0 && excess() > 0) c2 += excess();
    compact = c2 >= 0 && c2 <= 10 && c2 <= lo_bits && c1 <= hi_bits && hi_bits - c1 <= 9 && excess() <= 0 &&
              ((c1 == p1 && c2 == fine_bits(c1, part2_lds_entries(true))) || (sizing >> (c1 + c2)) >= 128);
    if (compact) { p1 = c1; p2 = c2; }
  }
  if (shard && !compact) return false;  // (distinct records travel in the one-word form only)
  if (!compact) p2 = fine_bits(p1, part2_lds_entries(false));
  // ---- no first level at all? ----
  bool direct = false;
  uint64_t coarse_used = 0;  // (direct) coarse buckets the scanned rows span
  // (a sharded query: every rank must decide alike -- what this rank's plan has learnt about its rows stays out of it)
  if (time_bits > 0 && time_buckets > 0 && ctx->part2_direct.load(std::memory_order_relaxed) != 0 &&
      (shard || plan->part2_direct.load(std::memory_order_relaxed) >= 0) && ctx->part2_compact.load(std::memory_order_relaxed) != 0 &&
      layout.fields[0].null_bit < 0 && lo_bits <= 63 && ap.row_mode != kRowsSelvec) {
    const int d1 = std::min(time_bits, 11);
    // fine buckets of at most the rows the LDS set takes in one go, where the rows allow it
    coarse_used = std::min<uint64_t>(1ull << d1, (time_buckets >> (time_bits - d1)) + 1);
    const uint64_t fine_rows = (uint64_t)part2_lds_entries(true) * 3 / 8 * rows_per_entry_x8 / 8;
    int d2 = 0;
    while (d2 < 10 && sizing / (coarse_used << d2) > fine_rows) ++d2;
    while (d2 < 10 && lo_bits + hi_bits - d1 - d2 > 63) ++d2;
    if (hi_bits - d1 <= 9 && d2 <= lo_bits && lo_bits + hi_bits - d1 - d2 <= 63 && sizing / (coarse_used << d2) <= 4 * fine_rows) {
      direct = true;
      compact = true;
      p1 = d1;
      p2 = d2;
    }
  }
  if (p2 < 0) return false;  // more rows than 2^21 fine buckets take
  const uint32_t P1 = 1u << p1;
  const uint32_t set_entries = part2_lds_entries(compact);
  if (shard && selected == 0) {
    // a rank with nothing to scan still takes part in the exchange: every bucket empty
    const uint32_t nf = 1u << (p1 + p2);
    shard->p1 = p1;
    shard->p2 = p2;
    shard->direct = direct;
    shard->lo_bits = lo_bits;
    shard->hi_bits = hi_bits;
    shard->n_fine = nf;
    shard->lds_entries = set_entries;
    shard->fine = Scratch(ctx, (size_t)nf * 16 + 64);
    PRISM_HIP(hipMemsetAsync(shard->fine.p, 0, (size_t)nf * 16 + 64, ctx->stream));
    shard->fine_count = shard->fine.as<uint32_t>();
    shard->fine_base = shard->fine.as<uint32_t>() + 2 * (size_t)nf;
    shard->records = Scratch(ctx, 256);
    return true;
  }

  PartParams q1;
  std::memset(&q1, 0, sizeof(q1));
  q1.p_bits = p1;
  q1.n_buckets = P1;
  q1.half_blocks = (ctx->part_half_blocks.load(std::memory_order_relaxed) != 0 && part_scatter_halves(words, P1)) ? 1 : 0;
  q1.n_wg = part_scan_workgroups(ctx, ap, &q1.tiles_per_wg, q1.half_blocks != 0);
  q1.chunk_rows = 1u << 30;
  q1.max_items = P1 + 4;
  q1.lds_entries = part_lds_entries(ap);
  q1.lds_limit = q1.lds_entries / 4 * 3;
  q1.rec_capacity = std::max<uint64_t>(1, selected);
  q1.by_hi = 1;
  q1.hi_bits = compact ? hi_bits : 0;
  Scratch rec[2], rec2[2], d_hist(ctx, (size_t)q1.n_wg * P1 * 4), d_totals(ctx, (size_t)P1 * 4), d_base(ctx, (size_t)(P1 + 1) * 8),
      d_items(ctx, (size_t)q1.max_items * sizeof(PartItem)), d_small(ctx, 256);
  for (int w = 0; w < 2 && !direct; ++w) {
    rec[w] = Scratch(ctx, std::max<uint64_t>(1, selected) * 8 + 256);
    q1.rec[w] = rec[w].as<uint64_t>();
  }
  q1.hist = d_hist.as<uint32_t>();
  q1.totals = d_totals.as<uint32_t>();
  q1.bucket_base = d_base.as<uint64_t>();
  q1.items = d_items.as<PartItem>();
  // one small block: [0] the list cursor, [8] the item count, [16] two flags, [24] the queue
  // of the first level (unused here); [32] the queue and [40] the flags of the second level
  q1.list_cursor = d_small.as<unsigned long long>();
  q1.n_items = (uint32_t*)(d_small.as<uint8_t>() + 8);
  q1.flags = (uint32_t*)(d_small.as<uint8_t>() + 16);
  q1.queue = (uint32_t*)(d_small.as<uint8_t>() + 24);
  PRISM_HIP(hipMemsetAsync(d_small.p, 0, 256, ctx->stream));
  if (!direct) launch_partitioned_group_by(ctx, ap, q1, false, selected * key_bytes, selected * (key_bytes + 16), 0, true);

  const uint32_t n_fine = 1u << (p1 + p2);
  // The fine buckets get regions of fixed size (no histogram pass over the records) unless
  // this plan's data overflowed one before, or the padded array would pass 2^32 records.
  const uint64_t fixed_capacity = part2_fixed_capacity(selected, n_fine);
  bool fixed = !shard && !direct && ctx->part2_fixed.load(std::memory_order_relaxed) != 0 && plan->part2_fixed.load(std::memory_order_relaxed) >= 0 &&
               fixed_capacity < (1ull << 32);
  // (no first level:) regions of one size for the fine buckets the rows can reach, half as large again as
  // their even share, and a spill list for what does not fit them -- no histogram pass either
  const uint32_t n_regions = direct ? (uint32_t)std::min<uint64_t>(n_fine, coarse_used << p2) : 0;
  const uint32_t region_cap = direct ? (uint32_t)(((((selected / std::max<uint64_t>(1, coarse_used)) >> p2) * 3 / 2 + 64) + 15) & ~15ull) : 0;
  const uint64_t spill_capacity = selected / 2 + 4096, region_records = (uint64_t)n_regions * region_cap;
  Scratch emitted;  // (a sharded query without a first level: where the last pass writes the buckets' distinct records)
  if (direct && shard) emitted = Scratch(ctx, std::max<uint64_t>({1, selected, region_records}) * 8 + 256);
  // (a sharded query takes the regions and the spill list too: its last pass then writes a bucket's distinct records
  // into the bucket's region, and a bucket with more of them than the region holds sends the plan back to the histogram)
  bool spill = direct && ctx->part2_fixed.load(std::memory_order_relaxed) != 0 && plan->part2_fixed.load(std::memory_order_relaxed) >= 0 &&
               region_records + spill_capacity < (1ull << 32);
  Scratch spill_rec, spill_fine, spill_small, d_pieces;
  // (the scatter pass by staging lines: a workgroup per piece of a coarse bucket's stretch of tiles; about three pieces a CU)
  const int ahead_keys = part2_keys_ahead(ap);
  // A coarse bucket is cut into at most 16 pieces (its fine buckets then are 17 runs for the last pass), of at
  // least a third of a CU's share of the tiles each; with fewer pieces than 1.5 a CU the tile-by-tile pass stays.
  const uint64_t tiles_per_coarse = ceil_div(ap.sd.n_tiles, std::max<uint64_t>(1, coarse_used));
  const uint32_t piece_tiles = (uint32_t)std::max<uint64_t>({8, ceil_div(ap.sd.n_tiles, (uint64_t)ctx->n_cus * 3), ceil_div(tiles_per_coarse, 16)});
  const uint32_t piece_max = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, ceil_div(tiles_per_coarse, (uint64_t)piece_tiles)));
  const uint32_t max_pieces = (uint32_t)(P1 + ap.sd.n_tiles / piece_tiles + 2);
  bool lines = spill && ctx->part2_lines.load(std::memory_order_relaxed) != 0 && plan->part2_lines.load(std::memory_order_relaxed) >= 0 &&
               (ahead_keys == 2 || ahead_keys == 3) && ap.sd.n_tiles < (1ull << 32) &&
               // (the rows of the tiles where the coarse bucket changes take the spill list: a coarse bucket should run through many tiles)
               ((ap.sd.n_tiles >= 64 * std::max<uint64_t>(1, coarse_used) && 2 * (ap.sd.n_tiles / piece_tiles) >= 3 * (uint64_t)ctx->n_cus) ||
                ctx->part2_lines.load(std::memory_order_relaxed) == 2) &&
               part2_lines_fit((uint32_t)p2, piece_tiles);
  const size_t n_runs = lines ? piece_max + 1 : 2;  // (sizes the arrays; a run without the lines uses two)
  const size_t spill_small_bytes = ((size_t)3 * (n_regions + 1) + 2 * n_runs * n_regions) * 4 + 64;
  if (spill) {
    spill_rec = Scratch(ctx, spill_capacity * 8 + 64);
    spill_fine = Scratch(ctx, spill_capacity * 4 + 64);
    spill_small = Scratch(ctx, spill_small_bytes);
    if (lines) d_pieces = Scratch(ctx, (size_t)max_pieces * sizeof(P2Piece) + (size_t)(P1 + 1) * 4 + 64 + 256);
  }
  for (int w = 0; w < 2; ++w)
    if (w == 0 || !compact)
      rec2[w] = Scratch(ctx, std::max<uint64_t>(fixed ? fixed_capacity : selected, spill ? region_records + spill_capacity : 0) * 8 + 256);
  Scratch d_fine(ctx, (size_t)n_fine * 16 + 64);
  struct { uint32_t flags[2]; uint64_t pad[2]; unsigned long long pairs[8]; uint32_t n_pieces, pieces_too_long; unsigned long long dropped; } h{};
  static_assert(sizeof(h) == 104, "layout of the small block from offset 40 on");
  for (;;) {
    PRISM_HIP(hipMemsetAsync(d_fine.p, 0, (size_t)n_fine * 4, ctx->stream));
    PRISM_HIP(hipMemsetAsync(d_small.as<uint8_t>() + 32, 0, 224, ctx->stream));
    if (!shard) distinct_outer_alloc(ctx, layout, selected, outer);
    Part2Params q2;
    std::memset(&q2, 0, sizeof(q2));
    q2.emit_out = !shard ? nullptr : direct ? emitted.as<uint64_t>() : rec[0].as<uint64_t>();  // (the first level's records are done with once the second has scattered them)
    q2.in_lo = rec[0].as<uint64_t>();
    q2.in_hi = rec[1].as<uint64_t>();
    q2.out_lo = rec2[0].as<uint64_t>();
    q2.out_hi = compact ? nullptr : rec2[1].as<uint64_t>();
    q2.out_capacity = fixed ? fixed_capacity : selected;
    q2.compact = compact ? 1 : 0;
    q2.lo_bits = lo_bits;
    q2.hi_bits = hi_bits;
    q2.coarse_base = q1.bucket_base;
    q2.p1_bits = (uint32_t)p1;
    q2.p2_bits = (uint32_t)p2;
    q2.n_rows = selected;
    q2.slice_rows = 1u << 16;
    q2.fine_count = d_fine.as<uint32_t>();
    q2.fine_cursor = d_fine.as<uint32_t>() + n_fine;
    q2.fine_base = d_fine.as<uint32_t>() + 2 * (size_t)n_fine;
    q2.fine_limit = fixed ? d_fine.as<uint32_t>() + 3 * (size_t)n_fine + 1 : nullptr;
    q2.lds_entries = set_entries;
    q2.x_null_bit = layout.fields[0].null_bit;
    if (!shard) {
      q2.dense = outer.dense ? 1 : 0;
      q2.capacity = outer.capacity;
      q2.g_keys = outer.keys.as<uint64_t>();
      q2.g_state = outer.state.as<int64_t>();
      q2.overflow = outer.overflow.as<uint32_t>();
    }
    q2.queue = (uint32_t*)(d_small.as<uint8_t>() + 32);
    q2.flags = (uint32_t*)(d_small.as<uint8_t>() + 40);
    q2.pairs = compact ? (unsigned long long*)(d_small.as<uint8_t>() + 64) : nullptr;  // [64, 128) of the block
    if (direct) {
      q2.direct = 1;
      q2.in_lo = q2.in_hi = nullptr;
      q2.coarse_base = nullptr;
      q2.n_tiles = ap.sd.n_tiles;
      // A slice keeps the LDS counters where its first rows put them: it should not run through more
      // coarse buckets than half of what the counters span (2048 counters, at most 8 coarse buckets).
      const uint64_t span = std::min<uint64_t>(8, 2048u >> p2), rows_per_coarse = selected / std::max<uint64_t>(1, coarse_used);
      const uint64_t slice_rows = std::max<uint64_t>(1, span / 2) * rows_per_coarse;
      q2.slice_tiles = (uint32_t)std::max<uint64_t>(2, std::min<uint64_t>(32, (slice_rows / (2 * kTileRows)) * 2));
      q2.strays = (unsigned long long*)(d_small.as<uint8_t>() + 48);
      if (spill) {
        PRISM_HIP(hipMemsetAsync(spill_small.p, 0, spill_small_bytes, ctx->stream));
        q2.fine_limit = d_fine.as<uint32_t>() + 3 * (size_t)n_fine + 1;
        q2.out_capacity = region_records + spill_capacity;
        q2.spill_rec = spill_rec.as<uint64_t>();
        q2.spill_fine = spill_fine.as<uint32_t>();
        q2.spill_cursor = (unsigned long long*)(d_small.as<uint8_t>() + 56);
        q2.spill_capacity = spill_capacity;
        q2.n_regions = n_regions;
        q2.region_cap = region_cap;
        q2.spill_count = spill_small.as<uint32_t>();
        q2.spill_base = q2.spill_count + (n_regions + 1);
        q2.spill_cur = q2.spill_base + (n_regions + 1);
        q2.seg_base_out = q2.spill_cur + (n_regions + 1);
        q2.seg_cnt_out = q2.seg_base_out + (lines ? piece_max + 1 : 2) * (size_t)n_regions;
        if (lines) {
          q2.lines = 1;
          q2.piece_tiles = piece_tiles;
          q2.piece_max = piece_max;
          q2.max_pieces = max_pieces;
          q2.piece_chunks = part2_piece_chunks(piece_tiles);
          q2.pieces = d_pieces.as<P2Piece>();
          q2.bounds = (uint32_t*)(d_pieces.as<P2Piece>() + max_pieces);
          q2.rare = (P2Rare*)(d_pieces.as<uint8_t>() + (((size_t)max_pieces * sizeof(P2Piece) + (size_t)(P1 + 1) * 4 + 63) & ~(size_t)63));
          q2.n_pieces = (uint32_t*)(d_small.as<uint8_t>() + 128);
          q2.dropped = (unsigned long long*)(d_small.as<uint8_t>() + 136);
          q2.pieces_too_long = q2.n_pieces + 1;
        }
      }
    }
    launch_distinct_second_level(ctx, q2, direct ? &ap : nullptr);
    {
      std::shared_ptr<void> h_landing = lease_pinned(ctx, sizeof(h));  // (page-locked: see table_flag)
      PRISM_HIP(hipMemcpyAsync(h_landing.get(), q2.flags, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      std::memcpy(&h, h_landing.get(), sizeof(h));
    }
    collect_timings(ctx);
    if (std::getenv("PRISM_WC_DEBUG"))
      std::fprintf(stderr, "[part2] p1 %d p2 %d compact %d fixed %d direct %d spill %d lines %d (%u tiles a piece, %u pieces a bucket) rows %llu: flags %u %u strays %llu spilled %llu pieces %u dropped %llu\n", p1, p2,
                   (int)compact, (int)fixed, (int)direct, (int)spill, (int)lines, piece_tiles, piece_max, (unsigned long long)selected, h.flags[0], h.flags[1],
                   (unsigned long long)h.pad[0], (unsigned long long)h.pad[1], h.n_pieces, h.dropped);
    // (rows out of time order: the next run sorts by the group key first again)
    if (direct && !shard && h.pad[0] > selected / 8) plan->part2_direct.store(-1, std::memory_order_relaxed);
    if (lines && h.pieces_too_long != 0) {
      // a time bucket with far more rows than the others: this plan scatters tile by tile from now on
      plan->part2_lines.store(-1, std::memory_order_relaxed);
      lines = false;
      continue;
    }
    if (spill && h.flags[1] != 0) {
      // the spill list ran over (few distinct pairs that repeat a great deal): exact places from a histogram from now on
      plan->part2_fixed.store(-1, std::memory_order_relaxed);
      spill = false;
      continue;
    }
    if (fixed && h.flags[1] != 0) {
      // a fine bucket outgrew its region (few distinct values that repeat a lot): exact places from now on
      plan->part2_fixed.store(-1, std::memory_order_relaxed);
      fixed = false;
      continue;
    }
    break;
  }
  if (h.flags[0] != 0 || (!shard && table_overflowed(ctx, outer))) {
    outer = GroupTable();
    return false;
  }
  if (shard) {
    shard->p1 = p1;
    shard->p2 = p2;
    shard->direct = direct;
    shard->lo_bits = lo_bits;
    shard->hi_bits = hi_bits;
    shard->n_fine = n_fine;
    shard->lds_entries = set_entries;
    shard->fine_count = d_fine.as<uint32_t>();
    shard->fine_base = d_fine.as<uint32_t>() + 2 * (size_t)n_fine;
    shard->records = direct ? std::move(emitted) : std::move(rec[0]);
    shard->fine = std::move(d_fine);
    return true;
  }
  if (compact) {
    unsigned long long pairs = 0;
    for (unsigned long long v : h.pairs) pairs += v;
    plan->distinct_pairs_per_1024.store((uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, pairs * 1024 / selected)), std::memory_order_relaxed);
  }
  return true;
}

}  // namespace

}  // namespace prism

// A group table kept in HBM between prism_gpu_aggregate and prism_gpu_state_finish.
struct prism_gpu_state {
  prism_gpu_plan* plan = nullptr;
  prism_gpu_ctx* ctx = nullptr;  // the context the table was built on: the plan's, or the child context of a streaming partition
  prism_gpu_table* table = nullptr;
  std::vector<prism::KeySrc> keys;
  prism::KeyLayout layout;
  prism::Lowered low;
  prism::GroupTable gt;
  std::vector<int> slot_ops;
  // the occupied groups in compact form (hash tables): [n] keys ([n] high words), [n_slots][n] accumulators
  bool compacted = false;
  uint64_t c_n = 0;
  prism::Scratch c_keys, c_keys_hi, c_state;
  // COUNT(DISTINCT): 1 = the table is the set of (x, group) pairs, 2 = the table counts the
  // pairs per group (slot 0: pairs folded in, slot 1: the distinct count)
  // 3 = no table: the distinct records of every fine bucket (`shard`), for the exchange of a sharded query
  int distinct = 0;
  uint64_t selected = 0;  // rows that went into the set (sizes the outer table)
  std::unique_ptr<prism::DistinctShard> shard;
};

namespace prism {

namespace {

}  // namespace

namespace {

// Share of the rows a leaf is expected to keep, from the column's min / max / NULL count
// over the scan and a uniform distribution in between.
double leaf_selectivity(const ScanSet& ss, const PredNode& n, const ColumnDict* dict) {
  const ColInfo ci = col_info(ss, n.column);
  const double rows = (double)std::max<uint64_t>(1, ss.rows);
  const double not_null = 1.0 - std::min(1.0, (double)ci.nulls / rows);
  if (n.kind == PredNode::IS_NULL) return 1.0 - not_null;
  if (n.kind == PredNode::IS_NOT_NULL) return not_null;
  if (dict) {  // a string predicate: the share of the dictionary's values that pass
    const std::vector<uint8_t> lut = dict_lut(n, *dict);
    size_t pass = 0;
    for (size_t c = 0; c + 1 < lut.size(); ++c) pass += lut[c] != 0;
    return not_null * (double)pass / (double)std::max<size_t>(1, lut.size() - 1);
  }
  if (ci.min > ci.max) return 0.0;
  const double lo = (double)ci.min, hi = (double)ci.max, lit = (double)n.lit_i64, range = hi - lo + 1.0;
  auto clamp01 = [](double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); };
  double f = 0.0;
  if (n.cmp_mask & kLt) f += clamp01((lit - lo) / range);
  if (n.cmp_mask & kEq) f += (lit >= lo && lit <= hi) ? 1.0 / range : 0.0;
  if (n.cmp_mask & kGt) f += clamp01((hi - lit) / range);
  return clamp01(f) * not_null;
}

// WHERE inside the aggregation kernel.  The predicate qualifies when it is one leaf, or a
// conjunction or a disjunction of at most kMaxFusedLeaves leaves, each a comparison of a
// fixed-width integer column with a literal, an IS [NOT] NULL, or a string comparison / LIKE
// on a column whose dictionary has at most 64 values (NOT has
// been pushed into the leaves at plan time, so the TRUE rows of AND / OR are the
// intersection / union of the leaves' TRUE rows).  Collects the leaves; `is_or` tells how
// they combine.
bool fusable_leaves(const prism_gpu_plan* p, const prism_gpu_table* t, int node, int combiner, std::vector<int>& leaves) {
  const PredNode& n = p->pred[node];
  if (n.kind == PredNode::AND || n.kind == PredNode::OR) {
    if (combiner >= 0 && combiner != (int)n.kind) return false;
    return fusable_leaves(p, t, n.lhs, (int)n.kind, leaves) && fusable_leaves(p, t, n.rhs, (int)n.kind, leaves);
  }
  const FieldType& f = t->fields[n.column];
  if (f.is_string) {
    // through the dictionary, when its codes fit one 64-bit set
    const ColumnDict& d = t->dicts[n.column];
    if (!d.alive || d.values.size() > 64) return false;
  } else {
    if (n.kind != PredNode::CMP && n.kind != PredNode::IS_NULL && n.kind != PredNode::IS_NOT_NULL) return false;
    if (f.is_float || (f.is_unsigned && f.width == 8)) return false;
  }
  leaves.push_back(node);
  return leaves.size() <= (size_t)kMaxFusedLeaves;
}

// Tries to move the plan's WHERE into the streaming GROUP BY kernel: on success `ap` carries
// the leaves (and any column only they read) and the caller runs with row_mode kRowsAll.
// `always`: also when few rows are expected to survive (the caller has no bitmap path).
bool try_fuse_predicate(const prism_gpu_plan* plan, const prism_gpu_table* table, ScanSet& ss, const std::vector<KeySrc>& keys,
                        const KeyLayout& layout, AggParams& ap, bool always) {
    std::vector<int> leaves;
    if (fusable_leaves(plan, table, plan->pred_root, -1, leaves)) {
      AggParams trial = ap;
      trial.n_keys = (int)layout.specs.size();
      for (size_t i = 0; i < layout.specs.size(); ++i) trial.keys[i] = layout.specs[i];
      trial.key_bits = plan->keys.empty() ? 0 : layout.total_bits;
      trial.key_words = 1;
      trial.lds_slots = lds_slots_for(trial.n_slots);
      trial.row_mode = kRowsAll;
      trial.n_pred = (int)leaves.size();
      trial.pred_or = plan->pred[plan->pred_root].kind == PredNode::OR ? 1 : 0;
      bool fits = true;
      // the kernel variant with 32-bit keys (the same test pick_stream_kernel makes)
      bool k32 = trial.key_bits <= 31;
      for (int k = 0; k < trial.n_keys; ++k) k32 = k32 && trial.keys[k].fast32 != 0;
      for (size_t l = 0; l < leaves.size() && fits; ++l) {
        const PredNode& n = plan->pred[leaves[l]];
        const FieldType& f = table->fields[n.column];
        int kind = n.kind == PredNode::CMP ? kLeafCmp : (n.kind == PredNode::IS_NULL ? kLeafIsNull : kLeafIsNotNull);
        long long lit = (long long)n.lit_i64;
        if (f.is_string && n.kind != PredNode::IS_NULL && n.kind != PredNode::IS_NOT_NULL) {
          kind = kLeafLut;
          const std::vector<uint8_t> lut = dict_lut(n, table->dicts[n.column]);
          uint64_t set = 0;
          for (size_t c = 0; c + 1 < lut.size() && c < 64; ++c) set |= (uint64_t)(lut[c] != 0) << c;
          lit = (long long)set;
        }
        // A column that is a GROUP BY key is loaded already: read it from there (WHERE on the
        // timestamp that is also bucketed is the everyday query).  The 32-bit-key kernel holds
        // the low word only, which is enough: value - base is below 2^32 for every row.
        int from_key = -1;
        for (size_t i = 0; i < keys.size() && from_key < 0; ++i)
          if (keys[i].column == n.column && (k32 ? trial.keys[i].width >= 4 : trial.keys[i].width == 8)) from_key = (int)i;
        if (from_key >= 0) {
          const KeySpec& ks = trial.keys[from_key];
          const __int128 rel = (__int128)lit - (__int128)ks.base;
          long long lit_k32 = rel < -1 ? -1ll : (rel > ((__int128)1 << 32) ? (1ll << 32) : (long long)rel);
          // (codes of the scan start at base: the set moves down with them)
          if (kind == kLeafLut) lit_k32 = ks.base >= 0 && ks.base < 64 ? (long long)((uint64_t)lit >> ks.base) : 0ll;
          trial.pred[l] = {-(from_key + 1), kind, n.cmp_mask, (uint32_t)(uint64_t)ks.base, lit, lit_k32};
          continue;
        }
        const int col = ss.use(n.column, f.is_string);  // strings: their dictionary codes
        int v = 0;
        while (v < trial.n_vals && trial.vals[v].col != col) ++v;
        if (v == trial.n_vals) {
          if (trial.n_vals >= kMaxVals) { fits = false; break; }
          trial.vals[v] = {col, f.is_string ? 4 : f.width, (f.is_string || f.is_unsigned) ? 1 : 0, 0};
          trial.n_vals += 1;
        }
        trial.pred[l] = {v, kind, n.cmp_mask, 0u, lit, 0ll};
      }
      // Few expected survivors: the bitmap pass followed by a gather of only those rows is
      // cheaper than streaming every column of every row through the aggregation kernel.
      double keep = trial.pred_or ? 0.0 : 1.0;
      for (int leaf : leaves) {
        const PredNode& ln = plan->pred[leaf];
        const double f = leaf_selectivity(ss, ln, table->fields[ln.column].is_string ? &table->dicts[ln.column] : nullptr);
        keep = trial.pred_or ? 1.0 - (1.0 - keep) * (1.0 - f) : keep * f;
      }
      if (fits && (always || keep >= 1.0 / 16) && group_by_streams(trial)) {
        const int n_keys = ap.n_keys, key_bits = ap.key_bits, key_words = ap.key_words, lds_slots = ap.lds_slots;
        ap = trial;
        // (the branches below set these for their own shape)
        ap.n_keys = n_keys; ap.key_bits = key_bits; ap.key_words = key_words; ap.lds_slots = lds_slots;
        return true;
      }
    }
  return false;
}

void execute_impl(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                  bool merge_input, ArrowArray* out, const prism_gpu_col_stats* stats_override, prism_gpu_state** capture,
                  bool emit_partial = false, prism_gpu_ctx* ctx_override = nullptr, std::vector<OutColumn>* cols_out = nullptr,
                  const DistinctShardRequest* shard_req = nullptr);
void execute_mixed(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions, ArrowArray* out,
                   bool emit_partial = false, prism_gpu_ctx* ctx_override = nullptr);
}

void execute_plan(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                  bool merge_input, ArrowArray* out) {
  if (plan->mixed && !merge_input) {
    execute_mixed(plan, table, partition_ids, n_partitions, out);
    return;
  }
  execute_impl(plan, table, partition_ids, n_partitions, merge_input, out, nullptr, nullptr);
}

void execute_plan_on(prism_gpu_ctx* ctx, prism_gpu_plan* plan, prism_gpu_table* table, bool merge_input, bool emit_partial,
                     ArrowArray* out) {
  if (plan->mixed && !merge_input) {
    execute_mixed(plan, table, nullptr, 0, out, emit_partial, ctx);
    return;
  }
  execute_impl(plan, table, nullptr, 0, merge_input, out, nullptr, nullptr, emit_partial, ctx);
}

namespace {

// With `capture` the GROUP BY stops once the group table is complete and hands the table
// over instead of producing rows; `stats_override` then replaces the scan's own column
// statistics so that every rank of a multi-GPU query packs keys the same way.
void execute_impl(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                  bool merge_input, ArrowArray* out, const prism_gpu_col_stats* stats_override, prism_gpu_state** capture,
                  bool emit_partial, prism_gpu_ctx* ctx_override, std::vector<OutColumn>* cols_out, const DistinctShardRequest* shard_req) {
  if (plan->mixed) fail(PRISM_GPU_EUNSUPPORTED, "COUNT(DISTINCT) next to other aggregates runs through prism_gpu_execute only");
  // (ctx_override: a child context of the plan's, for partitions that stream concurrently)
  prism_gpu_ctx* ctx = ctx_override ? ctx_override : plan->ctx;
  bind(ctx);
  if (capture && plan->keys.empty()) fail(PRISM_GPU_EUNSUPPORTED, "device-resident aggregate state needs a GROUP BY");
  HostTimer whole(ctx, "host:execute");
  if (table->ctx != ctx) fail(PRISM_GPU_EINVAL, "plan and table belong to different contexts");
  const std::vector<FieldType>& expect = merge_input ? plan->partial_fields : plan->fields;
  if (table->fields.size() != expect.size()) fail(PRISM_GPU_EINVAL, "table schema does not match the plan's input schema");
  for (size_t i = 0; i < expect.size(); ++i)
    if (table->fields[i].logical != expect[i].logical)
      fail(PRISM_GPU_EINVAL, "column " + std::to_string(i) + " (`" + table->fields[i].name + "`) has a different type than the plan expects");
  // A string key whose host dictionary gave up gets its codes from the device (exactly: see
  // table_device_dictionary); so does the column of a COUNT(DISTINCT).
  {
    std::vector<int> string_keys;
    for (size_t i = 0; i < plan->keys.size(); ++i) string_keys.push_back(merge_input ? (int)i : plan->keys[i].column);
    if (plan->distinct_plan) string_keys.push_back(merge_input ? (int)plan->keys.size() : plan->distinct_col);
    for (int c : string_keys)
      if (c >= 0 && (size_t)c < table->fields.size() && table->fields[(size_t)c].is_string) table_device_dictionary(table, c);
  }
  table_sync_stats(table);

  // ---- which chunks ----
  ScanSet ss;
  ss.ctx = ctx;
  ss.stats_override = stats_override;
  uint64_t chunks_pruned = 0;
  {
    std::lock_guard<std::mutex> chunks_lock(table->mu);  // appends may be adding chunks
    for (const Chunk& ch : table->chunks) {
      bool take = partition_ids == nullptr || n_partitions == 0;
      for (size_t i = 0; !take && i < n_partitions; ++i) take = partition_ids[i] == ch.partition_id;
      if (!take || ch.rows == 0) continue;
      if (plan->pred_root >= 0 && !merge_input && !chunk_may_match(plan, table, plan->pred_root, ch)) {
        chunks_pruned += 1;
        continue;
      }
      ss.chunks.push_back(&ch);
      ss.segs.push_back({ch.rows, ss.n_tiles});
      ss.n_tiles += ceil_div(ch.rows, kTileRows);
      ss.rows += ch.rows;
    }
  }
  if (chunks_pruned) {
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->stats.chunks_pruned += chunks_pruned;
  }
  if (ss.n_tiles * (uint64_t)kTileRows > 0xffffffffull)
    fail(PRISM_GPU_EUNSUPPORTED, "more than 2^32 rows in one scan; split the scan by partition");
  if (stats_override) {
    // Keys are packed from these statistics without a mask: a value outside them would land
    // in another key's bits, or outside a dense table.  They must cover this scan.
    std::vector<int> used;
    for (const KeyPlan& k : plan->keys) used.push_back(k.column);
    for (const AggPlan& a : plan->aggs) if (a.column >= 0) used.push_back(a.column);
    for (int c : used)
      for (const Chunk* ch : ss.chunks) {
        const ColStats& cs = ch->stats[c];
        const prism_gpu_col_stats& o = stats_override[c];
        const bool covered = (cs.min > cs.max || (o.min <= cs.min && cs.max <= o.max)) && (cs.null_count == 0 || o.nulls > 0);
        if (!covered)
          fail(PRISM_GPU_EINVAL, "the column statistics passed for `" + table->fields[c].name + "` do not cover partition " +
                                     std::to_string(ch->partition_id) + " of this scan (its values or NULLs fall outside them)");
      }
  }
  const uint64_t n_words = ss.n_tiles * kTileWords;
  // emit_partial: state columns whatever the plan's mode (the chunks of a streaming partition
  // are aggregated one by one and merged at the end)
  const bool partial = emit_partial || (plan->mode == PRISM_MODE_PARTIAL && !merge_input);

  // ---- lowering ----
  std::vector<KeySrc> keys;
  for (size_t i = 0; i < plan->keys.size(); ++i) {
    const KeyPlan& k = plan->keys[i];
    if (merge_input) keys.push_back({(int)i, false, 0, 0});
    else keys.push_back({k.column, k.kind == PRISM_KEY_DATE_BIN, k.stride_ms, k.origin_ms});
  }
  Lowered low;
  KeySrc distinct_src{-1, false, 0, 0};
  if (plan->distinct_plan) {
    low.slot(kSlotAddOne, -1);
    distinct_src.column = merge_input ? (int)plan->keys.size() : plan->distinct_col;
  } else {
    low = merge_input ? lower_merge(plan) : lower_raw(plan, ss);
  }

  AggParams ap;
  std::memset(&ap, 0, sizeof(ap));
  KeyLayout layout;
  KeyCursor cursor;
  if (!ss.chunks.empty() || capture) {
    if (plan->distinct_plan) {
      layout_keys(ss, table, {distinct_src}, cursor, layout);
      // (x, keys) beyond one word: x alone in the low word, the group key in the high one
      if (cursor.bit[0] + key_bits_needed(ss, table, keys) > 64) cursor.closed0 = true;
    }
    layout_keys(ss, table, keys, cursor, layout);
    fill_slots(table, ss, low, ap);
  }

  // ---- predicate ----
  Selection sel;
  uint64_t selected = ss.rows;
  PredEval pe{plan, table, ss, ctx, n_words, {}};
  const bool has_pred = plan->pred_root >= 0 && !merge_input && !ss.chunks.empty();
  // The common "COUNT(*) WHERE col <op> literal" shape needs no bitmap at all.
  const PredNode *range_a = nullptr, *range_b = nullptr;
  bool count_only = has_pred && plan->keys.empty() && !plan->distinct_plan && plan->aggs.size() == 1 &&
                    plan->aggs[0].func == PRISM_AGG_COUNT_STAR &&
                    one_column_range(plan, table, plan->pred_root, &range_a, &range_b);
  if (has_pred && count_only) ss.use(range_a->column, false);
  // Simple predicates are evaluated by the aggregation kernel itself: no bitmap pass, and
  // a predicate on a column the aggregates read anyway costs no memory traffic at all.
  bool fused = false;
  // A plan that partitions its rows evaluates the predicate as a bitmap (the partitioning
  // kernels take a selection, not a predicate).
  // (a sharded COUNT(DISTINCT) partitions its pairs: its WHERE is wanted as a bitmap)
  const bool prefers_part = !plan->keys.empty() && (plan->part_mode.load(std::memory_order_relaxed) == 1 || part_forced(ctx) || shard_req != nullptr);
  if (has_pred && !count_only && !plan->distinct_plan && layout.words == 1 && !prefers_part)
    fused = try_fuse_predicate(plan, table, ss, keys, layout, ap, false);
  if (has_pred && !count_only && !fused) {
    // register every column the predicate touches before the upload
    for (const PredNode& n : plan->pred)
      if (n.column >= 0) ss.use(n.column, leaf_uses_codes(table, ss, n));
  }
  ss.upload();
  ap.sd = ss.desc;
  ap.cols = ss.cols_dev;

  // a count many workgroups add to: kCountShards partial sums (kernels.hpp), added up here
  Scratch d_scalar(ctx, kCountBytes);
  if (has_pred && count_only) {
    const PredNode& n = *range_a;
    const FieldType& f = table->fields[n.column];
    PRISM_HIP(hipMemsetAsync(d_scalar.p, 0, kCountBytes, ctx->stream));
    {
      KernelTimer timer(ctx, "filter_count", ss.rows * f.width);
      launch_cmp_leaf(ctx, ss.desc, ss.col(ss.use(n.column, false)), f.width, f.is_unsigned, n.cmp_mask, n.lit_i64, nullptr,
                      d_scalar.as<unsigned long long>(), range_b ? range_b->cmp_mask : (kLt | kEq | kGt),
                      range_b ? range_b->lit_i64 : 0);
    }
    // (into a page-locked block: a copy into pageable memory goes through the runtime's own staging first)
    std::shared_ptr<void> landing = lease_pinned(ctx, kCountBytes);
    PRISM_HIP(hipMemcpyAsync(landing.get(), d_scalar.p, kCountBytes, hipMemcpyDeviceToHost, ctx->stream));
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    const unsigned long long c = count_total((const unsigned long long*)landing.get());
    collect_timings(ctx);
    std::vector<OutColumn> cols = make_columns(partial ? plan->partial_fields : plan->final_fields);
    cols[0].push_int((int64_t)c);
    {
      std::lock_guard<std::mutex> g(ctx->mu);
      ctx->stats.rows_scanned += ss.rows;
      ctx->stats.rows_selected += c;
      ctx->stats.groups_out += 1;
    }
    if (cols_out) *cols_out = std::move(cols);
    else export_record_batch(cols, out);
    return;
  }
  Scratch selvec;
  // Rows that passed a predicate evaluated inside the aggregation kernel: copied back
  // asynchronously, valid after the next stream synchronisation.  It lands in a leased
  // page-locked block rather than on this stack frame, so a copy still in flight when an
  // error unwinds the frame writes into pooled memory, not into a dead stack slot.
  std::shared_ptr<void> fused_landing;
  auto fused_slot = [&]() -> unsigned long long* {
    if (!fused_landing) {
      fused_landing = lease_pinned(ctx, kCountBytes);
      std::memset(fused_landing.get(), 0, kCountBytes);
    }
    return (unsigned long long*)fused_landing.get();
  };
  auto fused_count = [&]() { return count_total(fused_slot()); };
  if (fused) {
    ap.row_mode = kRowsAll;
    ap.selected_out = d_scalar.as<unsigned long long>();
    ap.selected_sharded = 1;
  }
  if (has_pred && !fused) {
    sel.all = false;
    sel.bits = pe.eval(plan->pred_root);
    PRISM_HIP(hipMemsetAsync(d_scalar.p, 0, kCountBytes, ctx->stream));
    launch_bitmap_count(ctx, sel.bits.as<uint64_t>(), n_words, d_scalar.as<unsigned long long>());
    {
      std::shared_ptr<void> landing = lease_pinned(ctx, kCountBytes);  // (page-locked: see table_flag)
      PRISM_HIP(hipMemcpyAsync(landing.get(), d_scalar.p, kCountBytes, hipMemcpyDeviceToHost, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      sel.count = selected = count_total((const unsigned long long*)landing.get());
    }
    // COUNT(*) alone is the number of TRUE rows: nothing left to aggregate
    bool only_count_star = plan->keys.empty() && !plan->distinct_plan && !capture && !plan->aggs.empty();
    for (const AggPlan& a : plan->aggs) only_count_star = only_count_star && a.func == PRISM_AGG_COUNT_STAR;
    if (only_count_star) {
      collect_timings(ctx);
      std::vector<OutColumn> count_cols = make_columns(partial ? plan->partial_fields : plan->final_fields);
      for (OutColumn& oc : count_cols) oc.push_int((int64_t)selected);
      {
        std::lock_guard<std::mutex> g(ctx->mu);
        ctx->stats.rows_scanned += ss.rows;
        ctx->stats.rows_selected += selected;
        ctx->stats.groups_out += 1;
      }
      if (cols_out) *cols_out = std::move(count_cols);
      else export_record_batch(count_cols, out);
      return;
    }
    ap.row_mode = kRowsBitmap;
    ap.sel_bits = sel.bits.as<uint64_t>();
    // Few survivors: compact them into a selection vector so that the aggregation
    // gathers only those rows instead of streaming every column in full.
    const bool part_likely = (prefers_part || layout.words == 2 || plan->distinct_plan) && !plan->keys.empty() && selected >= part_min_rows(ctx);
    if (selected * 8 < ss.rows && selected > 0 && !part_likely) {
      Scratch tile_offsets(ctx, (ss.n_tiles + 1) * 4);
      KernelTimer timer(ctx, "compact_selvec", n_words * 8 + selected * 4);
      launch_compact(ctx, sel.bits.as<uint64_t>(), ss.n_tiles, tile_offsets.as<uint32_t>(), nullptr,
                     d_scalar.as<unsigned long long>());
      selvec = Scratch(ctx, selected * 4);
      launch_compact(ctx, sel.bits.as<uint64_t>(), ss.n_tiles, tile_offsets.as<uint32_t>(), selvec.as<uint32_t>(), nullptr);
      PRISM_HIP(hipStreamSynchronize(ctx->stream));  // tile_offsets is released at scope end
      ap.row_mode = kRowsSelvec;
      ap.selvec = selvec.as<uint32_t>();
      ap.n_sel = selected;
    }
  }

  std::vector<OutColumn> cols = make_columns(partial ? plan->partial_fields : plan->final_fields);
  uint64_t groups_out = 0;
  uint64_t value_bytes = 0;  // algorithmic bytes per row of the aggregation kernel
  for (int j = 0; j < ap.n_vals; ++j) value_bytes += ap.vals[j].width;
  for (auto& s : layout.specs) value_bytes += s.width;

  // no input rows: an ungrouped aggregate still yields one row
  auto empty_input_row = [&]() {
    size_t c = 0;
    if (partial && plan->distinct_plan) cols[c++].push_null();
    else for (const AggPlan& a : plan->aggs) {
      bool is_count = a.func == PRISM_AGG_COUNT_STAR || a.func == PRISM_AGG_COUNT || a.func == PRISM_AGG_COUNT_DISTINCT;
      if (is_count) cols[c++].push_int(0);
      else if (a.func == PRISM_AGG_AVG && partial) { cols[c++].push_int(0); cols[c++].push_null(); }
      else cols[c++].push_null();
    }
    groups_out = 1;
  };
  if ((ss.chunks.empty() || selected == 0) && !capture) {
    if (plan->keys.empty()) empty_input_row();
  } else if (plan->keys.empty() && !plan->distinct_plan) {
    // ---- ungrouped aggregates ----
    GroupRows g;
    g.n = 1;
    g.keys.assign(1, 0);
    g.state.resize(ap.n_slots);
    const uint64_t agg_bytes = (ap.row_mode == kRowsSelvec ? selected : ss.rows) * value_bytes;
    if (ap.row_mode != kRowsSelvec) {
      // every row, or a bitmap-selected share of them: the streaming GROUP BY kernel with no
      // key columns, i.e. one group whose replicas in LDS keep the lanes of a wave apart
      std::vector<int> slot_ops;
      for (int s = 0; s < ap.n_slots; ++s) slot_ops.push_back(ap.slots[s].op);
      GroupTable gt;
      table_alloc(ctx, gt, true, 1, slot_ops);
      ap.n_keys = 0;
      ap.key_bits = 0;
      ap.key_words = 1;
      ap.lds_slots = lds_slots_for(ap.n_slots);
      ap.dense = 1;
      ap.capacity = 1;
      ap.g_keys = nullptr;
      ap.g_keys_hi = nullptr;
      ap.g_state = gt.state.as<int64_t>();
      ap.overflow = gt.overflow.as<uint32_t>();
      if (fused) PRISM_HIP(hipMemsetAsync(d_scalar.p, 0, kCountBytes, ctx->stream));
      launch_group_by(ctx, ap, agg_bytes, "agg_global");
      // entry 0 and the unused extra entry of every slot (into page-locked memory: see table_flag)
      std::shared_ptr<void> both_landing = lease_pinned(ctx, (size_t)ap.n_slots * 16);
      const int64_t* both = (const int64_t*)both_landing.get();
      PRISM_HIP(hipMemcpyAsync(both_landing.get(), gt.state.p, (size_t)ap.n_slots * 16, hipMemcpyDeviceToHost, ctx->stream));
      if (fused) PRISM_HIP(hipMemcpyAsync(fused_slot(), d_scalar.p, kCountBytes, hipMemcpyDeviceToHost, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      for (int s = 0; s < ap.n_slots; ++s) g.state[s] = both[(size_t)s * 2];
      if (fused) selected = fused_count();
    } else {
      // a selection vector of few rows: gather them
      std::vector<int64_t> init(ap.n_slots);
      for (int s = 0; s < ap.n_slots; ++s) init[s] = slot_identity_host(ap.slots[s].op);
      Scratch d_out(ctx, 256);
      PRISM_HIP(hipMemcpyAsync(d_out.p, init.data(), init.size() * 8, hipMemcpyHostToDevice, ctx->stream));
      {
        KernelTimer timer(ctx, "agg_global", agg_bytes);
        launch_agg_global(ctx, ap, d_out.as<int64_t>());
      }
      PRISM_HIP(hipMemcpyAsync(g.state.data(), d_out.p, g.state.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
    }
    // (a predicate evaluated in the kernel may turn out to select nothing)
    if (selected == 0) empty_input_row();
    else append_agg_columns(low, g, partial, cols, 0);
    groups_out = 1;
  } else {
    // ---- GROUP BY ----
    const int total_bits = layout.total_bits;
    std::vector<int> slot_ops;
    for (int s = 0; s < ap.n_slots; ++s) slot_ops.push_back(ap.slots[s].op);
    ap.n_keys = (int)layout.specs.size();
    for (size_t i = 0; i < layout.specs.size(); ++i) ap.keys[i] = layout.specs[i];
    ap.lds_slots = lds_slots_for(ap.n_slots);
    ap.key_bits = total_bits;
    ap.key_words = layout.words;
    ap.window_hint = 0;
    // COUNT(DISTINCT)'s inner level is a set: with two-word keys (always a hash table, whose
    // occupied entries are told by their keys) no accumulator is touched at all
    ap.keys_only = (plan->distinct_plan && layout.words == 2) ? 1 : 0;
    if (!plan->distinct_plan && !merge_input && keys.size() >= 1 && layout.words == 1) {
      // the lowest bit any time-typed key occupies: windows start at multiples of it, so
      // that a window holds whole time buckets with all their other key values
      int align = 64;
      for (size_t i = 0; i < keys.size(); ++i) {
        if (!(keys[i].date_bin || table->fields[keys[i].column].logical == PRISM_COL_TIMESTAMP)) continue;
        const KeyField& kf = layout.fields[i];
        if (kf.bits) align = std::min(align, kf.bit_pos);
        if (kf.null_bit >= 0) align = std::min(align, kf.null_bit);
      }
      if (align < 32) ap.window_hint = 1 + align;
    }
    uint64_t bound = selected;
    if (total_bits < 63) bound = std::min<uint64_t>(bound, 1ull << total_bits);
    GroupTable gt;
    GroupRows lead;  // ORDER BY COUNT(*) DESC LIMIT: the leading groups, when the aggregation pass already picked them
    // ---- can, and should, the rows be partitioned by key first? ----
    bool vals_nullable = false;
    for (int j = 0; j < ap.n_vals; ++j)
      for (const Chunk* ch : ss.chunks) vals_nullable = vals_nullable || ch->cols[ss.uses[ap.vals[j].col].column].validity != nullptr;
    // (nothing selected: nothing to partition -- and the partitioning passes size their work by the rows)
    bool part_ok = ap.row_mode != kRowsSelvec && selected > 0 && selected >= part_min_rows(ctx) && part_record_words(ap, vals_nullable) > 0;
    for (const Chunk* ch : ss.chunks) part_ok = part_ok && ch->rows < (1ull << 29);  // byte offsets inside a column fit 32 bits
    uint64_t key_bytes = 0;
    for (auto& sp : layout.specs) key_bytes += sp.width;
    // Two-word keys have no LDS stage on the streaming side (every row is an HBM atomic), so
    // they are partitioned whenever the expected groups fit the buckets' LDS tables.
    bool wide_fits = false;
    if (layout.words == 2 && part_ok && plan->part_mode.load(std::memory_order_relaxed) >= 0) {
      AggParams probe = ap;
      probe.key_words = 2;
      probe.keys_only = (plan->distinct_plan && layout.words == 2) ? 1 : 0;
      const int words = part_record_words(probe, vals_nullable);
      uint64_t est = plan->learnt_groups(merge_input).load(std::memory_order_relaxed);
      if (!est) est = plan->groups_hint;
      wide_fits = words > 0 && est > 0 && est <= (uint64_t)part_max_buckets(words) * (part_lds_entries(probe) / 2);
    }
    // (A state kept for an exchange between GPUs stays a directly indexed table whenever the
    // agreed key space allows one: whether to partition depends on this rank's rows and on what
    // its plan has learnt, and a rank that came with a list where the others hold dense tables
    // would enter a different collective.)
    const bool state_stays_dense = capture != nullptr && total_bits <= kDenseMaxBits;
    const bool use_part = part_ok && !state_stays_dense && ap.n_pred == 0 && plan->part_mode.load(std::memory_order_relaxed) >= 0 &&
                          (wide_fits || (layout.words == 1 && prefers_part) || (layout.words == 2 && part_forced(ctx)));
    // otherwise the streaming kernel runs, and reports it when its rows scatter
    ap.stop_if_scattered = (!use_part && part_ok && !state_stays_dense && plan->part_mode.load(std::memory_order_relaxed) == 0) ? 1 : 0;
    // The packed key indexes the table directly when the key space is small, or when it is
    // not much larger than the number of rows (say GROUP BY a timestamp in ms over sorted
    // data: neighbouring rows then update neighbouring entries, where a hash table would
    // scatter them over gigabytes).
    // (Not when the ranks of a multi-GPU query must take the same decision: the row count is local.)
    bool dense = total_bits <= kDenseMaxBits ||
                 (!ss.stats_override && total_bits <= kDenseMaxBitsLarge && (1ull << total_bits) <= 2 * bound);
    uint64_t capacity = dense ? (1ull << total_bits)
                              : pow2_ceil(std::max<uint64_t>(1024, 2 * (plan->groups_hint ? std::min<uint64_t>(plan->groups_hint, bound)
                                                                                         : std::min<uint64_t>(bound, 1ull << 24))));
    const uint64_t max_capacity = pow2_ceil(std::max<uint64_t>(1024, 2 * bound));
    if (!dense) {
      const uint64_t learned = plan->learned_capacity.load(std::memory_order_relaxed);
      if (learned > capacity) capacity = std::min(learned, max_capacity);
    }
    // COUNT(DISTINCT) over two-word pairs at scale: the set of pairs is too large for the
    // buckets of one partition level and, as one table in HBM, costs two random cache lines per
    // row; partitioned twice, the pairs are counted per group without the set ever being stored.
    GroupTable outer_q;
    bool outer_ready = false;
    // the group key's topmost field, when it is a time key without NULLs: its bits, and the time buckets the scan spans
    auto leading_time_key = [&](int& time_bits, uint64_t& time_buckets) {
      for (size_t i = 0; i < keys.size() && layout.fields.size() == keys.size() + 1; ++i) {
        const KeyField& kf = layout.fields[1 + i];
        const bool time_like = keys[i].date_bin || table->fields[keys[i].column].logical == PRISM_COL_TIMESTAMP;
        if (!time_like || kf.word != 1 || kf.bits == 0 || kf.null_bit >= 0 || kf.bit_pos + kf.bits != layout.word_bits[1]) continue;
        const ColInfo ci = col_info(ss, keys[i].column);
        if (ci.min > ci.max) continue;
        const long long lo_b = keys[i].date_bin ? bucket_number_host(ci.min, keys[i].stride, keys[i].origin) : ci.min;
        const long long hi_b = keys[i].date_bin ? bucket_number_host(ci.max, keys[i].stride, keys[i].origin) : ci.max;
        time_bits = kf.bits;
        time_buckets = (uint64_t)hi_b - (uint64_t)lo_b + 1;
      }
    };
    if (plan->distinct_plan && layout.words == 2 && !partial && !capture && part_ok && ap.n_pred == 0 &&
        selected >= (part_forced(ctx) ? 1ull : (1ull << 24)) && plan->part_mode.load(std::memory_order_relaxed) >= 0) {
      KernelTimer timer(ctx, "group_by", ss.rows * value_bytes);
      int time_bits = 0;
      uint64_t time_buckets = 0;
      leading_time_key(time_bits, time_buckets);
      outer_ready = partitioned_distinct(ctx, plan, ap, selected, key_bytes, layout, outer_q, nullptr, nullptr, time_bits, time_buckets);
      if (!outer_ready) plan->part_mode.store(-1, std::memory_order_relaxed);  // uneven data: the one-table way from now on
    }
    // The rank's part of a sharded COUNT(DISTINCT): the same two partition levels, ending with the
    // distinct records of every fine bucket instead of counts (the ranks decided together that
    // the plan takes this way: distinct_shard_possible).
    std::unique_ptr<DistinctShard> shard;
    if (shard_req && capture) {
      if (!(plan->distinct_plan && layout.words == 2 && !partial && ap.n_pred == 0 && ap.row_mode != kRowsSelvec))
        fail(PRISM_GPU_EINTERNAL, "the plan cannot produce the distinct records of a sharded COUNT(DISTINCT)");
      for (const Chunk* ch : ss.chunks)
        if (ch->rows >= (1ull << 29)) fail(PRISM_GPU_EUNSUPPORTED, "a sharded COUNT(DISTINCT) takes chunks of fewer than 2^29 rows");
      shard = std::make_unique<DistinctShard>();
      KernelTimer timer(ctx, "group_by", ss.rows * value_bytes);
      // (the statistics are the ranks' agreed ones: every rank finds the same time key and the same span)
      int time_bits = 0;
      uint64_t time_buckets = 0;
      leading_time_key(time_bits, time_buckets);
      if (!partitioned_distinct(ctx, plan, ap, selected, key_bytes, layout, outer_q, shard_req, shard.get(), time_bits, time_buckets))
        fail(PRISM_GPU_EINTERNAL, "the distinct pairs of this rank's partitions do not fit the fine buckets");
      outer_ready = true;
    }
    if (use_part && !outer_ready) {
      // ORDER BY COUNT(*) DESC LIMIT n, the whole answer made here: the aggregation pass may list only
      // the groups that can be among the first n (ClickBench's GROUP BY "UserID" ... LIMIT 10)
      uint32_t leading = 0;
      if (!capture && !partial && !merge_input && !plan->distinct_plan && ctx->part_wc_leading.load(std::memory_order_relaxed) != 0 &&
          plan->order_by >= 0 && plan->descending && plan->limit >= 1 && plan->limit <= 1024 && (size_t)plan->order_by >= keys.size() &&
          (size_t)plan->order_by - keys.size() < low.rules.size()) {
        const AggRule& r = low.rules[(size_t)plan->order_by - keys.size()];
        if ((r.func == PRISM_AGG_COUNT_STAR || r.func == PRISM_AGG_COUNT) && r.count_slot < 0 && r.main_slot == 0) leading = (uint32_t)plan->limit;
      }
      partitioned_group_by(ctx, plan, ap, selected, ss.rows, key_bytes, value_bytes, vals_nullable, slot_ops, gt, leading, &lead, merge_input);
      dense = false;
    }
    std::shared_ptr<void> flag_landing;  // (see below: the table's flag word, read with the groups)
    while (!use_part && !outer_ready) {
      {
        HostTimer ht(ctx, "host:table_alloc");
        table_alloc(ctx, gt, dense, capacity, slot_ops, layout.words);
      }
      ap.dense = dense ? 1 : 0;
      ap.capacity = capacity;
      ap.g_keys = gt.keys.as<uint64_t>();
      ap.g_keys_hi = gt.keys_hi.as<uint64_t>();
      ap.g_state = gt.state.as<int64_t>();
      ap.overflow = gt.overflow.as<uint32_t>();
      if (fused) PRISM_HIP(hipMemsetAsync(d_scalar.p, 0, kCountBytes, ctx->stream));
      launch_group_by(ctx, ap, (ap.row_mode == kRowsSelvec ? selected : ss.rows) * value_bytes);
      if (dense && !ap.stop_if_scattered) break;
      if (dense && !capture) {
        // Nothing but "the rows scatter" can come back from a directly indexed table: the groups are
        // read out at once, and the flag word lands with them (one wait instead of two; it is looked
        // at before the groups are used).
        flag_landing = lease_pinned(ctx, 64);
        *(volatile uint32_t*)flag_landing.get() = 0;
        PRISM_HIP(hipMemcpyAsync(flag_landing.get(), gt.overflow.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        break;
      }
      const uint32_t flag = table_flag(ctx, gt);
      if (flag == 3u && ap.stop_if_scattered) {
        // The rows mostly missed the LDS tables: this plan partitions its rows from now on.
        // Start over (the predicate may have been evaluated inside the kernel that just
        // stopped; the partitioning kernels want it as a bitmap).
        plan->part_mode.store(1, std::memory_order_relaxed);
        gt = GroupTable();
        sel.bits.reset();
        selvec.reset();
        PRISM_HIP(hipStreamSynchronize(ctx->stream));
        collect_timings(ctx);
        execute_impl(plan, table, partition_ids, n_partitions, merge_input, out, stats_override, capture, emit_partial, ctx_override,
                     cols_out, shard_req);
        return;
      }
      if (dense || flag == 0u) break;
      if (capacity >= max_capacity) fail(PRISM_GPU_EINTERNAL, "group table overflow at maximum capacity");
      capacity = std::min(max_capacity, capacity * 4);  // more groups than expected: grow and redo the pass
    }
    if (!dense && !use_part && !outer_ready) plan->learned_capacity.store(capacity, std::memory_order_relaxed);
    // lands with the next synchronisation of the stream (every path below has one before the statistics)
    if (fused) PRISM_HIP(hipMemcpyAsync(fused_slot(), d_scalar.p, kCountBytes, hipMemcpyDeviceToHost, ctx->stream));
    if (capture) {
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      collect_timings(ctx);
      auto st = std::make_unique<prism_gpu_state>();
      st->plan = plan;
      st->ctx = ctx;
      st->table = table;
      st->keys = keys;
      st->layout = layout;
      st->low = low;
      st->gt = std::move(gt);
      st->slot_ops = slot_ops;
      if (fused) selected = fused_count();
      st->distinct = shard ? 3 : plan->distinct_plan ? 1 : 0;
      st->shard = std::move(shard);
      st->selected = selected;
      {
        std::lock_guard<std::mutex> g(ctx->mu);
        ctx->stats.rows_scanned += ss.rows;
        ctx->stats.rows_selected += selected;
      }
      *capture = st.release();
      return;
    }
    if (lead.n) {
      append_key_columns(table, keys, layout, 0, lead, cols, 0);
      append_agg_columns(low, lead, false, cols, keys.size());
      groups_out = lead.n;
    } else if (!plan->distinct_plan) {
      if (!merge_input) gt.rows_bound = ss.rows;  // (merged counts add up over states: no such bound)
      groups_out = emit_groups(ctx, plan, table, keys, layout, low, gt, partial, cols);
    } else if (partial) {
      // inner GROUP BY (keys, x): one row per distinct pair
      GroupRows g = table_read(ctx, gt);
      append_key_columns(table, keys, layout, 1, g, cols, 0);
      append_key_columns(table, {distinct_src}, layout, 0, g, cols, keys.size());
      groups_out = g.n;
    } else {
      // outer GROUP BY keys: COUNT(x) over the distinct pairs
      if (outer_ready) {
        groups_out = distinct_emit_outer(ctx, plan, table, keys, layout, outer_q, cols);
      } else {
        GroupTable q;
        distinct_fold_outer(ctx, layout, selected, gt, q);
        groups_out = distinct_emit_outer(ctx, plan, table, keys, layout, q, cols);
      }
    }
    if (flag_landing) {
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      if (*(volatile uint32_t*)flag_landing.get() == 3u && ap.stop_if_scattered) {
        // the kernel stopped because its rows scatter: what was read out is no result.  This plan
        // partitions its rows from now on; start over.
        plan->part_mode.store(1, std::memory_order_relaxed);
        gt = GroupTable();
        sel.bits.reset();
        selvec.reset();
        collect_timings(ctx);
        execute_impl(plan, table, partition_ids, n_partitions, merge_input, out, stats_override, capture, emit_partial, ctx_override,
                     cols_out, shard_req);
        return;
      }
    }
  }
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  collect_timings(ctx);
  if (fused && !plan->keys.empty()) selected = fused_count();
  if (!partial) sort_and_limit(plan, cols);
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->stats.rows_scanned += ss.rows;
    ctx->stats.rows_selected += selected;
    ctx->stats.groups_out += groups_out;
  }
  if (cols_out) {
    *cols_out = std::move(cols);
    return;
  }
  HostTimer ht(ctx, "host:export");
  export_record_batch(cols, out);
}

// One row's group keys as bytes (validity byte, then the value: 8 bytes, or length + bytes).
void append_row_key(const std::vector<OutColumn>& cols, size_t n_keys, size_t row, std::string& key) {
  key.clear();
  for (size_t c = 0; c < n_keys; ++c) {
    const OutColumn& oc = cols[c];
    const bool ok = oc.valid_data()[row] != 0;
    key.push_back(ok ? 1 : 0);
    if (!ok) continue;
    if (oc.type.is_string) {
      const std::string& v = oc.dict ? (*oc.dict)[oc.code_at(row)] : oc.str[row];
      const uint32_t l = (uint32_t)v.size();
      key.append((const char*)&l, 4);
      key.append(v);
    } else {
      const int64_t v = oc.i64_data()[row];
      key.append((const char*)&v, 8);
    }
  }
}

// COUNT(DISTINCT) next to other aggregates or over several columns, MIN / MAX over strings:
// the sub-plans see the same rows and therefore the same groups; sub-plan 0's rows (one per
// group) give the order, the others' results are looked up by group key and placed in the
// order the plan lists its aggregates.  A string sub-plan lists (group, value) pairs: the
// byte-lexicographic extremes per group are picked from them here.
// `res[k]`: sub-plan k's rows.  Sub-plan 0 and the COUNT(DISTINCT) sub-plans come as result
// rows (partial == false) or as PARTIAL state rows (the plain aggregates' state columns; one
// row per distinct (group, value) pair); the string sub-plans always as (group, value, count)
// rows.  The output is the plan's result, or its PARTIAL state: per aggregate the plain state
// columns, a List of the group's distinct values (DataFusion's state of a distinct count), or
// the smallest / largest string so far.
void mixed_join(prism_gpu_plan* plan, std::vector<std::vector<OutColumn>>& res, bool partial, ArrowArray* out) {
  const size_t n_subs = plan->subs.size(), n_keys = plan->keys.size();
  const size_t rows = res[0].empty() ? 0 : res[0][0].rows();
  std::string key;
  std::unordered_map<std::string, size_t> group_row;  // group key -> row of sub-plan 0
  if (n_keys) {
    group_row.reserve(rows * 2);
    for (size_t i = 0; i < rows; ++i) {
      append_row_key(res[0], n_keys, i, key);
      group_row.emplace(key, i);
    }
  }
  auto group_of = [&](const std::vector<OutColumn>& cols, size_t row) -> size_t {
    if (!n_keys) return 0;
    append_row_key(cols, n_keys, row, key);
    auto it = group_row.find(key);
    if (it == group_row.end()) fail(PRISM_GPU_EINTERNAL, "the parts of the plan found different groups");
    return it->second;
  };
  // per sub-plan: for every group (row of sub-plan 0) the row of the sub-plan's result, or for
  // a string sub-plan the rows holding the group's smallest and largest value; for the pairs
  // of a distinct sub-plan in PARTIAL form, the pair rows sorted by group
  constexpr size_t kNone = ~(size_t)0;
  std::vector<std::vector<size_t>> row_of(n_subs), max_row_of(n_subs), pair_start(n_subs), pair_rows(n_subs);
  for (size_t k = 1; k < n_subs; ++k) {
    const size_t n_k = res[k].empty() ? 0 : res[k][0].rows();
    row_of[k].assign(rows, kNone);
    if (plan->sub_string_col[k] < 0 && partial) {
      const OutColumn& x = res[k][n_keys];
      std::vector<size_t> group(n_k, kNone);
      pair_start[k].assign(rows + 1, 0);
      for (size_t i = 0; i < n_k; ++i) {
        if (!x.valid_data()[i]) continue;  // COUNT(DISTINCT) ignores NULLs
        group[i] = rows ? group_of(res[k], i) : kNone;
        if (group[i] != kNone) pair_start[k][group[i] + 1] += 1;
      }
      for (size_t g = 0; g < rows; ++g) pair_start[k][g + 1] += pair_start[k][g];
      pair_rows[k].resize(pair_start[k][rows]);
      std::vector<size_t> at(pair_start[k].begin(), pair_start[k].end() - 1);
      for (size_t i = 0; i < n_k; ++i)
        if (group[i] != kNone) pair_rows[k][at[group[i]]++] = i;
      continue;
    }
    if (plan->sub_string_col[k] < 0) {
      if (n_k != rows) fail(PRISM_GPU_EINTERNAL, "the parts of the plan found different groups");
      for (size_t i = 0; i < n_k; ++i) row_of[k][group_of(res[k], i)] = i;
      continue;
    }
    max_row_of[k].assign(rows, kNone);
    const OutColumn& v = res[k][n_keys];
    for (size_t i = 0; i < n_k; ++i) {
      if (!v.valid_data()[i]) continue;  // NULLs are ignored
      const size_t g = group_of(res[k], i);
      const std::string& s = v.str_at(i);  // std::string compares bytes as unsigned: byte-lexicographic
      if (row_of[k][g] == kNone || s < v.str_at(row_of[k][g])) row_of[k][g] = i;
      if (max_row_of[k][g] == kNone || s > v.str_at(max_row_of[k][g])) max_row_of[k][g] = i;
    }
  }
  std::vector<OutColumn> cols = make_columns(partial ? plan->partial_fields : plan->final_fields);
  // where sub-plan 0's aggregates start among its columns (AVG has two state columns)
  std::vector<size_t> sub0_at;
  {
    size_t at = n_keys;
    for (const AggPlan& a : plan->subs[0]->aggs) {
      sub0_at.push_back(at);
      at += partial && a.func == PRISM_AGG_AVG ? 2 : 1;
    }
  }
  size_t c = n_keys;
  for (size_t i = 0; i < plan->aggs.size(); ++i) {
    const auto [k, idx] = plan->mixed_map[i];
    const size_t width = partial && plan->aggs[i].func == PRISM_AGG_AVG ? 2 : 1;
    if (k == 0) {
      for (size_t w = 0; w < width; ++w) {
        OutColumn& dst = cols[c + w];
        FieldType t = dst.type;
        // (an aggregate listed twice maps to the same column of sub-plan 0: copy, do not move)
        dst = res[0][sub0_at[(size_t)idx] + w];
        dst.type = t;
      }
      c += width;
      continue;
    }
    OutColumn& dst = cols[c++];
    if (plan->sub_string_col[(size_t)k] >= 0) {
      const OutColumn& v = res[(size_t)k][n_keys];
      const std::vector<size_t>& pick = plan->aggs[i].func == PRISM_AGG_MIN ? row_of[(size_t)k] : max_row_of[(size_t)k];
      for (size_t r = 0; r < rows; ++r) {
        if (pick[r] == kNone) dst.push_null();
        else dst.push_str(v.str_at(pick[r]));
      }
      continue;
    }
    if (partial) {
      const OutColumn& x = res[(size_t)k][n_keys];
      OutColumn& items = *dst.items;
      for (size_t r = 0; r < rows; ++r) {
        for (size_t p = pair_start[(size_t)k][r]; p < pair_start[(size_t)k][r + 1]; ++p) {
          const size_t at = pair_rows[(size_t)k][p];
          if (x.type.is_string) items.push_str(x.str_at(at));
          else items.push_int(x.i64_data()[at]);
        }
        if (items.rows() > 0x7fffffffull) fail(PRISM_GPU_EUNSUPPORTED, "more than 2^31 distinct values in one PARTIAL batch");
        dst.list_off.push_back((int32_t)items.rows());
        dst.valid.push_back(1);
      }
      continue;
    }
    OutColumn& src = res[(size_t)k][n_keys + (size_t)idx];
    src.materialise();
    for (size_t r = 0; r < rows; ++r) {
      const size_t at = row_of[(size_t)k][r];
      if (at == kNone) fail(PRISM_GPU_EINTERNAL, "the parts of the plan found different groups");
      if (!src.valid[at]) dst.push_null();
      else if (src.type.is_float) dst.push_float(src.f64[at]);
      else dst.push_int(src.i64[at]);
    }
  }
  for (size_t kc = 0; kc < n_keys; ++kc) {
    FieldType t = cols[kc].type;
    cols[kc] = std::move(res[0][kc]);
    cols[kc].type = t;
  }
  if (!partial) sort_and_limit(plan, cols);
  export_record_batch(cols, out);
}

// COUNT(DISTINCT) next to other aggregates or over several columns, MIN / MAX over strings:
// the sub-plans see the same rows and therefore the same groups; sub-plan 0's rows (one per
// group) give the order, the others' results are looked up by group key and placed in the
// order the plan lists its aggregates.  A string sub-plan lists (group, value) pairs: the
// byte-lexicographic extremes per group are picked from them.
void execute_mixed(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions, ArrowArray* out,
                   bool emit_partial, prism_gpu_ctx* ctx_override) {
  const size_t n_subs = plan->subs.size();
  const bool partial = emit_partial || plan->mode == PRISM_MODE_PARTIAL;
  std::vector<std::vector<OutColumn>> res(n_subs);
  for (size_t k = 0; k < n_subs; ++k)
    execute_impl(plan->subs[k].get(), table, partition_ids, n_partitions, false, nullptr, nullptr, nullptr,
                 partial && plan->sub_string_col[k] < 0, ctx_override, &res[k]);
  mixed_join(plan, res, partial, out);
}

// One cell of an Arrow array of the caller's (host memory) appended to an output column.
void append_cell(OutColumn& dst, const ArrowArray* a, int64_t row) {
  const int64_t at = a->offset + row;
  const uint8_t* bits = a->n_buffers > 0 ? (const uint8_t*)a->buffers[0] : nullptr;
  if (a->null_count != 0 && bits && !((bits[at >> 3] >> (at & 7)) & 1)) { dst.push_null(); return; }
  const FieldType& f = dst.type;
  if (f.is_string) {
    const int32_t* off = (const int32_t*)a->buffers[1];
    const char* data = (const char*)a->buffers[2];
    dst.push_str(std::string(data + off[at], (size_t)(off[at + 1] - off[at])));
  } else if (f.is_float) {
    dst.push_float(((const double*)a->buffers[1])[at]);
  } else if (f.width == 8) {
    dst.push_int(((const int64_t*)a->buffers[1])[at]);
  } else if (f.width == 4) {
    dst.push_int(f.is_unsigned ? (int64_t)((const uint32_t*)a->buffers[1])[at] : (int64_t)((const int32_t*)a->buffers[1])[at]);
  } else {
    dst.push_int(f.is_unsigned ? (int64_t)((const uint16_t*)a->buffers[1])[at] : (int64_t)((const int16_t*)a->buffers[1])[at]);
  }
}

}  // namespace

// AggregateExec(Final) -- or, with emit_partial, state -> state -- over PARTIAL batches of a plan
// whose aggregates are of several kinds: every batch is cut into the PARTIAL batches of the
// sub-plans (the plain state columns as they are; a List of distinct values as one (group,
// value) row per item; the smallest and the largest string as two (group, value) rows), the
// sub-plans merge theirs on the device, and the results are joined as in execute_mixed.
void merge_mixed(prism_gpu_plan* plan, ArrowArray* const* states, size_t n_states, bool emit_partial, ArrowArray* out, prism_gpu_ctx* ctx_override) {
  prism_gpu_ctx* ctx = ctx_override ? ctx_override : plan->ctx;
  bind(ctx);
  const size_t n_subs = plan->subs.size(), n_keys = plan->keys.size();
  const std::vector<FieldType>& in_fields = plan->partial_fields;
  // the plan's aggregates -> their columns in a PARTIAL batch
  std::vector<size_t> agg_at;
  {
    size_t at = n_keys;
    for (const AggPlan& a : plan->aggs) {
      agg_at.push_back(at);
      at += a.func == PRISM_AGG_AVG ? 2 : 1;
    }
    if (at != in_fields.size()) fail(PRISM_GPU_EINTERNAL, "PARTIAL schema of the plan and its aggregates disagree");
  }
  for (size_t b = 0; b < n_states; ++b) {
    const ArrowArray* batch = states[b];
    if (!batch) fail(PRISM_GPU_EINVAL, "null state array");
    if (batch->n_children != (int64_t)in_fields.size())
      fail(PRISM_GPU_EINVAL, "state batch has " + std::to_string(batch->n_children) + " columns, the plan's PARTIAL schema has " + std::to_string(in_fields.size()));
    if (batch->offset != 0) fail(PRISM_GPU_EUNSUPPORTED, "struct arrays with a non-zero offset are not supported");
    for (int64_t cidx = 0; cidx < batch->n_children; ++cidx) {
      const ArrowArray* a = batch->children[cidx];
      if (a->dictionary) fail(PRISM_GPU_EUNSUPPORTED, "dictionary-encoded columns in the PARTIAL batches of this plan");
      if (a->length < batch->length) fail(PRISM_GPU_EINVAL, "a state column is shorter than its batch");
    }
  }
  std::vector<std::vector<OutColumn>> res(n_subs);
  std::vector<std::unique_ptr<prism_gpu_table>> tables(n_subs);  // (string columns of `res` read the tables' dictionaries)
  for (size_t k = 0; k < n_subs; ++k) {
    prism_gpu_plan* sub = plan->subs[k].get();
    const bool string_sub = plan->sub_string_col[k] >= 0;
    tables[k] = std::make_unique<prism_gpu_table>();
    prism_gpu_table* const table = tables[k].get();
    table->ctx = ctx;
    table->fields = sub->partial_fields;
    table->dicts.resize(table->fields.size());
    for (size_t b = 0; b < n_states; ++b) {
      const ArrowArray* batch = states[b];
      std::vector<OutColumn> cols = make_columns(sub->partial_fields);
      auto keys_of = [&](int64_t r) {
        for (size_t kc = 0; kc < n_keys; ++kc) append_cell(cols[kc], batch->children[kc], r);
      };
      if (k == 0) {
        // the plain aggregates' state columns, in sub-plan 0's order
        std::vector<size_t> src;  // column of the batch for every column of the sub-plan's PARTIAL schema
        for (size_t kc = 0; kc < n_keys; ++kc) src.push_back(kc);
        std::vector<size_t> sub_cols(sub->aggs.size(), ~(size_t)0);
        for (size_t i = 0; i < plan->aggs.size(); ++i)
          if (plan->mixed_map[i].first == 0) sub_cols[(size_t)plan->mixed_map[i].second] = agg_at[i];
        for (size_t j = 0; j < sub->aggs.size(); ++j) {
          if (sub_cols[j] == ~(size_t)0) {
            // (no plain aggregate in the plan: sub-plan 0 counts rows only to list the groups)
            src.push_back(~(size_t)0);
            continue;
          }
          src.push_back(sub_cols[j]);
          if (sub->aggs[j].func == PRISM_AGG_AVG) src.push_back(sub_cols[j] + 1);
        }
        if (src.size() != cols.size()) fail(PRISM_GPU_EINTERNAL, "PARTIAL schema of the plain aggregates disagrees with the plan's");
        for (int64_t r = 0; r < batch->length; ++r)
          for (size_t cidx = 0; cidx < cols.size(); ++cidx) {
            if (src[cidx] == ~(size_t)0) cols[cidx].push_int(1);
            else append_cell(cols[cidx], batch->children[src[cidx]], r);
          }
      } else if (!string_sub) {
        // every item of the group's list: one (group, value) row; a group without items still is a group
        size_t list_col = ~(size_t)0;
        for (size_t i = 0; i < plan->aggs.size() && list_col == ~(size_t)0; ++i)
          if (plan->mixed_map[i].first == (int)k) list_col = agg_at[i];
        const ArrowArray* la = batch->children[list_col];
        if (la->n_children != 1 || la->n_buffers < 2 || !la->buffers[1]) fail(PRISM_GPU_EINVAL, "the state of a COUNT(DISTINCT) must be a List column");
        const int32_t* off = (const int32_t*)la->buffers[1] + la->offset;
        const ArrowArray* items = la->children[0];
        if (items->dictionary) fail(PRISM_GPU_EUNSUPPORTED, "dictionary-encoded list items");
        const uint8_t* lbits = la->null_count != 0 && la->n_buffers > 0 ? (const uint8_t*)la->buffers[0] : nullptr;
        for (int64_t r = 0; r < batch->length; ++r) {
          const int64_t lr = la->offset + r;
          const bool has = !lbits || ((lbits[lr >> 3] >> (lr & 7)) & 1);
          const int32_t lo = has ? off[r] : 0, hi = has ? off[r + 1] : 0;
          if (lo == hi) {
            keys_of(r);
            cols[n_keys].push_null();
          }
          for (int32_t j = lo; j < hi; ++j) {
            keys_of(r);
            append_cell(cols[n_keys], items, j);
          }
        }
      } else {
        // the smallest and the largest string so far: two (group, value) rows, each counted once
        std::vector<size_t> from;
        for (size_t i = 0; i < plan->aggs.size(); ++i)
          if (plan->mixed_map[i].first == (int)k) from.push_back(agg_at[i]);
        for (int64_t r = 0; r < batch->length; ++r)
          for (size_t cidx : from) {
            keys_of(r);
            append_cell(cols[n_keys], batch->children[cidx], r);
            cols[n_keys + 1].push_int(1);
          }
      }
      ArrowArray piece;
      export_record_batch(cols, &piece);
      table_append(table, (uint32_t)b, &piece, true);
    }
    execute_impl(sub, table, nullptr, 0, true, nullptr, nullptr, nullptr, emit_partial && !string_sub, ctx_override, &res[k]);
  }
  mixed_join(plan, res, emit_partial, out);
}

namespace {

}  // namespace

// ---- the entry point that allocates nothing ------------------------------------------------
// GROUP BY over resident partitions into a dense group table that lives, with the scan
// descriptors and the counters, in a fixed device array of the library (noalloc.hip) instead
// of scratch blocks of the pool; the result columns are written to buffers of the caller.
void execute_noalloc(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                     int64_t* out_values, uint8_t* out_valid, uint64_t out_stride, uint64_t* n_groups) {
  prism_gpu_ctx* ctx = plan->ctx;
  bind(ctx);
  if (table->ctx != ctx) fail(PRISM_GPU_EINVAL, "plan and table belong to different contexts");
  if (plan->keys.empty() || plan->distinct_plan || plan->mixed || plan->mode != PRISM_MODE_SINGLE || plan->order_by >= 0 || plan->limit >= 0)
    fail(PRISM_GPU_EUNSUPPORTED, "the allocation-free entry point runs GROUP BY plans in SINGLE mode without COUNT(DISTINCT), ORDER BY or LIMIT");
  if (table->fields.size() != plan->fields.size()) fail(PRISM_GPU_EINVAL, "table schema does not match the plan's input schema");
  for (size_t i = 0; i < plan->fields.size(); ++i)
    if (table->fields[i].logical != plan->fields[i].logical)
      fail(PRISM_GPU_EINVAL, "column " + std::to_string(i) + " (`" + table->fields[i].name + "`) has a different type than the plan expects");
  // one call at a time per device: every call uses the same array
  std::lock_guard<std::mutex> one_call(noalloc_mutex(ctx->device));
  uint8_t* const arena = (uint8_t*)noalloc_scratch(ctx->device);
  table_sync_stats(table);

  ScanSet ss;
  ss.ctx = ctx;
  {
    std::lock_guard<std::mutex> chunks_lock(table->mu);  // appends may be adding chunks
    for (const Chunk& ch : table->chunks) {
      bool take = partition_ids == nullptr || n_partitions == 0;
      for (size_t i = 0; !take && i < n_partitions; ++i) take = partition_ids[i] == ch.partition_id;
      if (!take || ch.rows == 0) continue;
      if (plan->pred_root >= 0 && !chunk_may_match(plan, table, plan->pred_root, ch)) continue;
      ss.chunks.push_back(&ch);
      ss.segs.push_back({ch.rows, ss.n_tiles});
      ss.n_tiles += ceil_div(ch.rows, kTileRows);
      ss.rows += ch.rows;
    }
  }
  *n_groups = 0;
  if (ss.chunks.empty()) return;
  if (ss.n_tiles * (uint64_t)kTileRows > 0xffffffffull) fail(PRISM_GPU_EUNSUPPORTED, "more than 2^32 rows in one scan; split the scan by partition");

  std::vector<KeySrc> keys;
  for (const KeyPlan& k : plan->keys) keys.push_back({k.column, k.kind == PRISM_KEY_DATE_BIN, k.stride_ms, k.origin_ms});
  Lowered low = lower_raw(plan, ss);
  AggParams ap;
  std::memset(&ap, 0, sizeof(ap));
  KeyLayout layout;
  KeyCursor cursor;
  layout_keys(ss, table, keys, cursor, layout);
  fill_slots(table, ss, low, ap);
  const uint64_t entries = layout.words == 1 && layout.total_bits <= kDenseMaxBits ? (1ull << layout.total_bits) : 0;
  if (entries == 0 || (entries + 1) * (uint64_t)ap.n_slots > kNoAllocMaxAccumulators)
    fail(PRISM_GPU_EUNSUPPORTED, "the group table of this plan over these partitions (" +
                                     (entries ? std::to_string(entries) + " entries of " + std::to_string(ap.n_slots) + " accumulators"
                                              : std::string("a hash table")) +
                                     ") exceeds the " + std::to_string(kNoAllocMaxAccumulators) +
                                     " accumulators of the allocation-free entry point; use prism_gpu_execute");
  if (plan->pred_root >= 0 && !try_fuse_predicate(plan, table, ss, keys, layout, ap, true))
    fail(PRISM_GPU_EUNSUPPORTED, "the allocation-free entry point only takes predicates that the aggregation kernel evaluates itself");
  // ---- the array: descriptors | counters | accumulators ----
  const size_t used = ss.upload(arena, kNoAllocDescriptorBytes - 256);
  (void)used;
  unsigned long long* const d_count = (unsigned long long*)(arena + kNoAllocDescriptorBytes - 256);  // [0..1] emit, [2] rows selected, [3] flag
  int64_t* const d_state = (int64_t*)(arena + kNoAllocDescriptorBytes);
  PRISM_HIP(hipMemsetAsync(d_count, 0, 64, ctx->stream));
  for (int s = 0; s < ap.n_slots; ++s) {
    int64_t* base = d_state + (uint64_t)s * (entries + 1);
    const long long id = slot_identity_host(ap.slots[s].op);
    if (id == 0) PRISM_HIP(hipMemsetAsync(base, 0, (entries + 1) * 8, ctx->stream));
    else launch_fill_i64(ctx, base, entries + 1, id);
  }
  ap.sd = ss.desc;
  ap.cols = ss.cols_dev;
  ap.n_keys = (int)layout.specs.size();
  for (size_t i = 0; i < layout.specs.size(); ++i) ap.keys[i] = layout.specs[i];
  ap.lds_slots = lds_slots_for(ap.n_slots);
  ap.key_bits = layout.total_bits;
  ap.key_words = 1;
  ap.row_mode = kRowsAll;
  ap.selected_out = d_count + 2;
  ap.dense = 1;
  ap.capacity = entries;
  ap.g_state = d_state;
  ap.overflow = (uint32_t*)(d_count + 3);
  {
    int align = 64;  // (the lowest bit a time-typed key occupies: see execute_impl)
    for (size_t i = 0; i < keys.size(); ++i) {
      if (!(keys[i].date_bin || table->fields[keys[i].column].logical == PRISM_COL_TIMESTAMP)) continue;
      const KeyField& kf = layout.fields[i];
      if (kf.bits) align = std::min(align, kf.bit_pos);
      if (kf.null_bit >= 0) align = std::min(align, kf.null_bit);
    }
    if (align < 32) ap.window_hint = 1 + align;
  }
  uint64_t value_bytes = 0;
  for (int j = 0; j < ap.n_vals; ++j) value_bytes += ap.vals[j].width;
  for (auto& sp : layout.specs) value_bytes += sp.width;
  launch_group_by(ctx, ap, ss.rows * value_bytes);
  const EmitColumns ec = emit_columns_of(table, keys, layout, low, false);
  launch_emit_columns(ctx, true, entries, nullptr, nullptr, d_state, ec, out_values, out_valid, out_stride, d_count, nullptr, nullptr);
  unsigned long long h[3] = {0, 0, 0};
  PRISM_HIP(hipMemcpyAsync(h, d_count, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  collect_timings(ctx);
  if (h[1]) fail(PRISM_GPU_EINTERNAL, "dictionary code out of range in result");
  *n_groups = h[0];
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->stats.rows_scanned += ss.rows;
    ctx->stats.rows_selected += ap.n_pred ? h[2] : ss.rows;
    ctx->stats.groups_out += h[0];
  }
  if (h[0] > out_stride)
    fail(PRISM_GPU_EINVAL, "the result has " + std::to_string(h[0]) + " groups, the output buffers hold " + std::to_string(out_stride) +
                               " per column (nothing was written past them)");
}

namespace {

void state_compact(prism_gpu_state* st) {
  if (st->compacted) return;
  prism_gpu_ctx* ctx = st->ctx;
  const GroupTable& g = st->gt;
  Scratch d_count(ctx, 256);
  PRISM_HIP(hipMemsetAsync(d_count.p, 0, 8, ctx->stream));
  const bool two = g.key_words == 2;
  launch_emit(ctx, g.dense, g.capacity, g.keys.as<uint64_t>(), two ? g.keys_hi.as<uint64_t>() : nullptr, g.state.as<int64_t>(),
              g.n_slots, nullptr, nullptr, nullptr, 0, d_count.as<unsigned long long>());
  unsigned long long n = 0;
  PRISM_HIP(hipMemcpyAsync(&n, d_count.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  st->c_n = n;
  st->c_keys = Scratch(ctx, std::max<uint64_t>(1, n) * 8);
  st->c_keys_hi = two ? Scratch(ctx, std::max<uint64_t>(1, n) * 8) : Scratch();
  st->c_state = Scratch(ctx, std::max<uint64_t>(1, n) * 8 * g.n_slots);
  if (n) {
    PRISM_HIP(hipMemsetAsync(d_count.p, 0, 8, ctx->stream));
    launch_emit(ctx, g.dense, g.capacity, g.keys.as<uint64_t>(), two ? g.keys_hi.as<uint64_t>() : nullptr, g.state.as<int64_t>(),
                g.n_slots, st->c_keys.as<uint64_t>(), two ? st->c_keys_hi.as<uint64_t>() : nullptr, st->c_state.as<int64_t>(), n,
                d_count.as<unsigned long long>());
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
  }
  st->compacted = true;
}

MergeParams merge_params(const prism_gpu_state* st, const GroupTable& g) {
  MergeParams mp{};
  mp.dense = g.dense ? 1 : 0;
  mp.capacity = g.capacity;
  mp.g_keys = g.keys.as<uint64_t>();
  mp.g_keys_hi = g.key_words == 2 ? g.keys_hi.as<uint64_t>() : nullptr;
  mp.g_state = g.state.as<int64_t>();
  mp.overflow = g.overflow.as<uint32_t>();
  mp.n_slots = g.n_slots;
  for (int s = 0; s < g.n_slots; ++s) mp.ops[s] = st->slot_ops[s];
  return mp;
}

}  // namespace

void scan_stats(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids, size_t n_partitions,
                prism_gpu_col_stats* out) {
  bind(plan->ctx);
  if (table->fields.size() != plan->fields.size()) fail(PRISM_GPU_EINVAL, "table schema does not match the plan's input schema");
  table_sync_stats(table);
  ScanSet ss;
  ss.ctx = plan->ctx;
  {
    std::lock_guard<std::mutex> chunks_lock(table->mu);  // appends may be adding chunks
    for (const Chunk& ch : table->chunks) {
      bool take = partition_ids == nullptr || n_partitions == 0;
      for (size_t i = 0; !take && i < n_partitions; ++i) take = partition_ids[i] == ch.partition_id;
      if (take && ch.rows) ss.chunks.push_back(&ch);
    }
  }
  for (size_t c = 0; c < table->fields.size(); ++c) {
    ColInfo ci = col_info(ss, (int)c);
    out[c].min = ci.min;
    out[c].max = ci.max;
    out[c].nulls = ci.nulls;
  }
}

prism_gpu_state* aggregate_state(prism_gpu_plan* plan, prism_gpu_table* table, const uint32_t* partition_ids,
                                 size_t n_partitions, const prism_gpu_col_stats* stats, const DistinctShardRequest* shard,
                                 prism_gpu_ctx* ctx_override) {
  prism_gpu_state* st = nullptr;
  execute_impl(plan, table, partition_ids, n_partitions, false, nullptr, stats, &st, false, ctx_override, nullptr, shard);
  return st;
}

bool state_merge_state(prism_gpu_state* dst, prism_gpu_state* src) {
  prism_gpu_ctx* ctx = dst->ctx;
  bind(ctx);
  if (dst->distinct || src->distinct || dst->ctx != src->ctx || dst->gt.n_slots != src->gt.n_slots || dst->slot_ops != src->slot_ops ||
      dst->gt.key_words != src->gt.key_words)
    return false;
  if (dst->gt.dense != src->gt.dense) return false;
  if (dst->gt.dense) {
    if (dst->gt.capacity != src->gt.capacity) return false;
    // entry by entry: the source's entries are their own keys
    MergeParams mp = merge_params(dst, dst->gt);
    mp.src_keys = nullptr;
    mp.src_state = src->gt.state.as<int64_t>();
    mp.src_stride = src->gt.capacity + 1;
    mp.n = src->gt.capacity;
    launch_merge_packed(ctx, mp);
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    collect_timings(ctx);
    return true;
  }
  const StateView v = state_view(src);
  state_merge_rows(dst, v.keys, v.keys_hi, v.state, v.entries);
  return true;
}

void state_emit(prism_gpu_state* st, bool partial, ArrowArray* out) {
  if (!partial) { state_finish(st, out); return; }
  prism_gpu_plan* plan = st->plan;
  prism_gpu_ctx* ctx = st->ctx;
  bind(ctx);
  if (st->distinct) fail(PRISM_GPU_EINTERNAL, "the PARTIAL form of a COUNT(DISTINCT) state is not read from a device state");
  HostTimer whole(ctx, "host:state_finish");
  std::vector<OutColumn> cols = make_columns(plan->partial_fields);
  const uint64_t groups = emit_groups(ctx, plan, st->table, st->keys, st->layout, st->low, st->gt, true, cols);
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  collect_timings(ctx);
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->stats.groups_out += groups;
  }
  export_record_batch(cols, out);
}

DistinctShard* state_shard(prism_gpu_state* st) { return st->shard.get(); }

prism_gpu_state* distinct_shard_merge(prism_gpu_state* local, const uint64_t* records, uint64_t n_records, const uint64_t* counts64,
                                      const uint64_t* recv_start, uint32_t n_own, int world, int rank) {
  prism_gpu_plan* plan = local->plan;
  prism_gpu_ctx* ctx = local->ctx;
  bind(ctx);
  const DistinctShard* sh = local->shard.get();
  if (!sh) fail(PRISM_GPU_EINTERNAL, "not the distinct records of a sharded COUNT(DISTINCT)");
  if (n_records >= (1ull << 32)) fail(PRISM_GPU_EUNSUPPORTED, "more than 2^32 distinct records meet on one rank");
  auto st = std::make_unique<prism_gpu_state>();
  st->plan = plan;
  st->ctx = ctx;
  st->table = local->table;
  st->keys = local->keys;
  st->layout = local->layout;  // (the pairs' layout: the outer rows are unpacked relative to it)
  st->low = local->low;
  st->slot_ops = {kSlotAddOne, kSlotAddOne};
  st->distinct = 2;
  st->selected = std::max<uint64_t>(1, n_records);
  distinct_outer_alloc(ctx, st->layout, st->selected, st->gt);
  Scratch runs(ctx, (size_t)world * std::max<uint32_t>(1, n_own) * 8), d_start(ctx, (size_t)world * 8 + 64);
  uint32_t* const seg_base = runs.as<uint32_t>();
  uint32_t* const seg_cnt = seg_base + (size_t)world * n_own;
  PRISM_HIP(hipMemcpyAsync(d_start.p, recv_start, (size_t)world * 8, hipMemcpyHostToDevice, ctx->stream));
  uint32_t* const flags = (uint32_t*)(d_start.as<uint8_t>() + (size_t)world * 8);
  PRISM_HIP(hipMemsetAsync(flags, 0, 64, ctx->stream));
  if (n_own) launch_shard_runs(ctx, (const unsigned long long*)counts64, d_start.as<unsigned long long>(), seg_base, seg_cnt, n_own, (uint32_t)world);
  Part2Params q;
  std::memset(&q, 0, sizeof(q));
  q.out_lo = const_cast<uint64_t*>(records);
  q.compact = 1;
  q.lo_bits = sh->lo_bits;
  q.hi_bits = sh->hi_bits;
  q.p1_bits = (uint32_t)sh->p1;
  q.p2_bits = (uint32_t)sh->p2;
  q.direct = sh->direct ? 1 : 0;  // (only tells the kernel how a bucket's number and a record's bits make the group key)
  q.n_rows = n_records;
  q.lds_entries = sh->lds_entries;
  q.x_null_bit = -1;
  q.dense = st->gt.dense ? 1 : 0;
  q.capacity = st->gt.capacity;
  q.g_keys = st->gt.keys.as<uint64_t>();
  q.g_state = st->gt.state.as<int64_t>();
  q.overflow = st->gt.overflow.as<uint32_t>();
  q.flags = flags;
  q.n_src = (uint32_t)world;
  q.world = (uint32_t)world;
  q.rank = (uint32_t)rank;
  q.seg_base = seg_base;
  q.seg_cnt = seg_cnt;
  launch_distinct_merge(ctx, q, n_own);
  uint32_t h_flags[2] = {0, 0};
  PRISM_HIP(hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, ctx->stream));
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  collect_timings(ctx);
  if (h_flags[0] != 0 || table_overflowed(ctx, st->gt)) fail(PRISM_GPU_EINTERNAL, "the distinct records of a fine bucket did not fit its owner's set");
  return st.release();
}

void state_describe(prism_gpu_state* st, prism_gpu_state_info* out) {
  bind(st->ctx);
  if (st->distinct || st->gt.key_words != 1)
    fail(PRISM_GPU_EUNSUPPORTED, "states of COUNT(DISTINCT) plans and of keys wider than 64 bits are exchanged by "
                                 "prism_gpu_execute_sharded only: they can be finished, not exported");
  std::memset(out, 0, sizeof(*out));
  out->dense = st->gt.dense ? 1 : 0;
  out->n_slots = st->gt.n_slots;
  if (st->gt.dense) {
    out->entries = st->gt.capacity;
    out->export_bytes = st->gt.capacity * 8 * (uint64_t)st->gt.n_slots;
  } else {
    state_compact(st);
    out->entries = st->c_n;
    out->export_bytes = st->c_n * 8 * (1 + (uint64_t)st->gt.n_slots);
  }
  for (int s = 0; s < st->gt.n_slots; ++s) {
    const int op = st->slot_ops[s];
    out->slot_merge[s] = op == kSlotMin ? PRISM_SLOT_MIN_I64 : op == kSlotMax ? PRISM_SLOT_MAX_I64
                         : op == kSlotAddF64 ? PRISM_SLOT_SUM_F64 : PRISM_SLOT_SUM_I64;
  }
}

void state_export(prism_gpu_state* st, void* device_dst, size_t bytes) {
  prism_gpu_ctx* ctx = st->ctx;
  bind(ctx);
  prism_gpu_state_info info;
  state_describe(st, &info);
  if (bytes != info.export_bytes) fail(PRISM_GPU_EINVAL, "export buffer has " + std::to_string(bytes) + " bytes, the state needs " + std::to_string(info.export_bytes));
  if (bytes == 0) return;
  uint8_t* dst = (uint8_t*)device_dst;
  if (st->gt.dense) {
    const uint64_t n = st->gt.capacity;
    for (int s = 0; s < st->gt.n_slots; ++s)
      PRISM_HIP(hipMemcpyAsync(dst + (uint64_t)s * n * 8, st->gt.state.as<int64_t>() + (uint64_t)s * (n + 1), n * 8,
                               hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    PRISM_HIP(hipMemcpyAsync(dst, st->c_keys.p, st->c_n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    PRISM_HIP(hipMemcpyAsync(dst + st->c_n * 8, st->c_state.p, st->c_n * 8 * (uint64_t)st->gt.n_slots, hipMemcpyDeviceToDevice,
                             ctx->stream));
  }
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
}

void state_import(prism_gpu_state* st, const void* device_src, size_t bytes) {
  prism_gpu_ctx* ctx = st->ctx;
  bind(ctx);
  if (!st->gt.dense) fail(PRISM_GPU_EINVAL, "only a dense state can be replaced; merge rows into a hash state instead");
  const uint64_t n = st->gt.capacity;
  if (bytes != n * 8 * (uint64_t)st->gt.n_slots) fail(PRISM_GPU_EINVAL, "import buffer size does not match the state");
  const uint8_t* src = (const uint8_t*)device_src;
  for (int s = 0; s < st->gt.n_slots; ++s)
    PRISM_HIP(hipMemcpyAsync(st->gt.state.as<int64_t>() + (uint64_t)s * (n + 1), src + (uint64_t)s * n * 8, n * 8,
                             hipMemcpyDeviceToDevice, ctx->stream));
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
}

void state_merge(prism_gpu_state* st, const void* device_src, uint64_t n_groups) {
  if (st->distinct || st->gt.key_words != 1)
    fail(PRISM_GPU_EUNSUPPORTED, "rows can be merged into states of plain GROUP BY plans with keys of at most 64 bits");
  state_merge_rows(st, (const uint64_t*)device_src, nullptr, (const int64_t*)device_src + n_groups, n_groups);
}

void state_merge_rows(prism_gpu_state* st, const uint64_t* src_keys, const uint64_t* src_keys_hi, const int64_t* src_state,
                      uint64_t n_groups) {
  prism_gpu_ctx* ctx = st->ctx;
  bind(ctx);
  if (n_groups == 0) return;
  if ((st->gt.key_words == 2) != (src_keys_hi != nullptr)) fail(PRISM_GPU_EINTERNAL, "merged rows and state differ in key width");
  if (!st->gt.dense) {
    // room for every incoming group being new; otherwise rebuild a larger table first
    state_compact(st);
    const uint64_t need = st->c_n + n_groups;
    if (st->gt.list || need * 2 > st->gt.capacity) {
      GroupTable bigger;
      table_alloc(ctx, bigger, false, pow2_ceil(std::max<uint64_t>(1024, need * 2)), st->slot_ops, st->gt.key_words);
      MergeParams own = merge_params(st, bigger);
      own.src_keys = st->c_keys.as<uint64_t>();
      own.src_keys_hi = st->gt.key_words == 2 ? st->c_keys_hi.as<uint64_t>() : nullptr;
      own.src_state = st->c_state.as<int64_t>();
      own.n = st->c_n;
      launch_merge_packed(ctx, own);
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
      st->gt = std::move(bigger);
    }
  }
  MergeParams mp = merge_params(st, st->gt);
  mp.src_keys = src_keys;
  mp.src_keys_hi = src_keys_hi;
  mp.src_state = src_state;
  mp.n = n_groups;
  launch_merge_packed(ctx, mp);
  if (table_overflowed(ctx, st->gt)) {
    PRISM_HIP(hipMemsetAsync(st->gt.overflow.p, 0, 4, ctx->stream));
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    fail(st->gt.dense ? PRISM_GPU_EINVAL : PRISM_GPU_EINTERNAL,
         st->gt.dense ? "merged keys fall outside the dense table: the states were not built from the same column statistics"
                      : "group table overflow while merging states");
  }
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  collect_timings(ctx);
  st->compacted = false;  // the compact copy no longer describes the table
}

void state_finish(prism_gpu_state* st, ArrowArray* out) {
  prism_gpu_plan* plan = st->plan;
  prism_gpu_ctx* ctx = st->ctx;
  bind(ctx);
  HostTimer whole(ctx, "host:state_finish");
  std::vector<OutColumn> cols = make_columns(plan->final_fields);
  uint64_t groups;
  if (st->distinct == 1) {
    GroupTable q;
    distinct_fold_outer(ctx, st->layout, st->selected, st->gt, q);
    groups = distinct_emit_outer(ctx, plan, st->table, st->keys, st->layout, q, cols);
  } else if (st->distinct == 2) {
    groups = distinct_emit_outer(ctx, plan, st->table, st->keys, st->layout, st->gt, cols);
  } else {
    groups = emit_groups(ctx, plan, st->table, st->keys, st->layout, st->low, st->gt, false, cols);
  }
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  collect_timings(ctx);
  sort_and_limit(plan, cols);
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->stats.groups_out += groups;
  }
  export_record_batch(cols, out);
}

void state_free(prism_gpu_state* st) {
  if (st) bind(st->ctx);
  delete st;
}

prism_gpu_state* state_distinct_outer(prism_gpu_state* inner) {
  if (inner->distinct != 1) fail(PRISM_GPU_EINTERNAL, "not the pair set of a COUNT(DISTINCT) plan");
  prism_gpu_ctx* ctx = inner->ctx;
  bind(ctx);
  auto st = std::make_unique<prism_gpu_state>();
  st->plan = inner->plan;
  st->ctx = inner->ctx;
  st->table = inner->table;
  st->keys = inner->keys;
  st->layout = inner->layout;  // (the pairs' layout: the outer rows are unpacked relative to it)
  st->low = inner->low;
  st->slot_ops = {kSlotAddOne, kSlotAddOne};
  st->distinct = 2;
  st->selected = inner->selected;
  distinct_fold_outer(ctx, inner->layout, inner->selected, inner->gt, st->gt);
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  collect_timings(ctx);
  return st.release();
}

StateView state_view(prism_gpu_state* st) {
  bind(st->ctx);
  StateView v;
  v.plan = st->plan;
  v.dense = st->gt.dense;
  v.n_slots = st->gt.n_slots;
  v.slot_ops = st->slot_ops;
  v.key_words = st->gt.key_words;
  v.distinct = st->distinct;
  if (st->distinct == 1) v.pair_shift = distinct_shape(st->layout).x_bits;
  if (v.dense) {
    v.entries = st->gt.capacity;
    v.dense_state = st->gt.state.as<int64_t>();
    v.dense_stride = st->gt.capacity + 1;
  } else {
    state_compact(st);
    v.entries = st->c_n;
    v.keys = st->c_keys.as<uint64_t>();
    v.keys_hi = st->gt.key_words == 2 ? st->c_keys_hi.as<uint64_t>() : nullptr;
    v.state = st->c_state.as<int64_t>();
  }
  return v;
}

uint64_t state_leading_rows(prism_gpu_state* st, Scratch& out_keys, Scratch& out_keys_hi, Scratch& out_state) {
  prism_gpu_plan* plan = st->plan;
  prism_gpu_ctx* ctx = st->ctx;
  bind(ctx);
  const GroupTable& g = st->gt;
  if (g.dense) fail(PRISM_GPU_EINTERNAL, "leading rows are read from hash states");
  const uint64_t entries = g.capacity + 1;
  TopKOrder order;
  std::vector<OutColumn> cols = make_columns(plan->final_fields);
  // (the device ranks groups of plain one-word states; the others send every group)
  if (st->distinct || g.key_words != 1 ||
      !(entries >= kTopKMinEntries && topk_order(plan, st->table, st->keys, st->layout, st->low, cols, order))) {
    state_compact(st);
    out_keys = Scratch(ctx, std::max<uint64_t>(1, st->c_n) * 8);
    out_state = Scratch(ctx, std::max<uint64_t>(1, st->c_n) * 8 * (uint64_t)g.n_slots);
    if (g.key_words == 2) {
      out_keys_hi = Scratch(ctx, std::max<uint64_t>(1, st->c_n) * 8);
      if (st->c_n) PRISM_HIP(hipMemcpyAsync(out_keys_hi.p, st->c_keys_hi.p, st->c_n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (st->c_n) {
      PRISM_HIP(hipMemcpyAsync(out_keys.p, st->c_keys.p, st->c_n * 8, hipMemcpyDeviceToDevice, ctx->stream));
      PRISM_HIP(hipMemcpyAsync(out_state.p, st->c_state.p, st->c_n * 8 * (uint64_t)g.n_slots, hipMemcpyDeviceToDevice, ctx->stream));
      PRISM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return st->c_n;
  }
  // at most the limit plus the entries of the last radix bin pass the threshold
  const uint64_t room = (uint64_t)std::max<int64_t>(1, plan->limit) + 4096;
  Scratch d_topk(ctx, sizeof(TopKState)), d_count(ctx, 256);
  launch_topk_threshold(ctx, false, g.capacity, g.keys.as<uint64_t>(), nullptr, g.state.as<int64_t>(), order,
                        (uint64_t)std::max<int64_t>(1, plan->limit), d_topk.as<TopKState>());
  out_keys = Scratch(ctx, room * 8);
  out_state = Scratch(ctx, room * 8 * (uint64_t)g.n_slots);
  PRISM_HIP(hipMemsetAsync(d_count.p, 0, 8, ctx->stream));
  launch_emit(ctx, false, g.capacity, g.keys.as<uint64_t>(), nullptr, g.state.as<int64_t>(), g.n_slots, out_keys.as<uint64_t>(),
              nullptr, out_state.as<int64_t>(), room, d_count.as<unsigned long long>(), &order, d_topk.as<TopKState>(), room);
  unsigned long long n = 0;
  PRISM_HIP(hipMemcpyAsync(&n, d_count.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  PRISM_HIP(hipStreamSynchronize(ctx->stream));
  collect_timings(ctx);
  if (n > room) fail(PRISM_GPU_EINTERNAL, "more leading groups than the radix select admits");
  // columns are `room` apart: close them up to n
  if (n && n < room) {
    Scratch packed(ctx, n * 8 * (uint64_t)g.n_slots);
    PRISM_HIP(hipMemcpy2DAsync(packed.p, n * 8, out_state.p, room * 8, n * 8, (size_t)g.n_slots, hipMemcpyDeviceToDevice, ctx->stream));
    PRISM_HIP(hipStreamSynchronize(ctx->stream));
    out_state = std::move(packed);
  }
  return n;
}

}  // namespace prism
