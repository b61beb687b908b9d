// Device code shared by the aggregation kernels: slot arithmetic, the date_bin bucket
// computation, and insertion into the HBM group table.
#pragma once

#include "device_utils.hpp"

namespace prism {

constexpr long long kI64Max = 0x7fffffffffffffffll;
constexpr long long kI64Min = -kI64Max - 1;

__device__ __forceinline__ long long slot_identity(int op) {
  return op == kSlotMin ? kI64Max : (op == kSlotMax ? kI64Min : 0ll);
}

// floor(n / d) for the divisor described by (magic, shift); shift < 0 means d == 1.
__device__ __forceinline__ uint64_t fast_div(uint64_t n, uint64_t magic, int shift) {
  if (shift < 0) return n;
  uint64_t q = __umul64hi(magic, n);
  uint64_t t = ((n - q) >> 1) + q;
  return t >> shift;
}

// Bucket number of a timestamp: DataFusion's date_bin rounds toward zero and then
// steps one more stride down for timestamps before the origin.
__device__ __forceinline__ long long bucket_number(long long ts, const KeySpec& k) {
  long long d = (long long)((uint64_t)ts - (uint64_t)k.origin);
  uint64_t a = d < 0 ? (0ull - (uint64_t)d) : (uint64_t)d;
  long long q = (long long)fast_div(a, k.magic, k.magic_shift);
  if (d < 0) q = -q - (k.stride > 1 ? 1 : 0);
  return q;
}

// Probes before an insertion gives up and reports overflow: far beyond any chain of a
// table at load <= 0.5, yet small enough that a pass over a table that turned out too
// small (it is re-run with a larger one) does not crawl.
constexpr uint64_t kMaxProbes = 512;

// Set once an insertion failed: the pass will be re-run, so kernels stop early.
__device__ __forceinline__ bool table_gave_up(const AggParams& p) {
  return __hip_atomic_load(p.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

template <int N>
struct RowValsT {
  long long v[N > 0 ? N : 1];
  uint32_t valid;  // bit j: value j is not NULL
};
using RowVals = RowValsT<kMaxVals>;

// Contribution of one row to slot s: returns false when the row does not touch it.
template <int N>
__device__ __forceinline__ bool slot_input(const AggParams& p, int s, const RowValsT<N>& rv, long long& out) {
  const SlotSpec& sl = p.slots[s];
  if (sl.op == kSlotAddOne) { out = 1; return true; }
  if (!((rv.valid >> sl.val) & 1u)) return false;
  long long v = 0;
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (j == sl.val) v = rv.v[j];  // select chain: a runtime index would put rv.v in scratch memory
  if (sl.op == kSlotAddValid) out = 1;
  else if (sl.op == kSlotAddF64) out = p.vals[sl.val].is_float ? v : __double_as_longlong((double)v);
  else out = v;
  return true;
}

template <typename P>
__device__ __forceinline__ void atomic_apply(P* addr, int op, long long x) {
  if (op == kSlotMin) atomicMin((long long*)addr, x);
  else if (op == kSlotMax) atomicMax((long long*)addr, x);
  else if (op == kSlotAddF64) atomicAdd((double*)addr, __longlong_as_double(x));
  else atomicAdd((unsigned long long*)addr, (unsigned long long)x);
}

// Entry of `key` in the HBM table, inserting it if absent; capacity (the extra
// entry) for the all-ones key; ~0 when the table is full.
__device__ __forceinline__ uint64_t global_entry_raw(int dense, uint64_t capacity, uint64_t* g_keys, uint32_t* overflow,
                                                     uint64_t key) {
  if (dense) return key;
  if (key == kEmptyKey) return capacity;
  uint64_t mask = capacity - 1;
  uint64_t h = hash64(key) & mask;
  const uint64_t max_probes = capacity < kMaxProbes ? capacity : kMaxProbes;
  for (uint64_t probe = 0; probe < max_probes; ++probe) {
    unsigned long long k = g_keys[h];
    if (k == key) return h;
    if (k == kEmptyKey) {
      unsigned long long old = atomicCAS((unsigned long long*)&g_keys[h], kEmptyKey, key);
      if (old == kEmptyKey || old == key) return h;
    }
    h = (h + 1) & mask;
  }
  *overflow = 1u;
  return ~0ull;
}

__device__ __forceinline__ uint64_t global_entry(const AggParams& p, uint64_t key) {
  return global_entry_raw(p.dense, p.capacity, p.g_keys, p.overflow, key);
}

// Two-word keys.  The low word claims the entry with a CAS; the winner then publishes
// the high word, which starts as all ones ("not written yet"; the planner keeps both
// words below all ones).  A lane that finds its low word in an entry whose high word
// is still unwritten simply runs the same probe again: the winner's store sits earlier
// in the loop body than that re-read, so lanes of one wave cannot wait on each other
// forever, and device-scope loads and stores make it visible across CUs.
__device__ __forceinline__ uint64_t global_entry2(const AggParams& p, uint64_t lo, uint64_t hi) {
  const uint64_t mask = p.capacity - 1;
  uint64_t h = hash64(lo ^ hash64(hi)) & mask;
  const uint64_t max_probes = p.capacity < kMaxProbes ? p.capacity : kMaxProbes;
  uint64_t probes = 0, result = ~0ull;
  bool done = false;
  while (!done) {
    bool retry = false;
    unsigned long long k = __hip_atomic_load((unsigned long long*)&p.g_keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == kEmptyKey) {
      unsigned long long old = atomicCAS((unsigned long long*)&p.g_keys[h], kEmptyKey, lo);
      if (old == kEmptyKey) {
        __hip_atomic_store((unsigned long long*)&p.g_keys_hi[h], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        result = h;
        done = true;
      }
      k = old;
    }
    if (!done && k == lo) {
      unsigned long long kh = __hip_atomic_load((unsigned long long*)&p.g_keys_hi[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (kh == hi) { result = h; done = true; }
      else if (kh == kEmptyKey) retry = true;
    }
    if (!done && !retry) {
      h = (h + 1) & mask;
      if (++probes >= max_probes) { *p.overflow = 1u; done = true; }
    }
  }
  return result;
}

template <int N>
__device__ __forceinline__ void global_update(const AggParams& p, uint64_t key, const RowValsT<N>& rv) {
  uint64_t e = global_entry(p, key);
  if (e == ~0ull) return;
  for (int s = 0; s < p.n_slots; ++s) {
    long long x;
    if (slot_input(p, s, rv, x)) atomic_apply(&p.g_state[(uint64_t)s * (p.capacity + 1) + e], p.slots[s].op, x);
  }
}

}  // namespace prism
