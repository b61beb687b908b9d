#!/usr/bin/env python3
"""Generates tests/golden/rand_small.arrow and tests/golden/rand_small_expected.json.

The expected results are computed WITHOUT the oracle and without the GPU library, by
engines that are independent of both: pyarrow.compute / Table.group_by and numpy.
They pin the SQL semantics the oracle restates (three-valued filters, NULL group keys,
NULL-skipping aggregates, date_bin flooring, LIKE, COUNT DISTINCT); the reference's own
tests pin only ungrouped MIN/MAX (src/prism-ingest/src/ingest.rs:224-226).

Run from the repository root:  python scripts/make_golden.py
"""
import json
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd import plan as P  # noqa: E402
from tests import helpers  # noqa: E402


def eval_pred(t: pa.Table, p):
    """Predicate -> nullable boolean array with Kleene logic (pyarrow kernels)."""
    if isinstance(p, P._Bool):
        if p.kind == abi.EXPR_NOT:
            return pc.invert(eval_pred(t, p.lhs))
        a, b = eval_pred(t, p.lhs), eval_pred(t, p.rhs)
        return pc.and_kleene(a, b) if p.kind == abi.EXPR_AND else pc.or_kleene(a, b)
    if isinstance(p, P._Cmp):
        c = t.column(p.column)
        lit = p.literal
        if pa.types.is_timestamp(c.type):
            c = c.cast(pa.int64())
        fn = {abi.CMP_EQ: pc.equal, abi.CMP_NEQ: pc.not_equal, abi.CMP_LT: pc.less, abi.CMP_LTEQ: pc.less_equal,
              abi.CMP_GT: pc.greater, abi.CMP_GTEQ: pc.greater_equal}[p.op]
        if isinstance(lit, int):
            return fn(c.cast(pa.int64()), pa.scalar(lit, pa.int64()))
        return fn(c, pa.scalar(lit))
    if isinstance(p, P._Like):
        m = pc.match_like(t.column(p.column), p.pattern, ignore_case=p.case_insensitive)
        return pc.invert(m) if p.negated else m
    if isinstance(p, P._IsNull):
        v = pc.is_null(t.column(p.column))
        return pc.invert(v) if p.negated else v
    raise TypeError(p)


def run(t: pa.Table, q: P.QueryPlan):
    """Rows (list of lists, JSON friendly) of the query result, independent of the oracle."""
    if q.predicate is not None:
        mask = pc.fill_null(eval_pred(t, q.predicate), False)  # FilterExec drops NULL and FALSE
        t = t.filter(mask)
    key_names = []
    for i, k in enumerate(q.keys):
        k = P.GroupKey(k) if isinstance(k, str) else k
        name = f"__k{i}"
        col = t.column(k.column)
        if k.kind == abi.KEY_DATE_BIN:
            ts = col.cast(pa.int64()).to_numpy(zero_copy_only=False).astype(np.int64)
            d = ts - k.origin_ms
            b = d - np.fmod(d, k.stride_ms)           # truncating remainder, like Rust's %
            b = np.where((d < 0) & (k.stride_ms > 1), b - k.stride_ms, b) + k.origin_ms
            col = pa.array(b, pa.int64())
        elif pa.types.is_timestamp(col.type):
            col = col.cast(pa.int64())
        t = t.append_column(name, col)
        key_names.append(name)
    aggs, post = [], []
    for a in q.aggs:
        c = a.column
        if c is not None and pa.types.is_timestamp(t.column(c).type):
            t = t.set_column(t.schema.get_field_index(c), c, t.column(c).cast(pa.int64()))
        if a.func == abi.AGG_COUNT_STAR:
            aggs.append(([], "count_all")); post.append("count_all")
        elif a.func == abi.AGG_COUNT:
            aggs.append((c, "count")); post.append(f"{c}_count")
        elif a.func == abi.AGG_SUM:
            aggs.append((c, "sum")); post.append(f"{c}_sum")
        elif a.func == abi.AGG_MIN:
            aggs.append((c, "min")); post.append(f"{c}_min")
        elif a.func == abi.AGG_MAX:
            aggs.append((c, "max")); post.append(f"{c}_max")
        elif a.func == abi.AGG_AVG:
            aggs.append((c, "mean")); post.append(f"{c}_mean")
        elif a.func == abi.AGG_COUNT_DISTINCT:
            aggs.append((c, "count_distinct", pc.CountOptions(mode="only_valid"))); post.append(f"{c}_count_distinct")
    g = pa.TableGroupBy(t, key_names, use_threads=False).aggregate(aggs)
    if not key_names and t.num_rows == 0 and g.num_rows == 0:
        raise AssertionError("pyarrow returned no row for an ungrouped aggregate")
    rows = []
    for i in range(g.num_rows):
        row = [g.column(k)[i].as_py() for k in key_names] + [g.column(p)[i].as_py() for p in post]
        rows.append(row)
    return rows


def main():
    t = helpers.random_table(2000, seed=42)
    out = os.path.join(ROOT, "tests", "golden")
    with pa.OSFile(os.path.join(out, "rand_small.arrow"), "wb") as f:
        with pa.ipc.new_file(f, t.schema) as w:
            w.write_table(t)
    expected = {}
    for name, q, n_keys in helpers.plan_catalogue():
        expected[name] = {"n_keys": n_keys, "rows": run(t, q)}
    with open(os.path.join(out, "rand_small_expected.json"), "w") as f:
        json.dump(expected, f, indent=0, sort_keys=True)
    print("wrote", len(expected), "expected results")


if __name__ == "__main__":
    main()
