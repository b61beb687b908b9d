#!/usr/bin/env python3
"""Randomised parity: random tables, plans, partitionings and execution routes (single,
PARTIAL -> merge, streaming, device-resident state), GPU against the CPU oracle.
usage: [FUZZ_OPTIONS=name=value,...] python scripts/fuzz_parity.py [iterations] [seed]   (needs a GPU)
Prints one line per failure with the seed that reproduces it; exit code 1 if any."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from oracle import pyoracle  # noqa: E402
from prism_amd import _cabi as abi  # noqa: E402
from prism_amd.engine import Context, Plan, PrismGpuError, Table  # noqa: E402
from prism_amd.plan import (QueryPlan, avg, col, count, count_distinct, count_star, date_bin, max_, min_,  # noqa: E402
                            sum_)
from tests import helpers  # noqa: E402

INT_COLS = ["value", "i16", "i32", "u16", "user", "ts"]
RANGES = {"value": (-50_000, 50_000), "i16": (0, 5), "i32": (-1000, 1000), "u16": (0, 65535),
          "ts": (1_700_000_000_000, 1_700_000_000_000 + 3 * 3600 * 1000)}


SCHEMA = helpers.random_table(100, seed=0).schema


def rand_leaf(rng, t):
    kind = rng.integers(0, 10)
    if kind < 5:
        c = INT_COLS[rng.integers(0, len(INT_COLS))]
        if c == "user":
            vals = [v for v in t.column("user").to_pylist() if v is not None]
            litv = int(vals[rng.integers(0, len(vals))]) if vals else 0
        else:
            lo, hi = RANGES[c]
            litv = int(rng.integers(lo - 2, hi + 2))
        op = rng.integers(0, 6)
        x = col(c)
        return [x == litv, x != litv, x < litv, x <= litv, x > litv, x >= litv][op]
    if kind < 6:
        c = ["value", "i32", "service", "url", "user"][rng.integers(0, 5)]
        return col(c).is_null() if rng.random() < 0.5 else col(c).is_not_null()
    if kind < 8:
        s = f"svc-{rng.integers(0, 13):02d}"
        x = col("service")
        return [x == s, x != s, x < s, x >= s, x.like("svc-0%"), x.like("%1")][rng.integers(0, 6)]
    pats = ["%google%", "%gogle%", "http://goog%", "%admin", "%/x/%", "%g_ogl%", "http://%/%/%", "%", "%bing%example%"]
    p = pats[rng.integers(0, len(pats))]
    return col("url").like(p) if rng.random() < 0.7 else col("url").not_like(p)


def rand_pred(rng, t, depth=0):
    if depth >= 3 or rng.random() < 0.45:
        return rand_leaf(rng, t)
    r = rng.random()
    if r < 0.4:
        return rand_pred(rng, t, depth + 1) & rand_pred(rng, t, depth + 1)
    if r < 0.8:
        return rand_pred(rng, t, depth + 1) | rand_pred(rng, t, depth + 1)
    return ~rand_pred(rng, t, depth + 1)


def rand_plan(rng, t):
    pred = rand_pred(rng, t) if rng.random() < 0.6 else None
    n_keys = int(rng.choice([0, 1, 1, 2, 2, 3]))
    keys, names = [], set()
    pool = ["service", "i16", "i32", "u16", "user", "value", "ts", "bin"]
    while len(keys) < n_keys:
        k = pool[rng.integers(0, len(pool))]
        if k in names:
            continue
        names.add(k)
        if k == "bin":
            stride = int(rng.choice([1000, 60_000, 3_600_000, 7, 86_400_000]))
            origin = int(rng.choice([0, 1_700_000_001_234, 1_700_010_000_000, -5]))
            keys.append(date_bin(stride, "ts", origin_ms=origin, name="bin"))
        else:
            keys.append(k)
    if rng.random() < 0.12:
        target = ["user", "service", "i32"][rng.integers(0, 3)]
        if target not in names:
            return QueryPlan([count_distinct(target)], keys=keys, predicate=pred), len(keys)
    fns = [lambda: count_star(), lambda: count(["value", "i32", "service"][rng.integers(0, 3)]),
           lambda: sum_(["value", "i16", "i32", "u16"][rng.integers(0, 4)]),
           lambda: min_(["value", "i16", "i32", "u16", "ts"][rng.integers(0, 5)]),
           lambda: max_(["value", "i16", "i32", "u16", "ts"][rng.integers(0, 5)]),
           lambda: avg(["value", "i32", "u16"][rng.integers(0, 3)])]
    aggs, seen = [], set()
    for _ in range(int(rng.integers(1, 6))):
        a = fns[rng.integers(0, len(fns))]()
        sig = (a.func, a.column)
        if sig not in seen:
            seen.add(sig)
            aggs.append(a)
    order_by = limit = None
    desc = False
    if keys and rng.random() < 0.25:
        order_by = int(rng.integers(0, len(keys) + len(aggs)))
        desc = bool(rng.random() < 0.6)
        limit = int(rng.choice([1, 3, 10, 50, 100_000]))
    return QueryPlan(aggs, keys=keys, predicate=pred, order_by=order_by, descending=desc, limit=limit), len(keys)


def check(got, want, q, n_keys):
    if q.limit is not None:
        helpers.assert_topk_match(got, want, q.order_by)
    else:
        helpers.assert_tables_match(got, want, n_keys)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = Context(0)
    # FUZZ_OPTIONS="part_force=1,part_min_rows=0": context options for the whole run (say, every
    # GROUP BY through the partitioned kernels)
    for opt in filter(None, os.environ.get("FUZZ_OPTIONS", "").split(",")):
        name, _, value = opt.partition("=")
        ctx.set_option(name, int(value))
    failures = 0
    unsupported = 0
    for it in range(iters):
        seed = seed0 * 100_000 + it
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1, 63, 64, 65, 4095, 4096, 4097, 10_000, 40_000, 120_000]))
        t = helpers.random_table(n, seed=seed, nulls=bool(rng.random() < 0.8)).cast(SCHEMA)  # (an all-NULL column would be typed null)
        parts = int(rng.integers(1, 6))
        g = Table(ctx, t.schema)
        try:
            g.append_table(t, parts)
            for _ in range(4):
                q, n_keys = rand_plan(rng, t)
                route = ["single", "partial", "stream", "state"][rng.integers(0, 4)]
                try:
                    want = pyoracle.execute(t, q)
                    if route == "single" or (route != "state" and q.limit is not None):
                        plan = Plan(ctx, t.schema, q)
                        try:
                            got = plan.execute(g)
                        finally:
                            plan.close()
                    elif route == "partial":
                        plan = Plan(ctx, t.schema, q.with_mode(abi.MODE_PARTIAL))
                        try:
                            got = plan.merge([plan.execute(g, [p]) for p in range(parts)])
                        finally:
                            plan.close()
                    elif route == "stream":
                        plan = Plan(ctx, t.schema, q)
                        try:
                            got = plan.run_batches(0, t.to_batches(max_chunksize=int(rng.integers(1, 5000))))
                        finally:
                            plan.close()
                    else:
                        plan = Plan(ctx, t.schema, q)
                        try:
                            try:
                                stats = plan.scan_stats(g)
                                states = [plan.aggregate(g, [p], stats) for p in range(parts)]
                            except PrismGpuError as e:
                                if e.code != abi.EUNSUPPORTED:
                                    raise
                                got = plan.execute(g)
                                states = []
                            if states:
                                infos = [s.describe() for s in states]
                                if infos[0].dense:
                                    bufs = []
                                    for s, info in zip(states, infos):
                                        b = torch.zeros(max(1, info.export_bytes // 8), dtype=torch.int64, device="cuda:0")
                                        torch.cuda.synchronize()
                                        s.export_to(b.data_ptr(), info.export_bytes)
                                        bufs.append(b)
                                    acc, e = bufs[0], infos[0].entries
                                    for b in bufs[1:]:
                                        for sl in range(infos[0].n_slots):
                                            v, w = acc[sl * e:(sl + 1) * e], b[sl * e:(sl + 1) * e]
                                            m = infos[0].slot_merge[sl]
                                            if m == abi.SLOT_SUM_I64:
                                                v += w
                                            elif m == abi.SLOT_SUM_F64:
                                                v.view(torch.float64).add_(w.view(torch.float64))
                                            elif m == abi.SLOT_MIN_I64:
                                                torch.minimum(v, w, out=v)
                                            else:
                                                torch.maximum(v, w, out=v)
                                    torch.cuda.synchronize()
                                    states[0].import_from(acc.data_ptr(), infos[0].export_bytes)
                                else:
                                    for s, info in zip(states[1:], infos[1:]):
                                        b = torch.zeros(max(1, info.export_bytes // 8), dtype=torch.int64, device="cuda:0")
                                        torch.cuda.synchronize()
                                        s.export_to(b.data_ptr(), info.export_bytes)
                                        states[0].merge_from(b.data_ptr(), info.entries)
                                got = states[0].finish()
                                for s in states:
                                    s.close()
                        finally:
                            plan.close()
                    check(got, want, q, n_keys)
                except PrismGpuError as e:
                    if e.code == abi.EUNSUPPORTED:
                        unsupported += 1
                        continue
                    failures += 1
                    print(f"FAIL seed={seed} n={n} parts={parts} route={route}: {e}\n  plan={q}", flush=True)
                except pyoracle.OracleError as e:
                    if e.code == abi.EUNSUPPORTED:
                        unsupported += 1
                        continue
                    failures += 1
                    print(f"ORACLE-FAIL seed={seed}: {e}\n  plan={q}", flush=True)
                except AssertionError as e:
                    failures += 1
                    print(f"MISMATCH seed={seed} n={n} parts={parts} route={route}: {str(e)[:300]}\n  plan={q}", flush=True)
                except Exception:  # noqa: BLE001
                    failures += 1
                    print(f"ERROR seed={seed} n={n} parts={parts} route={route}\n  plan={q}\n{traceback.format_exc()}", flush=True)
        finally:
            g.close()
        if it % 20 == 19:
            print(f"[fuzz] {it + 1}/{iters} tables, {failures} failures, {unsupported} unsupported", flush=True)
    ctx.close()
    print(f"[fuzz] done: {iters} tables x 4 plans, {failures} failures, {unsupported} unsupported")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
