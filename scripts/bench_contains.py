#!/usr/bin/env python3
"""Substring search (LIKE '%needle%' on a column without a dictionary) on the C4 table, for
needles of different lengths and match densities, with and without the window variant."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prism_amd import workloads
from prism_amd.engine import Context, Plan
from prism_amd.plan import QueryPlan, col, count_star

ctx = Context(0)
wl = workloads.c4()
rows = 100_000_000
table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, 100)
for needle in ["google", "zzgle.q", "goog", "oogle.bi", "zq", "e", "zzzzzzzzzzzz"]:
    for windows in (0, 1):
        ctx.set_option("contains_windows", windows)
        plan = Plan(ctx, wl.schema, QueryPlan([count_star()], predicate=col("URL").like(f"%{needle}%")))
        for _ in range(2):
            r = plan.execute(table)
        ctx.reset_stats(); ctx.set_profiling(True)
        for _ in range(5):
            r = plan.execute(table)
        ctx.set_profiling(False)
        k = ctx.stats()["kernels"]["like_contains"]
        print(f"needle {needle!r:16} windows={windows} matches={r.column(0)[0].as_py():>10} like_contains avg {k['total_ms']/k['launches']:.3f} ms", flush=True)
        plan.close()
table.close(); ctx.close()
