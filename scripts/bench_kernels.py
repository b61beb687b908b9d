#!/usr/bin/env python3
"""Per-kernel efficiency over a set of query shapes on the C2 table (HBM resident):
prints, per query, the step time and every kernel's average duration with the GB/s of
its algorithmic bytes.  usage: python scripts/bench_kernels.py [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import time  # noqa: E402

import torch  # noqa: E402,F401  (loads the HIP runtime torch ships before the library binds to one)

from prism_amd import workloads  # noqa: E402
from prism_amd.engine import Context, Plan  # noqa: E402
from prism_amd.plan import QueryPlan, avg, col, count, count_star, date_bin, max_, min_, sum_  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
    wl = workloads.WORKLOADS["C2"]()
    ctx = Context(0)
    table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, 32)
    queries = [
        ("COUNT(*) WHERE value >= 49000", QueryPlan([count_star()], predicate=col("value") >= 49_000)),
        ("COUNT(*) WHERE 10000 <= value < 30000", QueryPlan([count_star()], predicate=(col("value") >= 10_000) & (col("value") < 30_000))),
        ("SUM(value) WHERE timestamp >= noon + 1 h", QueryPlan([sum_("value")], predicate=col("timestamp") >= workloads.T0_MS + 13 * 3_600_000)),
        ("SUM MIN MAX(value) ungrouped", QueryPlan([sum_("value"), min_("value"), max_("value")])),
        ("AVG(value) ungrouped", QueryPlan([avg("value")])),
        ("GROUP BY service: COUNT SUM", QueryPlan([count_star(), sum_("value")], keys=["service"])),
        ("C2", wl.query),
        ("C2 + AVG MIN", QueryPlan([count_star(), sum_("value"), max_("value"), min_("value"), avg("value")],
                                   keys=[date_bin(60_000, "timestamp"), "service"])),
        ("C2 WHERE value >= 0 (100 %)", QueryPlan(wl.query.aggs, keys=wl.query.keys, predicate=col("value") >= 0)),
        ("C2 WHERE value >= 25000 (50 %)", QueryPlan(wl.query.aggs, keys=wl.query.keys, predicate=col("value") >= 25_000)),
        ("C2 WHERE value >= 49900 (0.2 %)", QueryPlan(wl.query.aggs, keys=wl.query.keys, predicate=col("value") >= 49_900)),
        ("C2 WHERE timestamp >= noon + 1 h (zone maps)", QueryPlan(wl.query.aggs, keys=wl.query.keys,
                                                               predicate=col("timestamp") >= workloads.T0_MS + 13 * 3_600_000)),
        ("C2 WHERE service LIKE '%-0%'", QueryPlan(wl.query.aggs, keys=wl.query.keys, predicate=col("service").like("%-0%"))),
        ("GROUP BY value (50k groups)", QueryPlan([count_star()], keys=["value"])),
        ("GROUP BY date_bin(1s) (86k groups)", QueryPlan([count_star(), max_("value")], keys=[date_bin(1000, "timestamp")])),
        ("GROUP BY timestamp (high card.) LIMIT 10", QueryPlan([count_star()], keys=["timestamp"], order_by=1, descending=True, limit=10)),
    ]
    for name, q in queries:
        plan = Plan(ctx, wl.schema, q)
        for _ in range(2):
            plan.execute(table)
        ctx.reset_stats()
        ctx.set_profiling(True)
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            out = plan.execute(table)
        ms = (time.perf_counter() - t0) / n * 1e3
        ctx.set_profiling(False)
        st = ctx.stats()["kernels"]
        parts = []
        for k, v in st.items():
            if not v["launches"]:
                continue
            if k.startswith("host:"):
                parts.append(f"{k} {v['total_ms'] / n:.3f} ms")
                continue
            avg_ms = v["total_ms"] / v["launches"]
            gbs = v["algorithmic_bytes"] / v["launches"] / max(avg_ms, 1e-9) / 1e6
            parts.append(f"{k} {avg_ms:.3f} ms x{v['launches'] // n} ({gbs:.0f} GB/s)")
        print(f"{name:42s} {ms:8.3f} ms/step  {rows / ms / 1e6:7.1f} G rows/s  groups={out.num_rows:<8d} " + "; ".join(parts), flush=True)
        plan.close()
    table.close()
    ctx.close()


if __name__ == "__main__":
    main()
