#!/usr/bin/env python3
"""Repeats a mix of queries on one context and reports free HBM and host RSS before and after:
a leak in the scratch pool, the pinned pool or the plans' learnt state shows as a drift.
usage: python scripts/soak.py [iterations]   (needs a GPU)"""
import os
import resource
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from prism_amd import workloads  # noqa: E402
from prism_amd.engine import Context, Plan  # noqa: E402
from prism_amd.plan import QueryPlan, col, count_star  # noqa: E402


def main() -> int:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ctx = Context(0)
    for name, value in (("part_min_rows", 0), ("part_wc_min_rows", 0), ("part2_lines", 2)):
        ctx.set_option(name, value)
    c3, c5, c2 = workloads.c3(id_bits=20), workloads.c5(), workloads.c2()
    tables = [(c3, workloads.build_gpu_table(ctx, c3, 0, 3_000_000, 3_000_000, 4)),
              (c5, workloads.build_gpu_table(ctx, c5, 0, 20_000_000, 20_000_000, 4)),
              (c2, workloads.build_gpu_table(ctx, c2, 0, 5_000_000, 5_000_000, 4))]
    queries = [(0, c3.query), (0, QueryPlan([count_star()], keys=["UserID"])), (1, c5.query), (2, c2.query),
               (2, QueryPlan([count_star()], predicate=col("value") > 0))]

    def snapshot():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss

    def one_round():
        for ti, q in queries:
            wl, table = tables[ti]
            plan = Plan(ctx, wl.schema, q)
            try:
                plan.execute(table)
                plan.execute(table)
            finally:
                plan.close()

    for _ in range(5):
        one_round()
    free0, rss0 = snapshot()
    for i in range(n):
        one_round()
    free1, rss1 = snapshot()
    print(f"[soak] {n} rounds x {len(queries)} plans x 2 runs: free HBM {free0 >> 20} -> {free1 >> 20} MiB, max RSS {rss0 >> 10} -> {rss1 >> 10} MiB")
    for _, t in tables:
        t.close()
    ctx.close()
    return 0 if free0 - free1 < (64 << 20) and rss1 - rss0 < (256 << 10) else 1


if __name__ == "__main__":
    sys.exit(main())
