#!/usr/bin/env python3
"""PARTIAL per partition, then merge, for a few rounds on one plan: prints what each merge launched
(a pass that runs twice every round is sized by something the partitions' runs overwrote).
usage: python scripts/debug_merge_rounds.py [C3|C5] [rows] [partitions]   (needs a GPU)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd import workloads  # noqa: E402
from prism_amd.engine import Context, Plan  # noqa: E402


def main() -> int:
    name = sys.argv[1] if len(sys.argv) > 1 else "C5"
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000_000
    parts = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    wl = workloads.c3(id_bits=22) if name == "C3" else workloads.WORKLOADS[name]()
    ctx = Context(0)
    for opt in sys.argv[4:]:
        k, _, v = opt.partition("=")
        ctx.set_option(k, int(v))
    table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, parts)
    plan = Plan(ctx, wl.schema, wl.query.with_mode(abi.MODE_PARTIAL))
    ctx.set_profiling(1)
    for r in range(4):
        states = [plan.execute(table, [p]) for p in range(parts)]
        ctx.reset_stats()
        t0 = time.perf_counter()
        out = plan.merge(states)
        ms = (time.perf_counter() - t0) * 1e3
        k = {n: (v["launches"], round(v["total_ms"], 2)) for n, v in ctx.stats()["kernels"].items() if v["launches"]}
        print(f"round {r}: merge {ms:.1f} ms, {out.num_rows} rows, state rows {sum(s.num_rows for s in states)}: {k}", flush=True)
    plan.close()
    table.close()
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
