"""Runs C5 with the partition levels' diagnostics on (PRISM_WC_DEBUG=1) and prints the sizes of the fine buckets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PRISM_WC_DEBUG"] = "1"
from prism_amd import workloads
from prism_amd.engine import Context, Plan

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
ctx = Context(0)
for o in sys.argv[2:]:
    k, v = o.split("=")
    ctx.set_option(k, int(v))
wl = workloads.c5()
table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, 64)
plan = Plan(ctx, wl.schema, wl.query)
ctx.set_profiling(True)
for i in range(3):
    ctx.reset_stats()
    t0 = time.perf_counter()
    r = plan.execute(table)
    dt = time.perf_counter() - t0
    ks = ctx.stats()["kernels"]
    print(i, f"{dt*1e3:.3f} ms", {k: round(v["total_ms"] / max(1, v["launches"]), 3) for k, v in ks.items() if not k.startswith("host")}, flush=True)
