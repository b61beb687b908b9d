"""Runs C3 through the histogram-free path with its diagnostics on (PRISM_WC_DEBUG=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PRISM_WC_DEBUG"] = "1"
from prism_amd import workloads
from prism_amd.engine import Context, Plan

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
ctx = Context(0)
ctx.set_option("part_force", 1)
wl = workloads.c3(id_bits=bits)
table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, 100)
plan = Plan(ctx, wl.schema, wl.query)
ctx.set_profiling(True)
for i in range(4):
    ctx.reset_stats()
    t0 = time.perf_counter()
    r = plan.execute(table)
    dt = time.perf_counter() - t0
    ks = ctx.stats()["kernels"]
    print(i, f"{dt*1e3:.3f} ms", {k: round(v["total_ms"] / max(1, v["launches"]), 4) for k, v in ks.items()}, flush=True)
print(r.to_pylist()[:3])
