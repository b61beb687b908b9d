"""C3's query (GROUP BY UserID, COUNT(*), top 10) over log-uniform ids of several widths: the
narrower the id, the larger the share of the most frequent ones (1 / (0.69 * bits) for the
first).  Prints the stage times of the partitioned path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prism_amd import workloads
from prism_amd.engine import Context, Plan

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = Context(0)
for bits in (18, 20, 22, 24):
    wl = workloads.c3(id_bits=bits)
    table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, 32)
    plan = Plan(ctx, wl.schema, wl.query)
    for _ in range(3):
        r = plan.execute(table)
    ctx.set_profiling(True)
    ctx.reset_stats()
    t0 = time.perf_counter(); n = 5
    for _ in range(n):
        r = plan.execute(table)
    ms = (time.perf_counter() - t0) / n * 1e3
    stages = {}
    for name, k in ctx.stats()["kernels"].items():
        if name.startswith("part") or name == "group_by":
            stages[name] = round(k["total_ms"] / n, 3)
    ctx.set_profiling(False)
    print(f"id_bits={bits:2d} {ms:7.3f} ms/query  {stages}", flush=True)
    plan.close(); table.close()

# half of the rows carry one id, the others are spread over ten million ids
import numpy as np
import pyarrow as pa
from prism_amd.engine import Table
rng = np.random.default_rng(1)
n = min(rows, 50_000_000)
ids = rng.integers(1, 10_000_000, n, dtype=np.int64) * 2654435761
ids[rng.random(n) < 0.5] = 7
t = pa.table({"UserID": pa.array(ids)})
wl = workloads.c3()
table = Table(ctx, t.schema)
table.append_table(t, 8)
plan = Plan(ctx, wl.schema, wl.query)
for _ in range(3):
    r = plan.execute(table)
assert r.column(1)[0].as_py() == int((ids == 7).sum()), r
ctx.set_profiling(True)
ctx.reset_stats()
t0 = time.perf_counter(); k = 5
for _ in range(k):
    r = plan.execute(table)
ms = (time.perf_counter() - t0) / k * 1e3
stages = {name: round(v["total_ms"] / k, 3) for name, v in ctx.stats()["kernels"].items() if name.startswith("part") or name == "group_by"}
print(f"half of {n} rows one id: {ms:7.3f} ms/query  {stages}", flush=True)
plan.close(); table.close()
ctx.close()
