"""GROUP BY shapes whose groups do not fit LDS, on the C2 table: ms per query, streaming vs partitioned."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prism_amd import workloads
from prism_amd.engine import Context, Plan
from prism_amd.plan import QueryPlan, count_star, sum_, max_, avg, date_bin, col

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
ctx = Context(0)
wl = workloads.c2()
table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, 32)
shapes = {
    "by value (50k groups) COUNT": QueryPlan([count_star()], keys=["value"]),
    "by value COUNT,SUM(ts-free) MAX": QueryPlan([count_star(), max_("value")], keys=["value"]),
    "by value,service (3.2M groups) COUNT": QueryPlan([count_star()], keys=["value", "service"]),
    "by date_bin(1s),value COUNT,SUM": QueryPlan([count_star(), sum_("value")], keys=[date_bin(1000, "timestamp"), "value"]),
}
for name, q in shapes.items():
    for mode, minrows in (("stream", 1 << 60), ("auto", 1 << 22)):
        ctx.set_option("part_min_rows", minrows)
        plan = Plan(ctx, wl.schema, q)
        for _ in range(2):
            r = plan.execute(table)
        t0 = time.perf_counter(); n = 3
        for _ in range(n):
            r = plan.execute(table)
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{name:45s} {mode:7s} {ms:9.2f} ms  {rows / ms / 1e6:8.1f} G rows/s  groups={r.num_rows}", flush=True)
        plan.close()
table.close(); ctx.close()
