"""Diagnostic: stream a table whose later chunks leave the first chunk's key range and print the statistics' names."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pyarrow as pa
from prism_amd.engine import Context, Plan
from prism_amd.plan import QueryPlan, count_star, date_bin, sum_

n = 60_000
i = np.arange(n)
ts = 1_700_000_000_000 + i * 5
ts[20_000:30_000] += 40 * 86_400_000
t = pa.table({"ts": pa.array(ts, pa.timestamp("ms")), "v": pa.array(i % 100, pa.int64())})
ctx = Context(0)
ctx.set_option("stream_chunk_rows", 4096)
plan = Plan(ctx, t.schema, QueryPlan([count_star(), sum_("v")], keys=[date_bin(1000, "ts")]))
ctx.reset_stats()
got = plan.run_batches(0, t.to_batches(max_chunksize=1500))
st = ctx.stats()
for k, v in st["kernels"].items():
    print(k, v["launches"])
print(got.num_rows, sum(got.column(1).to_pylist()))
