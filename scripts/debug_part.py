"""Debug aid: partitioned GROUP BY against the CPU oracle at sizes the unit tests do not reach."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle
from prism_amd import workloads
from prism_amd.engine import Context, Plan
from prism_amd.plan import QueryPlan, count_star, count_distinct, date_bin
from tests.helpers import rows_of

def diff(got, want, n_keys):
    g = {r[:n_keys]: r[n_keys:] for r in rows_of(got)}
    w = {r[:n_keys]: r[n_keys:] for r in rows_of(want)}
    bad = [(k, g.get(k), w.get(k)) for k in set(g) | set(w) if g.get(k) != w.get(k)]
    return len(g), len(w), bad[:5], len(bad)

def run(name, wl, rows, parts, q, n_keys, opts):
    ctx = Context(0)
    for k, v in opts.items():
        ctx.set_option(k, v)
    table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, parts)
    st = pyoracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, rows, rows, n_parts=parts, threads=16)
    want, _ = st.execute(q, threads=16)
    want = workloads.decode_dictionary_columns(wl, want)
    plan = Plan(ctx, wl.schema, q)
    for i in range(3):
        ctx.reset_stats()
        got = plan.execute(table)
        ks = {k: v["launches"] for k, v in ctx.stats()["kernels"].items() if not k.startswith("host")}
        print(name, opts, "run", i, diff(got, want, n_keys), ks, flush=True)
    plan.close(); table.close(); st.close(); ctx.close()

which = sys.argv[1] if len(sys.argv) > 1 else "all"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
wl5 = workloads.c5()
if which in ("all", "c5"):
    run("c5", wl5, rows, 8, wl5.query, 2, {})
    run("c5-old", wl5, rows, 8, wl5.query, 2, {"part_min_rows": 1 << 40})
if which in ("all", "c5pairs"):
    q = QueryPlan([count_star()], keys=[date_bin(3_600_000, "timestamp"), "service", "user"])
    run("c5pairs", wl5, rows // 4, 8, q, 3, {})
if which in ("all", "c3"):
    wl3 = workloads.c3()
    q = QueryPlan([count_star()], keys=["UserID"], groups_hint=1000)
    run("c3-hint", wl3, rows, 8, q, 1, {"part_force": 1})
