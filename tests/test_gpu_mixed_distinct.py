"""COUNT(DISTINCT x) next to other aggregates (SINGLE mode): the library runs the two halves
over the same rows and joins them on the group keys.  The expectation is built the same way
from two oracle runs (the oracle, like DataFusion's SingleDistinctToGroupBy rewrite, handles
plans whose aggregates are all COUNT(DISTINCT) or none)."""
import pyarrow as pa
import pytest

from tests import helpers
from tests.helpers import assert_tables_match

pytestmark = pytest.mark.gpu

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd.plan import QueryPlan, avg, col, count_distinct, count_star, date_bin, max_, sum_  # noqa: E402


def _expected(oracle, t, keys, aggs, predicate):
    """One oracle run per kind of aggregate (the plain ones; the COUNT(DISTINCT)s of each
    column), aligned on the group keys."""
    kinds = {}
    for i, a in enumerate(aggs):
        kinds.setdefault(a.column if a.func == abi.AGG_COUNT_DISTINCT else None, []).append(i)
    n_keys = len(keys)
    per_kind = {}
    for kind, idxs in kinds.items():
        res = oracle.execute(t, QueryPlan([aggs[i] for i in idxs], keys=keys, predicate=predicate))
        rows = {}
        for r in range(res.num_rows):
            key = tuple(res.column(c)[r].as_py() for c in range(n_keys))
            rows[key] = [res.column(n_keys + j)[r].as_py() for j in range(len(idxs))]
        per_kind[kind] = (res, rows)
    base, _ = next(iter(per_kind.values()))
    out_rows = []
    for r in range(base.num_rows):
        key = tuple(base.column(c)[r].as_py() for c in range(n_keys))
        vals = [None] * len(aggs)
        for kind, idxs in kinds.items():
            for j, i in enumerate(idxs):
                vals[i] = per_kind[kind][1][key][j]
        out_rows.append(list(key) + vals)
    cols = list(zip(*out_rows)) if out_rows else [[] for _ in range(n_keys + len(aggs))]
    arrays = [pa.array(list(cols[c]), type=base.column(c).type) for c in range(n_keys)]
    for i, a in enumerate(aggs):
        kind = a.column if a.func == abi.AGG_COUNT_DISTINCT else None
        src = per_kind[kind][0]
        arrays.append(pa.array(list(cols[n_keys + i]), type=src.column(n_keys + kinds[kind].index(i)).type))
    return pa.table(arrays, names=[f"c{c}" for c in range(len(arrays))])


@pytest.mark.parametrize("keys,aggs,pred", [
    (["service"], [count_star(), count_distinct("user"), sum_("value")], None),
    (["service", "i16"], [count_distinct("user"), max_("u16"), avg("value")], "filter"),
    ([], [count_distinct("i16"), count_star(), max_("value")], "filter"),
    ([date_bin(3_600_000, "ts")], [sum_("i32"), count_distinct("service")], None),
    (["service"], [count_distinct("user"), count_distinct("i16")], None),                       # two distinct columns
    (["i16"], [count_distinct("service"), count_star(), count_distinct("user"), count_distinct("service")], "filter"),
    ([], [count_distinct("user"), count_distinct("value")], None),
])
def test_count_distinct_beside_other_aggregates(gpu_ctx, keys, aggs, pred):
    from oracle import pyoracle
    from prism_amd.engine import Plan, Table

    t = helpers.random_table(40_000, seed=31, nulls=True)
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 3)
    predicate = (col("i32") > -500) if pred else None
    try:
        plan = Plan(gpu_ctx, t.schema, QueryPlan(aggs, keys=keys, predicate=predicate))
        try:
            got = plan.execute(g)
        finally:
            plan.close()
        want = _expected(pyoracle, t, keys, aggs, predicate)
        got = got.rename_columns(want.schema.names)
        assert_tables_match(got, want, len(keys))
    finally:
        g.close()


def test_order_by_the_distinct_count_and_what_is_refused(gpu_ctx):
    from oracle import pyoracle
    from prism_amd.engine import Plan, PrismGpuError, Table

    t = helpers.random_table(40_000, seed=32, nulls=False)
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 2)
    try:
        aggs = [count_star(), count_distinct("user")]
        plan = Plan(gpu_ctx, t.schema, QueryPlan(aggs, keys=["service"], order_by=2, descending=True, limit=3))
        try:
            got = plan.execute(g)
        finally:
            plan.close()
        want = _expected(pyoracle, t, ["service"], aggs, None)
        top = sorted(want.column(2).to_pylist(), reverse=True)[:3]
        assert got.num_rows == 3 and got.column(2).to_pylist() == top
        by_service = {r["service"]: r for r in want.rename_columns(["service", "n", "d"]).to_pylist()}
        assert want.num_rows > 3
        for r in got.rename_columns(["service", "n", "d"]).to_pylist():
            assert by_service[r["service"]] == r
        # no PARTIAL form, no streaming form
        with pytest.raises(PrismGpuError) as e:
            Plan(gpu_ctx, t.schema, QueryPlan(aggs, keys=["service"]).with_mode(abi.MODE_PARTIAL))
        assert e.value.code == abi.EUNSUPPORTED
        plan = Plan(gpu_ctx, t.schema, QueryPlan(aggs, keys=["service"]))
        try:
            with pytest.raises(PrismGpuError) as e:
                plan.run_batches(0, t.to_batches(max_chunksize=8192))
            assert e.value.code == abi.EUNSUPPORTED
        finally:
            plan.close()
    finally:
        g.close()
