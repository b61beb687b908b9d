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
    plain = [a for a in aggs if a.func != abi.AGG_COUNT_DISTINCT]
    distinct = [a for a in aggs if a.func == abi.AGG_COUNT_DISTINCT]
    a = oracle.execute(t, QueryPlan(plain, keys=keys, predicate=predicate))
    b = oracle.execute(t, QueryPlan(distinct, keys=keys, predicate=predicate))
    n_keys = len(keys)
    if n_keys:
        knames = a.schema.names[:n_keys]
        b = b.rename_columns(knames + [f"__d{i}" for i in range(len(distinct))])
        j = a.join(b, keys=knames, join_type="inner") if all(a.column(k).null_count == 0 for k in knames) else None
        if j is None:  # join keys with NULLs: align by hand
            rows_b = {tuple(r[k] for k in knames): r for r in b.to_pylist()}
            extra = {f"__d{i}": [] for i in range(len(distinct))}
            for r in a.to_pylist():
                m = rows_b[tuple(r[k] for k in knames)]
                for i in range(len(distinct)):
                    extra[f"__d{i}"].append(m[f"__d{i}"])
            j = a
            for name, vals in extra.items():
                j = j.append_column(name, pa.array(vals, pa.int64()))
    else:
        knames = []
        j = a
        for i in range(len(distinct)):
            j = j.append_column(f"__d{i}", b.column(i))
    cols, names, ia, ib = [j.column(k) for k in knames], list(knames), 0, 0
    for ag in aggs:
        if ag.func == abi.AGG_COUNT_DISTINCT:
            cols.append(j.column(f"__d{ib}"))
            ib += 1
        else:
            cols.append(j.column(a.schema.names[n_keys + ia]))
            ia += 1
        names.append(f"c{len(names)}")
    return pa.table(cols, names=names)


@pytest.mark.parametrize("keys,aggs,pred", [
    (["service"], [count_star(), count_distinct("user"), sum_("value")], None),
    (["service", "i16"], [count_distinct("user"), max_("u16"), avg("value")], "filter"),
    ([], [count_distinct("i16"), count_star(), max_("value")], "filter"),
    ([date_bin(3_600_000, "ts")], [sum_("i32"), count_distinct("service")], None),
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
