"""Pins the CPU oracle: against the reference's own fixture and against golden vectors
computed independently with pyarrow/numpy (scripts/make_golden.py).  No GPU needed."""
import json
import math
import os

import pyarrow as pa
import pytest

from oracle import pyoracle
from prism_amd import _cabi as abi
from prism_amd.plan import QueryPlan, avg, col, count_distinct, count_star, date_bin, max_, min_, sum_
from tests import helpers


@pytest.fixture(scope="module")
def demo():
    return helpers.demo_table()


def test_reference_pinned_min_max(demo):
    """The one numeric vector the reference's tests hold for this path:
    src/prism-ingest/src/ingest.rs:224-226 (and the 10 columns of :227)."""
    assert demo.num_rows == 34 and demo.num_columns == 10
    r = pyoracle.execute(demo, QueryPlan([max_("timestamp"), min_("timestamp")]))
    assert r.schema.types == [pa.timestamp("ms"), pa.timestamp("ms")]
    assert r.column(0).cast(pa.int64()).to_pylist() == [1698000995523]
    assert r.column(1).cast(pa.int64()).to_pylist() == [1698000992225]


def test_demo_log_derived_answers(demo):
    """Derived by hand from the same fixture (not pinned by the reference)."""
    r = pyoracle.execute(demo, QueryPlan([count_star(), sum_("bytes"), max_("bytes")], keys=["method"]))
    got = {m: (c, s, x) for m, c, s, x in helpers.rows_of(r)}
    assert got == {"DELETE": (13, 347590, 48600), "GET": (2, 83654, 44428), "HEAD": (4, 78885, 28663),
                   "OPTION": (4, 111733, 49236), "PATCH": (3, 28387, 14181), "POST": (5, 108229, 46657),
                   "PUT": (3, 85000, 45503)}
    assert sum(v[1] for v in got.values()) == 843478
    assert pyoracle.execute(demo, QueryPlan([count_star()], predicate=col("status") != "200")).column(0).to_pylist() == [30]
    assert pyoracle.execute(demo, QueryPlan([count_distinct("user-identifier")])).column(0).to_pylist() == [7]
    assert pyoracle.execute(demo, QueryPlan([count_star()], predicate=col("request").like("%admin%"))).column(0).to_pylist() == [3]
    r = pyoracle.execute(demo, QueryPlan([count_star()], keys=[date_bin(60_000, "timestamp")]))
    assert r.column(0).cast(pa.int64()).to_pylist() == [1698000960000] and r.column(1).to_pylist() == [34]


def test_prism_ql_plan_shapes(demo):
    """`table | count` and `table | count by col` (src/prism-ql/src/lower.rs:83-95):
    output schema and names as DataFusion's Aggregate would give them."""
    r = pyoracle.execute(demo, QueryPlan([count_star()]))
    assert r.schema.names == ["COUNT(*)"] and r.schema.types == [pa.int64()] and r.column(0).to_pylist() == [34]
    r = pyoracle.execute(demo, QueryPlan([count_star()], keys=["method"]))
    assert r.schema.names == ["method", "COUNT(*)"] and r.num_rows == 7


def _golden():
    t = pa.ipc.open_file(os.path.join(helpers.GOLDEN, "rand_small.arrow")).read_all()
    exp = json.load(open(os.path.join(helpers.GOLDEN, "rand_small_expected.json")))
    return t, exp


def _canon(v, typ):
    if v is None:
        return None
    if pa.types.is_timestamp(typ):
        return pa.scalar(v, typ).cast(pa.int64()).as_py()
    return v


@pytest.mark.parametrize("pname", [p[0] for p in helpers.plan_catalogue()])
def test_oracle_matches_independent_engines(pname):
    t, exp = _golden()
    _, q, n_keys = next(p for p in helpers.plan_catalogue() if p[0] == pname)
    got = pyoracle.execute(t, q)
    rows = [[_canon(c[i].as_py(), c.type) for c in got.columns] for i in range(got.num_rows)]
    want = exp[pname]["rows"]
    key = lambda r: tuple((0, 0) if v is None else (1, v) for v in r[:n_keys or len(r)])  # noqa: E731
    rows.sort(key=lambda r: tuple((0, "") if v is None else (1, str(v)) for v in r[:n_keys or len(r)]))
    want.sort(key=lambda r: tuple((0, "") if v is None else (1, str(v)) for v in r[:n_keys or len(r)]))
    assert len(rows) == len(want)
    for a, b in zip(rows, want):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            if isinstance(x, float) or isinstance(y, float):
                assert x is not None and y is not None and math.isclose(x, y, rel_tol=helpers.FLOAT_RTOL), (a, b)
            else:
                assert x == y, (a, b)
    del key


def test_partial_state_schema_follows_datafusion_names():
    t, _ = _golden()
    q = QueryPlan([count_star(), sum_("value"), max_("value"), avg("value"), sum_("u16"), min_("ts")],
                  keys=[date_bin(60_000, "ts"), "service"], mode=abi.MODE_PARTIAL)
    s = pyoracle.output_schema(t.schema, q)
    assert s.names == ["ts", "service", "COUNT(*)[count]", "SUM(value)[sum]", "MAX(value)[max]", "AVG(value)[count]",
                       "AVG(value)[sum]", "SUM(u16)[sum]", "MIN(ts)[min]"]
    assert s.types == [pa.timestamp("ms"), pa.string(), pa.int64(), pa.int64(), pa.int64(), pa.uint64(), pa.float64(),
                       pa.uint64(), pa.timestamp("ms")]


@pytest.mark.parametrize("pname", ["global_aggs", "by_minute_service_full", "by_user", "distinct_user_by_service",
                                   "distinct_user"])
def test_partial_then_final_equals_single(pname):
    t, _ = _golden()
    _, q, n_keys = next(p for p in helpers.plan_catalogue() if p[0] == pname)
    qp = q.with_mode(abi.MODE_PARTIAL)
    states = [pyoracle.execute(b, qp) for b in t.to_batches(max_chunksize=700)]
    helpers.assert_tables_match(pyoracle.merge(t.schema, states, q), pyoracle.execute(t, q), n_keys)


def test_threads_do_not_change_results():
    t, _ = _golden()
    q = QueryPlan([count_star(), sum_("value"), avg("i32")], keys=["service", "i16"])
    batches = t.to_batches(max_chunksize=300)
    helpers.assert_tables_match(pyoracle.execute(batches, q, threads=4), pyoracle.execute(t, q), 2)


def test_wrapping_sum_and_date_bin_before_origin():
    t = pa.table({"ts": pa.array([-120_000, -60_001, -60_000, -1, 0, 59_999, 60_000], pa.timestamp("ms")),
                  "v": pa.array([2**62, 2**62, 2**62, 2**62, 1, 2, 3], pa.int64())})
    r = pyoracle.execute(t, QueryPlan([sum_("v")]))
    assert r.column(0).to_pylist() == [(4 * 2**62 + 6 + 2**63) % 2**64 - 2**63]  # wraps like i64
    r = pyoracle.execute(t, QueryPlan([count_star()], keys=[date_bin(60_000, "ts")]))
    got = dict(zip(r.column(0).cast(pa.int64()).to_pylist(), r.column(1).to_pylist()))
    # exact negative multiples land one stride lower (the behaviour restated from
    # DataFusion 32's date_bin; unpinned by the reference)
    assert got == {-180_000: 1, -120_000: 2, -60_000: 1, 0: 2, 60_000: 1}
