"""ORDER BY <column> [DESC] LIMIT k over many groups: the leading groups are picked on
the GPU (radix select over the group table) and must be the rows the CPU oracle's full
sort puts first.  Ties at the cut are broken arbitrarily by the reference
(datafusion SortExec with fetch), so the check is: same ORDER BY value sequence, and
every returned row is a row of the unlimited result."""
import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu

from prism_amd.plan import QueryPlan, avg, count, count_star, date_bin, max_, min_, sum_  # noqa: E402

AGGS = [count_star(), sum_("value"), min_("i32"), max_("u16"), avg("value"), count("i32")]


def _table(n=120_000, seed=5):
    rng = np.random.default_rng(seed)
    mask = lambda p: rng.random(n) < p  # noqa: E731
    return pa.table({
        "ts": pa.array(1_700_000_000_000 + np.sort(rng.integers(-10**9, 10**9, n)), pa.timestamp("ms")),
        # ~30k groups, many of them with a single row, so that NULL aggregates are common
        "user": pa.array((rng.zipf(1.3, n) % 30_000).astype(np.int64) * 104729 - 17, mask=mask(0.01)),
        "value": pa.array(rng.integers(-50_000, 50_000, n).astype(np.int64), mask=mask(0.3)),
        "i32": pa.array(rng.integers(-1000, 1000, n).astype(np.int32), mask=mask(0.3)),
        "u16": pa.array(rng.integers(0, 65535, n).astype(np.uint16), mask=mask(0.3)),
    })


@pytest.fixture(scope="module")
def loaded(gpu_ctx):
    from prism_amd.engine import Table

    t = _table()
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 3)
    yield t, g
    g.close()


def _rows(tbl):
    cols = [c.to_pylist() for c in tbl.columns]
    return list(zip(*cols)) if cols else []


def _check(gpu_ctx, t, g, q, n_keys):
    from oracle import pyoracle
    from prism_amd.engine import Plan

    plan = Plan(gpu_ctx, t.schema, q)
    try:
        got = plan.execute(g)
    finally:
        plan.close()
    want = pyoracle.execute(t, q)
    full = pyoracle.execute(t, QueryPlan(q.aggs, q.keys, q.predicate))
    assert got.schema.names == want.schema.names
    assert got.num_rows == want.num_rows
    assert got.column(q.order_by).to_pylist() == want.column(q.order_by).to_pylist()
    by_key = {r[:n_keys]: r for r in _rows(full)}
    for r in _rows(got):
        ref = by_key[r[:n_keys]]
        for a, b in zip(r, ref):
            if isinstance(a, float):
                assert b is not None and abs(a - b) <= 1e-9 * max(1.0, abs(b))
            else:
                assert a == b


@pytest.mark.parametrize("descending", [True, False])
@pytest.mark.parametrize("order_by", range(0, 7))
@pytest.mark.parametrize("limit", [1, 10, 2500])
def test_topk_by_every_column(gpu_ctx, loaded, order_by, descending, limit):
    t, g = loaded
    q = QueryPlan(AGGS, keys=["user"], order_by=order_by, descending=descending, limit=limit)
    _check(gpu_ctx, t, g, q, 1)


def test_topk_limit_beyond_groups(gpu_ctx, loaded):
    t, g = loaded
    q = QueryPlan([count_star()], keys=["user"], order_by=1, descending=True, limit=1_000_000)
    _check(gpu_ctx, t, g, q, 1)


def test_topk_date_bin_key_and_two_keys(gpu_ctx, loaded):
    t, g = loaded
    for order_by, desc in [(0, True), (0, False), (1, True), (2, True)]:
        q = QueryPlan([sum_("value")], keys=[date_bin(1000, "ts"), "user"], order_by=order_by, descending=desc, limit=25)
        _check(gpu_ctx, t, g, q, 2)


def test_topk_after_merge(gpu_ctx, loaded):
    """Final merge of PARTIAL states applies the same device-side cut."""
    from oracle import pyoracle
    from prism_amd import _cabi as abi
    from prism_amd.engine import Plan

    t, g = loaded
    q = QueryPlan([count_star(), avg("value")], keys=["user"], order_by=2, descending=True, limit=20)
    plan = Plan(gpu_ctx, t.schema, q.with_mode(abi.MODE_PARTIAL))
    try:
        states = [plan.execute(g, [p]) for p in range(3)]
        got = plan.merge(states)
    finally:
        plan.close()
    want = pyoracle.execute(t, q)
    assert got.num_rows == want.num_rows
    a, b = got.column(2).to_pylist(), want.column(2).to_pylist()
    assert all((x is None and y is None) or abs(x - y) <= 1e-9 * max(1.0, abs(y)) for x, y in zip(a, b))
