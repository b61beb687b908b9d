"""Binary columns (proto ColumnType BINARY = 7, "present in some benchmark Parquet datasets",
proto/common/prism/common/v1/partition.proto:26-38; mapped to Arrow Binary by
src/prism-query/src/providers/schema_provider.rs:96-107), statistics passed in by the caller,
and partition pruning from the WHERE's time range."""
import numpy as np
import pyarrow as pa
import pytest

from tests.helpers import assert_tables_match

pytestmark = pytest.mark.gpu

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd.plan import QueryPlan, col, count, count_distinct, count_star, date_bin, max_, sum_  # noqa: E402


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle

    return pyoracle


def _binary_table(n=30_000, seed=5):
    rng = np.random.default_rng(seed)
    tags = [bytes([0xff, 0x00, i]) + b"tag" for i in range(9)] + [b"", b"\x00"]
    picks = rng.integers(0, len(tags), n)
    return pa.table({
        "blob": pa.array([tags[i] for i in picks], pa.binary(), mask=rng.random(n) < 0.07),
        "v": pa.array(rng.integers(-1000, 1000, n).astype(np.int64)),
        "k": pa.array(rng.integers(0, 5, n).astype(np.int16)),
    })


def test_binary_column_append_count_group_by_distinct_export(gpu_ctx, oracle):
    from prism_amd.engine import Plan, Table

    t = _binary_table()
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 3)
    try:
        for q, n_keys in [(QueryPlan([count_star(), count("blob")]), 0),
                          (QueryPlan([count_star(), sum_("v"), max_("v")], keys=["blob"]), 1),
                          (QueryPlan([count_star()], keys=["k", "blob"]), 2),
                          (QueryPlan([count_distinct("blob")], keys=["k"]), 1),
                          (QueryPlan([count_star()], predicate=col("blob").is_null()), 0)]:
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                got = plan.execute(g)
            finally:
                plan.close()
            if n_keys and "blob" in got.schema.names:
                assert got.schema.field("blob").type == pa.binary()  # not silently turned into Utf8
            assert_tables_match(got, oracle.execute(t, q), n_keys)
    finally:
        g.close()


def test_like_on_binary_is_unsupported_like_in_datafusion(gpu_ctx):
    """DataFusion 32 does not type-check LIKE on Binary; the plan is refused at creation so
    that the caller keeps its own operator (or casts the column to Utf8 first)."""
    from prism_amd.engine import Plan, PrismGpuError

    t = _binary_table(10)
    with pytest.raises(PrismGpuError) as e:
        Plan(gpu_ctx, t.schema, QueryPlan([count_star()], predicate=col("blob").like("%tag%")))
    assert e.value.code == abi.EUNSUPPORTED


def test_statistics_that_do_not_cover_the_scan_are_refused(gpu_ctx):
    from prism_amd import workloads
    from prism_amd.engine import Plan, PrismGpuError

    wl = workloads.c2()
    table = workloads.build_gpu_table(gpu_ctx, wl, 0, 200_000, 200_000, 2)
    plan = Plan(gpu_ctx, wl.schema, wl.query)
    try:
        stats = plan.scan_stats(table)
        st = plan.aggregate(table, None, stats)  # the scan's own statistics: fine
        st.close()
        narrow = stats.copy()
        narrow[2, 1] -= 1  # max(value) one too low
        with pytest.raises(PrismGpuError) as e:
            plan.aggregate(table, None, narrow)
        assert e.value.code == abi.EINVAL and "value" in e.value.msg
        late = stats.copy()
        late[0, 0] += 60_000  # min(timestamp) a minute too high
        with pytest.raises(PrismGpuError) as e:
            plan.aggregate(table, None, late)
        assert e.value.code == abi.EINVAL
    finally:
        plan.close()
        table.close()


def test_time_range_of_the_where_prunes_partitions_without_changing_the_result(gpu_ctx, oracle):
    """The caller lists only the partitions whose time range overlaps the plan's (what the
    catalogue call would do with time_range); the result equals the scan of everything."""
    from prism_amd import workloads
    from prism_amd.engine import Plan

    wl = workloads.c2()
    rows, parts = 480_000, 8
    table = workloads.build_gpu_table(gpu_ctx, wl, 0, rows, rows, parts)
    t0, day = workloads.T0_MS, workloads.DAY_MS
    lo, hi = t0 + day // 4, t0 + day // 2
    q = QueryPlan([count_star(), sum_("value")], keys=[date_bin(60_000, "timestamp"), "service"],
                  predicate=(col("timestamp") >= lo) & (col("timestamp") < hi))
    plan = Plan(gpu_ctx, wl.schema, q)
    try:
        start, end = plan.time_range("timestamp")
        assert (start, end) == (lo, hi - 1)
        # partition p of the generated table covers [t0 + day*p/parts, t0 + day*(p+1)/parts)
        keep = [p for p in range(parts) if t0 + day * p // parts <= end and t0 + day * (p + 1) // parts > start]
        assert 0 < len(keep) < parts
        assert_tables_match(plan.execute(table, keep), plan.execute(table), 2)
    finally:
        plan.close()
        table.close()
