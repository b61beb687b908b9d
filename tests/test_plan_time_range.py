"""prism_gpu_plan_time_range: the range of a column outside which no row can pass the WHERE
(what TableProvider::scan would hand to GetTablePartitions as time_range; the reference
passes None, src/prism-query/src/providers/table_provider.rs:86-93).  Host logic only: the
plans are created without a context."""
import pyarrow as pa
import pytest

from prism_amd.plan import QueryPlan, col, count_star

I64_MIN, I64_MAX = -(1 << 63), (1 << 63) - 1
SCHEMA = pa.schema([("timestamp", pa.timestamp("ms")), ("service", pa.string()), ("value", pa.int64())])
T = 1_700_000_000_000


def _range(pred, column="timestamp"):
    from prism_amd.engine import Plan

    p = Plan(None, SCHEMA, QueryPlan([count_star()], predicate=pred))
    try:
        return p.time_range(column)
    finally:
        p.close()


def test_no_predicate_is_unbounded():
    assert _range(None) == (I64_MIN, I64_MAX)


def test_half_open_interval_from_two_comparisons():
    assert _range((col("timestamp") >= T) & (col("timestamp") < T + 3_600_000)) == (T, T + 3_600_000 - 1)
    assert _range((col("timestamp") > T) & (col("timestamp") <= T + 5)) == (T + 1, T + 5)
    assert _range(col("timestamp") == T) == (T, T)


def test_other_columns_do_not_narrow_and_or_widens():
    assert _range((col("timestamp") >= T) & (col("value") < 10)) == (T, I64_MAX)
    assert _range((col("timestamp") >= T) | (col("value") < 10)) == (I64_MIN, I64_MAX)
    assert _range((col("timestamp") < T) | (col("timestamp") >= T + 100)) == (I64_MIN, I64_MAX)
    assert _range(((col("timestamp") >= T) & (col("timestamp") < T + 10)) |
                  ((col("timestamp") >= T + 100) & (col("timestamp") < T + 110))) == (T, T + 109)
    assert _range(col("service") == "svc-01") == (I64_MIN, I64_MAX)
    assert _range(col("value") > 5, "value") == (6, I64_MAX)


def test_not_is_pushed_into_the_comparisons():
    assert _range(~(col("timestamp") < T)) == (T, I64_MAX)
    assert _range(~((col("timestamp") < T) | (col("timestamp") >= T + 50))) == (T, T + 49)


def test_contradiction_is_an_empty_range():
    lo, hi = _range((col("timestamp") < T) & (col("timestamp") > T))
    assert lo > hi


def test_string_columns_have_no_range_and_such_a_plan_does_not_run():
    from prism_amd import _cabi as abi
    from prism_amd.engine import Plan, PrismGpuError

    p = Plan(None, SCHEMA, QueryPlan([count_star()]))
    try:
        with pytest.raises(PrismGpuError) as e:
            p.time_range("service")
        assert e.value.code == abi.EINVAL
    finally:
        p.close()
