"""WHERE evaluated inside the aggregation kernel: conjunctions / disjunctions of integer
comparisons and IS [NOT] NULL need no bitmap pass.  Every plan below must (a) give the
oracle's answer bit for bit, (b) report the oracle's number of selected rows and (c) have
run without a predicate kernel; plans the kernel cannot take, or that select little, must
fall back to the bitmap path and agree as well."""
import numpy as np
import pyarrow as pa
import pytest

from tests import helpers
from tests.helpers import assert_tables_match

pytestmark = pytest.mark.gpu

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd.plan import QueryPlan, avg, col, count, count_star, date_bin, max_, min_, sum_  # noqa: E402

PRED_KERNELS = {"pred_cmp", "pred_null", "pred_lut", "pred_str", "like_contains", "compact_selvec"}


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle

    return pyoracle


def _selected_rows(oracle, t, predicate):
    return oracle.execute(t, QueryPlan([count_star()], predicate=predicate)).column(0)[0].as_py()


def _run(gpu_ctx, oracle, t, table, q, n_keys, expect_fused):
    from prism_amd.engine import Plan

    plan = Plan(gpu_ctx, t.schema, q)
    try:
        gpu_ctx.reset_stats()
        got = plan.execute(table)
        st = gpu_ctx.stats()
    finally:
        plan.close()
    assert_tables_match(got, oracle.execute(t, q), n_keys)
    assert st["rows_selected"] == _selected_rows(oracle, t, q.predicate)
    ran = {k for k, v in st["kernels"].items() if v["launches"]}
    if expect_fused:
        assert not (ran & PRED_KERNELS), ran
    else:
        assert ran & PRED_KERNELS, ran


@pytest.mark.parametrize("nulls", [False, True])
def test_fused_predicates_match_the_oracle(gpu_ctx, oracle, nulls):
    from prism_amd.engine import Table

    t = helpers.random_table(50_000, seed=31, nulls=nulls)
    table = Table(gpu_ctx, t.schema)
    table.append_table(t, 3)
    keys2 = [date_bin(60_000, "ts"), "service"]
    aggs = [count_star(), sum_("value"), max_("value"), min_("i32"), avg("u16"), count("user")]
    try:
        cases = [
            # one leaf, on a column the aggregates read / do not read
            (QueryPlan(aggs, keys=keys2, predicate=col("value") >= 0), 2),
            (QueryPlan([count_star(), sum_("value")], keys=["service"], predicate=col("i16") != 0), 1),
            # range on one column, conjunction over four columns, disjunction
            (QueryPlan(aggs, keys=keys2, predicate=(col("i32") >= -500) & (col("i32") < 700)), 2),
            (QueryPlan([count_star(), sum_("value"), min_("i32")], keys=keys2,
                       predicate=(col("value") > -40_000) & (col("i32") <= 900) & (col("u16") >= 1000) & (col("i16") < 4)), 2),
            (QueryPlan(aggs, keys=["i16"], predicate=(col("value") < -20_000) | (col("u16") > 60_000) | (col("i32") == 7)), 1),
            # NOT is pushed into the leaves at plan time (Kleene): still one conjunction
            (QueryPlan(aggs, keys=keys2, predicate=~((col("value") > 10_000) | (col("i32") < -900))), 2),
            # validity leaves
            (QueryPlan(aggs, keys=["service"], predicate=col("value").is_not_null() & (col("i32") > -800)), 1),
            (QueryPlan([count_star(), max_("i32")], keys=["service"], predicate=col("value").is_null() | (col("value") > 20_000)), 1),
            # timestamps compare as 64-bit integers; 64-bit keys (user) take the wide-key kernel variant
            (QueryPlan([count_star(), sum_("value")], keys=["user"], predicate=col("ts") >= 1_700_000_000_000 + 3_600_000), 1),
            # leaves on GROUP BY key columns read the key's own load: bucketed timestamp (32-bit key form),
            # plain timestamp key, 8-byte and 4-byte integer keys, a 2-byte key (loaded again as a value),
            # a key column and a value column in one conjunction, literals outside the key's range
            (QueryPlan([count_star(), sum_("value")], keys=keys2, predicate=col("ts") >= 1_700_000_000_000 + 4_000_000), 2),
            (QueryPlan([count_star()], keys=keys2, predicate=(col("ts") >= 1_700_000_000_000 + 1_000_000)
                       & (col("ts") < 1_700_000_000_000 + 9_000_000)), 2),
            (QueryPlan([count_star(), max_("value")], keys=["ts"], predicate=col("ts") < 1_700_000_000_000 + 5_000_000), 1),
            (QueryPlan([count_star(), sum_("value")], keys=["user"], predicate=(col("user") > 20_000_000) | (col("user") < 0)), 1),
            (QueryPlan([count_star(), sum_("value")], keys=["i32", "service"], predicate=(col("i32") >= -300) & (col("value") < 10_000)), 2),
            (QueryPlan([count_star()], keys=["i16"], predicate=col("i16") != 0), 1),
            (QueryPlan([count_star(), sum_("value")], keys=keys2, predicate=col("ts") >= 0), 2),
            (QueryPlan([count_star(), sum_("value")], keys=keys2, predicate=col("ts") <= 2_000_000_000_000_000), 2),
            (QueryPlan([count_star(), min_("value")], keys=["user"], predicate=col("user").is_not_null() & (col("value") > -45_000)), 1),
            # string predicates on a dictionary of at most 64 values are a bit set of codes: on a key column,
            # on a column that is not a key, negated LIKE, IS NULL on the string column, mixed with integers
            (QueryPlan([count_star(), sum_("value")], keys=keys2, predicate=col("service").like("%-0%")), 2),
            (QueryPlan([count_star(), sum_("value")], keys=["i16"], predicate=col("service") == "svc-00"), 1),
            (QueryPlan([count_star(), max_("value")], keys=["service"], predicate=~col("service").like("svc-0_") & (col("value") > -30_000)), 1),
            (QueryPlan([count_star()], keys=keys2, predicate=(col("service") >= "svc-03") | col("service").is_null()), 2),
            (QueryPlan([count_star(), sum_("value")], predicate=(col("service") != "svc-01") & (col("i32") < 800)), 0),
            # ungrouped
            (QueryPlan([count_star(), sum_("value"), min_("value"), max_("u16"), avg("i32")],
                       predicate=(col("value") >= -10_000) & (col("value") < 30_000)), 0),
            (QueryPlan([sum_("value"), avg("value")], predicate=(col("i32") > -100) | (col("i16") == 2)), 0),
        ]
        for q, nk in cases:
            _run(gpu_ctx, oracle, t, table, q, nk, expect_fused=True)
        # shapes that must take the bitmap path: mixed AND / OR, a string leaf without a small dictionary, few
        # expected survivors,
        # more than four columns to load
        for q, nk in [
            (QueryPlan(aggs, keys=keys2, predicate=((col("value") > 0) & (col("i32") < 0)) | (col("u16") > 65_000)), 2),
            (QueryPlan(aggs, keys=keys2, predicate=(col("value") > 0) & col("url").like("%google%")), 2),
            (QueryPlan(aggs, keys=keys2, predicate=col("value") >= 49_000), 2),
            (QueryPlan([sum_("value"), sum_("i32"), sum_("u16"), sum_("user")], keys=keys2, predicate=col("i16") < 4), 2),
        ]:
            _run(gpu_ctx, oracle, t, table, q, nk, expect_fused=False)
    finally:
        table.close()


def test_fused_predicate_on_a_wide_key(gpu_ctx, oracle):
    """A key whose range needs the 64-bit key kernel: the leaf compares the raw 8-byte value."""
    from prism_amd.engine import Table

    rng = np.random.default_rng(9)
    n = 40_000
    wide = rng.integers(0, 400, n).astype(np.int64) * (2**62 // 400) - 2**61
    t = pa.table({"wide": pa.array(wide, mask=rng.random(n) < 0.05), "g": pa.array(rng.integers(0, 9, n).astype(np.int32)),
                  "value": pa.array(rng.integers(-100, 100, n).astype(np.int64))})
    table = Table(gpu_ctx, t.schema)
    table.append_table(t, 2)
    try:
        for q, nk in [(QueryPlan([count_star(), sum_("value")], keys=["wide"], predicate=col("wide") >= 0), 1),
                      (QueryPlan([count_star(), sum_("value")], keys=["wide"],
                                 predicate=(col("wide") < 2**60) & (col("wide") > -2**60) & (col("value") != 0) & (col("g") > 1)), 1),
                      (QueryPlan([count_star()], keys=["wide"], predicate=col("wide").is_null() | (col("wide") > 2**60)), 1)]:
            _run(gpu_ctx, oracle, t, table, q, nk, expect_fused=True)
    finally:
        table.close()


def test_fused_predicate_that_selects_nothing(gpu_ctx, oracle):
    """The kernel finds out that no row passes: the ungrouped plan still yields its one row
    (COUNT 0, SUM / MIN / AVG NULL although `value` has no NULLs and hence no validity
    counter), the grouped one yields no rows."""
    from prism_amd.engine import Plan, Table

    rng = np.random.default_rng(5)
    n = 20_000
    v = rng.integers(0, 1000, n).astype(np.int64)
    v[::2] = 5_000  # min/max make "value between 2000 and 4000" look selective enough; nothing is in there
    t = pa.table({"g": pa.array(rng.integers(0, 50, n).astype(np.int32)), "value": pa.array(v)})
    table = Table(gpu_ctx, t.schema)
    table.append_table(t, 2)
    pred = (col("value") >= 2_000) & (col("value") < 4_000)
    try:
        for q, nk in [(QueryPlan([count_star(), sum_("value"), min_("value"), avg("value")], predicate=pred), 0),
                      (QueryPlan([count_star(), sum_("value")], keys=["g"], predicate=pred), 1)]:
            for mode in (abi.MODE_SINGLE, abi.MODE_PARTIAL):
                plan = Plan(gpu_ctx, t.schema, q.with_mode(mode))
                try:
                    gpu_ctx.reset_stats()
                    got = plan.execute(table)
                    st = gpu_ctx.stats()
                    want = oracle.execute(t, q.with_mode(mode))
                finally:
                    plan.close()
                assert_tables_match(got, want, nk)
                assert st["rows_selected"] == 0
                assert not ({k for k, s in st["kernels"].items() if s["launches"]} & PRED_KERNELS)
    finally:
        table.close()


def test_fused_predicate_with_device_resident_state(gpu_ctx, oracle):
    """The multi-GPU path (aggregate -> merge -> finish) with a predicate the kernel evaluates."""
    from prism_amd.engine import Plan, Table

    t = helpers.random_table(30_000, seed=33)
    table = Table(gpu_ctx, t.schema)
    table.append_table(t, 4)
    from tests.test_gpu_state import _merge_on_one_gpu

    for keys, nk in [([date_bin(60_000, "ts"), "service"], 2), (["user"], 1)]:
        q = QueryPlan([count_star(), sum_("value"), max_("i32")], keys=keys,
                      predicate=(col("value") > -30_000) & (col("u16") < 50_000))
        plan = Plan(gpu_ctx, t.schema, q)
        try:
            gpu_ctx.reset_stats()
            got, _ = _merge_on_one_gpu(plan, table, [[0, 1], [2], [3]])
            st = gpu_ctx.stats()
            assert_tables_match(got, oracle.execute(t, q), nk)
            assert st["rows_selected"] == _selected_rows(oracle, t, q.predicate)
            assert not ({k for k, s in st["kernels"].items() if s["launches"]} & PRED_KERNELS)
        finally:
            plan.close()
    table.close()
