"""Streaming partitions (prism_gpu_partition_open / push_batch / finish) over several chunks.

The operator's input is a stream of RecordBatches per partition
(src/prism-query/src/providers/table_provider.rs:110-131 hands ParquetExec's output on batch by
batch).  A part gathers them into chunks; plans with a GROUP BY keep one group table per
partition in HBM and fold every chunk's groups into it on the device.  Chunks are made small
here (option "stream_chunk_rows") so that a few ten thousand rows already are many chunks;
every result must equal the CPU oracle's over the same rows, in SINGLE and in PARTIAL mode,
with the device-resident table and (option "stream_state" 0) with a PARTIAL batch per chunk."""
import numpy as np
import pyarrow as pa
import pytest

from tests import helpers
from tests.helpers import assert_tables_match, assert_topk_match

pytestmark = pytest.mark.gpu

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd.plan import QueryPlan, avg, col, count, count_star, date_bin, max_, min_, sum_  # noqa: E402


@pytest.fixture()
def oracle():
    from oracle import pyoracle
    return pyoracle


@pytest.fixture()
def small_chunks(gpu_ctx):
    gpu_ctx.set_option("stream_chunk_rows", 4096)
    yield gpu_ctx
    gpu_ctx.set_option("stream_chunk_rows", 1 << 20)
    gpu_ctx.set_option("stream_state", 1)


PLANS = [
    ("minute_service", QueryPlan([count_star(), sum_("value"), max_("value"), avg("i32"), min_("u16")], keys=[date_bin(60_000, "ts"), "service"]), 2),
    ("service_where", QueryPlan([count_star(), count("value"), sum_("i16")], keys=["service"], predicate=(col("i16") != 0) | col("url").like("%google%")), 1),
    ("user", QueryPlan([count_star(), sum_("value")], keys=["user"]), 1),
    ("user_i32", QueryPlan([count_star(), min_("value"), max_("ts")], keys=["user", "i32"]), 2),
    ("ts", QueryPlan([count_star()], keys=["ts"]), 1),
]


def _finished_chunk_states(ctx):
    """How often a device state was read out (a part that kept its groups in HBM does it once)."""
    k = ctx.stats()["kernels"].get("host:state_finish")
    return k["launches"] if k else 0


@pytest.mark.parametrize("state", [1, 0])
@pytest.mark.parametrize("pname", [p[0] for p in PLANS])
def test_chunks_of_a_partition_fold_into_one_result(small_chunks, oracle, pname, state):
    from prism_amd.engine import Plan

    ctx = small_chunks
    ctx.set_option("stream_state", state)
    _, q, n_keys = next(p for p in PLANS if p[0] == pname)
    t = helpers.random_table(50_000, seed=31)
    want = oracle.execute(t, q)
    batches = t.to_batches(max_chunksize=1000)
    plan = Plan(ctx, t.schema, q)
    try:
        ctx.reset_stats()
        got = plan.run_batches(0, batches)
        assert _finished_chunk_states(ctx) == (1 if state else 0)
    finally:
        plan.close()
    assert_tables_match(got, want, n_keys)
    # PARTIAL per partition (two parts over halves of the rows), then AggregateExec(Final)
    qp = q.with_mode(abi.MODE_PARTIAL)
    plan = Plan(ctx, t.schema, qp)
    try:
        half = len(batches) // 2
        states = [plan.run_batches(0, batches[:half]), plan.run_batches(1, batches[half:])]
        want_schema = oracle.output_schema(t.schema, qp)
        for s in states:
            assert s.schema.names == want_schema.names and [f.type for f in s.schema] == [f.type for f in want_schema]
        got = plan.merge(states)
    finally:
        plan.close()
    assert_tables_match(got, want, n_keys)
    assert_tables_match(oracle.merge(t.schema, states, q), want, n_keys)


def test_order_by_limit_over_a_streamed_partition(small_chunks, oracle):
    from prism_amd.engine import Plan

    t = helpers.random_table(40_000, seed=8)
    q = QueryPlan([count_star()], keys=["user"], order_by=1, descending=True, limit=10)
    plan = Plan(small_chunks, t.schema, q)
    try:
        got = plan.run_batches(0, t.to_batches(max_chunksize=2048))
    finally:
        plan.close()
    assert_topk_match(got, oracle.execute(t, q), 1)


def test_chunks_that_leave_the_first_chunks_key_range(small_chunks, oracle):
    """The part's table is laid out from its first chunk's statistics, widened.  Later chunks
    here break every assumption of that layout in turn -- timestamps far ahead and far behind,
    NULL keys where there were none, value NULLs where there were none, hundreds of new strings
    -- so the part has to set its table aside and start a new one several times; the result is
    the oracle's all the same."""
    from prism_amd.engine import Plan

    n = 60_000
    rng = np.random.default_rng(77)
    i = np.arange(n)
    ts = 1_700_000_000_000 + i * 5                                  # 5 ms apart: 20 s per chunk
    ts[20_000:30_000] += 40 * 86_400_000                            # forty days ahead
    ts[30_000:40_000] -= 300 * 86_400_000                           # most of a year behind
    svc = np.where(i < 25_000, i % 7, i % 900)                      # 7 names at first, 900 later
    t = pa.table({
        "ts": pa.array(ts, pa.timestamp("ms")),
        "service": pa.array([None if (k > 45_000 and k % 50 == 0) else f"service-{v:03d}" for k, v in enumerate(svc)]),
        "k": pa.array(np.where(i < 10_000, i % 10, rng.integers(-10**12, 10**12, n)), pa.int64(), mask=(i > 50_000) & (i % 3 == 0)),
        "v": pa.array(rng.integers(-1000, 1000, n), pa.int64(), mask=(i > 35_000) & (i % 11 == 0)),
    })
    ctx = small_chunks
    for q, n_keys in [(QueryPlan([count_star(), sum_("v"), count("v"), min_("v")], keys=[date_bin(1000, "ts"), "service"]), 2),
                      (QueryPlan([count_star(), max_("v")], keys=["k"]), 1),
                      (QueryPlan([count_star(), avg("v")], keys=["service", "k"], predicate=col("v") > -900), 2)]:
        plan = Plan(ctx, t.schema, q)
        try:
            ctx.reset_stats()
            got = plan.run_batches(0, t.to_batches(max_chunksize=1500))
            aside = ctx.stats()["kernels"].get("host:state_aside")
            assert aside and aside["launches"] >= 2  # (the last of them is finish's, which merges what was set aside)
        finally:
            plan.close()
        assert_tables_match(got, oracle.execute(t, q), n_keys)


def test_many_groups_per_chunk_stay_on_the_device(gpu_ctx, oracle):
    """A C3-shaped partition (src/prism-query/src/main.rs:57-60 groups by an id column): most rows
    of a chunk are groups of their own, the part's table is a hash table that grows as chunks
    arrive, and finish reads it once."""
    from prism_amd.engine import Plan

    n = 400_000
    rng = np.random.default_rng(3)
    ids = rng.integers(0, 150_000, n).astype(np.int64) * 1_000_003 + 17
    t = pa.table({"UserID": pa.array(ids), "w": pa.array(rng.integers(0, 100, n), pa.int32())})
    gpu_ctx.set_option("stream_chunk_rows", 50_000)
    try:
        for q, check in [(QueryPlan([count_star(), sum_("w")], keys=["UserID"]), assert_tables_match),
                         (QueryPlan([count_star()], keys=["UserID"], order_by=1, descending=True, limit=10), assert_topk_match)]:
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                gpu_ctx.reset_stats()
                got = plan.run_batches(0, t.to_batches(max_chunksize=8192))
                assert _finished_chunk_states(gpu_ctx) == 1
            finally:
                plan.close()
            check(got, oracle.execute(t, q), 1)
    finally:
        gpu_ctx.set_option("stream_chunk_rows", 1 << 20)


def test_a_string_key_outgrows_its_dictionary_mid_stream(small_chunks, oracle):
    """The part's table numbers a string key's values on the host until there are more than
    65536 of them; from then on the device numbers them per aggregation pass, and a group table
    that holds such codes is set aside as strings at once.  The groups before, across and after
    that point must add up to the oracle's."""
    from prism_amd.engine import Plan

    n = 120_000
    i = np.arange(n)
    t = pa.table({"u": pa.array([None if k % 1000 == 999 else f"user-{(k * 7919) % 90_001:06d}" for k in i]),
                  "v": pa.array((i % 13) - 6, pa.int64())})
    q = QueryPlan([count_star(), sum_("v")], keys=["u"])
    plan = Plan(small_chunks, t.schema, q)
    try:
        got = plan.run_batches(0, t.to_batches(max_chunksize=3000))
    finally:
        plan.close()
    assert_tables_match(got, oracle.execute(t, q), 1)
