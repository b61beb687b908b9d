"""GPU parity: every plan runs through the C ABI of libprism_gpu.so on the MI355X
and is compared with the CPU oracle on the same input (bit-exact for integers,
keys and counts; 1e-9 relative for Float64 AVG)."""
import numpy as np
import pyarrow as pa
import pytest

from tests import helpers
from tests.helpers import assert_tables_match, assert_topk_match

pytestmark = pytest.mark.gpu

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd.plan import QueryPlan, col, count_star, date_bin, max_, min_, sum_, avg, count_distinct  # noqa: E402


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle

    return pyoracle


@pytest.fixture(scope="module")
def rand_tables(gpu_ctx):
    from prism_amd.engine import Table

    out = {}
    for name, n, nulls, parts in [("small", 1000, True, 1), ("mid", 50_000, True, 3), ("nonull", 20_000, False, 2)]:
        t = helpers.random_table(n, seed=len(name) + n, nulls=nulls)
        g = Table(gpu_ctx, t.schema)
        g.append_table(t, parts)
        out[name] = (t, g)
    yield out
    for _, g in out.values():
        g.close()


PLANS = helpers.plan_catalogue()


@pytest.mark.parametrize("tname", ["small", "mid", "nonull"])
@pytest.mark.parametrize("pname", [p[0] for p in PLANS])
def test_single_mode(gpu_ctx, oracle, rand_tables, tname, pname):
    from prism_amd.engine import Plan

    _, q, n_keys = next(p for p in PLANS if p[0] == pname)
    t, g = rand_tables[tname]
    plan = Plan(gpu_ctx, t.schema, q)
    try:
        got = plan.execute(g)
    finally:
        plan.close()
    want = oracle.execute(t, q)
    assert_tables_match(got, want, n_keys)


@pytest.mark.parametrize("pname", [p[0] for p in PLANS if p[0].startswith(("by_", "global", "distinct", "count_all"))])
def test_partial_then_merge(gpu_ctx, oracle, rand_tables, pname):
    """PARTIAL per partition on the GPU, merged by the GPU (prism_gpu_merge) and by
    the oracle's Final stage; both must equal the oracle's single-pass result."""
    from prism_amd.engine import Plan

    _, q, n_keys = next(p for p in PLANS if p[0] == pname)
    t, g = rand_tables["mid"]
    qp = q.with_mode(abi.MODE_PARTIAL)
    plan = Plan(gpu_ctx, t.schema, qp)
    try:
        states = [plan.execute(g, [p]) for p in range(g.num_partitions)]
        want_partial_schema = oracle.output_schema(t.schema, qp)
        for s in states:
            assert s.schema.names == want_partial_schema.names
            assert [f.type for f in s.schema] == [f.type for f in want_partial_schema]
        got = plan.merge(states)
    finally:
        plan.close()
    want = oracle.execute(t, q)
    assert_tables_match(got, want, n_keys)
    via_oracle = oracle.merge(t.schema, states, q)
    assert_tables_match(via_oracle, want, n_keys)


def test_demo_log_known_answers(gpu_ctx, oracle):
    """The reference's fixture: MIN/MAX(timestamp) is pinned by
    src/prism-ingest/src/ingest.rs:224-226; the rest are derived."""
    from prism_amd.engine import Plan, Table

    t = helpers.demo_table()
    g = Table(gpu_ctx, t.schema)
    g.append_table(t)
    try:
        def run(q):
            p = Plan(gpu_ctx, t.schema, q)
            try:
                return p.execute(g)
            finally:
                p.close()

        r = run(QueryPlan([max_("timestamp"), min_("timestamp")]))
        assert r.column(0).cast(pa.int64()).to_pylist() == [1698000995523]
        assert r.column(1).cast(pa.int64()).to_pylist() == [1698000992225]
        r = run(QueryPlan([count_star(), sum_("bytes"), max_("bytes")], keys=["method"]))
        got = {m: (c, s, x) for m, c, s, x in helpers.rows_of(r)}
        assert got == {"DELETE": (13, 347590, 48600), "GET": (2, 83654, 44428), "HEAD": (4, 78885, 28663),
                       "OPTION": (4, 111733, 49236), "PATCH": (3, 28387, 14181), "POST": (5, 108229, 46657),
                       "PUT": (3, 85000, 45503)}
        assert run(QueryPlan([count_star()], predicate=col("status") != "200")).column(0).to_pylist() == [30]
        assert run(QueryPlan([count_distinct("user-identifier")])).column(0).to_pylist() == [7]
        assert run(QueryPlan([count_star()], predicate=col("request").like("%admin%"))).column(0).to_pylist() == [3]
        r = run(QueryPlan([count_star()], keys=[date_bin(60_000, "timestamp")]))
        assert r.column(0).cast(pa.int64()).to_pylist() == [1698000960000] and r.column(1).to_pylist() == [34]
        # prism-ql's two plan shapes (src/prism-ql/src/lower.rs:83-95)
        assert run(QueryPlan([count_star()])).column(0).to_pylist() == [34]
        for q in [QueryPlan([count_star()], keys=["method"]), QueryPlan([avg("bytes")], keys=["status", "protocol"])]:
            assert_tables_match(run(q), oracle.execute(t, q), len(q.keys))
    finally:
        g.close()


def test_order_by_limit(gpu_ctx, oracle, rand_tables):
    from prism_amd.engine import Plan

    t, g = rand_tables["mid"]
    q = QueryPlan([count_star()], keys=["user"], order_by=1, descending=True, limit=10)
    plan = Plan(gpu_ctx, t.schema, q)
    try:
        got = plan.execute(g)
    finally:
        plan.close()
    want = oracle.execute(t, q)
    assert_topk_match(got, want, 1)
    full = {k: c for k, c in helpers.rows_of(oracle.execute(t, QueryPlan([count_star()], keys=["user"])))}
    for k, c in helpers.rows_of(got):
        assert full[k] == c


def test_streaming_partition_api(gpu_ctx, oracle):
    """partition_open / push_batch / finish with small (8192-row) batches."""
    from prism_amd.engine import Plan

    t = helpers.random_table(30_000, seed=5)
    q = QueryPlan([count_star(), sum_("value")], keys=[date_bin(60_000, "ts"), "service"], predicate=col("i16") != 0)
    plan = Plan(gpu_ctx, t.schema, q)
    try:
        got = plan.run_batches(0, t.to_batches(max_chunksize=8192))
    finally:
        plan.close()
    assert_tables_match(got, oracle.execute(t, q), 2)


def test_empty_and_ragged_inputs(gpu_ctx, oracle):
    from prism_amd.engine import Plan, Table

    t = helpers.random_table(5000, seed=9)
    empty = t.slice(0, 0)
    for q, n_keys in [(QueryPlan([count_star(), sum_("value"), avg("value")]), 0),
                      (QueryPlan([count_star()], keys=["service"]), 1),
                      (QueryPlan([count_distinct("user")]), 0)]:
        plan = Plan(gpu_ctx, t.schema, q)
        g = Table(gpu_ctx, t.schema)
        try:
            got = plan.execute(g)  # no partitions at all
            assert_tables_match(got, oracle.execute(empty, q), n_keys)
            # ragged: partitions of 1, 63, 64, 65, 4095, 4097 rows and sliced (offset) batches
            off = 0
            for pid, n in enumerate([1, 63, 64, 65, 4095, 4097 - 4095 + 700]):
                g.append(pid, t.slice(off, n).combine_chunks().to_batches()[0])
                off += n
            sl = t.slice(3, off - 3)  # note: different rows than what was appended above
            got = plan.execute(g)
            assert_tables_match(got, oracle.execute(t.slice(0, off), q), n_keys)
            # a filter that selects nothing
            qn = QueryPlan(q.aggs, q.keys, predicate=col("value") > 10**9)
            pn = Plan(gpu_ctx, t.schema, qn)
            try:
                assert_tables_match(pn.execute(g), oracle.execute(t.slice(0, off), qn), n_keys)
            finally:
                pn.close()
            del sl
        finally:
            plan.close()
            g.close()


def test_sliced_batches_with_offsets(gpu_ctx, oracle):
    """Arrow arrays with a non-zero offset (validity bit offsets not byte aligned)."""
    from prism_amd.engine import Plan, Table

    t = helpers.random_table(3000, seed=11).combine_chunks()
    q = QueryPlan([count_star(), sum_("value"), max_("i32")], keys=["service"], predicate=col("url").like("%bing%"))
    g = Table(gpu_ctx, t.schema)
    plan = Plan(gpu_ctx, t.schema, q)
    try:
        b = t.to_batches()[0]
        g.append(0, b.slice(13, 1000))
        g.append(1, b.slice(1013, 1987))
        assert_tables_match(plan.execute(g), oracle.execute(t.slice(13, 2987), q), 1)
    finally:
        g.close()
    # the same rows as struct arrays whose own offset selects them (the children keep all their
    # rows; one child is itself sliced, so both offsets add up)
    import ctypes as C

    from prism_amd.engine import _check

    g = Table(gpu_ctx, t.schema)
    try:
        cols = [b.column(i) for i in range(b.num_columns)]
        cols[2] = pa.concat_arrays([cols[2].slice(0, 5), cols[2]]).slice(5)  # child offset 5
        st = pa.StructArray.from_arrays(cols, fields=list(t.schema))
        for part, (lo, n) in enumerate([(13, 1000), (1013, 1987)]):
            arr = abi.ArrowArray()
            st.slice(lo, n)._export_to_c(C.addressof(arr))
            assert arr.offset == lo and arr.length == n
            _check(g._lib.prism_gpu_table_append(g._h, part, C.byref(arr)))
        assert_tables_match(plan.execute(g), oracle.execute(t.slice(13, 2987), q), 1)
    finally:
        plan.close()
        g.close()


def test_dictionary_encoded_input(gpu_ctx, oracle):
    from prism_amd.engine import Plan, Table

    t = helpers.random_table(4000, seed=12)
    enc = t.set_column(t.schema.get_field_index("service"), "service", t.column("service").dictionary_encode())
    q = QueryPlan([count_star(), min_("value")], keys=["service"], predicate=col("service") != "svc-00")
    g = Table(gpu_ctx, t.schema)
    plan = Plan(gpu_ctx, t.schema, q)
    try:
        for i, b in enumerate(enc.combine_chunks().to_batches(max_chunksize=1500)):
            g.append(i, b)
        assert_tables_match(plan.execute(g), oracle.execute(t, q), 1)
    finally:
        plan.close()
        g.close()


@pytest.mark.parametrize("on_device", [1, 0])
def test_dictionary_encoded_input_is_expanded_to_the_same_column(gpu_ctx, oracle, on_device):
    """Dictionary arrays (Int32 indices, NULL rows, several batches with dictionaries of their
    own) become the same offsets / bytes / codes whether kernels expand them (the default) or the
    host does: a column of few values is read through its codes (GROUP BY, =), one of more than
    65536 values has no codes and is read through its bytes (LIKE over the raw strings)."""
    import numpy as np

    from prism_amd.engine import Plan, Table

    n = 90_000
    rng = np.random.default_rng(5)
    ids = rng.integers(0, 70_000, n)
    urls = pa.array([None if i % 97 == 0 else f"http://h{v % 13}.example/{'google' if v % 7 == 0 else 'q'}/{v}" for i, v in enumerate(ids)])
    svc = pa.array([None if i % 41 == 0 else f"svc-{v % 23:02d}" for i, v in enumerate(ids)])
    t = pa.table({"url": urls, "service": svc, "value": pa.array(rng.integers(-50, 50, n), pa.int64())})
    gpu_ctx.set_option("dict_expand_device", on_device)
    g = Table(gpu_ctx, t.schema)
    try:
        at = 0
        for i, rows in enumerate([1, 40_000, 4097, n - 44_098]):  # (every slice is encoded by itself: dictionaries differ)
            part = t.slice(at, rows)
            at += rows
            enc = pa.table({"url": part.column("url").combine_chunks().dictionary_encode(),
                            "service": part.column("service").combine_chunks().dictionary_encode(), "value": part.column("value")})
            g.append(i % 2, enc.to_batches()[0])
        for q, nk in [(QueryPlan([count_star(), sum_("value")], keys=["service"], predicate=col("url").like("%google%")), 1),
                      (QueryPlan([count_star()], predicate=col("url").like("http://h3.%") & (col("service") == "svc-03")), 0),
                      (QueryPlan([count_star(), min_("value")], predicate=col("url").is_null() | col("service").is_null()), 0)]:
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                assert_tables_match(plan.execute(g), oracle.execute(t, q), nk)
            finally:
                plan.close()
    finally:
        gpu_ctx.set_option("dict_expand_device", 1)
        g.close()


def test_wrapping_sum_and_extremes(gpu_ctx, oracle):
    from prism_amd.engine import Plan, Table

    big = 2**62
    t = pa.table({"k": pa.array([1, 1, 1, 2, 2, None], pa.int32()),
                  "v": pa.array([big, big, big, -2**63, 2**63 - 1, 5], pa.int64())})
    g = Table(gpu_ctx, t.schema)
    g.append_table(t)
    try:
        for q, nk in [(QueryPlan([sum_("v"), min_("v"), max_("v"), avg("v")], keys=["k"]), 1),
                      (QueryPlan([sum_("v"), min_("v"), max_("v")]), 0),
                      (QueryPlan([count_star()], keys=["v"]), 1)]:
            p = Plan(gpu_ctx, t.schema, q)
            try:
                assert_tables_match(p.execute(g), oracle.execute(t, q), nk)
            finally:
                p.close()
    finally:
        g.close()


def test_date_bin_before_origin(gpu_ctx, oracle):
    """Timestamps before the origin, including exact negative multiples of the
    stride.  The reference's behaviour here is unpinned by its own tests; the GPU
    path is only required to agree with the oracle's restatement."""
    from prism_amd.engine import Plan, Table

    ts = np.array([-120_000, -60_001, -60_000, -59_999, -1, 0, 1, 59_999, 60_000, 119_999, -180_000], dtype=np.int64)
    t = pa.table({"ts": pa.array(ts, pa.timestamp("ms")), "v": pa.array(np.arange(len(ts)), pa.int64())})
    g = Table(gpu_ctx, t.schema)
    g.append_table(t)
    try:
        for stride, origin in [(60_000, 0), (60_000, 30_000), (1, 0), (7, -3), (1000, 10**6)]:
            q = QueryPlan([count_star(), sum_("v")], keys=[date_bin(stride, "ts", origin_ms=origin)])
            p = Plan(gpu_ctx, t.schema, q)
            try:
                assert_tables_match(p.execute(g), oracle.execute(t, q), 1)
            finally:
                p.close()
    finally:
        g.close()


def test_unsupported_and_invalid_plans(gpu_ctx):
    from prism_amd.engine import Plan, PrismGpuError

    t = helpers.random_table(10, seed=1)
    with pytest.raises(PrismGpuError) as e:
        Plan(gpu_ctx, t.schema, QueryPlan([sum_("url")]))
    assert e.value.code == abi.EUNSUPPORTED
    with pytest.raises(PrismGpuError) as e:
        Plan(gpu_ctx, t.schema, QueryPlan([count_star()], keys=[date_bin(0, "ts")]))
    assert e.value.code == abi.EINVAL
    # COUNT(DISTINCT) beside other aggregates, or over several columns: the PARTIAL state of each is a List of its column's type
    for aggs in ([count_distinct("user"), count_star()], [count_distinct("user"), count_distinct("i16")]):
        p = Plan(gpu_ctx, t.schema, QueryPlan(aggs).with_mode(abi.MODE_PARTIAL))
        try:
            assert p.output_schema.field(0).type == pa.list_(pa.int64())
            assert p.output_schema.field(1).type == (pa.int64() if aggs[1].func == abi.AGG_COUNT_STAR else pa.list_(pa.int16()))
        finally:
            p.close()
    with pytest.raises(KeyError):
        Plan(gpu_ctx, t.schema, QueryPlan([count_star()], keys=["something"]))


def test_chunks_outside_the_predicate_range_are_pruned(gpu_ctx, oracle):
    """Zone maps: a WHERE on the (sorted) timestamp drops whole chunks before any kernel
    runs; the answer is the same as scanning everything."""
    from prism_amd.engine import Plan, Table

    t = helpers.random_table(40_000, seed=3, nulls=True)  # ts ascending over 3 hours
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 8)
    ts = t.column("ts").cast(pa.int64()).to_numpy()
    lo, hi = int(ts[len(ts) // 4]), int(ts[len(ts) // 2])
    try:
        for pred, expect_pruned in [((col("ts") >= lo) & (col("ts") < hi), True),
                                    ((col("ts") < lo) | (col("ts") >= hi), True),
                                    (col("ts") > int(ts[-1]), True),
                                    (col("i32") > 2000, True),        # above every value: all chunks go
                                    (col("i32") > -2000, False)]:
            q = QueryPlan([count_star(), sum_("value")], keys=["service"], predicate=pred)
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                before = gpu_ctx.stats()["chunks_pruned"]
                got = plan.execute(g)
                pruned = gpu_ctx.stats()["chunks_pruned"] - before
            finally:
                plan.close()
            assert_tables_match(got, oracle.execute(t, q), 1)
            assert (pruned > 0) == expect_pruned, (pred, pruned)
    finally:
        g.close()


def test_results_stay_valid_while_held_and_after_the_context_is_gone(oracle):
    """GROUP BY results point into page-locked blocks leased from a pool: many results held
    at once (more than the pool keeps), interleaved with further queries, must all keep their
    contents, also after the plan, the table and the context have been destroyed."""
    import gc

    from prism_amd.engine import Context, Plan, Table

    t = helpers.random_table(40_000, seed=21)
    q = QueryPlan([count_star(), sum_("value"), avg("value"), max_("i32")], keys=[date_bin(1_000, "ts"), "service"])
    want = oracle.execute(t, q)
    ctx = Context(0)
    table = Table(ctx, t.schema)
    table.append_table(t, 3)
    plan = Plan(ctx, t.schema, q)
    held = [plan.execute(table) for _ in range(12)]
    del held[3:7]  # blocks go back to the pool and are handed out again
    gc.collect()
    held += [plan.execute(table) for _ in range(6)]
    plan.close()
    table.close()
    ctx.close()
    for got in held:
        assert_tables_match(got, want, 2)
    del held
    gc.collect()
