"""GPU parity on the benchmark workloads at toy size: tables generated in HBM by the
device generators must aggregate to exactly what the CPU oracle computes over its own
generation of the same rows; plus the high-cardinality string (ClickBench URL-like)
path, where no dictionary exists and the raw-string kernels run."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from tests import helpers
from tests.helpers import assert_tables_match, assert_topk_match

pytestmark = pytest.mark.gpu

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd import workloads  # noqa: E402
from prism_amd.plan import QueryPlan, col, count_distinct, count_star, date_bin, max_, min_, sum_, avg  # noqa: E402


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle

    return pyoracle


def _both(gpu_ctx, oracle, wl, rows, parts, first=0, total=None, queries=None):
    from prism_amd.engine import Plan

    total = total or rows
    table = workloads.build_gpu_table(gpu_ctx, wl, first, rows, total, parts)
    st = oracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, first, rows, total, n_parts=parts)
    try:
        assert table.num_rows == rows
        for q, n_keys in queries or [(wl.query, wl.n_keys)]:
            plan = Plan(gpu_ctx, wl.schema, q)
            try:
                got = plan.execute(table)
            finally:
                plan.close()
            want, _ = st.execute(q)
            want = workloads.decode_dictionary_columns(wl, want)
            if q.limit is not None:
                assert_topk_match(got, want, q.order_by)
            else:
                assert_tables_match(got, want, n_keys)
    finally:
        st.close()
        table.close()


def test_c1_filter_count(gpu_ctx, oracle):
    wl = workloads.c1()
    _both(gpu_ctx, oracle, wl, 1_000_003, 10, queries=[
        (wl.query, 0),
        (QueryPlan([count_star(), sum_("AdvEngineID"), max_("AdvEngineID")], predicate=col("AdvEngineID") > 30), 0),
        (QueryPlan([count_star()], keys=["AdvEngineID"]), 1),
    ])


def test_c2_time_bucket_group_by(gpu_ctx, oracle):
    wl = workloads.c2()
    # a middle slice of a larger table, ragged partition sizes
    _both(gpu_ctx, oracle, wl, 700_001, 7, first=1_000_000, total=5_000_000, queries=[
        (wl.query, 2),
        (QueryPlan([count_star(), sum_("value"), max_("value"), min_("value"), avg("value")],
                   keys=[date_bin(1_000, "timestamp"), "service"], predicate=col("value") < 25_000), 2),
        (QueryPlan([count_star()], keys=[date_bin(3_600_000, "timestamp")]), 1),
        (QueryPlan([count_star(), sum_("value")], keys=["service"], predicate=col("value") >= 49_000), 1),
    ])


def test_c3_high_cardinality_group_by_topk(gpu_ctx, oracle):
    wl = workloads.c3(id_bits=18)
    _both(gpu_ctx, oracle, wl, 600_000, 6, queries=[
        (wl.query, 1),
        (QueryPlan([count_star()], keys=["UserID"]), 1),
        (QueryPlan([count_distinct("UserID")]), 0),
    ])


def test_c5_count_distinct_by_hour_service(gpu_ctx, oracle):
    wl = workloads.c5(n_services=16, user_bits=14)
    _both(gpu_ctx, oracle, wl, 500_000, 4, queries=[(wl.query, 2)])


def test_c2_partial_states_merge_across_shards(gpu_ctx, oracle):
    """Two 'ranks' each holding half of a table: PARTIAL per shard + Final merge equals
    the single pass over the whole table (the weak-scaling choreography of bench.py)."""
    from prism_amd.engine import Plan

    wl = workloads.c2()
    total = 400_000
    shards = [workloads.build_gpu_table(gpu_ctx, wl, r * total // 2, total // 2, total, 3) for r in range(2)]
    whole = workloads.build_gpu_table(gpu_ctx, wl, 0, total, total, 4)
    try:
        pp = Plan(gpu_ctx, wl.schema, wl.query.with_mode(abi.MODE_PARTIAL))
        ps = Plan(gpu_ctx, wl.schema, wl.query)
        try:
            merged = pp.merge([pp.execute(s) for s in shards])
            assert_tables_match(merged, ps.execute(whole), 2)
            assert sum(merged.column("COUNT(*)").to_pylist()) == total
        finally:
            pp.close()
            ps.close()
    finally:
        for t in shards + [whole]:
            t.close()


def _url_table(n: int, seed: int) -> pa.Table:
    rng = np.random.default_rng(seed)
    words = np.array(["google", "googl", "gogle", "example", "bing", "news", "shop", "a", "zz9", "GOOGLE"])
    urls = []
    for i in range(n):
        k = rng.integers(1, 6)
        urls.append("http://" + ".".join(words[rng.integers(0, len(words), k)]) + f"/p{i}")  # unique per row
    mask = rng.random(n) < 0.03
    return pa.table({
        "URL": pa.array(urls, mask=mask),
        "UserID": pa.array(rng.integers(0, 5000, n).astype(np.int64) * 1_000_003),
        "v": pa.array(rng.integers(-100, 100, n).astype(np.int32)),
    })


def test_c4_like_on_high_cardinality_strings(gpu_ctx, oracle):
    """> 65536 distinct URLs: the column has no dictionary, so LIKE '%google%' runs as the
    byte-stream search, other patterns and comparisons through the row-per-lane kernel,
    and the few survivors reach the GROUP BY through a compacted selection vector."""
    from prism_amd.engine import Plan, Table

    t = _url_table(150_000, seed=4)
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 3)
    try:
        queries = [
            (QueryPlan([count_star()], predicate=col("URL").like("%google%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").not_like("%google%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("%e%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("%google.google.google%")), 0),
            # every needle length class of the byte-stream search (1-3 bytes; 4; 5-6; 7 and more),
            # needles at the start and at the end of a row, and one that overlaps itself
            (QueryPlan([count_star()], predicate=col("URL").like("%go%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("%oog%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("%goog%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("%oogle%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("%e.bing.%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("%http://g%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("%/p77%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("%oo%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("%.google.goo%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("http://goo%/p1%")), 0),
            (QueryPlan([count_star()], predicate=col("URL").like("%/p12_")), 0),
            (QueryPlan([count_star()], predicate=col("URL").ilike("%GoOgLe.bing%")), 0),
            (QueryPlan([count_star()], predicate=col("URL") == "http://google/p7"), 0),
            (QueryPlan([count_star()], predicate=(col("URL") >= "http://g") & (col("URL") < "http://h")), 0),
            (QueryPlan([count_star(), sum_("v")], keys=["UserID"], predicate=col("URL").like("%google.bing%")), 1),
            (QueryPlan([count_star(), sum_("v")], keys=["UserID"], predicate=col("URL").like("%google%") & (col("v") > 0)), 1),
            (QueryPlan([count_star()], predicate=col("URL").is_null()), 0),
            (QueryPlan([count_star(), avg("v")], predicate=col("URL").like("%zz9.zz9.zz9.zz9%")), 0),
        ]
        for q, n_keys in queries:
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                assert_tables_match(plan.execute(g), oracle.execute(t, q), n_keys)
            finally:
                plan.close()
    finally:
        g.close()


def test_group_by_a_string_column_without_a_dictionary(gpu_ctx, oracle):
    """> 65536 distinct URLs: the host dictionary has given up, and the column's codes are
    numbered on the device (hash of the bytes, one representative row per hash, every row
    compared with its representative).  The reference groups by any Utf8 column
    (src/prism-query/src/providers/schema_provider.rs:96-107 maps them to plain strings).
    GROUP BY alone, beside another key, under a WHERE on the same column, as COUNT(DISTINCT),
    in PARTIAL mode with a merge, over some partitions only, and again after an append."""
    from prism_amd.engine import Plan, Table

    t = _url_table(150_000, seed=4)
    assert len(set(t.column("URL").to_pylist())) > 70_000
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 3)
    try:
        queries = [
            (QueryPlan([count_star()], keys=["URL"]), 1),
            (QueryPlan([count_star(), sum_("v"), max_("UserID")], keys=["URL"], predicate=col("URL").like("%google%")), 1),
            (QueryPlan([count_star(), avg("v")], keys=["UserID", "URL"], predicate=col("v") > 50), 2),
            (QueryPlan([count_distinct("URL")], keys=["v"]), 1),
            (QueryPlan([count_distinct("URL")]), 0),
        ]
        for q, n_keys in queries:
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                assert_tables_match(plan.execute(g), oracle.execute(t, q), n_keys)
            finally:
                plan.close()
        # PARTIAL per partition (the state's key column is Utf8), merged by the GPU and by the oracle
        q = QueryPlan([count_star(), min_("v")], keys=["URL"])
        qp = q.with_mode(abi.MODE_PARTIAL)
        plan = Plan(gpu_ctx, t.schema, qp)
        try:
            states = [plan.execute(g, [p]) for p in range(g.num_partitions)]
            got = plan.merge(states)
        finally:
            plan.close()
        want = oracle.execute(t, q)
        assert_tables_match(got, want, 1)
        assert_tables_match(oracle.merge(t.schema, states, q), want, 1)
        # rows arrive after the codes were numbered: the next query numbers them again
        more = _url_table(20_000, seed=5)
        g.append_table(more, 1)
        both = pa.concat_tables([t, more])
        q = QueryPlan([count_star(), sum_("v")], keys=["URL"])
        plan = Plan(gpu_ctx, t.schema, q)
        try:
            assert_tables_match(plan.execute(g), oracle.execute(both, q), 1)
        finally:
            plan.close()
    finally:
        g.close()


def test_keys_wider_than_64_bits(gpu_ctx, oracle):
    """GROUP BY over columns whose ranges need more than 64 bits together: two-word keys
    in the HBM table (also the shape of C5: a 63-bit id next to the group key)."""
    from prism_amd.engine import Plan, Table

    rng = np.random.default_rng(8)
    n = 60_000
    a = rng.integers(0, 300, n).astype(np.int64) * (2**62 // 300) - 2**61
    b = rng.integers(0, 200, n).astype(np.int64) * (2**40 // 200)
    t = pa.table({"a": pa.array(a, mask=rng.random(n) < 0.02), "b": pa.array(b), "c": pa.array(rng.integers(0, 9, n).astype(np.int32)),
                  "v": pa.array(rng.integers(-5, 5, n).astype(np.int64))})
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 2)
    try:
        for q, nk in [(QueryPlan([count_star(), sum_("v"), min_("v")], keys=["a", "b"]), 2),
                      (QueryPlan([count_star(), avg("v")], keys=["c", "a", "b"], predicate=col("v") != 0), 3),
                      (QueryPlan([count_distinct("a")], keys=["c"]), 1),
                      (QueryPlan([count_distinct("b")], keys=["a"]), 1),
                      (QueryPlan([count_star()], keys=["a", "b"], order_by=2, descending=True, limit=7), 2)]:
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                got, want = plan.execute(g), oracle.execute(t, q)
                if q.limit is not None:
                    assert_topk_match(got, want, q.order_by)
                else:
                    assert_tables_match(got, want, nk)
                if q.limit is None and not any(a.func == abi.AGG_COUNT_DISTINCT for a in q.aggs):
                    pp = Plan(gpu_ctx, t.schema, q.with_mode(abi.MODE_PARTIAL))
                    try:
                        states = [pp.execute(g, [p]) for p in range(2)]
                        assert_tables_match(pp.merge(states), want, nk)
                    finally:
                        pp.close()
            finally:
                plan.close()
    finally:
        g.close()


def test_count_distinct_over_many_groups_and_large_sets(gpu_ctx, oracle):
    """COUNT(DISTINCT x) GROUP BY g with 40 000 groups and sets of >= 2^16 entries: the second
    level counts pairs per range of groups in LDS, several passes; x wide (two-word set, only
    the keys are stored) or narrow (one-word set), with and without NULLs; PARTIAL states of
    the wide case merge to the same answer."""
    from prism_amd.engine import Plan, Table

    rng = np.random.default_rng(15)
    n = 300_000
    g_col = rng.integers(0, 40_000, n).astype(np.int32)
    wide = rng.integers(0, 50, n).astype(np.int64) * (2**62 // 50) - 2**61 + rng.integers(0, 3, n)
    narrow = rng.integers(0, 1 << 20, n).astype(np.int64)
    t = pa.table({"g": pa.array(g_col, mask=rng.random(n) < 0.01),
                  "wide_nulls": pa.array(wide, mask=rng.random(n) < 0.1), "wide": pa.array(wide),
                  "narrow_nulls": pa.array(narrow, mask=rng.random(n) < 0.1), "narrow": pa.array(narrow)})
    table = Table(gpu_ctx, t.schema)
    table.append_table(t, 3)
    try:
        for x in ("wide_nulls", "wide", "narrow_nulls", "narrow"):
            q = QueryPlan([count_distinct(x)], keys=["g"])
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                want = oracle.execute(t, q)
                assert_tables_match(plan.execute(table), want, 1)
            finally:
                plan.close()
            pp = Plan(gpu_ctx, t.schema, q.with_mode(abi.MODE_PARTIAL))
            try:
                assert_tables_match(pp.merge([pp.execute(table, [p]) for p in range(3)]), want, 1)
            finally:
                pp.close()
    finally:
        table.close()


def test_c4_generated_urls_like_and_compaction_variant(gpu_ctx, oracle):
    """C4 on device-generated URL-like strings (no dictionary: the raw-string kernels run),
    a middle slice of a larger table with ragged partitions; plus near-miss patterns and
    the GROUP BY variant that aggregates only the matching rows."""
    wl = workloads.c4()
    _both(gpu_ctx, oracle, wl, 300_007, 7, first=123_456, total=1_000_000, queries=[
        (wl.query, 0),
        (QueryPlan([count_star()], predicate=col("URL").like("%googl%")), 0),
        (QueryPlan([count_star()], predicate=col("URL").like("%gogle%") | col("URL").like("%google%")), 0),
        (QueryPlan([count_star()], predicate=~col("URL").like("%google%")), 0),
        (QueryPlan([count_star()], predicate=col("URL").like("%g_ogle%")), 0),
        (QueryPlan([count_star()], predicate=col("URL").like("a%")), 0),
        # more occurrences than rows: the position list overflows and the per-segment search runs
        (QueryPlan([count_star()], predicate=col("URL").like("%a%")), 0),
        (QueryPlan([count_star()], predicate=col("URL").like("%/%") & ~col("URL").like("%google%")), 0),
        (workloads.c4g().query, 1),
    ])


def test_full_size_c2_invariants(gpu_ctx):
    """The headline configuration at its full size (10^9 rows, 64 partitions), checked through
    properties that need no oracle: every row lands in exactly one group; the grouped sums and
    maxima fold to the ungrouped ones; PARTIAL states of two halves merge to the same rows as
    one pass; a filter and its complement split the rows."""
    from prism_amd.engine import Plan

    wl = workloads.c2()
    rows, parts = 1_000_000_000, 64
    table = workloads.build_gpu_table(gpu_ctx, wl, 0, rows, rows, parts)
    try:
        def run(q, partitions=None, mode=None):
            plan = Plan(gpu_ctx, wl.schema, q if mode is None else q.with_mode(mode))
            try:
                return plan.execute(table, partitions)
            finally:
                plan.close()

        full = run(wl.query)
        counts = np.array(full.column("COUNT(*)").to_pylist(), dtype=np.int64)
        sums = np.array(full.column("SUM(value)").to_pylist(), dtype=np.int64)
        maxs = np.array(full.column("MAX(value)").to_pylist(), dtype=np.int64)
        assert counts.sum() == rows and (counts > 0).all()
        assert full.num_rows <= 1440 * 64 + 64
        flat = run(QueryPlan([count_star(), sum_("value"), max_("value"), min_("value")]))
        assert flat.column(0).to_pylist() == [rows]
        assert flat.column(1).to_pylist() == [int(sums.sum())]
        assert flat.column(2).to_pylist() == [int(maxs.max())]
        assert 0 <= flat.column(3).to_pylist()[0] <= 5
        # two halves -> PARTIAL -> merge == one pass (order-insensitive, bit exact)
        plan = Plan(gpu_ctx, wl.schema, wl.query.with_mode(abi.MODE_PARTIAL))
        try:
            merged = plan.merge([plan.execute(table, range(0, 32)), plan.execute(table, range(32, 64))])
        finally:
            plan.close()
        assert_tables_match(merged, full, 2)
        # a predicate and its complement
        a = run(QueryPlan([count_star()], predicate=col("value") >= 12_345)).column(0).to_pylist()[0]
        b = run(QueryPlan([count_star()], predicate=col("value") < 12_345)).column(0).to_pylist()[0]
        assert a + b == rows and abs(a / rows - (1 - 12_345 / 50_000)) < 1e-3
    finally:
        table.close()


def test_large_c5_same_counts_on_every_route(oracle):
    """C5's query over 2.5e8 rows (large enough for the pairs to be partitioned twice), checked
    without the CPU: the distinct counts are the same whether the second partition level keeps
    one-word or two-word records, whether the scatter pass runs workgroups of 512 or of 1024
    threads, and when nothing is partitioned at all (one set of pairs in HBM, a different
    algorithm altogether); no group has more distinct users than rows; and a sample of the
    table agrees with the CPU checker."""
    from prism_amd.engine import Context, Plan

    wl = workloads.c5()
    rows, parts = 250_000_000, 16
    ctx = Context(0)
    table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, parts)

    def run(q, partitions=None):
        plan = Plan(ctx, wl.schema, q)
        try:
            ctx.reset_stats()
            out = plan.execute(table, partitions)
            return out, {k for k, v in ctx.stats()["kernels"].items() if v["launches"]}
        finally:
            plan.close()

    def sort(t):
        return t.take(pc.sort_indices(t, [(t.schema.names[0], "ascending"), (t.schema.names[1], "ascending")]))

    try:
        base, ran = run(wl.query)
        assert "part2_agg" in ran
        base = sort(base)
        # (part2_lines: the one-level scatter by staging lines, the default here -- rows in time order, a time
        # bucket runs through hundreds of tiles -- against the tile-by-tile one; part2_direct: against two levels)
        for name, value in (("part2_compact", 0), ("part_half_blocks", 0), ("part2_lines", 0), ("part2_direct", 0)):
            ctx.set_option(name, value)
            other, ran = run(wl.query)
            ctx.set_option(name, 1)
            assert "part2_agg" in ran
            assert sort(other).equals(base), name
        ctx.set_option("part_min_rows", 1 << 40)  # nothing is partitioned: the pairs go into one table in HBM
        other, ran = run(wl.query)
        ctx.set_option("part_min_rows", 4_194_304)
        assert "part2_agg" not in ran and sort(other).equals(base)
        counts, _ = run(QueryPlan([count_star()], keys=wl.query.keys))
        counts = sort(counts)
        assert counts.column(0).equals(base.column(0)) and counts.column(1).equals(base.column(1))
        distinct = np.array(base.column(2).to_pylist(), dtype=np.int64)
        n = np.array(counts.column(2).to_pylist(), dtype=np.int64)
        assert n.sum() == rows and (distinct >= 1).all() and (distinct <= n).all()
        # the first partition against the CPU checker
        got, _ = run(wl.query, range(1))
        st = oracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, workloads.partition_bounds(rows, parts, 0)[1], rows,
                               n_parts=8, threads=8)
        want, _ = st.execute(wl.query, threads=8)
        st.close()
        assert_tables_match(got, workloads.decode_dictionary_columns(wl, want), 2)
    finally:
        table.close()
        ctx.close()


def test_full_size_c1_and_c3_invariants(gpu_ctx):
    """C1 and C3 at their full sizes (10^8 rows): complementary filters split the rows; the
    top-10 of the high-cardinality GROUP BY does not depend on how the partitions are cut,
    and its counts are bounded by what the unlimited result holds."""
    from prism_amd.engine import Plan

    wl = workloads.c1()
    rows = 100_000_000
    table = workloads.build_gpu_table(gpu_ctx, wl, 0, rows, rows, 10)
    try:
        got = {}
        for name, pred in [("ne", col("AdvEngineID") != 0), ("eq", col("AdvEngineID") == 0), ("gt", col("AdvEngineID") > 0)]:
            plan = Plan(gpu_ctx, wl.schema, QueryPlan([count_star()], predicate=pred))
            try:
                got[name] = plan.execute(table).column(0).to_pylist()[0]
            finally:
                plan.close()
        assert got["ne"] + got["eq"] == rows and got["ne"] == got["gt"]
        assert abs(got["ne"] / rows - 0.0063) < 2e-4
        # the few survivors of a selective filter reach the GROUP BY through a selection vector
        # compacted over 24 415 tiles (the scan of the tile counts runs in several workgroups
        # beyond 8192 tiles): the groups' counts add up to what the filter alone counts
        gpu_ctx.reset_stats()
        plan = Plan(gpu_ctx, wl.schema, QueryPlan([count_star()], predicate=col("AdvEngineID") > 57))
        try:
            few = plan.execute(table).column(0).to_pylist()[0]
        finally:
            plan.close()
        gpu_ctx.reset_stats()
        plan = Plan(gpu_ctx, wl.schema, QueryPlan([count_star(), max_("AdvEngineID")], keys=["AdvEngineID"],
                                                   predicate=col("AdvEngineID") > 57))
        try:
            by = plan.execute(table)
        finally:
            plan.close()
        assert gpu_ctx.stats()["kernels"]["compact_selvec"]["launches"] >= 1
        keys, counts = by.column(0).to_pylist(), by.column(1).to_pylist()
        assert sorted(keys) == [58, 59, 60] and sum(counts) == few and by.column(2).to_pylist() == keys
        assert 0 < few < rows * 0.0063 * 0.1
    finally:
        table.close()

    wl = workloads.c3()
    tops = []
    for parts in (100, 7):
        table = workloads.build_gpu_table(gpu_ctx, wl, 0, rows, rows, parts)
        try:
            plan = Plan(gpu_ctx, wl.schema, wl.query)
            try:
                tops.append(plan.execute(table))
            finally:
                plan.close()
            if parts == 7:
                plan = Plan(gpu_ctx, wl.schema, QueryPlan([count_star()], keys=["UserID"], order_by=1, descending=True, limit=1000))
                try:
                    top1000 = plan.execute(table)
                finally:
                    plan.close()
        finally:
            table.close()
    assert tops[0].column(1).to_pylist() == tops[1].column(1).to_pylist()
    assert tops[0].column(1).to_pylist() == top1000.column(1).to_pylist()[:10]
    c = top1000.column(1).to_pylist()
    assert c == sorted(c, reverse=True) and len(set(top1000.column(0).to_pylist())) == 1000


def test_hash_table_growth_and_frozen_lds_at_scale(gpu_ctx, oracle):
    """Hundreds of thousands of groups with a far too small groups_hint: the HBM hash table overflows and
    is regrown several times (each failed pass aborts early), the LDS tables freeze, and the
    result still matches the oracle exactly; then the same plan again (the learned capacity)."""
    from prism_amd.engine import Plan

    wl = workloads.c3(id_bits=22)
    rows = 4_000_000
    table = workloads.build_gpu_table(gpu_ctx, wl, 0, rows, rows, 5)
    st = oracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, rows, rows, n_parts=5)
    try:
        q = QueryPlan([count_star(), max_("UserID")], keys=["UserID"], groups_hint=500)
        want, _ = st.execute(q)
        assert want.num_rows > 500_000
        plan = Plan(gpu_ctx, wl.schema, q)
        try:
            for _ in range(2):
                assert_tables_match(plan.execute(table), want, 1)
        finally:
            plan.close()
    finally:
        st.close()
        table.close()
