"""GPU parity of the partitioned GROUP BY (rows partitioned by key hash in HBM, buckets
aggregated in LDS): the same plans and the same oracle as the other parity tests, with
the context told to partition whatever can be partitioned, plus the cases only this path
has: heavy hitters whose bucket is aggregated in slices, LDS tables that overflow into the
HBM table, a result list that turns out too short, two-word keys, and the streaming
kernel handing a plan over when it finds its rows scattered."""
import pytest

from tests import helpers
from tests.helpers import assert_tables_match, assert_topk_match

pytestmark = pytest.mark.gpu

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd import workloads  # noqa: E402
from prism_amd.plan import QueryPlan, avg, col, count, count_distinct, count_star, date_bin, max_, min_, sum_  # noqa: E402


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle

    return pyoracle


@pytest.fixture(scope="module")
def part_ctx():
    """A context of its own, so that the options do not leak into the other tests."""
    from prism_amd.engine import Context

    ctx = Context(0)
    ctx.set_option("part_min_rows", 0)
    ctx.set_option("part_force", 1)
    yield ctx
    ctx.close()


@pytest.fixture(scope="module")
def rand_tables(part_ctx):
    from prism_amd.engine import Table

    out = {}
    for name, n, nulls, parts in [("small", 1000, True, 1), ("mid", 50_000, True, 3), ("nonull", 20_000, False, 2)]:
        t = helpers.random_table(n, seed=len(name) + n, nulls=nulls)
        g = Table(part_ctx, t.schema)
        g.append_table(t, parts)
        out[name] = (t, g)
    yield out
    for _, g in out.values():
        g.close()


PLANS = [p for p in helpers.plan_catalogue() if p[0].startswith(("by_", "distinct_"))]


def _kernels(ctx):
    return {k for k, v in ctx.stats()["kernels"].items() if v["launches"]}


@pytest.mark.parametrize("tname", ["small", "mid", "nonull"])
@pytest.mark.parametrize("pname", [p[0] for p in PLANS])
def test_catalogue_partitioned(part_ctx, oracle, rand_tables, tname, pname):
    from prism_amd.engine import Plan

    _, q, n_keys = next(p for p in PLANS if p[0] == pname)
    t, g = rand_tables[tname]
    plan = Plan(part_ctx, t.schema, q)
    try:
        got = plan.execute(g)
        again = plan.execute(g)  # sized from what the first run learned
    finally:
        plan.close()
    want = oracle.execute(t, q)
    assert_tables_match(got, want, n_keys)
    assert_tables_match(again, want, n_keys)


def test_partitioned_path_really_runs(part_ctx, oracle, rand_tables):
    from prism_amd.engine import Plan

    t, g = rand_tables["mid"]
    q = QueryPlan([count_star(), sum_("value"), min_("value")], keys=["user"])
    part_ctx.reset_stats()
    plan = Plan(part_ctx, t.schema, q)
    try:
        got = plan.execute(g)
    finally:
        plan.close()
    assert {"part_hist", "part_scatter", "part_agg"} <= _kernels(part_ctx)
    assert_tables_match(got, oracle.execute(t, q), 1)


@pytest.mark.parametrize("pname", ["by_user", "by_user_value", "by_i16_i32", "by_minute_service_full",
                                   "distinct_user_by_hour_service"])
def test_partial_then_merge_partitioned(part_ctx, oracle, rand_tables, pname):
    from prism_amd.engine import Plan

    _, q, n_keys = next(p for p in PLANS if p[0] == pname)
    t, g = rand_tables["mid"]
    plan = Plan(part_ctx, t.schema, q.with_mode(abi.MODE_PARTIAL))
    try:
        states = [plan.execute(g, [p]) for p in range(g.num_partitions)]
        got = plan.merge(states)
    finally:
        plan.close()
    assert_tables_match(got, oracle.execute(t, q), n_keys)


def _synth_both(ctx, oracle, wl, rows, parts, queries, hint=None):
    from prism_amd.engine import Plan

    table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, parts)
    st = oracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, rows, rows, n_parts=parts)
    try:
        for q, n_keys in queries:
            if hint is not None:
                q = q.with_groups_hint(hint) if hasattr(q, "with_groups_hint") else q
            plan = Plan(ctx, wl.schema, q)
            try:
                runs = [plan.execute(table), plan.execute(table)]
            finally:
                plan.close()
            want, _ = st.execute(q)
            want = workloads.decode_dictionary_columns(wl, want)
            for got in runs:
                if q.limit is not None:
                    assert_topk_match(got, want, q.order_by)
                else:
                    assert_tables_match(got, want, n_keys)
    finally:
        st.close()
        table.close()


def test_heavy_hitters_are_aggregated_in_slices(part_ctx, oracle):
    """Log-uniform ids: the most frequent id alone is 1/16 of the rows, far more than one
    work item holds, so its bucket is cut into slices that meet in the HBM table."""
    wl = workloads.c3(id_bits=16)
    _synth_both(part_ctx, oracle, wl, 3_000_000, 5, [
        (wl.query, 1),
        (QueryPlan([count_star()], keys=["UserID"]), 1),
    ])


def test_lds_tables_overflow_into_the_hbm_table(part_ctx, oracle):
    """A hint of very few groups gives 64 buckets; with ~10^6 groups every bucket holds
    more groups than an LDS table takes, and the rest goes through the HBM table (which
    has to grow) and a result list that was sized far too short."""
    wl = workloads.c3(id_bits=21)
    q = QueryPlan([count_star()], keys=["UserID"], groups_hint=100)
    _synth_both(part_ctx, oracle, wl, 4_000_000, 4, [(q, 1)])


def test_values_and_nulls_travel_with_the_records(part_ctx, oracle):
    import numpy as np
    import pyarrow as pa
    from prism_amd.engine import Plan, Table

    rng = np.random.default_rng(11)
    n = 400_000
    t = pa.table({
        "k": pa.array(rng.integers(0, 150_000, n).astype(np.int64) * 1_000_003 - 7, mask=rng.random(n) < 0.01),
        "a": pa.array(rng.integers(-1 << 40, 1 << 40, n).astype(np.int64), mask=rng.random(n) < 0.2),
        "b": pa.array(rng.integers(-30000, 30000, n).astype(np.int16), mask=rng.random(n) < 0.5),
        "c": pa.array(rng.integers(0, 1 << 31, n).astype(np.int32)),
    })
    g = Table(part_ctx, t.schema)
    g.append_table(t, 3)
    try:
        for q in [QueryPlan([count_star(), sum_("a"), min_("b"), max_("b"), avg("a"), count("b")], keys=["k"]),
                  QueryPlan([sum_("c"), max_("a")], keys=["k"], predicate=col("b") > 0),
                  QueryPlan([count_star(), avg("c")], keys=["k", "b"])]:
            plan = Plan(part_ctx, t.schema, q)
            try:
                got = plan.execute(g)
            finally:
                plan.close()
            assert_tables_match(got, oracle.execute(t, q), len(q.keys))
    finally:
        g.close()


def test_two_word_keys_and_count_distinct(part_ctx, oracle):
    """Two-word records: the scatter pass with two workgroups of 512 threads per CU (the
    default where the shape allows it), then with one of 1024."""
    wl = workloads.c5(n_services=16, user_bits=14)
    plain = QueryPlan([count_star()], keys=["user", "service"])  # two-word keys, no COUNT(DISTINCT)
    try:
        for halves in (1, 0):
            part_ctx.set_option("part_half_blocks", halves)
            _synth_both(part_ctx, oracle, wl, 1_500_000, 4, [(wl.query, 2), (plain, 2)])
    finally:
        part_ctx.set_option("part_half_blocks", 1)


def test_count_distinct_partitioned_twice(part_ctx, oracle):
    """COUNT(DISTINCT) over two-word pairs: coarse buckets by group, fine buckets by pair, LDS
    sets, per-group counts straight into the outer table (no set in HBM).  C5's shape, and a
    table whose 62-bit values have NULLs (a NULL is a pair of its own but is not counted)."""
    import numpy as np
    import pyarrow as pa
    from prism_amd.engine import Plan, Table

    wl = workloads.c5(n_services=16, user_bits=14)
    # records of one word at the second level (the bits their fine bucket does not imply:
    # 63-bit ids without NULLs next to a 12-bit group key allow it), then of two words
    rows = 1_500_000
    # (and both ways of placing the fine buckets: regions of fixed size, the default, and exact
    # places from a histogram pass over the records)
    part_ctx.set_option("part2_direct", 0)  # (the two levels: what rows in no particular order take)
    for compact, record_bytes, fixed in ((1, 8, 1), (0, 16, 1), (1, 8, 0), (0, 16, 0)):
        part_ctx.set_option("part2_compact", compact)
        part_ctx.set_option("part2_fixed", fixed)
        part_ctx.reset_stats()
        _synth_both(part_ctx, oracle, wl, rows, 4, [(wl.query, 2)])
        ran = _kernels(part_ctx)
        assert {"part2_layout" if fixed else "part2_hist", "part2_scatter", "part2_agg"} <= ran
        assert ("part2_hist" in ran) == (not fixed)
        agg = part_ctx.stats()["kernels"]["part2_agg"]
        assert agg["algorithmic_bytes"] == agg["launches"] * rows * record_bytes
    part_ctx.set_option("part2_compact", 1)
    part_ctx.set_option("part2_fixed", 1)
    # The generated rows are in time order and the group key starts with the hour: sorted by the
    # group key's top bits already, so there is no first level -- the one partition pass reads the columns
    # (its fine buckets: regions of one size plus a spill list sorted afterwards, or exact places from a histogram)
    part_ctx.set_option("part2_direct", 1)
    for fixed, halves in ((1, 0), (0, 0), (1, 1)):
        part_ctx.set_option("part2_fixed", fixed)
        part_ctx.set_option("part2_half_blocks", halves)
        part_ctx.reset_stats()
        _synth_both(part_ctx, oracle, wl, rows, 4, [(wl.query, 2)])
        ran = _kernels(part_ctx)
        assert {"part2_scatter", "part2_agg"} <= ran and not ({"part_scatter", "part_hist"} & ran), ran
        assert ({"part2_layout", "part2_spill"} <= ran and "part2_hist" not in ran) if fixed else \
               ("part2_hist" in ran and not ({"part2_layout", "part2_spill"} & ran)), ran
    part_ctx.set_option("part2_fixed", 1)
    part_ctx.set_option("part2_half_blocks", 0)

    rng = np.random.default_rng(21)
    n = 300_000
    a = rng.integers(0, 5000, n).astype(np.int64) * (2**62 // 5000) - 2**61
    t = pa.table({"a": pa.array(a, mask=rng.random(n) < 0.05),
                  "b": pa.array(rng.integers(0, 200, n).astype(np.int64) * (2**40 // 200), mask=rng.random(n) < 0.01),
                  "c": pa.array(rng.integers(0, 9, n).astype(np.int32))})
    g = Table(part_ctx, t.schema)
    g.append_table(t, 3)
    try:
        for q in [QueryPlan([count_distinct("a")], keys=["c", "b"]),
                  QueryPlan([count_distinct("a")], keys=["c"], predicate=col("c") > 2),
                  QueryPlan([count_distinct("a")], keys=["b"], order_by=1, descending=True, limit=5)]:
            plan = Plan(part_ctx, t.schema, q)
            try:
                got = plan.execute(g)
                again = plan.execute(g)
            finally:
                plan.close()
            want = oracle.execute(t, q)
            if q.limit is not None:
                assert_topk_match(got, want, q.order_by)
                assert_topk_match(again, want, q.order_by)
            else:
                assert_tables_match(got, want, len(q.keys))
                assert_tables_match(again, want, len(q.keys))
    finally:
        g.close()


def test_count_distinct_over_rows_out_of_time_order(part_ctx, oracle):
    """The same query over rows in no order at all: the pass that takes the rows as sorted by
    the hour still counts right (rows outside its LDS counters go through global atomics), the
    plan notices, and its next run sorts by the group key first."""
    import numpy as np
    import pyarrow as pa
    from prism_amd.engine import Plan, Table

    rng = np.random.default_rng(9)
    n = 600_000
    t = pa.table({"ts": pa.array(1_700_000_000_000 + rng.integers(0, 40 * 3_600_000, n), pa.timestamp("ms")),
                  "svc": pa.array(rng.integers(0, 12, n).astype(np.int32)),
                  "user": pa.array(rng.integers(0, 30_000, n).astype(np.int64) * (2**62 // 30_000) - 2**61)})  # (62-bit ids: the pair takes two words)
    q = QueryPlan([count_distinct("user")], keys=[date_bin(3_600_000, "ts"), "svc"])
    want = oracle.execute(t, q)
    g = Table(part_ctx, t.schema)
    g.append_table(t, 3)
    try:
        plan = Plan(part_ctx, t.schema, q)
        try:
            part_ctx.reset_stats()
            first = plan.execute(g)
            ran_first = _kernels(part_ctx)
            part_ctx.reset_stats()
            second = plan.execute(g)
            ran_second = _kernels(part_ctx)
        finally:
            plan.close()
        assert_tables_match(first, want, 2)
        assert_tables_match(second, want, 2)
        assert "part2_scatter" in ran_first and "part_scatter" not in ran_first, ran_first
        assert "part_scatter" in ran_second, ran_second
        # rows in time order within every partition: no first level, both times
        ts = t.column("ts").cast(pa.int64()).to_numpy()
        order = np.argsort(ts, kind="stable")
        sorted_t = t.take(pa.array(order))
        gs = Table(part_ctx, t.schema)
        gs.append_table(sorted_t, 3)
        try:
            plan = Plan(part_ctx, t.schema, q)
            try:
                for _ in range(2):
                    part_ctx.reset_stats()
                    assert_tables_match(plan.execute(gs), want, 2)
                    assert "part_scatter" not in _kernels(part_ctx)
            finally:
                plan.close()
        finally:
            gs.close()
    finally:
        g.close()


def test_count_distinct_scatter_by_staging_lines(part_ctx, oracle):
    """Rows in time order whose hours run through hundreds of scan tiles each: the one-level
    scatter takes a stretch of rows of one hour per workgroup (the hours' first tiles found by
    bisection) and copies full staging lines into the piece's part of the fine buckets' regions.
    Checked against the oracle and against the tile-by-tile scatter: in order; with some rows far
    from their place (they take the spill list); with a crawler's id in a third of the rows
    (dropped as repeats of a recently staged record); and with nearly all rows in one hour (the
    pieces would be too long: exact places from a histogram instead)."""
    import numpy as np
    import pyarrow as pa
    from prism_amd.engine import Plan, Table

    rng = np.random.default_rng(31)
    n, hours = 2_400_000, 5
    q = QueryPlan([count_distinct("user")], keys=[date_bin(3_600_000, "ts"), "svc"])

    def table(ts, heavy=0.0):
        user = rng.integers(0, 400_000, n).astype(np.int64) * (2**62 // 400_000) - 2**61  # (62-bit ids: the pair takes two words)
        if heavy:
            user[rng.random(n) < heavy] = 12345 * (2**62 // 400_000) - 2**61
        return pa.table({"ts": pa.array(1_700_000_000_000 + ts, pa.timestamp("ms")),
                         "svc": pa.array(rng.integers(0, 9, n).astype(np.int32)), "user": pa.array(user)})

    even = np.sort(rng.integers(0, hours * 3_600_000, n))
    astray = even.copy()
    far = rng.random(n) < 0.03
    astray[far] = rng.integers(0, hours * 3_600_000, int(far.sum()))
    uneven = np.sort(np.where(rng.random(n) < 0.97, rng.integers(0, 3_600_000, n), rng.integers(0, hours * 3_600_000, n)))
    for name, t, lines_expected in [("in order", table(even), True), ("astray", table(astray), True),
                                    ("crawler", table(even, heavy=0.35), True), ("one long hour", table(uneven), None)]:
        want = oracle.execute(t, q)
        g = Table(part_ctx, t.schema)
        g.append_table(t, 4)
        try:
            for lines in (2, 0):  # (2: also for tables of a few hundred tiles)
                part_ctx.set_option("part2_lines", lines)
                plan = Plan(part_ctx, t.schema, q)
                try:
                    part_ctx.reset_stats()
                    got = [plan.execute(g), plan.execute(g)]
                    ran = _kernels(part_ctx)
                finally:
                    plan.close()
                for r in got:
                    assert_tables_match(r, want, 2)
                assert "part2_scatter" in ran and "part_scatter" not in ran, (name, lines, ran)
                if lines_expected is not None:
                    assert ("part2_pieces" in ran) == (lines == 2), (name, lines, ran)
            if name == "in order":
                # a WHERE (the pieces read the selection bitmap) and a group key of the hour alone (two key columns)
                part_ctx.set_option("part2_lines", 2)
                for q2 in (QueryPlan([count_distinct("user")], keys=[date_bin(3_600_000, "ts"), "svc"], predicate=col("svc") != 3),
                           QueryPlan([count_distinct("user")], keys=[date_bin(3_600_000, "ts")])):
                    plan = Plan(part_ctx, t.schema, q2)
                    try:
                        part_ctx.reset_stats()
                        got2 = plan.execute(g)
                        ran = _kernels(part_ctx)
                    finally:
                        plan.close()
                    assert_tables_match(got2, oracle.execute(t, q2), len(q2.keys))
                    assert "part2_pieces" in ran, ran
        finally:
            part_ctx.set_option("part2_lines", 1)
            g.close()


def test_count_distinct_of_one_dominant_group(part_ctx, oracle):
    """Nine rows in ten belong to one group.  With values that hardly repeat its fine buckets
    hold far more distinct pairs than the LDS set takes: the attempt to take such a bucket in
    one go is given up and it goes class by class.  With values that repeat a lot the same
    buckets are just as large but fit in one go."""
    import numpy as np
    import pyarrow as pa
    from prism_amd.engine import Plan, Table

    rng = np.random.default_rng(33)
    n = 2_000_000
    c = np.where(rng.random(n) < 0.9, 17, rng.integers(0, 200, n)).astype(np.int32)
    for distinct_values in (1 << 40, 300):
        a = rng.integers(0, distinct_values, n).astype(np.int64) * ((2**62) // distinct_values) - 2**61
        t = pa.table({"a": pa.array(a), "c": pa.array(c)})
        g = Table(part_ctx, t.schema)
        g.append_table(t, 4)
        try:
            q = QueryPlan([count_distinct("a")], keys=["c"])
            plan = Plan(part_ctx, t.schema, q)
            try:
                part_ctx.reset_stats()
                got = plan.execute(g)
                agg = part_ctx.stats()["kernels"]["part2_agg"]
                # one-word records.  (300 values that repeat a lot make very uneven fine buckets: the
                # regions of fixed size overflow, and the level runs again with its histogram)
                assert agg["launches"] == (2 if distinct_values == 300 else 1)
                assert agg["algorithmic_bytes"] == agg["launches"] * n * 8
                assert ("part2_hist" in _kernels(part_ctx)) == (distinct_values == 300)
                # (the plan has learnt what share of the rows are distinct pairs: with few of
                # them the second run makes its fine buckets up to four times as large)
                again = plan.execute(g)
            finally:
                plan.close()
            want = oracle.execute(t, q)
            assert_tables_match(got, want, 1)
            assert_tables_match(again, want, 1)
        finally:
            g.close()


def test_forced_partitioning_of_nothing_and_states_for_an_exchange(part_ctx, oracle):
    """Partitioning forced at any size: a scan that selects no rows is not partitioned (the
    passes size their work by the rows), and a state kept for an exchange between GPUs stays a
    directly indexed table whenever the agreed key space allows one, whatever this rank's rows
    and plan would choose -- ranks that held different kinds of table would enter different
    collectives."""
    import torch
    from prism_amd.engine import Plan, Table

    t = helpers.random_table(5000, seed=77, nulls=False)
    g = Table(part_ctx, t.schema)
    g.append_table(t, 3)
    try:
        q = QueryPlan([count_star(), sum_("value")], keys=["i16"])
        plan = Plan(part_ctx, t.schema, q)
        try:
            assert plan.execute(g, [17]).num_rows == 0  # no such partition: nothing scanned
            stats = plan.scan_stats(g)
            states = [plan.aggregate(g, [p], stats) for p in (0, 1, 2, 17)]
            infos = [s.describe() for s in states]
            assert all(i.dense for i in infos) and len({i.entries for i in infos}) == 1
            # folded like a reduce between ranks would: slot by slot, in place
            e = infos[0].entries
            bufs = []
            for s_, info in zip(states, infos):
                b = torch.zeros(max(1, info.export_bytes // 8), dtype=torch.int64, device="cuda:0")
                torch.cuda.synchronize()
                s_.export_to(b.data_ptr(), info.export_bytes)
                bufs.append(b)
            acc = bufs[0]
            for b in bufs[1:]:
                for sl in range(infos[0].n_slots):
                    assert infos[0].slot_merge[sl] == abi.SLOT_SUM_I64
                    acc[sl * e:(sl + 1) * e] += b[sl * e:(sl + 1) * e]
            torch.cuda.synchronize()
            states[0].import_from(acc.data_ptr(), infos[0].export_bytes)
            got = states[0].finish()
            for s_ in states:
                s_.close()
        finally:
            plan.close()
        assert_tables_match(got, oracle.execute(t, q), 1)
    finally:
        g.close()


def test_streaming_kernel_hands_over_when_rows_scatter(oracle):
    """No forcing: the streaming kernel runs first, finds that its rows miss the LDS tables,
    stops, and the plan is partitioned from then on (its predicate, evaluated inside the
    streaming kernel on the first attempt, as a bitmap)."""
    import pyarrow.compute as pc
    from prism_amd.engine import Context, Plan

    ctx = Context(0)
    ctx.set_option("part_min_rows", 100_000)
    wl = workloads.c3(id_bits=22)
    rows, parts = 32_000_000, 8  # enough tiles per workgroup for the LDS tables to fill up a few times
    table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, parts)
    st = oracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, rows, rows, n_parts=parts, threads=8)

    def same(got, want):
        # millions of groups: compared as sorted columns rather than row by row in Python
        assert got.num_rows == want.num_rows
        for t in (got, want):
            assert t.schema.names == ["UserID", "COUNT(*)"]
        gi, wi = pc.sort_indices(got, [("UserID", "ascending")]), pc.sort_indices(want, [("UserID", "ascending")])
        assert got.take(gi).equals(want.take(wi))

    try:
        for q in [QueryPlan([count_star()], keys=["UserID"]),
                  QueryPlan([count_star()], keys=["UserID"], predicate=col("UserID") > (1 << 61))]:
            want, _ = st.execute(q, threads=8)
            plan = Plan(ctx, wl.schema, q)
            try:
                ctx.reset_stats()
                first = plan.execute(table)
                ran = _kernels(ctx)
                # (COUNT(*) alone over this many rows: partitioned without a histogram pass)
                assert "wc_agg" in ran, "the streaming kernel did not hand the plan over"
                ctx.reset_stats()
                second = plan.execute(table)
                assert "wc_agg" in _kernels(ctx) and "part_hist" not in _kernels(ctx)
            finally:
                plan.close()
            same(first, want)
            same(second, want)
    finally:
        st.close()
        table.close()
        ctx.close()


# ---- GROUP BY key -> COUNT(*) without a histogram pass (kernels_wc.hip) ----------------------

@pytest.fixture(scope="module")
def wc_ctx():
    from prism_amd.engine import Context

    ctx = Context(0)
    ctx.set_option("part_min_rows", 0)
    ctx.set_option("part_force", 1)
    ctx.set_option("part_wc_min_rows", 0)
    yield ctx
    ctx.close()


def _same_groups(got, want, names):
    import pyarrow.compute as pc

    assert got.num_rows == want.num_rows
    order = [(n, "ascending") for n in names]
    assert got.take(pc.sort_indices(got, order)).equals(want.take(pc.sort_indices(want, order)))


def test_histogram_free_count_by_key_small_tables(wc_ctx, oracle):
    """Every shape the path takes, on tables of a few tiles: a nullable key (keys loaded with
    their validity bits), two key columns in one word, a WHERE (selection bitmap), ORDER BY ..
    LIMIT over the list, a one-word COUNT(DISTINCT) whose pairs are counted the same way."""
    from prism_amd.engine import Plan, Table

    for n, nulls, parts in [(1000, True, 1), (50_000, True, 3), (20_000, False, 2)]:
        t = helpers.random_table(n, seed=n + 5, nulls=nulls)
        g = Table(wc_ctx, t.schema)
        g.append_table(t, parts)
        try:
            for q, n_keys in [(QueryPlan([count_star()], keys=["user"]), 1),
                              (QueryPlan([count_star()], keys=["i16", "i32"]), 2),
                              (QueryPlan([count_star()], keys=["user"], predicate=col("i32") > 0), 1),
                              (QueryPlan([count_star()], keys=["user"], order_by=1, descending=True, limit=7), 1),
                              (QueryPlan([count_star()], keys=[date_bin(60_000, "ts"), "service"]), 2),
                              (QueryPlan([count_distinct("i16")], keys=["service"]), 1)]:
                wc_ctx.reset_stats()
                plan = Plan(wc_ctx, t.schema, q)
                try:
                    got = plan.execute(g)
                    again = plan.execute(g)
                finally:
                    plan.close()
                assert "wc_scatter" in _kernels(wc_ctx) and "part_hist" not in _kernels(wc_ctx), q
                want = oracle.execute(t, q)
                for r in (got, again):
                    if q.limit is not None:
                        assert_topk_match(r, want, q.order_by)
                    else:
                        assert_tables_match(r, want, n_keys)
        finally:
            g.close()


def test_order_by_count_limit_lists_only_the_groups_that_can_lead(wc_ctx, oracle):
    """ORDER BY COUNT(*) DESC LIMIT n on the histogram-free path: the n-th largest total among the
    keys counted in LDS is known before the aggregation pass, and only groups that reach it are
    listed (no selection pass over millions of groups afterwards).  Same answer as with the
    option off and as the oracle's, for limits the hot keys cover and for one they do not."""
    from prism_amd.engine import Plan

    wl = workloads.c3(id_bits=20)
    rows, parts = 6_000_000, 6
    table = workloads.build_gpu_table(wc_ctx, wl, 0, rows, rows, parts)
    st = oracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, rows, rows, n_parts=parts, threads=8)
    try:
        full, _ = st.execute(QueryPlan([count_star()], keys=["UserID"]), threads=8)
        count_of = dict(zip(full.column(0).to_pylist(), full.column(1).to_pylist()))
        for limit in (1, 10, 100, 1024):
            q = QueryPlan([count_star()], keys=["UserID"], order_by=1, descending=True, limit=limit)
            want, _ = st.execute(q, threads=8)
            for leading in (1, 0):
                wc_ctx.set_option("part_wc_leading", leading)
                wc_ctx.reset_stats()
                plan = Plan(wc_ctx, wl.schema, q)
                try:
                    runs = [plan.execute(table), plan.execute(table)]
                finally:
                    plan.close()
                ran = _kernels(wc_ctx)
                assert "wc_agg" in ran and "part_hist" not in ran
                if limit <= 10:
                    assert ("topk_select" in ran) == (leading == 0), (limit, leading, sorted(ran))
                for got in runs:
                    assert got.num_rows == limit
                    assert_topk_match(got, want, q.order_by)
                    ids = got.column(0).to_pylist()
                    assert len(set(ids)) == limit and [count_of[i] for i in ids] == got.column(1).to_pylist()
        # ascending order, and an order by the key: every group is listed as before
        for q in [QueryPlan([count_star()], keys=["UserID"], order_by=1, descending=False, limit=5),
                  QueryPlan([count_star()], keys=["UserID"], order_by=0, descending=True, limit=5)]:
            want, _ = st.execute(q, threads=8)
            plan = Plan(wc_ctx, wl.schema, q)
            try:
                got = plan.execute(table)
            finally:
                plan.close()
            assert_topk_match(got, want, q.order_by)
    finally:
        wc_ctx.set_option("part_wc_leading", 1)
        st.close()
        table.close()


def test_profiling_levels_time_composite_launches_as_a_whole(wc_ctx):
    """prism_gpu_set_profiling: level 1 brackets every pass of a partitioned GROUP BY with events,
    level 2 only the launch as a whole ("group_by"); the passes are counted either way."""
    from prism_amd.engine import Plan

    wl = workloads.c3(id_bits=18)
    rows = 1_000_000
    table = workloads.build_gpu_table(wc_ctx, wl, 0, rows, rows, 2)
    plan = Plan(wc_ctx, wl.schema, QueryPlan([count_star()], keys=["UserID"]))
    try:
        plan.execute(table)
        for level, passes_timed in ((1, True), (2, False), (0, False)):
            wc_ctx.reset_stats()
            wc_ctx.set_profiling(level)
            plan.execute(table)
            wc_ctx.set_profiling(0)
            k = wc_ctx.stats()["kernels"]
            assert k["wc_scatter"]["launches"] == 1 and k["group_by"]["launches"] == 1
            assert (k["group_by"]["total_ms"] > 0) == (level != 0)
            assert (k["wc_scatter"]["total_ms"] > 0) == passes_timed and (k["wc_agg"]["total_ms"] > 0) == passes_timed
    finally:
        wc_ctx.set_profiling(0)
        plan.close()
        table.close()


def test_histogram_free_path_hot_keys_and_many_groups(wc_ctx, oracle):
    """Log-uniform ids: the frequent ids are found in the sample and counted in LDS (they would
    overflow any segment: one id alone holds 1/20 of the rows), the rest goes through the
    staging lines; 2e6 groups in 6e6 rows."""
    wl = workloads.c3(id_bits=20)
    rows, parts = 6_000_000, 6
    table = workloads.build_gpu_table(wc_ctx, wl, 0, rows, rows, parts)
    st = oracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, rows, rows, n_parts=parts, threads=8)
    try:
        from prism_amd.engine import Plan

        for q in [QueryPlan([count_star()], keys=["UserID"]), wl.query,
                  QueryPlan([count_star()], keys=["UserID"], predicate=col("UserID") > (1 << 61))]:
            want, _ = st.execute(q, threads=8)
            wc_ctx.reset_stats()
            plan = Plan(wc_ctx, wl.schema, q)
            try:
                runs = [plan.execute(table), plan.execute(table)]
            finally:
                plan.close()
            ran = _kernels(wc_ctx)
            assert "wc_agg" in ran and "part_hist" not in ran and "part_agg" not in ran
            for got in runs:
                if q.limit is not None:
                    assert_topk_match(got, want, q.order_by)
                else:
                    _same_groups(got, want, ["UserID"])
    finally:
        st.close()
        table.close()


def test_histogram_free_path_falls_back_when_a_segment_overflows(wc_ctx, oracle):
    """Rows sorted by key: every key's rows lie in one workgroup's share, so with few keys
    per bucket some (workgroup, bucket) segment takes far more rows than it holds.  The run
    is given up, the histogram path produces the result, and the plan remembers."""
    import numpy as np
    import pyarrow as pa
    from prism_amd.engine import Plan, Table

    n = 3_000_000
    # 500 rows a key, sorted: six times more keys than the hot table takes, each a few segments' worth
    keys = np.repeat(np.arange(6000, dtype=np.int64) * 7_777_777_777 + 13, n // 6000)
    t = pa.table({"k": pa.array(keys)})
    g = Table(wc_ctx, t.schema)
    g.append_table(t, 3)
    try:
        q = QueryPlan([count_star()], keys=["k"])
        plan = Plan(wc_ctx, t.schema, q)
        try:
            wc_ctx.reset_stats()
            got = plan.execute(g)
            ran = _kernels(wc_ctx)
            assert "wc_scatter" in ran and "part_agg" in ran  # tried, given up, done the other way
            wc_ctx.reset_stats()
            again = plan.execute(g)
            assert "wc_scatter" not in _kernels(wc_ctx)
        finally:
            plan.close()
        want = oracle.execute(t, q)
        assert_tables_match(got, want, 1)
        assert_tables_match(again, want, 1)
    finally:
        g.close()


def test_a_merge_is_sized_by_what_merges_learnt_not_by_a_partition(part_ctx, oracle):
    """One plan aggregates the partitions (PARTIAL) and then merges their states, the way the
    query engine uses it: a partition sees a fraction of the groups the merge sees.  The merge's
    list of groups is sized by what the previous merge found, so from the second round on it is
    one partitioned pass (sized by a partition's groups, its list was too short every time and
    the pass ran twice)."""
    from prism_amd.engine import Plan

    wl = workloads.c3(id_bits=20)
    rows, parts = 4_000_000, 8
    q = QueryPlan([count_star()], keys=["UserID"])
    table = workloads.build_gpu_table(part_ctx, wl, 0, rows, rows, parts)
    st = oracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, rows, rows, n_parts=parts)
    plan = Plan(part_ctx, wl.schema, q.with_mode(abi.MODE_PARTIAL))
    try:
        want, _ = st.execute(q)
        for round_ in range(3):
            states = [plan.execute(table, [p]) for p in range(parts)]
            assert max(s.num_rows for s in states) * 2 < want.num_rows  # (a partition's groups say little about the merge's)
            part_ctx.reset_stats()
            got = plan.merge(states)
            launches = part_ctx.stats()["kernels"]["group_by"]["launches"]
            assert_tables_match(got, want, 1)
            if round_ > 0:
                assert launches == 1, launches
    finally:
        plan.close()
        st.close()
        table.close()
