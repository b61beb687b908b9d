"""COUNT(DISTINCT x) next to other aggregates: the library runs the two halves over the same
rows and joins them on the group keys.  The expectation is built the same way from two oracle
runs (the oracle, like DataFusion's SingleDistinctToGroupBy rewrite, handles plans whose
aggregates are all COUNT(DISTINCT) or none).  In PARTIAL mode -- the mode the reference's
physical plan runs an aggregate's first stage in, per partition
(src/prism-query/src/providers/table_provider.rs:73-132 hands DataFusion one file group per
partition) -- the state of such a COUNT(DISTINCT) is a List of the group's distinct values
next to the ordinary state columns, and the state of a string MIN / MAX the string so far."""
import pyarrow as pa
import pytest

from tests import helpers
from tests.helpers import assert_tables_match

pytestmark = pytest.mark.gpu

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd.plan import QueryPlan, avg, col, count_distinct, count_star, date_bin, max_, sum_  # noqa: E402


def _kinds(aggs):
    kinds = {}
    for i, a in enumerate(aggs):
        kinds.setdefault(a.column if a.func == abi.AGG_COUNT_DISTINCT else None, []).append(i)
    return kinds


def _expected(oracle, t, keys, aggs, predicate, run=None):
    """One oracle run per kind of aggregate (the plain ones; the COUNT(DISTINCT)s of each
    column), aligned on the group keys.  `run(kind, plan)`: where a kind's result comes from
    (default: the oracle over `t`)."""
    kinds = _kinds(aggs)
    n_keys = len(keys)
    per_kind = {}
    for kind, idxs in kinds.items():
        q = QueryPlan([aggs[i] for i in idxs], keys=keys, predicate=predicate)
        res = run(kind, q) if run else oracle.execute(t, q)
        rows = {}
        for r in range(res.num_rows):
            key = tuple(res.column(c)[r].as_py() for c in range(n_keys))
            rows[key] = [res.column(n_keys + j)[r].as_py() for j in range(len(idxs))]
        per_kind[kind] = (res, rows)
    base, _ = next(iter(per_kind.values()))
    out_rows = []
    for r in range(base.num_rows):
        key = tuple(base.column(c)[r].as_py() for c in range(n_keys))
        vals = [None] * len(aggs)
        for kind, idxs in kinds.items():
            for j, i in enumerate(idxs):
                vals[i] = per_kind[kind][1][key][j]
        out_rows.append(list(key) + vals)
    cols = list(zip(*out_rows)) if out_rows else [[] for _ in range(n_keys + len(aggs))]
    arrays = [pa.array(list(cols[c]), type=base.column(c).type) for c in range(n_keys)]
    for i, a in enumerate(aggs):
        kind = a.column if a.func == abi.AGG_COUNT_DISTINCT else None
        src = per_kind[kind][0]
        arrays.append(pa.array(list(cols[n_keys + i]), type=src.column(n_keys + kinds[kind].index(i)).type))
    return pa.table(arrays, names=[f"c{c}" for c in range(len(arrays))])


@pytest.mark.parametrize("keys,aggs,pred", [
    (["service"], [count_star(), count_distinct("user"), sum_("value")], None),
    (["service", "i16"], [count_distinct("user"), max_("u16"), avg("value")], "filter"),
    ([], [count_distinct("i16"), count_star(), max_("value")], "filter"),
    ([date_bin(3_600_000, "ts")], [sum_("i32"), count_distinct("service")], None),
    (["service"], [count_distinct("user"), count_distinct("i16")], None),                       # two distinct columns
    (["i16"], [count_distinct("service"), count_star(), count_distinct("user"), count_distinct("service")], "filter"),
    ([], [count_distinct("user"), count_distinct("value")], None),
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


def _kind_states(oracle, t, states, keys, aggs, kind, q):
    """The PARTIAL batches of a plan of several kinds, cut down to the PARTIAL batches of one
    kind in the oracle's own layout: the plain aggregates' state columns as they are; a List
    column as one (group, value) row per item (and a NULL-value row per group, which a distinct
    count ignores)."""
    import pyarrow.compute as pc

    n_keys = len(keys)
    at, col_of = n_keys, []
    for a in aggs:
        col_of.append(at)
        at += 2 if a.func == abi.AGG_AVG else 1
    want = oracle.output_schema(t.schema, q.with_mode(abi.MODE_PARTIAL))
    out = []
    for s in states:
        s = s.combine_chunks()
        if kind is None:
            cols = list(range(n_keys))
            for i, a in enumerate(aggs):
                if a.func != abi.AGG_COUNT_DISTINCT:
                    cols += [col_of[i]] + ([col_of[i] + 1] if a.func == abi.AGG_AVG else [])
            out.append(pa.table([s.column(c) for c in cols], schema=want))
            continue
        i = next(i for i, a in enumerate(aggs) if a.func == abi.AGG_COUNT_DISTINCT and a.column == kind)
        lists = s.column(col_of[i]).chunk(0) if s.num_rows else pa.array([], type=s.schema.field(col_of[i]).type)
        assert pa.types.is_list(lists.type)
        parents = pc.list_parent_indices(lists)
        arrays = [pa.concat_arrays([s.column(c).chunk(0).take(parents), s.column(c).chunk(0)]) if s.num_rows else s.column(c).chunk(0)
                  for c in range(n_keys)]
        values = pc.list_flatten(lists)
        arrays.append(pa.concat_arrays([values, pa.nulls(s.num_rows, values.type)]) if s.num_rows else values)
        out.append(pa.table(arrays, schema=want))
    return out


@pytest.mark.parametrize("keys,aggs,pred", [
    (["service"], [count_star(), count_distinct("user"), sum_("value")], None),
    (["service", "i16"], [count_distinct("user"), max_("u16"), avg("value")], "filter"),
    ([], [count_distinct("i16"), count_star(), max_("value")], "filter"),
    ([date_bin(3_600_000, "ts")], [sum_("i32"), count_distinct("service")], None),
    (["service"], [count_distinct("user"), count_distinct("i16")], None),
    (["i16"], [count_distinct("service"), count_star(), count_distinct("user"), count_distinct("service")], "filter"),
    ([], [count_distinct("user"), count_distinct("value")], None),
])
def test_partial_then_merge_of_count_distinct_beside_other_aggregates(gpu_ctx, keys, aggs, pred):
    """PARTIAL per partition on the GPU; the states merged by the GPU (prism_gpu_merge) and,
    kind by kind, by the oracle's Final stage: both must equal the single-pass result."""
    from oracle import pyoracle
    from prism_amd.engine import Plan, Table

    t = helpers.random_table(40_000, seed=31, nulls=True)
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 3)
    predicate = (col("i32") > -500) if pred else None
    try:
        q = QueryPlan(aggs, keys=keys, predicate=predicate)
        plan = Plan(gpu_ctx, t.schema, q.with_mode(abi.MODE_PARTIAL))
        try:
            states = [plan.execute(g, [p]) for p in range(g.num_partitions)]
            got = plan.merge(states)
        finally:
            plan.close()
        for s in states:  # a List per COUNT(DISTINCT), of the column's type
            at = len(keys)
            for a in aggs:
                if a.func == abi.AGG_COUNT_DISTINCT:
                    assert s.schema.field(at).type == pa.list_(t.schema.field(a.column).type), s.schema
                at += 2 if a.func == abi.AGG_AVG else 1
        want = _expected(pyoracle, t, keys, aggs, predicate)
        assert_tables_match(got.rename_columns(want.schema.names), want, len(keys))
        via_oracle = _expected(pyoracle, t, keys, aggs, predicate,
                               run=lambda kind, qk: pyoracle.merge(t.schema, _kind_states(pyoracle, t, states, keys, aggs, kind, qk), qk))
        assert_tables_match(via_oracle, want, len(keys))
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
        assert want.num_rows > 3
        for r in got.rename_columns(["service", "n", "d"]).to_pylist():
            assert by_service[r["service"]] == r
        # the streaming form: every chunk becomes a PARTIAL batch (a List of distinct values per group), merged in finish
        gpu_ctx.set_option("stream_chunk_rows", 5000)
        try:
            for mode in (abi.MODE_SINGLE, abi.MODE_PARTIAL):
                plan = Plan(gpu_ctx, t.schema, QueryPlan(aggs, keys=["service"]).with_mode(mode))
                try:
                    streamed = plan.run_batches(0, t.to_batches(max_chunksize=2048))
                    if mode == abi.MODE_PARTIAL:
                        assert pa.types.is_list(streamed.schema.field(2).type)
                        streamed = plan.merge([streamed])
                finally:
                    plan.close()
                assert_tables_match(streamed.rename_columns(want.schema.names), want, 1)
        finally:
            gpu_ctx.set_option("stream_chunk_rows", 1 << 20)
    finally:
        g.close()


def _string_extremes(t, key_names, column, mask=None):
    """Byte-lexicographic MIN and MAX of `column` per group, NULLs ignored (the group exists even
    when all its values are NULL)."""
    out = {}
    keys = [t.column(k).to_pylist() for k in key_names]
    vals = t.column(column).to_pylist()
    keep = mask.to_pylist() if mask is not None else [True] * t.num_rows
    for r in range(t.num_rows):
        if not keep[r]:
            continue
        g = tuple(k[r] for k in keys)
        lo, hi = out.setdefault(g, (None, None))
        v = vals[r]
        if v is not None:
            b = v.encode() if isinstance(v, str) else v
            lo = v if lo is None or b < (lo.encode() if isinstance(lo, str) else lo) else lo
            hi = v if hi is None or b > (hi.encode() if isinstance(hi, str) else hi) else hi
        out[g] = (lo, hi)
    return out


def test_min_max_over_strings(gpu_ctx):
    """MIN / MAX over Utf8 are byte-lexicographic (DataFusion's min_max.rs compares the bytes):
    grouped, ungrouped, next to other aggregates, with NULLs and a filter, non-ASCII values."""
    import numpy as np
    import pyarrow.compute as pc

    from prism_amd.engine import Plan, PrismGpuError, Table
    from prism_amd.plan import min_

    rng = np.random.default_rng(5)
    n = 20_000
    words = ["zeta", "Zeta", "alpha", "alph", "", "ünï", "u", "~", "alpha0", " alpha"]
    t = pa.table({
        "k": pa.array(rng.integers(0, 40, n).astype(np.int64), mask=rng.random(n) < 0.05),
        "w": pa.array([words[int(i)] for i in rng.integers(0, len(words), n)], type=pa.string(), mask=rng.random(n) < 0.3),
        "path": pa.array([f"/p/{int(i):05d}" for i in rng.integers(0, 3000, n)], type=pa.string()),
        "v": pa.array(rng.integers(-100, 100, n).astype(np.int64)),
    })
    # one group whose strings are all NULL
    t = pa.concat_tables([t, pa.table({"k": pa.array([777] * 3, pa.int64()), "w": pa.array([None] * 3, pa.string()),
                                       "path": pa.array(["/x"] * 3), "v": pa.array([1, 2, 3], pa.int64())})])
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 3)
    try:
        q = QueryPlan([min_("w"), count_star(), max_("w"), max_("path"), sum_("v")], keys=["k"], predicate=col("v") > -50)
        plan = Plan(gpu_ctx, t.schema, q)
        try:
            got = plan.execute(g)
        finally:
            plan.close()
        assert got.schema.field(1).type == pa.string() and got.schema.field(4).type == pa.string()
        mask = pc.greater(t.column("v"), -50)
        ew, ep = _string_extremes(t, ["k"], "w", mask), _string_extremes(t, ["k"], "path", mask)
        rows = got.to_pylist()
        names = got.schema.names
        assert len(rows) == len(ew)
        for r in rows:
            key = (r[names[0]],)
            assert (r[names[1]], r[names[3]]) == ew[key], key
            assert r[names[4]] == ep[key][1]
        assert ew[(777,)] == (None, None)
        # ungrouped
        plan = Plan(gpu_ctx, t.schema, QueryPlan([max_("w"), min_("path")]))
        try:
            one = plan.execute(g).to_pylist()[0]
        finally:
            plan.close()
        assert list(one.values()) == [_string_extremes(t, [], "w")[()][1], _string_extremes(t, [], "path")[()][0]]
        # PARTIAL per partition: the state is the smallest / largest string so far (Utf8), merged by prism_gpu_merge
        plan = Plan(gpu_ctx, t.schema, q.with_mode(abi.MODE_PARTIAL))
        try:
            states = [plan.execute(g, [p]) for p in range(g.num_partitions)]
            merged = plan.merge(states)
        finally:
            plan.close()
        for s in states:
            assert [s.schema.field(c).type for c in (1, 3, 4)] == [pa.string()] * 3 and s.schema.names[1].endswith("[min]")
        assert sorted(merged.to_pylist(), key=lambda r: (r[names[0]] is None, r[names[0]])) == \
            sorted(rows, key=lambda r: (r[names[0]] is None, r[names[0]]))
        # SUM over strings stays refused
        with pytest.raises(PrismGpuError) as e:
            Plan(gpu_ctx, t.schema, QueryPlan([sum_("w")]))
        assert e.value.code == abi.EUNSUPPORTED
    finally:
        g.close()
