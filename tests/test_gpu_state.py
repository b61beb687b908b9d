"""Device-resident aggregate state (prism_gpu_aggregate .. prism_gpu_state_finish): the
exchange step of a multi-GPU query, exercised on one GPU by treating two sets of
partitions as two ranks.  The merged result must equal the CPU oracle's answer over all
rows: bit-exact for keys, counts and integer aggregates, 1e-9 relative for AVG."""
import numpy as np
import pytest

from tests import helpers
from tests.helpers import assert_tables_match

pytestmark = pytest.mark.gpu

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd.plan import QueryPlan, avg, col, count, count_distinct, count_star, date_bin, max_, min_, sum_  # noqa: E402

AGGS = [count_star(), sum_("value"), min_("i32"), max_("u16"), avg("value"), count("i32")]


@pytest.fixture(scope="module")
def loaded(gpu_ctx):
    from prism_amd.engine import Table

    t = helpers.random_table(60_000, seed=11, nulls=True)
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 4)
    yield t, g
    g.close()


def _merge_on_one_gpu(plan, g, shards, agree=True):
    """Aggregates each shard into its own state, merges all into the first the way the
    ranks of a node would, and returns the final table."""
    import torch

    dev = torch.device("cuda", 0)
    stats = plan.scan_stats(g) if agree else None
    states = [plan.aggregate(g, sh, stats) for sh in shards]
    try:
        infos = [s.describe() for s in states]
        bufs = []
        for s, info in zip(states, infos):
            b = torch.zeros(max(1, info.export_bytes // 8), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()  # the library copies on its own stream: no torch work may be pending on b
            s.export_to(b.data_ptr(), info.export_bytes)
            bufs.append(b)
        torch.cuda.synchronize()
        first = infos[0]
        if first.dense:
            assert all(i.dense and i.entries == first.entries for i in infos)
            acc = bufs[0].clone()
            for b in bufs[1:]:
                for sl in range(first.n_slots):
                    v, w = acc[sl * first.entries:(sl + 1) * first.entries], b[sl * first.entries:(sl + 1) * first.entries]
                    m = first.slot_merge[sl]
                    if m == abi.SLOT_SUM_I64:
                        v += w
                    elif m == abi.SLOT_SUM_F64:
                        v.view(torch.float64).add_(w.view(torch.float64))
                    elif m == abi.SLOT_MIN_I64:
                        torch.minimum(v, w, out=v)
                    else:
                        torch.maximum(v, w, out=v)
            torch.cuda.synchronize()
            states[0].import_from(acc.data_ptr(), first.export_bytes)
        else:
            for b, info in zip(bufs[1:], infos[1:]):
                states[0].merge_from(b.data_ptr(), info.entries)
        return states[0].finish(), first
    finally:
        for s in states:
            s.close()


@pytest.mark.parametrize("keys,n_keys", [(["service", "i16"], 2), ([date_bin(60_000, "ts"), "service"], 2),
                                         (["user"], 1), (["user", "i16"], 2)])
def test_two_shards_merge_like_two_ranks(gpu_ctx, loaded, keys, n_keys):
    from oracle import pyoracle
    from prism_amd.engine import Plan

    t, g = loaded
    q = QueryPlan(AGGS, keys=keys, predicate=col("i32") > -900)
    plan = Plan(gpu_ctx, t.schema, q)
    try:
        got, info = _merge_on_one_gpu(plan, g, [[0, 2], [1], [3]])
    finally:
        plan.close()
    assert_tables_match(got, pyoracle.execute(t, q), n_keys)
    assert bool(info.dense) == (keys[-1] != "user" and keys[0] != "user")


def test_hash_state_grows_when_merging_many_groups(gpu_ctx, loaded):
    """The receiving table starts with a handful of groups and takes in thousands."""
    from oracle import pyoracle
    from prism_amd.engine import Plan

    t, g = loaded
    q = QueryPlan([count_star(), max_("value")], keys=["user"])
    plan = Plan(gpu_ctx, t.schema, q)
    try:
        stats = plan.scan_stats(g)  # both states pack their keys from the same statistics
        small = plan.aggregate(g, [0xFFFFFFF0], stats)  # no such partition: an empty state
        big = plan.aggregate(g, None, stats)
        try:
            import torch

            info = big.describe()
            assert not info.dense and info.entries > 5000
            buf = torch.zeros(info.export_bytes // 8, dtype=torch.int64, device="cuda:0")
            torch.cuda.synchronize()
            big.export_to(buf.data_ptr(), info.export_bytes)
            small.merge_from(buf.data_ptr(), info.entries)
            small.merge_from(buf.data_ptr(), info.entries)  # twice: counts double, max stays
            got = small.finish()
        finally:
            small.close()
            big.close()
    finally:
        plan.close()
    want = pyoracle.execute(t, q)
    doubled = want.set_column(1, "COUNT(*)", [[2 * v for v in want.column(1).to_pylist()]])
    assert_tables_match(got, doubled.cast(want.schema), 1)


def test_merging_states_built_from_other_statistics_is_refused(gpu_ctx, loaded):
    """A dense table is indexed by the packed key: keys packed from other statistics must be
    rejected, not written out of bounds."""
    import torch

    from prism_amd.engine import Plan, PrismGpuError

    t, g = loaded
    plan = Plan(gpu_ctx, t.schema, QueryPlan([count_star()], keys=["i32"]))
    try:
        narrow = plan.aggregate(g, [0xFFFFFFF0])  # empty scan, own statistics: a one-entry table
        wide = plan.aggregate(g)
        try:
            info = wide.describe()
            assert info.dense and narrow.describe().entries == 1
            rows = torch.arange(2 * 50, dtype=torch.int64, device="cuda:0") + 5  # 50 (key, count) rows, keys >= 5
            torch.cuda.synchronize()
            with pytest.raises(PrismGpuError) as e:
                narrow.merge_from(rows.data_ptr(), 50)
            assert e.value.code == -1
        finally:
            narrow.close()
            wide.close()
    finally:
        plan.close()


def test_state_finish_applies_order_by_limit(gpu_ctx, loaded):
    from oracle import pyoracle
    from prism_amd.engine import Plan

    t, g = loaded
    q = QueryPlan([count_star()], keys=["user"], order_by=1, descending=True, limit=7)
    plan = Plan(gpu_ctx, t.schema, q)
    try:
        got, _ = _merge_on_one_gpu(plan, g, [[0, 1], [2, 3]])
    finally:
        plan.close()
    helpers.assert_topk_match(got, pyoracle.execute(t, q), 1)


def test_unsupported_plans_say_so(gpu_ctx, loaded):
    from oracle import pyoracle
    from prism_amd.engine import Plan, PrismGpuError

    t, g = loaded
    plan = Plan(gpu_ctx, t.schema, QueryPlan([count_star()]))  # no GROUP BY: no group table to keep
    try:
        with pytest.raises(PrismGpuError) as e:
            plan.aggregate(g)
        assert e.value.code == -2
    finally:
        plan.close()
    # COUNT(DISTINCT): the state (the set of pairs) can be finished, but has no export layout
    q = QueryPlan([count_distinct("user")], keys=["service"])
    plan = Plan(gpu_ctx, t.schema, q)
    try:
        st = plan.aggregate(g)
        try:
            with pytest.raises(PrismGpuError) as e:
                st.describe()
            assert e.value.code == -2
            assert_tables_match(st.finish(), pyoracle.execute(t, q), 1)
        finally:
            st.close()
    finally:
        plan.close()
