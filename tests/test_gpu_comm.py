"""The exchange inside the library (prism_gpu_comm_*, prism_gpu_execute_sharded): a one-rank
RCCL communicator on the real GPU, two and three ranks that share the GPU over the callbacks
transport (RCCL refuses two ranks on one device; the exchange code above the transport is the
same), and the error paths.  Results equal the CPU oracle's over all ranks' rows: bit-exact
for keys, counts and integer aggregates, 1e-9 relative for AVG."""
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

    t = helpers.random_table(60_000, seed=12, nulls=True)
    g = Table(gpu_ctx, t.schema)
    g.append_table(t, 4)
    yield t, g
    g.close()


def test_one_rank_rccl_communicator(gpu_ctx, loaded):
    """ncclCommInitRank with world 1: librccl is found and loaded, and both shapes of state
    (dense, hash) finish through the sharded entry point."""
    from oracle import pyoracle
    from prism_amd.engine import Comm, Plan

    t, g = loaded
    uid = Comm.unique_id()
    assert len(uid) == Comm.ID_BYTES
    comm = Comm.rccl(gpu_ctx, uid, 0, 1)
    try:
        for keys, n_keys in [(["service", "i16"], 2), (["user"], 1)]:
            q = QueryPlan(AGGS, keys=keys, predicate=col("i32") > -900)
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                got = plan.execute_sharded(comm, g)
            finally:
                plan.close()
            assert_tables_match(got, pyoracle.execute(t, q), n_keys)
        assert comm.stats()["collectives"] >= 2
    finally:
        comm.close()


def test_plans_without_device_state_are_refused_before_any_collective(gpu_ctx, loaded):
    from prism_amd.engine import Comm, Plan, PrismGpuError

    t, g = loaded
    comm = Comm.rccl(gpu_ctx, Comm.unique_id(), 0, 1)
    try:
        plan = Plan(gpu_ctx, t.schema, QueryPlan([count_star(), sum_("value")]))  # no GROUP BY
        try:
            with pytest.raises(PrismGpuError) as e:
                plan.execute_sharded(comm, g)
            assert e.value.code == abi.EUNSUPPORTED
        finally:
            plan.close()
        assert comm.stats()["collectives"] == 0
    finally:
        comm.close()


def test_count_distinct_and_wide_keys_through_the_sharded_entry_point(gpu_ctx, loaded):
    """One rank: the pair set is folded where it is; keys wider than 64 bits keep both words."""
    from oracle import pyoracle
    from prism_amd.engine import Comm, Plan

    t, g = loaded
    comm = Comm.rccl(gpu_ctx, Comm.unique_id(), 0, 1)
    try:
        for q, n_keys in [(QueryPlan([count_distinct("user")], keys=["service"]), 1),
                          (QueryPlan([count_distinct("i16")], keys=["service"]), 1),
                          (QueryPlan([count_distinct("value")], keys=[date_bin(3_600_000, "ts"), "service"]), 2),
                          (QueryPlan([count_star(), max_("i32")], keys=["user", "value", "ts"]), 3)]:
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                got = plan.execute_sharded(comm, g)
            finally:
                plan.close()
            assert_tables_match(got, pyoracle.execute(t, q), n_keys)
    finally:
        comm.close()


def test_a_failing_transport_is_reported_as_erccl(gpu_ctx, loaded):
    import ctypes as C

    from prism_amd.engine import Comm, Plan, PrismGpuError

    t, g = loaded
    tr = abi.Transport()
    keep = (abi.ALLGATHER_FN(lambda *_: 7), abi.REDUCE_FN(lambda *_: 7), abi.ALLTOALLV_FN(lambda *_: 7))
    tr.allgather, tr.reduce, tr.alltoallv = keep
    h = C.c_void_p()
    rc = gpu_ctx._lib.prism_gpu_comm_init_callbacks(gpu_ctx._h, C.byref(tr), 0, 2, C.byref(h))
    assert rc == 0
    comm = Comm(gpu_ctx, h, keep)
    plan = Plan(gpu_ctx, t.schema, QueryPlan([count_star()], keys=["service"]))
    try:
        with pytest.raises(PrismGpuError) as e:
            plan.execute_sharded(comm, g)
        assert e.value.code == abi.ERCCL and "allgather" in e.value.msg
    finally:
        plan.close()
        comm.close()
    # rank outside the world
    rc = gpu_ctx._lib.prism_gpu_comm_init_callbacks(gpu_ctx._h, C.byref(tr), 2, 2, C.byref(h))
    assert rc == abi.EINVAL


def _rank_worker(rank: int, world: int, port: int, q):
    """One of `world` processes sharing GPU 0: its own context, its own share of the tables,
    the exchange over the callbacks transport; the root checks against the oracle."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    from oracle import pyoracle
    from prism_amd import workloads
    from prism_amd.distributed import exchange_and_finish, make_comm
    from prism_amd.engine import Context, Plan
    from prism_amd.plan import (QueryPlan as Q, avg as avg_, col as col_, count_distinct as cd, count_star as cs, date_bin as db,
                                max_ as mx, min_ as mn, sum_ as sm)
    from tests.helpers import assert_tables_match, assert_topk_match

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ctx = Context(0)
    comm = make_comm(ctx)
    try:
        c3 = workloads.c3(id_bits=16)
        c2 = workloads.c2()
        c5 = workloads.c5(n_services=16, user_bits=14)
        cases = [
            (c2, c2.query, 400_000, 2, 0),                                                        # dense: in-place reduce
            (c3, c3.query, 300_000, 1, 0),                                                        # hash, ORDER BY .. LIMIT
            (c3, Q([cs(), mx("UserID"), avg_("UserID"), mn("UserID"), sm("UserID")], keys=["UserID"]), 300_000, 1, world - 1),
            (c3, Q([cs()], keys=["UserID"], order_by=0, descending=False, limit=5), 200_000, 1, 0),  # ordered by the key
            # COUNT(DISTINCT): pairs dealt to the owner of their group, counted there.  One-word pairs
            # in a dense set (reduced in place), in a hash set; two-word pairs (a 63-bit id): C5's shape
            (c2, Q([cd("value")], keys=["service"]), 200_000, 1, 0),
            (c2, Q([cd("value")], keys=[db(60_000, "timestamp"), "service"]), 200_000, 2, world - 1),
            (c5, c5.query, 300_000, 2, 0),
            (c5, Q([cs()], keys=["user", "service"]), 200_000, 2, 0),                              # plain two-word keys
        ]
        for wl, query, rows, n_keys, root in cases:
            total = rows * world
            table = workloads.build_gpu_table(ctx, wl, rows * rank, rows, total, 4)
            plan = Plan(ctx, wl.schema, query)
            try:
                # the last rank of a three-rank world scans nothing in the first case
                parts = [] if (world == 3 and rank == 2 and query is c2.query) else None
                got = exchange_and_finish(plan, table, parts, comm, root)
            finally:
                plan.close()
                table.close()
            if rank == root:
                n_have = total if not (world == 3 and query is c2.query) else rows * 2
                st = pyoracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, n_have, total, n_parts=4)
                try:
                    want, _ = st.execute(query)
                finally:
                    st.close()
                want = workloads.decode_dictionary_columns(wl, want)
                if query.limit is not None:
                    assert_topk_match(got, want, query.order_by)
                else:
                    assert_tables_match(got, want, n_keys)
            else:
                assert got is None
        assert comm.stats()["bytes_sent"] > 0
        # COUNT(DISTINCT) the way large scans take it (forced here on small ones): the pairs are
        # partitioned twice on every rank, the fine buckets' DISTINCT records go to the buckets'
        # owners, the owners count.  No set of pairs in HBM: k_group_by_wide does not run.
        ctx.set_option("part_force", 1)
        for wl, query, rows, n_keys, root, sharded in [
            (c5, c5.query, 1_500_000, 2, 0, True),
            (c5, Q([cd("user")], keys=[db(3_600_000, "timestamp"), "service"], predicate=col_("user") > (1 << 61)), 1_500_000, 2, world - 1, True),
            # a group key of four bits is narrower than the coarse buckets' number: no one-word records, and
            # one-word pairs are not partitioned twice at all -- the other way, on every rank alike
            (c5, Q([cd("user")], keys=["service"]), 200_000, 1, world - 1, False),
            (c2, Q([cd("value")], keys=["service"]), 100_000, 1, 0, False),
        ]:
            total = rows * world
            table = workloads.build_gpu_table(ctx, wl, rows * rank, rows, total, 4)
            plan = Plan(ctx, wl.schema, query)
            ctx.reset_stats()
            try:
                # the last rank of a three-rank world scans nothing in the first case
                parts = [] if (world == 3 and rank == 2 and query is c5.query) else None
                got = exchange_and_finish(plan, table, parts, comm, root)
            finally:
                plan.close()
                table.close()
            ran = {k for k, v in ctx.stats()["kernels"].items() if v["launches"]}
            assert ("part2_merge" in ran) == sharded, (query, ran)
            if rank == root:
                n_have = total if parts is None and not (world == 3 and query is c5.query) else rows * 2
                st = pyoracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, n_have, total, n_parts=4)
                try:
                    want, _ = st.execute(query)
                finally:
                    st.close()
                assert_tables_match(got, workloads.decode_dictionary_columns(wl, want), n_keys)
            else:
                assert got is None
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        comm.close()
        ctx.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_share_the_gpu_and_exchange_inside_the_library(world):
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert results == {r: "ok" for r in range(world)}, results


def _failing_rank_worker(rank: int, world: int, port: int, q):
    """Two ranks sharing GPU 0.  (a) COUNT(DISTINCT) over a string column whose dictionaries
    differ between the ranks; (b) one rank fails on purpose in each local phase of the sharded
    execution.  Every rank must come back with an error: nobody waits in a collective."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import pyarrow as pa
    import torch.distributed as dist

    from prism_amd import _cabi as abi_
    from prism_amd import workloads
    from prism_amd.distributed import exchange_and_finish, make_comm
    from prism_amd.engine import Context, Plan, PrismGpuError, Table
    from prism_amd.plan import QueryPlan as Q, count_distinct as cd, count_star as cs
    from tests.helpers import assert_tables_match

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ctx = Context(0)
    comm = make_comm(ctx)
    log = []
    try:
        # (a) the same strings, appended in a different order on each rank: different codes
        names = [f"user-{i:03d}" for i in range(40)]
        order = names if rank == 0 else names[::-1]
        t = pa.table({"g": pa.array([i % 5 for i in range(4000)], pa.int64()),
                      "who": pa.array([order[(i * 7 + i // 40) % 40] for i in range(4000)], pa.string())})
        g = Table(ctx, t.schema)
        g.append_table(t, 2)
        plan = Plan(ctx, t.schema, Q([cd("who")], keys=["g"]))
        try:
            exchange_and_finish(plan, g, None, comm, 0)
            log.append("no error for differing dictionaries")
        except PrismGpuError as e:
            if e.code != abi_.EINVAL or "dictionar" not in e.msg:
                log.append(f"differing dictionaries: {e.code} {e.msg}")
        finally:
            plan.close()
        # the same dictionary everywhere: the counts are right (40 names in every group)
        g2 = Table(ctx, t.schema)
        g2.set_dictionary("who", names)
        g2.append_table(t, 2)
        plan = Plan(ctx, t.schema, Q([cd("who")], keys=["g"]))
        try:
            got = exchange_and_finish(plan, g2, None, comm, 0)
        finally:
            plan.close()
        if rank == 0:
            both = pa.concat_tables([pa.table({"g": t.column("g"), "who": pa.array([names[(i * 7 + i // 40) % 40] for i in range(4000)])}),
                                     pa.table({"g": t.column("g"), "who": pa.array([names[::-1][(i * 7 + i // 40) % 40] for i in range(4000)])})])
            from oracle import pyoracle

            assert_tables_match(got, pyoracle.execute(both, Q([cd("who")], keys=["g"])), 1)
        g.close()
        g2.close()

        # (b) rank 1 fails in phase k: both ranks get an error, then the next query works again
        c3 = workloads.c3(id_bits=16)
        c2 = workloads.c2()
        rows = 100_000
        for wl, phases in ((c2, (1, 2)), (c3, (1, 2, 3, 4, 5))):
            table = workloads.build_gpu_table(ctx, wl, rows * rank, rows, rows * world, 2)
            plan = Plan(ctx, wl.schema, Q([cs()], keys=wl.query.keys))
            try:
                for phase in phases:
                    if rank == 1:
                        ctx.set_option("sharded_fail_phase", phase)
                    try:
                        exchange_and_finish(plan, table, None, comm, 0)
                        log.append(f"{wl.name} phase {phase}: no error on rank {rank}")
                    except PrismGpuError as e:
                        if rank == 1 and "injected" not in e.msg:
                            log.append(f"{wl.name} phase {phase}: rank 1 got {e.msg}")
                        if rank == 0 and "rank 1" not in e.msg:
                            log.append(f"{wl.name} phase {phase}: rank 0 got {e.msg}")
                    ctx.set_option("sharded_fail_phase", 0)
                    got = exchange_and_finish(plan, table, None, comm, 0)  # the communicator is still in step
                    if rank == 0 and sum(got.column("COUNT(*)").to_pylist()) != rows * world:
                        log.append(f"{wl.name} after phase {phase}: wrong counts")
            finally:
                plan.close()
                table.close()
        q.put((rank, "ok" if not log else "; ".join(log)))
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        comm.close()
        ctx.close()
        dist.destroy_process_group()


def test_a_failure_on_one_rank_is_an_error_on_every_rank():
    import socket

    import torch.multiprocessing as mp

    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert results == {r: "ok" for r in range(world)}, results


def _distinct_at_scale_worker(rank: int, world: int, port: int, q):
    """COUNT(DISTINCT) over 2.5e8 rows on two ranks that share GPU 0 (1.25e8 each): the sharded
    result on the root equals the result of one context over all the rows, which the other GPU
    tests check against the oracle; the set of pairs is never built in HBM."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    from prism_amd import workloads
    from prism_amd.distributed import exchange_and_finish, make_comm
    from prism_amd.engine import Context, Plan
    from tests.helpers import assert_tables_match

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ctx = Context(0)
    comm = make_comm(ctx)
    try:
        wl = workloads.c5()
        rows = 125_000_000
        total = rows * world
        table = workloads.build_gpu_table(ctx, wl, rows * rank, rows, total, 8)
        plan = Plan(ctx, wl.schema, wl.query)
        ctx.reset_stats()
        try:
            got = exchange_and_finish(plan, table, None, comm, 0)
        finally:
            table.close()
        ran = {k for k, v in ctx.stats()["kernels"].items() if v["launches"]}
        assert "part2_merge" in ran and "shard_gather" in ran, ran
        # (the rank's own part: rows in time order, an hour runs through hundreds of scan tiles -- one level, staging lines)
        assert "part2_pieces" in ran and "part2_hist" not in ran and "part_scatter" not in ran, ran
        assert comm.stats()["bytes_sent"] > rows  # the distinct records of the other rank's buckets: gigabytes
        if rank == 0:
            whole = workloads.build_gpu_table(ctx, wl, 0, total, total, 16)
            try:
                want = plan.execute(whole)
            finally:
                whole.close()
            assert got.num_rows == want.num_rows
            assert_tables_match(got, want, 2)
        plan.close()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        comm.close()
        ctx.close()
        dist.destroy_process_group()


def test_sharded_count_distinct_at_scale_exchanges_distinct_records():
    import socket

    import torch.multiprocessing as mp

    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_distinct_at_scale_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=400) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert results == {r: "ok" for r in range(world)}, results
