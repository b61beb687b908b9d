"""The N > 1 path on CPU: two gloo ranks shard the partitions, each produces PARTIAL
state, the states are all-gathered and merged.  The CPU oracle stands in for the GPU
engine here (it is only the checker's arithmetic; the choreography under test is
prism_amd.distributed)."""
import os
import socket
import sys

import pyarrow as pa
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import pyoracle
    from prism_amd import _cabi as abi
    from prism_amd.distributed import sharded_query
    from prism_amd.plan import QueryPlan, avg, col, count_distinct, count_star, date_bin, max_, sum_
    from tests import helpers

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = helpers.random_table(6000, seed=77)
        n_parts = 5
        parts = [t.slice(len(t) * p // n_parts, len(t) * (p + 1) // n_parts - len(t) * p // n_parts) for p in range(n_parts)]
        sizes = [p.nbytes for p in parts]
        plans = [
            (QueryPlan([count_star(), sum_("value"), max_("value"), avg("i32")], keys=[date_bin(60_000, "ts"), "service"],
                       predicate=col("i16") != 0), 2),
            (QueryPlan([count_distinct("user")], keys=["service"]), 1),
            (QueryPlan([count_star()], keys=["user"], order_by=1, descending=True, limit=5), 1),
        ]
        for qi, (q, n_keys) in enumerate(plans):
            qp = q.with_mode(abi.MODE_PARTIAL)

            def run_partial(mine):
                if not mine:
                    return pyoracle.execute(t.slice(0, 0), qp)
                return pyoracle.execute(pa.concat_tables([parts[i] for i in mine]), qp)

            got = sharded_query(run_partial, lambda states: pyoracle.merge(t.schema, states, q), list(range(n_parts)), sizes,
                                torch.device("cpu"))
            want = pyoracle.execute(t, q)
            if q.limit is not None:
                helpers.assert_topk_match(got, want, q.order_by)
            else:
                helpers.assert_tables_match(got, want, n_keys)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_partial_merge(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def test_partition_assignment_balances_by_size():
    sys.path.insert(0, ROOT)
    from prism_amd.distributed import assign_partitions

    sizes = [100, 10, 10, 10, 90, 40, 40]
    a = assign_partitions(sizes, 3)
    assert sorted(i for lst in a for i in lst) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in lst) for lst in a]
    assert max(loads) - min(loads) <= 20
    assert assign_partitions([5], 4) == [[0], [], [], []]
    assert assign_partitions([], 2) == [[], []]


def _collective_worker(rank: int, world: int, port: int, q):
    """The callbacks transport the library's exchange runs over when RCCL cannot be used,
    driven directly (no GPU): what each callback must deliver, between two gloo ranks."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ctypes as C

    import numpy as np
    import torch.distributed as dist

    from prism_amd import _cabi as abi
    from prism_amd.distributed import torch_cpu_callbacks

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        allgather, reduce, alltoallv = torch_cpu_callbacks()
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        u64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))  # noqa: E731

        # all-gather: rank order
        mine = np.array([10 * rank, 100 + rank, rank + 1], dtype=np.int64)
        all_ = np.zeros(3 * world, dtype=np.int64)
        rc_gather = allgather(None, ptr(mine), ptr(all_), mine.nbytes)

        # reductions onto rank 0: wrapping int64 sums, min, max, float64 sums
        s = np.array([2**62, 2**62, rank, -rank, 3], dtype=np.int64)
        mx = np.array([rank * 7, -rank, 3, 3, -(2**63)], dtype=np.int64)
        mn = np.array([rank * 7, -rank, 3, 3, 2**63 - 1], dtype=np.int64)
        f = np.array([0.5 + rank, -1.25, 0.0, 1e300, 3.0], dtype=np.float64)
        rcs = [reduce(None, ptr(s), 5, abi.REDUCE_I64, abi.REDUCE_SUM, 0), reduce(None, ptr(mx), 5, abi.REDUCE_I64, abi.REDUCE_MAX, 0),
               reduce(None, ptr(mn), 5, abi.REDUCE_I64, abi.REDUCE_MIN, 0), reduce(None, ptr(f), 5, abi.REDUCE_F64, abi.REDUCE_SUM, 0)]

        # all-to-all with uneven runs: rank r sends (r + 1) words to rank 0 and 2 * (r + 1) to rank 1, none to itself's ... 
        send_counts = np.array([rank + 1, 2 * (rank + 1)], dtype=np.uint64)
        send = (np.arange(int(send_counts.sum()), dtype=np.uint64) + 1000 * rank)
        recv_counts = np.array([(s_ + 1) * (1 if rank == 0 else 2) for s_ in range(world)], dtype=np.uint64)
        recv = np.zeros(int(recv_counts.sum()), dtype=np.uint64)
        rc_a2a = alltoallv(None, u64(send), u64(send_counts), u64(recv), u64(recv_counts))
        # an empty exchange is legal
        zero = np.zeros(world, dtype=np.uint64)
        empty = np.zeros(1, dtype=np.uint64)
        rc_empty = alltoallv(None, u64(empty), u64(zero), u64(empty), u64(zero))
        q.put((rank, [rc_gather, *rcs, rc_a2a, rc_empty], all_.tolist(), s.tolist(), mx.tolist(), mn.tolist(), f.tolist(), recv.tolist()))
    finally:
        dist.destroy_process_group()


def test_callbacks_transport_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_collective_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {}
    for _ in procs:
        r = q.get(timeout=120)
        out[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert out[r][0] == [0] * 7
        assert out[r][1] == [0, 100, 1, 10, 101, 2]
    _, _, s, mx, mn, f, _ = out[0]
    assert s == [-(2**63), -(2**63), 1, -1, 6]          # 2^62 + 2^62 wraps
    assert mx == [7, 0, 3, 3, -(2**63)]
    assert mn == [0, -1, 3, 3, 2**63 - 1]
    assert f == [2.0, -2.5, 0.0, 2e300, 6.0]
    # rank 0 receives word 0 of rank 0 and words 0..1 of rank 1; rank 1 the rest, in rank order
    assert out[0][6] == [0, 1000, 1001]
    assert out[1][6] == [1, 2, 1002, 1003, 1004, 1005]


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it: the script starts two ranks as
    child processes (torch.distributed.run) before it touches any GPU and relays rank 0's
    line.  --check-launch stops after the rendezvous, which needs no GPU."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--check-launch"], capture_output=True,
                       text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    got = json.loads(lines[0])
    assert got == {"check_launch": True, "n_gpus_asked": 2, "ranks_met": 2, "rank_sum": 1}
