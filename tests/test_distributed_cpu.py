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
    """Runs the collectives of the device-resident exchange on CPU tensors (gloo)."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist

    from prism_amd import _cabi as abi
    from prism_amd.distributed import agree_on_stats, gather_rows, reduce_dense_state

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        # statistics: min of mins, max of maxes, sum of NULL counts; equal digests
        local = np.array([[10 * rank, 100 + rank, rank + 1], [2**62, -(2**62), 0]], dtype=np.int64)
        stats, same = agree_on_stats(local, [0xFEEDFACECAFEBEEF], dev)
        _, differ = agree_on_stats(local, [rank], dev)
        # dense state: slots SUM_I64, SUM_I64 (wrapping), MAX, MIN, SUM_F64
        entries = 5
        merge = [abi.SLOT_SUM_I64, abi.SLOT_SUM_I64, abi.SLOT_MAX_I64, abi.SLOT_MIN_I64, abi.SLOT_SUM_F64]
        buf = torch.zeros(len(merge) * entries, dtype=torch.int64)
        buf[0:5] = torch.arange(5) + rank
        buf[5:10] = torch.tensor([2**62, 2**62, 1, 2, 3]) if rank == 0 else torch.tensor([2**62, 2**62, 1, 2, 3])
        buf[10:15] = torch.tensor([rank * 7, -rank, 3, 3, -(2**63)])
        buf[15:20] = torch.tensor([rank * 7, -rank, 3, 3, 2**63 - 1])
        buf[20:25] = torch.tensor([0.5 + rank, -1.25, 0.0, 1e300, 3.0], dtype=torch.float64).view(torch.int64)
        reduce_dense_state(buf, merge, entries, dst=0)
        # hash rows: rank r sends r + 1 rows of (key, 2 slots)
        n = rank + 1
        rows = torch.cat([torch.arange(n) + 100 * rank, torch.full((n,), rank), torch.full((n,), -rank)]).to(torch.int64)
        parts = gather_rows(rows, n, 2, dst=0)
        # all-to-all by key hash: every key must arrive at exactly one rank, with its slots
        from prism_amd.distributed import exchange_rows_all_to_all

        keys = torch.arange(50, dtype=torch.int64) * 1_000_003 + rank  # 50 keys per rank, disjoint
        cols = torch.cat([keys, keys * 2, keys * 3])
        got = exchange_rows_all_to_all(cols, 50, 2)
        m = got.numel() // 3
        gk, g1, g2 = got[:m], got[m:2 * m], got[2 * m:]
        a2a_ok = bool((g1 == gk * 2).all() and (g2 == gk * 3).all())
        q.put((rank, stats.tolist(), same, differ, buf.tolist() if rank == 0 else None,
               [(p.tolist(), k) for p, k in parts] if parts is not None else None, gk.tolist(), a2a_ok))
    finally:
        dist.destroy_process_group()


def test_state_exchange_collectives_two_ranks():
    import struct

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
        stats, same, differ = out[r][0], out[r][1], out[r][2]
        assert stats == [[0, 101, 3], [2**62, -(2**62), 0]]
        assert same is True and differ is False
    buf = out[0][3]
    assert buf[0:5] == [1, 3, 5, 7, 9]
    assert buf[5:10] == [-(2**63), -(2**63), 2, 4, 6]          # 2^62 + 2^62 wraps
    assert buf[10:15] == [7, 0, 3, 3, -(2**63)]
    assert buf[15:20] == [0, -1, 3, 3, 2**63 - 1]
    f = [struct.unpack("<d", struct.pack("<q", v))[0] for v in buf[20:25]]
    assert f == [2.0, -2.5, 0.0, 2e300, 6.0]
    parts = out[0][4]
    assert out[1][4] is None
    assert parts[0] == ([0, 0, 0], 1)
    assert parts[1] == ([100, 101, 1, 1, -1, -1], 2)
    # all-to-all: the two ranks' slices are disjoint, cover all 100 keys, and rows stayed intact
    k0, k1 = out[0][5], out[1][5]
    assert out[0][6] and out[1][6]
    assert not set(k0) & set(k1)
    assert sorted(k0 + k1) == sorted([i * 1_000_003 + r for i in range(50) for r in (0, 1)])
    assert len(k0) > 10 and len(k1) > 10
