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
