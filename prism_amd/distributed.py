"""Multi-GPU plumbing: one process per GPU, partitions sharded across ranks, partial
aggregates exchanged with torch.distributed (backend "nccl" is RCCL over xGMI on
ROCm; "gloo" on CPU for tests).

The reference already treats every ingest partition as an independent unit that
produces a Partial aggregate (one file group per partition,
src/prism-query/src/providers/table_provider.rs:94-109) and merges partials in a
Final stage; here the partitions of a table are dealt to the GPUs of a node, each
GPU runs scan -> filter -> partial aggregate over its share with no communication,
and the only exchange is the all-gather of the (small) PARTIAL state batches, after
which the Final merge runs.
"""
from __future__ import annotations

import io
from typing import List, Sequence

import pyarrow as pa


def assign_partitions(sizes: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-processing-time split of partitions over ranks by size (the
    `size` field of the catalogue's Partition, proto/common/prism/common/v1/partition.proto:5-9)."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    loads = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += sizes[i]
    for lst in out:
        lst.sort()
    return out


def _to_ipc(table: pa.Table) -> bytes:
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, table.schema) as w:
        for b in table.combine_chunks().to_batches():
            w.write_batch(b)
    return sink.getvalue()


def _from_ipc(buf: bytes) -> pa.Table:
    return pa.ipc.open_stream(io.BytesIO(buf)).read_all()


def all_gather_tables(table: pa.Table, device) -> List[pa.Table]:
    """All-gathers one Arrow table per rank (PARTIAL aggregate state).  The payload
    travels as a byte tensor on `device`: with the nccl backend that is an RCCL
    all-gather between HBM buffers."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    payload = _to_ipc(table)
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    lens = [int(s.item()) for s in sizes]
    cap = max(max(lens), 1)
    mine = torch.zeros(cap, dtype=torch.uint8, device=device)
    if payload:
        mine[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    bufs = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(bufs, mine)
    return [_from_ipc(bytes(b[:l].cpu().numpy().tobytes())) for b, l in zip(bufs, lens)]


def sharded_query(run_partial, merge_final, partitions: Sequence[int], sizes: Sequence[int], device):
    """Runs a query over `partitions` sharded across the ranks of the default
    process group.

    run_partial(my_partitions) -> pyarrow.Table   PARTIAL state of this rank's share
    merge_final(list_of_states) -> pyarrow.Table  AggregateExec(Final) over all shares
    Every rank returns the final table.
    """
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    mine = [partitions[i] for i in assign_partitions(sizes, world)[rank]]
    state = run_partial(mine)
    states = all_gather_tables(state, device)
    return merge_final(states)
