"""Multi-GPU plumbing: one process per GPU, partitions sharded across ranks, partial
aggregates exchanged with the library's RCCL communicator (torch.distributed only
launches the ranks and carries PARTIAL Arrow batches for the plans the device-resident
exchange does not cover; "gloo" on CPU for tests).

The reference already treats every ingest partition as an independent unit that
produces a Partial aggregate (one file group per partition,
src/prism-query/src/providers/table_provider.rs:94-109) and merges partials in a
Final stage; here the partitions of a table are dealt to the GPUs of a node, each
GPU runs scan -> filter -> partial aggregate over its share with no communication,
and the only exchange is the merge of the per-GPU group tables.

Two forms of that exchange:

* `exchange_and_finish`: the group tables stay in HBM and the library issues the
  collectives itself (prism_gpu_execute_sharded over a prism_gpu_comm: RCCL, or callbacks
  into this process group for tests).  The ranks first agree on the column statistics the
  keys are packed from (one tiny all-gather), so equal groups get equal packed keys
  everywhere.  A small key space then indexes a dense table and the merge is a reduce of
  the accumulator arrays onto the finishing rank (SUM / MIN / MAX per slot); otherwise the
  occupied (key, accumulators) rows are dealt to owner ranks by key hash, merged there, and
  the owners' (leading) groups meet on the finishing rank.
* `all_gather_tables` + `Plan.merge`: PARTIAL state batches in DataFusion's own
  layout, all-gathered as Arrow IPC bytes; works for every plan (COUNT(DISTINCT),
  ungrouped aggregates) and is what a DataFusion Final stage would consume.
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


# ---- device-resident exchange -------------------------------------------------------------

def torch_cpu_callbacks():
    """(allgather, reduce, alltoallv) of prism_gpu_transport over the default process group's
    CPU collectives (gloo).  Each takes the C arguments of the struct's member (pointers as
    integers or ctypes pointers) and returns 0, or 1 if the collective raised."""
    import ctypes as C
    import traceback

    import numpy as np
    import torch
    import torch.distributed as dist

    from . import _cabi as abi

    rank, world = dist.get_rank(), dist.get_world_size()

    def view(ptr, n_bytes, dtype=np.uint8):
        if n_bytes == 0:
            return np.empty(0, dtype=dtype)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n_bytes,)).view(dtype)

    def allgather(_user, mine, all_, n_bytes):
        try:
            t = torch.from_numpy(view(mine, n_bytes).copy())
            outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            view(all_, n_bytes * world)[:] = torch.cat(outs).numpy()
            return 0
        except Exception:  # noqa: BLE001 - reported to the library as a transport failure
            traceback.print_exc()
            return 1

    def reduce(_user, buf, n, dtype, op, root):
        try:
            a = view(buf, n * 8, np.float64 if dtype == abi.REDUCE_F64 else np.int64)
            t = torch.from_numpy(a.copy())
            dist.reduce(t, root, op={abi.REDUCE_SUM: dist.ReduceOp.SUM, abi.REDUCE_MIN: dist.ReduceOp.MIN,
                                     abi.REDUCE_MAX: dist.ReduceOp.MAX}[op])
            if rank == root:
                a[:] = t.numpy()
            return 0
        except Exception:  # noqa: BLE001
            traceback.print_exc()
            return 1

    def alltoallv(_user, send, send_counts, recv, recv_counts):
        try:
            sc = [int(send_counts[r]) for r in range(world)]
            rc = [int(recv_counts[r]) for r in range(world)]
            s = torch.from_numpy(view(send, sum(sc) * 8, np.int64).copy())
            out = torch.empty(sum(rc), dtype=torch.int64)
            dist.all_to_all_single(out, s, output_split_sizes=rc, input_split_sizes=sc)
            if sum(rc):
                view(recv, sum(rc) * 8, np.int64)[:] = out.numpy()
            return 0
        except Exception:  # noqa: BLE001
            traceback.print_exc()
            return 1

    return allgather, reduce, alltoallv


def make_comm(ctx):
    """The communicator of the default process group for prism_gpu_execute_sharded: RCCL when
    the group's backend is nccl (one GPU per rank), the callbacks transport over the group's
    CPU collectives otherwise (tests: several ranks sharing one GPU)."""
    import torch.distributed as dist

    from .engine import Comm

    if dist.get_backend() == "nccl":
        return Comm.from_process_group(ctx)
    return Comm.from_callbacks(ctx, dist.get_rank(), dist.get_world_size(), *torch_cpu_callbacks())


def exchange_and_finish(plan, table, partitions, comm, dst: int = 0):
    """One sharded GROUP BY query with the exchange done on device-resident group tables, by
    the library's own collectives (prism_gpu_execute_sharded): every rank aggregates its
    `partitions` of `table`; rank `dst` returns the final rows (pyarrow.Table), the others
    None.  Raises PrismGpuError(EUNSUPPORTED) on every rank for plans the device-resident
    state does not cover; use all_gather_tables + Plan.merge for those."""
    return plan.execute_sharded(comm, table, partitions, dst)
