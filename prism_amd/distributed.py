"""Multi-GPU plumbing: one process per GPU, partitions sharded across ranks, partial
aggregates exchanged with torch.distributed (backend "nccl" is RCCL over xGMI on
ROCm; "gloo" on CPU for tests).

The reference already treats every ingest partition as an independent unit that
produces a Partial aggregate (one file group per partition,
src/prism-query/src/providers/table_provider.rs:94-109) and merges partials in a
Final stage; here the partitions of a table are dealt to the GPUs of a node, each
GPU runs scan -> filter -> partial aggregate over its share with no communication,
and the only exchange is the merge of the per-GPU group tables.

Two forms of that exchange:

* `exchange_and_finish`: the group tables stay in HBM (prism_gpu_aggregate ..
  prism_gpu_state_finish).  The ranks first agree on the column statistics the keys
  are packed from (one tiny all-gather), so equal groups get equal packed keys
  everywhere.  A small key space then indexes a dense table and the merge is an RCCL
  reduce of the accumulator arrays onto the finishing rank (SUM / MIN / MAX per
  slot); otherwise the occupied (key, accumulators) rows are gathered and merged
  into the finishing rank's hash table.
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

def agree_on_stats(local_stats, digests: Sequence[int], device):
    """All ranks' column statistics ([n_columns, 3] int64: min, max, nulls) reduced to the
    statistics of the union, with ONE collective: an all-gather of the small vector, reduced
    locally.  `digests` are dictionary digests of the string key columns; they must be equal
    on every rank (string keys travel as dictionary codes).  Returns (stats, same_dicts)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    n = local_stats.shape[0]
    vec = np.concatenate([local_stats.reshape(-1), np.array(list(digests), dtype=np.uint64).view(np.int64)])
    mine = torch.from_numpy(vec.astype(np.int64)).to(device)
    all_ = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(all_, mine)
    m = torch.stack(all_).cpu().numpy()
    st = m[:, : n * 3].reshape(world, n, 3)
    out = np.stack([st[:, :, 0].min(axis=0), st[:, :, 1].max(axis=0), st[:, :, 2].sum(axis=0)], axis=1)
    dig = m[:, n * 3:]
    same = bool((dig == dig[0]).all())
    return out, same


def reduce_dense_state(buf, slot_merge: Sequence[int], entries: int, dst: int = 0) -> None:
    """Reduces a dense state buffer (int64 [n_slots * entries], slot-major) onto rank `dst`:
    runs of neighbouring slots with the same merge rule go out as one collective."""
    import torch
    import torch.distributed as dist

    from . import _cabi as abi

    ops = {abi.SLOT_SUM_I64: dist.ReduceOp.SUM, abi.SLOT_SUM_F64: dist.ReduceOp.SUM,
           abi.SLOT_MIN_I64: dist.ReduceOp.MIN, abi.SLOT_MAX_I64: dist.ReduceOp.MAX}
    s = 0
    n_slots = len(slot_merge)
    while s < n_slots:
        e = s + 1
        while e < n_slots and slot_merge[e] == slot_merge[s]:
            e += 1
        view = buf[s * entries: e * entries]
        if slot_merge[s] == abi.SLOT_SUM_F64:
            view = view.view(torch.float64)
        dist.reduce(view, dst, op=ops[slot_merge[s]])
        s = e


def gather_rows(buf, n_groups: int, n_slots: int, dst: int = 0):
    """Gathers every rank's exported hash-table rows (int64 [(1 + n_slots) * n_groups]) on
    rank `dst`.  Returns [(tensor, n_groups)] per rank there, None elsewhere."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    n = torch.tensor([n_groups], dtype=torch.int64, device=buf.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    counts = [int(x.item()) for x in sizes]
    cap = max(1, max(counts)) * (1 + n_slots)
    mine = torch.zeros(cap, dtype=torch.int64, device=buf.device)
    mine[: buf.numel()] = buf
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, parts, dst=dst)
    if rank != dst:
        return None
    return [(parts[r][: counts[r] * (1 + n_slots)].contiguous(), counts[r]) for r in range(world)]


def exchange_rows_all_to_all(buf, n_groups: int, n_slots: int):
    """Exported hash-table rows ([n] packed keys then [n_slots][n] accumulators, int64) are
    dealt to their owner ranks (hash(key) mod world) with one all-to-all; returns the rows this
    rank owns, in the same column-major layout."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    width = 1 + n_slots
    rows = buf.view(width, n_groups).t().contiguous() if n_groups else buf.new_empty((0, width))
    # multiplicative hash of the packed key; the HBM tables use another one, so a rank's slice
    # still spreads over its whole table
    h = (rows[:, 0] * -7046029254386353131) >> 40 if n_groups else rows[:, 0]
    owner = torch.remainder(h, world)
    order = torch.argsort(owner, stable=True)
    send = rows.index_select(0, order).contiguous()
    send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts)
    send_list = [int(x) for x in send_counts.tolist()]
    recv_list = [int(x) for x in recv_counts.tolist()]
    recv = send.new_empty((sum(recv_list), width))
    dist.all_to_all_single(recv, send, output_split_sizes=recv_list, input_split_sizes=send_list)
    return recv.t().contiguous().view(-1)


def _concat_sort_limit(parts, query):
    """Final rows of disjoint key slices -> one table, with the plan's ORDER BY / LIMIT applied
    (NULLS FIRST when descending, NULLS LAST when ascending)."""
    import pyarrow.compute as pc

    out = pa.concat_tables([p for p in parts if p is not None])
    if query.order_by is not None:
        name = out.schema.names[query.order_by]
        idx = pc.sort_indices(out, sort_keys=[(name, "descending" if query.descending else "ascending")],
                              null_placement="at_start" if query.descending else "at_end")
        out = out.take(idx)
    if query.limit is not None:
        out = out.slice(0, query.limit)
    return out.combine_chunks()


_AGREED_STATS: dict = {}


def exchange_and_finish(plan, table, partitions, device, dst: int = 0):
    """One sharded GROUP BY query with the exchange done on device-resident group tables.
    Every rank aggregates its `partitions` of `table`; rank `dst` returns the final rows
    (pyarrow.Table), the others None.  Raises PrismGpuError(EUNSUPPORTED) for plans the state
    interface does not cover; use all_gather_tables + Plan.merge for those.

    `device` is where the group tables live (a cuda device).  With the nccl backend (RCCL) the
    collectives run on those HBM buffers; with gloo (tests: several ranks sharing one GPU)
    the exported buffers are staged through host memory for the collective."""
    import torch
    import torch.distributed as dist

    from .engine import PrismGpuError

    rank = dist.get_rank()
    on_device = dist.get_backend() == "nccl"
    comm_device = device if on_device else torch.device("cpu")
    key_cols = [k.column if hasattr(k, "column") else k for k in plan.query.keys]
    digests = []
    for c in key_cols:
        if pa.types.is_string(plan.schema.field(c).type) or pa.types.is_binary(plan.schema.field(c).type):
            digests.append(table.dictionary_digest(c)[0])
    # The agreed statistics are metadata of (table contents, partitions): like a catalogue's
    # column statistics they are exchanged once and reused until the table changes (every rank
    # sees the same sequence of calls, so every rank takes the same branch here).
    cache_key = (id(plan), id(table), table.num_rows, None if partitions is None else tuple(partitions), tuple(digests))
    cached = _AGREED_STATS.get(cache_key)
    if cached is None:
        stats, same_dicts = agree_on_stats(plan.scan_stats(table, partitions), digests, comm_device)
        if not same_dicts:
            raise PrismGpuError(-2, "string key columns have different dictionaries on different ranks")
        if len(_AGREED_STATS) > 64:
            _AGREED_STATS.clear()
        _AGREED_STATS[cache_key] = stats
    else:
        stats = cached
    state = plan.aggregate(table, partitions, stats)
    try:
        info = state.describe()
        slot_merge = list(info.slot_merge[: info.n_slots])
        n_words = info.export_bytes // 8
        buf = torch.empty(max(1, n_words), dtype=torch.int64, device=device)
        state.export_to(buf.data_ptr(), info.export_bytes)
        comm = buf if on_device else buf.cpu()
        if info.dense:
            reduce_dense_state(comm[:n_words], slot_merge, info.entries, dst)
            if rank != dst:
                if on_device:
                    torch.cuda.current_stream().synchronize()
                return None
            if not on_device:
                buf.copy_(comm)
            torch.cuda.current_stream().synchronize()  # the library reads buf on its own stream
            state.import_from(buf.data_ptr(), info.export_bytes)
            return state.finish()
        if dist.get_world_size() == 1:
            return state.finish()
        # Large key space: hash-partitioned all-to-all.  Rank r becomes the owner of the keys
        # with hash(key) mod world == r, receives every other rank's rows for them, merges
        # them into an (initially empty) table of its own and finishes its slice; the slices
        # (already cut to LIMIT rows each) meet on rank `dst`.
        owned = plan.aggregate(table, [0xFFFFFFFF], stats)  # no such partition: an empty state, same key packing
        try:
            mine = exchange_rows_all_to_all(comm[:n_words], info.entries, info.n_slots)
            if mine.numel():
                rows = mine if on_device else mine.to(device)
                torch.cuda.current_stream().synchronize()
                owned.merge_from(rows.data_ptr(), rows.numel() // (1 + info.n_slots))
            part = owned.finish()
        finally:
            owned.close()
        parts = all_gather_tables(part, comm_device)
        if rank != dst:
            return None
        return _concat_sort_limit(parts, plan.query)
    finally:
        state.close()
