"""Python driver over the C ABI of libprism_gpu.so (include/prism_gpu.h).

This is the test/benchmark harness side of the boundary: it plays the role the
Rust `ExecutionPlan` shim plays inside prism-query, moving pyarrow RecordBatches
across the Arrow C Data Interface.  All compute happens in the HIP library; if
the library or a GPU is missing, every call fails loudly; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, List, Optional, Sequence

import pyarrow as pa

from . import _cabi as abi
from .plan import QueryPlan, lower

_HERE = os.path.dirname(os.path.abspath(__file__))
# PRISM_GPU_LIB points at another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("PRISM_GPU_LIB") or os.path.join(_HERE, "lib", "libprism_gpu.so")

# every symbol include/prism_gpu.h declares
EXPORTED_SYMBOLS = [
    "prism_gpu_init", "prism_gpu_ctx_free", "prism_gpu_last_error",
    "prism_gpu_table_create", "prism_gpu_table_append", "prism_gpu_table_num_rows",
    "prism_gpu_table_num_partitions", "prism_gpu_table_free", "prism_gpu_table_set_dictionary",
    "prism_gpu_table_synth",
    "prism_gpu_plan_create", "prism_gpu_plan_output_schema", "prism_gpu_plan_final_schema", "prism_gpu_plan_free",
    "prism_gpu_execute",
    "prism_gpu_partition_open", "prism_gpu_push_batch", "prism_gpu_finish", "prism_gpu_part_free",
    "prism_gpu_merge",
    "prism_gpu_scan_stats", "prism_gpu_aggregate", "prism_gpu_state_describe", "prism_gpu_state_export",
    "prism_gpu_state_import", "prism_gpu_state_merge", "prism_gpu_state_finish", "prism_gpu_state_free",
    "prism_gpu_table_dictionary_digest",
    "prism_gpu_set_profiling", "prism_gpu_get_stats", "prism_gpu_reset_stats",
    "prism_gpu_set_option", "prism_gpu_execute_noalloc", "prism_gpu_plan_time_range",
    "prism_gpu_comm_unique_id", "prism_gpu_comm_init", "prism_gpu_comm_init_callbacks", "prism_gpu_comm_free",
    "prism_gpu_execute_sharded", "prism_gpu_comm_get_stats",
    "prism_gpu_table_append_parquet", "prism_gpu_parquet_read",
]

_LIB: Optional[C.CDLL] = None


class PrismGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"prism_gpu error {code}: {msg}")
        self.code = code
        self.msg = msg


def load_library() -> C.CDLL:
    """Loads libprism_gpu.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `make -C prism_amd/csrc` (or __graft_entry__.build())")
    # PyTorch ships its own copy of the HIP runtime, and two copies cannot both drive the GPU
    # from one process: when torch is installed let its copy load first, so that this
    # library binds to the same one (torch.distributed carries the exchange between GPUs).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    P, V = C.POINTER, C.c_void_p
    L.prism_gpu_last_error.restype = C.c_char_p
    L.prism_gpu_init.argtypes = [C.c_int, P(V)]
    L.prism_gpu_ctx_free.argtypes = [V]
    L.prism_gpu_ctx_free.restype = None
    L.prism_gpu_table_create.argtypes = [V, P(abi.ArrowSchema), P(V)]
    L.prism_gpu_table_append.argtypes = [V, C.c_uint32, P(abi.ArrowArray)]
    L.prism_gpu_table_num_rows.argtypes = [V, P(C.c_uint64)]
    L.prism_gpu_table_num_partitions.argtypes = [V, P(C.c_uint32)]
    L.prism_gpu_table_free.argtypes = [V]
    L.prism_gpu_table_free.restype = None
    L.prism_gpu_table_set_dictionary.argtypes = [V, C.c_int32, P(C.c_char_p), P(C.c_uint64), C.c_uint32]
    L.prism_gpu_table_synth.argtypes = [V, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                        P(abi.SynthCol), C.c_uint32]
    L.prism_gpu_plan_create.argtypes = [V, P(abi.ArrowSchema), P(abi.PlanDesc), P(V)]
    L.prism_gpu_plan_output_schema.argtypes = [V, P(abi.ArrowSchema)]
    L.prism_gpu_plan_final_schema.argtypes = [V, P(abi.ArrowSchema)]
    L.prism_gpu_plan_free.argtypes = [V]
    L.prism_gpu_plan_free.restype = None
    L.prism_gpu_execute.argtypes = [V, V, P(C.c_uint32), C.c_size_t, P(abi.ArrowArray)]
    L.prism_gpu_partition_open.argtypes = [V, C.c_uint32, P(V)]
    L.prism_gpu_push_batch.argtypes = [V, P(abi.ArrowArray)]
    L.prism_gpu_finish.argtypes = [V, P(abi.ArrowArray)]
    L.prism_gpu_part_free.argtypes = [V]
    L.prism_gpu_part_free.restype = None
    L.prism_gpu_merge.argtypes = [V, P(P(abi.ArrowArray)), C.c_size_t, P(abi.ArrowArray)]
    L.prism_gpu_scan_stats.argtypes = [V, V, P(C.c_uint32), C.c_size_t, P(abi.ColStats)]
    L.prism_gpu_aggregate.argtypes = [V, V, P(C.c_uint32), C.c_size_t, P(abi.ColStats), P(V)]
    L.prism_gpu_state_describe.argtypes = [V, P(abi.StateInfo)]
    L.prism_gpu_state_export.argtypes = [V, V, C.c_size_t]
    L.prism_gpu_state_import.argtypes = [V, V, C.c_size_t]
    L.prism_gpu_state_merge.argtypes = [V, V, C.c_uint64]
    L.prism_gpu_state_finish.argtypes = [V, P(abi.ArrowArray)]
    L.prism_gpu_state_free.argtypes = [V]
    L.prism_gpu_state_free.restype = None
    L.prism_gpu_table_dictionary_digest.argtypes = [V, C.c_int32, P(C.c_uint64), P(C.c_uint32)]
    L.prism_gpu_set_profiling.argtypes = [V, C.c_int]
    L.prism_gpu_get_stats.argtypes = [V, P(abi.Stats)]
    L.prism_gpu_reset_stats.argtypes = [V]
    L.prism_gpu_set_option.argtypes = [V, C.c_char_p, C.c_int64]
    L.prism_gpu_plan_time_range.argtypes = [V, C.c_int32, P(C.c_int64), P(C.c_int64)]
    L.prism_gpu_execute_noalloc.argtypes = [V, V, P(C.c_uint32), C.c_size_t, V, V, C.c_uint64, P(C.c_uint64)]
    L.prism_gpu_table_append_parquet.argtypes = [V, C.c_uint32, C.c_char_p, C.c_size_t]
    L.prism_gpu_parquet_read.argtypes = [C.c_char_p, C.c_size_t, P(abi.ArrowSchema), P(abi.ArrowArray)]
    L.prism_gpu_comm_unique_id.argtypes = [V]
    L.prism_gpu_comm_init.argtypes = [V, V, C.c_int, C.c_int, P(V)]
    L.prism_gpu_comm_init_callbacks.argtypes = [V, P(abi.Transport), C.c_int, C.c_int, P(V)]
    L.prism_gpu_comm_free.argtypes = [V]
    L.prism_gpu_comm_free.restype = None
    L.prism_gpu_execute_sharded.argtypes = [V, V, V, P(C.c_uint32), C.c_size_t, C.c_int, P(abi.ArrowArray), P(C.c_int)]
    L.prism_gpu_comm_get_stats.argtypes = [V, P(abi.CommStats)]
    _LIB = L
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise PrismGpuError(rc, load_library().prism_gpu_last_error().decode("utf-8", "replace"))


class Context:
    """prism_gpu_ctx: one per process and GPU."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        _check(self._lib.prism_gpu_init(device, C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            self._lib.prism_gpu_ctx_free(self._h)
            self._h = C.c_void_p()

    def set_option(self, name: str, value: int) -> None:
        _check(self._lib.prism_gpu_set_option(self._h, name.encode(), int(value)))

    def set_profiling(self, on) -> None:
        """False / 0: off; True / 1: every timed launch; 2: composite launches as a whole only."""
        _check(self._lib.prism_gpu_set_profiling(self._h, int(on)))

    def reset_stats(self) -> None:
        _check(self._lib.prism_gpu_reset_stats(self._h))

    def stats(self) -> dict:
        s = abi.Stats()
        _check(self._lib.prism_gpu_get_stats(self._h, C.byref(s)))
        return {
            "rows_scanned": s.rows_scanned,
            "rows_selected": s.rows_selected,
            "groups_out": s.groups_out,
            "bytes_staged": s.bytes_staged,
            "staging_ms": s.staging_ms,
            "chunks_pruned": s.chunks_pruned,
            "kernels": {
                s.kernels[i].name.decode(): {
                    "launches": s.kernels[i].launches,
                    "total_ms": s.kernels[i].total_ms,
                    "algorithmic_bytes": s.kernels[i].algorithmic_bytes,
                }
                for i in range(s.n_kernels)
            },
        }


class Table:
    """prism_gpu_table: HBM-resident partitions of one table."""

    def __init__(self, ctx: Context, schema: pa.Schema):
        self.ctx = ctx
        self.schema = schema
        self._lib = ctx._lib
        self._h = C.c_void_p()
        cs = abi.export_schema(schema)
        try:
            _check(self._lib.prism_gpu_table_create(ctx._h, C.byref(cs), C.byref(self._h)))
        finally:
            abi.release_schema(cs)

    def append(self, partition_id: int, batch: pa.RecordBatch) -> None:
        arr = abi.export_batch(batch)  # moved into the library
        _check(self._lib.prism_gpu_table_append(self._h, partition_id, C.byref(arr)))

    def append_table(self, table: pa.Table, n_partitions: int = 1) -> None:
        """Splits a pyarrow Table into `n_partitions` contiguous partitions."""
        n = table.num_rows
        for p in range(n_partitions):
            lo, hi = n * p // n_partitions, n * (p + 1) // n_partitions
            for b in table.slice(lo, hi - lo).combine_chunks().to_batches():
                self.append(p, b)

    def append_parquet(self, partition_id: int, file_bytes: bytes) -> None:
        """prism_gpu_table_append_parquet: the library decodes the file (every row group a chunk)."""
        _check(self._lib.prism_gpu_table_append_parquet(self._h, partition_id, file_bytes, len(file_bytes)))

    def set_dictionary(self, column: str, values: Sequence[str]) -> None:
        enc = [v.encode("utf-8") for v in values]
        arr = (C.c_char_p * len(enc))(*enc)
        lens = (C.c_uint64 * len(enc))(*[len(v) for v in enc])
        _check(self._lib.prism_gpu_table_set_dictionary(self._h, self.schema.names.index(column), arr, lens, len(enc)))

    def synth(self, partition_id: int, seed: int, first_row: int, n_rows: int, total_rows: int,
              cols: Sequence[abi.SynthCol]) -> None:
        arr = (abi.SynthCol * len(cols))(*cols)
        _check(self._lib.prism_gpu_table_synth(self._h, partition_id, seed, first_row, n_rows, total_rows, arr,
                                               len(cols)))

    def dictionary_digest(self, column: str) -> tuple:
        """(digest, number of values) of a string column's dictionary."""
        d, n = C.c_uint64(), C.c_uint32()
        _check(self._lib.prism_gpu_table_dictionary_digest(self._h, self.schema.names.index(column), C.byref(d),
                                                           C.byref(n)))
        return d.value, n.value

    @property
    def num_rows(self) -> int:
        n = C.c_uint64()
        _check(self._lib.prism_gpu_table_num_rows(self._h, C.byref(n)))
        return n.value

    @property
    def num_partitions(self) -> int:
        n = C.c_uint32()
        _check(self._lib.prism_gpu_table_num_partitions(self._h, C.byref(n)))
        return n.value

    def close(self) -> None:
        if self._h:
            self._lib.prism_gpu_table_free(self._h)
            self._h = C.c_void_p()


class Plan:
    """prism_gpu_plan over an input schema."""

    def __init__(self, ctx: Optional[Context], schema: pa.Schema, query: QueryPlan):
        """`ctx` None: a plan that can be inspected (schemas, time_range) but not run."""
        self.ctx = ctx
        self.schema = schema
        self.query = query
        self._lib = ctx._lib if ctx is not None else load_library()
        self._lowered = lower(query, schema.names)
        self._h = C.c_void_p()
        cs = abi.export_schema(schema)
        try:
            _check(self._lib.prism_gpu_plan_create(ctx._h if ctx is not None else None, C.byref(cs), self._lowered.ptr(),
                                                   C.byref(self._h)))
        finally:
            abi.release_schema(cs)
        self.output_schema = self._schema(self._lib.prism_gpu_plan_output_schema)
        self.final_schema = self._schema(self._lib.prism_gpu_plan_final_schema)

    def _schema(self, fn) -> pa.Schema:
        s = abi.ArrowSchema()
        _check(fn(self._h, C.byref(s)))
        return abi.import_schema(s)

    def execute(self, table: Table, partitions: Optional[Iterable[int]] = None) -> pa.Table:
        out = abi.ArrowArray()
        if partitions is None:
            _check(self._lib.prism_gpu_execute(self._h, table._h, None, 0, C.byref(out)))
        else:
            ids = list(partitions)
            arr = (C.c_uint32 * max(1, len(ids)))(*ids)
            if not ids:
                arr[0] = 0xFFFFFFFF  # an id no partition has: scan nothing
            _check(self._lib.prism_gpu_execute(self._h, table._h, arr, max(1, len(ids)), C.byref(out)))
        return pa.Table.from_batches([abi.import_batch(out, self.output_schema)])

    def time_range(self, column: str) -> tuple:
        """(start, end) inclusive: values of `column` outside it cannot pass the WHERE."""
        lo, hi = C.c_int64(), C.c_int64()
        _check(self._lib.prism_gpu_plan_time_range(self._h, self.schema.names.index(column), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def execute_noalloc(self, table: Table, partitions, out_values_ptr: int, out_valid_ptr: int, out_stride: int) -> int:
        """prism_gpu_execute_noalloc: result columns into caller-provided (device) buffers;
        returns the number of groups."""
        arr, k = self._partition_args(partitions)
        n = C.c_uint64()
        _check(self._lib.prism_gpu_execute_noalloc(self._h, table._h, arr, k, C.c_void_p(out_values_ptr),
                                                   C.c_void_p(out_valid_ptr), out_stride, C.byref(n)))
        return n.value

    def run_batches(self, partition: int, batches: Iterable[pa.RecordBatch]) -> pa.Table:
        """Streaming form: partition_open / push_batch* / finish."""
        part = C.c_void_p()
        _check(self._lib.prism_gpu_partition_open(self._h, partition, C.byref(part)))
        try:
            for b in batches:
                arr = abi.export_batch(b)
                _check(self._lib.prism_gpu_push_batch(part, C.byref(arr)))
            out = abi.ArrowArray()
            _check(self._lib.prism_gpu_finish(part, C.byref(out)))
            return pa.Table.from_batches([abi.import_batch(out, self.output_schema)])
        finally:
            self._lib.prism_gpu_part_free(part)

    def merge(self, states: Sequence) -> pa.Table:
        """AggregateExec(Final) over PARTIAL outputs of this plan."""
        batches: List[pa.RecordBatch] = []
        for s in states:
            if isinstance(s, pa.Table):
                batches += s.combine_chunks().to_batches()
            else:
                batches.append(s)
        arrs = [abi.export_batch(b) for b in batches]
        ptrs = (C.POINTER(abi.ArrowArray) * max(1, len(arrs)))(*[C.pointer(a) for a in arrs])
        out = abi.ArrowArray()
        try:
            _check(self._lib.prism_gpu_merge(self._h, ptrs, len(arrs), C.byref(out)))
        finally:
            for a in arrs:
                abi.release_array(a)  # borrowed by the library, still ours
        return pa.Table.from_batches([abi.import_batch(out, self.final_schema)])

    @staticmethod
    def _partition_args(partitions: Optional[Iterable[int]]):
        if partitions is None:
            return None, 0
        ids = list(partitions)
        arr = (C.c_uint32 * max(1, len(ids)))(*ids)
        if not ids:
            arr[0] = 0xFFFFFFFF  # an id no partition has: scan nothing
        return arr, max(1, len(ids))

    def execute_sharded(self, comm: "Comm", table: Table, partitions: Optional[Iterable[int]] = None,
                        root: int = 0) -> Optional[pa.Table]:
        """prism_gpu_execute_sharded: every rank aggregates its `partitions`; the final rows
        arrive on rank `root` (None elsewhere).  The collectives are the library's own."""
        arr, k = self._partition_args(partitions)
        out = abi.ArrowArray()
        has = C.c_int(0)
        _check(self._lib.prism_gpu_execute_sharded(comm._h, self._h, table._h, arr, k, root, C.byref(out), C.byref(has)))
        if not has.value:
            return None
        return pa.Table.from_batches([abi.import_batch(out, self.final_schema)])

    def scan_stats(self, table: Table, partitions: Optional[Iterable[int]] = None):
        """Per input column (min, max, nulls) of the scan, as an int64 array [n_columns, 3]
        (nulls reinterpreted as int64)."""
        import numpy as np

        n = len(self.schema.names)
        out = (abi.ColStats * n)()
        arr, k = self._partition_args(partitions)
        _check(self._lib.prism_gpu_scan_stats(self._h, table._h, arr, k, out))
        return np.array([[c.min, c.max, c.nulls] for c in out], dtype=np.int64).reshape(n, 3)

    def aggregate(self, table: Table, partitions: Optional[Iterable[int]] = None, stats=None) -> "State":
        """scan -> filter -> GROUP BY; the group table stays in HBM behind the returned State.
        `stats`: [n_columns, 3] array of (min, max, nulls) agreed between the ranks."""
        cs = None
        if stats is not None:
            cs = (abi.ColStats * len(stats))(*[abi.ColStats(int(r[0]), int(r[1]), int(r[2])) for r in stats])
        arr, k = self._partition_args(partitions)
        h = C.c_void_p()
        _check(self._lib.prism_gpu_aggregate(self._h, table._h, arr, k, cs, C.byref(h)))
        return State(self, h)

    def close(self) -> None:
        if self._h:
            self._lib.prism_gpu_plan_free(self._h)
            self._h = C.c_void_p()


class State:
    """prism_gpu_state: a group table in HBM between Plan.aggregate and finish()."""

    def __init__(self, plan: Plan, handle):
        self.plan = plan
        self._lib = plan._lib
        self._h = handle

    def describe(self) -> abi.StateInfo:
        info = abi.StateInfo()
        _check(self._lib.prism_gpu_state_describe(self._h, C.byref(info)))
        return info

    def export_to(self, device_ptr: int, nbytes: int) -> None:
        _check(self._lib.prism_gpu_state_export(self._h, C.c_void_p(device_ptr), nbytes))

    def import_from(self, device_ptr: int, nbytes: int) -> None:
        _check(self._lib.prism_gpu_state_import(self._h, C.c_void_p(device_ptr), nbytes))

    def merge_from(self, device_ptr: int, n_groups: int) -> None:
        _check(self._lib.prism_gpu_state_merge(self._h, C.c_void_p(device_ptr), n_groups))

    def finish(self) -> pa.Table:
        out = abi.ArrowArray()
        _check(self._lib.prism_gpu_state_finish(self._h, C.byref(out)))
        return pa.Table.from_batches([abi.import_batch(out, self.plan.final_schema)])

    def close(self) -> None:
        if self._h:
            self._lib.prism_gpu_state_free(self._h)
            self._h = C.c_void_p()


def parquet_read(file_bytes: bytes) -> pa.Table:
    """prism_gpu_parquet_read: the library's own Parquet decode, as a pyarrow table (host only)."""
    lib = load_library()
    schema, arr = abi.ArrowSchema(), abi.ArrowArray()
    _check(lib.prism_gpu_parquet_read(file_bytes, len(file_bytes), C.byref(schema), C.byref(arr)))
    sch = abi.import_schema(schema)
    return pa.Table.from_batches([abi.import_batch(arr, sch)])


class Comm:
    """prism_gpu_comm: the ranks of one sharded query (one process per GPU)."""

    ID_BYTES = 128

    def __init__(self, ctx: Context, handle, keep=None):
        self._lib = ctx._lib
        self._h = handle
        self._keep = keep  # the callback objects must outlive the communicator

    @staticmethod
    def unique_id() -> bytes:
        """128 bytes that rank 0 makes and hands to the other ranks out of band."""
        buf = C.create_string_buffer(Comm.ID_BYTES)
        _check(load_library().prism_gpu_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, ctx: Context, unique_id: bytes, rank: int, world: int) -> "Comm":
        """RCCL communicator over the GPUs of the node (xGMI)."""
        h = C.c_void_p()
        _check(ctx._lib.prism_gpu_comm_init(ctx._h, C.c_char_p(unique_id), rank, world, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_process_group(cls, ctx: Context) -> "Comm":
        """RCCL communicator with the ranks of the default torch.distributed group; torch is
        used once, to hand rank 0's unique id to the others, and for nothing afterwards."""
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()
        # (a rank 0 that cannot make the id says so to the others instead of leaving them in
        # the broadcast: every rank then fails alike, and the caller's fallback stays in step)
        box = [None, None]
        if rank == 0:
            try:
                box[0] = cls.unique_id()
            except PrismGpuError as e:
                box[1] = (e.code, e.msg)
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise PrismGpuError(*box[1])
        return cls.rccl(ctx, box[0], rank, world)

    @classmethod
    def from_callbacks(cls, ctx: Context, rank: int, world: int, allgather, reduce, alltoallv) -> "Comm":
        """The callbacks transport (prism_gpu_transport): three Python callables with the C
        signatures of the struct's members, returning 0 on success."""
        t = abi.Transport()
        keep = (abi.ALLGATHER_FN(allgather), abi.REDUCE_FN(reduce), abi.ALLTOALLV_FN(alltoallv))
        t.user = None
        t.allgather, t.reduce, t.alltoallv = keep
        h = C.c_void_p()
        _check(ctx._lib.prism_gpu_comm_init_callbacks(ctx._h, C.byref(t), rank, world, C.byref(h)))
        return cls(ctx, h, keep)

    def stats(self) -> dict:
        st = abi.CommStats()
        _check(self._lib.prism_gpu_comm_get_stats(self._h, C.byref(st)))
        return {"collectives": st.collectives, "bytes_sent": st.bytes_sent, "exchange_ms": st.exchange_ms}

    def close(self) -> None:
        if self._h:
            self._lib.prism_gpu_comm_free(self._h)
            self._h = C.c_void_p()
