"""ctypes driver for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; the product path (prism_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import pyarrow as pa

from prism_amd import _cabi as abi
from prism_amd.plan import QueryPlan, lower

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None


def build() -> str:
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return os.path.join(_HERE, "liboracle.so")


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.prism_oracle_last_error.restype = C.c_char_p
        L.prism_oracle_execute.argtypes = [C.POINTER(abi.ArrowSchema), C.POINTER(C.POINTER(abi.ArrowArray)), C.c_size_t,
                                           C.POINTER(abi.PlanDesc), C.c_int, C.POINTER(abi.ArrowArray)]
        L.prism_oracle_merge.argtypes = [C.POINTER(abi.ArrowSchema), C.POINTER(abi.ArrowSchema),
                                         C.POINTER(C.POINTER(abi.ArrowArray)), C.c_size_t, C.POINTER(abi.PlanDesc),
                                         C.POINTER(abi.ArrowArray)]
        L.prism_oracle_output_schema.argtypes = [C.POINTER(abi.ArrowSchema), C.POINTER(abi.PlanDesc), C.c_int,
                                                 C.POINTER(abi.ArrowSchema)]
        L.prism_oracle_synth_create.argtypes = [C.POINTER(abi.ArrowSchema), C.c_uint64, C.c_uint64, C.c_uint64,
                                                C.c_uint64, C.c_uint32, C.POINTER(abi.SynthCol), C.c_uint32, C.c_int,
                                                C.POINTER(C.c_void_p)]
        L.prism_oracle_synth_free.argtypes = [C.c_void_p]
        L.prism_oracle_synth_execute.argtypes = [C.c_void_p, C.POINTER(abi.ArrowSchema), C.POINTER(abi.PlanDesc),
                                                 C.c_int, C.POINTER(abi.ArrowArray), C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


class OracleError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code


def _check(rc: int) -> None:
    if rc != 0:
        raise OracleError(rc, lib().prism_oracle_last_error().decode())


def output_schema(schema: pa.Schema, plan: QueryPlan, partial: Optional[bool] = None) -> pa.Schema:
    lp = lower(plan, schema.names)
    cs = abi.export_schema(schema)
    out = abi.ArrowSchema()
    try:
        part = (plan.mode == abi.MODE_PARTIAL) if partial is None else partial
        _check(lib().prism_oracle_output_schema(C.byref(cs), lp.ptr(), 1 if part else 0, C.byref(out)))
        return abi.import_schema(out)
    finally:
        abi.release_schema(cs)


def _as_batches(data) -> list:
    if isinstance(data, pa.Table):
        return data.combine_chunks().to_batches() or [pa.RecordBatch.from_pylist([], schema=data.schema)]
    if isinstance(data, pa.RecordBatch):
        return [data]
    return list(data)


def execute(data, plan: QueryPlan, threads: int = 1) -> pa.Table:
    """Run `plan` over a pyarrow Table / RecordBatch / list of RecordBatches."""
    batches = _as_batches(data)
    schema = batches[0].schema
    lp = lower(plan, schema.names)
    cs = abi.export_schema(schema)
    arrs = [abi.export_batch(b) for b in batches]
    ptrs = (C.POINTER(abi.ArrowArray) * len(arrs))(*[C.pointer(a) for a in arrs])
    out = abi.ArrowArray()
    try:
        _check(lib().prism_oracle_execute(C.byref(cs), ptrs, len(arrs), lp.ptr(), threads, C.byref(out)))
        osch = output_schema(schema, plan)
        return pa.Table.from_batches([abi.import_batch(out, osch)])
    finally:
        for a in arrs:
            abi.release_array(a)
        abi.release_schema(cs)


def merge(input_schema: pa.Schema, states: Sequence, plan: QueryPlan) -> pa.Table:
    """AggregateExec(Final) over PARTIAL outputs of `plan`."""
    states = [s.combine_chunks().to_batches()[0] if isinstance(s, pa.Table) else s for s in states]
    lp = lower(plan, input_schema.names)
    cs = abi.export_schema(input_schema)
    ss = abi.export_schema(states[0].schema)
    arrs = [abi.export_batch(b) for b in states]
    ptrs = (C.POINTER(abi.ArrowArray) * len(arrs))(*[C.pointer(a) for a in arrs])
    out = abi.ArrowArray()
    try:
        _check(lib().prism_oracle_merge(C.byref(cs), C.byref(ss), ptrs, len(arrs), lp.ptr(), C.byref(out)))
        osch = output_schema(input_schema, plan, partial=False)
        return pa.Table.from_batches([abi.import_batch(out, osch)])
    finally:
        for a in arrs:
            abi.release_array(a)
        abi.release_schema(cs)
        abi.release_schema(ss)


class SynthTable:
    """A synthetic table generated on the CPU with the generators of prism_synth.h."""

    def __init__(self, schema: pa.Schema, synth_cols, seed: int, first_row: int, n_rows: int, total_rows: int,
                 n_parts: int, threads: int = 1):
        self.schema = schema
        self._cs = abi.export_schema(schema)
        self._keep = synth_cols
        arr = (abi.SynthCol * len(synth_cols))(*synth_cols)
        self._h = C.c_void_p()
        _check(lib().prism_oracle_synth_create(C.byref(self._cs), seed, first_row, n_rows, total_rows, n_parts, arr,
                                               len(synth_cols), threads, C.byref(self._h)))
        self.rows = n_rows

    def execute(self, plan: QueryPlan, threads: int = 1):
        """Returns (pyarrow.Table, elapsed_ms of the scan/filter/aggregate/merge work)."""
        lp = lower(plan, self.schema.names)
        out = abi.ArrowArray()
        ms = C.c_double(0)
        _check(lib().prism_oracle_synth_execute(self._h, C.byref(self._cs), lp.ptr(), threads, C.byref(out),
                                                C.byref(ms)))
        # generated Utf8 columns are dictionary codes (the output schema has Int32 there),
        # except those made by the URL generator, which are real strings
        raw = {sc.column for sc in self._keep if sc.kind == abi.SYNTH_URL}
        fields = [pa.field(f.name, pa.int32() if pa.types.is_string(f.type) and i not in raw else f.type)
                  for i, f in enumerate(self.schema)]
        osch = output_schema(pa.schema(fields), plan)
        return pa.Table.from_batches([abi.import_batch(out, osch)]), ms.value

    def close(self):
        if self._h:
            lib().prism_oracle_synth_free(self._h)
            self._h = C.c_void_p()
            abi.release_schema(self._cs)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
