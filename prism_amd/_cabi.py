"""ctypes mirror of include/prism_gpu.h and include/arrow_c_data.h.

Only declarations live here: struct layouts, enum values, and helpers that move
pyarrow objects across the Arrow C Data Interface.  pyarrow is used by the test
driver and the benchmark harness to make and read RecordBatches; the library
itself never sees Python objects, only these C structs.
"""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 1

# prism_gpu_status
OK, EINVAL, EUNSUPPORTED, EOOM, EHIP, ERCCL, EINTERNAL = 0, -1, -2, -3, -4, -5, -6

# prism_gpu_col_type == proto ColumnType numbering
# (proto/common/prism/common/v1/partition.proto:26-38 in the reference)
COL_INT64, COL_UTF8, COL_TIMESTAMP, COL_INT16, COL_INT32, COL_UINT16, COL_BINARY = 1, 2, 3, 4, 5, 6, 7

# prism_gpu_expr_kind
(EXPR_COLUMN, EXPR_LIT_I64, EXPR_LIT_STR, EXPR_CMP, EXPR_AND, EXPR_OR, EXPR_NOT, EXPR_LIKE,
 EXPR_IS_NULL, EXPR_IS_NOT_NULL) = range(1, 11)
# prism_gpu_cmp_op
CMP_EQ, CMP_NEQ, CMP_LT, CMP_LTEQ, CMP_GT, CMP_GTEQ = range(1, 7)
LIKE_NEGATED, LIKE_CASE_INSENSITIVE = 1, 2
# prism_gpu_key_kind
KEY_COLUMN, KEY_DATE_BIN = 1, 2
# prism_gpu_agg_func
AGG_COUNT_STAR, AGG_COUNT, AGG_SUM, AGG_MIN, AGG_MAX, AGG_AVG, AGG_COUNT_DISTINCT = range(1, 8)
# prism_gpu_agg_mode
MODE_SINGLE, MODE_PARTIAL = 0, 1
# prism_synth_kind
SYNTH_TS_LINEAR, SYNTH_CATEGORICAL, SYNTH_UNIFORM, SYNTH_SPARSE, SYNTH_LOGUNIFORM_ID, SYNTH_URL = range(1, 7)


class ArrowSchema(C.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", C.c_char_p),
    ("name", C.c_char_p),
    ("metadata", C.c_char_p),
    ("flags", C.c_int64),
    ("n_children", C.c_int64),
    ("children", C.POINTER(C.POINTER(ArrowSchema))),
    ("dictionary", C.POINTER(ArrowSchema)),
    ("release", C.c_void_p),
    ("private_data", C.c_void_p),
]


class ArrowArray(C.Structure):
    pass


ArrowArray._fields_ = [
    ("length", C.c_int64),
    ("null_count", C.c_int64),
    ("offset", C.c_int64),
    ("n_buffers", C.c_int64),
    ("n_children", C.c_int64),
    ("buffers", C.POINTER(C.c_void_p)),
    ("children", C.POINTER(C.POINTER(ArrowArray))),
    ("dictionary", C.POINTER(ArrowArray)),
    ("release", C.c_void_p),
    ("private_data", C.c_void_p),
]


class Expr(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("op", C.c_int32),
        ("lhs", C.c_int32),
        ("rhs", C.c_int32),
        ("column", C.c_int32),
        ("flags", C.c_int32),
        ("i64", C.c_int64),
        ("str", C.c_char_p),
        ("str_len", C.c_uint64),
    ]


class Key(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("column", C.c_int32),
        ("stride_ms", C.c_int64),
        ("origin_ms", C.c_int64),
        ("name", C.c_char_p),
    ]


class Agg(C.Structure):
    _fields_ = [("func", C.c_int32), ("column", C.c_int32), ("name", C.c_char_p)]


class PlanDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("mode", C.c_int32),
        ("exprs", C.POINTER(Expr)),
        ("n_exprs", C.c_uint32),
        ("predicate", C.c_int32),
        ("keys", C.POINTER(Key)),
        ("n_keys", C.c_uint32),
        ("aggs", C.POINTER(Agg)),
        ("n_aggs", C.c_uint32),
        ("order_by", C.c_int32),
        ("descending", C.c_int32),
        ("limit", C.c_int64),
        ("groups_hint", C.c_uint64),
    ]


class SynthCol(C.Structure):
    _fields_ = [
        ("column", C.c_int32),
        ("kind", C.c_int32),
        ("p", C.c_int64 * 4),
        ("cdf", C.POINTER(C.c_uint32)),
        ("cdf_len", C.c_uint32),
    ]


MAX_KERNEL_STATS = 24


class ColStats(C.Structure):
    _fields_ = [("min", C.c_int64), ("max", C.c_int64), ("nulls", C.c_uint64)]


MAX_SLOTS = 24
SLOT_SUM_I64, SLOT_SUM_F64, SLOT_MIN_I64, SLOT_MAX_I64 = 0, 1, 2, 3


class StateInfo(C.Structure):
    _fields_ = [
        ("dense", C.c_int32),
        ("n_slots", C.c_int32),
        ("entries", C.c_uint64),
        ("export_bytes", C.c_uint64),
        ("slot_merge", C.c_int32 * MAX_SLOTS),
    ]


REDUCE_SUM, REDUCE_MIN, REDUCE_MAX = 0, 1, 2
REDUCE_I64, REDUCE_F64 = 0, 1

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                           C.POINTER(C.c_uint64))


class Transport(C.Structure):
    """prism_gpu_transport"""
    _fields_ = [
        ("user", C.c_void_p),
        ("allgather", ALLGATHER_FN),
        ("reduce", REDUCE_FN),
        ("alltoallv", ALLTOALLV_FN),
    ]


class CommStats(C.Structure):
    _fields_ = [
        ("collectives", C.c_uint64),
        ("bytes_sent", C.c_uint64),
        ("exchange_ms", C.c_double),
    ]


class KernelStat(C.Structure):
    _fields_ = [
        ("name", C.c_char * 48),
        ("launches", C.c_uint64),
        ("total_ms", C.c_double),
        ("algorithmic_bytes", C.c_uint64),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rows_scanned", C.c_uint64),
        ("rows_selected", C.c_uint64),
        ("groups_out", C.c_uint64),
        ("bytes_staged", C.c_uint64),
        ("staging_ms", C.c_double),
        ("n_kernels", C.c_uint32),
        ("kernels", KernelStat * MAX_KERNEL_STATS),
        ("chunks_pruned", C.c_uint64),
    ]


_RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p)


def export_schema(schema) -> ArrowSchema:
    """pyarrow.Schema -> ArrowSchema (caller owns it; release with release_schema)."""
    out = ArrowSchema()
    schema._export_to_c(C.addressof(out))
    return out


def export_batch(batch) -> ArrowArray:
    """pyarrow.RecordBatch -> struct-typed ArrowArray (zero copy)."""
    out = ArrowArray()
    batch._export_to_c(C.addressof(out))
    return out


def import_batch(arr: ArrowArray, schema):
    """ArrowArray (ownership moves to pyarrow) + pyarrow.Schema -> RecordBatch."""
    import pyarrow as pa

    return pa.RecordBatch._import_from_c(C.addressof(arr), schema)


def import_schema(s: ArrowSchema):
    import pyarrow as pa

    return pa.Schema._import_from_c(C.addressof(s))


def release_schema(s: ArrowSchema) -> None:
    if s.release:
        _RELEASE_FN(s.release)(C.addressof(s))


def release_array(a: ArrowArray) -> None:
    if a.release:
        _RELEASE_FN(a.release)(C.addressof(a))
