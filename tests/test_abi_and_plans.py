"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol
the header declares, and the plan builder lowers to the C structs as intended."""
import ctypes as C
import os
import re

import pyarrow as pa
import pytest

from prism_amd import _cabi as abi
from prism_amd import engine
from prism_amd.plan import QueryPlan, col, count_star, date_bin, lower, max_, sum_

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    """Every function the headers under include/ declare (the boundary, include/prism_gpu.h,
    and the test / benchmark support in include/prism_gpu_synth.h)."""
    names = set()
    for h in ("prism_gpu.h", "prism_gpu_synth.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(prism_gpu_[a-z_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = engine.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} is declared under include/ but not exported"
    assert sorted(engine.EXPORTED_SYMBOLS) == names


def test_struct_layouts_match_the_header():
    # sizes the C compiler gives the same structs (LP64): keeps _cabi.py honest
    assert C.sizeof(abi.ArrowSchema) == 72 and C.sizeof(abi.ArrowArray) == 80
    assert C.sizeof(abi.Expr) == 48 and C.sizeof(abi.Key) == 32 and C.sizeof(abi.Agg) == 16
    assert C.sizeof(abi.PlanDesc) == 80 and C.sizeof(abi.SynthCol) == 56
    assert C.sizeof(abi.KernelStat) == 72
    assert C.sizeof(abi.Stats) == 48 + 72 * abi.MAX_KERNEL_STATS + 8


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(engine.PrismGpuError) as e:
        engine.Context(0)
    assert e.value.code == abi.EHIP
    assert "no CPU fallback" in e.value.msg


SCHEMA = pa.schema([("bytes", pa.int64()), ("method", pa.string()), ("ts", pa.timestamp("ms"))])


def test_count_and_count_by_lowering():
    """The two shapes prism-ql lowers to (src/prism-ql/src/lower.rs:199-227)."""
    lp = lower(QueryPlan([count_star()]), SCHEMA.names)
    assert lp.desc.n_keys == 0 and lp.desc.n_aggs == 1 and lp.desc.aggs[0].func == abi.AGG_COUNT_STAR
    assert lp.desc.predicate == -1 and lp.desc.abi_version == abi.ABI_VERSION
    lp = lower(QueryPlan([count_star()], keys=["method"]), SCHEMA.names)
    assert lp.desc.n_keys == 1 and lp.desc.keys[0].column == 1 and lp.desc.keys[0].kind == abi.KEY_COLUMN


def test_unknown_column_is_rejected():
    """Counterpart of prism-ql's E0001 `column does not exist` diagnostic
    (src/prism-ql/src/lower.rs:229-260)."""
    with pytest.raises(KeyError, match="something"):
        lower(QueryPlan([count_star()], keys=["something"]), SCHEMA.names)


def test_predicate_tree_lowering():
    q = QueryPlan([sum_("bytes"), max_("bytes")], keys=[date_bin(60_000, "ts"), "method"],
                  predicate=~((col("bytes") > 10) & col("method").like("%E%")) | col("ts").is_null(),
                  order_by=2, descending=True, limit=5)
    lp = lower(q, SCHEMA.names)
    d = lp.desc
    kinds = [d.exprs[i].kind for i in range(d.n_exprs)]
    assert kinds.count(abi.EXPR_COLUMN) == 3 and abi.EXPR_LIKE in kinds and abi.EXPR_NOT in kinds
    root = d.exprs[d.predicate]
    assert root.kind == abi.EXPR_OR
    assert d.exprs[root.lhs].kind == abi.EXPR_NOT and d.exprs[root.rhs].kind == abi.EXPR_IS_NULL
    like = next(d.exprs[i] for i in range(d.n_exprs) if d.exprs[i].kind == abi.EXPR_LIKE)
    pat = d.exprs[like.rhs]
    assert C.string_at(pat.str, pat.str_len) == b"%E%"
    assert d.keys[0].kind == abi.KEY_DATE_BIN and d.keys[0].stride_ms == 60_000
    assert (d.order_by, d.descending, d.limit) == (2, 1, 5)
