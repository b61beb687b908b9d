"""Plan builder: a small expression vocabulary named after DataFusion's expr_fn
(`col`, `lit`, `count`, `sum`, `min`, `max`, `avg`, `date_bin`) that lowers to the
C structs of include/prism_gpu.h.

The reference builds the same shapes through DataFusion:
`LogicalPlanBuilder::aggregate(group_by, [count(*)])` in
src/prism-ql/src/lower.rs:83-95, `df.aggregate([], [max(col), min(col)])` in
src/prism-ingest/src/ingest.rs:67-77, and arbitrary SQL through `ctx.sql`
(src/prism-query/src/main.rs:89).  A `QueryPlan` is what remains of such a plan
between the scan and the final batches: WHERE, GROUP BY, aggregates, ORDER BY,
LIMIT.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

from . import _cabi as abi


# ---------------------------------------------------------------- predicates


class Predicate:
    def __and__(self, other: "Predicate") -> "Predicate":
        return _Bool(abi.EXPR_AND, self, other)

    def __or__(self, other: "Predicate") -> "Predicate":
        return _Bool(abi.EXPR_OR, self, other)

    def __invert__(self) -> "Predicate":
        return _Bool(abi.EXPR_NOT, self, None)


@dataclass
class _Bool(Predicate):
    kind: int
    lhs: Predicate
    rhs: Optional[Predicate]


@dataclass
class _Cmp(Predicate):
    op: int
    column: str
    literal: Union[int, str, bytes]


@dataclass
class _Like(Predicate):
    column: str
    pattern: Union[str, bytes]
    negated: bool = False
    case_insensitive: bool = False


@dataclass
class _IsNull(Predicate):
    column: str
    negated: bool = False


@dataclass
class Column:
    name: str

    def _cmp(self, op: int, other) -> Predicate:
        if isinstance(other, Lit):
            other = other.value
        return _Cmp(op, self.name, other)

    def __eq__(self, other):  # type: ignore[override]
        return self._cmp(abi.CMP_EQ, other)

    def __ne__(self, other):  # type: ignore[override]
        return self._cmp(abi.CMP_NEQ, other)

    def __lt__(self, other):
        return self._cmp(abi.CMP_LT, other)

    def __le__(self, other):
        return self._cmp(abi.CMP_LTEQ, other)

    def __gt__(self, other):
        return self._cmp(abi.CMP_GT, other)

    def __ge__(self, other):
        return self._cmp(abi.CMP_GTEQ, other)

    def like(self, pattern) -> Predicate:
        return _Like(self.name, pattern)

    def not_like(self, pattern) -> Predicate:
        return _Like(self.name, pattern, negated=True)

    def ilike(self, pattern) -> Predicate:
        return _Like(self.name, pattern, case_insensitive=True)

    def is_null(self) -> Predicate:
        return _IsNull(self.name)

    def is_not_null(self) -> Predicate:
        return _IsNull(self.name, negated=True)

    __hash__ = None  # type: ignore[assignment]


@dataclass
class Lit:
    value: Union[int, str, bytes]


def col(name: str) -> Column:
    return Column(name)


def lit(value) -> Lit:
    return Lit(value)


# ------------------------------------------------------- keys and aggregates


@dataclass
class GroupKey:
    column: str
    kind: int = abi.KEY_COLUMN
    stride_ms: int = 0
    origin_ms: int = 0
    name: Optional[str] = None


def date_bin(stride_ms: int, column: Union[str, Column], origin_ms: int = 0, name: Optional[str] = None) -> GroupKey:
    """date_bin(INTERVAL '<stride_ms> milliseconds', column, origin)."""
    cname = column.name if isinstance(column, Column) else column
    return GroupKey(cname, abi.KEY_DATE_BIN, int(stride_ms), int(origin_ms), name)


@dataclass
class Aggregate:
    func: int
    column: Optional[str] = None
    name: Optional[str] = None

    def alias(self, name: str) -> "Aggregate":
        return Aggregate(self.func, self.column, name)


def _cname(c) -> str:
    return c.name if isinstance(c, Column) else c


def count_star() -> Aggregate:
    return Aggregate(abi.AGG_COUNT_STAR)


def count(c) -> Aggregate:
    return Aggregate(abi.AGG_COUNT, _cname(c))


def sum_(c) -> Aggregate:
    return Aggregate(abi.AGG_SUM, _cname(c))


def min_(c) -> Aggregate:
    return Aggregate(abi.AGG_MIN, _cname(c))


def max_(c) -> Aggregate:
    return Aggregate(abi.AGG_MAX, _cname(c))


def avg(c) -> Aggregate:
    return Aggregate(abi.AGG_AVG, _cname(c))


def count_distinct(c) -> Aggregate:
    return Aggregate(abi.AGG_COUNT_DISTINCT, _cname(c))


# ----------------------------------------------------------------- the plan


@dataclass
class QueryPlan:
    """WHERE / GROUP BY / aggregates / ORDER BY / LIMIT over one table."""

    aggs: Sequence[Aggregate]
    keys: Sequence[Union[str, GroupKey]] = ()
    predicate: Optional[Predicate] = None
    mode: int = abi.MODE_SINGLE
    order_by: Optional[int] = None  # index into the output columns (keys first)
    descending: bool = False
    limit: Optional[int] = None
    groups_hint: int = 0

    def with_mode(self, mode: int) -> "QueryPlan":
        return QueryPlan(self.aggs, self.keys, self.predicate, mode, self.order_by, self.descending, self.limit,
                         self.groups_hint)


@dataclass
class LoweredPlan:
    """A prism_gpu_plan_desc plus the ctypes objects that keep its pointers alive."""

    desc: abi.PlanDesc
    _keep: list = field(default_factory=list)

    def ptr(self):
        return C.byref(self.desc)


def _bytes(v) -> bytes:
    return v.encode("utf-8") if isinstance(v, str) else bytes(v)


def lower(plan: QueryPlan, column_names: Sequence[str]) -> LoweredPlan:
    """QueryPlan -> prism_gpu_plan_desc against a schema's column names."""
    names = list(column_names)

    def cidx(name: str) -> int:
        if name not in names:
            raise KeyError(f"column `{name}` does not exist")
        return names.index(name)

    exprs: List[abi.Expr] = []
    keep: list = []

    def add(**kw) -> int:
        e = abi.Expr(kind=0, op=0, lhs=-1, rhs=-1, column=-1, flags=0, i64=0, str=None, str_len=0)
        for k, v in kw.items():
            setattr(e, k, v)
        exprs.append(e)
        return len(exprs) - 1

    def add_str(v) -> int:
        b = _bytes(v)
        buf = C.create_string_buffer(b, len(b) + 1)
        keep.append(buf)
        return add(kind=abi.EXPR_LIT_STR, str=C.cast(buf, C.c_char_p), str_len=len(b))

    def walk(p: Predicate) -> int:
        if isinstance(p, _Bool):
            l = walk(p.lhs)
            r = walk(p.rhs) if p.rhs is not None else -1
            return add(kind=p.kind, lhs=l, rhs=r)
        if isinstance(p, _Cmp):
            c = add(kind=abi.EXPR_COLUMN, column=cidx(p.column))
            if isinstance(p.literal, (str, bytes)):
                r = add_str(p.literal)
            else:
                r = add(kind=abi.EXPR_LIT_I64, i64=int(p.literal))
            return add(kind=abi.EXPR_CMP, op=p.op, lhs=c, rhs=r)
        if isinstance(p, _Like):
            c = add(kind=abi.EXPR_COLUMN, column=cidx(p.column))
            r = add_str(p.pattern)
            flags = (abi.LIKE_NEGATED if p.negated else 0) | (abi.LIKE_CASE_INSENSITIVE if p.case_insensitive else 0)
            return add(kind=abi.EXPR_LIKE, lhs=c, rhs=r, flags=flags)
        if isinstance(p, _IsNull):
            c = add(kind=abi.EXPR_COLUMN, column=cidx(p.column))
            return add(kind=abi.EXPR_IS_NOT_NULL if p.negated else abi.EXPR_IS_NULL, lhs=c)
        raise TypeError(f"not a predicate: {p!r}")

    root = walk(plan.predicate) if plan.predicate is not None else -1

    def cstr(s: Optional[str]):
        if s is None:
            return None
        buf = C.create_string_buffer(s.encode("utf-8"))
        keep.append(buf)
        return C.cast(buf, C.c_char_p)

    keys = []
    for k in plan.keys:
        if isinstance(k, str):
            k = GroupKey(k)
        keys.append(abi.Key(kind=k.kind, column=cidx(k.column), stride_ms=k.stride_ms, origin_ms=k.origin_ms,
                            name=cstr(k.name)))
    aggs = []
    for a in plan.aggs:
        aggs.append(abi.Agg(func=a.func, column=cidx(a.column) if a.column is not None else -1, name=cstr(a.name)))

    expr_arr = (abi.Expr * max(1, len(exprs)))(*exprs)
    key_arr = (abi.Key * max(1, len(keys)))(*keys)
    agg_arr = (abi.Agg * max(1, len(aggs)))(*aggs)
    keep += [expr_arr, key_arr, agg_arr]
    desc = abi.PlanDesc(
        abi_version=abi.ABI_VERSION,
        mode=plan.mode,
        exprs=C.cast(expr_arr, C.POINTER(abi.Expr)),
        n_exprs=len(exprs),
        predicate=root,
        keys=C.cast(key_arr, C.POINTER(abi.Key)),
        n_keys=len(keys),
        aggs=C.cast(agg_arr, C.POINTER(abi.Agg)),
        n_aggs=len(aggs),
        order_by=-1 if plan.order_by is None else int(plan.order_by),
        descending=1 if plan.descending else 0,
        limit=-1 if plan.limit is None else int(plan.limit),
        groups_hint=int(plan.groups_hint),
    )
    return LoweredPlan(desc, keep)
