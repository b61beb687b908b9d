"""The benchmark configurations: schema, synthetic generators and query of each.

The reference stages ClickBench `hits` into `s3://query/public/hits`
(dev/clickbench.sh:17) and demo logs into `demo/http_requests`
(misc/vector.json:19-38); neither dataset is in its tree, so every workload here
is generated from a seed with the counter-based generators of
include/prism_synth.h, which the GPU and the CPU checker evaluate identically.

    C1  hits-like:   COUNT(*) WHERE "AdvEngineID" <> 0
    C2  log events:  date_bin(1 minute, timestamp), service -> COUNT(*), SUM(value), MAX(value)
    C3  hits-like:   "UserID", COUNT(*) GROUP BY 1 ORDER BY 2 DESC LIMIT 10
    C4  hits-like:   COUNT(*) WHERE "URL" LIKE '%google%'  (URL-like strings, mean 64 bytes)
    C4G              "UserID", COUNT(*) WHERE "URL" LIKE '%google%' GROUP BY 1  (the compaction variant)
    C5  time series: date_bin(1 hour, timestamp), service -> COUNT(DISTINCT user)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Sequence

import pyarrow as pa

from . import _cabi as abi
from .plan import QueryPlan, col, count_distinct, count_star, date_bin, max_, sum_


def url_length_table(n: int = 1024, mu: float = 4.0, sigma: float = 0.6, lo: int = 8, hi: int = 512) -> List[int]:
    """Quantiles of clamp(lognormal(mu, sigma), lo, hi), in bytes: the length table of the
    URL generator.  Computed with the inverse normal CDF at the bin centres."""
    import math
    from statistics import NormalDist

    nd = NormalDist()
    out = []
    for i in range(n):
        z = nd.inv_cdf((i + 0.5) / n)
        out.append(int(min(hi, max(lo, round(math.exp(mu + sigma * z))))))
    return out


def zipf_cdf(n: int, s: float) -> List[int]:
    """Ascending u32 thresholds of a Zipf(s) law over n categories."""
    w = [1.0 / (k ** s) for k in range(1, n + 1)]
    tot = sum(w)
    acc, out = 0.0, []
    for x in w:
        acc += x
        out.append(min(0xFFFFFFFF, int(acc / tot * 4294967296.0)))
    out[-1] = 0xFFFFFFFF
    return out


def synth_col(column: int, kind: int, p: Sequence[int] = (), cdf: Sequence[int] = ()) -> abi.SynthCol:
    import ctypes as C

    sc = abi.SynthCol()
    sc.column = column
    sc.kind = kind
    for i, v in enumerate(p):
        sc.p[i] = int(v)
    if cdf:
        arr = (C.c_uint32 * len(cdf))(*cdf)
        sc._keep = arr  # keep the table alive with the struct
        sc.cdf = C.cast(arr, C.POINTER(C.c_uint32))
        sc.cdf_len = len(cdf)
    return sc


@dataclass
class Workload:
    name: str
    description: str
    schema: pa.Schema
    make_cols: Callable[[], List[abi.SynthCol]]
    query: QueryPlan
    n_keys: int
    seed: int
    default_rows: int
    default_partitions: int
    bytes_per_row: int          # algorithmic bytes per scanned row (columns the query references)
    bytes_per_group: int        # algorithmic bytes per output group
    dictionaries: dict          # column name -> list of strings (generated dictionary columns)


T0_MS = 1_700_000_000_000
DAY_MS = 86_400_000


def c1() -> Workload:
    schema = pa.schema([("AdvEngineID", pa.int16())])
    return Workload(
        "C1", 'SELECT COUNT(*) FROM hits WHERE "AdvEngineID" <> 0', schema,
        lambda: [synth_col(0, abi.SYNTH_SPARSE, (int(0.0063 * 2 ** 32), 60))],
        QueryPlan([count_star()], predicate=col("AdvEngineID") != 0), 0, seed=1,
        default_rows=100_000_000, default_partitions=10, bytes_per_row=2, bytes_per_group=8, dictionaries={})


def c2(n_services: int = 64) -> Workload:
    schema = pa.schema([("timestamp", pa.timestamp("ms")), ("service", pa.string()), ("value", pa.int64())])
    import os

    cdf = zipf_cdf(n_services, float(os.environ.get("PRISM_C2_ZIPF", "1.1")))  # env override: experiments only
    return Workload(
        "C2", "SELECT date_bin(INTERVAL '1 minute', timestamp), service, COUNT(*), SUM(value), MAX(value) "
              "FROM events GROUP BY 1, 2", schema,
        lambda: [synth_col(0, abi.SYNTH_TS_LINEAR, (T0_MS, DAY_MS)),
                 synth_col(1, abi.SYNTH_CATEGORICAL, (), cdf),
                 synth_col(2, abi.SYNTH_UNIFORM, (0, 50_000))],
        QueryPlan([count_star(), sum_("value"), max_("value")], keys=[date_bin(60_000, "timestamp"), "service"]),
        2, seed=2, default_rows=1_000_000_000, default_partitions=64, bytes_per_row=20, bytes_per_group=36,
        dictionaries={"service": [f"svc-{i:02d}" for i in range(n_services)]})


def c3(id_bits: int = 24) -> Workload:
    schema = pa.schema([("UserID", pa.int64())])
    return Workload(
        "C3", 'SELECT "UserID", COUNT(*) FROM hits GROUP BY "UserID" ORDER BY COUNT(*) DESC LIMIT 10', schema,
        lambda: [synth_col(0, abi.SYNTH_LOGUNIFORM_ID, (id_bits, 0x5EED))],
        QueryPlan([count_star()], keys=["UserID"], order_by=1, descending=True, limit=10), 1, seed=3,
        default_rows=100_000_000, default_partitions=100, bytes_per_row=8, bytes_per_group=16, dictionaries={})


def _c4(name: str, grouped: bool) -> Workload:
    schema = pa.schema([("URL", pa.string()), ("UserID", pa.int64())])
    lens = url_length_table()
    mean_len = sum(lens) / len(lens)
    cols = lambda: [synth_col(0, abi.SYNTH_URL, (int(0.02 * 2 ** 32), int(0.01 * 2 ** 32)), lens),  # noqa: E731
                    synth_col(1, abi.SYNTH_LOGUNIFORM_ID, (24, 0x5EED))]
    if grouped:
        q = QueryPlan([count_star()], keys=["UserID"], predicate=col("URL").like("%google%"))
        sql = 'SELECT "UserID", COUNT(*) FROM hits WHERE "URL" LIKE \'%google%\' GROUP BY "UserID"'
    else:
        q = QueryPlan([count_star()], predicate=col("URL").like("%google%"))
        sql = 'SELECT COUNT(*) FROM hits WHERE "URL" LIKE \'%google%\''
    # offsets 4 B + mean string bytes per row (+ 8 B UserID for the selected rows only: not counted)
    return Workload(name, sql, schema, cols, q, 1 if grouped else 0, seed=4, default_rows=100_000_000,
                    default_partitions=100, bytes_per_row=4 + int(round(mean_len)), bytes_per_group=16 if grouped else 8,
                    dictionaries={})


def c4() -> Workload:
    return _c4("C4", False)


def c4g() -> Workload:
    return _c4("C4G", True)


def c5(n_services: int = 256, user_bits: int = 26) -> Workload:
    schema = pa.schema([("timestamp", pa.timestamp("ms")), ("service", pa.string()), ("user", pa.int64())])
    cdf = zipf_cdf(n_services, 1.1)
    return Workload(
        "C5", "SELECT date_bin(INTERVAL '1 hour', timestamp), service, COUNT(DISTINCT user) FROM series GROUP BY 1, 2",
        schema,
        lambda: [synth_col(0, abi.SYNTH_TS_LINEAR, (T0_MS, 7 * DAY_MS)),
                 synth_col(1, abi.SYNTH_CATEGORICAL, (), cdf),
                 synth_col(2, abi.SYNTH_LOGUNIFORM_ID, (user_bits, 0xC5))],
        QueryPlan([count_distinct("user")], keys=[date_bin(3_600_000, "timestamp"), "service"]), 2, seed=5,
        default_rows=1_250_000_000, default_partitions=64, bytes_per_row=20, bytes_per_group=20,
        dictionaries={"service": [f"svc-{i:03d}" for i in range(n_services)]})


WORKLOADS = {"C1": c1, "C2": c2, "C3": c3, "C4": c4, "C4G": c4g, "C5": c5}


def partition_bounds(n_rows: int, n_parts: int, p: int):
    """Rows [lo, hi) of partition p when n_rows are split into n_parts contiguous runs
    (the same split the CPU checker's generator uses)."""
    return n_rows * p // n_parts, n_rows * (p + 1) // n_parts


def build_gpu_table(ctx, wl: Workload, first_row: int, n_rows: int, total_rows: int, n_parts: int):
    """Generates rows [first_row, first_row + n_rows) of the workload's table in HBM."""
    from .engine import Table

    t = Table(ctx, wl.schema)
    for name, values in wl.dictionaries.items():
        t.set_dictionary(name, values)
    cols = wl.make_cols()
    for p in range(n_parts):
        lo, hi = partition_bounds(n_rows, n_parts, p)
        if hi > lo:
            t.synth(p, wl.seed, first_row + lo, hi - lo, total_rows, cols)
    return t


def decode_dictionary_columns(wl: Workload, table: pa.Table) -> pa.Table:
    """The CPU checker's generated tables carry dictionary codes (Int32) where the
    GPU result carries the strings; this maps the codes to strings."""
    out = table
    for name, values in wl.dictionaries.items():
        if name in out.schema.names:
            i = out.schema.get_field_index(name)
            codes = out.column(i).to_pylist()
            out = out.set_column(i, name, pa.array([None if c is None else values[c] for c in codes], pa.string()))
    return out
