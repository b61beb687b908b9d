"""Shared test helpers: fixture tables, result comparison, plan catalogue."""
from __future__ import annotations

import datetime
import json
import math
import os

import numpy as np
import pyarrow as pa

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerance for float SUM/AVG stated by the task: 1e-9 relative.  Integer
# aggregates, group keys and counts are compared exactly.
FLOAT_RTOL = 1e-9


def _ts_ms(s: str) -> int:
    """RFC 3339 with nanoseconds -> epoch milliseconds, truncating (what
    to_timestamp_millis does in src/prism-ingest/src/ingest.rs:155)."""
    d, frac = s[:-1].split(".")
    base = datetime.datetime.strptime(d, "%Y-%m-%dT%H:%M:%S").replace(tzinfo=datetime.timezone.utc)
    return int(base.timestamp()) * 1000 + int(frac.ljust(9, "0")) // 1_000_000


def demo_table() -> pa.Table:
    """The reference's own ingest fixture (src/prism-ingest/src/testdata/demo.log),
    decoded the way prism-ingest does: all columns as inferred, `timestamp` as
    Timestamp(ms)."""
    rows = [json.loads(l) for l in open(os.path.join(GOLDEN, "demo.log"))]
    cols = {k: [r[k] for r in rows] for k in rows[0] if k != "timestamp"}
    cols["timestamp"] = pa.array([_ts_ms(r["timestamp"]) for r in rows], pa.timestamp("ms"))
    return pa.table(cols)


def _norm(v):
    if isinstance(v, datetime.datetime):
        return v.replace(tzinfo=None)
    return v


def rows_of(t: pa.Table):
    cols = [c.to_pylist() for c in t.columns]
    return [tuple(_norm(c[i]) for c in cols) for i in range(t.num_rows)]


def _sort_key(row, n_keys):
    return tuple((0, "") if v is None else (1, str(type(v).__name__), v) for v in row[:n_keys])


def assert_tables_match(got: pa.Table, want: pa.Table, n_keys: int, ordered: bool = False):
    """Same schema, same rows (order-insensitive unless `ordered`): integers,
    strings, timestamps exact; floats within FLOAT_RTOL."""
    assert got.schema.names == want.schema.names, (got.schema.names, want.schema.names)
    for g, w in zip(got.schema, want.schema):
        assert g.type == w.type, f"column {g.name}: {g.type} != {w.type}"
    assert got.num_rows == want.num_rows, f"{got.num_rows} rows != {want.num_rows}"
    gr, wr = rows_of(got), rows_of(want)
    if not ordered:
        gr.sort(key=lambda r: _sort_key(r, n_keys if n_keys else len(r)))
        wr.sort(key=lambda r: _sort_key(r, n_keys if n_keys else len(r)))
    for a, b in zip(gr, wr):
        for x, y in zip(a, b):
            if isinstance(x, float) and isinstance(y, float):
                assert math.isclose(x, y, rel_tol=FLOAT_RTOL, abs_tol=0.0) or x == y, (a, b)
            else:
                assert x == y, (a, b)


def assert_topk_match(got: pa.Table, want: pa.Table, order_col: int):
    """ORDER BY ... LIMIT k: ties are broken arbitrarily in the reference, so the
    sequence of ORDER BY values must match and every returned row must be a row
    the full result contains (checked by the caller when needed)."""
    assert got.schema.names == want.schema.names
    assert got.num_rows == want.num_rows
    assert got.column(order_col).to_pylist() == want.column(order_col).to_pylist()


def random_table(n: int, seed: int = 0, nulls: bool = True) -> pa.Table:
    """Mixed-type table exercising every supported column type."""
    rng = np.random.default_rng(seed)

    def maybe_null(arr, p=0.1):
        if not nulls:
            return arr
        return pa.array(np.asarray(arr), mask=rng.random(n) < p)

    ts = 1_700_000_000_000 + np.sort(rng.integers(0, 3 * 3600 * 1000, n))
    services = np.array([f"svc-{i:02d}" for i in range(12)])
    svc = services[np.minimum(rng.zipf(1.5, n) - 1, 11)]
    url_parts = np.array(["google", "example", "gogle", "googl", "bing", "admin", "x"])
    urls = ["http://" + "/".join(url_parts[rng.integers(0, len(url_parts), rng.integers(1, 5))]) for _ in range(n)]
    cols = {
        "ts": pa.array(ts, pa.timestamp("ms")),
        "service": pa.array(svc.tolist(), mask=(rng.random(n) < 0.05) if nulls else None),
        "value": maybe_null(rng.integers(-50_000, 50_000, n).astype(np.int64)),
        "i16": maybe_null((rng.integers(0, 5, n) * (rng.random(n) < 0.3)).astype(np.int16)),
        "i32": maybe_null(rng.integers(-1000, 1000, n).astype(np.int32)),
        "u16": maybe_null(rng.integers(0, 65535, n).astype(np.uint16)),
        "user": maybe_null(rng.integers(0, max(2, n // 3), n).astype(np.int64) * 7919 - 5),
        "url": pa.array(urls, mask=(rng.random(n) < 0.05) if nulls else None),
    }
    return pa.table(cols)


def plan_catalogue():
    """(name, QueryPlan, n_keys) over the columns of random_table()."""
    from prism_amd import _cabi as abi
    from prism_amd.plan import (QueryPlan, avg, col, count, count_distinct, count_star, date_bin, max_, min_, sum_)

    P = QueryPlan
    plans = [
        ("count_all", P([count_star()]), 0),
        ("count_i16_ne0", P([count_star()], predicate=col("i16") != 0), 0),
        ("count_i32_range", P([count_star()], predicate=(col("i32") >= -10) & (col("i32") < 250)), 0),
        ("count_u16_gt", P([count_star()], predicate=col("u16") > 40000), 0),
        ("count_value_le_or", P([count_star()], predicate=(col("value") <= -100) | (col("i16") == 3)), 0),
        ("count_not", P([count_star()], predicate=~((col("value") > 0) & (col("i16") == 0))), 0),
        ("count_not_or_null", P([count_star()], predicate=~((col("value") > 0) | col("i32").is_null())), 0),
        ("count_is_null", P([count_star()], predicate=col("value").is_null()), 0),
        ("count_ts_cmp", P([count_star()], predicate=col("ts") >= 1_700_000_000_000 + 3_600_000), 0),
        ("count_svc_eq", P([count_star()], predicate=col("service") == "svc-01"), 0),
        ("count_svc_lt", P([count_star()], predicate=col("service") < "svc-03"), 0),
        ("count_url_like", P([count_star()], predicate=col("url").like("%google%")), 0),
        ("count_url_not_like", P([count_star()], predicate=col("url").not_like("%google%")), 0),
        ("count_url_like_prefix", P([count_star()], predicate=col("url").like("http://goog%")), 0),
        ("count_url_like_underscore", P([count_star()], predicate=col("url").like("%/g_gle%")), 0),
        ("count_url_like_suffix", P([count_star()], predicate=col("url").like("%admin")), 0),
        ("global_aggs", P([count_star(), count("value"), sum_("value"), min_("value"), max_("value"), avg("value"),
                           sum_("u16"), min_("i16"), max_("ts"), min_("ts"), avg("i32")]), 0),
        ("global_aggs_filtered", P([sum_("value"), max_("i32"), avg("u16")], predicate=col("i16") != 0), 0),
        ("global_sparse_filter", P([sum_("value"), count("i32")], predicate=col("u16") > 64000), 0),
        ("by_service", P([count_star(), sum_("value"), max_("value")], keys=["service"]), 1),
        ("by_minute_service", P([count_star(), sum_("value"), max_("value")],
                                keys=[date_bin(60_000, "ts"), "service"]), 2),
        ("by_minute_service_full", P([count_star(), count("value"), sum_("value"), min_("value"), max_("value"),
                                      avg("value")], keys=[date_bin(60_000, "ts"), "service"]), 2),
        ("by_5s_origin", P([count_star()], keys=[date_bin(5_000, "ts", origin_ms=1_700_000_001_234)]), 1),
        ("by_hour_future_origin", P([count_star(), sum_("i32")],
                                    keys=[date_bin(3_600_000, "ts", origin_ms=1_700_010_000_000)]), 1),
        ("by_user", P([count_star()], keys=["user"]), 1),
        ("by_user_filtered", P([count_star(), sum_("value")], keys=["user"], predicate=col("url").like("%google%")), 1),
        ("by_user_sparse", P([count_star(), max_("value")], keys=["user"], predicate=col("u16") > 62000), 1),
        ("by_i16_i32", P([count_star(), avg("value")], keys=["i16", "i32"]), 2),
        ("by_user_value", P([count_star()], keys=["user", "u16"]), 2),
        ("distinct_user", P([count_distinct("user")]), 0),
        ("distinct_user_by_service", P([count_distinct("user")], keys=["service"]), 1),
        ("distinct_user_by_hour_service", P([count_distinct("user")], keys=[date_bin(3_600_000, "ts"), "service"]), 2),
        ("distinct_service", P([count_distinct("service")], keys=["i16"]), 1),
    ]
    return plans
