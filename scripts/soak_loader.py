#!/usr/bin/env python3
"""Loads the same Parquet files into a fresh table again and again (threads of the caller, as
bench.py --feed parquet does) and reports free HBM and host RSS before and after: a leak in the
table's slabs, the pool of page-locked blocks or the staging shows as a drift.
usage: python scripts/soak_loader.py [rounds]   (needs a GPU)"""
import io
import os
import resource
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pyarrow as pa  # noqa: E402
import pyarrow.parquet as pq  # noqa: E402
import torch  # noqa: E402

from prism_amd.engine import Context, Table  # noqa: E402


def main() -> int:
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(3)
    names = [f"svc-{i:02d}" for i in range(40)]
    files = []
    for f in range(8):
        n = 1_000_000
        t = pa.table({"timestamp": pa.array(1_700_000_000_000 + np.arange(n, dtype=np.int64) * 3 + f, pa.timestamp("ms")),
                      "service": pa.array([names[i] for i in rng.integers(0, len(names), n)]),
                      "value": pa.array(rng.integers(0, 50_000, n), pa.int64())})
        sink = io.BytesIO()
        pq.write_table(t, sink, version="2.6", data_page_version="2.0", compression="snappy", use_dictionary=True)
        files.append(sink.getvalue())
    schema = pa.schema([("timestamp", pa.timestamp("ms")), ("service", pa.string()), ("value", pa.int64())])
    ctx = Context(0)
    pool = ThreadPoolExecutor(8)

    def snapshot():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0] >> 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10

    def load():
        table = Table(ctx, schema)
        list(pool.map(lambda k: table.append_parquet(k, files[k]), range(len(files))))
        rows = table.num_rows
        table.close()
        return rows

    for _ in range(3):
        load()
    free0, rss0 = snapshot()
    for _ in range(rounds):
        assert load() == 8_000_000
    free1, rss1 = snapshot()
    print(f"[soak_loader] {rounds} loads of 8 files: free HBM {free0} -> {free1} MiB, max RSS {rss0} -> {rss1} MiB", flush=True)
    ctx.close()
    return 0 if abs(free0 - free1) <= 64 and rss1 - rss0 <= 256 else 1


if __name__ == "__main__":
    sys.exit(main())
