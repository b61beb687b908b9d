#!/usr/bin/env python3
"""Headline benchmark: rows/s scanned + aggregated on synthetic log events.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (predicate -> selection -> GROUP BY aggregate ->
final rows on the host) over the HBM-resident partitions of this rank.  At N = 1 the
workload is "Synthetic log events 1B rows: time_bucket(1m) GROUP BY service with
COUNT/SUM/MAX on one MI355X".  At N > 1 every rank holds its own 1B-row share of an
N-times larger table (weak scaling), produces PARTIAL state, the states are
all-gathered with RCCL and rank 0 runs the Final merge.

Rank 0 prints one JSON line.  `roofline` prices the dominant kernel (group_by) by
algorithmic bytes over its HIP-event duration; `cpu_baseline` times the CPU oracle
(a restatement of the reference's semantics, not DataFusion itself) on a bounded
sample of the same table on this host's cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


PCIE_BOUND_GBS = 64.0  # PCIe 5.0 x16, one direction


def stream_feed(args) -> int:
    print(json.dumps(stream_feed_line(args)), flush=True)
    return 0


def stream_feed_line(args) -> dict:
    """End-to-end figure: the native feeder (tests/c/stream_bench.cpp, one thread per partition
    as DataFusion polls them) pushes host Arrow batches through the C ABI; nothing is resident
    beforehand.  The rate is bounded by PCIe, not HBM, and priced against that."""
    import subprocess

    if args.workload not in ("C2", "C3") or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--feed stream measures the C2 or the C3 query on one GPU")
    exe = os.path.join(ROOT, "tests", "c", "stream_bench")
    if not os.path.exists(exe):
        raise SystemExit(f"{exe} is not built (python -c 'import __graft_entry__ as g; g.build()')")
    c3 = args.workload == "C3"
    rows = args.rows or (100_000_000 if c3 else 1_000_000_000)
    plain = bool(getattr(args, "plain_strings", False)) and not c3
    r = subprocess.run([exe, str(rows), str(args.feed_threads), str(args.steps), "8192", "c3" if c3 else "c2plain" if plain else "c2"], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(f"stream_bench failed ({r.returncode}): {r.stderr.strip()}")
    m = json.loads(r.stdout.strip().splitlines()[-1])
    line = {
        "metric": "rows/sec scanned+aggregated (node), end to end from host Arrow batches",
        "value": m["rows_per_s"],
        "unit": "rows/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": 1,
        "ms_per_step": m["seconds_avg"] * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": ("C3: GROUP BY UserID -> COUNT(*) as PARTIAL state per partition, merged with ORDER BY COUNT(*) DESC LIMIT 10 (`groups`: rows of the PARTIAL states); ids log-uniform over 2^24 ranks, "
                                "batches of 8192 rows in pageable host memory") if c3 else
                               ("C2: date_bin(1 minute, timestamp), service -> COUNT(*), SUM(value), MAX(value); "
                                + ("service as plain Utf8 (coded on the device), " if plain else "service dictionary encoded, ")
                                + "batches of 8192 rows in pageable host memory"),
                   "feed": "stream", "total_rows": rows, "partitions": m["threads"], "batch_rows": m["batch_rows"],
                   "groups": m["groups"]},
        "roofline": None,
        "staging": {"bound": "pcie", "host_bytes_per_row": m["host_bytes_per_row"], "achieved": m["host_to_hbm_GBps"],
                    "peak": PCIE_BOUND_GBS, "unit": "GB/s", "frac": m["host_to_hbm_GBps"] / PCIE_BOUND_GBS,
                    "bytes_staged_per_step": m["bytes_staged_last_step"],
                    "staging_thread_ms_per_step": m["staging_thread_ms_last_step"]},
        "cpu_baseline": None,
    }
    return line


DOMINANT = {"C1": "filter_count", "C2": "group_by", "C3": "group_by", "C4": "like_contains", "C4G": "like_contains", "C5": "group_by"}
SYMBOLS = {"group_by": ("k_group_by_stream", "k_part_agg", "k_part_scatter", "k_part_hist", "k_part_offsets",
                        "k_part_plan", "k_part2_hist", "k_part2_scan", "k_part2_scatter", "k_part2_lines", "k_part2_bounds", "k_part2_pieces", "k_spill_runs_lines", "k_part2_agg", "k_part2_agg_compact",
                        "k_group_by_wide", "k_group_by", "k_wc_sample", "k_wc_build", "k_wc_scatter", "k_wc_agg", "k_part2_layout",
                        "k_part2_hist_direct", "k_part2_layout_uniform", "k_spill_count", "k_spill_place", "k_spill_runs"),
           "filter_count": ("k_cmp_leaf",), "like_contains": ("k_contains_find", "k_contains_mark")}


def profile_tag(wl, rows, parts):
    return {("C2", 1_000_000_000, 64): "c2_1b", ("C3", 100_000_000, 100): "c3", ("C1", 1_000_000_000, 64): "c1_1b",
            ("C4", 100_000_000, 100): "c4", ("C5", 1_250_000_000, 64): "c5_1p25b",
            ("C3_17M", 100_000_000, 100): "c3_17m"}.get((wl.name, rows, parts))


def price_dominant_kernel(wl, rows, parts, stats, groups, world):
    """`roofline` of one workload: algorithmic bytes of a launch over the dominant kernel's
    average HIP-event duration (measured live, on the library's stream), against the 8 TB/s
    HBM peak; `traffic` is the committed PMC measurement of the same command (PMC passes
    cannot run inside this process), null when the configuration differs."""
    kname = DOMINANT[wl.name.split("_")[0]]
    ks = stats["kernels"].get(kname)
    if not ks or not ks["launches"]:
        return None
    avg_s = ks["total_ms"] / ks["launches"] / 1e3
    alg = rows * wl.bytes_per_row + (groups * wl.bytes_per_group if world == 1 else 0)
    achieved = alg / avg_s / 1e9
    traffic = traffic_source = profile_avg_ms = None
    tag = profile_tag(wl, rows, parts)
    if tag and world == 1:
        for rnd in ("r3", "r2"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", rnd, f"{tag}_traffic.json")))
                ks_prof = [tj["kernels"][k] for k in SYMBOLS[kname] if k in tj["kernels"] and tj["kernels"][k]["calls"] >= 3]
                if ks_prof:
                    traffic = sum((k["fetch_bytes_per_launch"] or 0) + (k["write_bytes_per_launch"] or 0) for k in ks_prof)
                    profile_avg_ms = sum(k["avg_ns"] for k in ks_prof) / 1e6
                    traffic_source = f"profiles/{rnd}/{tag}_traffic.json (rocprofv3 --pmc passes of the same command, not collected in this run)"
                    break
            except (OSError, KeyError, ValueError, TypeError):
                continue
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kname,
                "kernel_avg_ms": avg_s * 1e3, "algorithmic_bytes_per_launch": alg,
                "frac_uses": "kernel_avg_ms: HIP events around the launch, measured in this run",
                "traffic_source": traffic_source, "profile_kernel_avg_ms": profile_avg_ms}
    try:  # the streaming-read rate a plain sum kernel reaches on this part (committed measurement)
        ceil_gbs = json.load(open(os.path.join(ROOT, "profiles", "r1", "read_ceiling.json")))["read_GBps"]
        roofline["measured_read_ceiling"] = ceil_gbs
        roofline["frac_of_measured_ceiling"] = achieved / ceil_gbs
    except (OSError, KeyError, ValueError):
        pass
    return roofline


def pass_breakdown(ctx, step, stats, steps: int = 3):
    """The timed steps run with events around whole launches only (profiling level 2: an event
    between two passes of a partitioned GROUP BY delays the second pass by some microseconds).
    The per-pass durations come from a few more steps, untimed, with every event on; `stats`
    (of the timed steps) gets those passes' figures added under their names."""
    ctx.reset_stats()
    ctx.set_profiling(1)
    for _ in range(steps):
        step()
    ctx.set_profiling(0)
    detail = ctx.stats()["kernels"]
    for name, v in detail.items():
        have = stats["kernels"].get(name)
        if (have is None or not have["total_ms"]) and v["total_ms"]:
            stats["kernels"][name] = dict(v, breakdown_pass=True)
    return stats


def extra_workloads(ctx, steps: int, warmup: int, sample_rows_cap: int = 16_000_000):
    """The other configurations, a few steps each, for the same record as the headline: per
    workload the step time, the dominant kernel's HIP-event time, the roofline fraction on
    algorithmic bytes, the ratio of measured to algorithmic HBM traffic (from the committed
    profile of the same command) and whether a sample of the result equals the CPU oracle's."""
    import dataclasses

    from oracle import pyoracle
    from prism_amd import workloads
    from prism_amd.engine import Plan
    from tests.helpers import assert_tables_match, assert_topk_match

    c3_17m = dataclasses.replace(workloads.c3(id_bits=26), name="C3_17M")
    todo = [("C1@1e7", workloads.c1(), 10_000_000, 10), ("C1@1e9", workloads.c1(), 1_000_000_000, 64),
            ("C3", workloads.c3(), 100_000_000, 100), ("C3@17.6M groups", c3_17m, 100_000_000, 100),
            ("C4", workloads.c4(), 100_000_000, 100), ("C5@1.25e9", workloads.c5(), 1_250_000_000, 64)]
    out = {}
    for label, wl, rows, parts in todo:
        t_all = time.time()
        entry = {"query": wl.description, "rows": rows, "partitions": parts}
        try:
            table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, parts)
            plan = Plan(ctx, wl.schema, wl.query)
            try:
                # parity on the leading partitions (bounded: the oracle is a scalar restatement)
                k = max(1, min(parts, sample_rows_cap * parts // rows))
                sample = workloads.partition_bounds(rows, parts, k - 1)[1]
                st = pyoracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, sample, rows, n_parts=max(k, 16), threads=16)
                try:
                    want, _ = st.execute(wl.query, threads=min(16, os.cpu_count() or 1))
                finally:
                    st.close()
                want = workloads.decode_dictionary_columns(wl, want)
                got = plan.execute(table, range(k))
                if wl.query.limit is not None:
                    assert_topk_match(got, want, wl.query.order_by)
                else:
                    assert_tables_match(got, want, wl.n_keys)
                entry["parity"] = True
                entry["parity_sample_rows"] = sample
                for _ in range(warmup):
                    plan.execute(table)
                ctx.reset_stats()
                ctx.set_profiling(2)
                t0 = time.perf_counter()
                result = None
                for _ in range(steps):
                    result = plan.execute(table)
                elapsed = time.perf_counter() - t0
                ctx.set_profiling(False)
                stats = pass_breakdown(ctx, lambda: plan.execute(table), ctx.stats())
                roof = price_dominant_kernel(wl, rows, parts, stats, result.num_rows if wl.n_keys else 1, 1)
                entry.update({"steps": steps, "ms_per_step": elapsed / steps * 1e3, "rows_per_s": rows / (elapsed / steps),
                              "groups": result.num_rows})
                if roof:
                    entry.update({"kernel": roof["kernel"], "kernel_ms": roof["kernel_avg_ms"], "frac": roof["frac"],
                                  "algorithmic_bytes": roof["algorithmic_bytes_per_launch"], "traffic": roof["traffic"],
                                  "traffic_ratio": (roof["traffic"] / roof["algorithmic_bytes_per_launch"]) if roof["traffic"] else None,
                                  "traffic_source": roof["traffic_source"], "profile_kernel_avg_ms": roof["profile_kernel_avg_ms"]})
                entry["kernels"] = {n: v["total_ms"] / max(1, v["launches"]) for n, v in stats["kernels"].items()
                                    if not n.startswith("host:")}
            finally:
                plan.close()
                table.close()
        except Exception as e:  # noqa: BLE001 - one workload failing must not take the headline line with it
            entry["parity"] = entry.get("parity", False)
            entry["error"] = f"{type(e).__name__}: {e}"[:400]
        entry["wall_s"] = round(time.time() - t_all, 1)
        log(f"[workloads] {label}: " + json.dumps({k_: v for k_, v in entry.items() if k_ not in ('kernels', 'query')}))
        out[label] = entry
    return out


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher around it: starts N ranks of this script
    (one per GPU) as children through torch.distributed.run and relays rank 0's JSON line.
    This process never touches the GPU: the ranks are fresh processes, nothing is re-executed."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[launcher] starting {n} ranks: {' '.join(cmd)}")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.lstrip().startswith("{") and ('"metric"' in out or '"check_launch"' in out):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        log("[launcher] the ranks printed no result line")
        rc = 1
    return rc


def check_launch(args) -> int:
    """The ranks meet (gloo, CPU) and count themselves; rank 0 prints one line."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([1, rank], dtype=torch.int64)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"check_launch": True, "n_gpus_asked": args.gpus, "ranks_met": int(t[0]), "rank_sum": int(t[1])}), flush=True)
    dist.destroy_process_group()
    return 0


def parquet_feed(args) -> int:
    print(json.dumps(parquet_feed_line(args)), flush=True)
    return 0


def parquet_feed_line(args) -> dict:
    """Loader figure: files shaped like prism-ingest's (data page v2, dictionary with PLAIN
    fallback, Snappy, one file per partition; src/prism-ingest/src/ingest.rs:98-119) sit in host
    memory, as after an object-store GET; prism_gpu_table_append_parquet decodes them (library
    code, host threads) and stages the columns into HBM (dictionary-encoded string chunks as indices + dictionary, expanded
    by kernels); then the C2 query runs over them and is
    checked against the CPU oracle over the same rows.  Bounded by host decode, not by HBM."""
    import io
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np
    import pyarrow as pa
    import pyarrow.parquet as pq

    from prism_amd import workloads
    from prism_amd.engine import Context, Plan, Table

    if args.workload != "C2":
        raise SystemExit("--feed parquet measures the C2 query")
    n_files = args.partitions or 16
    rows = args.rows or 32_000_000
    per = rows // n_files
    rows = per * n_files
    wl = workloads.c2()
    names = wl.dictionaries["service"]
    rng = np.random.default_rng(7)
    w = 1.0 / np.arange(1, len(names) + 1) ** 1.1
    w /= w.sum()
    files, tables = [], []
    t0 = time.time()
    for f in range(n_files):
        ts = workloads.T0_MS + (np.arange(f * per, (f + 1) * per, dtype=np.int64) * workloads.DAY_MS) // rows
        codes = rng.choice(len(names), size=per, p=w).astype(np.int32)
        svc = pa.DictionaryArray.from_arrays(pa.array(codes), pa.array(names)).cast(pa.string())
        t = pa.table({"timestamp": pa.array(ts, pa.timestamp("ms")), "service": svc,
                      "value": pa.array(rng.integers(0, 50_000, per, dtype=np.int64))})
        sink = io.BytesIO()
        pq.write_table(t, sink, version="2.6", data_page_version="2.0", compression="snappy", use_dictionary=True)
        files.append(sink.getvalue())
        tables.append(t)
    file_bytes = sum(len(b) for b in files)
    decoded_bytes = sum(t.nbytes for t in tables)
    log(f"[parquet] wrote {n_files} files, {file_bytes / 1e6:.0f} MB for {rows:,} rows ({decoded_bytes / 1e6:.0f} MB decoded) in {time.time() - t0:.1f}s")

    ctx = Context(0)
    ctx.set_option("parquet_threads", args.parquet_threads)
    plan = Plan(ctx, wl.schema, wl.query)
    threads = max(1, min(args.feed_threads, n_files))

    pool = ThreadPoolExecutor(threads)  # (the caller's threads exist before the files arrive; the call into the library releases the interpreter lock)

    def load_all():
        table = Table(ctx, wl.schema)
        list(pool.map(lambda f: table.append_parquet(f, files[f]), range(n_files)))
        return table

    for _ in range(max(1, args.warmup)):
        load_all().close()  # warm-up: pinned ring, pools of page-locked blocks
    load_s, query_s, result = [], [], None
    locked_ms = None
    for _ in range(max(1, args.steps)):
        ctx.reset_stats()
        t0 = time.perf_counter()
        table = load_all()
        t1 = time.perf_counter()
        k = ctx.stats()["kernels"]
        locked_ms = k["host:append_locked"]["total_ms"] if "host:append_locked" in k else None
        host_ms = {n: (round(v["total_ms"], 3), v["launches"]) for n, v in k.items() if n.startswith("host:")}
        result = plan.execute(table)
        t2 = time.perf_counter()
        load_s.append(t1 - t0)
        query_s.append(t2 - t1)
        table.close()
    from oracle import pyoracle
    from tests.helpers import assert_tables_match

    want = pyoracle.execute(pa.concat_tables(tables), wl.query)
    assert_tables_match(result, want, wl.n_keys)
    plan.close()
    ctx.close()
    pool.shutdown()
    load = sorted(load_s)[len(load_s) // 2]
    log(f"[parquet] host timers of the last load (ms, calls): {host_ms}")
    line = {
        "metric": "rows/sec Parquet files in host memory -> HBM-resident columns (pages decoded on host threads; dictionary indices stay RLE / bit-packed and are expanded, with their strings, on the device)",
        "value": rows / load, "unit": "rows/s", "n_gpus": 1, "steps": len(load_s), "warmup": 1, "ms_per_step": load * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": "C2 schema (timestamp ms, service Utf8 dictionary-encoded, value Int64) as prism-ingest writes it: "
                               "data page v2, dictionary, Snappy, one file per partition", "feed": "parquet", "files": n_files,
                   "total_rows": rows, "file_bytes": file_bytes, "decoded_bytes": decoded_bytes, "caller_threads": threads,
                   "threads_per_file": args.parquet_threads},
        "roofline": None,
        "loader": {"bound": "host decode", "parquet_GBps": file_bytes / load / 1e9, "decoded_GBps": decoded_bytes / load / 1e9,
                   "load_ms": load * 1e3, "append_locked_ms_sum": locked_ms, "query_ms_after_load": sorted(query_s)[len(query_s) // 2] * 1e3,
                   "parity": "C2 result over the loaded rows equals the CPU oracle's over the same rows"},
        "cpu_baseline": None,
    }
    return line


def extra_feeds(args) -> dict:
    """The two ways rows reach HBM from outside, a few steps each, for the default run's record:
    host Arrow batches streamed through the C ABI (C2 at 10^9 rows, PCIe-bound, with the service
    column dictionary encoded and as plain Utf8; C3 at 10^8 rows as PARTIAL states merged) and
    Parquet files decoded by the library (C2-shaped, 32e6 rows).  A feed that fails is recorded as such."""
    import copy

    out = {}
    for name, fn, over in (("stream_C2", stream_feed_line, {"feed": "stream", "steps": 3, "rows": 0}),
                           ("stream_C2_plain_utf8", stream_feed_line, {"feed": "stream", "steps": 2, "rows": 0, "plain_strings": True}),
                           ("stream_C3", stream_feed_line, {"feed": "stream", "workload": "C3", "steps": 5, "rows": 0}),
                           ("parquet_C2", parquet_feed_line, {"feed": "parquet", "steps": 5, "warmup": 2, "rows": 0, "partitions": 0})):
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        t0 = time.time()
        try:
            line = fn(a)
            out[name] = {k: line[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "config") if k in line}
            for k in ("staging", "loader"):
                if k in line:
                    out[name][k] = line[k]
        except (Exception, SystemExit) as e:  # noqa: BLE001  (the headline must still be printed)
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["wall_s"] = round(time.time() - t0, 1)
        log(f"[feeds] {name}: {json.dumps(out[name])[:600]}")
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=["C1", "C2", "C3", "C4", "C4G", "C5"])
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the workload's)")
    ap.add_argument("--id-bits", type=int, default=0,
                    help="C3: ranks of the id generator are drawn from [1, 2^id_bits) (default 24: 6e6 groups in 1e8 rows; "
                         "26: 1.76e7 groups, ClickBench's UserID cardinality)")
    ap.add_argument("--partitions", type=int, default=0, help="partitions per GPU")
    ap.add_argument("--cpu-sample-partitions", type=int, default=8,
                    help="leading partitions of rank 0 the CPU baseline and the parity check run on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="auto", choices=["auto", "state", "arrow"],
                    help="merge between GPUs: device-resident group tables (state) or PARTIAL Arrow batches (arrow); "
                         "auto = state when the plan allows it.  With one GPU, state/arrow run the same code in a "
                         "one-rank group (to measure the exchange's overhead)")
    ap.add_argument("--parquet-threads", type=int, default=1, help="--feed parquet: threads decoding the column chunks of one file")
    ap.add_argument("--feed", default="resident", choices=["resident", "stream", "parquet"],
                    help="resident: partitions already in HBM when the timed region starts (the headline value). "
                         "stream: host Arrow batches of 8192 rows pushed through prism_gpu_push_batch on one thread "
                         "per partition, timed from the first push to the merged result (the end-to-end figure; "
                         "bounded by PCIe, reported separately, C2 only)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="prism_gpu_set_option on the context (for A/B runs of a kernel path)")
    ap.add_argument("--plain-strings", action="store_true", help="--feed stream, C2: the service column arrives as plain Utf8 instead of dictionary encoded")
    ap.add_argument("--feed-threads", type=int, default=16, help="partitions (= feeding threads) of --feed stream")
    ap.add_argument("--check-launch", action="store_true",
                    help="only bring the ranks up (gloo rendezvous + one all-reduce, no GPU) and print what met; "
                         "lets a box without GPUs check that --gpus N starts N ranks")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the `workloads` and `feeds` objects (C1, C3, C4, C5 and the two feeds run for a few steps after the headline's timed region)")
    args = ap.parse_args()
    if args.feed == "stream":
        return stream_feed(args)
    if args.feed == "parquet":
        return parquet_feed(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    if args.check_launch:
        return check_launch(args)

    # Exactly one line goes to stdout: RCCL prints a version banner there when the
    # communicator is created, so everything else written to fd 1 is sent to stderr and the
    # JSON line is written to the saved descriptor.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    grouped = world > 1 or args.exchange != "auto"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from prism_amd import _cabi as abi
    from prism_amd import workloads
    from prism_amd.distributed import all_gather_tables
    from prism_amd.engine import Comm, Context, Plan, PrismGpuError

    wl = workloads.WORKLOADS[args.workload]()
    if args.workload == "C3" and args.id_bits:
        import dataclasses

        wl = dataclasses.replace(workloads.c3(id_bits=args.id_bits), name="C3_17M" if args.id_bits == 26 else "C3")
    rows = args.rows or wl.default_rows
    parts = args.partitions or wl.default_partitions
    total_rows = rows * world
    first_row = rows * rank

    ctx = Context(local_rank)
    for opt in args.option:
        name, _, value = opt.partition("=")
        ctx.set_option(name, int(value))
    t0 = time.time()
    table = workloads.build_gpu_table(ctx, wl, first_row, rows, total_rows, parts)
    log(f"[rank {rank}] generated {rows:,} rows in {parts} partitions in {time.time() - t0:.1f}s")

    single = Plan(ctx, wl.schema, wl.query)
    partial = Plan(ctx, wl.schema, wl.query.with_mode(abi.MODE_PARTIAL)) if grouped else None

    # How the per-GPU aggregates meet: on the group tables in HBM when the plan allows it
    # (every rank takes the same decision: it depends on the plan only), else as PARTIAL
    # Arrow batches.
    exchange = "none (single GPU)"
    comm = None
    if grouped:
        exchange = args.exchange
        fallback_reason = None
        if exchange in ("auto", "state"):
            # the library's own RCCL communicator; torch.distributed only hands the id around
            try:
                comm = Comm.from_process_group(ctx)
                single.execute_sharded(comm, table, None, 0)
                exchange = "state"
            except PrismGpuError as e:
                if args.exchange == "state":
                    raise
                fallback_reason = str(e)
                exchange = "arrow"
            # every rank takes the same path: one that could not run the exchange in the library
            # (an unsupported plan fails on all of them alike; a transport error may not) moves
            # all of them to the PARTIAL Arrow exchange
            ok = torch.tensor([1 if exchange == "state" else 0], dtype=torch.int32, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                exchange = "arrow"
        log(f"[rank {rank}] exchange between GPUs: {exchange}" + (f" (library exchange not used: {fallback_reason})" if fallback_reason else ""))

    def step():
        if not grouped:
            return single.execute(table)
        if exchange == "state":
            return single.execute_sharded(comm, table, None, 0)  # no torch collective in here
        state = partial.execute(table)
        states = all_gather_tables(state, device)
        return partial.merge(states) if rank == 0 else None

    # ---- parity check + CPU baseline on a bounded sample (rank 0) ----
    cpu_baseline = None
    k = max(1, min(args.cpu_sample_partitions, parts))
    sample_rows = workloads.partition_bounds(rows, parts, k - 1)[1]
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import pyoracle
        from tests.helpers import assert_tables_match, assert_topk_match

        threads = max(1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)))
        threads = min(threads, os.cpu_count() or 1)
        t0 = time.time()
        # the shape DataFusion runs: the sample's partitions handed to a pool of threads, a Partial
        # aggregate per thread, then the Final fold (oracle/prism_oracle.cpp: run_threads)
        cpu_parts = max(k, threads * 4)
        st = pyoracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, first_row, sample_rows, total_rows,
                                 n_parts=cpu_parts, threads=threads)
        log(f"[cpu] generated {sample_rows:,} sample rows in {time.time() - t0:.1f}s")
        want, ms = st.execute(wl.query, threads=threads)
        st.close()
        # one core on a slice of the sample, for the per-core rate
        one_rows = max(1, min(sample_rows, max(sample_rows // 16, 2_000_000)))
        st1 = pyoracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, first_row, one_rows, total_rows, n_parts=1, threads=threads)
        _, ms1 = st1.execute(wl.query, threads=1)
        st1.close()
        want = workloads.decode_dictionary_columns(wl, want)
        got = single.execute(table, range(k))
        if wl.query.limit is not None:
            assert_topk_match(got, want, wl.query.order_by)
        else:
            assert_tables_match(got, want, wl.n_keys)
        log(f"[parity] GPU == CPU oracle on the first {k} partitions ({sample_rows:,} rows, {want.num_rows} groups)")
        cpu_baseline = {
            "value": sample_rows / (ms / 1e3),
            "unit": "rows/s",
            "cores": threads,
            "kind": "port",
            "sample": f"first {sample_rows} rows ({k} of {parts} partitions) of the same table in {cpu_parts} partitions handed to a "
                      f"pool of {threads} threads (Partial aggregate per thread, then the Final fold), {ms:.0f} ms; "
                      f"CPU restatement of the reference's semantics (not DataFusion 32)",
            "cores_available": os.cpu_count(),
            "one_core": {"value": one_rows / (ms1 / 1e3), "unit": "rows/s", "rows": one_rows, "ms": ms1},
        }

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.reset_stats()
    ctx.set_profiling(2)
    barrier()
    t0 = time.perf_counter()
    result = None
    for _ in range(args.steps):
        result = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.set_profiling(False)
    if grouped:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    stats = ctx.stats()
    if not grouped:
        stats = pass_breakdown(ctx, step, stats)
    if rank == 0:
        # full-size, size-independent checks: every row lands in exactly one group
        if args.workload in ("C2",):
            assert sum(result.column("COUNT(*)").to_pylist()) == total_rows, "COUNT(*) does not add up to the rows scanned"
        roofline = price_dominant_kernel(wl, rows, parts, stats, result.num_rows if wl.n_keys else 1, world)
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "rows/sec scanned+aggregated (node)",
            "value": total_rows / (elapsed / args.steps),
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": f"{wl.name}: {wl.description}", "rows_per_gpu": rows, "partitions_per_gpu": parts,
                       "total_rows": total_rows, "groups": result.num_rows,
                       "merge": {"none (single GPU)": "none (single GPU)",
                                 "state": "prism_gpu_execute_sharded: group tables stay in HBM, agreed key packing, the "
                                          "library's RCCL reduce of the dense accumulator arrays (or hash-partitioned "
                                          "all-to-all + merge of hash-table rows) onto rank 0",
                                 "arrow": "RCCL all-gather of PARTIAL Arrow batches + Final merge on rank 0"}[exchange]},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "exchange": comm.stats() if comm is not None and exchange == "state" else None,
            # (entries with `breakdown_pass`: single passes of a composite launch, from a few untimed steps after the timed ones)
            "kernels": {n: dict({"launches": v["launches"], "avg_ms": v["total_ms"] / max(1, v["launches"])},
                                **({"breakdown_pass": True} if v.get("breakdown_pass") else {}))
                        for n, v in stats["kernels"].items()},
        }
        if world == 1 and not grouped and args.workload == "C2" and not args.rows and not args.no_extra_workloads:
            # the other configurations, after the headline's timed region and with its table gone
            single.close()
            table.close()
            single = table = None
            line["workloads"] = extra_workloads(ctx, steps=5, warmup=3)
            line["feeds"] = extra_feeds(args)
        print(json.dumps(line), file=json_out, flush=True)

    if comm is not None:
        comm.close()
    if single is not None:
        single.close()
    if partial:
        partial.close()
    if table is not None:
        table.close()
    ctx.close()
    if grouped:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
