"""Parquet -> HBM: files written the way prism's ingest writes them (data page v2,
dictionary encoding, Snappy; src/prism-ingest/src/ingest.rs:98-119), one file per
partition, loaded through prism_amd.loader and queried; answers must equal the CPU
oracle's over the same rows."""
import os

import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from tests import helpers
from tests.helpers import assert_tables_match

pytestmark = pytest.mark.gpu

from prism_amd.plan import QueryPlan, avg, col, count_star, date_bin, max_, sum_  # noqa: E402


def test_parquet_partitions_round_trip(gpu_ctx, tmp_path):
    from oracle import pyoracle
    from prism_amd.engine import Plan, Table
    from prism_amd.loader import load_parquet

    t = helpers.random_table(30_000, seed=21, nulls=True)
    n_files = 3
    paths = []
    for i in range(n_files):
        lo, hi = t.num_rows * i // n_files, t.num_rows * (i + 1) // n_files
        p = os.path.join(tmp_path, f"part-{i}.parquet")
        pq.write_table(t.slice(lo, hi - lo), p, version="2.6", data_page_version="2.0", compression="snappy",
                       use_dictionary=True, row_group_size=4000)
        paths.append(p)
    g = Table(gpu_ctx, t.schema)
    try:
        assert load_parquet(g, paths, batch_rows=3000) == t.num_rows
        assert g.num_partitions == n_files and g.num_rows == t.num_rows
        for q, n_keys in [
            (QueryPlan([count_star(), sum_("value"), max_("u16"), avg("i32")], keys=[date_bin(600_000, "ts"), "service"]), 2),
            (QueryPlan([count_star()], predicate=col("url").like("%google%") & (col("i16") != 0)), 0),
            (QueryPlan([count_star()], keys=["user"], predicate=col("service") == "svc-01"), 1),
        ]:
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                got = plan.execute(g)
                one = plan.execute(g, [1])
            finally:
                plan.close()
            assert_tables_match(got, pyoracle.execute(t, q), n_keys)
            lo, hi = t.num_rows // n_files, t.num_rows * 2 // n_files
            assert_tables_match(one, pyoracle.execute(t.slice(lo, hi - lo), q), n_keys)
    finally:
        g.close()


def test_projection_leaves_other_columns_null(gpu_ctx, tmp_path):
    from oracle import pyoracle
    from prism_amd.engine import Plan, Table
    from prism_amd.loader import load_parquet

    t = helpers.random_table(5_000, seed=22, nulls=False)
    p = os.path.join(tmp_path, "one.parquet")
    pq.write_table(t, p, compression="snappy")
    g = Table(gpu_ctx, t.schema)
    try:
        load_parquet(g, [p], columns=["ts", "value"])
        q = QueryPlan([count_star(), sum_("value")], keys=[date_bin(3_600_000, "ts")])
        plan = Plan(gpu_ctx, t.schema, q)
        try:
            got = plan.execute(g)
        finally:
            plan.close()
        assert_tables_match(got, pyoracle.execute(t, q), 1)
        q2 = QueryPlan([count_star()], predicate=col("i32").is_null())
        plan = Plan(gpu_ctx, t.schema, q2)
        try:
            assert plan.execute(g).column(0).to_pylist() == [t.num_rows]
        finally:
            plan.close()
    finally:
        g.close()


def test_library_decodes_the_parquet_files_itself(gpu_ctx):
    """prism_gpu_table_append_parquet: the same files, decoded by the library's own reader
    (no pyarrow between the bytes and HBM), one file per partition, every row group a chunk."""
    import io

    from oracle import pyoracle
    from prism_amd.engine import Plan, PrismGpuError, Table

    t = helpers.random_table(30_000, seed=23, nulls=True)
    n_files = 3
    g = Table(gpu_ctx, t.schema)
    try:
        for i in range(n_files):
            lo, hi = t.num_rows * i // n_files, t.num_rows * (i + 1) // n_files
            sink = io.BytesIO()
            # (the second file's dictionaries overflow early: its string chunks start dictionary encoded and go
            # on PLAIN, which the loader must expand mid-chunk; the others stay dictionary encoded throughout)
            extra = {"dictionary_pagesize_limit": 2048, "data_page_size": 4096} if i == 1 else {}
            pq.write_table(t.slice(lo, hi - lo), sink, version="2.6", data_page_version="2.0", compression="snappy",
                           use_dictionary=True, row_group_size=4000, **extra)
            g.append_parquet(i, sink.getvalue())
        assert g.num_partitions == n_files and g.num_rows == t.num_rows
        for q, n_keys in [
            (QueryPlan([count_star(), sum_("value"), max_("u16"), avg("i32")], keys=[date_bin(600_000, "ts"), "service"]), 2),
            (QueryPlan([count_star()], predicate=col("url").like("%google%") & (col("i16") != 0)), 0),
            (QueryPlan([count_star()], keys=["user"], predicate=col("service") == "svc-01"), 1),
        ]:
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                got = plan.execute(g)
                one = plan.execute(g, [1])
            finally:
                plan.close()
            assert_tables_match(got, pyoracle.execute(t, q), n_keys)
            lo, hi = t.num_rows // n_files, t.num_rows * 2 // n_files
            assert_tables_match(one, pyoracle.execute(t.slice(lo, hi - lo), q), n_keys)
    finally:
        g.close()

    # a file without some of the table's columns: they read as NULLs; a column of another type is refused
    some = t.select(["ts", "value"]).slice(0, 5000)
    sink = io.BytesIO()
    pq.write_table(some, sink, data_page_version="2.0", compression="snappy")
    g = Table(gpu_ctx, t.schema)
    try:
        g.append_parquet(0, sink.getvalue())
        q = QueryPlan([count_star(), sum_("value")], keys=[date_bin(3_600_000, "ts")])
        plan = Plan(gpu_ctx, t.schema, q)
        try:
            assert_tables_match(plan.execute(g), pyoracle.execute(t.slice(0, 5000), q), 1)
        finally:
            plan.close()
        plan = Plan(gpu_ctx, t.schema, QueryPlan([count_star()], predicate=col("i32").is_null()))
        try:
            assert plan.execute(g).column(0).to_pylist() == [5000]
        finally:
            plan.close()
        wrong = io.BytesIO()
        pq.write_table(pa.table({"value": pa.array(["a", "b"])}), wrong)
        with pytest.raises(PrismGpuError) as e:
            g.append_parquet(1, wrong.getvalue())
        assert e.value.code == -2 and "`value`" in e.value.msg
    finally:
        g.close()


def test_a_batch_of_more_rows_than_a_chunk_holds_becomes_several_chunks(gpu_ctx):
    """prism_gpu_table_append takes a RecordBatch of any length (DataFusion's are small, a
    caller that concatenates a partition first is not): beyond 2^28 rows it is cut into chunks
    of the same partition.  COUNT / SUM / MIN / MAX and a GROUP BY equal numpy's."""
    import numpy as np

    from prism_amd.engine import Plan, Table
    from prism_amd.plan import min_

    n = (1 << 28) + 12_345
    v = (np.arange(n, dtype=np.int64) % 977).astype(np.int16)
    t = pa.table({"v": pa.array(v)})
    g = Table(gpu_ctx, t.schema)
    try:
        g.append_table(t, 1)
        del t
        assert g.num_rows == n and g.num_partitions == 1
        plan = Plan(gpu_ctx, g.schema, QueryPlan([count_star(), sum_("v"), min_("v"), max_("v")]))
        try:
            got = plan.execute(g).to_pylist()[0]
        finally:
            plan.close()
        assert list(got.values()) == [n, int(v.astype(np.int64).sum()), 0, 976]
        plan = Plan(gpu_ctx, g.schema, QueryPlan([count_star()], keys=["v"]))
        try:
            by_v = plan.execute(g)
        finally:
            plan.close()
        counts = dict(zip(by_v.column("v").to_pylist(), by_v.column("COUNT(*)").to_pylist()))
        want = np.bincount(v.astype(np.int64), minlength=977)
        assert counts == {i: int(c) for i, c in enumerate(want)}
    finally:
        g.close()


@pytest.mark.parametrize("nulls", [True, False])
@pytest.mark.parametrize("case", sorted(__import__("tests.test_parquet_reader", fromlist=["CASES"]).CASES))
def test_every_writer_configuration_lands_in_hbm_as_the_host_reader_decodes_it(gpu_ctx, case, nulls):
    """The writer configurations the host reader is checked on (data pages v1 / v2, with and
    without dictionary and Snappy, several row groups, dictionaries that overflow into PLAIN
    pages) through prism_gpu_table_append_parquet: string chunks that are dictionary encoded
    reach HBM as indices and are expanded by kernels, the others as decoded bytes; queries that
    read the codes (GROUP BY, =), the raw bytes (LIKE on the column without a dictionary) and
    the validity must equal the oracle's over prism_gpu_parquet_read's decode of the same file."""
    from oracle import pyoracle
    from prism_amd.engine import Plan, Table, parquet_read
    from tests.test_parquet_reader import CASES, _table, _write

    data = _write(_table(7000, seed=len(case), nulls=nulls), **CASES[case])
    t = parquet_read(data)
    g = Table(gpu_ctx, t.schema)
    try:
        gpu_ctx.reset_stats()
        g.append_parquet(0, data)
        assert g.num_rows == t.num_rows
        # chunks without NULLs that are dictionary encoded throughout leave the reader as the
        # file's RLE / bit-packed runs: a kernel makes the indices
        expanded = gpu_ctx.stats()["kernels"].get("rle_expand", {"launches": 0})["launches"]
        if case in ("like_prism_ingest", "v1_snappy_dict", "v2_row_groups") and not nulls:
            assert expanded > 0
        if not CASES[case]["use_dictionary"]:
            assert expanded == 0
        for q, n_keys in [
            (QueryPlan([count_star(), sum_("small64"), max_("i32")], keys=["method"]), 1),
            (QueryPlan([count_star()], predicate=col("url").like("%host7%") & (col("method") != "GET")), 0),
            (QueryPlan([count_star(), sum_("i16")], keys=["method"], predicate=col("url").like("http://host1%") | col("blob").is_null()), 1),
            (QueryPlan([count_star()], predicate=col("method").is_null() | col("url").is_null() | col("all_null").is_null()), 0),
        ]:
            plan = Plan(gpu_ctx, t.schema, q)
            try:
                got = plan.execute(g)
            finally:
                plan.close()
            assert_tables_match(got, pyoracle.execute(t, q), n_keys)
    finally:
        g.close()


def test_packed_indices_beyond_the_dictionary_are_refused(gpu_ctx):
    """A bit-packed dictionary index that points past the dictionary: the host reader refuses
    the file, and so does the loader, whose kernel expands the runs (the range check is the
    kernel's there)."""
    import io

    from prism_amd.engine import PrismGpuError, Table, parquet_read

    n = 3000
    t = pa.table({"s": pa.array([["a", "bb", "ccc"][i % 3] for i in range(n)]), "v": pa.array(range(n), pa.int64())})
    sink = io.BytesIO()
    pq.write_table(t, sink, compression="none", use_dictionary=True, data_page_version="1.0", write_statistics=False)
    data = bytearray(sink.getvalue())
    # indices 0,1,2,0,1,2,.. at two bits each, least significant first: 0x24 0x49 0x92 over and over
    at = bytes(data).find(b"\x24\x49\x92" * 4)
    assert at > 0
    g = Table(gpu_ctx, t.schema)
    try:
        g.append_parquet(0, bytes(data))
        assert g.num_rows == n
        data[at + 300] |= 0x03  # (the fourth entry of a dictionary of three)
        with pytest.raises(PrismGpuError) as e:
            parquet_read(bytes(data))
        assert "index" in e.value.msg
        with pytest.raises(PrismGpuError) as e:
            g.append_parquet(1, bytes(data))
        assert e.value.code == -1 and "index out of range" in e.value.msg
        assert g.num_rows == n and g.num_partitions == 1
    finally:
        g.close()


def test_hostile_run_headers_of_packed_indices_are_refused(gpu_ctx):
    """The loader keeps a string chunk's dictionary indices as the file's RLE / bit-packed runs
    (no NULLs in the chunk): a run header whose group count wraps the byte count, and a run
    that claims bytes beyond its page, are EINVAL there as they are in the host reader."""
    import numpy as np

    from prism_amd.engine import PrismGpuError, Table
    from tests.test_parquet_reader import _first_data_page, _page_header_fields, _write

    rng = np.random.default_rng(5)
    t = pa.table({"s": pa.array([f"v{int(i):03d}" for i in rng.integers(0, 200, 5000)])},
                 schema=pa.schema([pa.field("s", pa.string(), nullable=False)]))
    good = _write(t, data_page_version="1.0", compression="none", use_dictionary=True)
    _, body = _page_header_fields(good, _first_data_page(good))

    def with_header(width, header):
        data = bytearray(good)
        data[body] = width
        p = body + 1
        while True:
            b = header & 0x7F
            header >>= 7
            data[p] = b | (0x80 if header else 0)
            p += 1
            if not header:
                return bytes(data)

    g = Table(gpu_ctx, t.schema)
    try:
        g.append_parquet(0, good)
        assert g.num_rows == 5000
        for bad in (with_header(32, ((1 << 59) << 1) | 1),   # 2^59 groups of 32 bits: the byte count wraps to zero
                    with_header(8, ((1 << 20) << 1) | 1),    # a bit-packed run far longer than the page
                    with_header(8, 0 << 1)):                 # an empty RLE run
            with pytest.raises(PrismGpuError) as e:
                g.append_parquet(1, bad)
            assert e.value.code == -1, e.value.msg
        assert g.num_rows == 5000 and g.num_partitions == 1
    finally:
        g.close()


@pytest.mark.gpu
def test_column_buffers_of_middling_size_survive_a_refused_append(gpu_ctx):
    """Column buffers between a few hundred KiB and tens of MiB are cut from slabs the table
    owns; an append that is refused half way (its string column's indices point past the
    dictionary, after the fixed-width columns have their buffers) gives its buffers back, and the
    appends before and after it still answer as the oracle does."""
    import io

    import numpy as np

    from oracle import pyoracle
    from prism_amd.engine import Plan, PrismGpuError, Table

    rng = np.random.default_rng(11)
    names = [f"svc-{i:02d}" for i in range(23)]
    n = 300_000

    def part(k):
        codes = rng.integers(0, len(names), n).astype(np.int32)
        return pa.table({"timestamp": pa.array(1_700_000_000_000 + (np.arange(n, dtype=np.int64) + k * n) * 7, pa.timestamp("ms")),
                         "service": pa.DictionaryArray.from_arrays(pa.array(codes), pa.array(names)).cast(pa.string()),
                         "value": pa.array(rng.integers(-1000, 1000, n, dtype=np.int64))})

    def file_of(t):
        sink = io.BytesIO()
        pq.write_table(t, sink, compression="none", use_dictionary=True, data_page_version="1.0", write_statistics=False)
        return bytearray(sink.getvalue())

    query = QueryPlan([count_star(), sum_("value"), max_("timestamp")], keys=["service"])
    good = [part(k) for k in range(5)]
    schema = pa.schema([("timestamp", pa.timestamp("ms")), ("service", pa.string()), ("value", pa.int64())])
    g = Table(gpu_ctx, schema)
    plan = Plan(gpu_ctx, schema, query)
    try:
        for k in range(3):
            g.append_parquet(k, bytes(file_of(good[k])))
        bad = file_of(part(9))
        # 23 names, five bits an index: bytes of ones inside the service column's packed indices are indices of 31
        chunk = pq.ParquetFile(io.BytesIO(bytes(bad))).metadata.row_group(0).column(1)
        assert chunk.path_in_schema == "service"
        for pos in range(chunk.data_page_offset + 2000, chunk.data_page_offset + 2005):
            bad[pos] = 0xFF
        with pytest.raises(PrismGpuError) as e:
            g.append_parquet(7, bytes(bad))
        assert "index out of range" in e.value.msg
        assert g.num_rows == 3 * n
        for k in range(3, 5):
            g.append_parquet(k, bytes(file_of(good[k])))
        got = plan.execute(g)
        want = pyoracle.execute(pa.concat_tables(good), query)
        assert_tables_match(got, want, 1)
    finally:
        plan.close()
        g.close()


@pytest.mark.gpu
def test_plain_strings_are_coded_on_the_device_once_the_dictionary_has_them():
    """Plain Utf8 chunks of a low-cardinality column: a sample of the first chunk's rows seeds the
    table's dictionary on the host; a chunk that brings nothing new gets its codes from a kernel that
    looks every row up in the dictionary; a chunk with a new string is noticed by the kernel and coded on the
    host (after misses in a row the next appends are the host's without a try).  Codes are the same whichever way they
    come about: GROUP BY, = and IN over the column equal the oracle's, NULLs and empty strings
    included."""
    import numpy as np

    from oracle import pyoracle
    from prism_amd.engine import Context, Plan, Table

    rng = np.random.default_rng(31)
    names = ["", "a", "svc-one", "svc-two", "a-name-longer-than-eight-bytes", "ünï©ode", "svc-onf"]

    def chunk(n, pool, nulls=True):
        idx = rng.integers(0, len(pool), n)
        mask = (rng.random(n) < 0.1) if nulls else None
        return pa.table({"service": pa.array([pool[int(i)] for i in idx], pa.string(), mask=mask),
                         "value": pa.array(rng.integers(-100, 100, n), pa.int64())})

    ctx = Context(0)
    ctx.set_option("dict_lookup_min_rows", 0)
    chunks = [chunk(5000, names), chunk(7000, names), chunk(3000, names, nulls=False), chunk(4000, names + ["brand-new"]),
              chunk(2000, names), chunk(2000, names + ["x1"]), chunk(2000, names + ["x2"]), chunk(1000, names), chunk(1000, names),
              chunk(1000, names)]
    g = Table(ctx, chunks[0].schema)
    try:
        launches = []
        for k, c in enumerate(chunks):
            ctx.reset_stats()
            g.append(k, c.combine_chunks().to_batches()[0])
            stat = ctx.stats()["kernels"].get("dict_lookup")
            launches.append(stat["launches"] if stat else 0)
        # (a sample of the first chunk's rows seeds the dictionary;) the fourth chunk misses and is coded on the
        # host; the sixth and the seventh miss in a row, and the next two appends are the host's without a try
        assert launches == [1, 1, 1, 1, 1, 1, 1, 0, 0, 1], launches
        whole = pa.concat_tables(chunks)
        for q, n_keys in [(QueryPlan([count_star(), sum_("value")], keys=["service"]), 1),
                          (QueryPlan([count_star()], predicate=col("service") == "svc-one"), 0),
                          (QueryPlan([count_star()], predicate=col("service") == "brand-new"), 0),
                          (QueryPlan([count_star(), max_("value")], keys=["service"], predicate=col("service") != ""), 1)]:
            plan = Plan(ctx, whole.schema, q)
            try:
                assert_tables_match(plan.execute(g), pyoracle.execute(whole, q), n_keys)
            finally:
                plan.close()
    finally:
        g.close()
        ctx.close()
