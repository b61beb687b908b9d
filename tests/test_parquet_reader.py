"""The library's own Parquet decode (prism_gpu_parquet_read, host only) against pyarrow's, on
files written the way prism-ingest writes them (src/prism-ingest/src/ingest.rs:98-101: data
page v2, dictionary with PLAIN fallback, Snappy) and on the other shapes the reader claims:
data page v1, no dictionary, no compression, several row groups, required columns, all-NULL
and empty columns, dictionary overflow into PLAIN pages.  Bit-exact."""
import io

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from prism_amd import _cabi as abi
from prism_amd import engine


def _table(n: int, seed: int, nulls: bool = True) -> pa.Table:
    rng = np.random.default_rng(seed)

    def mask(p):
        return (rng.random(n) < p) if nulls else None

    words = ["GET", "POST", "", "a much longer method name that repeats", "ünï©ode", "x" * 300]
    urls = [f"http://host{int(i) % 977}/p{int(i)}?q={'z' * (int(i) % 37)}" for i in rng.integers(0, 10**9, n)]
    return pa.table({
        "i64": pa.array(rng.integers(-2**62, 2**62, n), type=pa.int64(), mask=mask(0.1)),
        "small64": pa.array(rng.integers(0, 50, n), type=pa.int64(), mask=mask(0.5)),
        "i32": pa.array(rng.integers(-2**31, 2**31 - 1, n), type=pa.int32(), mask=mask(0.05)),
        "i16": pa.array(rng.integers(-2**15, 2**15 - 1, n).astype(np.int16), mask=mask(0.2)),
        "u16": pa.array(rng.integers(0, 2**16 - 1, n).astype(np.uint16), mask=mask(0.2)),
        "ts": pa.array(rng.integers(1_600_000_000_000, 1_700_000_000_000, n), type=pa.timestamp("ms"), mask=mask(0.1)),
        "method": pa.array([words[int(i)] for i in rng.integers(0, len(words), n)], type=pa.string(), mask=mask(0.3)),
        "url": pa.array(urls, type=pa.string(), mask=mask(0.02)),
        "blob": pa.array([bytes(rng.integers(0, 256, int(l)).astype(np.uint8)) for l in rng.integers(0, 40, n)], type=pa.binary(),
                         mask=mask(0.1)),
        "all_null": pa.array([None] * n, type=pa.int64()),
    })


def _write(t: pa.Table, **kw) -> bytes:
    sink = io.BytesIO()
    pq.write_table(t, sink, **kw)
    return sink.getvalue()


CASES = {
    "like_prism_ingest": dict(data_page_version="2.0", compression="snappy", use_dictionary=True),
    "v1_snappy_dict": dict(data_page_version="1.0", compression="snappy", use_dictionary=True),
    "v2_plain_uncompressed": dict(data_page_version="2.0", compression="none", use_dictionary=False),
    "v1_plain_snappy_small_pages": dict(data_page_version="1.0", compression="snappy", use_dictionary=False, data_page_size=4096),
    "v2_row_groups": dict(data_page_version="2.0", compression="snappy", use_dictionary=True, row_group_size=1500),
    "v2_dictionary_falls_back_to_plain": dict(data_page_version="2.0", compression="snappy", use_dictionary=True,
                                              dictionary_pagesize_limit=2048, data_page_size=8192),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("nulls", [True, False])
def test_decode_matches_pyarrow(case, nulls):
    t = _table(7000, seed=len(case), nulls=nulls)
    data = _write(t, **CASES[case])
    got = engine.parquet_read(data)
    want = pq.read_table(io.BytesIO(data))
    assert got.schema.names == want.schema.names
    for name in want.schema.names:
        assert got.column(name).type == want.column(name).type, name
        assert got.column(name).combine_chunks().equals(want.column(name).combine_chunks()), name


def test_required_columns_empty_file_and_timestamp_units():
    schema = pa.schema([pa.field("a", pa.int64(), nullable=False), pa.field("s", pa.string(), nullable=False),
                        pa.field("us", pa.timestamp("us")), pa.field("ns", pa.timestamp("ns"))])
    t = pa.table({"a": pa.array([1, -2, 3], pa.int64()), "s": pa.array(["x", "", "yz"]),
                  "us": pa.array([1_700_000_000_123_456, None, -1_999], pa.timestamp("us")),
                  "ns": pa.array([1_700_000_000_123_456_789, 5, None], pa.timestamp("ns"))}, schema=schema)
    data = _write(t, data_page_version="2.0", compression="snappy", coerce_timestamps=None, version="2.6")
    got = engine.parquet_read(data)
    assert got.column("a").to_pylist() == [1, -2, 3] and got.column("s").to_pylist() == ["x", "", "yz"]
    # other units read as Timestamp(ms), divided the way arrow's cast does (towards zero)
    assert got.column("us").type == pa.timestamp("ms") and got.column("us").cast(pa.int64()).to_pylist() == [1_700_000_000_123, None, -1]
    assert got.column("ns").cast(pa.int64()).to_pylist() == [1_700_000_000_123, 0, None]
    empty = _write(t.slice(0, 0), data_page_version="2.0", compression="snappy")
    assert engine.parquet_read(empty).num_rows == 0


def test_what_is_not_covered_says_so():
    with pytest.raises(engine.PrismGpuError) as e:
        engine.parquet_read(b"PAR1 not really PAR1")
    assert e.value.code == abi.EINVAL
    with pytest.raises(engine.PrismGpuError) as e:
        engine.parquet_read(_write(pa.table({"f": pa.array([1.5, 2.5])})))
    assert e.value.code == abi.EUNSUPPORTED and "`f`" in e.value.msg
    with pytest.raises(engine.PrismGpuError) as e:
        engine.parquet_read(_write(pa.table({"l": pa.array([[1, 2], [3]])})))
    assert e.value.code == abi.EUNSUPPORTED
    with pytest.raises(engine.PrismGpuError) as e:
        engine.parquet_read(_write(pa.table({"a": pa.array([1, 2, 3])}), compression="zstd"))
    assert e.value.code == abi.EUNSUPPORTED and "codec" in e.value.msg
    # truncated in the middle of the pages: an error, not a crash
    good = _write(_table(500, 3), data_page_version="2.0", compression="snappy")
    broken = good[:200] + b"\x00" * 64 + good[264:]
    try:
        engine.parquet_read(broken)
    except engine.PrismGpuError as err:
        assert err.code in (abi.EINVAL, abi.EUNSUPPORTED)


# ---- hostile header fields: the decoder is reachable from the C ABI with bytes it did not write ----

def _varint_at(data: bytes, p: int):
    """(value, next position) of the ULEB128 varint at p."""
    v, shift = 0, 0
    while True:
        b = data[p]
        p += 1
        v |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return v, p


def _page_header_fields(data: bytes, pos: int):
    """Walks the Thrift compact PageHeader at `pos` and returns {(struct field, field): (offset of the
    zigzag varint, its byte length)} for the i32 fields of the header (struct field 0) and of its
    nested DataPageHeader (5) / DataPageHeaderV2 (8), and the offset of the page body."""
    out = {}

    def walk(p, outer):
        last = 0
        while True:
            b = data[p]
            p += 1
            if b == 0:
                return p
            delta, ty = b >> 4, b & 15
            if delta == 0:
                fid, p = _varint_at(data, p)
                fid = (fid >> 1) ^ -(fid & 1)
            else:
                fid = last + delta
            last = fid
            if ty in (1, 2):
                continue
            if ty in (5, 6):  # i32 / i64: zigzag varint
                _, q = _varint_at(data, p)
                out[(outer, fid)] = (p, q - p)
                p = q
            elif ty == 12:
                p = walk(p, fid)
            elif ty == 8:  # binary
                n, p = _varint_at(data, p)
                p += n
            else:
                raise AssertionError(f"unexpected thrift type {ty} in a page header")

    body = walk(pos, 0)
    return out, body


def _negate_in_place(data: bytearray, off: int, length: int):
    """Makes the zigzag varint at `off` decode to a negative number without changing its length."""
    data[off] |= 1  # zigzag: odd = negative


def _first_data_page(data: bytes, column: int = 0) -> int:
    md = pq.ParquetFile(io.BytesIO(data)).metadata
    return md.row_group(0).column(column).data_page_offset


@pytest.mark.parametrize("version", ["1.0", "2.0"])
def test_negative_num_values_in_a_page_header_is_einval(version):
    t = pa.table({"a": pa.array(np.arange(1000, dtype=np.int64))}, schema=pa.schema([pa.field("a", pa.int64(), nullable=False)]))
    data = bytearray(_write(t, data_page_version=version, compression="none", use_dictionary=False))
    fields, _ = _page_header_fields(bytes(data), _first_data_page(bytes(data)))
    off, ln = fields[(8 if version == "2.0" else 5, 1)]  # num_values
    _negate_in_place(data, off, ln)
    with pytest.raises(engine.PrismGpuError) as e:
        engine.parquet_read(bytes(data))
    assert e.value.code == abi.EINVAL


def test_negative_level_lengths_and_sizes_are_einval():
    t = pa.table({"a": pa.array([None if i % 7 == 0 else i for i in range(1000)], type=pa.int64())})
    good = _write(t, data_page_version="2.0", compression="none", use_dictionary=False)
    fields, _ = _page_header_fields(good, _first_data_page(good))
    # DataPageHeaderV2: 5 = definition_levels_byte_length, 6 = repetition_levels_byte_length (0: "negative zero"
    # stays zero, skipped); PageHeader: 2 = uncompressed_page_size, 3 = compressed_page_size
    for key in [(8, 5), (0, 2), (0, 3)]:
        data = bytearray(good)
        off, ln = fields[key]
        _negate_in_place(data, off, ln)
        with pytest.raises(engine.PrismGpuError) as e:
            engine.parquet_read(bytes(data))
        assert e.value.code == abi.EINVAL, key


def test_bit_packed_run_with_a_wrapping_group_count_is_einval():
    rng = np.random.default_rng(5)
    t = pa.table({"a": pa.array(rng.integers(0, 200, 5000), type=pa.int64())},
                 schema=pa.schema([pa.field("a", pa.int64(), nullable=False)]))
    data = bytearray(_write(t, data_page_version="1.0", compression="none", use_dictionary=True))
    _, body = _page_header_fields(bytes(data), _first_data_page(bytes(data)))
    # body[0] is the bit width of the dictionary indices, then the first run header of the hybrid:
    # bit width 32 and a bit-packed run of 2^59 groups, whose byte count wraps round to zero in 64 bits
    data[body] = 32
    header = ((1 << 59) << 1) | 1
    p = body + 1
    while True:
        b = header & 0x7F
        header >>= 7
        data[p] = b | (0x80 if header else 0)
        p += 1
        if not header:
            break
    with pytest.raises(engine.PrismGpuError) as e:
        engine.parquet_read(bytes(data))
    assert e.value.code == abi.EINVAL


def test_a_column_chunk_that_claims_bytes_outside_the_file_is_einval():
    t = pa.table({"a": pa.array(np.arange(100, dtype=np.int64))})
    good = _write(t, data_page_version="2.0", compression="none", use_dictionary=False)
    # chop the file's middle out but keep the footer: offsets in the metadata now point past the data
    md_len = int.from_bytes(good[-8:-4], "little")
    broken = good[:4] + good[-(md_len + 8):]
    with pytest.raises(engine.PrismGpuError) as e:
        engine.parquet_read(broken)
    assert e.value.code == abi.EINVAL


def test_plain_pages_with_and_without_nulls_in_one_chunk():
    """Snappy v2 pages of PLAIN fixed-width values: pages in which no row is NULL are
    decompressed where their values belong (also the chunk's last page, which has no room
    behind it for the decoder's slack and takes the other route), pages with NULLs are not, and
    the column's buffer starts uninitialised, so NULL rows must come out as written zeros.
    Required and optional columns, 4- and 8-byte values."""
    n = 60_000
    rng = np.random.default_rng(5)
    some = np.zeros(n, dtype=bool)
    some[20_000:35_000] = rng.random(15_000) < 0.3  # NULLs only in the pages of the middle
    schema = pa.schema([pa.field("req64", pa.int64(), nullable=False), pa.field("opt64", pa.int64()),
                        pa.field("opt32", pa.int32()), pa.field("ts", pa.timestamp("ms"), nullable=False)])
    t = pa.table({"req64": pa.array(rng.integers(-2**62, 2**62, n), pa.int64()),
                  "opt64": pa.array(rng.integers(-2**40, 2**40, n), pa.int64(), mask=some),
                  "opt32": pa.array(rng.integers(-2**31, 2**31 - 1, n), pa.int32(), mask=some),
                  "ts": pa.array(1_700_000_000_000 + np.arange(n, dtype=np.int64) * 3, pa.timestamp("ms"))}, schema=schema)
    for page_size in (4096, 65536, 1 << 20):
        data = _write(t, data_page_version="2.0", compression="snappy", use_dictionary=False, data_page_size=page_size)
        got = engine.parquet_read(data)
        want = pq.read_table(io.BytesIO(data))
        for name in want.schema.names:
            assert got.column(name).combine_chunks().equals(want.column(name).combine_chunks()), (name, page_size)
            # NULL rows hold zeros underneath (what the table's kernels read for them)
            col = got.column(name).combine_chunks()
            width = 4 if name == "opt32" else 8
            raw = np.frombuffer(col.buffers()[1], dtype=np.int32 if width == 4 else np.int64, count=n, offset=col.offset * width)
            if col.null_count:
                assert not raw[~np.asarray(col.is_valid())].any(), name


@pytest.mark.parametrize("width", [pa.int64(), pa.int32()])
def test_an_integer_dictionary_index_beyond_the_dictionary_is_einval(width):
    """Dense dictionary-encoded integer pages are decoded and looked up in one pass; an index
    past the dictionary must not be read and is EINVAL (200 values: 8-bit indices, so a byte of
    0xFF among the bit-packed indices is index 255)."""
    rng = np.random.default_rng(9)
    t = pa.table({"a": pa.array(rng.integers(0, 200, 5000), type=width)}, schema=pa.schema([pa.field("a", width, nullable=False)]))
    data = bytearray(_write(t, data_page_version="1.0", compression="none", use_dictionary=True))
    assert engine.parquet_read(bytes(data)).column("a").combine_chunks().equals(t.column("a").combine_chunks())
    _, body = _page_header_fields(bytes(data), _first_data_page(bytes(data)))
    assert data[body] == 8  # the indices' bit width
    data[body + 1000] = 0xFF
    with pytest.raises(engine.PrismGpuError) as e:
        engine.parquet_read(bytes(data))
    assert e.value.code == abi.EINVAL and "index out of range" in e.value.msg


def test_flipped_bytes_in_the_pages_are_an_error_or_a_decode_never_a_crash():
    """The reader is reachable from the C ABI with bytes it did not write: bytes of the page area
    (headers, levels, Snappy blocks, dictionary indices) are overwritten at random, a few per
    trial; every trial must end in a table of the file's row count or in EINVAL / EUNSUPPORTED."""
    t = _table(3000, seed=17, nulls=True)
    cases = [_write(t, data_page_version="2.0", compression="snappy", use_dictionary=True, data_page_size=4096),
             _write(t, data_page_version="1.0", compression="snappy", use_dictionary=False, data_page_size=4096),
             _write(t, data_page_version="2.0", compression="none", use_dictionary=True, dictionary_pagesize_limit=2048)]
    rng = np.random.default_rng(23)
    outcomes = {"decoded": 0, "refused": 0}
    for good in cases:
        md_len = int.from_bytes(good[-8:-4], "little")
        body_end = len(good) - md_len - 8
        for _ in range(120):
            data = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(4, body_end))
                data[at] = int(rng.integers(0, 256))
            try:
                got = engine.parquet_read(bytes(data))
                assert got.num_rows == t.num_rows
                outcomes["decoded"] += 1
            except engine.PrismGpuError as err:
                assert err.code in (abi.EINVAL, abi.EUNSUPPORTED), err.msg
                outcomes["refused"] += 1
    assert outcomes["decoded"] and outcomes["refused"], outcomes
