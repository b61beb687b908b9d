"""Parquet -> HBM loader: the step in front of the scan.

The reference gives DataFusion's ParquetExec one file group per partition and a column
projection (src/prism-query/src/providers/table_provider.rs:96-131); its ingest writes
one Parquet file per partition: data page v2, dictionary encoding with PLAIN fallback,
Snappy, Timestamp(ms) / Int64 / Utf8 columns (src/prism-ingest/src/ingest.rs:98-119,
145-157).  This module is the host-side form of that step: pyarrow's Parquet reader
decodes row groups into Arrow record batches, which prism_gpu_table_append stages
through pinned memory into HBM (strings get their table-wide dictionary codes there).
It is a caller of the C ABI, not part of it: decoding runs on the host.
"""
from __future__ import annotations

from typing import Iterable, Optional, Sequence, Union

import pyarrow as pa
import pyarrow.parquet as pq

from .engine import Table


def _conform(batch: pa.RecordBatch, schema: pa.Schema) -> pa.RecordBatch:
    """Projects and casts a decoded batch to the table schema (ParquetExec's schema
    adapter: missing columns become NULL, Binary <-> Utf8 and timestamp units are cast)."""
    cols = []
    for f in schema:
        i = batch.schema.get_field_index(f.name)
        if i < 0:
            cols.append(pa.nulls(batch.num_rows, f.type))
            continue
        a = batch.column(i)
        if pa.types.is_dictionary(a.type):
            if pa.types.is_string(f.type) or pa.types.is_binary(f.type):
                if a.type.index_type != pa.int32() or a.type.value_type != f.type:
                    a = a.cast(pa.dictionary(pa.int32(), f.type))
            else:
                a = a.cast(f.type)
        elif a.type != f.type:
            a = a.cast(f.type)
        cols.append(a)
    return pa.RecordBatch.from_arrays(cols, names=schema.names)


def load_parquet(table: Table, files: Sequence[Union[str, Iterable[str]]], columns: Optional[Sequence[str]] = None,
                 batch_rows: int = 1 << 20, first_partition: int = 0) -> int:
    """Appends Parquet files to `table`: entry i of `files` (a path, or several paths) becomes
    partition first_partition + i.  `columns` restricts decoding to the columns a query
    references (the others are filled with NULLs).  Returns the rows loaded."""
    total = 0
    want = list(columns) if columns is not None else table.schema.names
    for i, entry in enumerate(files):
        paths = [entry] if isinstance(entry, str) else list(entry)
        for path in paths:
            pf = pq.ParquetFile(path, read_dictionary=[c for c in want if pa.types.is_string(table.schema.field(c).type)])
            present = [c for c in want if c in pf.schema_arrow.names]
            for batch in pf.iter_batches(batch_size=batch_rows, columns=present):
                if batch.num_rows == 0:
                    continue
                table.append(first_partition + i, _conform(batch, table.schema))
                total += batch.num_rows
    return total
