"""prism_gpu_execute_noalloc (the entry point that allocates no device memory) against
prism_gpu_execute on the same plan and partitions, through the C ABI: two calls in a row,
the second over a smaller input, so that whatever the first call left in the library's
device array would show.  The whole-table result is also checked against the CPU oracle, so
that a fault shared by both entry points would not pass."""
import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu

from prism_amd import _cabi as abi  # noqa: E402
from prism_amd import workloads  # noqa: E402
from prism_amd.plan import QueryPlan, avg, col, count_star, date_bin, max_, min_, sum_  # noqa: E402


def _rows_from_buffers(plan, table_dicts, values, valid, n):
    """Decodes the [n_columns][stride] buffers into sorted Python rows."""
    cols = []
    for c, f in enumerate(plan.final_schema):
        v, ok = values[c, :n], valid[c, :n].astype(bool)
        if pa.types.is_string(f.type):
            d = table_dicts[f.name]
            col_ = [d[int(x)] if o else None for x, o in zip(v, ok)]
        elif pa.types.is_floating(f.type):
            col_ = [float(x) if o else None for x, o in zip(v.view(np.float64), ok)]
        elif pa.types.is_timestamp(f.type):
            col_ = [int(x) if o else None for x, o in zip(v, ok)]
        else:
            col_ = [int(x) if o else None for x, o in zip(v, ok)]
        cols.append(col_)
    return sorted(zip(*cols), key=lambda r: tuple((x is None, x) for x in r))


def _rows_from_arrow(t: pa.Table):
    cols = []
    for c, f in zip(t.columns, t.schema):
        if pa.types.is_timestamp(f.type):
            cols.append(c.cast(pa.int64()).to_pylist())
        else:
            cols.append(c.to_pylist())
    return sorted(zip(*cols), key=lambda r: tuple((x is None, x) for x in r))


def test_noalloc_matches_execute_twice_second_call_smaller(gpu_ctx):
    import torch
    from prism_amd.engine import Plan, PrismGpuError

    wl = workloads.c2()
    rows, parts = 600_000, 6
    table = workloads.build_gpu_table(gpu_ctx, wl, 0, rows, rows, parts)
    queries = [
        wl.query,
        QueryPlan([count_star(), sum_("value"), min_("value"), max_("value"), avg("value")],
                  keys=[date_bin(60_000, "timestamp"), "service"], predicate=(col("value") >= 1000) & (col("value") < 40_000)),
    ]
    stride = 1 << 17
    try:
        for q in queries:
            plan = Plan(gpu_ctx, wl.schema, q)
            n_cols = len(plan.final_schema)
            values = torch.full((n_cols, stride), -1, dtype=torch.int64, device="cuda:0")
            valid = torch.full((n_cols, stride), 7, dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            try:
                # first the whole table, then two of its partitions: a smaller input, fewer groups
                for partitions in (None, [1, 4]):
                    want = _rows_from_arrow(plan.execute(table, partitions))
                    n = plan.execute_noalloc(table, partitions, values.data_ptr(), valid.data_ptr(), stride)
                    got = _rows_from_buffers(plan, wl.dictionaries, values.cpu().numpy(), valid.cpu().numpy(), n)
                    assert n == len(want) and n > 0
                    assert got == want  # integers and keys exactly; AVG is the same division of the same sums
                    if partitions is None:
                        from oracle import pyoracle
                        from tests.helpers import assert_tables_match

                        st = pyoracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, rows, rows, n_parts=parts)
                        try:
                            oracle_rows, _ = st.execute(q)
                        finally:
                            st.close()
                        oracle_rows = workloads.decode_dictionary_columns(wl, oracle_rows)
                        assert_tables_match(plan.execute(table, None), oracle_rows, 2)
                        assert n == oracle_rows.num_rows
                # buffers too short: an error, and nothing written past them
                with pytest.raises(PrismGpuError) as e:
                    plan.execute_noalloc(table, None, values.data_ptr(), valid.data_ptr(), 16)
                assert e.value.code == abi.EINVAL
            finally:
                plan.close()
        # beyond the fixed array: a clear refusal, not an allocation
        big = Plan(gpu_ctx, wl.schema, QueryPlan([count_star()], keys=["value", "service"]))
        try:
            with pytest.raises(PrismGpuError) as e:
                big.execute_noalloc(table, None, 0, 0, 0)
            assert e.value.code == abi.EUNSUPPORTED and "accumulators" in e.value.msg
        finally:
            big.close()
    finally:
        table.close()
