"""The synthetic generators on the CPU: a pure-Python restatement of
include/prism_synth.h must agree with what the oracle's generated tables aggregate to,
so the generator is pinned independently of any GPU."""
import pyarrow as pa

from oracle import pyoracle
from prism_amd import _cabi as abi
from prism_amd import workloads
from prism_amd.plan import QueryPlan, count_star, max_, min_, sum_

M64 = (1 << 64) - 1


def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def column_key(seed, column):
    return mix64(seed ^ ((0xD6E8FEB86659FD93 * (column + 1)) & M64))


def value(kind, p, cdf, key, row, total):
    r = mix64((key + row) & M64)
    if kind == abi.SYNTH_TS_LINEAR:
        return p[0] + (row * p[1]) // total
    if kind == abi.SYNTH_CATEGORICAL:
        code = 0
        while code + 1 < len(cdf) and (r >> 32) >= cdf[code]:
            code += 1
        return code
    if kind == abi.SYNTH_UNIFORM:
        return p[0] + ((r * p[1]) >> 64)
    if kind == abi.SYNTH_SPARSE:
        return 1 + (((r & 0xFFFFFFFF) * p[1]) >> 32) if (r >> 32) < p[0] else 0
    if kind == abi.SYNTH_LOGUNIFORM_ID:
        r2 = mix64(r)
        k = (r * p[0]) >> 64
        rank = (1 << k) + (r2 & ((1 << k) - 1))
        return mix64(rank ^ p[1]) >> 1
    raise ValueError(kind)


def test_c2_generator_matches_python_restatement():
    wl = workloads.c2()
    n, total, first = 3000, 10_000, 2000
    cols = wl.make_cols()
    st = pyoracle.SynthTable(wl.schema, cols, wl.seed, first, n, total, n_parts=3)
    got, _ = st.execute(QueryPlan([count_star(), sum_("value"), max_("value"), min_("timestamp"), max_("timestamp")],
                                  keys=["service"]))
    st.close()
    cdf = workloads.zipf_cdf(64, 1.1)
    keys = [column_key(wl.seed, c) for c in range(3)]
    want = {}
    for row in range(first, first + n):
        ts = value(abi.SYNTH_TS_LINEAR, (workloads.T0_MS, workloads.DAY_MS), None, keys[0], row, total)
        svc = value(abi.SYNTH_CATEGORICAL, (), cdf, keys[1], row, total)
        v = value(abi.SYNTH_UNIFORM, (0, 50_000), None, keys[2], row, total)
        c, s, m, lo, hi = want.get(svc, (0, 0, -1, 1 << 62, -1))
        want[svc] = (c + 1, s + v, max(m, v), min(lo, ts), max(hi, ts))
    rows = {r[0]: (r[1], r[2], r[3], pa.scalar(r[4], pa.timestamp("ms")).cast(pa.int64()).as_py(),
                   pa.scalar(r[5], pa.timestamp("ms")).cast(pa.int64()).as_py())
            for r in zip(*[c.to_pylist() for c in got.columns])}
    assert rows == want


def test_c1_and_c3_generators():
    wl = workloads.c1()
    n = 20_000
    st = pyoracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, n, n, n_parts=2)
    got, _ = st.execute(wl.query)
    st.close()
    key = column_key(wl.seed, 0)
    p = (int(0.0063 * 2 ** 32), 60)
    vals = [value(abi.SYNTH_SPARSE, p, None, key, r, n) for r in range(n)]
    assert got.column(0).to_pylist() == [sum(1 for v in vals if v != 0)]
    assert 0 < got.column(0)[0].as_py() < n // 50 and max(vals) <= 60

    wl = workloads.c3(id_bits=10)
    n = 5000
    st = pyoracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, n, n, n_parts=2)
    got, _ = st.execute(QueryPlan([count_star()], keys=["UserID"]))
    st.close()
    key = column_key(wl.seed, 0)
    want = {}
    for r in range(n):
        v = value(abi.SYNTH_LOGUNIFORM_ID, (10, 0x5EED), None, key, r, n)
        want[v] = want.get(v, 0) + 1
    assert dict(zip(got.column(0).to_pylist(), got.column(1).to_pylist())) == want
    assert all(0 <= k < 2 ** 63 for k in want)


def test_final_stage_repartitioned_by_key_hash_equals_the_sequential_fold():
    """Many groups with integer keys: the checker's Final stage runs one hash slice of the
    partial tables per thread (RepartitionExec(Hash) + FinalPartitioned); same groups and
    aggregates as one thread folding everything."""
    import pyarrow.compute as pc
    from prism_amd.plan import max_, sum_

    wl = workloads.c3(id_bits=22)
    n = 700_000
    q = QueryPlan([count_star(), sum_("UserID"), max_("UserID")], keys=["UserID"])
    res = []
    for threads in (1, 4):
        st = pyoracle.SynthTable(wl.schema, wl.make_cols(), wl.seed, 0, n, n, n_parts=16, threads=4)
        t, _ = st.execute(q, threads=threads)
        st.close()
        res.append(t.take(pc.sort_indices(t, [("UserID", "ascending")])))
    assert res[0].num_rows > 200_000 and res[0].equals(res[1])
