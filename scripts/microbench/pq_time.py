import io, time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pyarrow as pa, pyarrow.parquet as pq
from prism_amd import workloads
from prism_amd.engine import parquet_read
wl=workloads.c2(); names=wl.dictionaries["service"]
rng=np.random.default_rng(7); per=2_000_000; rows=32_000_000
w=1.0/np.arange(1,len(names)+1)**1.1; w/=w.sum()
ts=workloads.T0_MS+(np.arange(0,per,dtype=np.int64)*workloads.DAY_MS)//rows
codes=rng.choice(len(names),size=per,p=w).astype(np.int32)
svc=pa.DictionaryArray.from_arrays(pa.array(codes),pa.array(names)).cast(pa.string())
cols={"timestamp":pa.array(ts,pa.timestamp("ms")),"service":svc,"value":pa.array(rng.integers(0,50_000,per,dtype=np.int64))}
for sel in (["timestamp"],["value"]):
    t=pa.table({k:cols[k] for k in sel})
    sink=io.BytesIO(); pq.write_table(t,sink,version="2.6",data_page_version="2.0",compression="snappy",use_dictionary=True)
    data=sink.getvalue()
    best=1e9
    for _ in range(7):
        t0=time.perf_counter(); parquet_read(data); best=min(best,time.perf_counter()-t0)
    print(sel, round(best*1e3,1),'ms')
