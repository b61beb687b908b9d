import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dev = torch.device("cuda",0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from prism_amd import workloads
from prism_amd.engine import Context, Plan
from prism_amd import distributed as D
ctx = Context(0); wl = workloads.WORKLOADS["C2"]()
rows = 1_000_000_000
table = workloads.build_gpu_table(ctx, wl, 0, rows, rows, 64)
plan = Plan(ctx, wl.schema, wl.query)
def T(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(6):
    t = [T()]
    st = plan.scan_stats(table, None); t.append(T())
    stats, same = D.agree_on_stats(st, [table.dictionary_digest("service")[0]], dev); t.append(T())
    state = plan.aggregate(table, None, stats); t.append(T())
    info = state.describe(); sm = list(info.slot_merge[:info.n_slots]); n = info.export_bytes//8
    buf = torch.empty(n, dtype=torch.int64, device=dev); state.export_to(buf.data_ptr(), info.export_bytes); t.append(T())
    D.reduce_dense_state(buf, sm, info.entries, 0); t.append(T())
    state.import_from(buf.data_ptr(), info.export_bytes); t.append(T())
    out = state.finish(); t.append(T())
    state.close(); t.append(T())
    if it >= 2: print(" ".join(f"{(b-a)*1e3:.3f}" for a,b in zip(t,t[1:])), "total", f"{(t[-1]-t[0])*1e3:.3f}", flush=True)
print("labels: scan_stats agree aggregate export reduce import finish close")
