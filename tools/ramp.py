import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from lr2rmats_amd import capi, workload
cfg = dict(workload.CONFIGS['cfg3'])
af, reads = workload.make_rank_workload(cfg, 0, 1)
e = capi.Engine(0)
e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
e.set_params(capi.default_params(full_level=3)); e.set_outputs(1)
e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
def region(k):
    e.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): e.run()
    e.sync(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print("first 5 (warmup):", round(region(5), 4))
for i in range(8): print("20 steps:", round(region(20), 4))
time.sleep(2.0)
print("after 2 s idle, 5:", round(region(5), 4))
for i in range(3): print("20 steps:", round(region(20), 4))
