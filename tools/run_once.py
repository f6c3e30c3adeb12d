import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lr2rmats_amd import capi, workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
cfg = dict(workload.CONFIGS['cfg3']); cfg['n_reads'] = N
af, reads = workload.make_rank_workload(cfg, 0, 1)
e = capi.Engine(0)
e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
e.set_params(capi.default_params(full_level=3))
e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
for _ in range(3):
    e.run(); e.sync()
print(e.sizes())
