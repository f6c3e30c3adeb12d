import sys, time, json, os
import numpy as np
sys.path.insert(0, '.')
from lr2rmats_amd import capi, synth, workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
cfg = dict(workload.CONFIGS['cfg3']); cfg['n_reads'] = N
af, reads = workload.make_rank_workload(cfg, 0, 1)
e = capi.Engine(0)
e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
for lvl, dis in ((3,0),(3,1),(5,0),(5,1)):
    e.set_params(capi.default_params(full_level=lvl, ss_dis=dis))
    e.run(); e.sync()
    tm = e.run_timed(5)
    r = e.download()
    print('lvl', lvl, 'dis', dis, 'fill %.3f' % tm['stage_ms']['classify_fast'], 'known', int(((r.info&1)!=0).sum()), flush=True)
