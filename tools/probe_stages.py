"""Quick timing probe on the GPU box (diagnostics, not part of the product)."""
import sys, time, json, os
import numpy as np
sys.path.insert(0, '.')
from lr2rmats_amd import capi, synth, workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
cfg = dict(workload.CONFIGS['cfg3']); cfg['n_reads'] = N
af, reads = workload.make_rank_workload(cfg, 0, 1)
e = capi.Engine(0)
t = time.time()
e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
print('set_annotation %.2f s' % (time.time() - t))
e.set_params(capi.default_params(full_level=3))
e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
for ab in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['0']):
    os.environ['L2R_ABLATE'] = ab
    e.run(); e.sync()
    tm = e.run_timed(5)
    print('ablate', ab, 'total %.3f' % tm['total_ms'], ' '.join('%s=%.3f' % (k[:10], v) for k, v in tm['stage_ms'].items()), flush=True)
