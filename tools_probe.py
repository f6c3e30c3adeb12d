"""Quick timing probe on the GPU box (not part of the product)."""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from lr2rmats_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nex = int(sys.argv[2]) if len(sys.argv) > 2 else 8
E = int(sys.argv[3]) if len(sys.argv) > 3 else 1500000
t = time.time(); anno = synth.make_annotation(E, 3, mean_tx_exons=nex + 1); af = anno.in_file_order()
reads = synth.make_reads(anno, N, nex, 3); print('gen', time.time() - t, 'tx', af.n_tx, 'exons', af.n_exons, 'ops', len(reads.cig), flush=True)
e = capi.Engine(0)
e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
e.set_params(capi.default_params(full_level=3))
t = time.time(); e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig); print('upload', time.time() - t)
e.run(); e.sync()
print('sizes', e.sizes())
tm = e.run_timed(5)
print(json.dumps(tm, indent=1))
print('reads/s', N / (tm['total_ms'] * 1e-3))
