"""Diagnostics: parity of a slice of config 3's reads (whole tiles) against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lr2rmats_amd import capi, workload
from oracle import pyoracle as po
t0, t1 = int(sys.argv[1]), int(sys.argv[2])
cfg = dict(workload.CONFIGS['cfg3'])
af, reads = workload.make_rank_workload(cfg, 0, 1)
sub = reads.slice(t0 * 256, t1 * 256)
e = capi.Engine(0)
e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
got = e.classify(sub, capi.default_params(full_level=3))
po.build()
want = po.classify_soa(sub.tid, sub.pos, sub.rev, sub.cig_off, sub.cig, af.tx_tid, af.tx_start, af.tx_end, af.tx_rev,
                       af.tx_ex_off, af.ex_start, af.ex_end, params=po.default_params(full_level=3))
a, b = got.info & 0x7f, want.info & 0x7f
d = np.nonzero(a != b)[0]
print("info diffs", len(d), d[:10], a[d[:10]], b[d[:10]], "flag diffs", int((got.ex_flag != want.ex_flag).sum()), "ref diffs", int((got.ref_tx != want.ref_tx).sum()))
import ctypes as C
cnt = (C.c_longlong * 4)()
lib = capi.load_library()
lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.l2r_debug_counters(e.ctx, cnt, 4)
print("redo reads %d, wide %d, compact %d, tiles %d" % tuple(cnt))
print("tids", np.unique(sub.tid), "tid of diff reads", sub.tid[d[:10]])
