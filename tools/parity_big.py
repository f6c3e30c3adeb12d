"""Diagnostics: full-size parity of the HIP path against the oracle, with the first differences."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lr2rmats_amd import capi, workload
from oracle import pyoracle as po
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
cfg = dict(workload.CONFIGS[os.environ.get('L2R_CFG', 'cfg3')]); cfg['n_reads'] = N
af, reads = workload.make_rank_workload(cfg, 0, 1)
e = capi.Engine(0)
e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
po.build()
levels = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [3]
for level in levels:
    got = e.classify(reads, capi.default_params(full_level=level))
    want = po.classify_soa(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, af.tx_tid, af.tx_start, af.tx_end, af.tx_rev,
                           af.tx_ex_off, af.ex_start, af.ex_end, params=po.default_params(full_level=level))
    print("-l", level)
    for name in ("ex_off", "ex_start", "ex_end", "ex_flag", "ref_tx"):
        a, b = getattr(got, name), getattr(want, name)
        m = min(len(a), len(b))
        d = np.nonzero(a[:m] != b[:m])[0]
        print(name, len(a), len(b), "diffs", len(d), d[:10])
    a, b = got.info & 0x7f, want.info & 0x7f
    d = np.nonzero(a != b)[0]
    print("info diffs", len(d), d[:10], a[d[:10]], b[d[:10]])
    if len(d):
        print("tiles of first diffs", d[:20] // 256, "lane", d[:20] % 256)
