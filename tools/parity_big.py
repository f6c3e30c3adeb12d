"""Diagnostics: full-size parity of the HIP path against the oracle, with the first differences.
tools/parity_big.py <reads> <levels>; L2R_CFG = workload config, L2R_DIS = -d, L2R_SJ=1: the second option set (-s -J 1 -j with the
junction table bench.py's second_pass leg makes from the first pass)."""
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
dis = int(os.environ.get("L2R_DIS", "0"))
for level in levels:
    extra = dict(ss_dis=dis)
    sj = None
    if os.environ.get("L2R_SJ"):
        from lr2rmats_amd import synth
        e.set_junctions(None)
        first = e.classify(reads, capi.default_params(full_level=level, ss_dis=dis))
        j = synth.make_junctions_fast(af, first.ex_off, first.ex_start, first.ex_end, reads.tid, seed=3, cover=0.8)
        sj = (j.tid, j.don, j.acc, j.uniq, j.multi)
        extra.update(split_trans=1, min_sj_cnt=1)
    e.set_junctions(sj)
    got = e.classify(reads, capi.default_params(full_level=level, **extra))
    want = po.classify_soa(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, af.tx_tid, af.tx_start, af.tx_end, af.tx_rev,
                           af.tx_ex_off, af.ex_start, af.ex_end, sj=sj, params=po.default_params(full_level=level, **extra))
    print("-l", level, extra, "junction rows", 0 if sj is None else len(sj[0]))
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
