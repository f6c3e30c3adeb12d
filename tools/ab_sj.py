"""Diagnostics on the GPU box: the second option set (-s -l 3 -J 1 -j SJ.tab, results + accepted list; bench.py's second_pass leg) under
L2R_ABLATE values, alternating on one box: tools/ab_sj.py <reads> <config> <ablate values ...>.  Bits 512 / 1024 / 2048 (k_tile): no read is a
candidate of the junction check / the whole check is skipped / no table lookups (cursor row and Q7 only) -- timing only, the results are wrong.  Not part of the product."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lr2rmats_amd import capi, workload, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
cfgname = sys.argv[2] if len(sys.argv) > 2 else 'cfg3'
vals = [int(x) for x in sys.argv[3:]] or [0]
rounds = int(os.environ.get("AB_ROUNDS", "3"))
iters = int(os.environ.get("AB_ITERS", "10"))
cfg = dict(workload.CONFIGS[cfgname]); cfg['n_reads'] = N
af, reads = workload.make_rank_workload(cfg, 0, 1)
e = capi.Engine(0)
e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
got = e.classify(reads, capi.default_params(full_level=3))
sj = synth.make_junctions_fast(af, got.ex_off, got.ex_start, got.ex_end, reads.tid, seed=3, cover=0.8)
del e
engines = {}
for v in vals:
    os.environ["L2R_ABLATE"] = str(v)
    e = capi.Engine(0)
    e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
    e.set_junctions((sj.tid, sj.don, sj.acc, sj.uniq, sj.multi))
    e.set_params(capi.default_params(full_level=3, split_trans=1, min_sj_cnt=1))
    e.set_outputs(capi.WANT_RESULTS | capi.WANT_ACCEPTED)
    e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
    e.run(); e.sync()
    engines[v] = e
acc = {v: [] for v in vals}
for _ in range(rounds):
    for v in vals:
        acc[v].append(engines[v].run_timed(iters))
for v in vals:
    best = min(acc[v], key=lambda t: t["total_ms"])
    print("ablate %4d: total %.4f ms (rounds: %s) %s" % (v, best["total_ms"], " ".join("%.4f" % t["total_ms"] for t in acc[v]),
                                                        {k.split(" ")[0]: round(x, 4) for k, x in best["kernel_ms"].items() if x > 0.003}))
