"""Diagnostics on the GPU box: A/B timing of engine variants selected by ONE environment variable read at l2r_create, alternating on one
box and one workload: tools/ab_env.py <reads> <config> <VAR> <value ...>  (L2R_LEVEL, L2R_DIS, L2R_WANT as in tools/ab.py).  Not part of the product."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lr2rmats_amd import capi, workload
N = int(sys.argv[1]); cfgname = sys.argv[2]; var = sys.argv[3]; vals = sys.argv[4:]
rounds = int(os.environ.get("AB_ROUNDS", "3")); iters = int(os.environ.get("AB_ITERS", "20"))
cfg = dict(workload.CONFIGS[cfgname]); cfg['n_reads'] = N
af, reads = workload.make_rank_workload(cfg, 0, 1)
engines = {}
for v in vals:
    os.environ[var] = v
    e = capi.Engine(0)
    e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
    e.set_params(capi.default_params(full_level=int(os.environ.get("L2R_LEVEL", "3")), ss_dis=int(os.environ.get("L2R_DIS", "0"))))
    e.set_outputs(int(os.environ.get("L2R_WANT", "1")))
    e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
    e.run(); e.sync(); e.run(); e.sync()
    engines[v] = e
acc = {v: [] for v in vals}
for _ in range(rounds):
    for v in vals:
        acc[v].append(engines[v].run_timed(iters))
for v in vals:
    best = min(acc[v], key=lambda t: t["total_ms"])
    print("%s=%s: total %.4f ms (rounds: %s) %s" % (var, v, best["total_ms"], " ".join("%.4f" % t["total_ms"] for t in acc[v]),
                                                   {k.split(" ")[0]: round(x, 4) for k, x in best["kernel_ms"].items() if x > 0.003}))
