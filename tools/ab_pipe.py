"""Diagnostics on the GPU box: A/B timing of the engine's kernel pipelines (L2R_PIPELINE, read at l2r_create) on ONE box and one
workload: tools/ab_pipe.py <reads> <config> <pipeline ...>.  Prints the per-kernel milliseconds of l2r_run_timed per pipeline,
interleaved over several rounds so that clock drift hits every variant alike, and checks that the pipelines' results are
identical.  Not part of the product."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lr2rmats_amd import capi, workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
cfgname = sys.argv[2] if len(sys.argv) > 2 else 'cfg3'
pipes = sys.argv[3:] or ['tile', 'slab']
rounds = int(os.environ.get("AB_ROUNDS", "3"))
iters = int(os.environ.get("AB_ITERS", "20"))
cfg = dict(workload.CONFIGS[cfgname]); cfg['n_reads'] = N
af, reads = workload.make_rank_workload(cfg, 0, 1)
engines, results = {}, {}
for v in pipes:
    os.environ["L2R_PIPELINE"] = v
    e = capi.Engine(0)
    e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
    e.set_params(capi.default_params(full_level=int(os.environ.get("L2R_LEVEL", "3")), ss_dis=int(os.environ.get("L2R_DIS", "0"))))
    e.set_outputs(int(os.environ.get("L2R_WANT", "1")))
    e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
    e.run(); e.sync()
    engines[v] = e
    results[v] = e.download()
ref = results[pipes[0]]
for v in pipes[1:]:
    r = results[v]
    same = all(np.array_equal(getattr(ref, k), getattr(r, k)) for k in ("ex_off", "ex_start", "ex_end", "ex_flag", "ref_tx", "info"))
    print("results %s == %s: %s" % (pipes[0], v, same))
acc = {v: [] for v in pipes}
for _ in range(rounds):
    for v in pipes:
        acc[v].append(engines[v].run_timed(iters))
for v in pipes:
    best = min(acc[v], key=lambda t: t["total_ms"])
    ks = {k.split(" ")[0]: round(x, 4) for k, x in best["kernel_ms"].items() if x > 0.003}
    print("%-8s total %.4f ms (rounds: %s) %s" % (v, best["total_ms"], " ".join("%.4f" % t["total_ms"] for t in acc[v]), ks))
