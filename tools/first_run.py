"""Diagnostics on the GPU box: what the FIRST run behind a fresh upload costs, kernel by kernel (l2r_run_timed(1): HIP events around every
launch), against the runs behind it.  tools/first_run.py <reads> <config>; L2R_PIPELINE as usual.  Not part of the product."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lr2rmats_amd import capi, workload, hostlib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
cfgname = sys.argv[2] if len(sys.argv) > 2 else 'cfg3'
cfg = dict(workload.CONFIGS[cfgname]); cfg['n_reads'] = N
af, reads = workload.make_rank_workload(cfg, 0, 1)
sm = hostlib.cigar_summaries(reads.cig_off, reads.cig)
e = capi.Engine(0)
e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
e.set_params(capi.default_params(full_level=3)); e.set_outputs(capi.WANT_RESULTS)
def short(tm):
    return "%.4f" % tm["total_ms"], {k.split(" ")[0]: round(x, 4) for k, x in tm["kernel_ms"].items() if x > 0.001}
for rep in range(4):
    e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, cig_summary=sm)
    print("upload %d: index %.4f ms" % (rep, e.upload_index_ms()))
    for k in range(3):
        print("   run %d:" % k, *short(e.run_timed(1)))
print("steady:", *short(e.run_timed(25)))
