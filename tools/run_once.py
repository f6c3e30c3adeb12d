"""Diagnostics on the GPU box: a few launches of the hot path on one config; with L2R_STAMPS=1 prints
the per-phase cycle sums of k_classify_fast (thread 0 of every tile).  Not part of the product."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lr2rmats_amd import capi, workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
cfgname = sys.argv[2] if len(sys.argv) > 2 else 'cfg3'
cfg = dict(workload.CONFIGS[cfgname]); cfg['n_reads'] = N
af, reads = workload.make_rank_workload(cfg, 0, 1)
e = capi.Engine(0)
e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
e.set_params(capi.default_params(full_level=int(os.environ.get("L2R_LEVEL", "3")), ss_dis=int(os.environ.get("L2R_DIS", "0"))))
e.set_outputs(int(os.environ.get("L2R_WANT", "1")))          # 1 = per-read results (what bench.py times), 3 = + accepted list
e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
lib = capi.load_library()
out = (C.c_ulonglong * 16)()
for _ in range(2):
    e.run(); e.sync()
if os.environ.get("L2R_STAMPS"):
    lib.l2r_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.l2r_debug_stamps(e.ctx, out, 16)
    e.run(); e.sync()
    lib.l2r_debug_stamps(e.ctx, out, 16)
    v = list(out)
    tot = sum(v) or 1
    tot = sum(v[:8]) or 1
    print("stamps (cycles of thread 0 summed over tiles):", [(i, x, round(100.0 * x / tot, 1)) for i, x in enumerate(v[:8]) if x])
    print("  slab pipeline (k_probe_slab, wave 0): 0 loads 1 staging + barrier 2 window pass 3 probe rounds 4 verdicts 5 write-out | last wave: 6 whole kernel 7 probe rounds; per tile:", [round(x / max(1, int(os.environ.get("L2R_TILES", "39074")))) for x in v[:8]])
    print("  one-kernel tile path (k_tile, wave 0): 0 records + sort + CIGAR heads asked for + staging 1 count walk + barrier 6 scan + publish + place walk 2 window pass 3 probe rounds 4 verdicts 7 first slot (counts in front) 5 write-out; per tile:", [round(x / max(1, int(os.environ.get("L2R_TILES", "39074")))) for x in v[:8]])
    print("  classic kernel: 0 CIGAR staging 7 walk 1 dictionary staging 2 window pass 3 probes 4 verdicts 5 counts 6 write-out")
    print("  one-walk kernel: 0 walk 1 staging 2 window pass + next span 3 probes + verdicts 4 offsets/map/write-out | 5 barrier waits of wave 0, 6 of the last wave, 7 descriptor work of the last wave (5-7 are not phases: compare with the sum of 0-4)")
    print("redo reasons [not fast, not in LDS, wide, other tid, not sane, window/compact]:", v[8:14])
cnt = (C.c_longlong * 24)()
lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.l2r_debug_counters(e.ctx, cnt, 24)
print("runs done again on the slab pipeline %d | k_tile_chunk declined %d tiles, %d handed on late" % (cnt[13], cnt[14], cnt[15]))
print("tiles of the chunked kernel by END entries [<=256, <=512, <=768, <=1024, more]:", list(cnt[16:21]), "| START entries > 128:", cnt[21], "> 256:", cnt[22], "| all:", cnt[23])
print("redo reads %d, wide entries %d, compact tx %d, tiles %d" % tuple(cnt[:4]), "| tiles of the 64-member kernel:", cnt[12])
print("tiles [fast, exons > LDS cap, bucket span, dictionary slice, window > 32, window scan, cursor behind window, off]:", list(cnt[4:12]))
# (a profile's per-kernel averages: enough back-to-back steps for the chip's clock to settle -- it rises over the first ~25, tools/ramp.py)
tm = e.run_timed(int(os.environ.get("L2R_ONCE_ITERS", "30")))
print(e.sizes(), tm)
