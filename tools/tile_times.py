"""Diagnostics on the GPU box (L2R_STAMPS=1): when the tiles of k_tile start, publish their exon counts and wait for the counts in
front of them, by the chip's 100 MHz clock: tools/tile_times.py <reads> <config>.  Not part of the product."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault("L2R_STAMPS", "1")
from lr2rmats_amd import capi, workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
cfgname = sys.argv[2] if len(sys.argv) > 2 else 'cfg3'
cfg = dict(workload.CONFIGS[cfgname]); cfg['n_reads'] = N
af, reads = workload.make_rank_workload(cfg, 0, 1)
e = capi.Engine(0)
e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
e.set_params(capi.default_params(full_level=3))
e.set_outputs(1)
e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
lib = capi.load_library()
for _ in range(3):
    e.run(); e.sync()
cnt = (C.c_longlong * 13)()
lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.l2r_debug_counters(e.ctx, cnt, 13)
T = int(cnt[3])
out = np.zeros(4 * T, np.uint32)
lib.l2r_debug_tile_times.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
rc = lib.l2r_debug_tile_times(e.ctx, out.ctypes.data_as(C.c_void_p), T)
assert rc == 0, rc
w = out.reshape(T, 4).astype(np.int64)
xcc = w[:, 0] & 7
start = (w[:, 0] >> 3) & 0x1fffffff
def rel(x, bits):
    m = (1 << bits) - 1
    return ((x - start) & m)
t0 = start.min()
s = (start - t0) / 100.0            # us
pub = rel(w[:, 1], 29) / 100.0; wb = rel(w[:, 2], 29) / 100.0; we = rel(w[:, 3], 29) / 100.0
print("tiles", T, "kernel span us", s.max() + we[np.argmax(s)])
print("per tile us: start->publish %.2f  start->wait begin %.2f  wait %.2f  (medians)" % (np.median(pub), np.median(wb), np.median(we - wb)))
print("wait us percentiles 50/90/99/max:", np.percentile(we - wb, [50, 90, 99, 100]))
# dispatch order: start time against tile number
order = np.argsort(s, kind="stable")
inv = np.empty(T, np.int64); inv[order] = np.arange(T)
print("tile number - dispatch rank: percentiles 1/50/99:", np.percentile(np.arange(T) - inv, [1, 50, 99]))
for x in range(8):
    m = xcc == x
    print("xcd", x, "tiles", int(m.sum()), "first tiles", np.nonzero(m)[0][:6], "mean start us", round(float(s[m].mean()), 1), "last start", round(float(s[m].max()), 1))
# which publication does a tile's wait end on?  the latest publication among the tiles in front of it
pub_abs = s + pub
pm = np.maximum.accumulate(pub_abs)
need = np.concatenate([[0.0], pm[:-1]])
wait_end = s + we
print("wait end - latest publication in front, us, percentiles 1/50/99:", np.percentile(wait_end - need, [1, 50, 99]))
late = need - (s + wb)
print("latest publication in front - wait begin (positive: had to wait), us percentiles 10/50/90:", np.percentile(late, [10, 50, 90]))
k = np.argmax(pub_abs[:-1] >= pm[:-1] - 1e-9)
blockers = np.nonzero(pub_abs >= pm - 1e-9)[0]
print("tiles that were the latest publication so far:", len(blockers), "first", blockers[:12])
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'tile_times.npy'), w)
