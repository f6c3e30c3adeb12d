"""Diagnostics on the GPU box: `lr2rmats filter` end to end on a synthetic SAM (reads with 1..4 alignments each), with the
CLI's own stage split (L2R_TIMING=1) and the decompressed output checked against the oracle on a prefix.  Not part of
the product.   tools/bench_filter.py [reads]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lr2rmats_amd import hostlib
from tests.test_filter import make_sam

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
d = tempfile.mkdtemp(prefix="l2r_filter_")
sam, out = os.path.join(d, "in.sam"), os.path.join(d, "out.bam")
t0 = time.perf_counter()
n_rec = make_sam(sam, n_reads, 77)
print("input: %d reads, %d records, %.1f MB of SAM (made in %.1f s)" % (n_reads, n_rec, os.path.getsize(sam) / 1e6, time.perf_counter() - t0))
for level in ("6", "1"):
    env = dict(os.environ, L2R_TIMING="1", L2R_BAM_LEVEL=level)
    t0 = time.perf_counter()
    with open(out, "wb") as fh:
        r = subprocess.run([hostlib.CLI_PATH, "filter", sam], stdout=fh, stderr=subprocess.PIPE, env=env)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    print("deflate level %s: wall %.2f s = %.2f M records/s, output %.1f MB" % (level, wall, n_rec / wall / 1e6, os.path.getsize(out) / 1e6))
    for line in r.stderr.decode().splitlines():
        if line.startswith("[timing]") or "Filtered" in line:
            print("   ", line)
if n_reads <= 50_000:
    from oracle import filter_oracle as fo
    want, keep = fo.expected_stream(sam)
    assert gzip.decompress(open(out, "rb").read()) == want
    print("output == oracle (%d records written)" % len(keep))
