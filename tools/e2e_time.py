"""Diagnostics: end-to-end wall clock of the C CLI (BAM + GTF in -> GTF/detail/summary/bed out) on config-3-like input."""
import os, subprocess, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lr2rmats_amd import synth, workload, hostlib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
cfg = dict(workload.CONFIGS['cfg3']); cfg['n_reads'] = N
anno = synth.make_annotation(cfg["anno_exons"], cfg["seed"], mean_tx_exons=cfg["n_exons"] + 1)
reads = synth.make_reads(anno, N, cfg["n_exons"], cfg["seed"] * 1000)
d = tempfile.mkdtemp(prefix="l2r_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
bam, gtf = os.path.join(d, "r.bam"), os.path.join(d, "a.gtf")
t0 = time.time(); synth.write_bam(reads, bam); anno.write_gtf(gtf)
print("inputs written in %.1f s: bam %.1f MB, gtf %.1f MB" % (time.time() - t0, os.path.getsize(bam) / 1e6, os.path.getsize(gtf) / 1e6), flush=True)
out = {k: os.path.join(d, k) for k in ("gtf", "detail", "summary", "bed")}
env = dict(os.environ); env["L2R_TIMING"] = "1"
runs = int(os.environ.get("E2E_RUNS", "1"))
for k in range(runs):
    t0 = time.time()
    r = subprocess.run([hostlib.BIN_PATH if hasattr(hostlib, "BIN_PATH") else os.path.join(os.path.dirname(hostlib.__file__), "bin", "lr2rmats"),
                        "update-gtf", "-l", "3", "-A", out["detail"], "-y", out["summary"], "-E", out["bed"], "-o", out["gtf"], bam, gtf],
                       env=env, stderr=subprocess.PIPE)
    dt = time.time() - t0
    if runs == 1:
        print(r.stderr.decode()[-1500:])
    else:
        print("run", k, " | ".join(l[len("[timing]"):].strip() for l in r.stderr.decode().splitlines() if l.startswith("[timing]")))
    print("rc", r.returncode, "wall %.2f s for %d reads; outputs:" % (dt, N), {k_: round(os.path.getsize(v) / 1e6, 1) for k_, v in out.items()}, flush=True)
