#!/usr/bin/env python3
"""bench.py -- lr2rmats update-gtf hot path on MI355X.

One "step" = one pass of the hot path (CIGAR -> exons, annotation sweep,
classification, per-read results in READ ORDER in HBM) over the rank's RESIDENT
read shard, and the step ends with the arrays l2r_download / l2r_device_view hand
to their consumers as they are.  It is NOT everything a fresh upload pays: the
one-kernel tile path reads a per-tile index of the records that l2r_upload_reads
makes once per read set (k_tile_index), and launches that a completed run has
shown to be empty (list kernels, the generic kernel) are dropped from later runs.
What ONE classification of fresh input costs the GPU -- index + first run -- is
measured beside the headline: `roofline.one_shot` (and `resident_input` for both
pipelines and both forms of the upload).
At N > 1 the shards are blocks of whole chromosomes, which is what
lr2rmats_amd/dist.py makes of a sorted input: the order-dependent host tail never
looks across chromosomes, so every rank merges and writes its own shard and the
step has NO collective (--exchange partitioned, default; the list sizes travel
once, after the timed region).  --exchange gathered times the other route of
dist.py (shards that cut through a chromosome, or -s with a junction table): the
compaction of the accepted records + their RCCL all-gatherv to every rank.

Workload at N=1: BASELINE.json configs[2] -- synthetic 10 M long reads, 8 exons/read
target, GENCODE-scale 1.5 M-exon GTF, pipeline option set `-l 3` (Snakefile:93).
Inputs are resident in HBM when the timed region starts.
--scaling strong (default) = BASELINE.json configs[3]: the SAME 10 M reads cut into N
chromosome-aligned shards (10 M / N reads per GPU).  --scaling weak: 10 M per GPU.
After the timed region rank 0 adds, at N=1: the dominant kernel's roofline figure
(HIP events on the engine's stream), the CPU oracle timed on one host core, and an
end-to-end run of the C CLI through files (BAM + GTF in, GTF / detail / summary / bed
out) on the same 10 M reads (--no-e2e skips it).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY_GBS = 6290.0


def cpu_baseline(af, reads, sample: int, level: int):
    """The oracle (CPU restatement = "port") timed on one host core over the first `sample` reads."""
    from oracle import pyoracle as po
    po.build()
    sub = reads.slice(0, min(sample, reads.n))
    p = po.default_params(full_level=level)
    t0 = time.perf_counter()
    res = po.classify_soa(sub.tid, sub.pos, sub.rev, sub.cig_off, sub.cig, af.tx_tid, af.tx_start, af.tx_end, af.tx_rev,
                          af.tx_ex_off, af.ex_start, af.ex_end, params=p)
    dt = time.perf_counter() - t0
    return sub.n / dt, sub, res, dt


def e2e_leg(af, reads, level: int):
    """End to end through files on the same reads: the C CLI reads a BAM and the GTF, classifies on the GPU and writes the
    updated GTF, detail.txt, summary.txt and novel_exon.bed (`lr2rmats update-gtf -l N -A -y -E -o`).  Wall clock of that
    one process; its own stage split (L2R_TIMING=1) is kept.  Inputs are written here first (not timed)."""
    import shutil
    import subprocess
    import tempfile
    from lr2rmats_amd import hostlib, synth
    d = tempfile.mkdtemp(prefix="l2r_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        bam, gtf = os.path.join(d, "reads.bam"), os.path.join(d, "anno.gtf")
        t0 = time.perf_counter()
        synth.write_bam_fast(reads, bam)
        af.write_gtf(gtf)
        t_in = time.perf_counter() - t0
        out = {k: os.path.join(d, k) for k in ("updated.gtf", "detail.txt", "summary.txt", "novel_exon.bed")}
        env = dict(os.environ)
        env["L2R_TIMING"] = "1"
        cmd = [hostlib.CLI_PATH, "update-gtf", "-l", str(level), "-A", out["detail.txt"], "-y", out["summary.txt"], "-E", out["novel_exon.bed"],
               "-o", out["updated.gtf"], bam, gtf]
        def one_run(run_env):
            # (every run writes NEW files: a run that overwrites the 3.9 GB of its predecessor pays 0.35-0.4 s when it closes them --
            #  tools/e2e_time.py with E2E_RUNS=3 shows it on every run but the first, with or without the annotation caches)
            for v in out.values():
                if os.path.exists(v):
                    os.remove(v)
            t0 = time.perf_counter()
            r = subprocess.run(cmd, env=run_env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            wall = time.perf_counter() - t0
            stages = {}
            for line in r.stderr.decode(errors="replace").splitlines():
                if line.startswith("[timing]"):
                    parts = line[len("[timing]"):].rsplit(None, 2)
                    if len(parts) == 3:
                        stages[parts[0].strip()] = float(parts[1])
            return r, wall, stages

        r, wall, stages = one_run(env)
        # the same command with the annotation caches on (L2R_ANNO_CACHE: parsed GTF + the engine's tables): the run that
        # fills them, then a run that reads them -- the pipeline's second update-gtf pass over the same GTF
        cenv = dict(env, L2R_ANNO_CACHE=os.path.join(d, "anno_cache"))
        _, cold_wall, _ = one_run(cenv)
        rw, warm_wall, warm_stages = one_run(cenv)
        # CPU side of the same command: the oracle's CLI (sequential C restatement with its own SAM / GTF readers and writers)
        # on a bounded sample of the same reads, as SAM text (it reads no BAM) -- "port", end to end, one core
        cpu_e2e = None
        parity_sample = None
        try:
            from oracle import pyoracle as po
            po.build()
            n_s = min(reads.n, 500_000)
            sam = os.path.join(d, "sample.sam")
            reads.slice(0, n_s).write_sam(sam)
            oo = {k: os.path.join(d, "cpu_" + k) for k in out}
            t0 = time.perf_counter()
            rc_o = po.run_cli(["update-gtf", "-l", str(level), "-A", oo["detail.txt"], "-y", oo["summary.txt"], "-E", oo["novel_exon.bed"],
                               "-o", oo["updated.gtf"], sam, gtf])
            cdt = time.perf_counter() - t0
            cpu_e2e = {"wall_s": round(cdt, 2), "rc": rc_o, "reads": n_s, "reads_per_s": round(n_s / cdt, 1), "cores": 1, "kind": "port",
                       "note": "oracle CLI, SAM text in (parses the whole GTF for the sample as well: the GTF stage does not shrink with the sample)"}
            # ... and the HIP CLI on the same sample file: the four graded files byte for byte against the oracle's
            go = {k: os.path.join(d, "gpu_" + k) for k in out}
            rs = subprocess.run([hostlib.CLI_PATH, "update-gtf", "-l", str(level), "-A", go["detail.txt"], "-y", go["summary.txt"], "-E", go["novel_exon.bed"],
                                 "-o", go["updated.gtf"], sam, gtf], env=dict(os.environ), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            same = {}
            for k in out:
                try:
                    with open(go[k], "rb") as fa, open(oo[k], "rb") as fb:
                        same[k] = fa.read() == fb.read()
                except OSError:
                    same[k] = False
            parity_sample = {"reads": n_s, "rc": rs.returncode, "files_identical": same, "all_identical": bool(rs.returncode == 0 and rc_o == 0 and all(same.values()))}
        except Exception as e:                                # the baseline must not take the line down
            cpu_e2e = {"error": str(e)[:200]}
        return {"parity_on_sample": parity_sample, "wall_s": round(wall, 3), "rc": r.returncode, "reads": reads.n, "reads_per_s": round(reads.n / wall, 1),
                "stages_s": stages, "cpu_port_same_command": cpu_e2e,
                "with_annotation_cache": {"filling_run_wall_s": round(cold_wall, 3), "warm_run_wall_s": round(warm_wall, 3), "rc": rw.returncode,
                                          "warm_run_reads_per_s": round(reads.n / warm_wall, 1), "warm_run_stages_s": warm_stages},
                "input_bytes": {"bam": os.path.getsize(bam), "gtf": os.path.getsize(gtf)},
                "outputs_bytes": {k: (os.path.getsize(v) if os.path.exists(v) else 0) for k, v in out.items()},
                "command": "lr2rmats update-gtf -l %d -A detail.txt -y summary.txt -E novel_exon.bed -o updated.gtf reads.bam anno.gtf" % level,
                "inputs_written_s": round(t_in, 1)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def pmc_traffic(kernels, config: str, n_reads: int):
    """HBM bytes per launch of the given kernels from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE need separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots"; tools/profile.sh + tools/pmc_csv_summary.py
    make the file).  Only reported when the file was measured on these kernels, this config and this read count.
    Returns ({kernel: bytes}, note)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json" if config == "cfg3" else "pmc_traffic_%s.json" % config)
    try:
        with open(path) as fh:
            t = json.load(fh)
    except (OSError, ValueError):
        return None, "no PMC file"
    if t.get("config") != config or t.get("reads") != n_reads:
        return None, "PMC file is for another config / read count"
    out = {}
    for k in kernels:
        ent = t.get("kernels", {}).get(k)
        if ent is None:
            return None, "PMC file has no kernel %s" % k
        out[k] = ent["hbm_bytes_per_launch"]
    return out, t.get("note", "profiles/pmc_traffic.json")


def second_pass(eng, af, reads, got, args, capi, workload, n_x):
    """The cold step under the pipeline's second option set: `-s -l 3 -J 1 -j SJ.tab` (the reference's junction
    check src/update_gtf.c:589-627,698-709,946-960) with a junction table made like the generator's (every annotated junction +
    the junctions of the reads, 80 % kept): k_validate_sj behind the classification, results + accepted list."""
    import numpy as np
    from lr2rmats_amd import synth
    if got is None:
        got = eng.download()
    t0 = time.perf_counter()
    sj = synth.make_junctions_fast(af, got.ex_off, got.ex_start, got.ex_end, reads.tid, seed=3, cover=0.8)
    t_tab = time.perf_counter() - t0
    eng.set_junctions((sj.tid, sj.don, sj.acc, sj.uniq, sj.multi))
    eng.set_params(capi.default_params(full_level=args.level, split_trans=1, min_sj_cnt=1))
    eng.set_outputs(capi.WANT_RESULTS | capi.WANT_ACCEPTED)
    eng.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
    eng.run(); eng.sync()
    tm = eng.run_timed(max(3, min(args.steps, 10)))
    _, _, m_acc, x_acc = eng.sizes()
    res = eng.download()
    unrel = int(((res.info & 16) != 0).sum())
    cand = int((((res.info & 4) != 0) & ((res.info & 1) == 0) & ((res.info & 2) != 0)).sum())
    n_sj = int(len(sj.don))
    abytes = workload.algorithmic_bytes(reads.n, int(reads.cig.shape[0]), n_x, af.n_tx, af.n_exons, n_sj)
    out = {"options": "-s -l %d -J 1 -j SJ.tab (results + accepted list)" % args.level, "junction_rows": n_sj, "junction_table_s": round(t_tab, 1),
           "ms_per_step": round(tm["total_ms"], 4), "reads_per_s": round(reads.n / (tm["total_ms"] * 1e-3), 1),
           "algorithmic_bytes_per_launch": abytes, "accepted_list_bytes": 20 * m_acc + 9 * x_acc,
           "frac_event_pass": round(abytes / (tm["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "candidates_checked": cand, "reads_with_unreliable_junction": unrel, "accepted_reads": m_acc, "accepted_exons": x_acc,
           "kernel_ms": {k.split(" ")[0]: round(v, 4) for k, v in tm["kernel_ms"].items()}}
    # back to the first option set (the e2e leg and anything behind this use their own engines, but leave this one as it was)
    eng.set_junctions(None)
    eng.set_params(capi.default_params(full_level=args.level))
    eng.set_outputs(capi.WANT_RESULTS | (capi.WANT_ACCEPTED if args.accepted else 0))
    eng.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
    return out


def gencode_leg(capi, workload, args):
    """A second workload beside the headline: config 3's reads and exon count against an annotation whose isoforms per gene are drawn
    heavy-tailed like a real one's (workload.CONFIGS["cfg3_gencode"]: log-normal, mean about 4.5, up to 200 -- most tiles on the 32-bit
    masks, isoform-rich loci on the 64-bit-mask and the chunked kernel, and, because reads follow the isoforms, sparse stretches that
    make small tiles).  The reference's sweep (src/update_gtf.c:796-822) has no such steps; this is where the mask kernels have theirs.
    An engine of its own; a 1 M-read slice is compared with the oracle bit for bit."""
    import ctypes as C
    import numpy as np
    cfg = dict(workload.CONFIGS["cfg3_gencode"])
    if args.reads:
        cfg["n_reads"] = args.reads
    af, reads = workload.make_rank_workload(cfg, 0, 1)
    eng = capi.Engine(0)
    try:
        eng.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
        eng.set_params(capi.default_params(full_level=args.level))
        eng.set_outputs(capi.WANT_RESULTS)
        eng.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
        eng.run(); eng.sync()
        # (like the headline: two identical regions of W + K steps back to back, the SECOND is the figure -- the chip's clock is still rising in
        #  the first, which is kept beside it)
        tm_first = eng.run_timed(args.warmup + args.steps)
        tm = eng.run_timed(args.warmup + args.steps)
        n_r, n_x, _, _ = eng.sizes()
        lib = capi.load_library()
        cnt = (C.c_longlong * 13)()
        lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.l2r_debug_counters(eng.ctx, cnt, 13)
        abytes = workload.algorithmic_bytes(n_r, int(reads.cig.shape[0]), n_x, af.n_tx, af.n_exons, 0)
        out = {"workload": "cfg3_gencode: %d reads x %.2f exons/read, %d-exon / %d-transcript GTF with log-normal isoforms per gene (max %d), update-gtf -l %d" % (
                   reads.n, n_x / max(n_r, 1), af.n_exons, af.n_tx, int(np.bincount(af.tx_gene).max()) if hasattr(af, "tx_gene") else -1, args.level),
               "ms_per_step": round(tm["total_ms"], 4), "reads_per_s": round(reads.n / (tm["total_ms"] * 1e-3), 1),
               "frac_event_pass": round(abytes / (tm["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "first_region_ms_per_step": round(tm_first["total_ms"], 4), "region": "the second of two regions of %d steps (HIP events on the engine's stream)" % (args.warmup + args.steps),
               "tiles": int(cnt[3]), "tiles_of_the_64_bit_mask_kernel": int(cnt[12]), "tiles_with_a_window_beyond_63_members": int(cnt[8]),
               "reads_on_the_redo_list": int(cnt[0]),
               "stage_ms": {k.split(" ")[0]: round(v, 4) for k, v in tm["kernel_ms"].items()}}
        if not args.no_cpu:
            from oracle import pyoracle as po
            po.build()
            sub = reads.slice(0, min(1_000_000, reads.n))
            want = po.classify_soa(sub.tid, sub.pos, sub.rev, sub.cig_off, sub.cig, af.tx_tid, af.tx_start, af.tx_end, af.tx_rev,
                                   af.tx_ex_off, af.ex_start, af.ex_end, params=po.default_params(full_level=args.level))
            got = eng.download()
            nx = int(want.ex_off[-1])
            out["parity_on_slice"] = {"reads": sub.n, "identical": bool(
                np.array_equal(got.ex_off[: sub.n + 1], want.ex_off) and np.array_equal(got.ex_start[:nx], want.ex_start)
                and np.array_equal(got.ex_end[:nx], want.ex_end) and np.array_equal(got.ex_flag[:nx], want.ex_flag)
                and np.array_equal(got.info[: sub.n] & 0x7f, want.info & 0x7f) and np.array_equal(got.ref_tx[: sub.n], want.ref_tx))}
        # ... and with a splice-site tolerance (-d 2): the 64-bit-mask and the chunked kernel probe within it as well
        eng.set_params(capi.default_params(full_level=args.level, ss_dis=2))
        eng.run(); eng.sync()
        tm2 = eng.run_timed(max(3, min(args.steps, 10)))
        lib.l2r_debug_counters(eng.ctx, cnt, 13)
        out["dis2"] = {"options": "-l %d -d 2" % args.level, "ms_per_step": round(tm2["total_ms"], 4), "over_the_d0_step": round(tm2["total_ms"] / tm["total_ms"], 3),
                       "reads_on_the_redo_list": int(cnt[0]), "stage_ms": {k.split(" ")[0]: round(v, 4) for k, v in tm2["kernel_ms"].items()}}
        return out
    finally:
        eng.close()


def _same(np, got, want, n_r):
    nx = int(want.ex_off[-1])
    return bool(np.array_equal(got.ex_off[: n_r + 1], want.ex_off) and np.array_equal(got.ex_start[:nx], want.ex_start)
                and np.array_equal(got.ex_end[:nx], want.ex_end) and np.array_equal(got.ex_flag[:nx], want.ex_flag)
                and np.array_equal(got.info[: n_r] & 0x7f, want.info & 0x7f) and np.array_equal(got.ref_tx[: n_r], want.ref_tx))


def ont_leg(capi, workload, args):
    """BASELINE configs[4] as ONE of its 8 GPUs sees it: rank 0's shard of the ONT-error-profile workload (2.5 M reads of ~330 CIGAR
    operations and ~14 exons, 3 micro-exons per read, against the 2 M-exon GTF) -- the long-CIGAR kernels (k_walk_slab_long + k_probe_slab).
    An engine of its own; a 150 k-read slice is compared with the oracle bit for bit."""
    import ctypes as C
    import numpy as np
    cfg = dict(workload.CONFIGS["cfg5"])
    cfg["n_reads"] = cfg["n_reads"] // 8
    af, reads = workload.make_rank_workload(cfg, 0, 8)
    eng = capi.Engine(0)
    try:
        eng.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
        eng.set_params(capi.default_params(full_level=args.level))
        eng.set_outputs(capi.WANT_RESULTS)
        eng.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
        eng.run(); eng.sync()
        tm = eng.run_timed(max(3, min(args.steps, 10)))
        n_r, n_x, _, _ = eng.sizes()
        lib = capi.load_library()
        cnt = (C.c_longlong * 13)()
        lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.l2r_debug_counters(eng.ctx, cnt, 13)
        abytes = workload.algorithmic_bytes(n_r, int(reads.cig.shape[0]), n_x, af.n_tx, af.n_exons, 0)
        out = {"workload": "BASELINE configs[4], rank 0 of 8: %d ONT-like reads x %.2f exons/read (%.1f CIGAR ops/read), %d-exon / %d-transcript GTF, update-gtf -l %d" % (
                   reads.n, n_x / max(n_r, 1), reads.cig.shape[0] / max(n_r, 1), af.n_exons, af.n_tx, args.level),
               "ms_per_step": round(tm["total_ms"], 4), "reads_per_s": round(reads.n / (tm["total_ms"] * 1e-3), 1),
               "algorithmic_bytes_per_launch": abytes,
               "frac_event_pass": round(abytes / (tm["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "reads_on_the_redo_list": int(cnt[0]), "tiles": int(cnt[3]),
               "kernel_ms": {k.split(" ")[0]: round(v, 4) for k, v in tm["kernel_ms"].items()}}
        if not args.no_cpu:
            from oracle import pyoracle as po
            po.build()
            sub = reads.slice(0, min(150_000, reads.n))
            want = po.classify_soa(sub.tid, sub.pos, sub.rev, sub.cig_off, sub.cig, af.tx_tid, af.tx_start, af.tx_end, af.tx_rev,
                                   af.tx_ex_off, af.ex_start, af.ex_end, params=po.default_params(full_level=args.level))
            out["parity_on_slice"] = {"reads": sub.n, "identical": _same(np, eng.download(), want, sub.n)}
        return out
    finally:
        eng.close()


def dis_leg(eng, af, reads, args, capi, workload, n_x, base_ms):
    """The headline workload with a splice-site tolerance: `-d 2` (src/update_gtf.c:717-779 with dis > 0; every mask kernel probes
    within the tolerance).  Same engine, same resident reads; a 150 k-read slice against the oracle."""
    import ctypes as C
    import numpy as np
    eng.set_params(capi.default_params(full_level=args.level, ss_dis=2))
    eng.run(); eng.sync()
    tm = eng.run_timed(max(3, min(args.steps, 10)))
    lib = capi.load_library()
    cnt = (C.c_longlong * 13)()
    lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.l2r_debug_counters(eng.ctx, cnt, 13)
    out = {"options": "-l %d -d 2" % args.level, "ms_per_step": round(tm["total_ms"], 4), "over_the_d0_step": round(tm["total_ms"] / base_ms, 3),
           "reads_on_the_redo_list": int(cnt[0]), "redo_share": round(cnt[0] / max(reads.n, 1), 6),
           "kernel_ms": {k.split(" ")[0]: round(v, 4) for k, v in tm["kernel_ms"].items()}}
    if not args.no_cpu:
        from oracle import pyoracle as po
        po.build()
        sub = reads.slice(0, min(150_000, reads.n))
        want = po.classify_soa(sub.tid, sub.pos, sub.rev, sub.cig_off, sub.cig, af.tx_tid, af.tx_start, af.tx_end, af.tx_rev,
                               af.tx_ex_off, af.ex_start, af.ex_end, params=po.default_params(full_level=args.level, ss_dis=2))
        out["parity_on_slice"] = {"reads": sub.n, "identical": _same(np, eng.download(), want, sub.n)}
    eng.set_params(capi.default_params(full_level=args.level))
    eng.run(); eng.sync()
    return out


def pipelines_leg(af, reads, args, capi, tile_ms, abytes):
    """What ONE classification of fresh input costs the GPU, next to the resident-input step of the headline.  The one-kernel tile path
    reads per-tile slot records and op statistics that l2r_upload_reads makes once per read set (k_tile_index: a function of the
    records alone, no option changes it) -- work on every record that the timed step of the headline does not contain.  Here, per
    pipeline and per form of the upload (with the reader's per-record CIGAR summaries, l2r_reads::cig_summary, the index touches no
    CIGAR; without them it walks every one), a FRESH upload is followed by its first run, five times:
        one_shot ms = k_tile_index (HIP events inside the upload, l2r_upload_index_ms) + the first run behind it (l2r_run_timed(1): with
        the launches of the list kernels and the generic kernel that later runs of the same upload drop once seen empty).
    The slab pipeline (L2R_PIPELINE=slab) has no index: everything happens inside every run."""
    import os
    import statistics
    old = os.environ.get("L2R_PIPELINE")
    out = {}
    sm = getattr(reads, "cig_summary", None)
    try:
        for name in ("slab", "tile"):
            os.environ["L2R_PIPELINE"] = name
            e = capi.Engine(0)
            e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
            e.set_params(capi.default_params(full_level=args.level)); e.set_outputs(capi.WANT_RESULTS)
            e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, cig_summary=sm)          # (first upload: allocations)
            e.run(); e.sync()
            forms = {}
            for form, summ in (("with_reader_summaries", sm), ("engine_walks_the_cigars", None)):
                if form == "with_reader_summaries" and sm is None:
                    continue
                idx, first, wall = [], [], []
                for _ in range(5):
                    t0 = time.perf_counter()
                    e.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, cig_summary=summ)
                    idx.append(e.upload_index_ms() if name == "tile" else 0.0)
                    first.append(e.run_timed(1)["total_ms"])
                    e.sync()
                    wall.append((time.perf_counter() - t0) * 1e3)
                i_ms, f_ms = statistics.median(idx), statistics.median(first)
                forms[form] = {"index_ms": round(i_ms, 4), "first_run_ms": round(f_ms, 4), "ms": round(i_ms + f_ms, 4),
                               "frac": round(abytes / ((i_ms + f_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "upload_plus_first_run_wall_ms": round(statistics.median(wall), 2)}
                if name == "slab":
                    break                                        # (no index: the form of the upload changes nothing)
            for _ in range(2):                                   # (like the headline: the second of two regions, the clock has settled)
                tm = e.run_timed(args.warmup + args.steps)
            out[name] = {"ms_per_step": round(tm["total_ms"], 4), "frac": round(abytes / (tm["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "one_shot": forms}
            e.close()
    finally:
        if old is None:
            os.environ.pop("L2R_PIPELINE", None)
        else:
            os.environ["L2R_PIPELINE"] = old
    out["note"] = ("ms_per_step: steady state on a resident upload (tile: the headline's path, its index made at upload; slab: everything inside the "
                   "step).  one_shot: index + first run of a fresh upload = what one l2r_classify costs the GPU; the product calls it once per read set")
    return out


def c_route_leg(world: int, dev_map, args, capi, workload, dev_index: int):
    """N > 1, rank 0, BEHIND the process group: the multi-GPU route of the C CLI itself (host/cmds.c, `L2R_GPUS=N lr2rmats update-gtf ...`: the
    parent forks a child per GPU before any HIP call, the children classify shards of the records and -- for `-s` with a junction table --
    gather their results on child 0 over RCCL / xGMI, l2r_xchg_*) on a bounded read set of the same configuration, as FRESH CHILD PROCESSES of
    this rank (nothing is exec'ed in place of a rank): the pipeline's second command with every output (the per-read results travel) and with
    `-o new.gtf -E bed` alone (the accepted reads travel, SURVEY 8(e)'s message), each against the one-GPU run of the same command, file by
    file.  The torch.distributed route above and this one are two implementations of the same exchange; a scaling run exercises both."""
    import filecmp
    import shutil
    import subprocess
    import tempfile
    from lr2rmats_amd import hostlib, synth
    cfg = dict(workload.CONFIGS[args.config]); cfg["n_reads"] = args.c_route_reads
    af, reads = workload.make_rank_workload(cfg, 0, 1)
    d = tempfile.mkdtemp(prefix="l2r_croute_", dir=os.environ.get("TMPDIR", "/tmp"))
    out = {"reads": reads.n, "children": world}
    try:
        bam, gtf, tab = os.path.join(d, "reads.bam"), os.path.join(d, "anno.gtf"), os.path.join(d, "SJ.out.tab")
        synth.write_bam_fast(reads, bam)
        af.write_gtf(gtf)
        e = capi.Engine(dev_index)
        e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
        e.set_junctions(None)
        res = e.classify(reads, capi.default_params(full_level=args.level))
        n_x = int(res.ex_start.shape[0])
        e.close()
        sj = synth.make_junctions_fast(af, res.ex_off, res.ex_start, res.ex_end, reads.tid, 5, cover=0.8)
        sj.write(tab)
        base = ["update-gtf", "-s", "-l", str(args.level), "-J", "1", "-j", tab]
        env_n = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID"):
            env_n.pop(k, None)
        env_n["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        env_1 = dict(env_n); env_1.pop("L2R_GPUS", None)
        env_n["L2R_GPUS"] = str(world)
        if dev_map:                                           # (tests: several ranks on one GPU -- the children share it too, RCCL cannot run there)
            env_n["L2R_GPU_MAP"] = ",".join(str(dev_map[k % len(dev_map)]) for k in range(world))
        for name, outs in (("per_read_results", ("updated.gtf", "detail.txt", "summary.txt", "novel_exon.bed")), ("accepted_reads_alone", ("updated.gtf", "novel_exon.bed"))):
            def cmd(tag):
                o = {k: os.path.join(d, "%s.%s.%s" % (name, tag, k)) for k in outs}
                c = [hostlib.CLI_PATH] + base + ["-o", o["updated.gtf"], "-E", o["novel_exon.bed"]]
                if "detail.txt" in o:
                    c += ["-A", o["detail.txt"], "-y", o["summary.txt"]]
                return c + [bam, gtf], o
            def run_bounded(argv, env, limit=240.0):
                # (a child of its own session, ended with its whole group at the limit: a route that has never met two physical GPUs must not
                #  be able to hold the line back)
                import signal
                pr = subprocess.Popen(argv, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
                try:
                    _, e = pr.communicate(timeout=limit)
                    return pr.returncode, e.decode(errors="replace")
                except subprocess.TimeoutExpired:
                    try:
                        os.killpg(pr.pid, signal.SIGKILL)
                    except OSError:
                        pass
                    _, e = pr.communicate()
                    return -999, e.decode(errors="replace") + "\n[bench.py] ended after %.0f s" % limit
            c1, o1 = cmd("one")
            rc1, _ = run_bounded(c1, env_1)
            cn, on = cmd("many")
            t0 = time.perf_counter()
            rcn, err = run_bounded(cn, env_n)
            wall = time.perf_counter() - t0
            for ln in err.splitlines():                          # what ncclCommInitRank saw on every rank: the first hardware run proves its world by it
                if "RCCL communicator" in ln or "gathered route" in ln or "L2R_GPUS" in ln:
                    print("bench.py c_route[%s]: %s" % (name, ln), file=sys.stderr)
            same = rc1 == 0 and rcn == 0 and all(os.path.exists(on[k]) and filecmp.cmp(o1[k], on[k], shallow=False) for k in outs)
            ranks_seen = sorted({ln.split("rank ")[1].split(" ")[0] for ln in err.splitlines() if "RCCL communicator: rank " in ln})
            n_acc = int(((res.info & 128) != 0).sum())
            out[name] = {"wall_s": round(wall, 3), "rc": [rc1, rcn], "files_identical": bool(same),
                         "exchange": "rccl" if "exchange: RCCL" in err else ("shm" if "exchange: shared memory" in err else "none (partitioned or one child)"),
                         "rccl_ranks_reporting": ranks_seen,
                         # an upper bound of what travels to child 0 (its own shard does not): 12 bytes per read + 9 per exon, or the accepted records alone
                         "bytes_to_rank0_at_most": (12 * reads.n + 9 * n_x) if name == "per_read_results" else None,
                         "stderr_tail": None if same else err[-600:]}
        out["accepted_reads_before_the_junction_check"] = n_acc
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def launcher_argv(args, port: int):
    """The command `bench.py --gpus N` starts when no launcher has set WORLD_SIZE: the shape the driver uses itself."""
    passed = [a for a in sys.argv[1:] if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + passed


def self_launch(args) -> int:
    import socket
    import subprocess
    with socket.socket() as so:                       # a free port for the rendezvous
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = launcher_argv(args, port)
    if args.dry_launch:
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in r.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    elif r.returncode == 0:
        print("bench.py: the launcher returned 0 without a JSON line", file=sys.stderr)
        return 4
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--reads", type=int, default=0, help="override reads per rank")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=10_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--exchange", choices=("partitioned", "gathered"), default="partitioned")
    ap.add_argument("--accepted", action="store_true", help="also compact the accepted-novel list (always on with --exchange gathered)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--emulate-world", type=int, default=0, help="diagnostics, one GPU: run the shard rank 0 would own in a strong-scaling run of this many GPUs")
    ap.add_argument("--no-second-pass", action="store_true", help="skip the pipeline's second option set (-s -l 3 -J 1 -j SJ.tab) at N=1")
    ap.add_argument("--no-gencode", action="store_true", help="skip the second workload (heavy-tailed isoforms per gene) at N=1")
    ap.add_argument("--no-ont", action="store_true", help="skip the ONT shard (BASELINE configs[4], rank 0 of 8) at N=1")
    ap.add_argument("--no-dis", action="store_true", help="skip the -d 2 leg at N=1")
    ap.add_argument("--no-pipelines", action="store_true", help="skip the tile / slab comparison (the step without the upload's op index) at N=1")
    ap.add_argument("--no-c-route", action="store_true", help="N > 1: skip the C CLI's own multi-GPU run (L2R_GPUS=N) behind the process group")
    ap.add_argument("--c-route-reads", type=int, default=1_000_000, help="N > 1: reads of the C route's leg")
    ap.add_argument("--dry-launch", action="store_true", help="print the launcher command a plain `bench.py --gpus N` would start, and exit")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python3 bench.py --gpus N` without a launcher: start one process per GPU as a CHILD (nothing here has touched HIP yet,
        # and nothing is exec'ed), relay rank 0's JSON line and the launcher's return code
        sys.exit(self_launch(args))
    if args.dry_launch:
        print(json.dumps({"launch": None, "note": "one process: no launcher needed"}))
        sys.exit(0)
    if world != args.gpus:
        print("bench.py: --gpus %d under WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist
    from lr2rmats_amd import capi, workload

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(3)
    # L2R_BENCH_DEVICES="0,0": ranks -> devices (tests put two ranks on one GPU; RCCL wants a GPU per rank, so such a run needs
    # L2R_BENCH_BACKEND=gloo, whose collectives take host tensors)
    dev_map = [int(x) for x in os.environ.get("L2R_BENCH_DEVICES", "").split(",") if x.strip() != ""]
    dev_index = dev_map[local_rank % len(dev_map)] if dev_map else local_rank
    backend = os.environ.get("L2R_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    coll_dev = device if backend == "nccl" else torch.device("cpu")      # where the small collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            print("bench.py: process group of %d ranks under --gpus %d" % (dist.get_world_size(), args.gpus), file=sys.stderr)
            sys.exit(2)
    rccl_world = dist.get_world_size() if world > 1 else 1

    cfg = dict(workload.CONFIGS[args.config])
    if args.reads:
        cfg["n_reads"] = args.reads
    w_world = args.emulate_world if (args.emulate_world > 1 and world == 1) else world
    if args.scaling == "strong":
        # configs[3]: one read set of cfg["n_reads"] cut into `world` chromosome-aligned shards
        cfg["n_reads"] = cfg["n_reads"] // w_world + (1 if rank < cfg["n_reads"] % w_world else 0)
    af, reads = workload.make_rank_workload(cfg, rank, w_world)
    # what a reader knows of every record's CIGAR while it converts it (host/aln_reader.c for SAM / BAM input; here the same C loop over the
    # generator's arrays): the engine's upload then makes its tile index without touching a CIGAR
    from lr2rmats_amd import hostlib
    reads.cig_summary = hostlib.cigar_summaries(reads.cig_off, reads.cig)

    eng = capi.Engine(dev_index)
    eng.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
    eng.set_params(capi.default_params(full_level=args.level))
    # the step produces the per-read results (SURVEY.md 8(d) bytes); the compacted accepted list only where the exchange sends it
    gather_headline = world > 1 and args.exchange == "gathered"
    eng.set_outputs(capi.WANT_RESULTS | (capi.WANT_ACCEPTED if (gather_headline or args.accepted) else 0))
    eng.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, first_read_index=rank * reads.n, cig_summary=reads.cig_summary)

    gathered = {}

    def exchange():
        # counts of (records, exons) per rank, then five padded all-gathers of engine-owned HBM (workload.all_gatherv)
        _, _, m, x = eng.sizes()
        v = eng.device_view()
        cnt = torch.tensor([m, x], dtype=torch.int64, device=coll_dev)
        allc = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(allc, cnt)
        allc = [c.tolist() for c in allc]
        # (the exon arrays are chunked per tile, see include/lr2rmats_hip.h: the per-record offsets travel with the records)
        parts = (("rec", v.acc_rec, 16, 0), ("ex_off", v.acc_ex_off, 4, 0), ("ex_start", v.acc_ex_start, 4, 1), ("ex_end", v.acc_ex_end, 4, 1),
                 ("ex_flag", v.acc_ex_flag, 1, 1))
        for name, ptr, width, which in parts:
            mine = workload.device_bytes(ptr, (m if which == 0 else x) * width, device)
            outs, _ = workload.all_gatherv(mine, counts=[c[which] * width for c in allc])
            gathered[name] = outs
        return sum(c[0] for c in allc), sum(c[1] for c in allc)

    def timed_region(steps: int, warmup: int, gather: bool):
        """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; max over ranks.
        One step = one pass of the hot path over the resident batch: asynchronous launches on the engine's stream.  Passes are
        independent (each overwrites the results of the one before) and stream order keeps them apart, so the host does not
        wait between them: the region ends with one synchronisation of the engine's stream + device.  Only the gathered
        exchange needs the sizes of a pass on the host, and waits for them."""
        def step():
            eng.run()
            if gather:
                eng.sync()
                return exchange()
            return None
        for _ in range(warmup):
            step()
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for _ in range(steps):
            last = step()
        eng.sync()                  # (the engine launches on a stream of its own: torch.cuda.synchronize() alone would do, this says it)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, last

    # Two identical regions, W warm-up steps + K timed steps each, back to back.  The chip raises its clock over the first ~15 ms of sustained
    # work (tools/ramp.py: 0.658 -> 0.613 ms per step over the first ~25 steps on one box) and the contract's W = 5 warm-up steps are 3 ms of
    # that: the FIRST region runs on the ramp and is reported as `roofline.cold_start`; the SECOND one is the line's `value` / `ms_per_step` /
    # `roofline.frac` -- a read set is classified in a stream of such steps, on the clock the chip settles at.
    dt_cold, _ = timed_region(args.steps, args.warmup, gather_headline)
    dt, last = timed_region(args.steps, args.warmup, gather_headline)

    n_r, n_x, n_acc, n_acc_x = eng.sizes()
    per_rank = [[reads.n, n_x, n_acc, n_acc_x]]
    if world > 1:
        # reads / exons / accepted records of every rank (outside the timed region): the first hardware run explains itself
        tr = torch.tensor(per_rank[0], dtype=torch.int64, device=coll_dev)
        allr = [torch.zeros_like(tr) for _ in range(world)]
        dist.all_gather(allr, tr)
        per_rank = [t.tolist() for t in allr]
    total_reads = sum(p[0] for p in per_rank)
    value = total_reads * args.steps / dt

    # the OTHER route of dist.py in the same line (N > 1): a second, shorter timed region with the exchange the headline did not use,
    # so that one scaling run reports both (partitioned: no collective in a step; gathered: compaction + RCCL all-gatherv of the accepted list)
    other = None
    if world > 1:
        o_gather = not gather_headline
        eng.set_outputs(capi.WANT_RESULTS | (capi.WANT_ACCEPTED if (o_gather or args.accepted) else 0))
        o_steps = max(2, min(args.steps, 5))
        o_dt, o_last = timed_region(o_steps, 1, o_gather)
        _, _, o_acc, o_acc_x = eng.sizes()
        tr = torch.tensor([o_acc, o_acc_x], dtype=torch.int64, device=coll_dev)
        allr = [torch.zeros_like(tr) for _ in range(world)]
        dist.all_gather(allr, tr)
        o_per = [t.tolist() for t in allr]
        other = {"exchange": "gathered" if o_gather else "partitioned", "steps": o_steps, "warmup": 1,
                 "ms_per_step": round(o_dt / o_steps * 1e3, 4), "value": round(total_reads * o_steps / o_dt, 1), "unit": "reads/s",
                 "exchange_bytes_per_rank_per_step": [0 if not o_gather else 20 * q[0] + 9 * q[1] for q in o_per],
                 "gathered_records_exons": list(o_last) if o_last else None}
        eng.set_outputs(capi.WANT_RESULTS | (capi.WANT_ACCEPTED if (gather_headline or args.accepted) else 0))

    out = None
    if rank == 0:
        # `achieved` = the path's algorithmic bytes over the time of one step OF THE TIMED REGION ABOVE -- the clock the
        # headline value comes from.  The path is two kernels of comparable length (walk: CIGAR -> exons; probe: annotation
        # window + site probes -> verdicts + the read-order write-out) with the redo list behind; no single one of them moves the path's
        # algorithmic bytes.  A second pass with HIP events on the engine's stream (l2r_run_timed: back-to-back cold runs, then per-stage
        # events) gives the kernels one by one; its own total is kept beside the headline as `frac_event_pass`.
        tm = eng.run_timed(max(3, min(args.steps, 10)))
        stage = tm["stage_ms"]
        kern = {k.split(" ")[0]: v for k, v in tm["kernel_ms"].items()}
        live = {k: v for k, v in kern.items() if v > 0.02 * tm["total_ms"]}
        dom = max(live, key=lambda k: live[k])
        abytes = workload.algorithmic_bytes(n_r, int(reads.cig.shape[0]), n_x, af.n_tx, af.n_exons, 0)
        step_s = dt / args.steps
        ach = abytes / step_s / 1e9
        ach_ev = abytes / (tm["total_ms"] * 1e-3) / 1e9
        launched = [k for k in kern if k not in ("k_validate_sj", "k_scan_accepted", "k_gather_accepted") or args.accepted or gather_headline]
        per_kernel, traffic_note = pmc_traffic(sorted(live), args.config, reads.n)
        traffic = None if per_kernel is None else int(sum(per_kernel.values()))
        roof = {"bound": "hbm", "kernel": " + ".join(sorted(launched, key=lambda k: -kern[k])) + " (every kernel of a step on resident input)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "frac_of_measured_copy_peak": round(ach / HBM_COPY_GBS, 4),
                "clock": "the timed region of this line (the second of two identical W + K regions): algorithmic bytes / ms_per_step",
                "cold_start": {"ms_per_step": round(dt_cold / args.steps * 1e3, 4), "frac": round(abytes / (dt_cold / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                               "note": "the same W warm-up + K timed steps right in front of the headline region, from an idle chip (its clock is still rising: tools/ramp.py)"},
                "frac_event_pass": round(ach_ev / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_note,
                "traffic_over_algorithmic": None if traffic is None else round(traffic / abytes, 3),
                "algorithmic_bytes_per_launch": abytes,
                "all_kernels_ms": round(tm["total_ms"], 4),
                "all_kernels_achieved_GBs": round(ach_ev, 1),
                "step_is_cold": False, "results_layout": "read order (ex_off / ex_start / ex_end / ex_flag / info / ref_tx as l2r_download copies them)",
                "dominant_kernel": {"name": dom, "ms": round(live[dom], 4),
                                    "hbm_bytes_per_launch": None if per_kernel is None else per_kernel[dom],
                                    "hbm_GBs": None if per_kernel is None else round(per_kernel[dom] / (live[dom] * 1e-3) / 1e9, 1)},
                "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
                "kernel_hbm_bytes": per_kernel,
                "stage_ms": {k: round(v, 4) for k, v in stage.items()}}
        # the same cold step with the compacted accepted-novel list as well (K5: wave ballot / prefix-sum compaction of the accepted
        # reads, what `update-gtf ... > new.gtf` and the multi-GPU all-gatherv consume)
        with_accepted = None
        if world == 1:
            eng.set_outputs(capi.WANT_RESULTS | capi.WANT_ACCEPTED)
            eng.run(); eng.sync()
            tma = eng.run_timed(max(3, min(args.steps, 10)))
            _, _, m_acc, x_acc = eng.sizes()
            with_accepted = {"ms_per_step": round(tma["total_ms"], 4), "reads_per_s": round(reads.n / (tma["total_ms"] * 1e-3), 1),
                             "accepted_reads": m_acc, "accepted_exons": x_acc, "extra_bytes": 20 * m_acc + 9 * x_acc,
                             "kernel_ms": {k.split(" ")[0]: round(v, 4) for k, v in tma["kernel_ms"].items()}}
            eng.set_outputs(capi.WANT_RESULTS | (capi.WANT_ACCEPTED if args.accepted else 0))
            eng.run(); eng.sync()
        cpu = None
        got = None
        if world == 1 and not args.no_cpu:
            rate, sub, ores, cdt = cpu_baseline(af, reads, args.cpu_sample, args.level)
            # the same launch also serves as a parity spot check of the sample (bit exact)
            got = eng.download()
            nx = int(ores.ex_off[-1])
            ok = (np.array_equal(got.ex_off[: sub.n + 1], ores.ex_off) and np.array_equal(got.ex_start[:nx], ores.ex_start)
                  and np.array_equal(got.ex_end[:nx], ores.ex_end) and np.array_equal(got.ex_flag[:nx], ores.ex_flag)
                  and np.array_equal(got.info[: sub.n] & 0x7f, ores.info & 0x7f) and np.array_equal(got.ref_tx[: sub.n], ores.ref_tx))
            cpu = {"value": round(rate, 1), "unit": "reads/s", "cores": 1, "kind": "port",
                   "sample": "first %d reads of the same workload, oracle/ (sequential C restatement: comparison core only, "
                             "structure-of-arrays in, no parsing, no lists), %.1f s" % (sub.n, cdt),
                   "parity_on_sample": bool(ok),
                   # the reference itself cannot be built or shipped (htslib); its survey-time probe on a 2.1 GHz Xeon, BASELINE.md section 2
                   "reference_core_reads_per_s": 3.0e5, "reference_e2e_reads_per_s": 3.2e4,
                   "reference_note": "BASELINE.md section 2 (survey-time probe of the unmodified reference, other host; indicative only)"}
        # the pipeline's SECOND option set (Snakefile:170): -s -l 3 -J 1 -j SJ.tab, i.e. the junction check
        # (src/update_gtf.c:946-960) behind the classification, with the generator's 80 %-cover junction table, per-read results AND the
        # accepted list (what -A / -E / -y and the GTF on stdout consume)
        second = None
        if world == 1 and not args.no_second_pass:
            try:
                second = second_pass(eng, af, reads, got, args, capi, workload, n_x)
            except Exception as e:                                # (must not take the line down)
                second = {"error": str(e)[:300]}
        dis2 = None
        if world == 1 and not args.no_dis and args.config == "cfg3":
            try:
                dis2 = dis_leg(eng, af, reads, args, capi, workload, n_x, tm["total_ms"])
            except Exception as e:                                # (must not take the line down)
                dis2 = {"error": str(e)[:300]}
        ont = None
        if world == 1 and not args.no_ont and args.config == "cfg3":
            try:
                ont = ont_leg(capi, workload, args)
            except Exception as e:                                # (must not take the line down)
                ont = {"error": str(e)[:300]}
        gencode = None
        if world == 1 and not args.no_gencode:
            try:
                gencode = gencode_leg(capi, workload, args)
            except Exception as e:                                # (must not take the line down)
                gencode = {"error": str(e)[:300]}
        pipes = None
        if world == 1 and args.config == "cfg3" and not args.no_pipelines:
            try:
                pipes = pipelines_leg(af, reads, args, capi, tm["total_ms"], abytes)
            except Exception as e:                                # (must not take the line down)
                pipes = {"error": str(e)[:300]}
        if pipes and "tile" in pipes and "slab" in pipes:
            # the cost of ONE classification of fresh input (index + first run), beside the resident-input step that `frac` is: which form
            # of the figure clears the north star's 0.40 and which does not is said here, not left to the reader
            t_os, s_os = pipes["tile"]["one_shot"], pipes["slab"]["one_shot"]
            best = t_os.get("with_reader_summaries") or t_os.get("engine_walks_the_cigars")
            roof["one_shot"] = {"ms": best["ms"], "frac": best["frac"], "index_ms": best["index_ms"], "first_run_ms": best["first_run_ms"],
                                "upload": "with the reader's per-record CIGAR summaries (l2r_reads::cig_summary)" if "with_reader_summaries" in t_os else "the engine walks the CIGARs",
                                "tile_engine_walks_the_cigars": t_os.get("engine_walks_the_cigars"),
                                "slab_pipeline_no_index": next(iter(s_os.values()), None),
                                "note": "k_tile_index at upload + the first run of that upload: what one l2r_classify of fresh records costs the GPU; `frac` above is the step on resident input"}
        e2e = None
        if world == 1 and not args.no_e2e:
            e2e = e2e_leg(af, reads, args.level)
        out = {
            "metric": "long-read alignments classified/sec; achieved HBM GB/s vs roofline",
            "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # (the line's value / ms_per_step / roofline.frac come from the SECOND of two identical W + K regions: roofline.cold_start is the first)
            "headline_region": 2, "warmup_effective_steps": 2 * args.warmup + args.steps,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: synthetic %d long reads (%d per GPU) x %.2f exons/read (%.1f CIGAR ops/read), "
                                   "%d-exon / %d-transcript GTF, update-gtf -l %d" % (
                                       ({"cfg2": 1, "cfg3": 2, "cfg5": 4}.get(args.config, 2) + (1 if (world > 1 and args.scaling == "strong" and args.config == "cfg3") else 0)),
                                       total_reads, reads.n, n_x / max(n_r, 1), reads.cig.shape[0] / max(n_r, 1), af.n_exons, af.n_tx, args.level),
                       "reads_per_gpu": reads.n, "total_reads": total_reads, "accepted_reads_rank0": n_acc,
                       "exchange": "none (1 GPU)" if world == 1 else (
                           "partitioned: chromosome-aligned shards merge and write on their own rank; no collective inside a step"
                           if not gather_headline else
                           "gathered: RCCL all-gatherv (padded all_gather_into_tensor) of %s accepted records / %s exons to every rank" % (last[0], last[1])),
                       "parallelism": "reads sharded over %d GPU(s), annotation replicated" % world,
                       "rccl_world_size": rccl_world, "collective_backend": backend if world > 1 else None,
                       "reads_per_rank": [p[0] for p in per_rank], "exons_per_rank": [p[1] for p in per_rank],
                       # bytes a rank SENDS per step: the partitioned route exchanges nothing inside a step (16 summary counters once,
                       # behind the timed region); the gathered route all-gathers its accepted records (16 + 4 bytes each) and their exons (9 bytes each)
                       "exchange_bytes_per_rank_per_step": [0 if not gather_headline else 20 * p[2] + 9 * p[3] for p in per_rank]},
            "roofline": roof,
            "other_exchange": other,
            "with_accepted": with_accepted,
            "second_pass": second,
            "isoform_rich": gencode,
            "dis2": dis2,
            "ont_shard": ont,
            "resident_input": pipes,
            "cpu_baseline": cpu,
            "e2e": e2e,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if world > 1 and rank == 0 and out is not None and not args.no_c_route:
        try:
            out["c_route"] = c_route_leg(world, dev_map, args, capi, workload, dev_index)
        except Exception as e:                                    # (must not take the line down)
            out["c_route"] = {"error": str(e)[:300]}
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
