"""Diagnostics on the GPU box: the HIP path against the oracle on many small random workloads -- annotation shape (isoforms per
gene from 1 to 160: windows of every size, dictionary keys in parts), exons per read, ONT-like CIGARs, every -l level, the CIGAR
thresholds (-e / -i / -t), -d 0 .. 9, sparse and dense coverage; every case also checks the compacted accepted list against the
accepted reads of the full result, and every fourth runs with L2R_SEG_MAX=0 (the scans in launches of their own).  tools/fuzz_parity.py [rounds] [seed]; prints the first differing case and
exits 1.  Not part of the product."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lr2rmats_amd import capi, synth
from oracle import pyoracle as po

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
po.build()
bad = 0
for it in range(rounds):
    tpg = int(rng.choice([1, 3, 5, 12, 40, 70, 110, 160]))
    n_ex = int(rng.integers(2, 15))
    anno_exons = int(rng.choice([3000, 20000, 120000]))
    n_reads = int(rng.choice([3000, 40000, 150000]))
    ont = bool(rng.random() < 0.2)
    seed = int(rng.integers(1, 1 << 30))
    level = int(rng.integers(0, 6))
    prm = dict(full_level=level, min_exon=int(rng.choice([3, 3, 3, 0, 1, 25])), min_intron=int(rng.choice([3, 3, 40])), max_delet=int(rng.choice([50, 50, 4])),
               ss_dis=int(rng.choice([0, 0, 1, 2, 5, 9])))
    if it % 4 == 3:
        os.environ["L2R_SEG_MAX"] = "0"
    else:
        os.environ.pop("L2R_SEG_MAX", None)
    anno = synth.make_annotation(anno_exons, seed, mean_tx_exons=n_ex + 1, tx_per_gene=tpg)
    af = anno.in_file_order()
    reads = synth.make_reads(anno, n_reads, n_ex, seed + 7, ont=ont, micro_exons=3 if ont else 0, xs_conflict_frac=0.02 if ont else 0.0)
    # every third case with a junction table (-j): made from the oracle's first pass over the same reads like the generator's (a share of
    # the annotated junctions + of the reads' own), with -J 1 .. 3, -M and -s drawn -- the junction check inside k_tile (rows staged in LDS or
    # looked up in HBM), k_validate_sj behind the other kernels
    sj = None
    if it % 3 == 1 and not ont:
        base = po.classify_soa(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, af.tx_tid, af.tx_start, af.tx_end, af.tx_rev,
                               af.tx_ex_off, af.ex_start, af.ex_end, params=po.default_params(**prm))
        j = synth.make_junctions(af, base.ex_off, base.ex_start, base.ex_end, reads.tid, seed + 11, cover=float(rng.choice([0.3, 0.7, 0.95])))
        sj = (j.tid, j.don, j.acc, j.uniq, j.multi)
        prm.update(split_trans=int(rng.integers(0, 2)), min_sj_cnt=int(rng.integers(1, 4)), use_multi=int(rng.integers(0, 2)))
    e = capi.Engine(0)
    e.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
    e.set_junctions(sj)
    got = e.classify(reads, capi.default_params(**prm))
    acc = e.download_accepted()
    idx = np.nonzero((got.info & 128) != 0)[0]
    lens = (got.info[idx] >> 8).astype(np.int64)
    gidx = synth._ragged_gather_index(got.ex_off[idx], lens)
    acc_ok = (np.array_equal(acc.read_index, idx) and np.array_equal(np.diff(acc.ex_off), lens) and np.array_equal(acc.ex_start, got.ex_start[gidx])
              and np.array_equal(acc.ex_end, got.ex_end[gidx]) and np.array_equal(acc.ex_flag, got.ex_flag[gidx]))
    cnt = None
    try:
        import ctypes as C
        lib = capi.load_library()
        c = (C.c_longlong * 13)()
        lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.l2r_debug_counters(e.ctx, c, 13)
        cnt = (int(c[0]), int(c[1]), int(c[3]), int(c[12]))
    except Exception:
        pass
    e.close()
    want = po.classify_soa(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, af.tx_tid, af.tx_start, af.tx_end, af.tx_rev,
                           af.tx_ex_off, af.ex_start, af.ex_end, sj=sj, params=po.default_params(**prm))
    diffs = {} if acc_ok else {"accepted_list": 1}
    for name in ("ex_off", "ex_start", "ex_end", "ex_flag", "ref_tx"):
        a, b = getattr(got, name), getattr(want, name)
        if len(a) != len(b) or np.any(a != b):
            diffs[name] = int(np.sum(a[:min(len(a), len(b))] != b[:min(len(a), len(b))])) + abs(len(a) - len(b))
    d = np.nonzero((got.info & 0x7f) != (want.info & 0x7f))[0]
    if len(d):
        diffs["info"] = (len(d), d[:5].tolist())
    tag = "tpg %3d n_ex %2d anno %6d reads %6d ont %d sj %d seed %d %s (redo, keys in parts, tiles, wide list: %s)" % (tpg, n_ex, anno_exons, n_reads, ont, 0 if sj is None else len(sj[0]), seed, prm, cnt)
    print(("DIFF " if diffs else "ok   ") + tag, diffs if diffs else "", flush=True)
    bad += 1 if diffs else 0
    if diffs:
        break
print("fuzz_parity: %d rounds, %d with differences" % (it + 1, bad))
sys.exit(1 if bad else 0)
