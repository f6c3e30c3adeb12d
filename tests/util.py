"""Shared helpers of the parity tests: seeded cases and array comparison."""
from __future__ import annotations

import numpy as np

from lr2rmats_amd import synth


def make_case(seed, n_reads=20000, n_exons=5, anno_exons=20000, *, shuffle=False, long_tx=0, ont=False, micro=0,
              xs=0.0, unsorted=False, full_frac=0.7, nchr=24):
    anno = synth.make_annotation(anno_exons, seed, nchr=nchr, shuffle_within_gene=shuffle, long_tx_per_chrom=long_tx,
                                 mean_tx_exons=max(2, n_exons + 1))
    reads = synth.make_reads(anno, n_reads, n_exons, seed, full_frac=full_frac, ont=ont, micro_exons=micro,
                             xs_conflict_frac=xs, unsorted=unsorted)
    return anno, anno.in_file_order(), reads


def oracle_run(po, af, reads, params, sj=None):
    return po.classify_soa(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig,
                           af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end,
                           sj=sj, params=params)


def junction_table(anno, reads, base_result, seed, cover=0.8):
    j = synth.make_junctions(anno, base_result.ex_off, base_result.ex_start, base_result.ex_end, reads.tid, seed, cover=cover)
    return j, (j.tid, j.don, j.acc, j.uniq, j.multi)


def to_engine_params(capi, op):
    return capi.default_params(min_exon=op.min_exon, min_intron=op.min_intron, max_delet=op.max_delet, ss_dis=op.ss_dis,
                               end_dis=op.end_dis, full_level=op.full_level, split_trans=op.split_trans,
                               use_multi=op.use_multi, min_sj_cnt=op.min_sj_cnt, force_strand=op.force_strand,
                               single_exon_ovlp_frac=op.single_exon_ovlp_frac)


def assert_same_result(got, want, n_sj, split):
    """Bit-exact comparison of an engine Result with an oracle Result."""
    np.testing.assert_array_equal(got.ex_off, want.ex_off)
    np.testing.assert_array_equal(got.ex_start, want.ex_start)
    np.testing.assert_array_equal(got.ex_end, want.ex_end)
    np.testing.assert_array_equal(got.ex_flag, want.ex_flag)
    np.testing.assert_array_equal(got.ref_tx, want.ref_tx)
    n_ex = np.diff(want.ex_off)
    np.testing.assert_array_equal(got.info >> 8, n_ex.astype(np.uint32))
    np.testing.assert_array_equal(got.info & 0x7f, want.info & 0x7f)
    # ACCEPTED = what update_gtf.c:946-960 sends to novel_T
    w = want.info
    cand = ((w & 4) != 0) & ((w & 1) == 0) & ((w & 2) != 0)
    acc = cand & ((n_sj == 0) | ((w & 64) != 0) | bool(split))
    np.testing.assert_array_equal((got.info & 128) != 0, acc)


# ---- seeded whole-file fixtures (tests/golden/seeds.sha256, made by tools/make_seed_hashes.py) ------------------------------
SEED_FILES = ("updated.gtf", "detail.txt", "novel_exon.bed")
SEED_SETS = ("first", "second")          # `-l 3` and `-s -l 3 -J 1 -j SJ.tab -A -E -y`
SEEDS = (1, 2, 3, 4, 5)


def seed_inputs(po, seed, d):
    """BASELINE configs[1] with generator seed `seed` (100 k reads x 5 exons, 50 k-exon GTF) as files in directory `d`:
    reads.sam, anno.gtf and the junction table of the pipeline's second pass (80 % of the novel junctions of the first pass,
    from the ORACLE's first-pass classification: the inputs do not depend on the engine under test)."""
    import os
    anno = synth.make_annotation(50_000, seed)
    reads = synth.make_reads(anno, 100_000, 5, seed)
    af = anno.in_file_order()
    sam, gtf, tab = os.path.join(d, "reads.sam"), os.path.join(d, "anno.gtf"), os.path.join(d, "SJ.out.tab")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    base = oracle_run(po, af, reads, po.default_params(full_level=3))
    j, _ = junction_table(af, reads, base, seed, cover=0.8)
    j.write(tab)
    return sam, gtf, tab


def seed_args(which, sam, gtf, tab, out):
    """argv of `update-gtf` for option set `which`; out = {file name: path} for SEED_FILES + summary.txt."""
    extra = ["-l", "3"] if which == "first" else ["-s", "-l", "3", "-J", "1", "-j", tab]
    return ["update-gtf"] + extra + ["-A", out["detail.txt"], "-E", out["novel_exon.bed"], "-y", out["summary.txt"], "-o", out["updated.gtf"], sam, gtf]


def sha256_file(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def read_seed_hashes():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seeds.sha256")
    out = {}
    with open(path) as fh:
        for line in fh:
            if line.strip() and not line.startswith("#"):
                digest, name = line.split()
                out[name] = digest
    return out
