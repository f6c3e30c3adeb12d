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
