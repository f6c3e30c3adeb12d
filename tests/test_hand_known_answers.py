"""Hand-derived known answers for the order-dependent tail (merge_trans / check_iden / split pieces / summary /
novel_exon.bed): the expected files under tests/golden/hand/ are literals written by hand from the reference's
rules (derivations: tests/golden/hand/README.md), NOT by the oracle.  The CPU suite holds the oracle CLI to them,
the GPU suite the HIP CLI, byte for byte."""
import filecmp
import os

import pytest

H = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hand")

# case -> (sub-command + options, input, uses the annotation, {output kind: expected file})
CASES = {
    "upd": (["update-gtf", "-l", "5"], "upd.sam", True,
            {"gtf": "upd.updated.gtf", "detail": "upd.detail.txt", "summary": "upd.summary.txt", "bed": "upd.novel_exon.bed"}),
    "upd_c": (["update-gtf", "-l", "5", "-c"], "upd.sam", True,
              {"gtf": "upd_c.updated.gtf", "detail": "upd.detail.txt", "summary": "upd_c.summary.txt", "bed": "upd_c.novel_exon.bed"}),
    "split": (["update-gtf", "-s", "-l", "5", "-J", "1", "-j", os.path.join(H, "split_sj.tab")], "split.sam", True,
              {"gtf": "split.updated.gtf", "detail": "split.detail.txt", "summary": "split.summary.txt", "bed": "split.novel_exon.bed"}),
    # -d: two annotation donors inside one read donor's tolerance (pair counting, update_gtf.c:736-752,770); -l 1 / 2 / 4 end rules (:629-696)
    "dis2": (["update-gtf", "-l", "5", "-d", "2"], "dis.sam", "dis_anno.gtf", {"gtf": "dis.updated.gtf", "detail": "dis2.detail.txt"}),
    "dis0": (["update-gtf", "-l", "5", "-d", "0"], "dis.sam", "dis_anno.gtf", {"gtf": "dis.updated.gtf", "detail": "dis0.detail.txt"}),
    "ends_l1": (["update-gtf", "-l", "1"], "ends.sam", "ends_anno.gtf", {"gtf": "ends_l1.updated.gtf", "detail": "ends.detail.txt"}),
    "ends_l2": (["update-gtf", "-l", "2"], "ends.sam", "ends_anno.gtf", {"gtf": "ends_l2.updated.gtf", "detail": "ends.detail.txt"}),
    "ends_l4": (["update-gtf", "-l", "4"], "ends.sam", "ends_anno.gtf", {"gtf": "ends_l4.updated.gtf", "detail": "ends.detail.txt"}),
    # -l 3 (both ends: overlap the terminal exon or overlap nothing) against -l 2 / -l 4, with reads whose chains contain each other (Q8)
    "ends3_l2": (["update-gtf", "-l", "2"], "ends3.sam", "ends_anno.gtf", {"gtf": "ends3_l2.updated.gtf", "detail": "ends3.detail.txt"}),
    "ends3_l3": (["update-gtf", "-l", "3"], "ends3.sam", "ends_anno.gtf", {"gtf": "ends3_l3.updated.gtf", "detail": "ends3.detail.txt"}),
    "ends3_l4": (["update-gtf", "-l", "4"], "ends3.sam", "ends_anno.gtf", {"gtf": "ends3_l4.updated.gtf", "detail": "ends3.detail.txt"}),
    # junction table without -s: a count below -J (with and without --use-multi; the SHORT option -M wants an argument, update_gtf.c:999),
    # a cursor row beyond the read (Q7: unsupported, no flag)
    "sj": (["update-gtf", "-l", "5", "-J", "3", "-j", os.path.join(H, "sj.tab")], "sj.sam", "sj_anno.gtf", {"gtf": "sj.updated.gtf", "detail": "sj.detail.txt"}),
    "sj_m": (["update-gtf", "-l", "5", "-J", "3", "--use-multi", "-j", os.path.join(H, "sj.tab")], "sj.sam", "sj_anno.gtf",
             {"gtf": "sj_m.updated.gtf", "detail": "sj_m.detail.txt"}),
    # CIGAR -> exons (gen_exon, bam2gtf.c:31-78: Q4 micro-exon drop and intron fusion, absorbed N / D, clips, =/X, XS strand, unmapped record)
    # under two threshold sets, through bam2gtf and through update-gtf's classification kernels
    "cigar_b2g": (["bam2gtf"], "cigar.sam", False, {"gtf": "cigar.bam2gtf.gtf"}),
    "cigar_b2g_t": (["bam2gtf", "-e", "10", "-i", "100", "-t", "5"], "cigar.sam", False, {"gtf": "cigar_t.bam2gtf.gtf"}),
    "cigar_upd": (["update-gtf", "-l", "5"], "cigar_m.sam", "cigar_anno.gtf", {"detail": "cigar.detail.txt"}),
    "cigar_upd_t": (["update-gtf", "-l", "5", "-e", "10", "-i", "100", "-t", "5"], "cigar_m.sam", "cigar_anno.gtf", {"detail": "cigar_t.detail.txt"}),
    # the annotation reader's quirks (gtf.c:317-326,468-521): tag found inside a longer tag, id / name fall-backs, a comment between a
    # transcript's rows, a chromosome that is not in the header, an empty line, rows cut behind byte 1 023
    "gtfq": (["update-gtf", "-l", "5"], "gtfq.sam", "gtfq_anno.gtf", {"gtf": "gtfq.updated.gtf", "detail": "gtfq.detail.txt"}),
    # records that are not coordinate sorted: the sweep's cursor only moves forward (update_gtf.c:792-802), a read in front of it finds nothing
    "uns": (["update-gtf", "-l", "5"], "uns.sam", "uns_anno.gtf", {"gtf": "uns.updated.gtf", "detail": "uns.detail.txt"}),
    # -m g: read-like transcripts from a GTF (read_gtf_trans, gtf.c:524-595): ids / names of the input kept, gene taken from the annotation, exons sorted,
    # strand flipped to the reference transcript's
    "mg": (["update-gtf", "-m", "g", "-b", os.path.join(H, "uns.sam"), "-l", "5"], "mg_reads.gtf", "uns_anno.gtf", {"gtf": "mg.updated.gtf", "detail": "mg.detail.txt"}),
    "uniq": (["unique-gtf"], "uniq.sam", False, {"gtf": "uniq.unique.gtf"}),
    "uniq_s": (["unique-gtf", "-s"], "uniq.sam", False, {"gtf": "uniq_s.unique.gtf"}),
}


def _argv(case, tmp_path):
    cmd, inp, with_anno, expect = CASES[case]
    out = {k: str(tmp_path / ("%s.%s" % (case, k))) for k in ("gtf", "detail", "summary", "bed")}
    args = list(cmd)
    if with_anno:
        args += ["-A", out["detail"], "-y", out["summary"], "-E", out["bed"]]
    args += [os.path.join(H, inp)]
    if with_anno:
        args += [os.path.join(H, with_anno if isinstance(with_anno, str) else "anno.gtf")]
    return args, out, expect


def _check(out, expect, case):
    for k, name in expect.items():
        want = os.path.join(H, name)
        if not filecmp.cmp(out[k], want, shallow=False):
            got = open(out[k]).read().split("\n")
            exp = open(want).read().split("\n")
            diff = [(i, g, e) for i, (g, e) in enumerate(zip(got, exp)) if g != e][:3]
            raise AssertionError("%s: %s differs from the hand-derived %s (lines %d vs %d); first differences: %r" %
                                 (case, k, name, len(got), len(exp), diff))


def test_literal_files_are_current():
    """The committed files are exactly what literal_files.py holds (nobody regenerated them from a program's output)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("literal_files", os.path.join(H, "literal_files.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for name, rows in m.FILES.items():
        assert open(os.path.join(H, name)).read() == "".join(r + "\n" for r in rows), name


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_matches_hand_derived_files(oracle, tmp_path, case):
    args, out, expect = _argv(case, tmp_path)
    assert oracle.run_cli(args, stdout_path=out["gtf"]) == 0
    _check(out, expect, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_hip_cli_matches_hand_derived_files(tmp_path, case):
    from lr2rmats_amd import hostlib
    args, out, expect = _argv(case, tmp_path)
    r = hostlib.run_cli(args, stdout_path=out["gtf"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    _check(out, expect, case)
