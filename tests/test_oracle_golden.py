"""CPU: the oracle against the known-answer vectors recorded from the reference."""
import filecmp
import os

import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "toy")

# SURVEY.md Appendix D.2, literal (tabs): `update-gtf -l 3` detail rows
D2_ROWS = [
    "r1_perfect\tchr1\t-\t1\tENSG00000186891\tTNFRSF18\t4\t1138888,1139779,1140750,1141765\t1139340,1139866,1140872,1141951\t0\tNA\t1\t1\t0\tNA\t0\tNA\t",
    "r4_single\tchr1\t-\t2\tNA\tNA\t1\t1138900\t1139299\t1\t0\t0\tNA\t0\tNA\t0\tNA\t",
    "r3_skip\tchr1\t-\t1\tENSG00000186891\tTNFRSF18\t3\t1138888,1139779,1141765\t1139340,1139866,1141951\t0\tNA\t1\t1\t1\t1\t0\tNA\t",
    "r2_trunc\tchr1\t-\t0\tENSG00000186891\tTNFRSF18\t3\t1139779,1140750,1141765\t1139866,1140872,1141951\t0\tNA\t0\tNA\t0\tNA\t0\tNA\t",
    "r5_unrec\tchr1\t+\t2\tNA\tNA\t2\t2000000,2000600\t2000099,2000699\t2\t0,1\t2\t0,1\t1\t0\t0\tNA\t",
]


def _run(oracle, tmp_path, extra, tag):
    out = {k: str(tmp_path / (tag + "." + k)) for k in ("gtf", "detail", "summary", "bed")}
    rc = oracle.run_cli(["update-gtf"] + extra + ["-A", out["detail"], "-y", out["summary"], "-E", out["bed"],
                                                os.path.join(G, "toy.sam"), os.path.join(G, "original.gtf")], stdout_path=out["gtf"])
    assert rc == 0
    return out


@pytest.mark.parametrize("level", ["3", "5"])
def test_detail_rows_match_reference_record(oracle, tmp_path, level):
    out = _run(oracle, tmp_path, ["-l", level], "l" + level)
    lines = open(out["detail"]).read().split("\n")
    assert lines[0].startswith("ReadName\tchr\tstrand\tNovel\tGeneID")
    assert lines[1:6] == D2_ROWS


def test_level3_files(oracle, tmp_path):
    out = _run(oracle, tmp_path, ["-l", "3"], "l3")
    gtf = open(out["gtf"]).read().split("\n")
    tr = [l for l in gtf if "\ttranscript\t" in l]
    assert [l.split('transcript_id "')[1].split('"')[0] for l in tr] == ["r1_perfect", "r3_skip"]
    assert all(l.endswith('transcript_cov "1";') for l in tr)
    ex = [int(l.split("\t")[3]) for l in gtf[1:5]]
    assert ex == sorted(ex, reverse=True)                       # '-' strand: exons descending
    assert os.path.getsize(out["bed"]) == 0
    summ = dict(l.rstrip("\n").split("\t") for l in open(out["summary"]) if "\t" in l)
    assert summ["Updated_Genes"] == "1" and summ["Added_Novel_Transcripts"] == "2"
    assert summ["Added_Novel_Sites"] == "1" and summ["Added_Novel_Splice_Junctions"] == "1"
    assert summ["Known_Transcripts_from_BAM"] == "1" and summ["Novel_Transcript_from_BAM"] == "2"
    assert summ["Novel_Transcript_from_BAM_with_All_Reliable_Junction"] == "2"
    assert summ["Unrecognized_Transcript_from_BAM"] == "2"
    for k, ext in (("gtf", "updated.gtf"), ("detail", "detail.txt"), ("summary", "summary.txt"), ("bed", "novel_exon.bed")):
        assert filecmp.cmp(out[k], os.path.join(G, "expect_l3." + ext), shallow=False)


def test_sj_cases(oracle, tmp_path):
    # D.2: supporting junction -> same two transcripts
    o = _run(oracle, tmp_path, ["-s", "-l", "3", "-J", "1", "-j", os.path.join(G, "sj_support.tab")], "sup")
    assert filecmp.cmp(o["gtf"], os.path.join(G, "expect_l3.updated.gtf"), shallow=False)
    # D.2: junction table far away -> only r3_skip.split.0, printed with tid/start/end/strand zeroed (Q2, Q7)
    o = _run(oracle, tmp_path, ["-s", "-l", "3", "-J", "1", "-j", os.path.join(G, "sj_far.tab")], "far")
    g = open(o["gtf"]).read().split("\n")
    assert g[0].startswith('chr1\tlr2rmats\ttranscript\t0\t0\t.\t+\t.\tgene_id "ENSG00000186891"; transcript_id "r3_skip.split.0";')
    assert [l.split("\t")[3:5] + [l.split("\t")[6]] for l in g[1:4]] == [["1138888", "1139340", "-"], ["1139779", "1139866", "-"], ["1141765", "1141951", "-"]]
    assert len([l for l in g if l]) == 4
    summ = dict(l.rstrip("\n").split("\t") for l in open(o["summary"]) if "\t" in l)
    assert summ["Added_Novel_Partial-read_Transcripts"] == "1"
    assert summ["Novel_Transcript_from_BAM_with_Unreliable_Junction"] == "2"
    assert all(l.split("\t")[15:17] == ["0", "NA"] for l in open(o["detail"]).read().split("\n")[1:6])
    # D.2: a different junction -> only r1_perfect; r3_skip gets unreliable junction 1
    o = _run(oracle, tmp_path, ["-s", "-l", "3", "-J", "1", "-j", os.path.join(G, "sj_other.tab")], "oth")
    g = [l for l in open(o["gtf"]).read().split("\n") if "\ttranscript\t" in l]
    assert len(g) == 1 and 'transcript_id "r1_perfect"' in g[0]
    row = open(o["detail"]).read().split("\n")[3].split("\t")
    assert row[0] == "r3_skip" and row[15:17] == ["1", "1"]
    for tag, name in (("sup", "sj_support"), ("far", "sj_far"), ("oth", "sj_other")):
        for k, ext in (("gtf", "updated.gtf"), ("detail", "detail.txt"), ("summary", "summary.txt")):
            assert filecmp.cmp(str(tmp_path / (tag + "." + k)), os.path.join(G, "expect_%s.%s" % (name, ext)), shallow=False)


def test_unmapped_record_aborts(oracle, tmp_path):
    # D.2 / Q9: an unmapped record makes update-gtf abort (SIGABRT), bam2gtf skips it
    sam = tmp_path / "u.sam"
    sam.write_text("@SQ\tSN:chr1\tLN:10000000\nu1\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n")
    rc = oracle.run_cli(["update-gtf", str(sam), os.path.join(G, "original.gtf")], stdout_path=str(tmp_path / "o"))
    assert rc == -6
    rc = oracle.run_cli(["bam2gtf", str(sam)], stdout_path=str(tmp_path / "o2"))
    assert rc == 0 and os.path.getsize(tmp_path / "o2") == 0


def test_bam2gtf_golden(oracle, tmp_path):
    rc = oracle.run_cli(["bam2gtf", os.path.join(G, "toy.sam")], stdout_path=str(tmp_path / "b.gtf"))
    assert rc == 0
    assert filecmp.cmp(str(tmp_path / "b.gtf"), os.path.join(G, "expect.bam2gtf.gtf"), shallow=False)


def test_quirk_reads_known_answers(oracle):
    """Q1 / Q4 / Q5 outcomes worked out by hand from SURVEY.md Appendix A (the same reads the GPU suite runs)."""
    from tests import test_gpu_edges as edges, util
    af, reads, pos_of = edges.quirk_case()
    want5 = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=5))
    want2 = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=2))
    edges.check_quirk_outcomes(want5, want2, pos_of)


def test_full_length_levels_known_answers(oracle):
    """check_full / set_full at -l 1..5, outcomes worked out by hand from SURVEY.md Appendix A.4."""
    from tests import test_gpu_edges as edges, util
    af, reads, expect = edges.full_length_case()
    res = {l: util.oracle_run(oracle, af, reads, oracle.default_params(full_level=l)) for l in (1, 2, 3, 4, 5)}
    edges.check_full_length_outcomes(res, expect)
