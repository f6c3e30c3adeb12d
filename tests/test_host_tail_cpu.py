"""CPU: the C host side (readers + sequential tail + writers) through its staged API.

The per-read results come from the oracle here (no GPU in this container); what is under test is
that the host parsers produce the arrays the oracle's own parsers produce, and that routing / split /
merge / writers turn identical per-read results into byte-identical files.
"""
import filecmp
import os
import subprocess
import sys

import numpy as np
import pytest

from lr2rmats_amd import hostlib, synth
from tests import util

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "toy")
OUTS = ("gtf", "detail", "summary", "bed", "known", "novel", "unrec", "all")


def _paths(tmp, tag):
    return {k: str(tmp / ("%s.%s" % (tag, k))) for k in OUTS}


def _args(extra, o, sam, gtf):
    return ["update-gtf"] + extra + ["-A", o["detail"], "-y", o["summary"], "-E", o["bed"], "-k", o["known"], "-v", o["novel"],
                                     "-u", o["unrec"], "-a", o["all"], "-o", o["gtf"], sam, gtf]


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from lr2rmats_amd import hostlib
from oracle import pyoracle as po
job = hostlib.Job(sys.argv[1:])
a, r, sj, p = job.annotation_arrays(), job.read_arrays(), job.junction_arrays(), job.prm
op = po.default_params(min_exon=p.min_exon, min_intron=p.min_intron, max_delet=p.max_delet, ss_dis=p.ss_dis, end_dis=p.end_dis,
                       full_level=p.full_level, split_trans=p.split_trans, use_multi=p.use_multi, min_sj_cnt=p.min_sj_cnt,
                       force_strand=p.force_strand, single_exon_ovlp_frac=p.single_exon_ovlp_frac)
res = po.classify_soa(r["tid"], r["pos"], r["rev"], r["cig_off"], r["cig"], a["tx_tid"], a["tx_start"], a["tx_end"], a["tx_rev"],
                      a["tx_ex_off"], a["ex_start"], a["ex_end"], sj=sj, params=op)
info = (res.info & 0x7f) | (np.diff(res.ex_off).astype(np.uint32) << 8)
sys.exit(job.finish(res.ex_off, res.ex_start, res.ex_end, res.ex_flag, info, res.ref_tx))
"""


def _host_with_oracle_results(argv, env=None):
    """Staged host run in a child process (the C code exits the process on fatal errors)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", _CHILD % root] + argv, stderr=subprocess.PIPE, env=env).returncode


def _compare(oracle, tmp_path, extra, sam, gtf, tag):
    oo, ho = _paths(tmp_path, tag + ".o"), _paths(tmp_path, tag + ".h")
    assert oracle.run_cli(_args(extra, oo, sam, gtf)) == 0
    assert _host_with_oracle_results(_args(extra, ho, sam, gtf)) == 0
    for k in OUTS:
        assert filecmp.cmp(oo[k], ho[k], shallow=False), (tag, k)
    return oo


@pytest.mark.parametrize("extra,tag", [(["-l", "3"], "l3"), (["-l", "5"], "l5"),
                                       (["-s", "-l", "3", "-J", "1", "-j", os.path.join(G, "sj_far.tab")], "far"),
                                       (["-s", "-l", "3", "-J", "1", "-j", os.path.join(G, "sj_other.tab")], "oth"),
                                       (["-l", "3", "-j", os.path.join(G, "sj_support.tab"), "-S", "mysrc"], "sup")])
def test_toy_files_identical(oracle, tmp_path, extra, tag):
    _compare(oracle, tmp_path, extra, os.path.join(G, "toy.sam"), os.path.join(G, "original.gtf"), tag)


@pytest.fixture(scope="module")
def synth_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("syn")
    anno = synth.make_annotation(6000, 31, nchr=6, shuffle_within_gene=True, long_tx_per_chrom=1)
    reads = synth.make_reads(anno, 6000, 5, 31, xs_conflict_frac=0.03)
    sam, gtf = str(d / "r.sam"), str(d / "a.gtf")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    return d, anno, reads, sam, gtf


@pytest.mark.parametrize("extra,tag", [(["-l", "3"], "a"), (["-l", "5", "-d", "3", "-c"], "b"), (["-l", "1", "-e", "8", "-i", "150"], "c"),
                                       (["-l", "4", "-f", "0.5", "-D", "20"], "d")])
def test_synthetic_files_identical(oracle, tmp_path, synth_files, extra, tag):
    d, anno, reads, sam, gtf = synth_files
    oo = _compare(oracle, tmp_path, extra, sam, gtf, tag)
    assert os.path.getsize(oo["gtf"]) > 1000 and os.path.getsize(oo["detail"]) > 100000


@pytest.mark.parametrize("extra,tag", [(["-l", "3", "-J", "2"], "j1"), (["-s", "-l", "3", "-J", "3", "-M", "x"], "j2"), (["-s", "-l", "5", "-d", "2"], "j3")])
def test_synthetic_with_junction_table(oracle, tmp_path, synth_files, extra, tag):
    d, anno, reads, sam, gtf = synth_files
    af = anno.in_file_order()
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=5))
    j, _ = util.junction_table(af, reads, base, 31, cover=0.6)
    tab = str(d / "sj.tab")
    # shuffle the rows and add a chromosome that is not in the header: the reader sorts and interns it
    order = np.random.default_rng(1).permutation(len(j.don))
    j.chrom = [j.chrom[i] for i in order] + ["chrUn_x"]
    for f in ("tid", "don", "acc", "strand", "uniq", "multi"):
        setattr(j, f, np.concatenate([getattr(j, f)[order], [7]]))
    j.write(tab)
    oo = _compare(oracle, tmp_path, extra + ["-j", tab], sam, gtf, tag)
    if "-s" in extra:
        assert ".split." in open(oo["gtf"]).read()


def test_gtf_reader_quirks(oracle, tmp_path, synth_files):
    """Q10-Q12: long lines split at 1023 bytes, blank / short lines re-using stale fields, tag substring lookup,
    id<->name fallbacks, chromosomes that are not in the header."""
    d, anno, reads, sam, gtf = synth_files
    lines = open(gtf).read().split("\n")
    out = []
    for i, l in enumerate(lines):
        out.append(l)
        if i == 40:
            out.append("")                                                   # blank line: stale type/attrs
        if i == 60 and "\texon\t" in l:
            out.append(l + " note \"" + "x" * 1500 + "\";")                 # > 1023 bytes
        if i == 80 and "\texon\t" in l:
            f = l.split("\t"); f[8] = 'ref_gene_id "zz"; ' + f[8]; out.append("\t".join(f))
        if i == 100 and "\texon\t" in l:
            f = l.split("\t"); f[8] = 'transcript_id "only_id_%d"; gene_name "only_name";' % i; out.append("\t".join(f))
        if i == 120 and "\texon\t" in l:
            f = l.split("\t"); f[0] = "chrNotInHeader"; f[8] = f[8].replace("SYNT", "ZZZT"); out.append("\t".join(f))
        if i == 140:
            out.append("chr1 synth exon 5 9")                                # short line, space separated
    g2 = str(tmp_path / "quirk.gtf")
    open(g2, "w").write("\n".join(out))
    _compare(oracle, tmp_path, ["-l", "3"], sam, g2, "q")


def test_unmapped_record_aborts_like_reference(tmp_path):
    sam = tmp_path / "u.sam"
    sam.write_text("@SQ\tSN:chr1\tLN:10000000\nu1\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n")
    rc = _host_with_oracle_results(["update-gtf", str(sam), os.path.join(G, "original.gtf")])
    assert rc == -6                                                           # SIGABRT, as the reference (Q9)


def test_usage_errors_return_1(tmp_path):
    assert _host_with_oracle_results(["update-gtf", "only_one_positional"]) == 1
    assert _host_with_oracle_results(["update-gtf", "-m", "x", "a", "b"]) == 1


def test_bam_reader_equals_sam_reader(tmp_path, synth_files):
    """The BGZF/BAM route of the alignment reader yields the arrays of the SAM text route."""
    d, anno, reads, sam, gtf = synth_files
    bam = str(tmp_path / "r.bam")
    synth.write_bam(reads, bam)
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from lr2rmats_amd import hostlib
a = hostlib.Job(["update-gtf", sys.argv[1], sys.argv[3]]).read_arrays()
b = hostlib.Job(["update-gtf", sys.argv[2], sys.argv[3]]).read_arrays()
assert a["tid"].size > 1000
for k in a:
    assert np.array_equal(a[k], b[k]), k
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code, sam, bam, gtf], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    # and the arrays are the generator's
    j = hostlib.Job(["update-gtf", bam, gtf])
    ra = j.read_arrays()
    np.testing.assert_array_equal(ra["tid"], reads.tid)
    np.testing.assert_array_equal(ra["pos"], reads.pos)
    np.testing.assert_array_equal(ra["rev"], reads.rev)
    np.testing.assert_array_equal(ra["cig_off"], reads.cig_off)
    np.testing.assert_array_equal(ra["cig"], reads.cig)
    # ... and the reader's per-record CIGAR summaries (l2r_reads::cig_summary) are what the numpy form of the rule gives
    np.testing.assert_array_equal(ra["cig_summary"], synth.cigar_summary(reads.cig_off, reads.cig))
    j.close()


@pytest.mark.parametrize("threads,part", [("3", ""), ("8", ""), ("5", "40"), ("2", "7")])
def test_threaded_tail_is_byte_identical(oracle, tmp_path, threads, part):
    """The tail on several host threads (parts cut at coverage gaps -- chromosome boundaries and, inside a chromosome, reads that
    start behind the end of every read in front of them -- into memory streams, written out in order) against the oracle's single
    sequential pass.  `part`: reads per part (L2R_TAIL_PART_READS): a few reads per part put a cut into nearly every gap between
    two loci, and none between two reads that overlap (merge_trans across a would-be cut)."""
    anno = synth.make_annotation(6000, 61, nchr=5, shuffle_within_gene=True)
    reads = synth.make_reads(anno, 6000, 5, 61, xs_conflict_frac=0.02)
    sam, gtf = str(tmp_path / "r.sam"), str(tmp_path / "a.gtf")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    oo, ho = _paths(tmp_path, "t.o"), _paths(tmp_path, "t.h")
    assert oracle.run_cli(_args(["-l", "3"], oo, sam, gtf)) == 0
    env = dict(os.environ, L2R_THREADS=threads)
    if part:
        env["L2R_TAIL_PART_READS"] = part
    assert _host_with_oracle_results(_args(["-l", "3"], ho, sam, gtf), env=env) == 0
    for k in OUTS:
        assert filecmp.cmp(oo[k], ho[k], shallow=False), (threads, part, k)
    assert os.path.getsize(oo["detail"]) > 100000


def test_threaded_tail_counts_a_gene_on_two_chromosomes_like_the_sequential_run(oracle, tmp_path):
    """merge_gene (src/update_gtf.c:181-189) compares the gene_id with the list's last entry before the tid break: a gene
    id that continues on the next chromosome is counted once.  Every gene of this annotation carries the same id, so the
    sequential run counts 1 updated / 1 known gene; parts that start with empty lists would count one per part."""
    anno = synth.make_annotation(4000, 62, nchr=4)
    reads = synth.make_reads(anno, 24000, 5, 62)
    sam, gtf, gtf1 = str(tmp_path / "r.sam"), str(tmp_path / "a.gtf"), str(tmp_path / "one_gene.gtf")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    import re
    with open(gtf) as fi, open(gtf1, "w") as fo:
        for l in fi:
            fo.write(re.sub(r'gene_id "[^"]*"', 'gene_id "GX"', l))
    oo, ho = _paths(tmp_path, "g.o"), _paths(tmp_path, "g.h")
    assert oracle.run_cli(_args(["-l", "3"], oo, sam, gtf1)) == 0
    for env in (dict(os.environ, L2R_THREADS="4"), dict(os.environ, L2R_THREADS="4", L2R_TAIL_PART_READS="25")):
        assert _host_with_oracle_results(_args(["-l", "3"], ho, sam, gtf1), env=env) == 0
        for k in OUTS:
            assert filecmp.cmp(oo[k], ho[k], shallow=False), k
    summ = dict(l.rstrip("\n").split("\t") for l in open(oo["summary"]) if "\t" in l)
    assert summ["Updated_Genes"] == "1" and summ["Genes_of_Known_Transcripts_from_BAM"] == "1", summ


def test_bgzf_blocks_inflated_on_several_threads(tmp_path):
    """A BAM of a few hundred BGZF blocks read with 1 and with 5 inflate threads, and window by window, gives the generator's arrays; a
    gzip-compressed SAM (one gzip member, not BGZF) still goes through the sequential route."""
    import gzip
    anno = synth.make_annotation(8000, 71, nchr=3)
    reads = synth.make_reads(anno, 60000, 6, 71, ont=True)
    bam, gtf, samgz = str(tmp_path / "big.bam"), str(tmp_path / "a.gtf"), str(tmp_path / "r.sam.gz")
    synth.write_bam(reads, bam)
    anno.write_gtf(gtf)
    assert os.path.getsize(bam) > 8 * 20000
    sam = str(tmp_path / "r.sam")
    reads.write_sam(sam)
    with open(sam, "rb") as fi, gzip.open(samgz, "wb", compresslevel=1) as fo:
        fo.write(fi.read())
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from lr2rmats_amd import hostlib
j = hostlib.Job(["update-gtf", sys.argv[1], sys.argv[2]])
a = j.read_arrays()
np.savez(sys.argv[3], **a)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (w1 / w3: the BAM in windows of one block and of about three: the records and the header cross window borders, every batch
    #  is appended to the arrays of the batches before it -- aln_reader.c h_aln_stream)
    for tag, fn, thr, win in (("t1", bam, "1", ""), ("t5", bam, "5", ""), ("gz", samgz, "5", ""), ("w1", bam, "3", "65600"), ("w3", bam, "2", "200000")):
        out = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ, L2R_THREADS=thr)
        if win:
            env["L2R_READ_WINDOW"] = win
        r = subprocess.run([sys.executable, "-c", code, fn, gtf, out], stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        z = np.load(out)
        np.testing.assert_array_equal(z["tid"], reads.tid)
        np.testing.assert_array_equal(z["pos"], reads.pos)
        np.testing.assert_array_equal(z["rev"], reads.rev)
        np.testing.assert_array_equal(z["cig_off"], reads.cig_off)
        np.testing.assert_array_equal(z["cig"], reads.cig)
    # a block whose stored CRC32 does not match its inflated bytes is refused (htslib refuses it too)
    raw = bytearray(open(bam, "rb").read())
    bsize = (raw[16] | (raw[17] << 8)) + 1                  # first block; its CRC32 sits 8 bytes before the block's end
    raw[bsize - 8] ^= 0x5a
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(raw))
    r = subprocess.run([sys.executable, "-c", code, bad, gtf, str(tmp_path / "bad.npz")], stderr=subprocess.PIPE, env=dict(os.environ, L2R_THREADS="3"))
    assert r.returncode != 0 and b"CRC32 mismatch" in r.stderr


def test_parsed_gtf_cache(tmp_path, synth_files):
    """L2R_ANNO_CACHE: the second job over the same GTF + header reads the parsed arrays back (same arrays, same names);
    a changed GTF (other size / mtime), another header or a damaged cache file is parsed anew."""
    d, anno, reads, sam, gtf = synth_files[:5]
    cache = tmp_path / "cache"
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from lr2rmats_amd import hostlib
j = hostlib.Job(["update-gtf", "-o", sys.argv[4], sys.argv[1], sys.argv[2]], open_outputs=False)
a = j.annotation_arrays()
np.savez(sys.argv[3], **a)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def arrays(tag, gtf_file, env):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, sam, gtf_file, out, str(tmp_path / "o.gtf")], stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        return dict(np.load(out))

    plain = arrays("plain", gtf, dict(os.environ))
    env = dict(os.environ, L2R_ANNO_CACHE=str(cache))
    first = arrays("first", gtf, env)
    files = sorted(os.listdir(cache))
    assert len(files) == 1 and files[0].startswith("l2r_gtf_") and files[0].endswith(".parsed")
    stamp = os.path.getmtime(cache / files[0])
    second = arrays("second", gtf, env)
    assert os.path.getmtime(cache / files[0]) == stamp and sorted(os.listdir(cache)) == files       # read, not rewritten
    for k in plain:
        np.testing.assert_array_equal(plain[k], first[k])
        np.testing.assert_array_equal(plain[k], second[k])
    # the names come back too: the updated GTF written from cached arrays is the uncached one
    outs = {}
    for tag, e in (("n", dict(os.environ)), ("c", env)):
        o = str(tmp_path / (tag + ".det"))
        assert _host_with_oracle_results(["update-gtf", "-l", "3", "-a", o, "-o", str(tmp_path / (tag + ".gtf")), sam, gtf], env=e) == 0
        outs[tag] = (o, str(tmp_path / (tag + ".gtf")))
    assert filecmp.cmp(outs["n"][0], outs["c"][0], shallow=False) and filecmp.cmp(outs["n"][1], outs["c"][1], shallow=False)
    # a GTF with one transcript less under the same name: other size -> other key
    lines = open(gtf).read().splitlines(keepends=True)
    gtf2 = str(tmp_path / "cut.gtf")
    open(gtf2, "w").write("".join(lines[: len(lines) * 2 // 3]))
    cut_plain, cut_cached = arrays("cp", gtf2, dict(os.environ)), arrays("cc", gtf2, env)
    assert cut_plain["tx_tid"].shape[0] < plain["tx_tid"].shape[0]
    for k in cut_plain:
        np.testing.assert_array_equal(cut_plain[k], cut_cached[k])
    assert len(os.listdir(cache)) == 2
    # damage: a truncated file and one with a flipped offset are both ignored (and replaced)
    path = cache / files[0]
    raw = bytearray(open(path, "rb").read())
    open(path, "wb").write(bytes(raw[: len(raw) // 2]))
    again = arrays("again", gtf, env)
    raw[64 + 8 * 3 + 100] ^= 0xff
    open(path, "wb").write(bytes(raw))
    again2 = arrays("again2", gtf, env)
    for k in plain:
        np.testing.assert_array_equal(plain[k], again[k])
        np.testing.assert_array_equal(plain[k], again2[k])


def test_gtf_input_mode(oracle, tmp_path):
    """`update-gtf -m g -b hdr.sam reads.gtf anno.gtf`: read-like transcripts from a GTF (here the reads' own bam2gtf
    output, trans_name != trans_id after editing) take the alignment records' place; files equal the oracle's."""
    anno = synth.make_annotation(6000, 81, nchr=4, shuffle_within_gene=True)
    reads = synth.make_reads(anno, 4000, 5, 81, xs_conflict_frac=0.02)
    sam, gtf, rgtf = str(tmp_path / "r.sam"), str(tmp_path / "a.gtf"), str(tmp_path / "reads.gtf")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    assert oracle.run_cli(["bam2gtf", sam], stdout_path=rgtf) == 0
    # give every transcript a transcript_name that differs from its transcript_id
    lines = []
    for l in open(rgtf):
        f = l.rstrip("\n").split("\t")
        if len(f) > 8 and 'transcript_id "' in f[8]:
            tid = f[8].split('transcript_id "')[1].split('"')[0]
            if "transcript_name" not in f[8]:
                f[8] = f[8].rstrip() + ' transcript_name "N_%s";' % tid
        lines.append("\t".join(f))
    open(rgtf, "w").write("\n".join(lines) + "\n")
    oo, ho = _paths(tmp_path, "g.o"), _paths(tmp_path, "g.h")
    extra = ["-m", "g", "-b", sam, "-l", "3"]
    assert oracle.run_cli(_args(extra, oo, rgtf, gtf)) == 0
    assert _host_with_oracle_results(_args(extra, ho, rgtf, gtf)) == 0
    for k in OUTS:
        assert filecmp.cmp(oo[k], ho[k], shallow=False), k
    assert os.path.getsize(oo["detail"]) > 50000 and "N_" in open(oo["detail"]).read(4000)


def test_cigar_summaries_saturate_and_match_the_numpy_form():
    """l2r_reads::cig_summary as host/aln_reader.c makes it (h_cigar_summaries) against synth.cigar_summary (numpy) on hand-made CIGARs at the
    fields' limits: no N at all, N / D / stretches of 65535 and more (16-bit fields saturate), an empty CIGAR, clips and insertions
    (they advance nothing), a stretch made of several ops, and against literal words for three of them."""
    M, I, D, N, S, H, EQ, X = 0, 1, 2, 3, 4, 5, 7, 8

    def cig(ops):
        return [(l << 4) | o for l, o in ops]
    reads = [
        cig([(100, M)]),                                                        # no N: shortest N / stretch = 65535
        cig([(50, M), (70000, N), (10, EQ), (5, X), (3, D), (65535, N), (20, M)]),      # N beyond and at 65535; stretch 10 + 5 + 3 = 18
        cig([(5, S), (30, M), (4, I), (70000, D), (200, N), (8, M), (120, N), (9, M), (2, H)]),     # D beyond 65535; first stretch not counted: 8
        [],                                                                     # empty
        cig([(1, M), (1, N), (70000, M), (1, N), (1, M)]),                      # stretch beyond 65535
    ]
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    flat = np.array([w for r in reads for w in r], np.uint32)
    a = hostlib.cigar_summaries(off, flat)
    b = synth.cigar_summary(off, flat)
    np.testing.assert_array_equal(a, b)
    assert list(a[0]) == [100, 0 | (65535 << 16), 0 | (65535 << 16)]
    assert list(a[1]) == [50 + 70000 + 10 + 5 + 3 + 65535 + 20, 2 | (65535 << 16), 3 | (18 << 16)]
    assert list(a[2]) == [30 + 70000 + 200 + 8 + 120 + 9, 2 | (120 << 16), 65535 | (8 << 16)]
    assert list(a[3]) == [0, 0 | (65535 << 16), 0 | (65535 << 16)]
    assert list(a[4]) == [70004, 2 | (1 << 16), 0 | (65535 << 16)]
    # and on random ONT-like records
    anno = synth.make_annotation(3000, 5)
    r = synth.make_reads(anno, 2000, 6, 5, ont=True, micro_exons=2)
    np.testing.assert_array_equal(hostlib.cigar_summaries(r.cig_off, r.cig), synth.cigar_summary(r.cig_off, r.cig))
