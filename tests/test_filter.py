"""`lr2rmats filter` (reference src/bam_filter.c): hand-worked known answers for the restatement in oracle/, the host's
SAM -> BAM encoder and BGZF writer against an independent encoder (CPU), and the HIP path through the CLI (GPU)."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from lr2rmats_amd import hostlib, synth
from oracle import filter_oracle as fo

HDR = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:2000000\n@SQ\tSN:chr2\tLN:1500000\n@SQ\tSN:chr3\tLN:900000\n@PG\tID:aligner\tPN:x\n"


def _seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def _line(qname, flag, rname, pos, cigar, seq, nm, extra=(), qual=None, mapq=60):
    aux = ([] if nm is None else ["NM:i:%d" % nm]) + list(extra)
    return "\t".join([qname, str(flag), rname, str(pos), str(mapq), cigar, "*", "0", "0", seq, qual or "*"] + aux) + "\n"


def _qlen(cigar):
    return sum(l for (l, op) in fo.parse_cigar(cigar) if op in (0, 1, 4, 7, 8))


# ---------------------------------------------------------------------------------------------------- hand-worked answers
# (score = aligned query length - NM + deleted bases; every number below is worked out from src/bam_filter.c:61-86,128-154)
HAND = [
    # name   cigar           NM   expectation
    ("a", "10S80M10S", 5),        # 0  qlen 80/100 = .80 ok; score 75 >= .75*80 = 60          -> alone in its group: written
    ("b", "40S60M", 0),           # 1  60/100 = .60 < .67                                       -> dropped
    ("c", "100M", 30),            # 2  score 70 < 75                                            -> dropped
    ("c", "100M", 25),            # 3  score 75 >= 75 (not <)                                   -> kept, alone (record 2 is invisible): written
    ("d", "50M1000N50M", 0),      # 4  score 100, one intron                                    -> written
    ("e", "50M5D50M", 7),         # 5  score 100 - 7 + 5 = 98                                   -> written
    ("f", "100M", 10),            # 6  90 } second 89 >= .98 * 90 = 88.2                        -> the read is not retained
    ("f", "100M", 11),            # 7  89 }
    ("g", "100M", 10),            # 8  90 } second 88 < 88.2                                    -> record 8 written
    ("g", "100M", 12),            # 9  88 }
    ("h", "100M", 10),            # 10 90 } equal scores: the second one only raises s_score to 90 -> not retained
    ("h", "100M", 10),            # 11 90 }
    ("i", "100M", 20),            # 12 80 } the later record is the best; second best 80 < 93.1 -> record 13 written
    ("i", "100M", 5),             # 13 95 }
    ("j", "100M", 3),             # 14 97 } one name, a dropped record of another name between   (15: 0.5 coverage, dropped)
    ("k", "50S50M", 0),           # 15      } the loop never sees record 15: 14 and 16 are ONE group; 97 vs 96 >= 95.06 -> not retained
    ("j", "100M", 4),             # 16 96 }
    ("l", "100M", 2),             # 17 98, last group of the file                               -> written
]
HAND_WRITTEN = [0, 3, 4, 5, 8, 13, 17]


def _hand_sam(path):
    rng = np.random.default_rng(1)
    with open(path, "w") as fh:
        fh.write(HDR)
        for k, (name, cigar, nm) in enumerate(HAND):
            fh.write(_line(name, 0, "chr1", 1000 + 10 * k, cigar, _seq(rng, _qlen(cigar)), nm))


def test_oracle_hand_worked_choices(tmp_path):
    sam = str(tmp_path / "hand.sam")
    _hand_sam(sam)
    _, keep = fo.expected_stream(sam)
    assert keep == HAND_WRITTEN
    # -i 1: only the read with an intron stays; -s 1.0: the 90/89 read is retained as well (89 < 90), equal scores still not
    assert fo.expected_stream(sam, min_intron_n=1)[1] == [4]
    assert fo.expected_stream(sam, sec_rat=1.0)[1] == [0, 3, 4, 5, 6, 8, 13, 14, 17]
    # -v 0.5 lets record 1 (0.60) and record 15 (0.50, not < 0.5) through: 15 now splits the two "j" records into two groups
    assert fo.expected_stream(sam, cov_rate=0.5)[1] == [0, 1, 3, 4, 5, 8, 13, 14, 15, 16, 17]
    # -q 0.70 keeps record 2 (70 >= 70): records 2 and 3 form one group, 70 < .98 * 75 -> record 3 written
    assert fo.expected_stream(sam, map_qual=0.70)[1] == HAND_WRITTEN


def test_oracle_hard_clips_and_remove_overlap():
    # a hard clip is subtracted from l_qseq although SEQ does not hold the clipped bases (:74-76)
    r = fo.Record(_line("x", 0, "chr1", 101, "10H90M", "A" * 90, 0))
    assert fo.score_record(r, 0, fo.COV_RATIO, fo.MAP_QUAL, ()) == (80, 0)
    # remove_overlap compares the 0-based position with 1-based transcript coordinates (:48-59): the read covers 1-based
    # 101..190, pos = 100, pos + rlen - 1 = 189
    assert fo.score_record(r, 0, fo.COV_RATIO, fo.MAP_QUAL, [(0, 190, 300)]) is not None      # 190 > 189: no overlap seen
    assert fo.score_record(r, 0, fo.COV_RATIO, fo.MAP_QUAL, [(0, 189, 300)]) is None
    assert fo.score_record(r, 0, fo.COV_RATIO, fo.MAP_QUAL, [(0, 1, 99)]) is not None          # pos 100 > end 99
    assert fo.score_record(r, 0, fo.COV_RATIO, fo.MAP_QUAL, [(0, 1, 100)]) is None
    # the walk stops at the first transcript of a larger tid: an overlapping transcript behind it is never reached
    assert fo.score_record(r, 0, fo.COV_RATIO, fo.MAP_QUAL, [(1, 1, 10), (0, 1, 1000)]) is not None
    assert fo.score_record(r, 1, fo.COV_RATIO, fo.MAP_QUAL, [(0, 1, 1000), (1, 150, 160), (2, 1, 5)]) is None


# ---------------------------------------------------------------------------------------------------- synthetic input

def make_sam(path, n_reads, seed, with_unmapped=True):
    """Reads with one to four alignments each (consecutive lines of one name), clips, introns, deletions, insertions, NM values
    on both sides of the thresholds, unmapped records, names that come back after another read, aux tags of every type."""
    rng = np.random.default_rng(seed)
    chroms = ["chr1", "chr2", "chr3"]
    lines = []
    for r in range(n_reads):
        name = "read%05d/%d" % (r, int(rng.integers(1, 9)))
        k = int(rng.choice([1, 1, 2, 2, 3, 4]))
        qlen = int(rng.integers(60, 400))
        seq = _seq(rng, qlen)
        qual = "".join(chr(33 + int(q)) for q in rng.integers(0, 42, qlen)) if r % 3 else None
        base_nm = int(rng.integers(0, qlen // 3))
        for a in range(k):
            if with_unmapped and rng.random() < 0.03:
                lines.append("\t".join([name, "4", "*", "0", "0", "*", "*", "0", "0", seq, qual or "*"]) + "\n")
                continue
            left = int(rng.choice([0, 0, 0, 5, 20, qlen // 3])); right = int(rng.choice([0, 0, 3, 15, qlen // 4]))
            clip_l = "SH"[int(rng.integers(0, 2))]; clip_r = "SH"[int(rng.integers(0, 2))]
            body = qlen - left - right
            ops, used = [], 0
            n_seg = int(rng.integers(1, 5))
            for s in range(n_seg):
                seg = body - used if s == n_seg - 1 else max(1, int((body - used) * rng.random() * 0.6))
                if seg <= 0:
                    break
                if rng.random() < 0.3 and seg > 6:
                    i = int(rng.integers(1, 4))
                    ops += [(seg - i - 1, "M"), (i, "I"), (1, "M")]
                else:
                    ops.append((seg, "=" if rng.random() < 0.1 else "M"))
                used += seg
                if s < n_seg - 1 and used < body:
                    ops.append((int(rng.integers(1, 30)), "D") if rng.random() < 0.35 else (int(rng.integers(60, 5000)), "N"))
            if ops and ops[-1][1] in "DN":
                ops.pop()
            cig = ("%d%s" % (left, clip_l) if left else "") + "".join("%d%s" % o for o in ops) + ("%d%s" % (right, clip_r) if right else "")
            # hard clips are not part of SEQ
            s2 = seq[(left if (left and clip_l == "H") else 0): qlen - (right if (right and clip_r == "H") else 0)]
            q2 = None if qual is None else qual[(left if (left and clip_l == "H") else 0): qlen - (right if (right and clip_r == "H") else 0)]
            nm = max(0, base_nm + int(rng.integers(-6, 7)))
            extra = ["AS:i:%d" % int(rng.integers(-40000, 70000)), "XS:A:%s" % "+-"[int(rng.integers(0, 2))],
                     "ms:f:%.3f" % rng.random(), "MD:Z:%dA%d" % (a, r % 50), "ts:A:+"]
            if r % 7 == 0:
                extra.append("ZB:B:S,1,22,333")
            if r % 11 == 0:
                extra.append("ZC:B:f,1.5,-2.25")
            if r % 13 == 0:
                extra.append("ZI:i:-70000")
            flag = (16 if rng.random() < 0.5 else 0) | (256 if a else 0)
            lines.append(_line(name, flag, chroms[int(rng.integers(0, 3))], int(rng.integers(1, 800000)), cig, s2, nm, extra, q2,
                               mapq=int(rng.integers(0, 61))))
        if r % 17 == 0 and r:                                 # an earlier name again, right after a record that may be dropped
            prev = lines[-1].split("\t")
            lines.append(_line("read%05d/%d" % (r - 1, 1), 0, "chr1", 5, "90M10S", _seq(rng, 100), 1))
            lines.append("\t".join([name] + prev[1:]))
    with open(path, "w") as fh:
        fh.write(HDR)
        fh.writelines(lines)
    return len(lines)


def _inflate(path_or_bytes):
    raw = open(path_or_bytes, "rb").read() if isinstance(path_or_bytes, str) else path_or_bytes
    assert raw[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]), "no BGZF end-of-file block"
    return gzip.decompress(raw)


def test_host_sam_to_bam_equals_the_independent_encoder(tmp_path):
    """host/filter.c reader + encoder + BGZF writer (no GPU): every record of a SAM file as BAM == oracle's encoding; the
    BAM it wrote, read again (BGZF reader, records referenced in place), gives the same stream; blocks stay within 64 KiB."""
    sam = str(tmp_path / "in.sam")
    make_sam(sam, 1500, 5)
    header, refs, recs = fo.parse_sam(sam)
    idx = {name: i for i, (name, _) in enumerate(refs)}
    want = fo.header_bytes(header, refs) + b"".join(fo.encode_record(r, idx) for r in recs)
    out1, out2 = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    code = "import sys; sys.path.insert(0, %r); from lr2rmats_amd import hostlib; sys.exit(hostlib.records_to_bam(sys.argv[1], sys.argv[2]))" % \
        os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for src, dst in ((sam, out1), (out1, out2)):
        r = subprocess.run([sys.executable, "-c", code, src, dst], stderr=subprocess.PIPE, env=dict(os.environ, L2R_THREADS="3"))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert _inflate(dst) == want
    raw = open(out1, "rb").read()
    at, n_blocks = 0, 0
    while at < len(raw):
        bsize = (raw[at + 16] | (raw[at + 17] << 8)) + 1
        assert raw[at:at + 4] == b"\x1f\x8b\x08\x04" and bsize <= 65536
        at += bsize; n_blocks += 1
    assert at == len(raw) and n_blocks > 5


def test_host_encoder_long_cigar_goes_into_the_cg_tag(tmp_path):
    """A CIGAR of more than 65535 operations does not fit the BAM record's 16-bit count: it is stored in CG:B,I behind a
    <l_seq>S<ref len>N placeholder (SAMv1 4.2.2; htslib does this when it writes).  The host encoder and the independent one
    agree on the bytes, and the record read back from that BAM scores with its REAL CIGAR (the reader swaps it back in)."""
    rng = np.random.default_rng(3)
    n_pairs = 33_000                                     # 66 000 operations: 1M 1I alternating, one intron in the middle
    cigar = "1M1I" * (n_pairs // 2) + "500N" + "1M1I" * (n_pairs // 2)
    qlen = 2 * n_pairs
    sam = str(tmp_path / "long.sam")
    with open(sam, "w") as fh:
        fh.write(HDR)
        fh.write(_line("long", 0, "chr1", 1000, cigar, _seq(rng, qlen), 7, ["AS:i:5"]))
        fh.write(_line("short", 16, "chr2", 50, "20M", _seq(rng, 20), 0))
    header, refs, recs = fo.parse_sam(sam)
    idx = {name: i for i, (name, _) in enumerate(refs)}
    want = fo.header_bytes(header, refs) + b"".join(fo.encode_record(r, idx) for r in recs)
    assert b"CGBI" in want
    out1, out2 = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    code = "import sys; sys.path.insert(0, %r); from lr2rmats_amd import hostlib; sys.exit(hostlib.records_to_bam(sys.argv[1], sys.argv[2]))" % \
        os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for src, dst in ((sam, out1), (out1, out2)):
        r = subprocess.run([sys.executable, "-c", code, src, dst], stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert _inflate(dst) == want


# ---------------------------------------------------------------------------------------------------- the HIP path (CLI)

def _run_filter(args, stdout_path):
    r = hostlib.run_cli(["filter"] + args, stdout_path=stdout_path)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stderr.decode()


@pytest.mark.gpu
def test_hip_filter_hand_worked_choices(tmp_path):
    sam, out = str(tmp_path / "hand.sam"), str(tmp_path / "out.bam")
    _hand_sam(sam)
    for args, kw in (([], {}), (["-i", "1"], dict(min_intron_n=1)), (["-s", "1.0"], dict(sec_rat=1.0)), (["-v", "0.5"], dict(cov_rate=0.5)),
                     (["-q", "0.70"], dict(map_qual=0.70))):
        err = _run_filter(args + [sam], out)
        want, keep = fo.expected_stream(sam, **kw)
        assert _inflate(out) == want, args
        assert "[bam_filter] Filtered alignments: %d\n" % len(keep) in err
    assert fo.expected_stream(sam)[1] == HAND_WRITTEN


@pytest.mark.gpu
@pytest.mark.parametrize("opts,kw", [([], {}), (["-v", "0.8", "-q", "0.9", "-s", "0.95", "-i", "1"], dict(cov_rate=0.8, map_qual=0.9, sec_rat=0.95, min_intron_n=1)),
                                     (["--coverage", "0.3", "--map-quality", "0.5", "--sec-rat", "1.01"], dict(cov_rate=0.3, map_qual=0.5, sec_rat=1.01))])
def test_hip_filter_equals_the_oracle(tmp_path, opts, kw):
    """SAM text, gzip SAM and BAM input; the decompressed output stream is the oracle's, byte for byte."""
    sam, out = str(tmp_path / "in.sam"), str(tmp_path / "out.bam")
    n = make_sam(sam, 6000, 11)
    want, keep = fo.expected_stream(sam, **kw)
    assert 500 < len(keep) < n
    _run_filter(opts + [sam], out)
    assert _inflate(out) == want
    # the same records as a BAM file (independent encoder + BGZF blocks), and as gzip-compressed SAM
    header, refs, recs = fo.parse_sam(sam)
    idx = {name: i for i, (name, _) in enumerate(refs)}
    bam = str(tmp_path / "in.bam")
    open(bam, "wb").write(fo.bgzf_blocks(fo.header_bytes(header, refs) + b"".join(fo.encode_record(r, idx) for r in recs)))
    _run_filter(opts + [bam], out)
    assert _inflate(out) == want
    samgz = str(tmp_path / "in.sam.gz")
    with open(sam, "rb") as fi, gzip.open(samgz, "wb", compresslevel=1) as fz:
        fz.write(fi.read())
    _run_filter(opts + [samgz], out)
    assert _inflate(out) == want


@pytest.mark.gpu
def test_hip_filter_remove_gtf(tmp_path):
    """-r: records that overlap a transcript of the GTF go (the reference's own comparison, 0-based vs 1-based); the GTF has a
    chromosome the header does not know and its chromosomes are not in header order (the walk stops at a larger tid)."""
    sam, out, gtf = str(tmp_path / "in.sam"), str(tmp_path / "out.bam"), str(tmp_path / "rm.gtf")
    make_sam(sam, 5000, 21)
    rng = np.random.default_rng(4)
    rows, spans = [], []
    for chrom, tid in (("chr2", 1), ("chrUn", -1), ("chr1", 0), ("chr3", 2), ("chr1", 0)):
        for g in range(40):
            a = int(rng.integers(1, 780000)); b = a + int(rng.integers(200, 9000)); c = b + int(rng.integers(100, 3000)); d = c + int(rng.integers(100, 2000))
            t = "%s_t%d_%d" % (chrom, g, len(rows))
            for (s, e) in ((a, b), (c, d)):
                rows.append('%s\tx\texon\t%d\t%d\t.\t+\t.\tgene_id "g%s"; transcript_id "%s";\n' % (chrom, s, e, t, t))
            spans.append((tid, a, d))
    open(gtf, "w").writelines(rows)
    want, keep = fo.expected_stream(sam, spans=spans)
    want0, keep0 = fo.expected_stream(sam)
    assert 100 < len(keep) < len(keep0)
    _run_filter(["-r", gtf, sam], out)
    assert _inflate(out) == want


@pytest.mark.gpu
def test_hip_filter_feeds_update_gtf(oracle, tmp_path):
    """The pipeline's order (Snakefile:90-93): filter, sort, update-gtf.  The filtered BAM (already coordinate sorted
    here: one alignment per read, reads in order) goes straight into update-gtf; files equal the oracle CLI on the SAM."""
    anno = synth.make_annotation(6000, 91, nchr=4)
    reads = synth.make_reads(anno, 5000, 5, 91)
    sam, gtf, nm_sam = str(tmp_path / "r.sam"), str(tmp_path / "a.gtf"), str(tmp_path / "nm.sam")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    rng = np.random.default_rng(2)
    with open(sam) as fi, open(nm_sam, "w") as fo_:              # the synthetic records carry no SEQ / NM: give them both
        for l in fi:
            if l.startswith("@"):
                fo_.write(l); continue
            f = l.rstrip("\n").split("\t")
            f[9] = _seq(rng, _qlen(f[5])); f[10] = "*"
            fo_.write("\t".join(f + ["NM:i:%d" % int(rng.integers(0, 4))]) + "\n")
    fbam = str(tmp_path / "filtered.bam")
    _run_filter([nm_sam], fbam)
    want, keep = fo.expected_stream(nm_sam)
    assert _inflate(fbam) == want and reads.n - 20 < len(keep) <= reads.n
    kept_sam = str(tmp_path / "kept.sam")                       # the oracle CLI gets the same records as SAM text
    body = [l for l in open(sam) if not l.startswith("@")]
    with open(kept_sam, "w") as fh:
        fh.writelines([l for l in open(sam) if l.startswith("@")] + [body[i] for i in keep])
    a, b = str(tmp_path / "o.gtf"), str(tmp_path / "h.gtf")
    assert oracle.run_cli(["update-gtf", "-l", "3", kept_sam, gtf], stdout_path=a) == 0
    r = hostlib.run_cli(["update-gtf", "-l", "3", fbam, gtf], stdout_path=b)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.gpu
def test_c_abi_filter_score_and_select_at_size(tmp_path):
    """l2r_filter_score / l2r_filter_select through the C-ABI on 300 k records (numpy restatement of the two loops as the
    checker: the same float / double arithmetic, vectorised), with a -r span table of 20 k transcripts."""
    from lr2rmats_amd import capi
    rng = np.random.default_rng(8)
    n = 300_000
    n_ops = rng.integers(1, 12, n)
    off = np.zeros(n + 1, np.int64); np.cumsum(n_ops, out=off[1:])
    ops = rng.choice(np.array([0, 0, 0, 1, 2, 3, 7, 8], np.uint32), size=int(off[-1]))
    lens = rng.integers(1, 400, size=int(off[-1])).astype(np.uint32)
    first, last = off[:-1], off[1:] - 1
    clip = rng.random(n)
    ops[first[clip < 0.3]] = 4; ops[first[(clip >= 0.3) & (clip < 0.4)]] = 5
    ops[last[clip > 0.7]] = 4
    cig = (lens << 4) | ops
    flag = np.where(rng.random(n) < 0.02, 4, 0).astype(np.uint16)
    tid = rng.integers(0, 5, n).astype(np.int32); pos = rng.integers(0, 3_000_000, n).astype(np.int32)
    opl = lens.astype(np.int64)
    rid = np.repeat(np.arange(n), n_ops)
    qsum = np.bincount(rid, weights=np.where(np.isin(ops, [0, 1, 4, 7, 8]), opl, 0), minlength=n).astype(np.int64)
    l_qseq = np.maximum(qsum + rng.integers(-3, 4, n), 0).astype(np.int32)          # (not always consistent with the CIGAR: any input is defined)
    nm = rng.integers(0, 200, n).astype(np.int32)
    T = 20_000
    sp_tid = np.sort(rng.integers(-1, 6, T)).astype(np.int32); rng.shuffle(sp_tid[: T // 10])
    sp_start = rng.integers(1, 3_000_000, T).astype(np.int32); sp_end = (sp_start + rng.integers(10, 3000, T)).astype(np.int32)
    prm = capi.CFilterParams(0.67, 0.75, 0.98, 1)
    eng = capi.Engine(0)
    try:
        drop, score, intron = eng.filter_score(flag, tid, pos, l_qseq, nm, off, cig, prm, (sp_tid, sp_start, sp_end))
        # ---- checker
        w_in = np.bincount(rid, weights=(ops == 3), minlength=n).astype(np.int32)
        w_del = np.bincount(rid, weights=np.where(ops == 2, opl, 0), minlength=n).astype(np.int64)
        w_rlen = np.bincount(rid, weights=np.where(np.isin(ops, [0, 2, 3, 7, 8]), opl, 0), minlength=n).astype(np.int64)
        qlen = l_qseq.astype(np.int64).copy()
        f_op, l_op = ops[first], ops[last]
        qlen -= np.where((f_op == 4) | (f_op == 5), opl[first], 0)
        qlen -= np.where((n_ops > 1) & ((l_op == 4) | (l_op == 5)), opl[last], 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            d = (flag & 4) != 0
            d |= (qlen.astype(np.float64) / l_qseq.astype(np.float64)) < np.float64(np.float32(0.67))
        s = qlen - nm + w_del
        d |= s.astype(np.float32) < np.float32(0.75) * qlen.astype(np.float32)
        hit = np.zeros(n, bool)
        for v in range(5):                                   # remove_overlap: transcripts of tid v in front of the first larger tid
            stop = np.nonzero(sp_tid > v)[0]
            stop = int(stop[0]) if stop.size else T
            sel = np.nonzero(sp_tid[:stop] == v)[0]
            if not sel.size:
                continue
            o = np.argsort(sp_start[sel], kind="stable")
            st, pe = sp_start[sel][o], np.maximum.accumulate(sp_end[sel][o])
            m = np.nonzero(tid == v)[0]
            k = np.searchsorted(st, (pos[m].astype(np.int64) + w_rlen[m] - 1), side="right")
            hit[m] = (k > 0) & (pe[np.maximum(k - 1, 0)] >= pos[m])
        d |= hit & ~d
        np.testing.assert_array_equal(drop != 0, d)
        np.testing.assert_array_equal(score, np.where(d, 0, s).astype(np.int32))
        np.testing.assert_array_equal(intron, np.where((flag & 4) != 0, 0, w_in))
        assert 0.2 < d.mean() < 0.95 and hit.sum() > 1000
        # ---- select: groups of 1..6 kept rows
        kept = np.nonzero(~d)[0]
        gl = rng.integers(1, 7, kept.size); goff = np.concatenate([[0], np.cumsum(gl)]); goff = goff[goff < kept.size]
        goff = np.concatenate([goff, [kept.size]]).astype(np.int64)
        ks, ki = score[kept], intron[kept]
        win = eng.filter_select(goff, ks, ki, prm)
        want = np.full(goff.size - 1, -1, np.int64)
        for g in range(goff.size - 1):
            a, b = int(goff[g]), int(goff[g + 1])
            best, bs, ss = a, int(ks[a]), 0
            for k in range(a + 1, b):
                if ks[k] > bs:
                    ss, bs, best = bs, int(ks[k]), k
                elif ks[k] > ss:
                    ss = int(ks[k])
            if np.float32(ss) < np.float32(0.98) * np.float32(bs) and ki[best] >= 1:
                want[g] = best
        np.testing.assert_array_equal(win, want)
        assert (want >= 0).sum() > 1000
    finally:
        eng.close()


@pytest.mark.gpu
def test_hip_filter_empty_and_header_only_inputs(tmp_path):
    """No records at all, and records that all fail: the output is the header and the end-of-file block."""
    out = str(tmp_path / "out.bam")
    empty = str(tmp_path / "empty.sam")
    open(empty, "w").write(HDR)
    err = _run_filter([empty], out)
    assert _inflate(out) == fo.expected_stream(empty)[0] and "Filtered alignments: 0" in err
    rng = np.random.default_rng(6)
    bad = str(tmp_path / "bad.sam")
    with open(bad, "w") as fh:
        fh.write(HDR)
        fh.write(_line("a", 0, "chr1", 10, "60S40M", _seq(rng, 100), 0))          # coverage 0.4
        fh.write(_line("b", 4, "*", 0, "*", _seq(rng, 30), None))                   # unmapped
        fh.write(_line("c", 0, "chr1", 10, "100M", _seq(rng, 100), 60))             # identity 0.4
    err = _run_filter([bad], out)
    want, keep = fo.expected_stream(bad)
    assert keep == [] and _inflate(out) == want and "Filtered alignments: 0" in err


@pytest.mark.gpu
def test_hip_filter_long_cigar_record(tmp_path):
    """The 66 000-operation record through the whole sub-command, as SAM and as the BAM made from it (CG tag)."""
    rng = np.random.default_rng(3)
    n_pairs = 33_000
    cigar = "1M1I" * (n_pairs // 2) + "500N" + "1M1I" * (n_pairs // 2)
    sam, out = str(tmp_path / "long.sam"), str(tmp_path / "out.bam")
    with open(sam, "w") as fh:
        fh.write(HDR)
        fh.write(_line("long", 0, "chr1", 1000, cigar, _seq(rng, 2 * n_pairs), 7))
        fh.write(_line("long", 256, "chr2", 1000, "%dM" % (2 * n_pairs), _seq(rng, 2 * n_pairs), 40000))
    want, keep = fo.expected_stream(sam, min_intron_n=1)
    assert keep == [0]
    _run_filter(["-i", "1", sam], out)
    assert _inflate(out) == want
    bam = str(tmp_path / "in.bam")
    header, refs, recs = fo.parse_sam(sam)
    idx = {name: i for i, (name, _) in enumerate(refs)}
    open(bam, "wb").write(fo.bgzf_blocks(fo.header_bytes(header, refs) + b"".join(fo.encode_record(r, idx) for r in recs)))
    _run_filter(["-i", "1", bam], out)
    assert _inflate(out) == want


@pytest.mark.gpu
def test_hip_filter_reads_nm_only_behind_the_coverage_test(tmp_path):
    """gtf_filter() looks for the NM tag after the coverage test (src/bam_filter.c:77-80): a record that fails coverage is dropped
    without one; a record that passes coverage without one ends the reference's run (a null pointer there, a message here)."""
    rng = np.random.default_rng(7)
    ok, out = str(tmp_path / "ok.sam"), str(tmp_path / "out.bam")
    with open(ok, "w") as fh:
        fh.write(HDR)
        fh.write(_line("a", 0, "chr1", 10, "60S40M", _seq(rng, 100), None))       # coverage 0.4, no NM: dropped, no error
        fh.write(_line("b", 0, "chr1", 10, "100M", _seq(rng, 100), 1))
    err = _run_filter([ok], out)
    assert "Filtered alignments: 1" in err
    bad = str(tmp_path / "bad.sam")
    with open(bad, "w") as fh:
        fh.write(HDR)
        fh.write(_line("a", 0, "chr1", 10, "100M", _seq(rng, 100), None))         # passes coverage, no NM
    r = hostlib.run_cli(["filter", bad], stdout_path=out)
    assert r.returncode == 1 and b"no NM tag" in r.stderr
