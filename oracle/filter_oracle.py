"""TEST INFRASTRUCTURE ONLY -- CPU restatement of `lr2rmats filter` (reference src/bam_filter.c).

Only tests/ may import this (see oracle/oracle.h for the rule).  Plain Python over SAM text: small inputs.

    parse_sam()            header lines + records (the eleven mandatory fields and the aux fields, as text)
    score_record()         gtf_filter()      src/bam_filter.c:61-86   (+ remove_overlap() :48-59)
    select()               the loop of bam_filter()   :128-154
    encode_record() ...    SAM line -> BAM record bytes, SAMv1 sections 1.4 / 4.2 (what htslib's sam_parse1 + bam_write1
                           put on disk); written from the specification, independently of host/filter.c
    expected_stream()      the uncompressed BAM stream `filter` has to write: header + the chosen records

parity: the reference itself cannot be built here (htslib is an empty submodule); this restatement is
pinned by the hand-worked cases of tests/test_filter.py (scores and choices derived on paper from the reference's
source) -- "parity unpinned by reference runs", like the rest of oracle/.
"""
from __future__ import annotations

import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

COV_RATIO, MAP_QUAL, SEC_RATIO, MIN_INTRON_NUM = 0.67, 0.75, 0.98, 0      # src/bam_filter.c:10-12, src/gtf.h:123
OPS = "MIDNSHP=XB"
REF_CONSUMING = {0, 2, 3, 7, 8}                                               # bam_cigar2rlen: M D N = X


class Record:
    __slots__ = ("qname", "flag", "rname", "pos", "mapq", "cigar", "rnext", "pnext", "tlen", "seq", "qual", "aux", "line")

    def __init__(self, line: str):
        f = line.rstrip("\r\n").split("\t")
        assert len(f) >= 11, line
        self.line = line
        self.qname, self.flag, self.rname, self.pos, self.mapq = f[0], int(f[1], 0), f[2], int(f[3]), int(f[4])
        self.cigar = [] if f[5] == "*" else parse_cigar(f[5])
        if f[5] == "*":
            self.flag |= 4                                                    # sam_parse1: "treated as unmapped"
        self.rnext, self.pnext, self.tlen, self.seq, self.qual = f[6], int(f[7]), int(f[8]), f[9], f[10]
        self.aux = [a for a in f[11:] if a]


def parse_cigar(s: str) -> List[Tuple[int, int]]:
    out, n = [], 0
    for ch in s:
        if ch.isdigit():
            n = n * 10 + ord(ch) - 48
        else:
            out.append((n, OPS.index(ch)))
            n = 0
    return out


def parse_sam(path: str):
    header, recs = [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith("@"):
                header.append(line if line.endswith("\n") else line + "\n")
            elif line.strip():
                recs.append(Record(line))
    refs = []
    for h in header:
        if h.startswith("@SQ"):
            d = dict(x.split(":", 1) for x in h.rstrip("\n").split("\t")[1:] if ":" in x)
            refs.append((d["SN"], int(d.get("LN", 0))))
    return header, refs, recs


def aux_nm(rec: Record) -> Optional[int]:
    """bam_aux2i(bam_aux_get(b, "NM")): the value for an integer tag, 0 for another type, None when absent (the reference
    dereferences NULL then, src/bam_filter.c:78-80)."""
    for a in rec.aux:
        if a.startswith("NM:"):
            return int(a[5:]) if a[3] == "i" else 0
    return None


def score_record(rec: Record, tid: int, cov_rate, map_qual, spans) -> Optional[Tuple[int, int]]:
    """gtf_filter() src/bam_filter.c:61-86: None = filtered out, else (score, intron_n)."""
    if rec.flag & 4:                                                           # :63
        return None
    c = rec.cigar
    intron_n = sum(1 for (l, op) in c if op == 3)                              # :68-71
    del_len = sum(l for (l, op) in c if op == 2)
    l_qseq = 0 if rec.seq == "*" else len(rec.seq)
    cigar_qlen = l_qseq                                                        # :73-76
    if c and c[0][1] in (4, 5):
        cigar_qlen -= c[0][0]
    if len(c) > 1 and c[-1][1] in (4, 5):
        cigar_qlen -= c[-1][0]
    with np.errstate(divide="ignore", invalid="ignore"):
        if np.float64(cigar_qlen + 0.0) / np.float64(l_qseq) < np.float64(np.float32(cov_rate)):      # :77 (double arithmetic, float option)
            return None
    ed = aux_nm(rec)                                                           # :78-80
    assert ed is not None, "no NM tag: the reference dereferences NULL here"
    s = cigar_qlen - ed + del_len
    if np.float32(s) < np.float32(map_qual) * np.float32(cigar_qlen):          # :81 (int < float * int: float arithmetic)
        return None
    if remove_overlap(rec, tid, spans):                                        # :82
        return None
    return s, intron_n                                                         # :83


def remove_overlap(rec: Record, tid: int, spans) -> bool:
    """src/bam_filter.c:48-59; spans = (tid, start, end) of the -r transcripts in file order.  pos is 0-based, the
    transcript coordinates 1-based: compared as they are."""
    pos = rec.pos - 1
    rlen = sum(l for (l, op) in rec.cigar if op in REF_CONSUMING)
    for (t, start, end) in spans:
        if tid == t and not (pos > end or start > pos + rlen - 1):
            return True
        if tid < t:
            return False
    return False


def select(recs: Sequence[Record], tids: Sequence[int], cov_rate=COV_RATIO, map_qual=MAP_QUAL, sec_rat=SEC_RATIO,
           min_intron_n=MIN_INTRON_NUM, spans=()) -> List[int]:
    """bam_filter() src/bam_filter.c:128-154: indices of the records that are written, in order."""
    out: List[int] = []
    lqname, best, b_score, s_score, b_intron = "", -1, 0, 0, 0
    sec = np.float32(sec_rat)

    def flush():
        if lqname != "" and np.float32(s_score) < sec * np.float32(b_score) and b_intron >= min_intron_n:     # :141, :149
            out.append(best)

    for i, rec in enumerate(recs):
        r = score_record(rec, tids[i], cov_rate, map_qual, spans)
        if r is None:                                                          # :129 continue
            continue
        score, intron_n = r
        if rec.qname == lqname:                                                # :131
            if score > b_score:
                best, s_score, b_score, b_intron = i, b_score, score, intron_n
            elif score > s_score:
                s_score = score
        else:
            flush()
            best, b_score, s_score, b_intron, lqname = i, score, 0, intron_n, rec.qname
    flush()
    return out


# ---------------------------------------------------------------------------------------- SAM -> BAM (SAMv1 4.2)

def reg2bin(beg: int, end: int) -> int:
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def encode_aux(a: str) -> bytes:
    tag, typ, val = a[:2].encode(), a[3], a[5:]
    if typ == "A":
        return tag + b"A" + val[:1].encode()
    if typ == "i":
        x = int(val)
        if x < 0:
            fmt = ("c", "<b") if x >= -128 else ("s", "<h") if x >= -32768 else ("i", "<i")
        else:
            fmt = ("C", "<B") if x <= 255 else ("S", "<H") if x <= 65535 else ("I", "<I")
        return tag + fmt[0].encode() + struct.pack(fmt[1], x)
    if typ == "f":
        return tag + b"f" + struct.pack("<f", float(val))
    if typ in "ZH":
        return tag + typ.encode() + val.encode() + b"\0"
    if typ == "B":
        parts = val.split(",")
        st = parts[0]
        code = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[st]
        vals = [float(x) if st == "f" else int(x) for x in parts[1:]]
        return tag + b"B" + st.encode() + struct.pack("<I", len(vals)) + struct.pack("<%d%s" % (len(vals), code), *vals)
    raise ValueError(a)


def encode_record(rec: Record, ref_index: dict) -> bytes:
    tid = -1 if rec.rname == "*" else ref_index[rec.rname]
    mtid = tid if rec.rnext == "=" else (-1 if rec.rnext == "*" else ref_index.get(rec.rnext, -1))
    rlen = sum(l for (l, op) in rec.cigar if op in REF_CONSUMING) if rec.cigar else 1
    l_seq = 0 if rec.seq == "*" else len(rec.seq)
    name = rec.qname.encode() + b"\0"
    cig = b"".join(struct.pack("<I", (l << 4) | op) for (l, op) in rec.cigar)
    seq = bytearray((l_seq + 1) // 2)
    for k in range(l_seq):
        seq[k >> 1] |= NT16.get(rec.seq[k].upper(), 15) << (0 if k & 1 else 4)
    qual = (b"\xff" * l_seq) if rec.qual == "*" else bytes(ord(ch) - 33 for ch in rec.qual)
    aux = b"".join(encode_aux(a) for a in rec.aux)
    n_cig = len(rec.cigar)
    if n_cig > 65535:                                    # the real CIGAR moves into CG:B,I (SAMv1 4.2.2)
        aux += b"CGBI" + struct.pack("<I", n_cig) + cig
        cig = struct.pack("<II", (l_seq << 4) | 4, (rlen << 4) | 3)
        n_cig = 2
    pos0 = rec.pos - 1
    core = struct.pack("<iiBBHHHIiii", tid, pos0, len(name), rec.mapq, reg2bin(pos0, pos0 + rlen), n_cig, rec.flag & 0xffff,
                       l_seq, mtid, rec.pnext - 1, rec.tlen)
    body = core + name + cig + bytes(seq) + qual + aux
    return struct.pack("<I", len(body)) + body


def header_bytes(header_lines: Sequence[str], refs: Sequence[Tuple[str, int]]) -> bytes:
    text = "".join(header_lines).encode()
    out = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(refs))
    for (name, ln) in refs:
        out += struct.pack("<I", len(name) + 1) + name.encode() + b"\0" + struct.pack("<I", ln)
    return out


def expected_stream(sam_path: str, spans=(), **opts) -> Tuple[bytes, List[int]]:
    """The uncompressed BAM stream of `lr2rmats filter [opts] sam_path`, and the indices of the records in it."""
    header, refs, recs = parse_sam(sam_path)
    idx = {name: i for i, (name, _) in enumerate(refs)}
    tids = [-1 if r.rname == "*" else idx[r.rname] for r in recs]
    keep = select(recs, tids, spans=spans, **opts)
    return header_bytes(header, refs) + b"".join(encode_record(recs[i], idx) for i in keep), keep


def bgzf_blocks(data: bytes, payload: int = 0xff00, level: int = 1) -> bytes:
    """`data` as a BGZF file (tests feed it to the product as a BAM input)."""
    import zlib
    out = bytearray()
    for at in list(range(0, len(data), payload)) + [None]:
        chunk = b"" if at is None else data[at:at + payload]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp
        out += struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    return bytes(out)
