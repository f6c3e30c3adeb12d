#!/usr/bin/env python3
"""Hand-derived known answers for the ORDER-DEPENDENT half of update-gtf / unique-gtf.

Every byte of every file this script writes is a literal below: nothing here calls the oracle, the
product, or any code that computes an expected value.  The expected outputs were worked out BY HAND
from the reference's rules (file:line into /root/reference/src), and the derivation of each row is
in README.md next to this file.  Running the script only re-materialises the committed files:

    python tests/golden/hand/literal_files.py            # rewrites the files in this directory

The tests (tests/test_hand_known_answers.py) compare the oracle CLI (CPU suite) and the HIP CLI
(GPU suite) with the committed files byte for byte.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
T = "\t"


def gtf_rows(chrom, strand, gid, gname, tid, tname, exons):
    """One `transcript` row (ignored by the reader, src/gtf.c:478) + one `exon` row per exon, in the order given."""
    attr = 'gene_id "%s"; transcript_id "%s"; gene_name "%s"; transcript_name "%s";' % (gid, tid, gname, tname)
    lo, hi = min(s for s, _ in exons), max(e for _, e in exons)
    rows = [T.join([chrom, "hand", "transcript", str(lo), str(hi), ".", strand, ".", attr])]
    rows += [T.join([chrom, "hand", "exon", str(s), str(e), ".", strand, ".", attr]) for s, e in exons]
    return rows


# --------------------------------------------------------------------------------------------------
# Annotation (GTF FILE ORDER matters: the sweep of check_with_anno_trans walks it, update_gtf.c:792-835).
# TFAR sits in front of TA2 although TA2 starts earlier (unsorted GTF, GENCODE style): a read that ends
# before TFAR stops there and never sees TA2 (:799-800); TR1 is a '-' gene written with descending exons.
ANNO = []
ANNO += gtf_rows("chrA", "+", "GA", "ga", "TA1", "ta1", [(1000, 1100), (2000, 2100), (3000, 3100), (4000, 4100)])
ANNO += gtf_rows("chrA", "+", "GF", "gf", "TFAR", "tfar", [(9000, 9100), (9200, 9300)])
ANNO += gtf_rows("chrA", "-", "GN", "gn", "TA2", "ta2", [(2900, 2950), (2000, 2100)])
ANNO += gtf_rows("chrA", "-", "GR", "gr", "TR1", "tr1", [(13000, 13100), (12000, 12100), (11000, 11100)])
ANNO += gtf_rows("chrB", "+", "GB", "gb", "TB1", "tb1", [(1000, 1100), (2000, 2100), (3000, 3100), (4000, 4100)])
ANNO += gtf_rows("chrB", "+", "GC", "gc", "TC1", "tc1", [(20000, 20100), (21000, 21100), (22000, 22100), (23000, 23100)])
ANNO += gtf_rows("chrB", "+", "GS", "gs", "TS1", "ts1", [(50000, 50099)])

SAM_HEADER = ["@HD\tVN:1.0\tSO:coordinate", "@SQ\tSN:chrA\tLN:100000000", "@SQ\tSN:chrB\tLN:100000000"]


def sam(name, flag, chrom, pos, cigar, *tags):
    return T.join([name, str(flag), chrom, str(pos), "60", cigar, "*", "0", "0", "*", "*"] + list(tags))


# --------------------------------------------------------------------------------------------------
# Case "upd": update-gtf -l 5  (and the same with -c).  README.md section 1.
UPD_SAM = SAM_HEADER + [
    sam("a1", 0, "chrA", 1000, "101M899N101M399N101M399N101M899N101M"),
    sam("a2", 0, "chrA", 1000, "101M899N101M399N101M399N101M899N201M"),
    sam("c1", 0, "chrA", 1000, "101M899N101M899N51M"),
    sam("c2", 0, "chrA", 1000, "101M899N101M899N6501M"),
    sam("k1", 0, "chrA", 2000, "101M899N101M899N101M"),
    sam("a3", 0, "chrA", 2050, "51M399N101M399N51M"),
    sam("r1", 0, "chrA", 11000, "101M899N101M399N101M"),
    sam("u1", 0, "chrA", 20000, "101M399N101M"),
    sam("d2", 16, "chrB", 1000, "101M949N51M1399N101M"),
    sam("b1", 0, "chrB", 2050, "51M899N101M"),
    sam("b2", 0, "chrB", 2050, "51M899N101M899N151M"),
    sam("s1", 0, "chrB", 50020, "100M"),
    sam("s2", 0, "chrB", 50021, "100M"),
]

DETAIL_HEADER = T.join(["ReadName", "chr", "strand", "Novel", "GeneID", "GeneName", "ExonCount", "ExonStart", "ExonEnd", "NovelExonCount",
                        "NovelExonIndex", "NovelSiteCount", "NovelSiteIndex", "NovelJunctionCount", "NovelJunctionIndex",
                        "UnreliableJunctionCount", "UnreliableJunctionIndex"])


def detail(name, chrom, strand, novel, gid, gname, starts, ends, nov_exon, nov_site, nov_junc, unrel):
    """One detail.txt row (update_gtf.c:297-419): every field is followed by a tab except a non-empty last list."""
    def lst(v):
        return (str(len(v)), ",".join(str(x) for x in v) if v else "NA")
    f = [name, chrom, strand, str(novel), gid, gname, str(len(starts)), ",".join(map(str, starts)), ",".join(map(str, ends))]
    for v in (nov_exon, nov_site, nov_junc):
        f += list(lst(v))
    f += list(lst(unrel))
    return T.join(f) + ("" if unrel else T)


UPD_DETAIL = [
    DETAIL_HEADER,
    detail("a1", "chrA", "+", 1, "GA", "ga", [1000, 2000, 2500, 3000, 4000], [1100, 2100, 2600, 3100, 4100], [2], [1, 4, 5], [1, 2], []),
    detail("a2", "chrA", "+", 1, "GA", "ga", [1000, 2000, 2500, 3000, 4000], [1100, 2100, 2600, 3100, 4200], [2, 4], [1, 4, 5], [1, 2], []),
    detail("c1", "chrA", "+", 1, "GA", "ga", [1000, 2000, 3000], [1100, 2100, 3050], [2], [1], [], []),
    detail("c2", "chrA", "-", 1, "GN", "gn", [1000, 2000, 3000], [1100, 2100, 9500], [2], [1], [], []),
    detail("k1", "chrA", "+", 0, "GA", "ga", [2000, 3000, 4000], [2100, 3100, 4100], [], [], [], []),
    detail("a3", "chrA", "+", 1, "GA", "ga", [2050, 2500, 3000], [2100, 2600, 3050], [0, 1, 2], [1, 2, 3], [0, 1], []),
    detail("r1", "chrA", "-", 1, "GR", "gr", [11000, 12000, 12500], [11100, 12100, 12600], [2], [1], [1], []),
    detail("u1", "chrA", "+", 2, "NA", "NA", [20000, 20500], [20100, 20600], [0, 1], [0, 1], [0], []),
    detail("d2", "chrB", "+", 1, "GB", "gb", [1000, 2050, 3500], [1100, 2100, 3600], [1, 2], [1, 3], [0, 1], []),
    detail("b1", "chrB", "+", 1, "GB", "gb", [2050, 3000], [2100, 3100], [0], [1], [], []),
    detail("b2", "chrB", "+", 1, "GB", "gb", [2050, 3000, 4000], [2100, 3100, 4150], [0, 2], [1], [], []),
    detail("s1", "chrB", "+", 0, "GS", "gs", [50020], [50119], [0], [], [], []),
    detail("s2", "chrB", "+", 2, "NA", "NA", [50021], [50120], [0], [], [], []),
]


def gtf_block(tchrom, start, end, strand, gid, gname, name, cov, xchrom, xstrand, exons):
    """print_read_trans (gtf.c:607-632): transcript row with transcript_cov, exon rows without; exons descending when
    the TRANSCRIPT is '-'; every exon row carries its own chromosome and strand."""
    attr = 'gene_id "%s"; transcript_id "%s"; gene_name "%s"; transcript_name "%s";' % (gid, name, gname, name)
    rows = [T.join([tchrom, "lr2rmats", "transcript", str(start), str(end), ".", strand, ".", attr + ' transcript_cov "%d";' % cov])]
    order = list(reversed(exons)) if strand == "-" else list(exons)
    rows += [T.join([xchrom, "lr2rmats", "exon", str(s), str(e), ".", xstrand, ".", attr]) for s, e in order]
    return rows


_A = gtf_block("chrA", 1000, 4200, "+", "GA", "ga", "a1", 2, "chrA", "+", [(1000, 1100), (2000, 2100), (2500, 2600), (3000, 3100), (4000, 4200)])
_R1 = gtf_block("chrA", 11000, 12600, "-", "GR", "gr", "r1", 1, "chrA", "-", [(11000, 11100), (12000, 12100), (12500, 12600)])
_D2 = gtf_block("chrB", 1000, 3600, "+", "GB", "gb", "d2", 1, "chrB", "+", [(1000, 1100), (2050, 2100), (3500, 3600)])
_B1 = gtf_block("chrB", 2050, 3100, "+", "GB", "gb", "b1", 1, "chrB", "+", [(2050, 2100), (3000, 3100)])
# without -c: c2 merges into c1's entry (cov 2, last exon and transcript end extended to 9500, strand stays '+')
UPD_GTF = _A + gtf_block("chrA", 1000, 9500, "+", "GA", "ga", "c1", 2, "chrA", "+", [(1000, 1100), (2000, 2100), (3000, 9500)]) + _R1 + _D2 + _B1
# with -c: c2 ('-' after it took TA2's strand) is skipped at c1's and a1's entries (update_gtf.c:149) and becomes its own entry
UPD_GTF_C = (_A + gtf_block("chrA", 1000, 3050, "+", "GA", "ga", "c1", 1, "chrA", "+", [(1000, 1100), (2000, 2100), (3000, 3050)])
             + gtf_block("chrA", 1000, 9500, "-", "GN", "gn", "c2", 1, "chrA", "-", [(1000, 1100), (2000, 2100), (3000, 9500)]) + _R1 + _D2 + _B1)


def bed(chrom, start, end, kind, score, strand):
    return T.join([chrom, str(start - 1), str(end), kind + "_exon", str(score), strand])


UPD_BED = [bed("chrA", 2500, 2600, "I", 2, "+"), bed("chrA", 3000, 9500, "T", 2, "+"), bed("chrA", 12500, 12600, "T", 1, "-"),
           bed("chrB", 2050, 2100, "I", 2, "+"), bed("chrB", 3500, 3600, "T", 1, "+")]
UPD_BED_C = [bed("chrA", 2500, 2600, "I", 2, "+"), bed("chrA", 3000, 3050, "T", 1, "+"), bed("chrA", 3000, 9500, "T", 1, "-"),
             bed("chrA", 12500, 12600, "T", 1, "-"), bed("chrB", 2050, 2100, "I", 2, "+"), bed("chrB", 3500, 3600, "T", 1, "+")]


def summary(anno_genes, anno_tx, upd_genes, added, partial, exons, sites, juncs, known, known_genes, uniq_known,
            reliable, uniq_reliable, unreliable, uniq_unreliable, unrec, uniq_unrec):
    """update_gtf.c:530-569, literal text."""
    return [
        "==== Annotaion ====", "Genes_of_annotation_GTF\t%d" % anno_genes, "Transcripts_of_annotation_GTF\t%d" % anno_tx, "",
        "===================", "", "==== Updated information ====", "Updated_Genes\t%d" % upd_genes, "Added_Novel_Transcripts\t%d" % added,
        "Added_Novel_Full-read_Transcripts\t%d" % (added - partial), "Added_Novel_Partial-read_Transcripts\t%d" % partial,
        "Added_Novel_Exons\t%d" % exons, "Added_Novel_Sites\t%d" % sites, "Added_Novel_Splice_Junctions\t%d" % juncs, "",
        "=============================", "", "==== Known information ====", "Known_Transcripts_from_BAM\t%d" % known,
        "Genes_of_Known_Transcripts_from_BAM\t%d" % known_genes, "Uniq_Known_Transcripts_from_BAM\t%d" % uniq_known, "",
        "===========================", "", "==== Novel information ====", "Novel_Transcript_from_BAM\t%d" % (reliable + unreliable),
        "Novel_Transcript_from_BAM_with_All_Reliable_Junction\t%d" % reliable,
        "Uniq_Novel_Transcript_from_BAM_with_All_Reliable_Junction\t%d" % uniq_reliable,
        "Novel_Transcript_from_BAM_with_Unreliable_Junction\t%d" % unreliable,
        "Uniq_Novel_Transcript_from_BAM_with_Unreliable_Junction\t%d" % uniq_unreliable, "",
        "===========================", "", "==== Unrecognized information ====", "Unrecognized_Transcript_from_BAM\t%d" % unrec,
        "Uniq_Unrecognized_Transcript_from_BAM\t%d" % uniq_unrec, "", "=================================="]


UPD_SUMMARY = summary(7, 7, 3, 5, 0, 5, 7, 5, 2, 2, 2, 9, 5, 0, 0, 2, 2)
# -c: c2 is its own entry (gene GN), its novel exon its own bed row; the unique novel list is merged with -c as well
UPD_SUMMARY_C = summary(7, 7, 4, 6, 0, 6, 7, 5, 2, 2, 2, 9, 6, 0, 0, 2, 2)

# --------------------------------------------------------------------------------------------------
# Case "uniq": unique-gtf (and the same with -s = force strand).  README.md section 2.
UNIQ_SAM = SAM_HEADER + [
    sam("u1", 0, "chrA", 100, "101M99N101M99N101M"),
    sam("u2", 0, "chrA", 90, "111M99N101M99N151M"),
    sam("u3", 0, "chrA", 95, "106M99N101M99N121M"),
    sam("u4", 0, "chrA", 150, "51M99N101M"),
    sam("v1", 0, "chrA", 1000, "101M99N101M"),
    sam("v2", 0, "chrA", 1000, "101M99N101M99N101M"),
    sam("w1", 0, "chrA", 2000, "101M99N101M99N101M99N101M"),
    sam("w2", 0, "chrA", 2450, "51M99N101M99N101M"),
    sam("x1", 0, "chrA", 3000, "101M499N101M199N101M"),
    sam("x2", 0, "chrA", 3200, "101M99N101M"),
    sam("x3", 0, "chrA", 3650, "51M199N101M"),
    sam("y1", 0, "chrA", 5000, "101M99N101M"),
    sam("y2", 16, "chrA", 5000, "101M99N101M"),
    sam("z1", 0, "chrA", 6000, "100M"),
    sam("z2", 0, "chrA", 6020, "100M"),
    sam("z3", 0, "chrA", 6041, "100M"),
    sam("q1", 0, "chrB", 1000, "50000001M"),
    sam("q2", 0, "chrB", 10001001, "50000001M"),
]


def ublock(chrom, start, end, strand, name, cov, exons):
    return gtf_block(chrom, start, end, strand, name, name, name, cov, chrom, strand, exons)


_U = [
    ublock("chrA", 90, 650, "+", "u1", 3, [(90, 200), (300, 400), (500, 650)]),
    ublock("chrA", 1000, 1300, "+", "v1", 1, [(1000, 1100), (1200, 1300)]),
    ublock("chrA", 2000, 2700, "+", "w1", 1, [(2000, 2100), (2200, 2300), (2400, 2500), (2600, 2700)]),
    ublock("chrA", 3000, 4000, "+", "x1", 1, [(3000, 3100), (3600, 3700), (3900, 4000)]),
    ublock("chrA", 3200, 3500, "+", "x2", 1, [(3200, 3300), (3400, 3500)]),
    ublock("chrA", 3650, 4000, "+", "x3", 1, [(3650, 3700), (3900, 4000)]),
]
_Z = [
    ublock("chrA", 6000, 6119, "+", "z1", 2, [(6000, 6119)]),
    ublock("chrA", 6041, 6140, "+", "z3", 1, [(6041, 6140)]),
    ublock("chrB", 1000, 60001001, "+", "q1", 2, [(1000, 60001001)]),
]
UNIQ_GTF = sum(_U, []) + ublock("chrA", 5000, 5300, "+", "y1", 2, [(5000, 5100), (5200, 5300)]) + sum(_Z, [])
UNIQ_GTF_S = (sum(_U, []) + ublock("chrA", 5000, 5300, "+", "y1", 1, [(5000, 5100), (5200, 5300)])
              + ublock("chrA", 5000, 5300, "-", "y2", 1, [(5000, 5100), (5200, 5300)]) + sum(_Z, []))

# --------------------------------------------------------------------------------------------------
# Case "split": update-gtf -s -l 5 -J 1 -j split_sj.tab.  README.md section 3.
SPLIT_SAM = SAM_HEADER + [
    sam("a1", 0, "chrA", 1000, "101M899N101M399N101M399N101M899N101M"),
    sam("p1", 0, "chrB", 1000, "101M899N101M399N101M399N101M899N101M899N101M"),
    sam("w1", 0, "chrB", 20050, "51M899N101M"),
    sam("p2", 0, "chrB", 20900, "51M69N81M899N101M1399N101M"),
    sam("w2", 0, "chrB", 20960, "21M19N101M899N101M"),
]
# STAR SJ.out.tab columns: chr, first intron base, last intron base, strand, motif, annotated, unique, multi, overhang
SPLIT_SJ = [T.join(r) for r in [
    ("chrA", "2101", "2499", "1", "1", "0", "5", "0", "30"),
    ("chrA", "2601", "2999", "1", "1", "0", "5", "0", "30"),
    ("chrB", "2101", "2499", "1", "1", "0", "5", "0", "30"),
    ("chrB", "2601", "2999", "1", "1", "0", "5", "0", "30"),
    ("chrB", "20981", "20999", "1", "1", "0", "3", "0", "30"),
    ("chrB", "90000", "90100", "1", "1", "0", "9", "0", "30"),
]]
SPLIT_DETAIL = [
    DETAIL_HEADER,
    detail("a1", "chrA", "+", 1, "GA", "ga", [1000, 2000, 2500, 3000, 4000], [1100, 2100, 2600, 3100, 4100], [2], [1, 4, 5], [1, 2], []),
    detail("p1", "chrB", "+", 1, "GB", "gb", [1000, 2000, 2500, 3000, 4000, 5000], [1100, 2100, 2600, 3100, 4100, 5100], [2, 5], [1, 4, 5, 8], [1, 2, 4], [4]),
    detail("w1", "chrB", "+", 1, "GC", "gc", [20050, 21000], [20100, 21100], [0], [1], [], []),
    detail("p2", "chrB", "+", 1, "GC", "gc", [20900, 21020, 22000, 23500], [20950, 21100, 22100, 23600], [0, 1, 3], [0, 1, 3], [0, 2], [0, 2]),
    detail("w2", "chrB", "+", 1, "GC", "gc", [20960, 21000, 22000], [20980, 21100, 22100], [0], [0, 1], [0], []),
]
SPLIT_GTF = (
    # a1's entry got cov 2 from the split piece of the chrB read p1 (Q2: the piece has tid 0 and is compared with every entry)
    gtf_block("chrA", 1000, 4100, "+", "GA", "ga", "a1", 2, "chrA", "+", [(1000, 1100), (2000, 2100), (2500, 2600), (3000, 3100), (4000, 4100)])
    + gtf_block("chrB", 20050, 21100, "+", "GC", "gc", "w1", 1, "chrB", "+", [(20050, 20100), (21000, 21100)])
    # Q2: the piece's transcript row prints chr_name[0], 0, 0, '+'; its exon rows carry the real chromosome and strand
    + gtf_block("chrA", 0, 0, "+", "GC", "gc", "p2.split.0", 1, "chrB", "+", [(21020, 21100), (22000, 22100)])
    + gtf_block("chrB", 20960, 22100, "+", "GC", "gc", "w2", 1, "chrB", "+", [(20960, 20980), (21000, 21100), (22000, 22100)]))
SPLIT_BED = [bed("chrA", 2500, 2600, "I", 2, "+"), bed("chrB", 20050, 20100, "T", 1, "+"), bed("chrB", 21020, 21100, "T", 1, "+"),
             bed("chrB", 20960, 20980, "T", 1, "+")]
SPLIT_SUMMARY = summary(7, 7, 2, 4, 1, 4, 7, 3, 0, 0, 0, 3, 3, 2, 2, 0, 0)

# --------------------------------------------------------------------------------------------------
# Case "dis": update-gtf -l 5 -d 2 (and -d 0 beside it) on an annotation with TWO donors inside one read donor's tolerance.
# README.md section 4.  check_splice_site (update_gtf.c:717-779) counts identical_site_n per (annotation site, read site) PAIR.
DIS_ANNO = gtf_rows("chrA", "+", "GD", "gd", "TD1", "td1", [(500, 600), (1000, 1100), (1102, 1103), (2000, 2100)])
DIS_SAM = SAM_HEADER + [
    sam("rd1", 0, "chrA", 1000, "102M898N101M"),      # exons (1000,1101) (2000,2100): its donor 1101 is within 2 of the donors 1100 AND 1103
    sam("rd2", 0, "chrA", 1000, "101M899N101M"),      # exons (1000,1100) (2000,2100): its donor 1100 is within 2 of 1100 only
]
# -d 2: rd1 -- acceptor pair (1000, its own first start: Q1) + donor pairs (1100,1101) and (1103,1101) = 3 pairs for 2 read sites: NOT known
# (:770), has a known site; every novel flag is cleared (exon 0 within 2 of (1000,1100), junction within 2 of (1103,2000)).
# rd2 -- pairs (1000,1000) and (1100,1100), donor 1103 is 3 away: 2 pairs = 2 sites = known; its junction (1100,2000) is no
# annotated junction within 2 ((1100,1102) and (1103,2000) both miss one end), so that flag stays.
DIS2_DETAIL = [
    DETAIL_HEADER,
    detail("rd1", "chrA", "+", 1, "GD", "gd", [1000, 2000], [1101, 2100], [], [], [], []),
    detail("rd2", "chrA", "+", 0, "GD", "gd", [1000, 2000], [1100, 2100], [], [], [0], []),
]
# -d 0: rd1's donor 1101 matches nothing (site 0, exon 0 and junction 0 stay novel), its acceptor pair still makes it "has known site"
DIS0_DETAIL = [
    DETAIL_HEADER,
    detail("rd1", "chrA", "+", 1, "GD", "gd", [1000, 2000], [1101, 2100], [0], [0], [0], []),
    detail("rd2", "chrA", "+", 0, "GD", "gd", [1000, 2000], [1100, 2100], [], [], [0], []),
]
DIS_GTF = gtf_block("chrA", 1000, 2100, "+", "GD", "gd", "rd1", 1, "chrA", "+", [(1000, 1101), (2000, 2100)])      # rd2 is known: not in the updated list

# --------------------------------------------------------------------------------------------------
# Case "ends": update-gtf -l 1 / -l 2 / -l 4: the full-length rules (check_full / set_full, update_gtf.c:629-696) seen through the
# updated GTF (only full-length reads are routed, :943).  README.md section 5.
ENDS_ANNO = gtf_rows("chrA", "+", "GE", "ge", "TE1", "te1", [(1000, 1100), (2000, 2100), (3000, 3100)])
ENDS_SAM = SAM_HEADER + [
    sam("e4", 0, "chrA", 1020, "71M909N101M899N101M"),     # (1020,1090) (2000,2100) (3000,3100): first exons overlap, first ends differ
    sam("e1", 0, "chrA", 1050, "51M899N101M899N51M"),      # (1050,1100) (2000,2100) (3000,3050): first end and last start are the annotation's
    sam("e2", 0, "chrA", 1150, "101M749N101M899N101M"),    # (1150,1250) ...: the first exon overlaps NO annotation exon
    sam("e3", 0, "chrA", 2050, "51M899N101M"),             # (2050,2100) (3000,3100): the first exon overlaps the annotation's SECOND exon
]
ENDS_DETAIL = [
    DETAIL_HEADER,
    detail("e4", "chrA", "+", 1, "GE", "ge", [1020, 2000, 3000], [1090, 2100, 3100], [0], [0, 1], [0], []),
    detail("e1", "chrA", "+", 1, "GE", "ge", [1050, 2000, 3000], [1100, 2100, 3050], [0, 2], [1], [], []),
    detail("e2", "chrA", "+", 1, "GE", "ge", [1150, 2000, 3000], [1250, 2100, 3100], [0], [0, 1], [0], []),
    detail("e3", "chrA", "+", 1, "GE", "ge", [2050, 3000], [2100, 3100], [0], [1], [], []),
]
_E4 = gtf_block("chrA", 1020, 3100, "+", "GE", "ge", "e4", 1, "chrA", "+", [(1020, 1090), (2000, 2100), (3000, 3100)])
_E1 = gtf_block("chrA", 1050, 3050, "+", "GE", "ge", "e1", 1, "chrA", "+", [(1050, 1100), (2000, 2100), (3000, 3050)])
_E2 = gtf_block("chrA", 1150, 3100, "+", "GE", "ge", "e2", 1, "chrA", "+", [(1150, 1250), (2000, 2100), (3000, 3100)])
ENDS_GTF_L1 = _E1                     # -l 1: first END and last START must be the annotation's: e1 alone
ENDS_GTF_L2 = _E4 + _E1               # -l 2: terminal exons must overlap the annotation's terminal exons: e4 too
ENDS_GTF_L4 = _E4 + _E1 + _E2         # -l 4: left side only, and a first exon that overlaps NO annotation exon counts (lnoth): e2 too; e3's overlaps an inner one

# Case "ends3": the same annotation with two more reads in front -- -l 3 (both ends, README.md section 7) against -l 4 and -l 2, and what
# merge_trans makes of reads whose junction chains contain each other once they are routed.
ENDS3_SAM = SAM_HEADER + [
    sam("e5", 0, "chrA", 1000, "101M899N51M"),             # (1000,1100) (2000,2050): the LAST exon overlaps the annotation's second exon only
    sam("e6", 0, "chrA", 1000, "101M899N101M399N101M"),    # (1000,1100) (2000,2100) (2500,2600): the last exon overlaps NO annotation exon
] + ENDS_SAM[len(SAM_HEADER):]
ENDS3_DETAIL = [
    DETAIL_HEADER,
    detail("e5", "chrA", "+", 1, "GE", "ge", [1000, 2000], [1100, 2050], [1], [1], [], []),
    detail("e6", "chrA", "+", 1, "GE", "ge", [1000, 2000, 2500], [1100, 2100, 2600], [2], [1], [1], []),
] + ENDS_DETAIL[1:]
_E5 = gtf_block("chrA", 1000, 2050, "+", "GE", "ge", "e5", 1, "chrA", "+", [(1000, 1100), (2000, 2050)])
_E6 = gtf_block("chrA", 1000, 2600, "+", "GE", "ge", "e6", 1, "chrA", "+", [(1000, 1100), (2000, 2100), (2500, 2600)])
ENDS3_GTF_L2 = _E4 + _E1               # e5, e6: their last exons do not overlap the annotation's last exon
ENDS3_GTF_L3 = _E6 + _E4 + _E1 + _E2   # e6's last exon overlaps nothing (rnoth): full; e5's overlaps an inner exon: not; nothing merges
ENDS3_GTF_L4 = _E5 + _E4 + _E2         # left side only: e5 is routed first, and e6 and e1 -- chains that BEGIN with e5's only junction -- vanish into it (Q8)

# --------------------------------------------------------------------------------------------------
# Case "cigar": CIGAR -> exons (gen_exon, bam2gtf.c:31-78) through `bam2gtf` and through `update-gtf -l 5 -A` with an annotation on
# another chromosome, under the default thresholds (-e 3 -i 3 -t 50) and under -e 10 -i 100 -t 5.  README.md section 6.
CIG_SAM = SAM_HEADER + [
    sam("g1", 0, "chrA", 1000, "5S50M2I50M100N30M"),
    sam("g2", 0, "chrA", 2000, "100M200N2M300N100M"),
    sam("g3", 0, "chrA", 3000, "50M2N50M10N50M"),
    sam("g4", 0, "chrA", 4000, "50M50D50M51D50M"),
    sam("g5", 0, "chrA", 5000, "1M100N2M100N50M"),
    sam("g6", 0, "chrA", 6000, "3H10=5X100N1M"),
    sam("g7", 16, "chrA", 7000, "50M100N50M", "XS:A:+"),
    sam("g8", 0, "chrA", 7500, "50M100N50M", "XS:A:-"),
    sam("g11", 0, "chrA", 8000, "20M100N10M100N20M"),
    sam("g12", 0, "chrA", 8500, "20M100N9M100N20M"),
    sam("g9", 0, "chrA", 9000, "50M100N50M", "XS:i:5"),
    sam("g10", 4, "*", 0, "*"),
]
CIG_SAM_MAPPED = CIG_SAM[:-1]          # (update-gtf: read_bam_trans does not skip an unmapped record -- it dies on it, bam2gtf.c:95,100)
CIG_ANNO = gtf_rows("chrB", "+", "GX", "gx", "TX1", "tx1", [(1000, 1100), (2000, 2100)])


def b2g(chrom, strand, name, exons):
    """print_trans (gtf.c:597-604): transcript row + exon rows, gene_id and transcript_id only, always ascending."""
    attr = 'gene_id "%s"; transcript_id "%s";' % (name, name)
    rows = [T.join([chrom, "lr2rmats", "transcript", str(exons[0][0]), str(exons[-1][1]), ".", strand, ".", attr])]
    return rows + [T.join([chrom, "lr2rmats", "exon", str(a), str(b), ".", strand, ".", attr]) for a, b in exons]


# name -> (strand, exons under the defaults, exons under -e 10 -i 100 -t 5 where they differ)
_CIG = [
    ("g1", "+", [(1000, 1099), (1200, 1229)], None),
    ("g2", "+", [(2000, 2099), (2602, 2701)], None),
    ("g3", "+", [(3000, 3101), (3112, 3161)], [(3000, 3161)]),
    ("g4", "+", [(4000, 4149), (4201, 4250)], [(4000, 4049), (4100, 4149), (4201, 4250)]),
    ("g5", "+", [(5000, 5000), (5203, 5252)], None),
    ("g6", "+", [(6000, 6014), (6115, 6115)], None),
    ("g7", "+", [(7000, 7049), (7150, 7199)], None),
    ("g8", "-", [(7500, 7549), (7650, 7699)], None),
    ("g11", "+", [(8000, 8019), (8120, 8129), (8230, 8249)], None),
    ("g12", "+", [(8500, 8519), (8620, 8628), (8729, 8748)], [(8500, 8519), (8729, 8748)]),
    ("g9", "-", [(9000, 9049), (9150, 9199)], None),
]
CIG_B2G = sum((b2g("chrA", st, nm, ex) for nm, st, ex, _ in _CIG), [])
CIG_B2G_T = sum((b2g("chrA", st, nm, alt or ex) for nm, st, ex, alt in _CIG), [])


def _unrec(name, strand, exons):
    n = len(exons)
    return detail(name, "chrA", strand, 2, "NA", "NA", [a for a, _ in exons], [b for _, b in exons], list(range(n)), list(range(2 * (n - 1))), list(range(n - 1)), [])


CIG_DETAIL = [DETAIL_HEADER] + [_unrec(nm, st, ex) for nm, st, ex, _ in _CIG]
CIG_DETAIL_T = [DETAIL_HEADER] + [_unrec(nm, st, alt or ex) for nm, st, ex, alt in _CIG]


# --------------------------------------------------------------------------------------------------
# Case "sj": update-gtf -l 5 -J 3 -j sj.tab without -s, and the same with -M.  README.md section 8.
SJ_ANNO = (gtf_rows("chrA", "+", "GJ1", "gj1", "TJ1", "tj1", [(1000, 1100), (2000, 2100), (3000, 3100)])
           + gtf_rows("chrA", "+", "GJ2", "gj2", "TJ2", "tj2", [(5000, 5100), (6000, 6100)])
           + gtf_rows("chrA", "+", "GJ3", "gj3", "TJ3", "tj3", [(6900, 7000), (9000, 9100)]))
SJ_SAM = SAM_HEADER + [
    sam("j1", 0, "chrA", 1000, "101M899N101M399N101M399N101M"),
    sam("j2", 0, "chrA", 5000, "101M399N101M"),
    sam("j3", 0, "chrA", 6900, "101M999N101M"),
]
SJ_TAB = [T.join(r) for r in [
    ("chrA", "2101", "2499", "1", "1", "0", "2", "2", "30"),
    ("chrA", "2601", "2999", "1", "1", "0", "5", "0", "30"),
    ("chrA", "7001", "7999", "1", "1", "0", "3", "0", "30"),
]]


def _sj_detail(j1_unrel):
    return [
        DETAIL_HEADER,
        detail("j1", "chrA", "+", 1, "GJ1", "gj1", [1000, 2000, 2500, 3000], [1100, 2100, 2600, 3100], [2], [1, 4, 5], [1, 2], j1_unrel),
        detail("j2", "chrA", "+", 1, "GJ2", "gj2", [5000, 5500], [5100, 5600], [1], [1], [0], []),
        detail("j3", "chrA", "+", 1, "GJ3", "gj3", [6900, 8000], [7000, 8100], [1], [1], [0], []),
    ]


SJ_DETAIL = _sj_detail([1])            # -J 3: the row of junction 1 has 2 unique reads
SJ_DETAIL_M = _sj_detail([])           # -M: 2 unique + 2 multi
_J1 = gtf_block("chrA", 1000, 3100, "+", "GJ1", "gj1", "j1", 1, "chrA", "+", [(1000, 1100), (2000, 2100), (2500, 2600), (3000, 3100)])
_J3 = gtf_block("chrA", 6900, 8100, "+", "GJ3", "gj3", "j3", 1, "chrA", "+", [(6900, 7000), (8000, 8100)])
SJ_GTF = _J3                           # j1 has an unreliable junction, j2 is unsupported without one (Q7): neither is routed without -s
SJ_GTF_M = _J1 + _J3


# --------------------------------------------------------------------------------------------------
# Case "gtfq": the annotation reader's quirks (read_anno_trans / gtf_add_info, gtf.c:317-326,468-521) seen through
# update-gtf -l 5.  README.md section 9.  Rows are written out one by one here (no gtf_rows): their ORDER and their
# attribute TEXT are the subject.
def _exon_row(chrom, s, e, attr):
    return T.join([chrom, "hand", "exon", str(s), str(e), ".", "+", ".", attr])


_Q1_ATTR = 'ref_gene_id "RG1"; gene_id "G1"; transcript_id "T1"; gene_name "g1"; transcript_name "t1";'      # "gene_id" is first met inside "ref_gene_id"
_QZ_ATTR = 'gene_id "GZ"; transcript_id "TZ"; gene_name "gz"; transcript_name "tz";'
_Q2_ATTR = 'transcript_id "T2"; gene_name "g2";'                                                                 # no gene_id, no transcript_name
_Q4_HEAD = 'gene_id "G4"; transcript_id "T4"; note "'
_Q4_LONG = _Q4_HEAD + "aaaa " * 280 + '"; gene_name "late4"; transcript_name "latet4";'                          # 1 500 bytes of attributes: the names lie behind byte 1 023
_Q4_ATTR = 'gene_id "G4"; transcript_id "T4"; gene_name "g4"; transcript_name "t4";'
_Q5_ATTR = 'gene_id "G5"; transcript_id "T5"; gene_name "g5"; transcript_name "t5";'
_Q5_HEAD = _exon_row("chrA", 21000, 21100, 'gene_id "G5"; transcript_id "T5"; note "')
# the row is cut behind its 1 023rd byte (fgets(line, 1024), gtf.c:476): the second piece reads "bb cc exon dd ee..."
_Q5_LONG = _Q5_HEAD + "b" * (1023 - len(_Q5_HEAD)) + 'bb cc exon dd ee"; gene_name "late5";'
GTFQ_ANNO = [
    _exon_row("chrA", 1000, 1100, _Q1_ATTR),
    "#" + _exon_row("chrA", 1500, 1600, 'gene_id "GC"; transcript_id "TC";'),     # a comment (gtf.c:477); read as a row it would cut T1 in two
    _exon_row("chrA", 2000, 2100, _Q1_ATTR),
    _exon_row("chrZ", 1000, 1100, _QZ_ATTR),                                      # chrZ is not in the header: tid -1, before every read (gtf.c:481)
    _exon_row("chrZ", 2000, 2100, _QZ_ATTR),
    _exon_row("chrA", 5000, 5100, _Q2_ATTR),
    _exon_row("chrA", 6000, 6100, _Q2_ATTR),
    "",                                                                           # an empty line: sscanf assigns nothing, the row in front is taken again
    _exon_row("chrA", 7000, 7100, _Q2_ATTR),
    _exon_row("chrA", 10000, 10100, _Q4_LONG),
    _exon_row("chrA", 11000, 11100, _Q4_ATTR),
    _exon_row("chrA", 20000, 20100, _Q5_ATTR),
    _Q5_LONG,
    _exon_row("chrA", 22000, 22100, _Q5_ATTR),
]
GTFQ_SAM = SAM_HEADER + [
    sam("q1", 0, "chrA", 1000, "101M899N101M"),
    sam("q2", 0, "chrA", 6000, "101M899N101M"),
    sam("q4", 0, "chrA", 10000, "101M899N101M"),
    sam("q5", 0, "chrA", 21000, "101M899N101M"),
]
GTFQ_DETAIL = [
    DETAIL_HEADER,
    detail("q1", "chrA", "+", 1, "RG1", "g1", [1000, 2000], [1100, 2100], [], [1], [], []),
    detail("q2", "chrA", "+", 1, "g2", "g2", [6000, 7000], [6100, 7100], [], [], [], []),
    detail("q4", "chrA", "+", 1, "G4", "G4", [10000, 11000], [10100, 11100], [], [1], [], []),
    detail("q5", "chrA", "+", 1, "G5", "g5", [21000, 22000], [21100, 22100], [], [], [], []),
]
GTFQ_GTF = (gtf_block("chrA", 1000, 2100, "+", "RG1", "g1", "q1", 1, "chrA", "+", [(1000, 1100), (2000, 2100)])
            + gtf_block("chrA", 6000, 7100, "+", "g2", "g2", "q2", 1, "chrA", "+", [(6000, 6100), (7000, 7100)])
            + gtf_block("chrA", 10000, 11100, "+", "G4", "G4", "q4", 1, "chrA", "+", [(10000, 10100), (11000, 11100)])
            + gtf_block("chrA", 21000, 22100, "+", "G5", "g5", "q5", 1, "chrA", "+", [(21000, 21100), (22000, 22100)]))


# --------------------------------------------------------------------------------------------------
# Case "uns": records that are NOT coordinate sorted -- the sweep's cursor only moves forward (check_with_anno_trans,
# update_gtf.c:792-802: `if (*last_anno_i == i) ++(*last_anno_i)`).  README.md section 10.
UNS_ANNO = (gtf_rows("chrA", "+", "GA", "ga", "TA", "ta", [(1000, 1100), (2000, 2100)])
            + gtf_rows("chrA", "+", "GB", "gb", "TB", "tb", [(5000, 5100), (6000, 6100)]))
UNS_SAM = ["@HD\tVN:1.0\tSO:unsorted", "@SQ\tSN:chrA\tLN:100000000", "@SQ\tSN:chrB\tLN:100000000",
           sam("rB", 0, "chrA", 5000, "101M899N101M"),
           sam("rA", 0, "chrA", 1000, "101M899N101M"),
           sam("rC", 0, "chrA", 5000, "101M899N101M")]
UNS_DETAIL = [
    DETAIL_HEADER,
    detail("rB", "chrA", "+", 1, "GB", "gb", [5000, 6000], [5100, 6100], [], [1], [], []),
    detail("rA", "chrA", "+", 2, "NA", "NA", [1000, 2000], [1100, 2100], [0, 1], [0, 1], [0], []),      # the cursor has passed TA for good
    detail("rC", "chrA", "+", 1, "GB", "gb", [5000, 6000], [5100, 6100], [], [1], [], []),
]
UNS_GTF = gtf_block("chrA", 5000, 6100, "+", "GB", "gb", "rB", 2, "chrA", "+", [(5000, 5100), (6000, 6100)])      # rC merges into rB's entry: cov 2


# --------------------------------------------------------------------------------------------------
# Case "mg": `update-gtf -m g -b uns.sam`: read-like transcripts from a GTF instead of alignments (read_gtf_trans, gtf.c:524-595).
# README.md section 11.  Annotation: uns_anno.gtf.
MG_READS = (gtf_rows("chrA", "-", "GX", "gx", "X1", "x1n", [(6000, 6100), (5000, 5100)])
            + [T.join(["chrA", "hand", "exon", "50000", "50099", ".", "+", ".", 'transcript_id "X2"; gene_name "gz";'])])
MG_DETAIL = [
    DETAIL_HEADER,
    detail("x1n", "chrA", "+", 1, "GB", "gb", [5000, 6000], [5100, 6100], [], [1], [], []),
    detail("X2", "chrA", "+", 2, "NA", "NA", [50000], [50099], [0], [], [], []),
]
_MG_ATTR = 'gene_id "GB"; transcript_id "X1"; gene_name "gb"; transcript_name "x1n";'
MG_GTF = [T.join(["chrA", "lr2rmats", "transcript", "5000", "6100", ".", "+", ".", _MG_ATTR + ' transcript_cov "1";']),
          T.join(["chrA", "lr2rmats", "exon", "5000", "5100", ".", "+", ".", _MG_ATTR]),
          T.join(["chrA", "lr2rmats", "exon", "6000", "6100", ".", "+", ".", _MG_ATTR])]


FILES = {
    "anno.gtf": ANNO,
    "cigar.sam": CIG_SAM, "cigar_m.sam": CIG_SAM_MAPPED, "cigar_anno.gtf": CIG_ANNO, "cigar.bam2gtf.gtf": CIG_B2G, "cigar_t.bam2gtf.gtf": CIG_B2G_T,
    "cigar.detail.txt": CIG_DETAIL, "cigar_t.detail.txt": CIG_DETAIL_T,
    "dis_anno.gtf": DIS_ANNO, "dis.sam": DIS_SAM, "dis2.detail.txt": DIS2_DETAIL, "dis0.detail.txt": DIS0_DETAIL, "dis.updated.gtf": DIS_GTF,
    "ends_anno.gtf": ENDS_ANNO, "ends.sam": ENDS_SAM, "ends.detail.txt": ENDS_DETAIL,
    "sj_anno.gtf": SJ_ANNO, "sj.sam": SJ_SAM, "sj.tab": SJ_TAB, "sj.detail.txt": SJ_DETAIL, "sj_m.detail.txt": SJ_DETAIL_M,
    "sj.updated.gtf": SJ_GTF, "sj_m.updated.gtf": SJ_GTF_M,
    "ends3.sam": ENDS3_SAM, "ends3.detail.txt": ENDS3_DETAIL, "ends3_l2.updated.gtf": ENDS3_GTF_L2, "ends3_l3.updated.gtf": ENDS3_GTF_L3,
    "ends3_l4.updated.gtf": ENDS3_GTF_L4,
    "ends_l1.updated.gtf": ENDS_GTF_L1, "ends_l2.updated.gtf": ENDS_GTF_L2, "ends_l4.updated.gtf": ENDS_GTF_L4,
    "upd.sam": UPD_SAM, "upd.detail.txt": UPD_DETAIL, "upd.updated.gtf": UPD_GTF, "upd.novel_exon.bed": UPD_BED, "upd.summary.txt": UPD_SUMMARY,
    "upd_c.updated.gtf": UPD_GTF_C, "upd_c.novel_exon.bed": UPD_BED_C, "upd_c.summary.txt": UPD_SUMMARY_C,
    "uniq.sam": UNIQ_SAM, "uniq.unique.gtf": UNIQ_GTF, "uniq_s.unique.gtf": UNIQ_GTF_S,
    "split.sam": SPLIT_SAM, "split_sj.tab": SPLIT_SJ, "split.detail.txt": SPLIT_DETAIL, "split.updated.gtf": SPLIT_GTF,
    "split.novel_exon.bed": SPLIT_BED, "split.summary.txt": SPLIT_SUMMARY,
    "mg_reads.gtf": MG_READS, "mg.detail.txt": MG_DETAIL, "mg.updated.gtf": MG_GTF,
    "uns_anno.gtf": UNS_ANNO, "uns.sam": UNS_SAM, "uns.detail.txt": UNS_DETAIL, "uns.updated.gtf": UNS_GTF,
    "gtfq_anno.gtf": GTFQ_ANNO, "gtfq.sam": GTFQ_SAM, "gtfq.detail.txt": GTFQ_DETAIL, "gtfq.updated.gtf": GTFQ_GTF,
}

if __name__ == "__main__":
    for name, rows in FILES.items():
        with open(os.path.join(HERE, name), "w") as fh:
            fh.write("".join(r + "\n" for r in rows))
    print("wrote %d files into %s" % (len(FILES), HERE))
