"""GPU: inputs built to break the fast path's assumptions -- against the oracle, bit exact, on every kernel pipeline.

    annotation rows on chromosomes the BAM header does not have (tid -1) between the others
    a locus of 300 isoforms (windows far wider than the 32-bit membership masks: the redo list carries the load)
    duplicated transcripts, duplicated and overlapping exons inside a transcript
    reads that touch a transcript / an exon by exactly one base, at every -l level
    -d 2 with a junction table on ONT-like reads (long CIGARs, micro-exons), with and without -s
and what share of the reads the redo list (k_classify_generic) takes on each of them.
"""
import numpy as np
import pytest

from lr2rmats_amd import synth
from tests import util
from tests.test_gpu_edges import _anno, _chain, _reads, _run, pipeline  # noqa: F401  (pipeline: autouse fixture)

pytestmark = pytest.mark.gpu


def _sorted_rows(rows):
    return sorted(rows, key=lambda r: (r[0], r[1]))


def test_annotation_rows_without_a_chromosome_in_the_header(oracle):
    # read_anno_trans() keeps transcripts of unknown chromosomes with tid -1 (src/gtf.c:468-521); they sit between the
    # others in file order and can never match, but the sweep cursor walks over them
    rng = np.random.default_rng(3)
    txs = []
    for g in range(120):
        base = 5_000 + g * 4_000
        ex = [(base + 300 * k, base + 300 * k + 120) for k in range(int(rng.integers(2, 7)))]
        txs.append((0 if g < 80 else 1, g & 1, ex))
        if g % 3 == 0:
            txs.append((-1, 0, [(base + 10, base + 90), (base + 400, base + 520)]))      # same coordinates, no chromosome
    af = _anno(txs)
    rows = []
    for i in range(2500):
        t = txs[int(rng.integers(len(txs)))]
        if t[0] < 0:
            continue
        ex = [list(x) for x in t[2]]
        if i % 4 == 0:
            ex[0][0] += int(rng.integers(0, 30))
        if i % 5 == 0 and len(ex) > 2:
            del ex[1]
        p, ops = _chain([tuple(x) for x in ex])
        rows.append((t[0], p, i & 1, ops))
    cnt = [0, 0, 0, 0]
    got, want = _run(oracle, af, _reads(_sorted_rows(rows)), counters=cnt, full_level=3)
    # (chains that begin with a transcript's first exon are never "known": Q1 -- they count as "has a known site")
    assert ((want.info & 2) != 0).sum() > 200 and (want.ref_tx >= 0).sum() > 500
    assert not np.isin(want.ref_tx[want.ref_tx >= 0], np.nonzero(af.tx_tid < 0)[0]).any()


@pytest.mark.parametrize("level", [3, 5])
def test_locus_of_300_isoforms(oracle, level, pipeline):
    # every tile of this locus sees ~300 overlapping transcripts: the window fits no mask width and most exons are shared by
    # transcripts more than 64 apart in file order.  Results stay exact; the share of the redo list is reported by the counters.
    rng = np.random.default_rng(9)
    pool = [(20_000 + 600 * k, 20_000 + 600 * k + 140) for k in range(30)]
    txs = []
    for t in range(300):
        keep = sorted(set([0, 29] + list(rng.choice(np.arange(1, 29), size=int(rng.integers(6, 20)), replace=False))))
        ex = [pool[k] for k in keep]
        if t % 5 == 0:
            ex[1] = (ex[1][0] - int(rng.integers(1, 30)), ex[1][1])
        txs.append((0, t & 1, ex))
    # a quiet neighbour locus on the same chromosome: its tiles must stay on the fast path
    for g in range(200):
        base = 400_000 + g * 3_000
        txs.append((0, 0, [(base, base + 100), (base + 500, base + 650), (base + 1200, base + 1300)]))
    af = _anno(txs)
    rows = []
    for i in range(6000):
        t = txs[int(rng.integers(300))][2]
        a = int(rng.integers(0, len(t) - 2))
        ex = [list(x) for x in t[a:a + int(rng.integers(2, 9))]]
        if i % 3 == 0:
            ex[-1][1] -= int(rng.integers(0, 50))
        p, ops = _chain([tuple(x) for x in ex])
        rows.append((0, p, i & 1, ops))
    for i in range(6000):
        t = txs[300 + int(rng.integers(200))][2]
        p, ops = _chain(t if i % 2 else t[:2])
        rows.append((0, p, 0, ops))
    cnt = [0, 0, 0, 0]
    reads = _reads(_sorted_rows(rows))
    got, want = _run(oracle, af, reads, counters=cnt, full_level=level)
    assert ((want.info & 1) != 0).sum() > 1000
    # classic pipeline: the crowded locus is the redo list's, the quiet one is not (one tile may straddle the two); the slab pipeline
    # takes the crowded tiles' windows 63 transcripts at a time (k_probe_slab_chunked, every chunk with the dictionary entries that
    # matter for it, fewer transcripts per chunk where they need more entries than are staged) and leaves only the one tile that
    # straddles the two loci (its bucket span does not fit the staged directories) to the generic kernel
    if pipeline in ("tile", "slab"):
        assert cnt[0] <= 256, cnt
    else:
        assert 5000 <= cnt[0] <= 6000 + 256, cnt


@pytest.mark.parametrize("ss_dis", [0, 2, 7])
@pytest.mark.parametrize("level", [1, 3, 5])
@pytest.mark.parametrize("n_iso,gapped", [(40, False), (62, False), (44, True)])
def test_locus_of_33_to_63_isoforms_stays_off_the_redo_list(oracle, level, n_iso, gapped, ss_dis, pipeline):
    """Windows of 33 .. 63 transcripts: the slab pipeline classifies their tiles with the 64-bit-mask kernel
    (l2r_wide.hip.h) instead of the redo list; results are exact on every pipeline.  `gapped`: unrelated transcripts of
    a chromosome the header does not have (tid -1: skipped, not a stop) sit between the isoforms in file order, so the
    window's members are not consecutive.  `ss_dis`: -d, the splice-site tolerance (src/update_gtf.c:717-779) -- the 64-bit-mask
    kernel probes within it like the 32-bit one (probe_near64)."""
    rng = np.random.default_rng(100 + n_iso)
    pool = [(50_000 + 500 * k, 50_000 + 500 * k + 120) for k in range(26)]
    txs = []
    for t in range(n_iso):
        keep = sorted(set([0, 25] + list(rng.choice(np.arange(1, 25), size=int(rng.integers(5, 16)), replace=False))))
        ex = [pool[k] for k in keep]
        if t % 4 == 0:
            ex[1] = (ex[1][0] - int(rng.integers(1, 25)), ex[1][1])
        if t % 6 == 0:
            ex[-2] = (ex[-2][0], ex[-2][1] + int(rng.integers(1, 25)))
        txs.append((0, t & 1, ex))
        if gapped and t % 3 == 0:
            txs.append((-1, 0, [(1_000 + 50 * t, 1_020 + 50 * t), (9_000, 9_100)]))      # (a larger tid would END the sweep, :799)
    if n_iso >= 60:                                   # one single-exon member as well
        txs.append((0, 0, [(52_000, 56_000)]))
    af = _anno(txs)
    rows = []
    iso = [t for t in txs if t[0] == 0]
    for i in range(5000):
        t = iso[int(rng.integers(len(iso)))][2]
        if len(t) < 3:
            ex = [list(t[0])]
            ex[0][0] += int(rng.integers(0, 300)); ex[0][1] -= int(rng.integers(0, 300))
        else:
            a = int(rng.integers(0, len(t) - 2))
            ex = [list(x) for x in t[a:a + int(rng.integers(2, 9))]]
            if i % 3 == 0:
                ex[-1][1] -= int(rng.integers(0, 40))
            if i % 5 == 0:
                ex[0][0] += int(rng.integers(0, 40))
            if i % 11 == 0 and len(ex) > 2:
                del ex[1]
        p, ops = _chain([tuple(x) for x in ex])
        rows.append((0, p, i & 1, ops))
    cnt = [0, 0, 0, 0, 0]
    got, want = _run(oracle, af, _reads(_sorted_rows(rows)), counters=cnt, full_level=level, ss_dis=ss_dis)
    assert ((want.info & 1) != 0).sum() > 100 and ((want.info & 2) != 0).sum() > 1000
    if pipeline in ("tile", "slab"):
        assert cnt[4] >= 15 and cnt[0] <= 300, cnt          # the locus's tiles took the 64-member kernel, (almost) nothing the redo list


def test_duplicated_transcripts_and_overlapping_exons(oracle):
    # the same transcript three times in a row; transcripts whose exons repeat or overlap each other (the reference
    # sorts a transcript's exons by (start, end) and keeps all of them, src/gtf.c:468-521); a transcript that is one
    # exon of another
    a = [(1_000, 1_200), (2_000, 2_150), (3_000, 3_300)]
    txs = [(0, 0, a), (0, 0, a), (0, 0, a),
           (0, 0, [(1_000, 1_200), (1_000, 1_200), (2_000, 2_150)]),                  # duplicated exon
           (0, 0, [(1_000, 1_200), (1_100, 1_400), (2_000, 2_150), (2_100, 2_300)]),  # overlapping exons
           (0, 1, [(2_000, 2_150)]),                                                  # an exon of the others, other strand
           (0, 0, [(5_000, 5_100), (5_050, 5_400), (5_300, 5_350), (6_000, 6_100)])]
    af = _anno(txs)
    chains = [a, a[:2], a[1:], [(1_100, 1_400), (2_000, 2_150)], [(1_000, 1_200), (2_100, 2_300)], [(1_050, 1_200), (2_000, 2_100)],
              [(2_000, 2_150)], [(2_010, 2_140)], [(5_000, 5_100), (6_000, 6_100)], [(5_050, 5_400), (6_000, 6_100)],
              [(5_300, 5_350), (6_000, 6_100)], [(5_000, 5_400), (6_000, 6_100)], [(1_000, 1_400), (2_000, 2_300)]]
    rows = []
    for k, ex in enumerate(chains * 20):
        p, ops = _chain(ex)
        rows.append((0, p, k & 1, ops))
    for level in (1, 2, 3, 4, 5):
        got, want = _run(oracle, af, _reads(_sorted_rows(rows)), full_level=level)
    assert ((want.info & 1) != 0).any() and ((want.info & 1) == 0).any()
    assert len(set(want.ref_tx.tolist())) >= 4


@pytest.mark.parametrize("level", [1, 2, 3, 4, 5])
def test_one_base_contacts_at_every_level(oracle, level):
    # reads whose first / last exon meets a transcript's exon by exactly one base, misses it by one, or overlaps by
    # two; and reads that touch the transcript's span by one base (Q5) -- check_full()'s comparisons are all <= / < on
    # these (src/update_gtf.c:629-681), one level after the other
    tx = [(10_000, 10_200), (11_000, 11_100), (12_000, 12_300), (13_000, 13_050)]
    af = _anno([(0, 0, tx), (0, 1, [(20_000, 20_100)]), (0, 0, [tx[0], tx[2], tx[3]])])
    rows = []
    for d in (-2, -1, 0, 1, 2):
        # left end around the first exon's END, right end around the last exon's START
        rows.append((0, *_chain([(10_200 + d, 10_260 + max(d, 0)), (11_000, 11_100)])))
        rows.append((0, *_chain([(11_000, 11_100), (12_000, 12_300), (12_940 + d, 13_000 + d)])))
        # ends around exon boundaries of INNER exons
        rows.append((0, *_chain([(11_100 + d, 11_180 + max(d, 0)), (12_000, 12_300)])))
        rows.append((0, *_chain([(10_000, 10_200), (10_940 + d, 11_000 + d)])))
        # single-exon reads around the single-exon transcript
        rows.append((0, *_chain([(19_900, 20_000 + d)])))
        rows.append((0, *_chain([(20_100 + d, 20_250)])))
        # exact chain whose ends move by d
        rows.append((0, *_chain([(10_000 + d, 10_200), (11_000, 11_100), (12_000, 12_300), (13_000, 13_050 + d)])))
    rows = [(r[0], r[1], 0, r[2]) for r in rows]
    got, want = _run(oracle, af, _reads(_sorted_rows(rows)), full_level=level)
    assert ((want.info & 4) != 0).any() and (level == 5 or ((want.info & 4) == 0).any())       # (-l 5: every read counts as full length)


@pytest.mark.parametrize("split", [0, 1])
def test_splice_distance_with_junction_table_on_ont_like_reads(oracle, split):
    # -d 2 moves every site comparison to a +-2 neighbourhood, the junction table decides acceptance (and with -s the
    # split), the records are ONT-like: ~50 CIGAR operations per read, micro-exons, XS tags that disagree with the flag
    anno, af, reads = util.make_case(23, n_reads=12000, n_exons=6, anno_exons=12000, ont=True, micro=3, xs=0.03)
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3, ss_dis=2))
    j, sj = util.junction_table(af, reads, base, 23, cover=0.6)
    cnt = [0, 0, 0, 0]
    got, want = _run(oracle, af, reads, sj=sj, counters=cnt, full_level=3, ss_dis=2, split_trans=split, min_sj_cnt=2)
    assert ((want.info & 32) != 0).sum() > 100 and ((want.info & 64) != 0).sum() > 10 and ((want.ex_flag & 16) != 0).sum() > 10
    # the mask kernels take -d 2 themselves (probe_near): nothing of this input is the generic kernel's
    assert cnt[0] == 0, cnt


def test_redo_share_of_the_benchmark_workload_shape(oracle, pipeline):
    # the shape bench.py measures (BASELINE configs[2], smaller): (almost) nothing may leave the fast path
    anno, af, reads = util.make_case(29, n_reads=200000, n_exons=8, anno_exons=150000)
    cnt = [0, 0, 0, 0]
    got, want = _run(oracle, af, reads, counters=cnt, full_level=3)
    assert cnt[0] <= reads.n // 200, cnt


def test_exons_of_64_kb_and_more(oracle, pipeline):
    """The slab rows keep an exon as {start, 16-bit length}: a read with an exon of 65 536 bases or more is stored densely
    instead (found out by the walk itself) and classified by the generic kernel; lengths 65 534 .. 65 537 sit on both sides."""
    txs = [(0, 0, [(1_000, 1_200), (2_000, 70_000), (80_000, 80_300)]),
           (0, 1, [(100_000, 100_100), (100_500, 100_500 + 65_535), (200_000, 200_100)]),
           (1, 0, [(5_000, 5_100), (6_000, 6_200)])]
    af = _anno(txs)
    rows = []
    for L in (65_534, 65_535, 65_536, 65_537, 90_000):
        rows.append((0, *_chain([(1_000, 1_200), (2_000, 2_000 + L - 1), (2_000 + L + 9_999, 2_000 + L + 10_299)])))
        rows.append((0, *_chain([(100_000, 100_100), (100_500, 100_500 + L - 1), (200_000, 200_100)])))
        rows.append((0, *_chain([(3_000, 3_000 + L - 1)])))                    # single long exon
        rows.append((0, *_chain([(500, 600), (3_000, 3_000 + L - 1)])))        # long LAST exon
    for k in range(300):                                                        # ordinary neighbours in the same tiles
        rows.append((0, *_chain([(1_000 + k % 150, 1_200), (2_000, 2_300)])))
        rows.append((1, *_chain([(5_000, 5_100), (6_000, 6_200 - (k % 7))])))
    rows = [(r[0], r[1], 0, r[2]) for r in rows]
    cnt = [0, 0, 0, 0, 0]
    got, want = _run(oracle, af, _reads(_sorted_rows(rows)), counters=cnt, full_level=3)
    lens = (want.ex_end - want.ex_start + 1)
    assert (lens >= 65_536).sum() >= 10 and (lens == 65_535).sum() >= 3
    if pipeline in ("tile", "slab"):
        assert 12 <= cnt[0] <= 80, cnt            # the reads with an exon of 65 536 bases or more (+ the tile a 200 kb span pushes off the fast path)


def test_more_exons_than_staged_positions_with_empty_inner_exons(oracle, pipeline):
    """`-e 0` keeps empty inner exons, so a read can have one exon per CIGAR op + 1 -- more than the bound the tiles were cut by
    ((ops + 3) / 2): a tile of such reads has more exons than the probe kernel stages (l2r_slab.hip.h SLAB_POS_CAP).  The reads
    behind the cap are written directly and classified by the generic kernel; everything is still the oracle's, bit for bit."""
    txs = [(0, 0, [(1_000, 1_100), (1_200, 1_300), (1_400, 1_500), (1_600, 1_700)])]
    af = _anno(txs)
    M_, N_ = 0, 3
    rows = []
    for k in range(700):
        # 10M then ten times (5N back to back 5N: an EMPTY exon between them) ... : 21 ops, 12 exons with -e 0, 2 + with -e 1
        ops = [(10, M_)]
        for _ in range(10):
            ops += [(5, N_), (5, N_)]
        ops = ops[:-1] + [(20, M_)]
        rows.append((0, 900 + (k % 3), k & 1, ops))
    for k in range(300):
        rows.append((0, *_chain([(1_000 + k % 50, 1_100), (1_200, 1_300)])[0:1], 0, _chain([(1_000 + k % 50, 1_100), (1_200, 1_300)])[1]))
    reads = _reads(_sorted_rows(rows))
    cnt = [0, 0, 0, 0]
    got, want = _run(oracle, af, reads, counters=cnt, full_level=3, min_exon=0)
    assert int(np.diff(want.ex_off).max()) >= 12
    if pipeline in ("tile", "slab"):
        assert cnt[0] > 0, cnt                      # some reads did not fit the staged positions
    _run(oracle, af, reads, full_level=3, min_exon=1)


def test_reads_far_from_their_tiles_first_read_and_an_outlier_between_neighbours(oracle, pipeline):
    """A slab row keeps an exon's start relative to the tile's first base in 18 bits, so the upload ends a tile where the reads would
    begin 2^17 bases apart: sparse input (here 5 reads every 400 kb) becomes small tiles on the mask path, not tiles for the generic
    kernel.  A densely stored read (an exon of 16 kb or more) that sits BETWEEN staged neighbours is written directly -- the
    write-out has to leave its positions alone."""
    txs = []
    for g in range(60):
        base = 10_000 + g * 400_000
        txs.append((0, g & 1, [(base, base + 100), (base + 300, base + 400), (base + 900, base + 1_000)]))
    af = _anno(txs)
    rows = []
    for g in range(60):
        base = 10_000 + g * 400_000
        for k in range(5):                                                       # 300 reads over 24 Mb: ~1.2 tiles
            rows.append((0, *_chain([(base + k, base + 100), (base + 300, base + 400), (base + 900, base + 1_000 - k)])))
    # a dense locus with a long-exon read in the middle of its tile
    for k in range(400):
        rows.append((1, *_chain([(5_000 + k % 40, 5_100), (5_300, 5_400)])))
        if k == 200:
            rows.append((1, *_chain([(5_020, 5_100), (5_300, 5_300 + 70_000)])))
    rows = [(r[0], r[1], 0, r[2]) for r in rows]
    txs.append((1, 0, [(5_000, 5_100), (5_300, 5_400)]))
    af = _anno(txs)
    cnt = [0, 0, 0, 0]
    got, want = _run(oracle, af, _reads(_sorted_rows(rows)), counters=cnt, full_level=3)
    assert ((want.info & 2) != 0).sum() > 100            # (chains that begin with a transcript's first exon are never "known": Q1)
    if pipeline in ("tile", "slab"):
        assert cnt[0] <= 2 and cnt[3] >= 60, cnt         # only the long-exon read is the generic kernel's; one tile per sparse locus


@pytest.mark.parametrize("level", [3, 5])
def test_row_word_limits_of_the_slab(oracle, level, pipeline):
    """A slab row is one word: the exon's start relative to the tile's first base in 18 bits, its length in 14.  Around every limit of
    that format, next to staged neighbours: exons of 16383 / 16384 / 16385 bases (the last two leave the slab), reads that begin just
    below / at / above 2^17 bases behind the tile's first read (the upload begins a new tile at 2^17), reads whose last exon starts
    at 2^18 - 2 / 2^18 - 1 / 2^18 bases behind the tile's first base (the last two leave the slab)."""
    lo = 1_000_000
    txs = [(0, 0, [(lo, lo + 100), (lo + 300, lo + 400), (lo + 900, lo + 1_000)]),
           (0, 1, [(lo + 200, lo + 260), (lo + 20_000, lo + 20_100), (lo + 262_100, lo + 262_400)]),
           (0, 0, [(lo + 131_000, lo + 131_200), (lo + 131_400, lo + 131_500), (lo + 140_000, lo + 140_100)])]
    rows = []
    for k in range(150):                                                         # the tile's staged majority, first read at lo
        rows.append((0, *_chain([(lo + (k % 50), lo + 100), (lo + 300, lo + 400), (lo + 900, lo + 1_000 - (k % 7))])))
    for ln in (16_382, 16_383, 16_384, 16_385, 40_000):                          # exon lengths around 2^14 - 1
        rows.append((0, *_chain([(lo + 10, lo + 100), (lo + 300, lo + 300 + ln - 1)])))
        rows.append((0, *_chain([(lo + 12, lo + 12 + ln - 1), (lo + 12 + ln + 200, lo + 12 + ln + 300)])))
    for d in (131_070, 131_071, 131_072, 131_073, 131_400):                      # read starts around 2^17 behind the tile's first base
        rows.append((0, *_chain([(lo + d, lo + d + 90), (lo + d + 300, lo + d + 380)])))
        rows.append((0, *_chain([(lo + 131_000, lo + 131_200), (lo + 131_400, lo + 131_500), (lo + 140_000, lo + 140_100)])))
    for d in (262_141, 262_142, 262_143, 262_144, 262_145):                      # last exon starts around 2^18 - 1 behind the TILE's base
        rows.append((0, *_chain([(lo + 200, lo + 260), (lo + 20_000, lo + 20_100), (lo + d, lo + d + 250)])))
    for d in (262_141, 262_142, 262_143, 262_144, 262_145):                      # ... and behind the READ's own base (a far read)
        s = lo + 140_000
        rows.append((0, *_chain([(s, s + 100), (s + 500, s + 600), (s + d, s + d + 80)])))
    rows = [(r[0], r[1], i & 1, r[2]) for i, r in enumerate(rows)]
    af = _anno(txs)
    cnt = [0, 0, 0, 0]
    got, want = _run(oracle, af, _reads(_sorted_rows(rows)), counters=cnt, full_level=level)
    assert ((want.info & 2) != 0).sum() > 100 and len(rows) <= 256


@pytest.mark.parametrize("ss_dis", [0, 2, 7])
@pytest.mark.parametrize("level", [1, 3, 5])
@pytest.mark.parametrize("n_iso", [70, 150])
def test_locus_beyond_the_mask_width_is_taken_in_chunks(oracle, level, n_iso, ss_dis, pipeline):
    """More overlapping transcripts than a 64-bit window holds, and exons shared by transcripts that lie more than 64 apart in the
    annotation (dictionary keys in several entries): the slab pipeline takes the tile's window 63 members at a time
    (k_probe_slab_chunked) and leaves nothing of it to the generic kernel; results are exact on every pipeline.  The known break,
    the LAST transcript with a shared site and the sticky full-length evidence all have to survive the chunk boundaries: the reads
    copy transcripts from every part of the file order."""
    rng = np.random.default_rng(500 + n_iso)
    pool = [(30_000 + 700 * k, 30_000 + 700 * k + 150) for k in range(12)]
    txs = []
    for t in range(n_iso):
        keep = sorted(set([0, 11] + list(rng.choice(np.arange(1, 11), size=int(rng.integers(3, 9)), replace=False))))
        ex = [pool[k] for k in keep]
        if t % 7 == 3:
            ex[0] = (ex[0][0] + int(rng.integers(1, 40)), ex[0][1])                # another first base: the first exon matches nothing else
        if t % 9 == 4:
            ex[-1] = (ex[-1][0], ex[-1][1] + int(rng.integers(1, 60)))
        txs.append((0, t & 1, ex))
    txs.append((0, 0, [(31_000, 36_500)]))                                        # single-exon members, early and late in file order
    txs.append((0, 1, [(31_200, 37_000)]))
    af = _anno(txs)
    rows = []
    for i in range(4000):
        t = txs[int(rng.integers(n_iso))][2]
        mode = i % 6
        if mode == 5:
            ex = [[31_000 + int(rng.integers(0, 400)), 36_000 + int(rng.integers(0, 900))]]      # one exon over the single-exon members
        else:
            a = 0 if mode < 2 else int(rng.integers(0, len(t) - 1))
            ex = [list(x) for x in (t if mode == 0 else t[a:a + int(rng.integers(2, 8))])]
            if mode == 3:
                ex[-1][1] -= int(rng.integers(0, 40))
            if mode == 4 and len(ex) > 2:
                del ex[1]
        p, ops = _chain([tuple(x) for x in ex])
        rows.append((0, p, i & 1, ops))
    cnt = [0, 0, 0, 0, 0]
    got, want = _run(oracle, af, _reads(_sorted_rows(rows)), counters=cnt, full_level=level, ss_dis=ss_dis)
    assert ((want.info & 1) != 0).sum() > 100 and ((want.info & 2) != 0).sum() > 1000 and len(np.unique(want.ref_tx)) > 15
    assert cnt[1] > 0                                                             # keys in several entries exist
    if pipeline in ("tile", "slab"):
        assert cnt[0] == 0, cnt                                                   # nothing left to the generic kernel


def test_reads_of_300_and_9000_exons_between_ordinary_neighbours(oracle, pipeline):
    """The walk hands a read's exon count and its place among its tile's exons to the probe kernels in ONE word (8 + 13 bits,
    l2r_slab.hip.h SlabArgs::pl); a tile with a read of 256 exons or more, or with 8192 exons or more, is "fat" and uses the two
    whole words instead.  Only outliers (reads beyond 24 slab rows) make such tiles: a 300-exon read and a 9000-exon read sit in
    the middle of tiles of ordinary reads here, whose results must not move."""
    txs = [(0, 0, [(1_000, 1_100), (1_300, 1_400), (1_600, 1_700)]), (1, 0, [(5_000, 5_100), (5_300, 5_400)])]
    af = _anno(txs)
    rows = []
    for k in range(600):
        rows.append((0, *_chain([(1_000 + k % 60, 1_100), (1_300, 1_400), (1_600, 1_700 - k % 5)])))
        rows.append((1, *_chain([(5_000 + k % 30, 5_100), (5_300, 5_400)])))
    rows.append((0, *_chain([(1_020 + 40 * j, 1_030 + 40 * j) for j in range(300)])))          # 300 exons: its count does not fit 8 bits
    rows.append((1, *_chain([(5_010 + 30 * j, 5_020 + 30 * j) for j in range(9_000)])))        # 9000 exons: the places behind it do not fit 13 bits
    rows = [(r[0], r[1], 0, r[2]) for r in rows]
    cnt = [0, 0, 0, 0]
    got, want = _run(oracle, af, _reads(_sorted_rows(rows)), counters=cnt, full_level=3)
    n_ex = np.diff(want.ex_off)
    assert int((n_ex == 300).sum()) == 1 and int((n_ex == 9_000).sum()) == 1
    assert ((want.info & 2) != 0).sum() > 500
    if pipeline == "slab":
        assert 2 <= cnt[0] <= 300, cnt                # the two long reads (and the reads of their tiles that no longer fit the staged positions)
    if pipeline == "tile":
        assert 1 <= cnt[0] <= 300, cnt                # k_tile has no rows to outgrow: the 300-exon read is classified on the mask path; the 9000-exon read's tile keeps the slab form


@pytest.mark.parametrize("min_exon", [0, 1, 3])
def test_long_cigars_with_cut_ops_crowded_into_a_few_words(oracle, min_exon, pipeline):
    """The wave-cooperative walk of long CIGARs (l2r_kernels.hip.h wave_chunk_try) keeps a lane's FIRST and LAST cut op of its six or
    eight CIGAR words; three cut ops in one lane's words re-walk the round two words per lane.  Reads with bursts of micro-exons
    (N M N M N ... next to each other, with and without cutting deletions between them) at every place of the op stream -- inside
    a lane, across lanes, across the 384- and 512-op rounds -- between stretches of M / I / D noise, of 40 to 1500 ops: one round
    of six words, one of eight, several of eight, and the two-words-per-lane rounds in each of them."""
    M_, I_, D_, N3 = 0, 1, 2, 3
    rng = np.random.default_rng(77 + min_exon)
    txs = [(0, 0, [(2_000 + 700 * k, 2_300 + 700 * k) for k in range(12)]), (0, 1, [(2_050, 2_300), (2_700, 3_000), (4_100, 4_400)])]
    af = _anno(txs)
    rows = []
    for i in range(700):
        target = int(rng.choice([40, 200, 370, 384, 390, 500, 512, 520, 700, 1_030, 1_500]))
        ops = []
        n_cuts = 0
        burst_at = sorted(int(x) for x in rng.integers(0, max(target - 12, 1), size=int(rng.integers(1, 4))))
        while len(ops) < target:
            if burst_at and len(ops) >= burst_at[0] and n_cuts < 18:
                burst_at.pop(0)
                for _ in range(int(rng.integers(2, 5))):            # 2 .. 4 cut ops with 1 .. 6-base exons between them
                    cut = (int(rng.integers(51, 90)), D_) if rng.random() < 0.3 else (int(rng.integers(3, 400)), N3)
                    ops.append(cut)
                    ops.append((int(rng.integers(1, 7)), M_))
                    n_cuts += 1
                continue
            r = rng.random()
            if r < 0.55 or not ops or ops[-1][1] != M_:
                ops.append((int(rng.integers(1, 40)), M_))
            elif r < 0.75:
                ops.append((int(rng.integers(1, 4)), I_))
            elif r < 0.97:
                ops.append((int(rng.integers(1, 51)), D_))            # (up to max_delet: no cut)
            elif n_cuts < 18:
                ops.append((int(rng.integers(3, 2_000)), N3))
                n_cuts += 1
        if ops[-1][1] != M_:
            ops.append((int(rng.integers(5, 40)), M_))
        rows.append((0, 1_000 + int(rng.integers(0, 3_000)), i & 1, ops))
    cnt = [0, 0, 0, 0]
    got, want = _run(oracle, af, _reads(_sorted_rows(rows)), counters=cnt, full_level=3, min_exon=min_exon)
    n_ex = np.diff(want.ex_off)
    assert n_ex.max() >= 8 and (n_ex >= 4).sum() > 300


@pytest.mark.parametrize("split", [0, 1])
def test_junction_check_in_blocks_of_more_exons_than_the_mapped_positions(oracle, split, pipeline):
    """k_validate_sj maps the exon positions of a block of 256 reads to their reads (6144 positions, l2r_kernels.hip.h SJ_MAP_CAP) and
    spreads the junction lookups over the positions; a block of full-length reads of a 30 .. 40-exon gene has more exons than that,
    and the read that straddles position 6144 is not mapped: its positions have no owner (they were looked at all the
    same, with whatever an earlier workgroup had left in LDS).  Reads with skipped exons (novel junctions) sit all over such blocks,
    the straddling ones included; the junction table supports some of the novel junctions and not others."""
    rng = np.random.default_rng(77)
    gene = [(10_000 + 400 * k, 10_000 + 400 * k + 150) for k in range(40)]
    txs = [(0, 0, gene), (0, 1, gene[:12] + [(gene[12][0], gene[12][1] + 40)]), (1, 0, [(5_000, 5_100), (5_300, 5_400)])]
    af = _anno(txs)
    rows = []
    for i in range(900):
        n = int(rng.integers(30, 41))
        ex = [list(x) for x in gene[:n]]
        ex[0][0] += int(rng.integers(0, 100))                       # sorted input with varied starts
        if i % 3 == 0:
            del ex[int(rng.integers(2, n - 2))]                     # a skipped exon: one novel junction
        if i % 7 == 0:
            del ex[int(rng.integers(2, len(ex) - 2))]
        if i % 5 == 0:
            ex[-1][1] -= int(rng.integers(0, 30))
        rows.append((0, *_chain([tuple(x) for x in ex])))
    rows = [(r[0], r[1], i & 1, r[2]) for i, r in enumerate(rows)]
    reads = _reads(_sorted_rows(rows))
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=5))          # (-l 5: every read counts as full length)
    assert int(np.diff(base.ex_off).reshape(-1)[:256].sum()) > 6144 + 40          # the first block alone is beyond the mapped positions
    j, sj = util.junction_table(af, reads, base, 77, cover=0.5)
    for _ in range(3):                                              # (stale LDS: what the positions hold differs from launch to launch)
        got, want = _run(oracle, af, reads, sj=sj, full_level=5, split_trans=split, min_sj_cnt=1)
    assert ((want.info & 32) != 0).sum() > 200 and ((want.info & 16) != 0).sum() > 20 and ((want.info & 64) != 0).sum() > 20


@pytest.mark.parametrize("route", ["wide_instance", "slab_form", "inexact"])
def test_wide_tiles_on_every_route_of_the_tile_path(oracle, monkeypatch, route):
    """A locus of 40 isoforms on the one-kernel tile path: its tiles (windows of 33 .. 63 transcripts) are classified by k_tile's WIDE
    instance straight from their CIGARs (default), by k_probe_slab_wide from the slab form k_tile gives them (L2R_WIDE_DIRECT=0), and --
    with a threshold that is borderline inside the tiles (-i 450 against introns of 380 .. 24 500 bases: the tiles are not exact) -- by the
    slab form again although the WIDE instance is on.  Exact on each."""
    monkeypatch.setenv("L2R_PIPELINE", "tile")
    if route == "slab_form":
        monkeypatch.setenv("L2R_WIDE_DIRECT", "0")
    rng = np.random.default_rng(4040)
    pool = [(50_000 + 500 * k, 50_000 + 500 * k + 120) for k in range(26)]
    txs = []
    for t in range(40):
        keep = sorted(set([0, 25] + list(rng.choice(np.arange(1, 25), size=int(rng.integers(5, 16)), replace=False))))
        txs.append((0, t & 1, [pool[k] for k in keep]))
    af = _anno(txs)
    rows = []
    for i in range(6000):
        t = txs[int(rng.integers(len(txs)))][2]
        a = int(rng.integers(0, len(t) - 2))
        ex = [list(x) for x in t[a:a + int(rng.integers(2, 9))]]
        if i % 3 == 0:
            ex[-1][1] -= int(rng.integers(0, 40))
        if i % 7 == 0 and len(ex) > 2:
            del ex[1]
        p, ops = _chain([tuple(x) for x in ex])
        rows.append((0, p, i & 1, ops))
    cnt = [0, 0, 0, 0, 0]
    kw = dict(full_level=3)
    if route == "inexact":
        kw["min_intron"] = 450
    got, want = _run(oracle, af, _reads(_sorted_rows(rows)), counters=cnt, **kw)
    assert ((want.info & 2) != 0).sum() > 1000
    assert cnt[4] >= 15 and cnt[0] <= 300, cnt              # the locus's tiles are 64-bit-mask tiles, (almost) nothing on the redo list


@pytest.mark.parametrize("shape", ["far_members", "chromosome_edge", "end_of_annotation", "many_chunks"])
def test_chunk_lists_at_their_limits(oracle, shape, pipeline):
    """k_tile_chunk (l2r_tchunk.hip.h) takes a window in stretches of 63 CONSECUTIVE transcripts of the annotation's file order:
    far_members        a locus whose 150 isoforms are written in two halves with 20 000 transcripts of LOWER coordinates between them (file
                       order is arbitrary, src/update_gtf.c:796-822 walks it as it is): more stretches than the window scan looks at -- the
                       tile's reads are the generic kernel's, results exact;
    chromosome_edge    the isoform-rich locus is the last of its chromosome, the stretch that ends its window runs into the next chromosome's
                       transcripts (behind every read) and a quiet locus there must stay untouched;
    end_of_annotation  ... and the last of the annotation: the last stretch is cut short by the annotation's end;
    many_chunks        400 isoforms around one another: seven stretches, the sweep's state carried through all of them."""
    rng = np.random.default_rng(17)
    pool = [(50_000 + 700 * k, 50_000 + 700 * k + 160) for k in range(24)]

    def isoform(t):
        keep = sorted(set([0, 23] + list(rng.choice(np.arange(1, 23), size=int(rng.integers(5, 16)), replace=False))))
        return (0, t & 1, [pool[k] for k in keep])
    n_iso = 400 if shape == "many_chunks" else 150
    iso = [isoform(t) for t in range(n_iso)]
    # (transcripts in front of the locus: all of them end below 50 000)
    if shape == "far_members":
        quiet0 = [(0, 0, [(1_000 + 2 * g, 1_000 + 2 * g + 1)]) for g in range(20_000)]
    else:
        quiet0 = [(0, 0, [(1_000 + 40 * g, 1_000 + 40 * g + 10), (1_000 + 40 * g + 20, 1_000 + 40 * g + 30)]) for g in range(50)]
    quiet1 = [(1, 0, [(5_000 + 3_000 * g, 5_000 + 3_000 * g + 100), (5_000 + 3_000 * g + 500, 5_000 + 3_000 * g + 650)]) for g in range(100)]
    if shape == "far_members":
        txs = iso[:75] + quiet0 + iso[75:] + quiet1
    elif shape == "end_of_annotation":
        txs = quiet0 + iso                                  # (nothing behind the locus)
    else:
        txs = quiet0 + iso + quiet1
    af = _anno(txs)
    rows = []
    for i in range(5000):
        t = iso[int(rng.integers(n_iso))][2]
        a = int(rng.integers(0, len(t) - 2))
        ex = [list(x) for x in t[a:a + int(rng.integers(2, 9))]]
        if i % 4 == 0:
            ex[-1][1] -= int(rng.integers(0, 60))
        if i % 7 == 0:
            ex[0][0] += int(rng.integers(0, 60))
        p, ops = _chain([tuple(x) for x in ex])
        rows.append((0, p, i & 1, ops))
    if shape not in ("end_of_annotation",):
        for i in range(2000):
            t = quiet1[int(rng.integers(len(quiet1)))][2]
            p, ops = _chain(t)
            rows.append((1, p, 0, ops))
    cnt = [0, 0, 0, 0]
    for level in (3, 5):
        got, want = _run(oracle, af, _reads(_sorted_rows(rows)), counters=cnt, full_level=level)
        assert ((want.info & 2) != 0).sum() > 500
        if pipeline == "tile" and shape != "far_members":
            assert cnt[0] <= 256, cnt                       # (the isoform-rich tiles stay off the redo list)
        if pipeline == "tile" and shape == "far_members":
            assert cnt[0] >= 4000, cnt                      # (the window scan gives up: the locus' reads are the generic kernel's)


@pytest.mark.parametrize("variants", [1, 24, 25, 150])
def test_terminal_exon_groups_of_a_chunk(oracle, variants, pipeline):
    """k_tile_chunk takes the full-length evidence of levels 1-4 (src/update_gtf.c:803-822: the read's terminal exons against the
    member's) once per DISTINCT first / last exon of a chunk's 63 transcripts (TcGroup, l2r_tchunk.hip.h), member by member when a
    chunk has more than 24 of a kind.  150 isoforms around one another whose first exon ends and whose last exon begins in `variants`
    different places: one group of each kind; exactly 24 (the cap); 25 (one more: member by member); every isoform its own.  Reads
    that begin / end with an isoform's terminal exon (full length at level 1 only with the very same inner boundary), with a
    shortened one, and inside the isoform."""
    rng = np.random.default_rng(23 + variants)
    pool = [(50_000 + 700 * k, 50_000 + 700 * k + 160) for k in range(1, 23)]

    def isoform(t):
        v = t % variants
        keep = sorted(rng.choice(np.arange(len(pool)), size=int(rng.integers(5, 14)), replace=False))
        return (0, t & 1, [(49_000, 49_200 + 3 * v)] + [pool[k] for k in keep] + [(70_000 + 3 * ((v * 7) % variants), 70_900)])
    iso = [isoform(t) for t in range(150)]
    quiet0 = [(0, 0, [(1_000 + 40 * g, 1_000 + 40 * g + 10), (1_000 + 40 * g + 20, 1_000 + 40 * g + 30)]) for g in range(50)]
    quiet1 = [(1, 0, [(5_000 + 3_000 * g, 5_000 + 3_000 * g + 100), (5_000 + 3_000 * g + 500, 5_000 + 3_000 * g + 650)]) for g in range(20)]
    af = _anno(quiet0 + iso + quiet1)
    rows = []
    for i in range(6000):
        t = iso[int(rng.integers(len(iso)))][2]
        kind = i % 5
        if kind == 0:
            ex = [list(x) for x in t]                                        # the whole isoform
        elif kind == 1:
            ex = [list(x) for x in t[:int(rng.integers(2, len(t)))]]         # from its first exon on
        elif kind == 2:
            ex = [list(x) for x in t[int(rng.integers(0, len(t) - 2)):]]     # up to its last exon
        else:
            a = int(rng.integers(0, len(t) - 2))
            ex = [list(x) for x in t[a:a + int(rng.integers(2, 9))]]
        if i % 3 == 0:
            ex[0][0] += int(rng.integers(0, 150))                            # (the outer ends move, the inner boundaries stay)
        if i % 4 == 0:
            ex[-1][1] -= int(rng.integers(0, 150))
        if i % 11 == 0 and len(ex) > 2:
            ex[0][1] += 3                                                    # (another variant's first exon, or nobody's)
        p, ops = _chain([tuple(x) for x in ex])
        rows.append((0, p, i & 1, ops))
    cnt = [0, 0, 0, 0]
    for level in (1, 2, 3, 4):
        got, want = _run(oracle, af, _reads(_sorted_rows(rows)), counters=cnt, full_level=level)
        assert ((want.info & 4) != 0).sum() > 300 and ((want.info & 4) == 0).sum() > 300       # (I_FULL: full-length and not, both kinds of verdict)
        if pipeline == "tile" and variants <= 25:
            assert cnt[0] <= 256, cnt                       # (150 variants: more START entries than k_tile_chunk stages -- slab form, the old kernel, some reads generic)
