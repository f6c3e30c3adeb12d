"""GPU: hand-built corner cases of the fast classification path against the oracle -- bit exact.

Each case targets one precondition of k_classify_fast (lr2rmats_amd/csrc/l2r_kernels.hip.h): the bucket grid at
the end of a chromosome, transcript windows wider than the membership masks, tiles that straddle chromosomes,
many tiles per persistent workgroup, crowded dictionary buckets, reads far longer than the dictionary span."""
import os

import numpy as np
import pytest

from lr2rmats_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu

M, N_ = 0, 3


def _reads(rows, chrom_names=("chr1", "chr2", "chr3")):
    """rows: (tid, pos0, rev, [(len, op), ...]) sorted by (tid, pos)."""
    tid = np.array([r[0] for r in rows], np.int32)
    pos = np.array([r[1] for r in rows], np.int32)
    rev = np.array([r[2] for r in rows], np.uint8)
    off = np.zeros(len(rows) + 1, np.int64)
    np.cumsum([len(r[3]) for r in rows], out=off[1:])
    cig = np.array([(l << 4) | op for r in rows for (l, op) in r[3]], np.uint32)
    return synth.Reads(list(chrom_names), tid, pos, rev, rev.copy(), np.zeros(len(rows), np.uint8), off, cig)


def _anno(txs):
    """txs: (tid, rev, [(start, end), ...]) in file order."""
    tx_tid = np.array([t[0] for t in txs], np.int32)
    tx_rev = np.array([t[1] for t in txs], np.uint8)
    off = np.zeros(len(txs) + 1, np.int64)
    np.cumsum([len(t[2]) for t in txs], out=off[1:])
    es = np.array([e[0] for t in txs for e in t[2]], np.int32)
    ee = np.array([e[1] for t in txs for e in t[2]], np.int32)
    return synth.Annotation(["chr1", "chr2", "chr3"], tx_tid, tx_rev, np.arange(len(txs), dtype=np.int32), off, es, ee, len(txs), None)


def _chain(exons):
    """CIGAR of a read made of the given exons (1-based closed), and its 0-based position."""
    ops = []
    for k, (s, e) in enumerate(exons):
        if k:
            ops.append((s - exons[k - 1][1] - 1, N_))
        ops.append((e - s + 1, M))
    return exons[0][0] - 1, ops


@pytest.fixture(autouse=True, params=["tile", "slab", "classic"])
def pipeline(request, monkeypatch):
    """Every case runs on each of the engine's two kernel pipelines (l2r_engine.hip: L2R_PIPELINE is read by l2r_create;
    records the chosen pipeline cannot take -- unsorted, long CIGARs -- fall to the classic one by themselves)."""
    monkeypatch.setenv("L2R_PIPELINE", request.param)
    return request.param


def _run(oracle, af, reads, counters=None, sj=None, **kw):
    want = util.oracle_run(oracle, af, reads, oracle.default_params(**kw), sj)
    eng = capi.Engine(0)
    try:
        eng.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
        eng.set_junctions(sj)
        got = eng.classify(reads, capi.default_params(**kw))
        # the accepted list (chunks placed by the classification kernel and by k_gather_accepted, mixed) == the
        # accepted reads of the full result, in read order
        acc = eng.download_accepted()
        idx = np.nonzero((got.info & 128) != 0)[0]
        lens = (got.info[idx] >> 8).astype(np.int64)
        g = synth._ragged_gather_index(got.ex_off[idx], lens)
        np.testing.assert_array_equal(acc.read_index, idx)
        np.testing.assert_array_equal(np.diff(acc.ex_off), lens)
        np.testing.assert_array_equal(acc.ex_start, got.ex_start[g])
        np.testing.assert_array_equal(acc.ex_end, got.ex_end[g])
        np.testing.assert_array_equal(acc.ex_flag, got.ex_flag[g])
        if counters is not None:            # [reads the generic kernel took, wide entries, compact transcripts, tiles]
            import ctypes as C
            lib = capi.load_library()
            cnt = (C.c_longlong * 13)()
            lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            lib.l2r_debug_counters(eng.ctx, cnt, 13)
            counters[:] = list(cnt)[:4] + [cnt[12]]      # (... , tiles the 64-member kernel took)
    finally:
        eng.close()
    util.assert_same_result(got, want, 0 if sj is None else len(sj[0]), kw.get("split_trans", 0))
    return got, want


@pytest.mark.parametrize("level", [3, 4])
def test_read_over_last_exon_of_chromosome(oracle, level):
    # the read's first exon overlaps only the LAST exon of the chromosome's last transcripts: "overlaps some exon"
    # must be seen although no probe key lies that far (the bucket grid has to cover exon ends)
    base = 83_291_818
    tx = [(base, base + 156), (base + 1899, base + 2007), (base + 29_888, base + 30_106)]
    af = _anno([(0, 0, tx), (0, 0, [tx[0], tx[2]]), (1, 0, [(1000, 1200), (5000, 5100)])])
    rows = []
    for shift in (0, 7, 50, 160, 217, 218, 219, 600):
        ex = [(tx[2][0] + 162 + shift, tx[2][0] + 318 + shift), (tx[2][0] + 2061 + shift, tx[2][0] + 2169 + shift)]
        p, ops = _chain(ex)
        rows.append((0, p, 0, ops))
    rows.sort(key=lambda r: (r[0], r[1]))
    got, want = _run(oracle, af, _reads(rows), full_level=level)
    assert ((want.info & 4) == 0).any() and ((want.info & 4) != 0).any()


def test_window_wider_than_masks_and_crowded_buckets(oracle):
    # 80 isoforms of one locus sharing exons (dictionary entries with more than 64 members, windows of more than
    # 32 transcripts) and 40 alternative ends inside one 512-bp bucket
    rng = np.random.default_rng(5)
    pool = [(10_000 + 700 * k, 10_000 + 700 * k + 150) for k in range(14)]
    txs = []
    for t in range(80):
        keep = sorted(set([0, 13] + list(rng.choice(np.arange(1, 13), size=8, replace=False))))
        ex = [pool[k] for k in keep]
        if t % 2:
            ex[-2] = (ex[-2][0], ex[-2][1] + int(t // 2) + 1)        # crowded bucket: many ends for one start
        txs.append((0, t & 1, ex))
    txs.sort(key=lambda t: (t[2][0][0], t[2][-1][1]))
    af = _anno(txs)
    rows = []
    for i in range(3000):
        t = txs[int(rng.integers(len(txs)))][2]
        a = int(rng.integers(0, len(t) - 2))
        ex = [list(x) for x in t[a:a + int(rng.integers(2, 7))]]
        if i % 3 == 0:
            ex[0][0] += int(rng.integers(0, 40))
        if i % 7 == 0:
            ex[-1][1] -= int(rng.integers(0, 40))
        p, ops = _chain([tuple(x) for x in ex])
        rows.append((0, p, int(i & 1), ops))
    rows.sort(key=lambda r: (r[0], r[1]))
    got, want = _run(oracle, af, _reads(rows), full_level=3)
    assert ((want.info & 1) != 0).sum() > 100 and ((want.info & 2) != 0).sum() > 100


def test_tiles_straddling_chromosomes_and_many_tiles_per_workgroup(oracle, monkeypatch):
    # small chromosomes: nearly every tile holds reads of two or three chromosomes; a grid of 2 workgroups walks
    # over all tiles (the persistent loop and its register prefetch)
    monkeypatch.setenv("L2R_FAST_GRID", "2")
    anno, af, reads = util.make_case(31, n_reads=9000, n_exons=5, anno_exons=3000, nchr=24)
    want = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
    eng = capi.Engine(0)
    try:
        eng.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
        got = eng.classify(reads, capi.default_params(full_level=3))
    finally:
        eng.close()
    util.assert_same_result(got, want, 0, 0)
    assert len(np.unique(reads.tid)) >= 20


def test_long_reads_and_long_cigars(oracle):
    # reads of 40..90 exons (more than a tile's LDS share when they cluster) and ONT-like CIGARs: the tile falls back
    # to smaller tiles / HBM exons / the generic kernel without changing a byte
    anno, af, reads = util.make_case(32, n_reads=3000, n_exons=60, anno_exons=40000, ont=True, micro=3, xs=0.02)
    _run(oracle, af, reads, full_level=3)
    _run(oracle, af, reads, full_level=1, min_exon=1)


def test_dictionary_slice_and_span_overflow(oracle):
    """Tiles whose dictionary slices do not fit the staged tables (more than 224 distinct exons in the tile's span) and
    reads that span more than the staged bucket directory (a 400-kb intron): everything falls back to the generic
    kernel, nothing changes in the results."""
    rng = np.random.default_rng(9)
    txs, pool = [], []
    for k in range(320):                                     # 320 distinct exons inside 64 kb
        s = 20_000 + 200 * k
        pool.append((s, s + 60 + int(rng.integers(0, 80))))
    for t in range(60):
        keep = sorted(rng.choice(np.arange(len(pool)), size=12, replace=False))
        txs.append((0, t & 1, [pool[k] for k in keep]))
    txs.append((0, 0, [(600_000, 600_200), (1_000_400, 1_000_600)]))          # one junction 400 kb long
    txs.sort(key=lambda t: (t[2][0][0], t[2][-1][1]))
    af = _anno(txs)
    rows = []
    for i in range(1500):
        t = txs[int(rng.integers(len(txs)))][2]
        a = int(rng.integers(0, max(1, len(t) - 2)))
        ex = [list(x) for x in t[a:a + int(rng.integers(2, 6))]]
        if i % 4 == 0:
            ex[0][0] += int(rng.integers(0, 30))
        p, ops = _chain([tuple(x) for x in ex])
        rows.append((0, p, int(i & 1), ops))
    rows.sort(key=lambda r: (r[0], r[1]))
    got, want = _run(oracle, af, _reads(rows), full_level=3)
    assert ((want.info & 1) != 0).sum() > 50


def test_reads_longer_than_the_lds_tile(oracle):
    """150-exon reads: even the smallest tile (32 reads) has more exons than the LDS image holds, so the walk writes
    its exons straight to HBM and the generic kernel classifies them."""
    n_ex, n_reads = 150, 200
    ex = [(50_000 + 400 * k, 50_000 + 400 * k + 120) for k in range(n_ex + 40)]
    af = _anno([(0, 0, ex[:n_ex]), (0, 1, ex[5:n_ex + 20:2]), (1, 0, [(100, 300), (900, 1200)])])
    rng = np.random.default_rng(11)
    rows = []
    for i in range(n_reads):
        a = int(rng.integers(0, 30))
        chain = [list(x) for x in ex[a:a + n_ex]]
        if i % 3 == 0:
            chain[0][0] += 7
        if i % 5 == 0:
            del chain[60]
        p, ops = _chain([tuple(x) for x in chain])
        rows.append((0, p, 0, ops))
    rows.sort(key=lambda r: (r[0], r[1]))
    got, want = _run(oracle, af, _reads(rows), full_level=3)
    assert int((got.info >> 8).max()) >= 149


@pytest.mark.parametrize("level", [1, 3, 5])
def test_windows_with_gaps_stay_on_the_fast_path(oracle, level):
    # Two chromosome-long transcripts at the head of the file pin the reference's cursor: every tile's window is
    # {the long ones} + {the local genes}, with hundreds of transcripts in between that end before the tile.  Genes
    # hold their transcripts in arbitrary order (GENCODE style), so sweeps end at different members per read.
    rng = np.random.default_rng(11)
    txs = [(0, 0, [(1_000, 1_200), (1_990_000, 1_990_300)]), (0, 1, [(1_100, 1_200), (5_000, 5_050), (1_995_000, 1_995_100)])]
    genes = []
    for g in range(300):
        base = 10_000 + 6_000 * g
        pool = [(base + 400 * k, base + 400 * k + 120) for k in range(10)]
        iso = []
        for t in range(4):
            keep = sorted(set([0] + list(rng.choice(np.arange(1, 10), size=5, replace=False))))
            iso.append((0, g & 1, [pool[k] for k in keep]))
        order = rng.permutation(4)
        genes.append([iso[i] for i in order])
        txs.extend(genes[-1])
    af = _anno(txs)
    rows = []
    for i in range(30000):                  # about 2.5 genes = 10 transcripts under a tile of 256 reads
        iso = genes[int(rng.integers(len(genes)))]
        t = iso[int(rng.integers(4))][2]
        a = int(rng.integers(0, len(t) - 1))
        ex = [list(x) for x in t[a:a + int(rng.integers(1, 5))]]
        if i % 4 == 0:
            ex[0][0] -= int(rng.integers(0, 30))
        if i % 9 == 0:
            ex[-1][1] += int(rng.integers(0, 30))
        p, ops = _chain([tuple(x) for x in ex])
        rows.append((0, p, int(i & 1), ops))
    rows.append((0, 999, 0, _chain([(1_000, 1_200), (1_990_000, 1_990_300)])[1]))        # the long transcript itself
    rows.sort(key=lambda r: (r[0], r[1]))
    cnt = [0, 0, 0, 0]
    got, want = _run(oracle, af, _reads(rows), counters=cnt, full_level=level)
    assert ((want.info & 1) != 0).sum() > 2000
    # the long read covers the whole chromosome: the tile that holds it may go to the generic kernel, no other
    if not os.environ.get("L2R_ABLATE"):
        assert cnt[0] <= 2 * 256, cnt


def test_gencode_style_annotation_keeps_the_fast_path(oracle):
    # transcripts in arbitrary order inside their gene and one chromosome-long transcript per chromosome that pins the
    # reference's cursor: every tile's window has gaps (the long transcript + the local genes).  Bit exact, and the
    # generic kernel sees (almost) nothing.
    anno, af, reads = util.make_case(41, n_reads=120000, n_exons=6, anno_exons=60000, shuffle=True, long_tx=1)
    cnt = [0, 0, 0, 0]
    for level in (3, 5):
        got, want = _run(oracle, af, reads, counters=cnt, full_level=level)
        assert ((want.info & 1) != 0).sum() > 10000 and ((want.info & 2) != 0).sum() > 10000
        if not os.environ.get("L2R_ABLATE"):
            assert cnt[0] < reads.n // 20, cnt           # (sparse input: a few tiles span more than the staged buckets)


def quirk_case():
    """Hand-built reads whose classification hinges on quirks Q1, Q4, Q5 of the reference;
    returns (annotation, reads, position of row i in the sorted read set)."""
    af = _anno([(0, 0, [(100, 200), (300, 400), (500, 600)]),          # Q1 transcript
                (0, 0, [(10_000, 10_100)])])                          # Q5 transcript (one exon)
    I, D = 1, 2
    rows = [
        # Q1 (update_gtf.c:746): the acceptor loop looks at read exon j's OWN start.  A chain that begins with the
        # transcript's first exon has no matching acceptor -> not known, only "has known site" ...
        (0, *_chain([(100, 200), (300, 400)])),
        # ... the same kind of chain beginning at an annotated acceptor is known
        (0, *_chain([(300, 400), (500, 600)])),
        # Q4 (bam2gtf.c:52-66, -e 3): a 2-base inner exon is dropped and its two introns fuse; a 2-base FIRST or LAST exon stays
        (0, 2_000, [(50, M), (100, N_), (2, M), (100, N_), (50, M)]),
        (0, 3_000, [(2, M), (100, N_), (50, M)]),
        (0, 4_000, [(50, M), (100, N_), (2, M)]),
        # ... and an insertion / a short deletion (<= -t 50) do not cut, a long deletion does
        (0, 5_000, [(30, M), (5, I), (20, M), (50, D), (10, M), (51, D), (40, M)]),
        # Q5 (update_gtf.c:786-801, <=): touching the transcript by its first / last base only is no overlap
        (0, *_chain([(9_900, 10_000)])),
        (0, *_chain([(9_900, 10_001)])),
        (0, *_chain([(10_099, 10_200)])),
        (0, *_chain([(10_100, 10_200)])),
    ]
    rows = [(r[0], r[1], 0, r[2]) for r in rows]
    order = sorted(range(len(rows)), key=lambda i: (rows[i][0], rows[i][1]))
    pos_of = {orig: new for new, orig in enumerate(order)}
    return af, _reads([rows[i] for i in order]), pos_of


def check_quirk_outcomes(want5, want2, pos_of):
    """The documented outcomes (want5 / want2: results at -l 5 / -l 2)."""
    info = lambda w, i: int(w.info[pos_of[i]])
    exons = lambda i: list(zip(want5.ex_start[want5.ex_off[pos_of[i]]:want5.ex_off[pos_of[i] + 1]].tolist(),
                               want5.ex_end[want5.ex_off[pos_of[i]]:want5.ex_off[pos_of[i] + 1]].tolist()))
    assert (info(want5, 0) & 1) == 0 and (info(want5, 0) & 2) != 0    # Q1: not known, has a known site
    assert (info(want5, 1) & 1) != 0 and int(want5.ref_tx[pos_of[1]]) == 0     # known
    assert exons(2) == [(2_001, 2_050), (2_253, 2_302)]              # Q4: inner micro-exon gone, introns fused
    assert exons(3) == [(3_001, 3_002), (3_103, 3_152)]              # first exon is never length-checked
    assert exons(4) == [(4_001, 4_050), (4_151, 4_152)]              # nor the last
    assert exons(5) == [(5_001, 5_110), (5_162, 5_201)]              # I ignored, D 50 absorbed, D 51 cuts
    # Q5 at -l 2 (full <=> first and last exons overlap; the only exon of the transcript is both): a read that merely
    # touches the transcript never reaches check_full
    assert [(info(want2, i) & 4) != 0 for i in (6, 7, 8, 9)] == [False, True, True, False]


def test_reference_quirks_q1_q4_q5(oracle):
    """The GPU path agrees with the oracle on the quirk reads, and the oracle shows the documented outcome."""
    af, reads, pos_of = quirk_case()
    _, want5 = _run(oracle, af, reads, full_level=5)
    _, want2 = _run(oracle, af, reads, full_level=2)
    check_quirk_outcomes(want5, want2, pos_of)


def full_length_case():
    """check_full / set_full (update_gtf.c:629-696) worked out by hand: transcript (100,200)(300,400)(500,600);
    read A ends inside the transcript (its last exon is the transcript's MIDDLE exon), read B's last exon is novel."""
    af = _anno([(0, 0, [(100, 200), (300, 400), (500, 600)])])
    rows = [(0, *_chain([(150, 200), (300, 400)])),         # A
            (0, *_chain([(160, 200), (700, 800)]))]         # B
    rows = [(r[0], r[1], 0, r[2]) for r in rows]
    # level:            1      2      3      4     5
    expect = {"A": [False, False, False, True, True],      # left end anchored; right end overlaps a non-terminal exon
              "B": [False, False, True, True, True]}       # right end overlaps nothing of the transcript: not held against it at -l 3
    return af, _reads(rows), expect


def check_full_length_outcomes(results_by_level, expect):
    for name, i in (("A", 0), ("B", 1)):
        got = [(int(results_by_level[l].info[i]) & 4) != 0 for l in (1, 2, 3, 4, 5)]
        assert got == expect[name], (name, got)


def test_full_length_levels_known_answers(oracle):
    af, reads, expect = full_length_case()
    res = {l: _run(oracle, af, reads, full_level=l)[1] for l in (1, 2, 3, 4, 5)}
    check_full_length_outcomes(res, expect)


@pytest.mark.parametrize("dis", [1, 2, 7, 20, 64])          # (64 = DIS_MASK_MAX: tolerance windows of 129 bases straddle the 512-bp buckets everywhere)
@pytest.mark.parametrize("level", [1, 3, 5])
def test_splice_distance_on_the_mask_path(oracle, dis, level, pipeline):
    """-d > 0 (src/update_gtf.c:717-779 with dis > 0) on the mask kernels: every probe looks at the annotation sites within the
    tolerance (probe_near).  Reads whose sites sit 0 .. dis + 1 bases off an annotation site on either side, exons that start /
    end up to dis outside the first / last annotated base of their tile (the staged slices reach that far), annotation sites just
    inside and just outside the read's own span (an annotation site only counts inside the overlap span, the flags know no span),
    single-exon reads.  Nothing goes to the generic kernel: no transcript here has two sites within the tolerance of each other."""
    rng = np.random.default_rng(40 + dis)
    txs = []
    for g in range(40):
        base = 10_000 + g * 9_000
        ex, x = [], base
        for _ in range(int(rng.integers(3, 8))):
            ln = int(rng.integers(60, 300))
            ex.append((x, x + ln))
            x += ln + int(rng.integers(120, 900))
        txs.append((0 if g < 25 else 1, g & 1, ex))
        if g % 3 == 0 and len(ex) > 3:                        # an isoform that skips an exon
            txs.append((0 if g < 25 else 1, g & 1, ex[:1] + ex[2:]))
    af = _anno(txs)
    rows = []
    for i in range(4000):
        t = txs[int(rng.integers(len(txs)))]
        ex = [list(x) for x in t[2]]
        a = int(rng.integers(0, len(ex) - 1)); b = int(rng.integers(a + 1, len(ex))) + 1
        ex = ex[a:b]
        mode = i % 8
        if mode == 1:                                            # every site jittered within the tolerance
            for q in ex:
                q[0] += int(rng.integers(-dis, dis + 1)); q[1] += int(rng.integers(-dis, dis + 1))
        elif mode == 2:                                          # one site just outside it
            q = ex[int(rng.integers(len(ex)))]
            q[int(rng.integers(2))] += (dis + 1) * (1 if rng.integers(2) else -1)
        elif mode == 3:                                          # the read ends right at / just inside / just outside a donor
            ex[-1][1] = ex[-1][0] + int(rng.integers(0, 2 * dis + 3))
        elif mode == 4:                                          # ... and begins around an acceptor
            ex[0][0] = ex[0][1] - int(rng.integers(0, 2 * dis + 3))
        elif mode == 5 and len(ex) > 2:                          # skipped exon with jitter
            del ex[1]; ex[0][1] += int(rng.integers(-dis, dis + 1))
        elif mode == 6:
            ex = [ex[0]]                                         # single exon
        elif mode == 7:
            ex[0][0] -= int(rng.integers(0, dis + 2)); ex[-1][1] += int(rng.integers(0, dis + 2))
        ex = [(max(1, x), max(max(1, x), y)) for x, y in ex]
        ok = all(ex[k][1] + 3 < ex[k + 1][0] for k in range(len(ex) - 1))
        if not ok:
            continue
        p, ops = _chain(ex)
        rows.append((t[0], p, i & 1, ops))
    rows.sort(key=lambda r: (r[0], r[1]))
    cnt = [0, 0, 0, 0]
    got, want = _run(oracle, af, _reads(rows), counters=cnt, full_level=level, ss_dis=dis)
    k = (want.info & 1) != 0
    assert k.sum() > 200 and (((want.info & 2) != 0) & ~k).sum() > 200
    if pipeline in ("tile", "slab"):
        assert cnt[0] == 0, cnt                                  # nothing for the generic kernel


def test_splice_distance_with_two_annotation_sites_inside_the_tolerance(oracle, pipeline):
    """identical_site_n counts (annotation site, read site) PAIRS (update_gtf.c:735-750): transcript A has donors at 1100 and 1104,
    so at -d 3 a read donor at 1101 .. 1103 makes TWO pairs with it -- 3 pairs for the read's 2 sites: A is NOT "known" although every
    read site matches one of its sites, and the sweep goes on to B (donor 1100 only), which is.  The masks cannot count pairs: a
    read with such a member in its window goes to the generic kernel and comes out as the reference has it."""
    txs = [(0, 0, [(500, 600), (1_000, 1_100), (1_102, 1_104), (1_400, 1_500)]),      # A: donors 600, 1100, 1104
           (0, 0, [(500, 600), (1_000, 1_100), (1_400, 1_500)]),                      # B
           (0, 1, [(4_500, 4_600), (5_000, 5_100), (5_300, 5_400)])]
    af = _anno(txs)
    rows = []
    for k in range(300):
        rows.append((0, *_chain([(1_000, 1_100 + (k % 6)), (1_400 - (k % 3), 1_500)])))
        rows.append((0, *_chain([(5_000, 5_100 + (k % 5) - 2), (5_300, 5_400)])))
    rows = [(r[0], r[1], 0, r[2]) for r in rows]
    rows.sort(key=lambda r: (r[0], r[1]))
    cnt = [0, 0, 0, 0]
    got, want = _run(oracle, af, _reads(rows), counters=cnt, full_level=3, ss_dis=3)
    known = (want.info & 1) != 0
    assert known.sum() > 300 and (want.ref_tx[known] == 1).sum() >= 100 and (want.ref_tx[known] == 0).sum() >= 50      # B for the twin-donor reads, A for donor 1100 itself
    if pipeline in ("tile", "slab"):
        assert 0 < cnt[0] <= 300, cnt                            # the reads around the twin donors, nobody else


def test_scans_in_launches_of_their_own(oracle, pipeline, monkeypatch):
    """Shards beyond 262 k tiles (67 M reads) scan the tiles' exon counts and the deferred accepted counts with k_scan_u32 in launches of
    their own instead of the segmented scans (l2r_kernels.hip.h SEG_MAX); L2R_SEG_MAX=0 takes that path on a small input: results and
    accepted list as ever, with and without a junction table."""
    monkeypatch.setenv("L2R_SEG_MAX", "0")
    anno, af, reads = util.make_case(31, n_reads=30000, n_exons=6, anno_exons=20000)
    got, want = _run(oracle, af, reads, full_level=3)
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
    j, sj = util.junction_table(af, reads, base, 31, cover=0.7)
    _run(oracle, af, reads, sj=sj, full_level=3, split_trans=1, min_sj_cnt=1)
    assert ((want.info & 7) == 6).sum() > 1000            # full, has a known site, not known: the reads the accepted list holds


def test_sparse_stretches_make_their_own_tiles_small_and_no_others(oracle, pipeline):
    """Tiles of the slab pipeline are cut by span one by one (a tile's reads begin less than 2^17 bases apart): a sparse stretch of
    the read set makes ITS tiles small.  The upload used to halve every tile of the read set when a sample of 256-read windows
    held sparse ones (a rule the classic pipeline needs): twice the tiles on an annotation with busy loci between quiet stretches.
    Ten busy loci of 3000 reads each, between them stretches of 600 reads one kb apart (256 of those span more than the staged
    bucket directory, 128 do not: the old rule chose 128 for every tile -- 235 tiles for the busy loci instead of 120)."""
    if pipeline not in ("tile", "slab"):
        pytest.skip("the classic pipeline keeps its rule")
    exons = [(0, 300), (900, 1_200), (2_000, 2_400)]
    txs, rows = [], []
    rng = np.random.default_rng(11)

    def locus(base, n_reads):
        txs.append((0, 0, [(base + s, base + e) for s, e in exons]))
        for _ in range(n_reads):
            j = int(rng.integers(0, 40))
            rows.append((0, *_chain([(base + exons[0][0] + j, base + exons[0][1]), (base + exons[1][0], base + exons[1][1]),
                                     (base + exons[2][0], base + exons[2][1] - j)])))

    at = 10_000
    for k in range(10):
        locus(at, 3_000)
        at += 5_000
        if k < 9:
            for _ in range(600):
                locus(at, 1)
                at += 1_000
    rows = sorted(((r[0], r[1], 0, r[2]) for r in rows), key=lambda r: (r[0], r[1]))
    cnt = [0, 0, 0, 0]
    _run(oracle, _anno(txs), _reads(rows), counters=cnt, full_level=3)
    busy = 10 * -(-3_000 // 256)
    assert busy <= cnt[3] <= busy + 9 * 7 + 10, cnt              # (a stretch of 600 kb: five to six tiles by span, plus the seams)
