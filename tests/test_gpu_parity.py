"""GPU: the HIP path, called through the C-ABI, against the CPU oracle -- bit exact."""
import numpy as np
import pytest

from lr2rmats_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["tile", "slab", "classic"])
def engine(request):
    """One engine per kernel pipeline (l2r_create reads L2R_PIPELINE): every case of this file runs on both."""
    import os
    old = os.environ.get("L2R_PIPELINE")
    os.environ["L2R_PIPELINE"] = request.param
    try:
        e = capi.Engine(0)
        e.pipeline = request.param
    finally:
        if old is None:
            del os.environ["L2R_PIPELINE"]
        else:
            os.environ["L2R_PIPELINE"] = old
    yield e
    e.close()


def _set_anno(engine, af):
    engine.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)


def _check(engine, oracle, af, reads, op, sj=None):
    want = util.oracle_run(oracle, af, reads, op, sj)
    engine.set_junctions(sj)
    got = engine.classify(reads, util.to_engine_params(capi, op))
    util.assert_same_result(got, want, 0 if sj is None else len(sj[0]), op.split_trans)
    return got, want


@pytest.mark.parametrize("level", [1, 2, 3, 4, 5])
def test_levels(engine, oracle, level):
    anno, af, reads = util.make_case(11, n_reads=30000, n_exons=5, anno_exons=20000)
    _set_anno(engine, af)
    got, want = _check(engine, oracle, af, reads, oracle.default_params(full_level=level))
    # sanity: the case exercises all three classes
    k = (want.info & 1) != 0
    ks = ((want.info & 2) != 0) & ~k
    assert k.sum() > 100 and ks.sum() > 100 and (~k & ~ks).sum() > 100


@pytest.mark.parametrize("dis", [0, 2, 7])
def test_splice_distance(engine, oracle, dis):
    anno, af, reads = util.make_case(12, n_reads=20000, n_exons=6, anno_exons=15000)
    _set_anno(engine, af)
    _check(engine, oracle, af, reads, oracle.default_params(full_level=3, ss_dis=dis))
    # -d > 0 stays on the mask kernels (probe_near): a zero redo share
    import ctypes as C
    lib = capi.load_library()
    cnt = (C.c_longlong * 13)()
    lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.l2r_debug_counters(engine.ctx, cnt, 13)
    assert cnt[0] == 0, list(cnt)


@pytest.mark.parametrize("opts", [dict(min_exon=1), dict(min_exon=10, min_intron=200), dict(max_delet=3), dict(min_intron=0)])
def test_cigar_thresholds_ont(engine, oracle, opts):
    anno, af, reads = util.make_case(13, n_reads=8000, n_exons=6, anno_exons=15000, ont=True, micro=3, xs=0.02)
    _set_anno(engine, af)
    _check(engine, oracle, af, reads, oracle.default_params(full_level=3, **opts))
    # whichever pipeline the upload chose (-t 3 cuts these reads into dozens of exons: more than a slab has rows, so long-CIGAR input
    # of that kind stays with the classic kernels): the generic kernel is left with next to nothing
    import ctypes as C
    lib = capi.load_library()
    cnt = (C.c_longlong * 13)()
    lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.l2r_debug_counters(engine.ctx, cnt, 13)
    assert cnt[0] <= reads.n // 50, list(cnt)


def test_unsorted_gtf_and_long_transcripts(engine, oracle):
    anno, af, reads = util.make_case(14, n_reads=20000, n_exons=5, anno_exons=15000, shuffle=True, long_tx=2)
    _set_anno(engine, af)
    _check(engine, oracle, af, reads, oracle.default_params(full_level=3))
    _check(engine, oracle, af, reads, oracle.default_params(full_level=5, ss_dis=3))


@pytest.mark.parametrize("split,multi,mincnt,dis", [(0, 0, 1, 0), (1, 0, 1, 0), (1, 1, 5, 0), (0, 0, 3, 2)])
def test_junction_validation(engine, oracle, split, multi, mincnt, dis):
    anno, af, reads = util.make_case(15, n_reads=20000, n_exons=5, anno_exons=15000)
    _set_anno(engine, af)
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
    _, sj = util.junction_table(af, reads, base, 15, cover=0.7)
    op = oracle.default_params(full_level=3, split_trans=split, use_multi=multi, min_sj_cnt=mincnt, ss_dis=dis)
    got, want = _check(engine, oracle, af, reads, op, sj)
    assert ((want.info & 32) != 0).sum() > 100 and ((want.info & 64) != 0).sum() > 10
    assert ((want.ex_flag & 16) != 0).sum() > 10
    engine.set_junctions(None)


def test_sparse_junction_table_q7(engine, oracle):
    # a table with rows on a few chromosomes only: cursor exhausted / beyond the read (Q7)
    anno, af, reads = util.make_case(16, n_reads=10000, n_exons=5, anno_exons=10000)
    _set_anno(engine, af)
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=5))
    j, sj = util.junction_table(af, reads, base, 16, cover=0.5)
    keep = (j.tid % 5) == 2
    sj = tuple(x[keep] for x in sj)
    _check(engine, oracle, af, reads, oracle.default_params(full_level=5, split_trans=1), sj)
    engine.set_junctions(None)


def test_unsorted_reads_history_cursor(engine, oracle):
    anno, af, reads = util.make_case(17, n_reads=15000, n_exons=5, anno_exons=10000, unsorted=True)
    assert not reads.sorted
    _set_anno(engine, af)
    _check(engine, oracle, af, reads, oracle.default_params(full_level=3))
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
    _, sj = util.junction_table(af, reads, base, 17, cover=0.7)
    _check(engine, oracle, af, reads, oracle.default_params(full_level=3, split_trans=1), sj)
    engine.set_junctions(None)


def test_edge_shapes(engine, oracle):
    anno, af, reads = util.make_case(18, n_reads=3000, n_exons=4, anno_exons=5000)
    _set_anno(engine, af)
    op = oracle.default_params(full_level=3)
    # empty input
    empty = reads.slice(0, 0)
    got = engine.classify(empty, util.to_engine_params(capi, op))
    assert got.info.size == 0 and got.ex_start.size == 0 and got.ex_off.tolist() == [0]
    # one read; a ragged count that is not a multiple of the tile
    for n in (1, 257, 1023):
        _check(engine, oracle, af, reads.slice(0, n), op)
    # reads whose CIGAR is empty ('*') or starts with N: zero-length first exon
    r = reads.slice(0, 300)
    cig = r.cig.copy()
    cig[r.cig_off[5]] = (100 << 4) | 3            # first op of read 5 becomes an N
    r.cig = cig
    _check(engine, oracle, af, r, op)
    # many-exon reads: more than 64 exons per read and an oversize tile (HBM fallback path)
    n_big = 300
    ops_per = 2 * 90 - 1
    big = np.empty(n_big * ops_per, np.uint32)
    big[0::2][:] = (30 << 4) | 0
    tmp = big.reshape(n_big, ops_per)
    tmp[:, 1::2] = (120 << 4) | 3
    tmp[:, 0::2] = (30 << 4) | 0
    from lr2rmats_amd.synth import Reads
    pos = np.sort(np.random.default_rng(5).integers(10_000, 400_000, n_big)).astype(np.int32)
    rb = Reads(reads.chrom_names, np.zeros(n_big, np.int32), pos, np.zeros(n_big, np.uint8), np.zeros(n_big, np.uint8),
               np.zeros(n_big, np.uint8), np.arange(n_big + 1, dtype=np.int64) * ops_per, tmp.ravel())
    _check(engine, oracle, af, rb, op)


def test_single_exon_fraction_float_compare(engine, oracle):
    # Q6: float ratio compare at several -f values including ones that are not exactly representable
    anno, af, reads = util.make_case(19, n_reads=20000, n_exons=3, anno_exons=10000, full_frac=0.2)
    _set_anno(engine, af)
    for f in (0.8, 0.5, 0.3333333, 1.0, 0.95):
        _check(engine, oracle, af, reads, oracle.default_params(full_level=5, single_exon_ovlp_frac=f))


@pytest.mark.parametrize("n_reads,ablate", [(600000, None), (70000, "256")])
def test_runs_in_a_row_on_one_upload(engine, oracle, monkeypatch, n_reads, ablate):
    """From the second run on the launches a completed run has shown to be empty are skipped (k_probe_slab's lists, the generic kernel),
    and the words the tiles' exon counts add up in take turns run by run: five runs in a row, the same results.  600 k reads: two whole
    super-blocks of 1024 tiles and a partial one; L2R_ABLATE=256: no tile's exon count is known ahead, every tile adds its own inside the run."""
    anno, af, reads = util.make_case(21, n_reads=n_reads, n_exons=6, anno_exons=20000)
    _set_anno(engine, af)
    op = oracle.default_params(full_level=3)
    want = util.oracle_run(oracle, af, reads, op)
    if ablate:
        monkeypatch.setenv("L2R_ABLATE", ablate)
    engine.set_junctions(None)
    engine.set_params(util.to_engine_params(capi, op))
    engine.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
    for _ in range(5):
        engine.run(); engine.sync()
        util.assert_same_result(engine.download(), want, 0, 0)
    if ablate:
        monkeypatch.delenv("L2R_ABLATE")
        engine.set_params(util.to_engine_params(capi, op))


@pytest.mark.parametrize("opts", [dict(), dict(min_exon=40, min_intron=150, max_delet=2)])
def test_upload_with_the_readers_cigar_summaries(engine, oracle, opts):
    """l2r_reads::cig_summary (what a reader knows of a record's CIGAR while it converts it): the upload's tile index then touches no CIGAR
    (k_tile_index<true>) -- same tiles, same statistics, same results as with the engine's own CIGAR walk, under default thresholds (every
    tile exact) and under thresholds that are borderline in most tiles (counts published from k_tile / the slab pipeline)."""
    from lr2rmats_amd import synth
    anno, af, reads = util.make_case(33, n_reads=150000, n_exons=7, anno_exons=30000)
    _set_anno(engine, af)
    op = oracle.default_params(full_level=3, **opts)
    want = util.oracle_run(oracle, af, reads, op)
    engine.set_junctions(None)
    engine.set_params(util.to_engine_params(capi, op))
    sm = synth.cigar_summary(reads.cig_off, reads.cig)
    for summary in (None, sm, None, sm):
        engine.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, cig_summary=summary)
        assert engine.upload_index_ms() >= 0.0          # (0 where the upload makes no tile index: the classic pipeline)
        for _ in range(2):
            engine.run(); engine.sync()
            util.assert_same_result(engine.download(), want, 0, 0)


@pytest.mark.parametrize("how", ["a_tile_never_publishes", "device_too_small"])
def test_starved_look_back_falls_back_to_the_slab_pipeline(oracle, monkeypatch, how):
    """k_tile's tiles wait for the exon counts of the tiles in front of them; a count that never comes (here: L2R_ABLATE bit 15 -- tile 3 does not
    publish, with bit 8 -- no count is known ahead) ends every wait at the poll limit, and l2r_sync does the same resident upload again on the
    slab pipeline: results bit exact, the counter says so, later runs stay there.  A device that cannot hold the workgroups the look-back
    needs (L2R_TILE_STARVED stands in for the occupancy test of l2r_create) never takes the tile path."""
    import ctypes as C
    anno, af, reads = util.make_case(44, n_reads=40000, n_exons=6, anno_exons=20000)
    op = oracle.default_params(full_level=3)
    want = util.oracle_run(oracle, af, reads, op)
    if how == "a_tile_never_publishes":
        monkeypatch.setenv("L2R_ABLATE", str(256 + 32768))
    else:
        monkeypatch.setenv("L2R_TILE_STARVED", "1")
    monkeypatch.setenv("L2R_PIPELINE", "tile")
    eng = capi.Engine(0)
    try:
        eng.set_annotation(af.tx_tid, af.tx_start, af.tx_end, af.tx_rev, af.tx_ex_off, af.ex_start, af.ex_end)
        eng.set_junctions(None)
        eng.set_params(util.to_engine_params(capi, op))
        eng.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig)
        lib = capi.load_library()
        lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.l2r_stage_kernel.restype = C.c_char_p
        for k in range(3):
            eng.run(); eng.sync()
            util.assert_same_result(eng.download(), want, 0, 0)
            cnt = (C.c_longlong * 14)()
            lib.l2r_debug_counters(eng.ctx, cnt, 14)
            assert cnt[13] == (1 if how == "a_tile_never_publishes" else 0), list(cnt)
            assert b"k_walk_slab" in lib.l2r_stage_kernel(eng.ctx, 0)          # (the last launch was the slab pipeline's)
    finally:
        eng.close()


def _check_accepted_list(engine, got, first):
    """The accepted list of the last launch == the accepted reads of its full result, in read order."""
    acc = engine.download_accepted()
    idx = np.nonzero((got.info & 128) != 0)[0]
    np.testing.assert_array_equal(acc.read_index, idx + first)
    np.testing.assert_array_equal(acc.rec["info"], got.info[idx])
    np.testing.assert_array_equal(acc.rec["ref_tx"], got.ref_tx[idx])
    lens = (got.info[idx] >> 8).astype(np.int64)
    np.testing.assert_array_equal(np.diff(acc.ex_off), lens)
    from lr2rmats_amd.synth import _ragged_gather_index
    g = _ragged_gather_index(got.ex_off[idx], lens)
    np.testing.assert_array_equal(acc.ex_start, got.ex_start[g])
    np.testing.assert_array_equal(acc.ex_end, got.ex_end[g])
    np.testing.assert_array_equal(acc.ex_flag, got.ex_flag[g])
    return acc


def test_accepted_compaction_in_read_order(engine, oracle, monkeypatch):
    anno, af, reads = util.make_case(20, n_reads=25000, n_exons=5, anno_exons=15000)
    _set_anno(engine, af)
    op = oracle.default_params(full_level=3)
    engine.set_junctions(None)
    got = engine.classify(reads, util.to_engine_params(capi, op), first_read_index=1 << 33)
    acc = _check_accepted_list(engine, got, 1 << 33)
    assert acc.rec.shape[0] > 1000
    # the same list when k_gather_accepted places every tile's exons (what a junction table or redo reads lead to)
    monkeypatch.setenv("L2R_ABLATE", "2")
    got2 = engine.classify(reads, util.to_engine_params(capi, op), first_read_index=1 << 33)
    np.testing.assert_array_equal(got2.info, got.info)
    _check_accepted_list(engine, got2, 1 << 33)
    monkeypatch.delenv("L2R_ABLATE")
    # ... and with a junction table (acceptance decided by k_validate_sj)
    base = util.oracle_run(oracle, af, reads, op)
    _, sj = util.junction_table(af, reads, base, 20, cover=0.7)
    engine.set_junctions(sj)
    got3 = engine.classify(reads, util.to_engine_params(capi, oracle.default_params(full_level=3, min_sj_cnt=1)), first_read_index=5)
    assert 0 < int(((got3.info & 128) != 0).sum()) < int(((got.info & 128) != 0).sum())
    _check_accepted_list(engine, got3, 5)
    engine.set_junctions(None)


def test_full_size_properties(engine, oracle):
    """Config-2 size (100k reads x 5 exons, 50k-exon GTF) against the oracle, plus size-independent
    properties: idempotence (same launch twice) and shard invariance (two halves == whole)."""
    anno, af, reads = util.make_case(2, n_reads=100000, n_exons=5, anno_exons=50000)
    _set_anno(engine, af)
    op = oracle.default_params(full_level=3)
    got, want = _check(engine, oracle, af, reads, op)
    again = engine.classify(reads, util.to_engine_params(capi, op))
    for a, b in ((got.info, again.info), (got.ex_flag, again.ex_flag), (got.ref_tx, again.ref_tx)):
        np.testing.assert_array_equal(a, b)
    h = reads.n // 2
    lo = engine.classify(reads.slice(0, h), util.to_engine_params(capi, op))
    hi = engine.classify(reads.slice(h, reads.n), util.to_engine_params(capi, op), first_read_index=h)
    np.testing.assert_array_equal(np.concatenate([lo.info, hi.info]), got.info)
    np.testing.assert_array_equal(np.concatenate([lo.ex_flag, hi.ex_flag]), got.ex_flag)
    np.testing.assert_array_equal(np.concatenate([lo.ref_tx, hi.ref_tx]), got.ref_tx)


def test_config3_full_size(engine, oracle):
    """BASELINE config 3 at full size (10 M reads, 1.46 M-exon GTF, the bench workload): idempotence of the whole launch,
    shard invariance of a 1 M-read slice out of the middle, and that slice bit-exact against the oracle."""
    from lr2rmats_amd import workload
    af, reads = workload.make_rank_workload(dict(workload.CONFIGS["cfg3"]), 0, 1)
    _set_anno(engine, af)
    engine.set_junctions(None)
    op = oracle.default_params(full_level=3)
    prm = util.to_engine_params(capi, op)
    got = engine.classify(reads, prm)
    again = engine.classify(reads, prm)
    for name in ("ex_off", "ex_start", "ex_end", "ex_flag", "info", "ref_tx"):
        np.testing.assert_array_equal(getattr(got, name), getattr(again, name))
    assert got.ex_start.size == int(got.ex_off[-1]) == int((got.info >> 8).sum())
    _check_accepted_list(engine, got, 0)
    lo, hi = 4_500_000 - 4_500_000 % 256 + 128, 5_500_000            # not aligned to the tiles of the whole run
    part = reads.slice(lo, hi)
    sub = engine.classify(part, prm, first_read_index=lo)
    want = util.oracle_run(oracle, af, part, op)
    util.assert_same_result(sub, want, 0, 0)
    a, b = int(got.ex_off[lo]), int(got.ex_off[hi])
    np.testing.assert_array_equal(got.info[lo:hi], sub.info)
    np.testing.assert_array_equal(got.ref_tx[lo:hi], sub.ref_tx)
    np.testing.assert_array_equal(got.ex_flag[a:b], sub.ex_flag)
    np.testing.assert_array_equal(got.ex_start[a:b], sub.ex_start)


@pytest.mark.parametrize("cfgname,lo,ss_dis", [("cfg3_iso40", 4_200_000 + 77, 0), ("cfg3_gencode", 6_100_000 + 131, 0), ("cfg3_gencode", 2_300_000 + 19, 2)])
def test_isoform_rich_annotations_full_size(engine, oracle, cfgname, lo, ss_dis):
    """Config 3's reads against isoform-rich annotations at FULL size (10 M reads): `cfg3_iso40` (40 isoforms per gene: every
    window holds 33 .. 63 transcripts, k_probe_slab_wide and k_probe_slab_chunked classify nearly everything) and `cfg3_gencode`
    (isoforms per gene log-normal, up to 200: all three probe kernels and small tiles).  Idempotence, the accepted list, a
    1 M-read slice out of the middle bit-exact against the oracle and equal to the same reads of the whole run; nothing is left to
    the generic kernel (src/update_gtf.c:796-822: the reference's sweep knows no window limit).  `ss_dis` 2: the same with a splice-site
    tolerance (-d 2, src/update_gtf.c:717-779) -- all three probe kernels look within it."""
    import ctypes as C
    from lr2rmats_amd import workload
    if engine.pipeline not in ("tile", "slab"):
        pytest.skip("the classic pipeline leaves windows beyond 32 transcripts to the generic kernel: covered at small size")
    af, reads = workload.make_rank_workload(dict(workload.CONFIGS[cfgname]), 0, 1)
    _set_anno(engine, af)
    engine.set_junctions(None)
    op = oracle.default_params(full_level=3, ss_dis=ss_dis)
    prm = util.to_engine_params(capi, op)
    got = engine.classify(reads, prm)
    lib = capi.load_library()
    cnt = (C.c_longlong * 13)()
    lib.l2r_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.l2r_debug_counters(engine.ctx, cnt, 13)
    redo_share = cnt[0] / reads.n
    again = engine.classify(reads, prm)
    for name in ("ex_off", "ex_start", "ex_end", "ex_flag", "info", "ref_tx"):
        np.testing.assert_array_equal(getattr(got, name), getattr(again, name))
    _check_accepted_list(engine, got, 0)
    hi = lo + 1_000_000
    part = reads.slice(lo, hi)
    sub = engine.classify(part, prm, first_read_index=lo)
    want = util.oracle_run(oracle, af, part, op)
    util.assert_same_result(sub, want, 0, 0)
    a, b = int(got.ex_off[lo]), int(got.ex_off[hi])
    np.testing.assert_array_equal(got.info[lo:hi], sub.info)
    np.testing.assert_array_equal(got.ref_tx[lo:hi], sub.ref_tx)
    np.testing.assert_array_equal(got.ex_flag[a:b], sub.ex_flag)
    np.testing.assert_array_equal(got.ex_start[a:b], sub.ex_start)
    assert redo_share <= 1e-4, (cfgname, list(cnt))


def test_config5_shard(engine, oracle):
    """BASELINE config 5 at the size one of its 8 GPUs sees (2.5 M ONT-like reads on rank 0's chromosomes = the density of
    the 20 M-read set, 12 exons + 3 micro-exons, ~400 CIGAR ops per read, 2 M-exon GTF) with the pipeline's second-pass
    options (-s -l 3 -J 1 -j): idempotence, the accepted list, and a 150 k-read slice bit-exact against the oracle --
    without and with the junction table -- plus shard invariance of that slice."""
    from lr2rmats_amd import workload
    cfg = dict(workload.CONFIGS["cfg5"]); cfg["n_reads"] = 2_500_000
    af, reads = workload.make_rank_workload(cfg, 0, 8)
    _set_anno(engine, af)
    engine.set_junctions(None)
    op = oracle.default_params(full_level=3)
    prm = util.to_engine_params(capi, op)
    got = engine.classify(reads, prm)
    again = engine.classify(reads, prm)
    for name in ("ex_off", "ex_start", "ex_end", "ex_flag", "info", "ref_tx"):
        np.testing.assert_array_equal(getattr(got, name), getattr(again, name))
    _check_accepted_list(engine, got, 0)
    lo, hi = 1_200_000 + 77, 1_350_000
    part = reads.slice(lo, hi)
    want = util.oracle_run(oracle, af, part, op)
    sub = engine.classify(part, prm, first_read_index=lo)
    util.assert_same_result(sub, want, 0, 0)
    a, b = int(got.ex_off[lo]), int(got.ex_off[hi])
    np.testing.assert_array_equal(got.info[lo:hi], sub.info)
    np.testing.assert_array_equal(got.ref_tx[lo:hi], sub.ref_tx)
    np.testing.assert_array_equal(got.ex_flag[a:b], sub.ex_flag)
    np.testing.assert_array_equal(got.ex_start[a:b], sub.ex_start)
    np.testing.assert_array_equal(got.ex_end[a:b], sub.ex_end)
    # second pass: short-read junction table, split at unsupported junctions
    _, sj = util.junction_table(af, part, want, 5, cover=0.8)
    op2 = oracle.default_params(full_level=3, split_trans=1, min_sj_cnt=1)
    got2, want2 = _check(engine, oracle, af, part, op2, sj)
    assert ((want2.info & 32) != 0).sum() > 1000 and ((want2.ex_flag & 16) != 0).sum() > 100
    _check_accepted_list(engine, got2, 0)
    engine.set_junctions(None)


def test_annotation_table_cache(oracle, tmp_path):
    """l2r_set_annotation_cache: the first engine builds the tables and stores them (state 1), the second reads them
    (state 2) and classifies bit-exactly; another annotation gets a file of its own; a truncated file and one with a
    flipped payload byte are ignored and rewritten (state 1)."""
    import os
    anno, af, reads = util.make_case(19, n_reads=20000, n_exons=5, anno_exons=15000, shuffle=True, long_tx=1)
    op = oracle.default_params(full_level=3)
    want = util.oracle_run(oracle, af, reads, op)
    cache = tmp_path / "cache"

    def run(af_, reads_, want_):
        e = capi.Engine(0)
        try:
            e.set_annotation_cache(str(cache))
            e.set_annotation(af_.tx_tid, af_.tx_start, af_.tx_end, af_.tx_rev, af_.tx_ex_off, af_.ex_start, af_.ex_end)
            state = e.annotation_cache_state()
            got = e.classify(reads_, util.to_engine_params(capi, op))
        finally:
            e.close()
        util.assert_same_result(got, want_, 0, 0)
        return state

    assert run(af, reads, want) == 1
    files = sorted(os.listdir(cache))
    assert len(files) == 1 and files[0].endswith(".tables")
    assert run(af, reads, want) == 2 and sorted(os.listdir(cache)) == files
    anno2, af2, reads2 = util.make_case(20, n_reads=5000, n_exons=5, anno_exons=8000)
    want2 = util.oracle_run(oracle, af2, reads2, op)
    assert run(af2, reads2, want2) == 1 and len(os.listdir(cache)) == 2
    assert run(af2, reads2, want2) == 2
    path = cache / files[0]
    raw = bytearray(open(path, "rb").read())
    open(path, "wb").write(bytes(raw[: len(raw) - 1000]))
    assert run(af, reads, want) == 1                       # truncated: rebuilt + rewritten
    assert run(af, reads, want) == 2
    raw[len(raw) // 2] ^= 0x10
    open(path, "wb").write(bytes(raw))
    assert run(af, reads, want) == 1                       # payload checksum
    assert run(af, reads, want) == 2


@pytest.mark.parametrize("opts", [dict(), dict(min_intron=66000), dict(max_delet=69000), dict(min_intron=70001, max_delet=70000, min_exon=19)])
def test_summaries_at_their_saturation_limits(engine, oracle, opts):
    """Summary fields saturate at 65535: an N or a D operation of 70 000 bases with thresholds on either side of it, a stretch of 18 bases against
    -e 19 -- the tiles' exactness must come out on the safe side (a tile that might not be exact counts in k_tile), results as the oracle's."""
    from lr2rmats_amd import synth
    from tests.test_gpu_edges import _anno, _reads
    M, D, N = 0, 2, 3
    rows = []
    for i in range(600):
        base = 10_000 + 37 * i
        rows.append((0, base, i & 1, [(50, M), (70000, N), (10, M), (8, M), (300, N), (40, M), (70000, D), (25, M), (65535, N), (30, M)]))
        rows.append((0, base + 5, 0, [(60, M), (500, N), (18, M), (400, N), (33, M)]))
    rows.sort(key=lambda r: (r[0], r[1]))
    reads = _reads(rows)
    af = _anno([(0, 0, [(10_001, 10_050), (80_051, 80_068)]), (0, 1, [(10_100, 10_200), (10_701, 10_718), (11_119, 11_151)])]).in_file_order()
    _set_anno(engine, af)
    op = oracle.default_params(full_level=3, **opts)
    want = util.oracle_run(oracle, af, reads, op)
    engine.set_junctions(None)
    engine.set_params(util.to_engine_params(capi, op))
    sm = synth.cigar_summary(reads.cig_off, reads.cig)
    for summary in (sm, None):
        engine.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, cig_summary=summary)
        for _ in range(2):
            engine.run(); engine.sync()
            util.assert_same_result(engine.download(), want, 0, 0)
