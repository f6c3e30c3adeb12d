"""CPU, world_size 2, gloo: the sharded update-gtf driver gives the single-process files.

The shard classifier is the oracle here (no GPU in this container); what is under test is the N > 1
path itself: sharding by bytes, the all-gatherv with unequal sizes, offset rebasing, rank-0 tail."""
import filecmp
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from lr2rmats_amd import synth, workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, %r)
from lr2rmats_amd import dist as l2r_dist, capi
from oracle import pyoracle as po

def classify(job, lo, hi):
    a, r, sj, p = job.annotation_arrays(), job.read_arrays(), job.junction_arrays(), job.prm
    op = po.default_params(min_exon=p.min_exon, min_intron=p.min_intron, max_delet=p.max_delet, ss_dis=p.ss_dis, end_dis=p.end_dis,
                           full_level=p.full_level, split_trans=p.split_trans, use_multi=p.use_multi, min_sj_cnt=p.min_sj_cnt,
                           force_strand=p.force_strand, single_exon_ovlp_frac=p.single_exon_ovlp_frac)
    c0 = int(r["cig_off"][lo])
    res = po.classify_soa(r["tid"][lo:hi], r["pos"][lo:hi], r["rev"][lo:hi], r["cig_off"][lo:hi + 1] - c0, r["cig"][c0:int(r["cig_off"][hi])],
                          a["tx_tid"], a["tx_start"], a["tx_end"], a["tx_rev"], a["tx_ex_off"], a["ex_start"], a["ex_end"], sj=sj, params=op)
    info = (res.info & 0x7f) | (np.diff(res.ex_off).astype(np.uint32) << 8)
    return capi.Result(res.ex_off, res.ex_start, res.ex_end, res.ex_flag, info, res.ref_tx)

sys.exit(l2r_dist.run(sys.argv[1:], classify=classify, backend="gloo"))
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_shard_bounds():
    b = workload.shard_bounds(10, 3)
    assert b[0][0] == 0 and b[-1][1] == 10 and all(b[i][1] == b[i + 1][0] for i in range(2))
    w = np.array([1, 1, 1, 1, 100, 1, 1, 1], float)
    b = workload.shard_bounds(8, 2, w)
    assert b[0][0] == 0 and b[1][1] == 8 and b[0][1] == b[1][0]
    assert workload.algorithmic_bytes(10, 150, 80, 5, 50) == 4 * 150 + 120 + 640 + (400 - 40) + 80 + 400 + 100


def _run_ranks(world, argv, extra_env=None, stdout_path=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.update(extra_env or {})
        out = open(stdout_path, "wb") if (stdout_path and rank == 0) else subprocess.DEVNULL
        procs.append((subprocess.Popen([sys.executable, "-c", _WORKER % ROOT] + argv, env=env, stderr=subprocess.PIPE, stdout=out), out))
    for p, out in procs:
        _, err = p.communicate(timeout=280)
        if out is not subprocess.DEVNULL:
            out.close()
        assert p.returncode == 0, err.decode()[-3000:]


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("dist")
    anno = synth.make_annotation(5000, 51, nchr=5, shuffle_within_gene=True)
    reads = synth.make_reads(anno, 5001, 5, 51)
    sam, gtf = str(d / "r.sam"), str(d / "a.gtf")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    return d, anno, reads, sam, gtf


KEYS = ("gtf", "detail", "bed", "summary", "known", "novel", "unrec", "all")


def _args(o, sam, gtf, extra=()):
    return ["update-gtf", "-l", "3"] + list(extra) + ["-A", o["detail"], "-E", o["bed"], "-y", o["summary"], "-k", o["known"], "-v", o["novel"],
                                                        "-u", o["unrec"], "-a", o["all"], "-o", o["gtf"], sam, gtf]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,route", [(2, "partitioned"), (3, "partitioned"), (2, "gathered")])
def test_ranks_equal_single_process(oracle, tmp_path, inputs, world, route):
    d, anno, reads, sam, gtf = inputs
    single = {k: str(tmp_path / ("s." + k)) for k in KEYS}
    multi = {k: str(tmp_path / ("m." + k)) for k in KEYS}
    assert oracle.run_cli(_args(single, sam, gtf)) == 0
    _run_ranks(world, _args(multi, sam, gtf), {"L2R_DIST_GATHER": "1"} if route == "gathered" else None)
    for k in KEYS:
        assert filecmp.cmp(single[k], multi[k], shallow=False), (world, route, k)
    assert os.path.getsize(single["detail"]) > 100000
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]


@pytest.mark.timeout(300)
def test_split_with_junctions_takes_the_gathered_route_and_stdout(oracle, tmp_path, inputs):
    # -s with a junction table: split pieces are compared across chromosomes (Q2), so the tail runs once on rank 0;
    # the updated GTF goes to stdout here
    d, anno, reads, sam, gtf = inputs
    from tests import util
    af = anno.in_file_order()
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
    j, _ = util.junction_table(af, reads, base, 51, cover=0.7)
    tab = str(tmp_path / "SJ.out.tab")
    j.write(tab)
    extra = ["-s", "-J", "1", "-j", tab]
    a, b = str(tmp_path / "s.gtf"), str(tmp_path / "m.gtf")
    det_a, det_b = str(tmp_path / "s.det"), str(tmp_path / "m.det")
    assert oracle.run_cli(["update-gtf", "-l", "3"] + extra + ["-A", det_a, sam, gtf], stdout_path=a) == 0
    _run_ranks(2, ["update-gtf", "-l", "3"] + extra + ["-A", det_b, sam, gtf], stdout_path=b)
    assert filecmp.cmp(a, b, shallow=False) and filecmp.cmp(det_a, det_b, shallow=False)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("extra", [(), ("-s",)])
def test_gathered_route_sends_only_the_accepted_reads_when_no_output_lists_every_read(oracle, tmp_path, inputs, extra):
    # `update-gtf ... -o new.gtf -v novel.gtf -E bed`: the message of the all-gatherv is the accepted list (records + exons),
    # rank 0 sorts the chunks into input order and runs the tail over those reads alone
    d, anno, reads, sam, gtf = inputs
    from tests import util
    af = anno.in_file_order()
    sj_args = []
    if extra:
        base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
        j, _ = util.junction_table(af, reads, base, 52, cover=0.6)
        tab = str(tmp_path / "SJ.out.tab")
        j.write(tab)
        sj_args = ["-s", "-J", "1", "-j", tab]
    names = ("gtf", "novel", "bed")
    single = {k: str(tmp_path / ("s." + k)) for k in names}
    multi = {k: str(tmp_path / ("m." + k)) for k in names}
    args = lambda o: ["update-gtf", "-l", "3"] + sj_args + ["-v", o["novel"], "-E", o["bed"], "-o", o["gtf"], sam, gtf]
    assert oracle.run_cli(args(single)) == 0
    _run_ranks(3, args(multi), {"L2R_DIST_GATHER": "1", "L2R_DIST_TRACE": str(tmp_path / "trace")})
    for k in names:
        assert filecmp.cmp(single[k], multi[k], shallow=False), k
    assert os.path.getsize(single["gtf"]) > 10000
    assert open(str(tmp_path / "trace")).read().strip() == "gathered accepted"


@pytest.mark.timeout(300)
@pytest.mark.parametrize("grouped", [True, False])
def test_unsorted_records_are_classified_by_one_rank(oracle, tmp_path, inputs, grouped):
    # records that are not coordinate sorted: the reference's cursors depend on every earlier record, so a shard that
    # starts in the middle would classify differently -- rank 0 takes them all; files = single process
    d, anno, reads, sam, gtf = inputs
    rng = np.random.default_rng(7)
    lines = open(sam).read().splitlines(keepends=True)
    hdr = [l for l in lines if l.startswith("@")]
    body = [l for l in lines if not l.startswith("@")]
    if grouped:         # chromosome blocks stay together, order inside a block is shuffled
        by = {}
        for l in body:
            by.setdefault(l.split("\t")[2], []).append(l)
        body = []
        for k in by:
            body += [by[k][i] for i in rng.permutation(len(by[k]))]
    else:
        body = [body[i] for i in rng.permutation(len(body))]
    usam = str(tmp_path / "u.sam")
    open(usam, "w").write("".join(hdr + body))
    single = {k: str(tmp_path / ("s." + k)) for k in KEYS}
    multi = {k: str(tmp_path / ("m." + k)) for k in KEYS}
    assert oracle.run_cli(_args(single, usam, gtf)) == 0
    _run_ranks(2, _args(multi, usam, gtf), {"L2R_DIST_TRACE": str(tmp_path / "trace")})
    for k in KEYS:
        assert filecmp.cmp(single[k], multi[k], shallow=False), k
    assert open(str(tmp_path / "trace")).read().strip() == "one rank, gathered full"
    # and the sorted run of the same records is a different classification for some reads: the test would not notice
    # a wrong route otherwise
    s2 = {k: str(tmp_path / ("t." + k)) for k in KEYS}
    assert oracle.run_cli(_args(s2, sam, gtf)) == 0
    assert not filecmp.cmp(single["summary"], s2["summary"], shallow=False) or not filecmp.cmp(single["known"], s2["known"], shallow=False)


@pytest.mark.timeout(300)
def test_partitioned_route_counts_a_gene_id_that_spans_chromosomes_once(oracle, tmp_path, inputs):
    # merge_gene compares the gene_id with the list's last entry before its tid break (src/update_gtf.c:181-189): with one
    # gene_id on every chromosome the sequential summary says 1 gene; the ranks exchange their lists' last ids to agree
    import re
    d, anno, reads, sam, gtf = inputs
    gtf1 = str(tmp_path / "one_gene.gtf")
    with open(gtf) as fi, open(gtf1, "w") as fo:
        for l in fi:
            fo.write(re.sub(r'gene_id "[^"]*"', 'gene_id "GX"', l))
    single = {k: str(tmp_path / ("s." + k)) for k in KEYS}
    multi = {k: str(tmp_path / ("m." + k)) for k in KEYS}
    assert oracle.run_cli(_args(single, sam, gtf1)) == 0
    _run_ranks(3, _args(multi, sam, gtf1), {"L2R_DIST_TRACE": str(tmp_path / "trace")})
    assert open(str(tmp_path / "trace")).read().strip() == "partitioned"
    for k in KEYS:
        assert filecmp.cmp(single[k], multi[k], shallow=False), k
    assert "Updated_Genes\t1\n" in open(single["summary"]).read()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 5])
def test_bam_input_is_loaded_shard_by_shard(oracle, tmp_path, inputs, world):
    # coordinate-sorted BAM on the partitioned route: every rank extracts only its chromosome-aligned shard of the records
    # (hostlib.Job(rank, world)); 5 ranks over 5 chromosomes, and 2; files = the single-process files
    d, anno, reads, sam, gtf = inputs
    bam = str(tmp_path / "r.bam")
    synth.write_bam(reads, bam)
    single = {k: str(tmp_path / ("s." + k)) for k in KEYS}
    multi = {k: str(tmp_path / ("m." + k)) for k in KEYS}
    assert oracle.run_cli(_args(single, sam, gtf)) == 0
    _run_ranks(world, _args(multi, bam, gtf), {"L2R_DIST_TRACE": str(tmp_path / "trace"), "L2R_DIST_SHARD_TRACE": str(tmp_path / "shards")})
    assert open(str(tmp_path / "trace")).read().strip() == "partitioned"
    for k in KEYS:
        assert filecmp.cmp(single[k], multi[k], shallow=False), (world, k)
    # every rank reported a shard of its own: together they cover the file once, cut at chromosome boundaries
    rows = sorted(tuple(int(x) for x in open(str(tmp_path / ("shards.%d" % r))).read().split()) for r in range(world))
    assert all(r[3] == reads.n and r[0] == 1 for r in rows)
    spans = sorted((r[1], r[2]) for r in rows)
    assert spans[0][0] == 0 and spans[-1][1] == reads.n and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    for lo, hi in spans:
        assert lo == hi or lo == 0 or reads.tid[lo - 1] != reads.tid[lo]
    assert sum(1 for lo, hi in spans if hi > lo) >= 2


@pytest.mark.timeout(300)
def test_partitioned_route_to_stdout(oracle, tmp_path, inputs):
    d, anno, reads, sam, gtf = inputs
    a, b = str(tmp_path / "s.gtf"), str(tmp_path / "m.gtf")
    assert oracle.run_cli(["update-gtf", "-l", "3", sam, gtf], stdout_path=a) == 0
    _run_ranks(2, ["update-gtf", "-l", "3", sam, gtf], {"TMPDIR": str(tmp_path)}, stdout_path=b)
    assert filecmp.cmp(a, b, shallow=False)


def test_aligned_shard_bounds():
    tid = np.repeat(np.arange(5, dtype=np.int32), [10, 50, 5, 20, 15])
    b = workload.aligned_shard_bounds(tid, 3)
    assert b[0][0] == 0 and b[-1][1] == 100 and all(b[i][1] == b[i + 1][0] for i in range(2))
    for lo, hi in b:
        assert lo == hi or lo == 0 or tid[lo - 1] != tid[lo]
    assert workload.aligned_shard_bounds(np.array([0, 1, 0], np.int32), 2) is None
    b = workload.aligned_shard_bounds(np.zeros(10, np.int32), 4)          # one chromosome, four ranks: three empty shards
    assert sum(hi - lo for lo, hi in b) == 10 and sum(1 for lo, hi in b if hi > lo) == 1


@pytest.mark.timeout(600)
def test_every_rank_inflates_only_its_block_range(oracle, tmp_path):
    """A coordinate-sorted BGZF BAM on the partitioned route: rank r inflates the blocks from its byte target (file size * r / world) to
    the end of its records -- its share of the file plus at most one chromosome and a block -- and finds its first record without the
    records in front of it (host/aln_reader.c h_read_alignments_blocks); the ranks' ranges meet (dist.py checks that, the trace says
    what each rank inflated) and the files are the single-process files."""
    world, nchr = 3, 12
    anno = synth.make_annotation(12000, 77, nchr=nchr, shuffle_within_gene=True)
    reads = synth.make_reads(anno, 150000, 5, 77)
    sam, gtf, bam = str(tmp_path / "r.sam"), str(tmp_path / "a.gtf"), str(tmp_path / "r.bam")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    synth.write_bam(reads, bam)
    single = {k: str(tmp_path / ("s." + k)) for k in KEYS}
    multi = {k: str(tmp_path / ("m." + k)) for k in KEYS}
    assert oracle.run_cli(_args(single, sam, gtf)) == 0
    _run_ranks(world, _args(multi, bam, gtf), {"L2R_DIST_TRACE": str(tmp_path / "trace"), "L2R_DIST_SHARD_TRACE": str(tmp_path / "shards")})
    assert open(str(tmp_path / "trace")).read().strip() == "partitioned"
    for k in KEYS:
        assert filecmp.cmp(single[k], multi[k], shallow=False), k
    rows = [tuple(int(x) for x in open(str(tmp_path / ("shards.%d" % r))).read().split()) for r in range(world)]
    fsz = os.path.getsize(bam)
    assert fsz > 20 * 65536 // 4                                # enough blocks for the ranges to mean something
    spans = [(r[1], r[2]) for r in rows]
    assert spans[0][0] == 0 and spans[-1][1] == reads.n and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    for lo, hi in spans:
        assert lo == hi or lo == 0 or reads.tid[lo - 1] != reads.tid[lo]
    for r in rows:
        assert r[0] == 1 and r[3] == reads.n and r[5] == fsz and r[4] > 0, r          # (r[4] == 0: the fallback ran, every rank inflated the file)
        assert r[4] <= fsz * (1.0 / world + 1.5 / nchr) + 3 * 65536, (r, fsz)
    assert sum(r[4] for r in rows) < 1.6 * fsz


@pytest.mark.timeout(600)
def test_block_ranges_that_cannot_be_trusted_fall_back(oracle, tmp_path):
    """The second half of this BAM is not coordinate sorted: rank 0 reads its block range without noticing, rank 1 finds no chain of sorted
    records to begin at and reads the file the other way.  The ranks compare notes (dist.py), rank 0 reloads, and the run takes the
    route of unsorted input: one rank classifies everything; files = the single-process files of the same records."""
    anno = synth.make_annotation(6000, 78, nchr=6, shuffle_within_gene=True)
    reads = synth.make_reads(anno, 60000, 5, 78)
    rng = np.random.default_rng(78)
    perm = np.arange(reads.n)
    half = reads.n // 2
    perm[half:] = half + rng.permutation(reads.n - half)
    lens = np.diff(reads.cig_off)[perm]
    g = synth._ragged_gather_index(reads.cig_off[perm], lens)
    off = np.zeros(reads.n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    mixed = synth.Reads(reads.chrom_names, reads.tid[perm], reads.pos[perm], reads.rev[perm], reads.flag_rev[perm], reads.has_xs[perm], off, reads.cig[g],
                        reads.name_base, reads.chrom_len, False)
    sam, gtf, bam = str(tmp_path / "r.sam"), str(tmp_path / "a.gtf"), str(tmp_path / "r.bam")
    mixed.write_sam(sam)
    anno.write_gtf(gtf)
    synth.write_bam(mixed, bam)
    single = {k: str(tmp_path / ("s." + k)) for k in KEYS}
    multi = {k: str(tmp_path / ("m." + k)) for k in KEYS}
    assert oracle.run_cli(_args(single, sam, gtf)) == 0
    _run_ranks(2, _args(multi, bam, gtf), {"L2R_DIST_TRACE": str(tmp_path / "trace")})
    assert open(str(tmp_path / "trace")).read().strip() == "one rank, gathered full"
    for k in KEYS:
        assert filecmp.cmp(single[k], multi[k], shallow=False), k


@pytest.mark.parametrize("world", [2, 3])
def test_a_bam_of_one_block_takes_the_fallback_of_the_block_ranges(oracle, tmp_path, world):
    """A sorted BAM so small that it is ONE BGZF block (plus the end-of-file block): every rank's byte target lies in that block, the
    block ranges cannot be told apart -- the ranks' notes do not meet (dist.py), every rank reopens its job with L2R_DIST_BLOCKS=0 and the
    shards are cut from the records of the whole file.  Files = the single-process files (the branch had no test)."""
    anno = synth.make_annotation(1500, 91, nchr=4)
    reads = synth.make_reads(anno, 400, 4, 91)
    sam, gtf, bam = str(tmp_path / "r.sam"), str(tmp_path / "a.gtf"), str(tmp_path / "r.bam")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    synth.write_bam(reads, bam)
    single = {k: str(tmp_path / ("s." + k)) for k in KEYS}
    multi = {k: str(tmp_path / ("m." + k)) for k in KEYS}
    assert oracle.run_cli(_args(single, sam, gtf)) == 0
    _run_ranks(world, _args(multi, bam, gtf))
    for k in KEYS:
        assert filecmp.cmp(single[k], multi[k], shallow=False), k
