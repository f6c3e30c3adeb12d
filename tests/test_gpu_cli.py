"""GPU: the `lr2rmats` C binary end to end (file in -> files out) against the oracle CLI -- byte identical."""
import filecmp
import os

import numpy as np
import pytest

from lr2rmats_amd import hostlib, synth
from tests import util

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "toy")
OUTS = ("gtf", "detail", "summary", "bed", "known", "novel", "unrec", "all")


def _paths(tmp, tag):
    return {k: str(tmp / ("%s.%s" % (tag, k))) for k in OUTS}


def _args(extra, o, sam, gtf):
    return ["update-gtf"] + extra + ["-A", o["detail"], "-y", o["summary"], "-E", o["bed"], "-k", o["known"], "-v", o["novel"],
                                     "-u", o["unrec"], "-a", o["all"], "-o", o["gtf"], sam, gtf]


def _compare(oracle, tmp_path, extra, aln, gtf, tag, oracle_aln=None):
    oo, ho = _paths(tmp_path, tag + ".o"), _paths(tmp_path, tag + ".h")
    assert oracle.run_cli(_args(extra, oo, oracle_aln or aln, gtf)) == 0
    r = hostlib.run_cli(_args(extra, ho, aln, gtf))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    for k in OUTS:
        assert filecmp.cmp(oo[k], ho[k], shallow=False), (tag, k)
    return oo


@pytest.mark.parametrize("name", ["l3", "sj_support", "sj_far", "sj_other"])
def test_toy_golden_files(tmp_path, name):
    """The committed known-answer files (values recorded from the reference)."""
    extra = ["-l", "3"] if name == "l3" else ["-s", "-l", "3", "-J", "1", "-j", os.path.join(G, name + ".tab")]
    o = {k: str(tmp_path / k) for k in ("gtf", "detail", "summary", "bed")}
    r = hostlib.run_cli(["update-gtf"] + extra + ["-A", o["detail"], "-y", o["summary"], "-E", o["bed"],
                                                 os.path.join(G, "toy.sam"), os.path.join(G, "original.gtf")], stdout_path=o["gtf"])
    assert r.returncode == 0, r.stderr.decode()
    for k, ext in (("gtf", "updated.gtf"), ("detail", "detail.txt"), ("summary", "summary.txt"), ("bed", "novel_exon.bed")):
        assert filecmp.cmp(o[k], os.path.join(G, "expect_%s.%s" % (name, ext)), shallow=False), (name, k)


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("syn")
    anno = synth.make_annotation(20000, 41, nchr=8, shuffle_within_gene=True, long_tx_per_chrom=1)
    reads = synth.make_reads(anno, 40000, 5, 41, xs_conflict_frac=0.02)
    sam, bam, gtf = str(d / "r.sam"), str(d / "r.bam"), str(d / "a.gtf")
    reads.write_sam(sam)
    synth.write_bam(reads, bam)
    anno.write_gtf(gtf)
    return d, anno, reads, sam, bam, gtf


@pytest.mark.parametrize("extra,tag", [(["-l", "3"], "a"), (["-l", "5", "-d", "2", "-c"], "b"), (["-l", "2", "-e", "6", "-i", "120", "-t", "10"], "c")])
def test_sam_and_bam_input(oracle, tmp_path, files, extra, tag):
    d, anno, reads, sam, bam, gtf = files
    _compare(oracle, tmp_path, extra, sam, gtf, tag + "s")
    _compare(oracle, tmp_path, extra, bam, gtf, tag + "b", oracle_aln=sam)


def test_pipeline_second_pass_options(oracle, tmp_path, files):
    """Snakefile:170 option set: -s -l 3 -J 1 -j SJ.tab -y -a -A -k -v -u -E."""
    d, anno, reads, sam, bam, gtf = files
    af = anno.in_file_order()
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
    j, _ = util.junction_table(af, reads, base, 41, cover=0.7)
    tab = str(d / "SJ.out.tab")
    j.write(tab)
    oo = _compare(oracle, tmp_path, ["-s", "-l", "3", "-J", "1", "-j", tab], bam, gtf, "p2", oracle_aln=sam)
    assert ".split." in open(oo["gtf"]).read()


def test_ont_like_reads(oracle, tmp_path):
    anno = synth.make_annotation(8000, 43, nchr=4)
    reads = synth.make_reads(anno, 5000, 8, 43, ont=True, micro_exons=3, xs_conflict_frac=0.02)
    sam, gtf = str(tmp_path / "o.sam"), str(tmp_path / "o.gtf")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    _compare(oracle, tmp_path, ["-l", "3"], sam, gtf, "ont")


def test_unsorted_bam(oracle, tmp_path):
    anno = synth.make_annotation(8000, 44, nchr=4)
    reads = synth.make_reads(anno, 8000, 5, 44, unsorted=True)
    sam, gtf = str(tmp_path / "u.sam"), str(tmp_path / "u.gtf")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    _compare(oracle, tmp_path, ["-l", "3"], sam, gtf, "uns")


def test_bam2gtf_and_unique_gtf(oracle, tmp_path, files):
    d, anno, reads, sam, bam, gtf = files
    for cmd, extra in (("bam2gtf", ["-e", "5"]), ("unique-gtf", []), ("unique-gtf", ["-I", "-s"])):
        a, b = str(tmp_path / "o.out"), str(tmp_path / "h.out")
        assert oracle.run_cli([cmd] + extra + [sam], stdout_path=a) == 0
        r = hostlib.run_cli([cmd] + extra + [bam], stdout_path=b)
        assert r.returncode == 0, r.stderr.decode()[-1000:]
        assert filecmp.cmp(a, b, shallow=False), (cmd, extra)
        assert os.path.getsize(a) > 10000
    # bam2gtf converts a BAM window by window (aln_reader.c h_aln_stream): windows of one BGZF block = many engine batches
    a, b = str(tmp_path / "o.out"), str(tmp_path / "hw.out")
    assert oracle.run_cli(["bam2gtf", sam], stdout_path=a) == 0
    r = hostlib.run_cli(["bam2gtf", bam], stdout_path=b, env={"L2R_READ_WINDOW": 65600})
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    assert filecmp.cmp(a, b, shallow=False)


def test_gtf_input_mode(oracle, tmp_path, files):
    """`update-gtf -m g -b`: the reads' bam2gtf output as read-like input; the engine gets synthesized M/N CIGARs."""
    d, anno, reads, sam, bam, gtf = files
    rgtf = str(tmp_path / "reads.gtf")
    assert oracle.run_cli(["bam2gtf", sam], stdout_path=rgtf) == 0
    _compare(oracle, tmp_path, ["-m", "g", "-b", sam, "-l", "3"], rgtf, gtf, "mg")          # (the oracle reads SAM headers only)
    _compare(oracle, tmp_path, ["-m", "g", "-b", sam, "-l", "5", "-d", "1"], rgtf, gtf, "mg2")


def _run_all_outputs(tmp_path, tag, extra, aln, gtf, env=None):
    o = _paths(tmp_path, tag)
    r = hostlib.run_cli(_args(extra, o, aln, gtf), env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return o


@pytest.mark.parametrize("kind", ["sorted", "unsorted", "unsorted_sj"])
def test_sharded_single_gpu_run_is_byte_identical(oracle, tmp_path, files, kind):
    """A single-GPU run cut into many engine shards (L2R_CHUNK_READS; the engine's own limit is 2^32 reads + ops per
    shard) writes the bytes of the unsharded run -- also for unsorted input, whose two sequential cursors
    (src/update_gtf.c:938) the engine carries from shard to shard."""
    d, anno, reads, sam, bam, gtf = files
    extra = ["-l", "3"]
    aln = bam
    if kind != "sorted":
        ur = synth.make_reads(anno, 30000, 5, 47, unsorted=True)
        aln = str(tmp_path / "u.sam")
        ur.write_sam(aln)
        if kind == "unsorted_sj":
            af = anno.in_file_order()
            base = util.oracle_run(oracle, af, ur, oracle.default_params(full_level=3))
            j, _ = util.junction_table(af, ur, base, 47, cover=0.6)
            tab = str(tmp_path / "SJ.out.tab")
            j.write(tab)
            extra = ["-s", "-l", "3", "-J", "1", "-j", tab]
    whole = _run_all_outputs(tmp_path, "whole", extra, aln, gtf)
    # unsorted input: also against the oracle (sequential cursors)
    if kind != "sorted":
        oo = _paths(tmp_path, "orc")
        assert oracle.run_cli(_args(extra, oo, aln, gtf)) == 0
        for k in OUTS:
            assert filecmp.cmp(oo[k], whole[k], shallow=False), (kind, "oracle", k)
    for chunk in (7001, 1500):
        part = _run_all_outputs(tmp_path, "c%d" % chunk, extra, aln, gtf, env={"L2R_CHUNK_READS": chunk})
        for k in OUTS:
            assert filecmp.cmp(whole[k], part[k], shallow=False), (kind, chunk, k)


@pytest.mark.parametrize("extra", [["-l", "3"], ["-l", "5", "-c"], "sj"])
def test_accepted_route_writes_the_same_files(oracle, tmp_path, files, extra):
    """`update-gtf ... > new.gtf` with at most -v / -E asks the engine for the compacted accepted list only
    (l2r_download_accepted, the pipeline's first pass, Snakefile:93); the files equal the ones of the full-result route
    and of the oracle.  Sharded too."""
    d, anno, reads, sam, bam, gtf = files
    if extra == "sj":
        af = anno.in_file_order()
        base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
        j, _ = util.junction_table(af, reads, base, 43, cover=0.7)
        tab = str(tmp_path / "SJ.out.tab")
        j.write(tab)
        extra = ["-s", "-l", "3", "-J", "1", "-j", tab]

    def run(tag, env, binary=True):
        o = {k: str(tmp_path / ("%s.%s" % (tag, k))) for k in ("gtf", "bed", "novel")}
        args = ["update-gtf"] + extra + ["-E", o["bed"], "-v", o["novel"], "-o", o["gtf"], bam if binary else sam, gtf]
        if binary:
            r = hostlib.run_cli(args, env=env)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
        else:
            assert oracle.run_cli(args) == 0
        return o

    want = run("orc", None, binary=False)
    for tag, env in (("acc", None), ("full", {"L2R_ROUTE": "full"}), ("acc_sharded", {"L2R_CHUNK_READS": 9000})):
        got = run(tag, env)
        for k in want:
            assert filecmp.cmp(want[k], got[k], shallow=False), (tag, k)
        assert os.path.getsize(got["gtf"]) > 1000


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.timeout(600)
@pytest.mark.parametrize("route", ["partitioned", "gathered_full", "gathered_accepted", "unsorted", "partitioned_in_pieces", "gathered_accepted_in_pieces"])
def test_ranks_on_the_engine_write_the_single_gpu_files(oracle, tmp_path, files, route):
    """`python -m lr2rmats_amd.dist` with the real engine behind every rank: three ranks share GPU 0 (gloo as the
    transport: RCCL wants one GPU per rank, the driver's 8-GPU run covers that); the shard results are read from the
    engines' HBM buffers (l2r_device_view_get), gathered, and the files equal the ones of the one-process CLI."""
    import subprocess
    import sys
    d, anno, reads, sam, bam, gtf = files
    aln = bam
    pieces = route.endswith("_in_pieces")               # a rank's shard beyond the engine's shard limit: several uploads per rank (dist.py)
    route = route.replace("_in_pieces", "")
    if route == "unsorted":
        aln = str(tmp_path / "u.bam")
        synth.write_bam(synth.make_reads(anno, 20000, 5, 45, unsorted=True), aln)
    if route == "gathered_accepted":
        names = ("gtf", "bed", "novel")
        args = lambda o: ["update-gtf", "-l", "3", "-E", o["bed"], "-v", o["novel"], "-o", o["gtf"], aln, gtf]
    else:
        names = OUTS
        args = lambda o: _args(["-l", "3"], o, aln, gtf)
    one = {k: str(tmp_path / ("one." + k)) for k in names}
    many = {k: str(tmp_path / ("many." + k)) for k in names}
    r = hostlib.run_cli(args(one))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    world, port, procs = 3, _free_port(), []
    trace = str(tmp_path / "trace")
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   L2R_DIST_BACKEND="gloo", L2R_DIST_TRACE=trace, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if route.startswith("gathered"):
            env["L2R_DIST_GATHER"] = "1"
        if pieces:
            env["L2R_CHUNK_READS"] = "1500"
        procs.append(subprocess.Popen([sys.executable, "-m", "lr2rmats_amd.dist"] + args(many), env=env, stderr=subprocess.PIPE,
                                      stdout=subprocess.DEVNULL, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    for p in procs:
        _, err = p.communicate(timeout=500)
        assert p.returncode == 0, err.decode()[-3000:]
    for k in names:
        assert filecmp.cmp(one[k], many[k], shallow=False), (route, k)
    want = {"partitioned": "partitioned", "gathered_full": "gathered full", "gathered_accepted": "gathered accepted",
            "unsorted": "one rank, gathered full"}[route]
    assert open(trace).read().strip() == want


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_gpus,stdout", [(3, False), (5, True)])
def test_c_cli_with_several_gpu_children_writes_the_single_gpu_files(tmp_path, files, n_gpus, stdout):
    """`L2R_GPUS=N lr2rmats update-gtf ...` (host/cmds.c update_gtf_multi): the C binary parses the inputs once, forks one child per
    device before any HIP call, every child classifies a chromosome-aligned shard and runs the tail on it, the parent joins the
    parts.  Here the children share GPU 0 (L2R_GPU_MAP); more children than chromosomes leaves some shards empty.  Every file
    equals the one-process run's, the updated GTF also when it goes to stdout."""
    d, anno, reads, sam, bam, gtf = files
    one, many = _paths(tmp_path, "one"), _paths(tmp_path, "many")
    r = hostlib.run_cli(_args(["-l", "3"], one, bam, gtf))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    env = {"L2R_GPUS": n_gpus, "L2R_GPU_MAP": ",".join(["0"] * n_gpus), "L2R_THREADS": 3, "L2R_TAIL_PART_READS": 500}
    args = _args(["-l", "3"], many, bam, gtf)
    if stdout:
        i = args.index("-o")
        del args[i:i + 2]
        r = hostlib.run_cli(args, stdout_path=many["gtf"], env=env)
    else:
        r = hostlib.run_cli(args, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    for k in OUTS:
        assert filecmp.cmp(one[k], many[k], shallow=False), k
    assert not [f for f in os.listdir(str(tmp_path)) if ".part" in f]
    # an option set that needs one stream of records says so and runs on one GPU
    r = hostlib.run_cli(["update-gtf", "-m", "g", "-b", sam, "-l", "3", gtf, gtf], env={"L2R_GPUS": 2, "L2R_GPU_MAP": "0,0"})
    assert r.returncode == 0 and b"running on one GPU" in r.stderr
    if n_gpus == 5:
        # more children than the input has chromosomes: some get no records -- said, not silent -- and the files are still the same
        eleven = _paths(tmp_path, "eleven")
        r = hostlib.run_cli(_args(["-l", "3"], eleven, bam, gtf), env={"L2R_GPUS": 11, "L2R_GPU_MAP": ",".join(["0"] * 11), "L2R_THREADS": 2})
        assert r.returncode == 0 and b"without records" in r.stderr, r.stderr.decode()[-2000:]
        for k in OUTS:
            assert filecmp.cmp(one[k], eleven[k], shallow=False), k
    # a child for a device the node does not have: refused before any child starts, nothing left behind
    bad = _paths(tmp_path, "bad")
    r = hostlib.run_cli(_args(["-l", "3"], bad, bam, gtf), env={"L2R_GPUS": 2, "L2R_GPU_MAP": "0,97"})
    assert r.returncode != 0 and b"would run on device 97" in r.stderr
    assert not [f for f in os.listdir(str(tmp_path)) if ".part" in f or f.startswith("l2r_gtf_")]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n_gpus,xchg", [(3, "shm"), (5, "shm"), (1, "rccl")])
def test_c_cli_gathered_route_over_several_children(oracle, tmp_path, files, n_gpus, xchg):
    """The pipeline's second command on several GPUs FROM C (Snakefile:170: `update-gtf -s -l 3 -J 1 -j SJ.out.tab ...`): -s with a
    junction table cannot be cut into independent shards (split pieces are compared across chromosomes, src/update_gtf.c:837-913, Q2),
    so `L2R_GPUS=N lr2rmats update-gtf` forks one child per device, the children classify shards cut ANYWHERE (not at chromosome
    boundaries), their per-read results are gathered on child 0 -- over RCCL from the engines' HBM (l2r_xchg_*: ncclSend / ncclRecv to
    rank 0), or through memory shared since before the fork where RCCL cannot run -- and child 0 runs the order-dependent tail once.
    Here: 3 and 5 children on GPU 0 with the shared-memory transport, and a world of ONE through real RCCL (id from rank 0,
    ncclCommInitRank, the counts' all-gather, the gather itself).  Every file equals the one-process run's."""
    d, anno, reads, sam, bam, gtf = files
    af = anno.in_file_order()
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
    j, _ = util.junction_table(af, reads, base, 41, cover=0.7)
    tab = str(tmp_path / "SJ.out.tab")
    j.write(tab)
    extra = ["-s", "-l", "3", "-J", "1", "-j", tab]
    one, many = _paths(tmp_path, "one"), _paths(tmp_path, "many")
    r = hostlib.run_cli(_args(extra, one, bam, gtf))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    env = {"L2R_GPUS": n_gpus, "L2R_GPU_MAP": ",".join(["0"] * n_gpus), "L2R_THREADS": 3, "L2R_XCHG": xchg}
    if n_gpus == 1:
        env["L2R_MULTI_ROUTE"] = "gathered"
    r = hostlib.run_cli(_args(extra, many, bam, gtf), env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"gathered route" in r.stderr and (b"RCCL" if xchg == "rccl" else b"shared memory") in r.stderr
    for k in OUTS:
        assert filecmp.cmp(one[k], many[k], shallow=False), (n_gpus, xchg, k)
    assert not [f for f in os.listdir(str(tmp_path)) if ".part" in f]
    if n_gpus == 3:
        # the same route when the updated GTF goes to stdout, and forced on an option set that could be partitioned
        so = _paths(tmp_path, "so")
        args = _args(["-l", "3"], so, bam, gtf)
        i = args.index("-o")
        del args[i:i + 2]
        r = hostlib.run_cli(args, stdout_path=so["gtf"], env=dict(env, L2R_MULTI_ROUTE="gathered"))
        assert r.returncode == 0 and b"gathered route" in r.stderr, r.stderr.decode()[-2000:]
        ref = _paths(tmp_path, "ref")
        r = hostlib.run_cli(_args(["-l", "3"], ref, bam, gtf))
        assert r.returncode == 0
        for k in OUTS:
            assert filecmp.cmp(ref[k], so[k], shallow=False), ("stdout", k)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n_gpus,xchg", [(3, "shm"), (4, "shm"), (1, "rccl")])
def test_c_cli_gathered_route_moves_the_accepted_reads_alone(oracle, tmp_path, files, n_gpus, xchg):
    """SURVEY.md 8(e)'s message from C: when no output of the run wants every read (`update-gtf -s -l 3 -J 1 -j SJ.out.tab -v novel.gtf
    -E novel_exon.bed -o updated.gtf`: no detail table, no known / unrecognised / all lists, no summary), the gathered route sends the
    compacted accepted-novel records alone to child 0 (l2r_xchg_gather_accepted over RCCL, or shared memory) and the tail runs on them
    (h_job_finish_accepted).  The three files equal the one-process run's; L2R_GATHER_ALL=1 sends the per-read results instead."""
    d, anno, reads, sam, bam, gtf = files
    af = anno.in_file_order()
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
    j, _ = util.junction_table(af, reads, base, 43, cover=0.7)
    tab = str(tmp_path / "SJ.out.tab")
    j.write(tab)

    def args(o):
        return ["update-gtf", "-s", "-l", "3", "-J", "1", "-j", tab, "-v", o["novel"], "-E", o["bed"], "-o", o["gtf"], bam, gtf]
    one, many, full = _paths(tmp_path, "one"), _paths(tmp_path, "many"), _paths(tmp_path, "full")
    r = hostlib.run_cli(args(one))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert os.path.getsize(one["gtf"]) > 10000 and os.path.getsize(one["bed"]) > 100
    env = {"L2R_GPUS": n_gpus, "L2R_GPU_MAP": ",".join(["0"] * n_gpus), "L2R_THREADS": 3, "L2R_XCHG": xchg}
    if n_gpus == 1:
        env["L2R_MULTI_ROUTE"] = "gathered"
    r = hostlib.run_cli(args(many), env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"gathered route" in r.stderr and b"the accepted reads alone" in r.stderr and (b"RCCL" if xchg == "rccl" else b"shared memory") in r.stderr
    for k in ("gtf", "novel", "bed"):
        assert filecmp.cmp(one[k], many[k], shallow=False), (n_gpus, xchg, k)
    r = hostlib.run_cli(args(full), env=dict(env, L2R_GATHER_ALL="1"))
    assert r.returncode == 0 and b"the per-read results" in r.stderr, r.stderr.decode()[-2000:]
    for k in ("gtf", "novel", "bed"):
        assert filecmp.cmp(one[k], full[k], shallow=False), ("all", n_gpus, xchg, k)


def _n_devices():
    import torch
    return torch.cuda.device_count()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("accepted_only", [False, True])
def test_c_cli_gathered_route_over_rccl_between_two_gpus(oracle, tmp_path, files, accepted_only):
    """The RCCL transport with PEERS (the other gathered-route tests reach it with a world of one, where no ncclSend / ncclRecv is issued):
    L2R_GPUS=2, a GPU per child, both forms of the message -- the per-read results (offsets at[k] * width, exon offsets re-based on rank 0)
    and the accepted reads alone (tile_rchunk gathered for order_accepted).  Needs two devices: skipped on the one-GPU boxes of the pool."""
    if _n_devices() < 2:
        pytest.skip("needs two GPUs")
    d, anno, reads, sam, bam, gtf = files
    af = anno.in_file_order()
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
    j, _ = util.junction_table(af, reads, base, 47, cover=0.7)
    tab = str(tmp_path / "SJ.out.tab")
    j.write(tab)
    if accepted_only:
        keys = ("gtf", "novel", "bed")

        def args(o):
            return ["update-gtf", "-s", "-l", "3", "-J", "1", "-j", tab, "-v", o["novel"], "-E", o["bed"], "-o", o["gtf"], bam, gtf]
    else:
        keys = OUTS

        def args(o):
            return _args(["-s", "-l", "3", "-J", "1", "-j", tab], o, bam, gtf)
    one, many = _paths(tmp_path, "one"), _paths(tmp_path, "many")
    r = hostlib.run_cli(args(one))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    r = hostlib.run_cli(args(many), env={"L2R_GPUS": 2, "L2R_THREADS": 3, "L2R_XCHG": "rccl"})
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"gathered route" in r.stderr and b"RCCL" in r.stderr
    assert b"RCCL communicator: rank 0 of 2" in r.stderr and b"RCCL communicator: rank 1 of 2" in r.stderr      # (what ncclCommInitRank saw)
    for k in keys:
        assert filecmp.cmp(one[k], many[k], shallow=False), (accepted_only, k)


def test_c_cli_gathered_shard_beyond_one_upload_runs_on_one_gpu(oracle, tmp_path, files):
    """A child of the gathered route classifies its shard in ONE upload; where that does not fit (here: L2R_CHUNK_READS below the shard size)
    the run is declined before anything is forked and takes the one-GPU path, upload by upload -- same files, and it says so."""
    d, anno, reads, sam, bam, gtf = files
    af = anno.in_file_order()
    base = util.oracle_run(oracle, af, reads, oracle.default_params(full_level=3))
    j, _ = util.junction_table(af, reads, base, 53, cover=0.7)
    tab = str(tmp_path / "SJ.out.tab")
    j.write(tab)
    extra = ["-s", "-l", "3", "-J", "1", "-j", tab]
    one, many = _paths(tmp_path, "one"), _paths(tmp_path, "many")
    r = hostlib.run_cli(_args(extra, one, bam, gtf))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    r = hostlib.run_cli(_args(extra, many, bam, gtf), env={"L2R_GPUS": 3, "L2R_GPU_MAP": "0,0,0", "L2R_XCHG": "shm", "L2R_CHUNK_READS": 1500})
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"would not fit one upload: running on one GPU" in r.stderr and b"gathered route" not in r.stderr
    for k in OUTS:
        assert filecmp.cmp(one[k], many[k], shallow=False), k
