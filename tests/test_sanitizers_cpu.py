"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer builds of the checker and of the host C code, run over the known-answer sets and
a batch of random small workloads (sanitizers on the host C and the restatement; GPU ASan does not exist here).

  * oracle/_build/lr2rmats_oracle_asan (make -C oracle asan): the toy set, every hand-derived case, 20 random workloads x option sets;
  * lr2rmats_amd/lib/libl2r_host_asan.so (make -C lr2rmats_amd/host asan-lib): the readers (SAM / BAM / GTF / junction table), the
    order-dependent tail and the writers, driven from a child process that feeds h_job_finish with oracle-made results -- the same
    harness as tests/test_host_tail_cpu.py, no GPU call is made.
A finding of either sanitizer ends the process (halt_on_error / -fno-sanitize-recover is not needed: any report fails the test)."""
import filecmp
import os
import subprocess
import sys

import numpy as np
import pytest

from lr2rmats_amd import synth
from tests import test_hand_known_answers as hand
from tests import test_host_tail_cpu as tail

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "toy")
ORACLE_ASAN = os.path.join(ROOT, "oracle", "_build", "lr2rmats_oracle_asan")
HOST_ASAN = os.path.join(ROOT, "lr2rmats_amd", "lib", "libl2r_host_asan.so")
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=1", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}
MARKS = (b"AddressSanitizer", b"runtime error:", b"LeakSanitizer", b"UndefinedBehaviorSanitizer")


@pytest.fixture(scope="module")
def oracle_asan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return ORACLE_ASAN


@pytest.fixture(scope="module")
def host_asan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "lr2rmats_amd", "host"), "asan-lib"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    pre = []
    for lib in ("libasan.so", "libubsan.so"):
        p = subprocess.run(["gcc", "-print-file-name=" + lib], stdout=subprocess.PIPE).stdout.decode().strip()
        assert os.path.isabs(p) and os.path.exists(p), "no %s beside gcc" % lib
        pre.append(os.path.realpath(p))
    env = dict(os.environ)
    env.update(SAN_ENV)
    env.update({"LD_PRELOAD": ":".join(pre), "L2R_HOST_LIB": HOST_ASAN})
    return env


def _clean(stderr: bytes, what):
    for m in MARKS:
        assert m not in stderr, "%s: sanitizer report\n%s" % (what, stderr.decode(errors="replace")[-3000:])


def _oracle(exe, args, stdout_path=None, expect_rc=0):
    env = dict(os.environ)
    env.update(SAN_ENV)
    out = open(stdout_path, "wb") if stdout_path else subprocess.DEVNULL
    try:
        r = subprocess.run([exe] + args, stdout=out, stderr=subprocess.PIPE, env=env)
    finally:
        if stdout_path:
            out.close()
    _clean(r.stderr, args)
    assert r.returncode == expect_rc, (args, r.returncode, r.stderr.decode(errors="replace")[-1000:])


def test_oracle_asan_toy_and_hand_cases(oracle_asan, tmp_path):
    for tag, extra in (("l3", ["-l", "3"]), ("l5", ["-l", "5"]), ("sup", ["-s", "-l", "3", "-J", "1", "-j", os.path.join(G, "sj_support.tab")]),
                       ("far", ["-s", "-l", "3", "-J", "1", "-j", os.path.join(G, "sj_far.tab")]),
                       ("oth", ["-s", "-l", "3", "-J", "1", "-j", os.path.join(G, "sj_other.tab")])):
        o = {k: str(tmp_path / (tag + "." + k)) for k in ("gtf", "detail", "summary", "bed")}
        _oracle(oracle_asan, ["update-gtf"] + extra + ["-A", o["detail"], "-y", o["summary"], "-E", o["bed"], os.path.join(G, "toy.sam"), os.path.join(G, "original.gtf")], o["gtf"])
    assert filecmp.cmp(str(tmp_path / "l3.detail"), os.path.join(G, "expect_l3.detail.txt"), shallow=False)
    _oracle(oracle_asan, ["bam2gtf", os.path.join(G, "toy.sam")], str(tmp_path / "b2g.gtf"))
    for case in sorted(hand.CASES):
        args, out, expect = hand._argv(case, tmp_path)
        _oracle(oracle_asan, args, out["gtf"])
        hand._check(out, expect, case)


def _random_case(seed, tmp_path):
    rng = np.random.default_rng([seed, 0x5a])
    tpg = int(rng.choice([1, 3, 5, 12, 40, 90]))
    n_ex = int(rng.integers(2, 12))
    ont = bool(rng.random() < 0.25)
    anno = synth.make_annotation(int(rng.choice([1500, 6000])), seed, mean_tx_exons=n_ex + 1, tx_per_gene=tpg, shuffle_within_gene=bool(seed & 1),
                                 long_tx_per_chrom=int(seed % 3 == 0))
    reads = synth.make_reads(anno, int(rng.choice([400, 2500])), n_ex, seed + 7, ont=ont, micro_exons=2 if ont else 0, xs_conflict_frac=0.05 if ont else 0.0)
    sam, gtf = str(tmp_path / ("s%d.sam" % seed)), str(tmp_path / ("s%d.gtf" % seed))
    reads.write_sam(sam, sort_order="coordinate")
    anno.write_gtf(gtf)
    opts = ["-l", str(int(rng.integers(1, 6))), "-d", str(int(rng.choice([0, 0, 2, 7]))), "-e", str(int(rng.choice([3, 1, 25]))),
            "-i", str(int(rng.choice([3, 40]))), "-t", str(int(rng.choice([50, 4])))]
    if seed % 4 == 1:
        opts += ["-c"]
    return sam, gtf, opts


@pytest.mark.parametrize("block", range(4))
def test_oracle_asan_random_workloads(oracle_asan, tmp_path, block):
    for seed in range(100 + 5 * block, 105 + 5 * block):
        sam, gtf, opts = _random_case(seed, tmp_path)
        o = {k: str(tmp_path / ("r%d.%s" % (seed, k))) for k in ("gtf", "detail", "summary", "bed", "known", "novel", "unrec", "all")}
        _oracle(oracle_asan, ["update-gtf"] + opts + ["-A", o["detail"], "-y", o["summary"], "-E", o["bed"], "-k", o["known"], "-v", o["novel"], "-u", o["unrec"],
                              "-a", o["all"], sam, gtf], o["gtf"])
        _oracle(oracle_asan, ["unique-gtf", sam], str(tmp_path / ("r%d.uniq" % seed)))
        assert os.path.getsize(o["detail"]) > 100


def _host(env, argv):
    root = ROOT
    r = subprocess.run([sys.executable, "-c", tail._CHILD % root] + argv, stderr=subprocess.PIPE, env=env)
    _clean(r.stderr, argv)
    return r.returncode, r.stderr


def test_host_asan_readers_tail_and_writers(oracle, host_asan, tmp_path):
    """libl2r_host_asan.so: SAM, BAM and GTF readers, junction table, routing / split / merge, writers (one and several threads) -- the
    files equal the oracle CLI's byte for byte, and neither sanitizer has anything to say."""
    # (the child really runs the sanitized library under the sanitizers' runtimes)
    probe = ("import sys; sys.path.insert(0, %r); from lr2rmats_amd import hostlib; hostlib.load_library(); m = open('/proc/self/maps').read(); "
             "print(hostlib.LIB_PATH); print('libasan' in m, 'libubsan' in m, 'libl2r_host_asan.so' in m)") % ROOT
    r = subprocess.run([sys.executable, "-c", probe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=host_asan)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-1500:]
    assert r.stdout.decode().split("\n")[:2] == [HOST_ASAN, "True True True"], r.stdout
    # toy set with every output, a junction table and -s
    for tag, extra in (("a", ["-l", "3"]), ("b", ["-s", "-l", "3", "-J", "1", "-j", os.path.join(G, "sj_other.tab")])):
        oo, ho = tail._paths(tmp_path, tag + ".o"), tail._paths(tmp_path, tag + ".h")
        assert oracle.run_cli(tail._args(extra, oo, os.path.join(G, "toy.sam"), os.path.join(G, "original.gtf"))) == 0
        rc, err = _host(host_asan, tail._args(extra, ho, os.path.join(G, "toy.sam"), os.path.join(G, "original.gtf")))
        assert rc == 0, err.decode(errors="replace")[-1500:]
        for k in tail.OUTS:
            assert filecmp.cmp(oo[k], ho[k], shallow=False), (tag, k)
    # the hand-derived reader-quirk annotation (rows cut behind byte 1 023, an empty line, a comment, a chromosome outside the header)
    H = hand.H
    oo, ho = tail._paths(tmp_path, "q.o"), tail._paths(tmp_path, "q.h")
    rc, err = _host(host_asan, tail._args(["-l", "5"], ho, os.path.join(H, "gtfq.sam"), os.path.join(H, "gtfq_anno.gtf")))
    assert rc == 0, err.decode(errors="replace")[-1500:]
    assert filecmp.cmp(ho["gtf"], os.path.join(H, "gtfq.updated.gtf"), shallow=False) and filecmp.cmp(ho["detail"], os.path.join(H, "gtfq.detail.txt"), shallow=False)
    # random workloads: SAM and BAM input, the tail on several threads
    for seed in (201, 202, 203, 204, 205, 206):
        sam, gtf, opts = _random_case(seed, tmp_path)
        inp = sam
        if seed & 1:
            inp = str(tmp_path / ("s%d.bam" % seed))
            env_py = [sys.executable, "-c", "import sys; sys.path.insert(0, %r); from lr2rmats_amd import hostlib; sys.exit(hostlib.load_library().h_records_to_bam(sys.argv[1].encode(), sys.argv[2].encode()))" % ROOT, sam, inp]
            r = subprocess.run(env_py, stderr=subprocess.PIPE, env=host_asan)
            _clean(r.stderr, "h_records_to_bam")
            assert r.returncode == 0, r.stderr.decode(errors="replace")[-1500:]
        oo, ho = tail._paths(tmp_path, "r%d.o" % seed), tail._paths(tmp_path, "r%d.h" % seed)
        assert oracle.run_cli(tail._args(opts, oo, sam, gtf)) == 0
        env = dict(host_asan)
        if seed % 3 == 0:
            env["L2R_THREADS"] = "3"; env["L2R_TAIL_PART_READS"] = "300"
        rc, err = _host(env, tail._args(opts, ho, inp, gtf))
        assert rc == 0, err.decode(errors="replace")[-1500:]
        for k in tail.OUTS:
            assert filecmp.cmp(oo[k], ho[k], shallow=False), (seed, k)
