"""bench.py's launcher: `python3 bench.py --gpus N` (the shape of the driver's N = 1 command) starts one process per GPU itself.
CPU: the command it would start (--dry-launch).  GPU: a world of two ranks sharing device 0 (gloo as the transport, RCCL
wants a GPU per rank) must come back as ONE JSON line that says so."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    return env


def test_dry_launch_prints_the_launcher_command():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "7", "--warmup", "2", "--dry-launch"], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    cmd = json.loads(r.stdout.decode().strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    at = cmd.index(BENCH)
    assert cmd[at + 1:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"]          # the same arguments, without --dry-launch


def test_one_process_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, "--dry-launch"], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    assert json.loads(r.stdout.decode().strip().splitlines()[-1])["launch"] is None


def test_a_world_that_does_not_match_gpus_is_refused():
    env = _env(); env["WORLD_SIZE"] = "2"; env["RANK"] = "0"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 2


def test_fast_junction_table_is_the_generators(oracle):
    from lr2rmats_amd import synth
    anno = synth.make_annotation(12000, 11); af = anno.in_file_order()
    reads = synth.make_reads(anno, 6000, 5, 11)
    res = oracle.classify_soa(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, af.tx_tid, af.tx_start, af.tx_end, af.tx_rev,
                              af.tx_ex_off, af.ex_start, af.ex_end, params=oracle.default_params(full_level=3))
    a = synth.make_junctions(anno, res.ex_off, res.ex_start, res.ex_end, reads.tid, 3, cover=1.0)
    b = synth.make_junctions_fast(af, res.ex_off, res.ex_start, res.ex_end, reads.tid, 3, cover=1.0)
    assert len(a.don) == len(b.don) > 0
    assert np.array_equal(a.tid, b.tid) and np.array_equal(a.don, b.don) and np.array_equal(a.acc, b.acc)
    c = synth.make_junctions_fast(af, res.ex_off, res.ex_start, res.ex_end, reads.tid, 3, cover=0.8)
    assert 0.7 * len(a.don) < len(c.don) < 0.9 * len(a.don)
    key = (c.tid.astype(np.int64) << 56) | (c.don.astype(np.int64) << 25) | (c.acc - c.don)
    assert np.all(np.diff(key) > 0)                                                # sorted by (tid, don, acc), no duplicates


@pytest.mark.gpu
def test_self_launched_world_of_two_on_one_gpu():
    env = _env()
    env["L2R_BENCH_DEVICES"] = "0,0"; env["L2R_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "cfg2", "--reads", "60000", "--c-route-reads", "50000"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_world_size"] == 2 and d["config"]["collective_backend"] == "gloo"
    assert d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert sum(d["config"]["reads_per_rank"]) == d["config"]["total_reads"] == 60000
    # both routes in one line: the headline is partitioned (no collective in a step), the other one gathers the accepted list
    assert d["other_exchange"]["exchange"] == "gathered" and d["other_exchange"]["value"] > 0
    assert sum(d["other_exchange"]["exchange_bytes_per_rank_per_step"]) > 0
    # ... and behind the process group the C CLI's own multi-GPU run (L2R_GPUS=2: here two children on GPU 0, shared-memory transport): both forms
    # of the gathered route against the one-GPU run of the same command
    cr = d["c_route"]
    assert "error" not in cr, cr
    assert cr["children"] == 2 and cr["reads"] == 50000
    for form in ("per_read_results", "accepted_reads_alone"):
        assert cr[form]["files_identical"] is True and cr[form]["exchange"] == "shm" and cr[form]["rc"] == [0, 0], cr[form]


@pytest.mark.gpu
def test_one_gpu_line_has_the_contract_fields():
    """`bench.py` at N = 1 on a small configuration: ONE JSON line with the driver's fields, the roofline object (from the timed region) and the
    CPU baseline object, parity of the timed path against the oracle on the sample."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--config", "cfg2", "--reads", "60000", "--cpu-sample", "60000",
                        "--no-gencode", "--no-e2e"], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "headline_region", "warmup_effective_steps"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "int32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 60000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-2            # (ms_per_step is rounded to four digits)
    ro = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and 0 < ro["frac"] < 1
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["parity_on_sample"] is True
    assert d["second_pass"]["ms_per_step"] > 0 and d["with_accepted"]["ms_per_step"] > 0
