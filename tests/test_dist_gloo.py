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


@pytest.mark.timeout(300)
def test_two_ranks_equal_single_process(oracle, tmp_path):
    anno = synth.make_annotation(5000, 51, nchr=4, shuffle_within_gene=True)
    reads = synth.make_reads(anno, 5001, 5, 51)
    sam, gtf = str(tmp_path / "r.sam"), str(tmp_path / "a.gtf")
    reads.write_sam(sam)
    anno.write_gtf(gtf)
    single = {k: str(tmp_path / ("s." + k)) for k in ("gtf", "detail", "bed", "summary")}
    multi = {k: str(tmp_path / ("m." + k)) for k in ("gtf", "detail", "bed", "summary")}
    args = lambda o: ["update-gtf", "-l", "3", "-A", o["detail"], "-E", o["bed"], "-y", o["summary"], "-o", o["gtf"], sam, gtf]
    assert oracle.run_cli(args(single)) == 0
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", _WORKER % ROOT] + args(multi), env=env, stderr=subprocess.PIPE))
    for p in procs:
        _, err = p.communicate(timeout=280)
        assert p.returncode == 0, err.decode()[-3000:]
    for k in single:
        assert filecmp.cmp(single[k], multi[k], shallow=False), k
    assert os.path.getsize(single["detail"]) > 100000
