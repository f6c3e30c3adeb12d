"""CPU: the per-rank BGZF block ranges of a coordinate-sorted BAM (host/aln_reader.c h_read_alignments_blocks) through the CLI's
`bam-shards` diagnostic: for every world size the ranks' ranges meet, together they hold every record once, and no rank inflates
much more than its share.  (What dist.py does with them: tests/test_dist_gloo.py.)"""
import re

import pytest

from lr2rmats_amd import hostlib, synth


@pytest.fixture(scope="module")
def bams(tmp_path_factory):
    d = tmp_path_factory.mktemp("shards")
    anno = synth.make_annotation(8000, 91, nchr=10, shuffle_within_gene=True)
    reads = synth.make_reads(anno, 90000, 5, 91)
    a, b = str(d / "slow.bam"), str(d / "fast.bam")
    synth.write_bam(reads.slice(0, 20000), a)          # (the two writers cut their BGZF blocks differently)
    synth.write_bam_fast(reads, b)
    return (a, 20000), (b, reads.n)


@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("world", [1, 2, 3, 7, 16])
def test_ranges_meet_and_cover_the_file(bams, which, world):
    path, n = bams[which]
    r = hostlib.run_cli(["bam-shards", path, str(world)])
    out = r.stdout.decode()
    assert r.returncode == 0, out + r.stderr.decode()[-2000:]
    assert out.strip().splitlines()[-1] == "ranges meet, %d records" % n
    rows = [tuple(int(x) for x in re.findall(r"-?\d+", l)) for l in out.splitlines() if l.startswith("rank")]
    assert len(rows) == world and sum(q[-1] for q in rows) == n
    fsz = rows[0][6]
    for q in rows:                                        # (rank, start block, offset, end block, offset, inflated, file size, records)
        assert q[5] <= fsz / world + fsz / 5 + 4 * 65536, q


def test_small_windows_and_threads(bams, monkeypatch):
    path, n = bams[1]
    monkeypatch.setenv("L2R_READ_WINDOW", "70000")
    monkeypatch.setenv("L2R_THREADS", "3")
    r = hostlib.run_cli(["bam-shards", path, "4"])
    assert r.returncode == 0 and r.stdout.decode().strip().splitlines()[-1] == "ranges meet, %d records" % n
