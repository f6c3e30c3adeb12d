#!/usr/bin/env python3
"""tools/make_seed_hashes.py -- writes tests/golden/seeds.sha256: sha256 of the three graded files (updated.gtf, detail.txt,
novel_exon.bed) of `update-gtf` on BASELINE configs[1] (100 k reads x 5 exons, 50 k-exon GTF) for generator seeds 1-5 and both
option sets of SURVEY.md 8(d), as produced by the CPU oracle's CLI (oracle/).  The inputs are regenerated from the seeds by
tests/util.seed_inputs; tests/test_seed_hashes.py checks the oracle (CPU) and the HIP CLI (GPU) against this file."""
import os
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import pyoracle as po          # noqa: E402
from tests import util                     # noqa: E402


def main():
    po.build()
    lines = ["# sha256 of the oracle CLI's outputs; name = seed<generator seed>.<option set>.<file>  (tools/make_seed_hashes.py)"]
    for seed in util.SEEDS:
        with tempfile.TemporaryDirectory(prefix="l2r_seed_") as d:
            sam, gtf, tab = util.seed_inputs(po, seed, d)
            for which in util.SEED_SETS:
                out = {k: os.path.join(d, which + "." + k) for k in util.SEED_FILES + ("summary.txt",)}
                rc = po.run_cli(util.seed_args(which, sam, gtf, tab, out))
                assert rc == 0, (seed, which, rc)
                for k in util.SEED_FILES:
                    lines.append("%s  seed%d.%s.%s" % (util.sha256_file(out[k]), seed, which, k))
                    print(lines[-1], os.path.getsize(out[k]))
    with open(os.path.join(ROOT, "tests", "golden", "seeds.sha256"), "w") as fh:
        fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
