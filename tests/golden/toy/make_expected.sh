#!/bin/sh
# Regenerates expect_* from the oracle (run inside tests/golden/toy after `make -C oracle`).
set -e
O=../../../oracle/_build/lr2rmats_oracle
$O update-gtf -l 3 -A expect_l3.detail.txt -y expect_l3.summary.txt -E expect_l3.novel_exon.bed toy.sam original.gtf > expect_l3.updated.gtf
for t in sj_support sj_far sj_other; do
  $O update-gtf -s -l 3 -J 1 -j $t.tab -A expect_$t.detail.txt -y expect_$t.summary.txt -E expect_$t.novel_exon.bed toy.sam original.gtf > expect_$t.updated.gtf
done
$O bam2gtf toy.sam > expect.bam2gtf.gtf
