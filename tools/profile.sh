#!/bin/bash
# tools/profile.sh <out_dir> [reads] [config] -- on the GPU box: rocprofv3 kernel statistics and the PMC passes of the hot
# path (tools/run_once.py), one pass per counter group (gfx950 SQ has 8 slots, FETCH_SIZE and WRITE_SIZE do not share a
# pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Counters are collected with --kernel-trace only.
set -u
out=${1:-gpurun_out/prof}; reads=${2:-10000000}; cfg=${3:-cfg3}
mkdir -p "$out"; export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d "$out/$tag" -o "$tag" -- python3 tools/run_once.py "$reads" "$cfg" > "$out/$tag.log" 2>&1 < /dev/null; }
run stats --kernel-trace --stats
run sq1 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
run sq2 --kernel-trace --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
python3 tools/pmc_csv_summary.py "$out" > "$out/summary.json"
cat "$out/summary.json"
