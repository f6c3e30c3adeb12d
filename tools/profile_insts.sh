#!/bin/bash
# tools/profile_insts.sh <out_dir> <reads> <config> -- instruction mix and issue/wait cycles per kernel (SQ counters, two passes)
set -u
out=${1:-gpurun_out/profi}; reads=${2:-10000000}; cfg=${3:-cfg3}
mkdir -p "$out"; export TMPDIR=/tmp
run() { tag=$1; shift; rocprofv3 "$@" --output-format csv -d "$out/$tag" -o "$tag" -- python3 tools/run_once.py "$reads" "$cfg" > "$out/$tag.log" 2>&1; }
run i1 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
run i2 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run i3 --kernel-trace --pmc SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU
python3 tools/pmc_csv_summary.py "$out" > "$out/summary.json"
