#!/bin/bash
# tools/profile_wait.sh <out_dir> -- latency-side SQ counters of the hot path: queue-depth integrals (LEVEL / INSTS = mean latency)
set -u
out=${1:-gpurun_out/profw}; reads=${2:-10000000}; cfg=${3:-cfg3}
mkdir -p "$out"; export TMPDIR=/tmp
run() { tag=$1; shift; rocprofv3 "$@" --output-format csv -d "$out/$tag" -o "$tag" -- python3 tools/run_once.py "$reads" "$cfg" > "$out/$tag.log" 2>&1; }
run w1 --kernel-trace --pmc SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY
run w2 --kernel-trace --pmc SQ_WAVES SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA
run w3 --kernel-trace --pmc SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU
python3 tools/pmc_csv_summary.py "$out" > "$out/summary.json"
python3 - "$out/summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if 'fused' in k or 'classify_fast' in k:
        print(k, json.dumps({c:round(x,1) for c,x in v.items()}, indent=0))
        if v.get('SQ_INSTS_VMEM_RD'): print(' mean VMEM latency (quad-cycles?)', v['SQ_INST_LEVEL_VMEM']/(v['SQ_INSTS_VMEM_RD']+v['SQ_INSTS_VMEM_WR']))
        if v.get('SQ_INSTS_LDS'): print(' mean LDS latency', v['SQ_INST_LEVEL_LDS']/v['SQ_INSTS_LDS'])
        if v.get('SQ_INSTS_SMEM'): print(' mean SMEM latency', v['SQ_INST_LEVEL_SMEM']/v['SQ_INSTS_SMEM'])
PY
