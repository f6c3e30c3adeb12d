#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_ab
for a in 0 12288; do
  L2R_ABLATE=$a L2R_ONCE_ITERS=5 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc_ab/a$a -o a$a -- python3 tools/run_once.py 10000000 cfg3 > gpurun_out/pmc_ab/a$a.log 2>&1 < /dev/null
  f=$(find gpurun_out/pmc_ab/a$a -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][-40:]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVES': cnt[k]+=1
for k,v in acc.items():
    if v.get('SQ_WAVES',0)>1e5*1: print(k, {c: round(x/max(1,v['SQ_WAVES']),1) for c,x in v.items() if c!='SQ_WAVES'}, 'launches', cnt[k])
PY
done
