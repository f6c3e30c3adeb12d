#!/bin/bash
# tools/calib.sh <out_dir> -- on the GPU box: the shader clock under the hot path (GRBM_GUI_ACTIVE cycles over the kernel durations)
# next to the VALU counters, so that VALU busy = SQ_ACTIVE_INST_VALU * 4 / SIMDs / cycles can be priced on the real clock.
set -u
out=${1:-gpurun_out/calib}; reads=${2:-10000000}; cfg=${3:-cfg3}
mkdir -p "$out"; export TMPDIR=/tmp
run() { tag=$1; shift; rocprofv3 "$@" --output-format csv -d "$out/$tag" -o "$tag" -- python3 tools/run_once.py "$reads" "$cfg" > "$out/$tag.log" 2>&1; }
run stats --kernel-trace --stats
run clk --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
python3 tools/pmc_csv_summary.py "$out" > "$out/summary.json"
python3 - "$out/summary.json" <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if v.get("SQ_WAVES", 0) > 10000 and "avg_ns" in v:
        t = v["avg_ns"] * 1e-9
        g = v.get("GRBM_GUI_ACTIVE", 0.0)
        print(k, "ms %.4f" % (t * 1e3), "GRBM cycles", g, "clock GHz %.3f" % (g / t / 1e9 if t else 0), "VALU/wave %.0f" % (v["SQ_INSTS_VALU"] / v["SQ_WAVES"]),
              "VALU busy (GRBM clock) %.3f" % (v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / g if g else 0), "SQ_BUSY/32/GRBM %.3f" % (v.get("SQ_BUSY_CYCLES", 0) / 32 / g if g else 0))
P
