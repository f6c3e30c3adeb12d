#!/bin/bash
# tools/prof_stats.sh <out_dir> <reads> <config> [ablate values ...] -- on the GPU box: rocprofv3 kernel statistics of tools/run_once.py,
# once per L2R_ABLATE value (diagnostics).  Every step has a time limit; nothing reads from the terminal.
out=$1; reads=$2; cfg=$3; shift 3
mkdir -p "$out"; export TMPDIR=/tmp
for a in "${@:-0}"; do
  L2R_ABLATE=$a timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/a$a" -o "a$a" -- python3 tools/run_once.py "$reads" "$cfg" > "$out/a$a.log" 2>&1 < /dev/null
  echo "== ablate $a (rc $?)"
  f=$(find "$out/a$a" -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" "$out/a${a}_kernel_stats.csv"; head -9 "$f" | cut -d, -f1-4 | cut -c1-120; else tail -3 "$out/a$a.log"; fi
done
