#!/bin/bash
# tools/ab_libs.sh <reads> <config> <lib.so ...> -- diagnostics on the GPU box: tools/ab.py once per engine build, twice around
# (the builds alternate on one box, so clock drift and box-to-box spread hit them alike).  "cur" = the in-tree library.
reads=$1; cfg=$2; shift 2
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = cur ]; then unset L2R_HIP_LIB; else export L2R_HIP_LIB="$PWD/$lib"; fi
    echo "== $lib"; python3 tools/ab.py "$reads" "$cfg" ${AB_VALS:-0} 2>&1 | tail -n ${AB_TAIL:-1}
  done
done
