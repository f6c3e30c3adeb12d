#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass on gfx950,
MI355X_MICROARCH.md "rocprofv3 PMC slots") into profiles/pmc_traffic.json, which bench.py reports as
roofline.traffic.

    tools/pmc_traffic.py <fetch_dir> <write_dir> <config> <reads> [out.json]

Both directories hold `*_counter_collection.csv` of `rocprofv3 --kernel-trace --pmc X --output-format csv`.
FETCH_SIZE / WRITE_SIZE are in KiB.  Correction applied as the guide prescribes: on gfx950 FETCH_SIZE counts
128-byte requests of wide coalesced reads as 64 bytes, so it is doubled; WRITE_SIZE is taken as is.  The kernels
here mix 16-byte and 4-byte per-lane accesses, for which the guide calls the absolute value uncalibrated: the
raw counters are kept next to the corrected sum.
"""
import collections
import csv
import glob
import json
import os
import sys

STAGE_OF = {"k_pass_a": "pass_a", "k_classify_fast": "classify_fast", "k_classify_generic": "classify_generic",
            "k_gather_accepted": "gather_accepted", "k_scan_u32": "scan", "k_validate_sj": "validate_sj",
            "k_count_accepted": "count_accepted"}


def per_kernel(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under %s" % d)
    per_dispatch = collections.defaultdict(float)
    name = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = (f, row["Dispatch_Id"])
            per_dispatch[k] += float(row["Counter_Value"])
            name[k] = row["Kernel_Name"]
    agg = collections.defaultdict(list)
    for k, v in per_dispatch.items():
        agg[name[k]].append(v)
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch_dir, write_dir, config, reads = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for kname, fkib in fetch.items():
        short = kname.split("(")[0].split("::")[-1].split("<")[0]
        stage = STAGE_OF.get(short)
        if not stage:
            continue
        wkib = write.get(kname, 0.0)
        res[stage] = {"config": config, "reads": reads, "kernel": short,
                      "fetch_size_kib_raw": round(fkib, 1), "write_size_kib_raw": round(wkib, 1),
                      "hbm_bytes_per_launch": int((2.0 * fkib + wkib) * 1024),
                      "note": "profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate passes"}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
