#!/usr/bin/env python3
"""Turn the summary of tools/profile.sh (tools/pmc_csv_summary.py: per-kernel means of the rocprofv3 passes) into
profiles/pmc_traffic.json, which bench.py reports as roofline.traffic.

    tools/pmc_traffic.py <summary.json> <config> <reads> [out.json]

FETCH_SIZE and WRITE_SIZE come from separate `rocprofv3 --kernel-trace --pmc X` passes (they do not share a pass on
gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots") and are in KiB.  Correction applied as the guide prescribes: on
gfx950 FETCH_SIZE counts 128-byte requests of wide coalesced reads as 64 bytes, so it is doubled; WRITE_SIZE is taken
as is.  The kernels here mix 16-byte and 4-byte per-lane accesses, for which the guide calls the absolute value
uncalibrated: the raw counters are kept next to the corrected sum.
"""
import json
import os
import sys


def main():
    summary, config, reads = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")
    res = {"config": config, "reads": reads, "source": summary,
           "note": "profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate passes, mean per launch",
           "kernels": {}}
    for name, d in json.load(open(summary)).items():
        if "FETCH_SIZE" not in d and "WRITE_SIZE" not in d:
            continue
        short = name.split("<")[0].strip()
        # (several instances of one template -- k_tile's plain and WIDE one -- share the short name: the one launched most stands for it)
        if short in res["kernels"] and res["kernels"][short].get("calls", 0) >= d.get("calls", 0):
            continue
        res["kernels"][short] = {"calls": d.get("calls", 0), "fetch_size_kib_raw": round(d.get("FETCH_SIZE", 0.0), 1), "write_size_kib_raw": round(d.get("WRITE_SIZE", 0.0), 1),
                                 "hbm_bytes_per_launch": int((2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024),
                                 "avg_ns": d.get("avg_ns")}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
