#!/usr/bin/env python3
"""Per-launch HBM traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md's HBM section prescribes: both counters are in KiB-free 'KB' units of 1024 B as rocprofv3 reports
them; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is DOUBLED; WRITE_SIZE is exact for 16-B/lane stores.

    python3 tools/pmc_traffic.py <dir with fetch/ and write/ subdirs> <kernel-name substring> [out.json]
"""
import csv
import glob
import json
import sys


def per_launch(d, kernel, counter):
    vals = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                # one row per (dispatch, [dimension instance]): sum the instances of a dispatch
                vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    if not vals:
        raise SystemExit("no %s rows for kernel *%s* under %s" % (counter, kernel, d))
    v = sorted(vals.values())
    v = v[len(v) // 8: len(v) - len(v) // 8] if len(v) >= 16 else v          # drop warm-up / outliers
    return sum(v) / len(v), len(vals)


def main():
    d, kernel = sys.argv[1], sys.argv[2]
    fetch_kb, n1 = per_launch(d + "/fetch", kernel, "FETCH_SIZE")
    write_kb, n2 = per_launch(d + "/write", kernel, "WRITE_SIZE")
    out = {"kernel": kernel, "launches": [n1, n2], "FETCH_SIZE_KB_raw": round(fetch_kb, 1), "WRITE_SIZE_KB": round(write_kb, 1),
           "fetch_bytes_corrected": int(2 * fetch_kb * 1024), "write_bytes": int(write_kb * 1024),
           "traffic_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
           "correction": "FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B), WRITE_SIZE as reported; KB = 1024 B"}
    print(json.dumps(out))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
