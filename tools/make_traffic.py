#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes over bench.py (one with FETCH_SIZE, one with WRITE_SIZE)
into profiles/traffic.json: HBM bytes per launch of each executor kernel.

    tools/make_traffic.py <fetch_dir> <write_dir> <batch> <out.json>

Corrections for gfx950: both counters are in KiB; FETCH_SIZE tallies the 128-byte requests of
wide streaming reads at 64 bytes, so it is doubled (checked here on the grayscale kernel, whose
reads are exactly 3 bytes per pixel: the corrected figure must land within 2 % of that).
WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import collections
import csv
import glob
import json
import sys

sys.path.insert(0, ".")
from bench import KERNEL_OF_GROUP, W, H  # noqa: E402


def means(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "iak::" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch_dir, write_dir, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, write = means(fetch_dir, "FETCH_SIZE"), means(write_dir, "WRITE_SIZE")
    px = batch * W * H
    res = {}
    for group, kern in KERNEL_OF_GROUP.items():
        if kern not in fetch or kern not in write:
            continue
        rd = fetch[kern] * 1024 * 2
        wr = write[kern] * 1024
        res[group] = {"kernel": kern, "batch": batch, "FETCH_SIZE_KiB_raw": round(fetch[kern], 1),
                      "WRITE_SIZE_KiB_raw": round(write[kern], 1), "read_bytes": int(rd), "write_bytes": int(wr),
                      "hbm_bytes_per_launch": int(rd + wr), "hbm_bytes_per_px": round((rd + wr) / px, 3)}
    g = res.get("gray+hist")
    if g:
        ratio = g["read_bytes"] / (3.0 * px)
        g["calibration"] = f"reads / (3 B/px) = {ratio:.4f}"
        if abs(ratio - 1) > 0.02:
            raise SystemExit(f"FETCH_SIZE calibration off: {ratio}")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
