#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes holding SQ_ACTIVE_INST_VALU into profiles/valu.json: the vector
pipes' issue time per launch of each executor kernel, per workload.

    tools/make_valu.py <out.json> <workload>=<pmc dir> [<workload>=<pmc dir> ...]

SQ_ACTIVE_INST_VALU counts, summed over the chip's SIMDs, the cycles (in units of 4 clocks) during
which a vector ALU instruction was executing; x 4 gives SIMD-cycles, which bench.py divides by the
number of SIMDs and the clock to get the share of a launch spent issuing vector instructions."""
import collections
import csv
import glob
import json
import sys


def means(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "iak::" not in r["Kernel_Name"]:
                continue
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    out = sys.argv[1]
    res = {}
    for spec in sys.argv[2:]:
        workload, d = spec.split("=", 1)
        res[workload] = {k: {"SQ_ACTIVE_INST_VALU": round(v, 1), "valu_busy_simd_cycles_per_launch": round(4 * v, 1)}
                         for k, v in sorted(means(d, "SQ_ACTIVE_INST_VALU").items())}
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
