#!/usr/bin/env python3
"""Print the iak:: rows of a rocprofv3 kernel_stats.csv: calls, average µs, short name."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "iak::" in r["Name"]:
            print(f'{int(r["Calls"]):5d} {float(r["AverageNs"])/1e3:10.1f} us  {r["Name"].split("(")[0].replace("void ", "")}')
