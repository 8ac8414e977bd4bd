#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc csv: mean counter value per kernel name (iak kernels only)."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "iak::" not in k: continue
        k = k.split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
