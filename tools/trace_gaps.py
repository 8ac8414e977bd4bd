#!/usr/bin/env python3
"""Per-step timeline of the bench from a rocprofv3 kernel trace: for the last steps, each kernel's
duration and the gap to the kernel before it.  usage: python tools/trace_gaps.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps start at a grayscale kernel
starts = [i for i, r in enumerate(rows) if "gray_kernel" in r[2]]
tail = starts[-40:-1]
acc = {}
gaps = {}
n = 0
for a, b in zip(tail[:-1], tail[1:]):
    n += 1
    for i in range(a, b):
        name = rows[i][2].split("(")[0][-60:]
        acc[name] = acc.get(name, 0) + rows[i][1] - rows[i][0]
        gaps[name] = gaps.get(name, 0) + rows[i][0] - rows[i - 1][1]
step = (rows[tail[-1]][0] - rows[tail[0]][0]) / (len(tail) - 1)
print(f"step {step / 1e3:.1f} us over {n} steps")
for k in acc:
    print(f"{k:62s} dur {acc[k] / n / 1e3:7.1f} us   gap before {gaps[k] / n / 1e3:6.1f} us")
print(f"sum of durations {sum(acc.values()) / n / 1e3:.1f} us, sum of gaps {sum(gaps.values()) / n / 1e3:.1f} us")
