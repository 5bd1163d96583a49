#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 kernel trace: per kernel of the step its start offset, duration and the idle
gap since the previous kernel ended, averaged over the last N steps.  A step = the kernels from one launch of `first` to the next.
usage: step_timeline.py <..._kernel_trace.csv> [first-kernel-substring=mask_compact] [steps=20]"""
import csv
import re
import sys

path = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "mask_compact"
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
if len(starts) < 3:
    sys.exit("fewer than 3 steps in the trace")
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-nsteps:]
shape = None
acc = {}
tot = 0.0
for a, b in steps:
    names = [rows[i]["Kernel_Name"] for i in range(a, b)]
    if shape is None:
        shape = names
    if names != shape:
        continue
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = int(rows[a - 1]["End_Timestamp"]) if a > 0 else t0
    for j, i in enumerate(range(a, b)):
        s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        d = acc.setdefault(j, [0.0, 0.0, 0.0, 0])
        d[0] += (s - t0) / 1e3; d[1] += (e - s) / 1e3; d[2] += (s - prev_end) / 1e3; d[3] += 1
        prev_end = e
    tot += (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
n = acc[0][3]
print("steps averaged: %d; step period %.1f us" % (n, tot / n))
print("%-64s %10s %10s %10s" % ("kernel", "start us", "dur us", "gap us"))
sd = sg = 0.0
for j, name in enumerate(shape):
    nm = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", name)).split("(")[0][:64]
    d = acc[j]
    print("%-64s %10.1f %10.1f %10.1f" % (nm, d[0] / n, d[1] / n, d[2] / n))
    sd += d[1] / n; sg += d[2] / n
print("sum of durations %.1f us, sum of gaps %.1f us" % (sd, sg))
