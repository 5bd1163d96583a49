#!/usr/bin/env python3
"""Print name / calls / average us / total ms from a rocprofv3 *_kernel_stats.csv (usage: kstats.py file.csv [n])"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in rows[:n]:
    name = re.sub(r"^void ", "", r["Name"])
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.split("(")[0][:56]
    print(f'{name:56s} calls={int(r["Calls"]):5d} avg={float(r["AverageNs"]) / 1e3:10.1f} us total={float(r["TotalDurationNs"]) / 1e6:9.2f} ms')
