#!/usr/bin/env python3
"""Static check of the odf_gemm_kernel variants' K loop (usage: check_gemm_isa.py odf_dev.s).
Flags s_waitcnt vmcnt(...) between a stage's first direct-to-LDS load and its MFMA block: such a wait exposes
the memory latency of every stage (hipcc inserts them when its wait-count bookkeeping is confused)."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r"^(_ZN12_GLOBAL__N_115odf_gemm_kernelILi(\d+)ELi(\d+)EEEvNS_8GemmArgsE):", s, re.M):
    name, mb, nx = m.group(1), int(m.group(2)), int(m.group(3))
    body = s[m.end():s.index(".Lfunc_end", m.end())].split("\n")
    # the K loop body: from the last s_barrier before the first MFMA back to ... simpler: walk instructions,
    # state machine: after an LDS-DMA inside a loop ("in Loop"/"Inner Loop" label seen) until the first MFMA
    seen_loop = False
    in_stage = False
    waits = []
    c = Counter()
    for l in body:
        if re.match(r"^\.LBB", l):
            if "Loop" in l:
                seen_loop = True
            continue
        op = re.match(r"^\s+([a-z_0-9]+)", l)
        if not op:
            continue
        op = op.group(1)
        if "mfma" in op:
            in_stage = False
            c["mfma"] += 1
        if not seen_loop:
            continue
        if op.startswith("global_load_lds") and c["mfma"] > 0 or (op.startswith("global_load_lds") and c["dma"] > 0):
            in_stage = True
        if op.startswith("global_load_lds"):
            c["dma"] += 1
        if in_stage and op == "s_waitcnt" and "vmcnt" in l:
            waits.append(l.split(";")[0].strip())
    flag = "BAD" if waits else "ok"
    bad += bool(waits)
    print(f"MB={mb:2d} NX={nx}: {flag} {waits[:4]}")
sys.exit(1 if bad else 0)
