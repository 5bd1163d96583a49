#!/usr/bin/env python3
"""Compact view of the split-bf16 GEMM's stage loop in the device assembly (hipcc -S --cuda-device-only).
usage: gemm3_isa.py odf_dev.s [MB NX NW]"""
import re, sys
s = open(sys.argv[1]).read()
mb, nx, nw = (sys.argv[2:5] + ["10", "1", "4"])[:3] if len(sys.argv) > 2 else ("10", "1", "4")
m = re.search(r"^_ZN12_GLOBAL__N_116odf_gemm3_kernelILi%sELi%sELi%sELb0EEEvNS_8GemmArgsE:" % (mb, nx, nw), s, re.M)
end = s.index(".Lfunc_end", m.end())
body = s[m.end():end].split("\n")
tail = s[end:end + 6000]
g = lambda k: re.search(k + r"[:\s]+(\d+)", tail).group(1)
print("vgpr", g("NumVgprs"), "spill", g("ScratchSize"), "occ", g("Occupancy"), "lds", g("LDSByteSize"))
# the stage loop = the basic block with the most MFMAs
blocks, cur = [], []
for l in body:
    if re.match(r"^\.LBB", l):
        blocks.append(cur); cur = []
    cur.append(l)
blocks.append(cur)
blk = max(blocks, key=lambda b: sum("v_mfma" in x for x in b))
out = []
for l in blk:
    mm = re.match(r"^\s+([a-z_0-9]+)\s*(.*)", l)
    if not mm: continue
    op = mm.group(1)
    if op.startswith("v_mfma"): op = "MFMA"
    elif op.startswith("s_waitcnt"): op = "W[" + mm.group(2).split(";")[0].strip() + "]"
    elif op.startswith("s_barrier"): op = "BARRIER"
    elif op.startswith("s_"): op = "s"
    elif op.startswith("ds_read"): op = "DSR"
    elif op.startswith("buffer_load"): op = "BUF"
    elif op.startswith("global_load_lds"): op = "GLDS"
    elif op.startswith("scratch"): op = "SCRATCH"
    elif op.startswith("v_"): op = "v"
    out.append(op)
res, prev, cnt = [], None, 0
for o in out:
    if o == prev: cnt += 1
    else:
        if prev: res.append(f"{prev}{cnt}" if cnt > 1 else prev)
        prev, cnt = o, 1
res.append(prev)
print(" ".join(res))
