#!/usr/bin/env python3
"""Where a work item's cycles go in the fused contraction kernel: the DIAGNOSTIC build (make -C fibers.jl_amd/csrc phase) marks the
phases of the third work item of waves 0 (an "early" wave: MFMA block first, next split afterwards) and 4 (a "late" wave) of one
workgroup with s_memtime (shader cycles at the constant 100 MHz x clock ratio ... s_memtime counts shader-clock cycles).
Prints per stage: split | requests | MFMA block | second split | wait for loads | barrier, then the epilogue.
usage: python tools/phase_profile.py [gqi|dsi]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FIBERS_HIP_LIB", os.path.join(ROOT, "fibers.jl_amd", "libfibers_hip_phase.so"))
import numpy as np, torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
kind = sys.argv[1] if len(sys.argv) > 1 else "gqi"
dev = torch.device("cuda", 0); L = fj.lib()
bval, bvec = phantom.scheme_gqi() if kind == "gqi" else phantom.scheme_dsi()
dwi, _ = phantom.make_dwi_torch((140, 140, 140), bval, bvec, seed=3, device=dev)
mask = torch.ones(140 ** 3, dtype=torch.uint8, device=dev)
plan = fj.OdfPlan(kind, bval, bvec, fj.sphere_642, sigma=1.25, hann_width=32, device=0)
out = fj.odf_rec_device(plan, dwi, mask, normalize=True)
for _ in range(200): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
L.fib_debug_phase_stamps.argtypes = [C.c_void_p]; L.fib_debug_phase_stamps.restype = C.c_int
assert L.fib_debug_phase_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(2, 256)
names = {1: "top", 2: "split", 3: "requests", 4: "mfma", 5: "split2", 6: "loads", 7: "barrier", 8: "epilogue"}
for w, label in ((0, "wave 0 (early)"), (1, "wave 4 (late)")):
    t = (st[w] >> np.uint64(8)).astype(np.int64); ids = (st[w] & np.uint64(255)).astype(int)
    n = int((ids > 0).sum())
    print("%s: %d marks" % (label, n))
    tot = {}
    rows = []
    cur = {}
    for i in range(1, n):
        d = int(t[i] - t[i - 1]); k = ids[i]
        if ids[i] == 1:                                   # a new stage starts: what lies before it is not part of a phase
            rows.append(cur); cur = {}
            continue
        cur[names.get(k, str(k))] = d
        tot[names.get(k, str(k))] = tot.get(names.get(k, str(k)), 0) + d
    rows.append(cur)
    for i, r in enumerate(rows[:20]):
        print("  stage %2d: " % i + "  ".join("%s %5d" % (k, r[k]) for k in ("split", "requests", "mfma", "split2", "loads", "barrier", "epilogue") if k in r))
    print("  totals (s_memtime ticks): " + "  ".join("%s %d" % (k, v) for k, v in tot.items()), " item span %d" % int(t[n - 1] - t[0]))
