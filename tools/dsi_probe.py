#!/usr/bin/env python3
"""DSI reconstruction on the 140^3 x 515-frame phantom: step and kernel times, checksum.  python tools/dsi_probe.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom

dev = torch.device("cuda", 0)
SHAPE = (140, 140, 140); nvox = 140 ** 3
bval, bvec = phantom.scheme_dsi()
dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=5, device=dev)
L = fj.lib()
for name, mask in (("ones", torch.ones(nvox, dtype=torch.uint8, device=dev)), ("ball", phantom.ball_mask_torch(SHAPE, dev))):
    plan = fj.OdfPlan("dsi", bval, bvec, fj.sphere_642, hann_width=32, device=0)
    out = fj.odf_rec_device(plan, dwi, mask)
    for _ in range(2): fj.odf_rec_device(plan, dwi, mask, out=out)
    torch.cuda.synchronize()
    L.fib_profile_enable(1); L.fib_profile_reset()
    t0 = time.perf_counter()
    for _ in range(5): fj.odf_rec_device(plan, dwi, mask, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    ms, n = C.c_double(0), C.c_int64(0)
    r = []
    for k in ("odf_gemm", "dsi_fold", "odf_peaks", "zero_dead"):
        L.fib_profile_get(k.encode(), C.byref(ms), C.byref(n)); r.append("%s %.3f" % (k, ms.value / max(n.value, 1)))
    print(name, "step %.3f ms  %.0f Mvox/s " % (dt * 1e3, int(mask.sum()) / dt / 1e6), " ".join(r),
          " sum(odf) %.9e sum(pdf) %.9e" % (float(out["odf"].double().sum()), float(out["pdf"].double().sum())), flush=True)
    plan.close()
