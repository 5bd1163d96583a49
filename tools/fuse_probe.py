#!/usr/bin/env python3
"""Timing experiment: which part of the fused epilogue costs what (FIBERS_FUSE_SKIP masks, results are wrong on purpose)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0)
SHAPE = (140, 140, 140)
bval, bvec = phantom.scheme_gqi()
dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=3, device=dev)
nvox = dwi.shape[1]
mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25, device=0)
out = fj.odf_rec_device(plan, dwi, mask, normalize=False)
masks = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4, 8, 16, 31, -1]
res = {m: [] for m in masks}
for rnd in range(4):
    for m in masks:
        if m < 0:
            os.environ["FIBERS_ODF_UNFUSED"] = "1"; os.environ.pop("FIBERS_FUSE_SKIP", None)
        else:
            os.environ.pop("FIBERS_ODF_UNFUSED", None); os.environ["FIBERS_FUSE_SKIP"] = str(m)
        fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
        torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / 10 * 1e3)
for m in masks:
    print("skip %3d: step ms min %.3f med %.3f" % (m, min(res[m]), float(np.median(res[m]))))
