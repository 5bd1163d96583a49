#!/usr/bin/env python3
"""GQI step on z-slabs of the 140^3 x 270 volume (what one of N ranks gets): time per voxel against the slab's slice count.  A slab of an
odd number of 140 x 140 slices has a voxel count that is 16 mod 32: every other frame / output row of it starts 64 bytes off a cache line."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0); L = fj.lib()
bval, bvec = phantom.scheme_gqi()
full, _ = phantom.make_dwi_torch((140, 140, 140), bval, bvec, seed=3, device=dev)
plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25, device=0)
for nzs in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,17,18,35,36,70").split(",")]:
    n = nzs * 19600
    dwi = full[:, :n].contiguous()
    mask = torch.ones(n, dtype=torch.uint8, device=dev)
    out = fj.odf_rec_device(plan, dwi, mask, normalize=True)
    for _ in range(100): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
    torch.cuda.synchronize(); L.fib_profile_enable(1); L.fib_profile_reset()
    t0 = time.perf_counter()
    for _ in range(200): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 200 * 1e3; L.fib_profile_enable(0)
    ms, cnt = C.c_double(), C.c_int64(); L.fib_profile_get(b"odf_gemm", C.byref(ms), C.byref(cnt))
    k = ms.value / max(cnt.value, 1)
    print("%3d slices (%8d voxels, %% 32 = %2d): step %.3f ms, kernel %.3f ms = %.3f us per 1000 voxels" % (nzs, n, n % 32, wall, k, k * 1e6 / n), flush=True)
    del dwi, mask, out
