#!/usr/bin/env python3
"""Step time of the GQI kernel selected by FIBERS_ODF_PIPE (and FIBERS_HIP_LIB) on the 140^3 phantom."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402

dev = torch.device("cuda", 0)
bval, bvec = phantom.scheme_gqi()
dwi, _ = phantom.make_dwi_torch((140, 140, 140), bval, bvec, seed=3, device=dev, noise_frac=0.1)
mask = torch.ones(dwi.shape[1], dtype=torch.uint8, device=dev)
plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25, device=0)
for _ in range(3):
    fj.odf_rec_device(plan, dwi, mask, normalize=False)
torch.cuda.synchronize()
res = []
for rep in range(int(os.environ.get("FIBERS_TIME_REPS", "3"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fj.odf_rec_device(plan, dwi, mask, normalize=False)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20)
import ctypes as C
from fibers_jl_amd import _lib
L = _lib.lib()
L.fib_profile_enable(1); L.fib_profile_reset()
for _ in range(10):
    fj.odf_rec_device(plan, dwi, mask, normalize=False)
torch.cuda.synchronize()
parts = []
for k in (b"mask_compact", b"odf_gemm", b"odf_peaks", b"odfmax_refine", b"zero_dead"):
    ms, n = C.c_double(), C.c_int64()
    L.fib_profile_get(k, C.byref(ms), C.byref(n))
    if n.value:
        parts.append("%s %.3f" % (k.decode(), ms.value / n.value))
L.fib_profile_enable(0)
print("   kernels (ms per launch):", ", ".join(parts))
print("%s pipe=%s: step ms %s" % (os.path.basename(os.environ.get("FIBERS_HIP_LIB", "default")), os.environ.get("FIBERS_ODF_PIPE", "0"), " ".join("%.3f" % r for r in res)))
