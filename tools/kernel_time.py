#!/usr/bin/env python3
"""GQI step on the 140^3 x 270 phantom with whatever library FIBERS_HIP_LIB names: contraction-kernel time (hipEvents) and step wall
time at steady state.  Used to compare timing-only experiment builds (results of those are wrong by construction)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0); L = fj.lib()
kind = sys.argv[1] if len(sys.argv) > 1 else "gqi"
bval, bvec = phantom.scheme_gqi() if kind == "gqi" else phantom.scheme_dsi()
dwi, _ = phantom.make_dwi_torch((140, 140, 140), bval, bvec, seed=3, device=dev)
mask = torch.ones(140 ** 3, dtype=torch.uint8, device=dev)
plan = fj.OdfPlan(kind, bval, bvec, fj.sphere_642, sigma=1.25, hann_width=32, device=0)
out = fj.odf_rec_device(plan, dwi, mask, normalize=True)
for _ in range(60): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
torch.cuda.synchronize()
L.fib_profile_enable(1); L.fib_profile_reset()
n = 60
t0 = time.perf_counter()
for _ in range(n): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
L.fib_profile_enable(0)
ms, cnt = C.c_double(), C.c_int64()
L.fib_profile_get(b"odf_gemm", C.byref(ms), C.byref(cnt))
print("%-40s %s kernel %.3f ms  step %.3f ms" % (os.path.basename(os.environ.get("FIBERS_HIP_LIB", "libfibers_hip.so")), kind, ms.value / max(cnt.value, 1), wall), flush=True)
