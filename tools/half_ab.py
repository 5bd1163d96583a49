#!/usr/bin/env python3
"""A/B of the fused GQI kernel: two independent 4-wave workgroups per CU on half-stage ring buffers (default) against one 8-wave
workgroup per CU (FIBERS_ODF_HALF=0), interleaved rounds in one process; kernel time, step time, agreement of the outputs."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0); L = fj.lib()
shape = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "140,140,140").split(","))
nvox = shape[0] * shape[1] * shape[2]
bval, bvec = phantom.scheme_gqi()
dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=3, device=dev)
mask = torch.ones(nvox, dtype=torch.uint8, device=dev) if len(sys.argv) <= 3 else phantom.ball_mask_torch(shape, dev)
plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25, device=0)
outs = {}
def run(half, n=40):
    if half: os.environ.pop("FIBERS_ODF_HALF", None)
    else: os.environ["FIBERS_ODF_HALF"] = "0"
    o = outs.setdefault(half, fj.odf_rec_device(plan, dwi, mask, normalize=True))
    for _ in range(10): fj.odf_rec_device(plan, dwi, mask, out=o, normalize=True)
    torch.cuda.synchronize()
    L.fib_profile_enable(1); L.fib_profile_reset()
    t0 = time.perf_counter()
    for _ in range(n): fj.odf_rec_device(plan, dwi, mask, out=o, normalize=True)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    L.fib_profile_enable(0)
    ms, cnt = C.c_double(), C.c_int64()
    L.fib_profile_get(b"odf_gemm", C.byref(ms), C.byref(cnt))
    return ms.value / cnt.value, wall
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    k8, s8 = run(False); k4, s4 = run(True)
    print("round %d: 8-wave WG kernel %.3f ms step %.3f | 2 x 4-wave WG kernel %.3f ms step %.3f | ratio %.3f" % (r, k8, s8, k4, s4, k8 / k4), flush=True)
a, b = outs[True], outs[False]
nn = lambda t: torch.nan_to_num(t, nan=-7.0, posinf=-8.0, neginf=-9.0)
print("odf identical", torch.equal(nn(a["odf"]), nn(b["odf"])), "peaks identical", all(torch.equal(a["peak"][k], b["peak"][k]) for k in range(3)),
      "qa identical", all(torch.equal(nn(a["qa"][k]), nn(b["qa"][k])) for k in range(3)), "odfmax", a["odfmax"].tolist(), b["odfmax"].tolist())
