#!/usr/bin/env python3
"""A/B of the fused GQI kernel with and without anti-phase wave halves (FIBERS_ODF_ANTI): outputs bit-identical, step time."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the switch this tool flips exists in the DIAGNOSTIC build only (csrc/common.h ab_env; make -C fibers.jl_amd/csrc stamp)
os.environ.setdefault("FIBERS_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fibers.jl_amd", "libfibers_hip_stamp.so"))
import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402

dev = torch.device("cuda", 0)
bval, bvec = phantom.scheme_gqi()
shape = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (140, 140, 140)
dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=3, device=dev, noise_frac=0.1)
dwi[5, 100] = float("nan"); dwi[7, 2000] = float("inf"); dwi[:, 3000] = 0.0; dwi[:, 3001] = -1.0
mask = torch.ones(dwi.shape[1], dtype=torch.uint8, device=dev)
plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25, device=0)
res = {}
for flag in ("3", "0"):
    os.environ["FIBERS_ODF_ANTI"] = flag
    o = fj.odf_rec_device(plan, dwi, mask, normalize=False)
    torch.cuda.synchronize()
    res[flag] = [o["odf"].clone()] + [t.clone() for t in o["peak"]] + [t.clone() for t in o["qa"]] + [o["odfmax"].clone()]
nn = lambda t: torch.nan_to_num(t, nan=-7.0, posinf=-8.0, neginf=-9.0)
print("identical:", all(torch.equal(nn(x), nn(y)) for x, y in zip(res["3"], res["0"])))
for rep in range(2):
    for flag in ("0", "1", "2", "3"):
        os.environ["FIBERS_ODF_ANTI"] = flag
        for _ in range(3):
            fj.odf_rec_device(plan, dwi, mask, normalize=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fj.odf_rec_device(plan, dwi, mask, normalize=False)
        e1.record()
        torch.cuda.synchronize()
        print("anti=%s step ms %.3f" % (flag, e0.elapsed_time(e1) / 20))
