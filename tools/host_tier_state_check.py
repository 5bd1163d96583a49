import os, sys, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np, torch
import host_tier_probe as htp
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0)
def leg(tag):
    r = htp.leg_odf("gqi", reps=3, dev=dev)
    print(tag, "e2e %.1f ms  h2d %.1f GB/s d2h %.1f GB/s" % (r["e2e_pcie_ms"], r.get("h2d_gbs_while_copying", 0), r.get("d2h_gbs_while_copying", 0)), flush=True)
leg("fresh")
# GPU memory churn: what bench.py does before its host-tier legs (tens of GB allocated and released through torch)
xs = [torch.empty(int(4e9), dtype=torch.uint8, device=dev) for _ in range(30)]
for x in xs: x.fill_(1)
torch.cuda.synchronize(); del xs; torch.cuda.empty_cache()
leg("after 120 GB of torch allocations + empty_cache")
# compute churn: the contraction kernel for 3 s (the chip at its power cap, hot)
bval, bvec = phantom.scheme_gqi()
dwi, _ = phantom.make_dwi_torch((140,140,140), bval, bvec, seed=3, device=dev)
plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, device=0)
mask = torch.ones(140**3, dtype=torch.uint8, device=dev)
out = fj.odf_rec_device(plan, dwi, mask)
t0 = time.time()
while time.time() - t0 < 4.0:
    for _ in range(50): fj.odf_rec_device(plan, dwi, mask, out=out)
    torch.cuda.synchronize()
del dwi, out
leg("right after 4 s of the GQI kernel at the power cap")
time.sleep(3)
leg("3 s later")
