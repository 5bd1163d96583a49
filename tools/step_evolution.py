import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0)
bval, bvec = phantom.scheme_gqi()
dwi, _ = phantom.make_dwi_torch((140,140,140), bval, bvec, seed=3, device=dev)
mask = torch.ones(140**3, dtype=torch.uint8, device=dev)
plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25, device=0)
out = fj.odf_rec_device(plan, dwi, mask, normalize=False)
torch.cuda.synchronize()
time.sleep(1.0)
ts = []
for blk in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5 * 1e3)
print("ms/step per block of 5 steps:", " ".join("%.3f" % t for t in ts))
