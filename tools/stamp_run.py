import os, sys, torch
sys.path.insert(0, "/root/repo")
os.environ.setdefault("FIBERS_ODF_PIPE", "1")
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0)
bval, bvec = phantom.scheme_gqi()
dwi, _ = phantom.make_dwi_torch((140, 140, 140), bval, bvec, seed=3, device=dev, noise_frac=0.1)
mask = torch.ones(dwi.shape[1], dtype=torch.uint8, device=dev)
plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25, device=0)
fj.odf_rec_device(plan, dwi, mask, normalize=False)
torch.cuda.synchronize()
