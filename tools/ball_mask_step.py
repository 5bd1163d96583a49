import os, sys, ctypes as C, time
sys.path.insert(0, os.getcwd())
import torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0); L = fj.lib()
bval, bvec = phantom.scheme_gqi()
dwi, _ = phantom.make_dwi_torch((140,140,140), bval, bvec, seed=3, device=dev)
mask = phantom.ball_mask_torch((140,140,140), dev)
plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25, device=0)
out = fj.odf_rec_device(plan, dwi, mask, normalize=True)
for _ in range(40): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
torch.cuda.synchronize(); L.fib_profile_enable(1); L.fib_profile_reset()
t0=time.perf_counter()
for _ in range(40): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
torch.cuda.synchronize(); wall=(time.perf_counter()-t0)/40*1e3; L.fib_profile_enable(0)
r=[]
for nm in (b"mask_compact", b"odf_gemm", b"odf_post", b"qa_normalize"):
    ms,cnt=C.c_double(),C.c_int64(); L.fib_profile_get(nm,C.byref(ms),C.byref(cnt)); r.append("%s %.3f"%(nm.decode(), ms.value/max(cnt.value,1)))
print(os.path.basename(os.environ.get("FIBERS_HIP_LIB","libfibers_hip.so")), "ball mask step %.3f ms |"%wall, " | ".join(r))
