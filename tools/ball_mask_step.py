import os, sys, ctypes as C, time
sys.path.insert(0, os.getcwd())
import torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0); L = fj.lib()
rec = os.environ.get("KIND", "gqi")
bval, bvec = phantom.scheme_gqi() if rec == "gqi" else phantom.scheme_dsi()
dwi, _ = phantom.make_dwi_torch((140,140,140), bval, bvec, seed=3, device=dev)
mask = phantom.ball_mask_torch((140,140,140), dev)
kind = os.environ.get("MASK", "ball")
if kind == "ones":
    mask = torch.ones(140**3, dtype=torch.uint8, device=dev)
elif kind == "slab":                                   # as many voxels as the ball holds, as one contiguous run
    n = int(mask.sum().item()); mask = torch.zeros(140**3, dtype=torch.uint8, device=dev); mask[:n] = 1
elif kind == "rows":                                 # the ball's voxel count as full x-rows (no ragged row ends), every other row
    n = int(mask.sum().item()); m = torch.zeros(140*140, 140, dtype=torch.uint8, device=dev); m[: 2 * (n // 140) : 2] = 1; mask = m.reshape(-1)
elif kind.startswith("blocks"):                      # every other block of B voxels, offset by O voxels: MASK=blocks:B:O
    _, B, O = kind.split(":"); B = int(B); O = int(O)
    n = int(mask.sum().item()); idx = torch.arange(140**3, device=dev)
    mask = ((((idx - O) // B) % 2 == 0) & (idx >= O) & (idx < O + 2 * n)).to(torch.uint8)
mask = mask.reshape(-1).contiguous()
plan = fj.OdfPlan(rec, bval, bvec, fj.sphere_642, sigma=1.25, hann_width=32, device=0)
out = fj.odf_rec_device(plan, dwi, mask, normalize=True)
for _ in range(40): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
torch.cuda.synchronize(); L.fib_profile_enable(1); L.fib_profile_reset()
t0=time.perf_counter()
for _ in range(40): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)
torch.cuda.synchronize(); wall=(time.perf_counter()-t0)/40*1e3; L.fib_profile_enable(0)
r=[]
for nm in (b"mask_compact", b"odf_gemm", b"odf_post", b"qa_normalize"):
    ms,cnt=C.c_double(),C.c_int64(); L.fib_profile_get(nm,C.byref(ms),C.byref(cnt)); r.append("%s %.3f"%(nm.decode(), ms.value/max(cnt.value,1)))
print(os.path.basename(os.environ.get("FIBERS_HIP_LIB","libfibers_hip.so")), "%s mask (%d voxels) step %.3f ms |"%(kind, int(mask.sum().item()), wall), " | ".join(r))
