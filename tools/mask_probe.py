import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0)
SHAPE = (140, 140, 140); nvox = 140 ** 3
bval, bvec = phantom.scheme_gqi()
dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=3, device=dev)
L = fj.lib()
for name, mask in (("ones", torch.ones(nvox, dtype=torch.uint8, device=dev)), ("ball", phantom.ball_mask_torch(SHAPE, dev))):
    plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, device=0)
    out = fj.odf_rec_device(plan, dwi, mask, normalize=False)
    for _ in range(3): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=False)
    torch.cuda.synchronize()
    L.fib_profile_enable(1); L.fib_profile_reset()
    import time
    t0 = time.perf_counter()
    for _ in range(10): fj.odf_rec_device(plan, dwi, mask, out=out, normalize=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    ms, n = C.c_double(0), C.c_int64(0)
    r = []
    for k in ("odf_gemm", "odf_peaks", "zero_dead", "mask_compact"):
        L.fib_profile_get(k.encode(), C.byref(ms), C.byref(n)); r.append("%s %.3f" % (k, ms.value / max(n.value, 1)))
    print(name, "live voxels", int(mask.sum()), "step %.3f ms" % (dt * 1e3), " ".join(r), flush=True)
