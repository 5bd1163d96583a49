import os, sys, torch
sys.path.insert(0, "/root/repo")
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0)
shape = (40, 40, 40); nvox = 40 ** 3
bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=4, device=dev)
ball = phantom.ball_mask_torch(shape, dev, radius=17.3)
outs = {}
for mode in ("f32", "bf16x3"):
    os.environ["FIBERS_ODF_GEMM"] = mode
    plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642)
    for name, m in (("ones", torch.ones(nvox, dtype=torch.uint8, device=dev)), ("ball", ball)):
        o = fj.odf_rec_device(plan, dwi, m)
        outs[mode, name] = o["odf"].clone()
    plan.close()
for name in ("ones", "ball"):
    a, b = outs["f32", name], outs["bf16x3", name]
    d = (a - b).abs().max(0).values
    bad = torch.nonzero(d > 1.0).flatten()
    print(name, "bad voxels", bad.numel(), bad[:10].tolist())
    if bad.numel():
        v = int(bad[0]); print(" live", int(ball[v]), "f32", a[:4, v].tolist(), "bf16", b[:4, v].tolist(), "nz rows f32", int((a[:, v] != 0).sum()), "bf16", int((b[:, v] != 0).sum()))
