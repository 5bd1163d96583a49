import os, sys
sys.path.insert(0, os.getcwd())
# the switch this tool flips exists in the DIAGNOSTIC build only (csrc/common.h ab_env; make -C fibers.jl_amd/csrc stamp)
os.environ.setdefault("FIBERS_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fibers.jl_amd", "libfibers_hip_stamp.so"))
import numpy as np, torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0)
rng = np.random.default_rng(12345)
bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, device=0)
b5, g5 = phantom.scheme_dsi()
plan5 = fj.OdfPlan("dsi", b5, g5, fj.sphere_642, device=0)
def rec(plan, dwi, mask, unit):
    os.environ["FIBERS_ODF_LIST"] = unit
    o = fj.odf_rec_device(plan, dwi, mask)
    torch.cuda.synchronize()
    return {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in o.items()}
def same(a, b):
    for k in a:
        xs, ys = (a[k], b[k]) if isinstance(a[k], list) else ([a[k]], [b[k]])
        for x, y in zip(xs, ys):
            if not torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)):
                return k
    return None
bad = 0
for it in range(60):
    shape = tuple(int(4 * x) if it % 3 else int(x) for x in rng.integers(3, 24, 3))
    nvox = int(np.prod(shape))
    if nvox % 4:
        shape = (shape[0] * 4, shape[1], shape[2]); nvox *= 4
    kind = it % 4
    if kind == 0: m = (rng.random(nvox) < rng.uniform(0.01, 1.0))
    elif kind == 1:
        m = np.zeros(nvox, bool)
        for _ in range(int(rng.integers(1, 12))):
            a = int(rng.integers(0, nvox)); m[a:a + int(rng.integers(1, 400))] = True
    elif kind == 2: m = phantom.ball_mask_torch(shape, dev).reshape(-1).cpu().numpy().astype(bool)
    else: m = np.ones(nvox, bool); m[int(rng.integers(0, nvox))] = False
    mask = torch.from_numpy(m.astype(np.uint8)).to(dev)
    use5 = it % 5 == 0
    p, bv, gv = (plan5, b5, g5) if use5 else (plan, bval, bvec)
    dwi, _ = phantom.make_dwi_torch(shape, bv, gv, seed=100 + it, device=dev)
    a, b, c = rec(p, dwi, mask, "quads"), rec(p, dwi, mask, "octets"), rec(p, dwi, mask, "auto")
    k = same(a, b) or same(a, c)
    dead = (mask == 0)
    nz = float(a["odf"][:, dead].abs().max()) if bool(dead.any()) else 0.0
    if k or nz != 0.0:
        bad += 1; print("MISMATCH it", it, shape, "dsi" if use5 else "gqi", "key", k, "dead max", nz, flush=True)
print("soak done:", 60 - bad, "of 60 identical")
