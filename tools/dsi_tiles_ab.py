"""DSI at 140^3 x 515: odf_dsi2_kernel (two M tiles, peaks on chip) against the three-tile path + separate peak kernel (FIB_ODF_SEPARATE_PEAKS): step and kernel times, agreement of the outputs."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0)
L = fj.lib()
shape = (140, 140, 140)
nvox = 140**3
b5, g5 = phantom.scheme_dsi()
d5, _ = phantom.make_dwi_torch(shape, b5, g5, seed=5, device=dev)
mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
res = {}
for name, sep in (("two_tiles", False), ("three_tiles", True)):
    p5 = fj.OdfPlan("dsi", b5, g5, fj.sphere_642, hann_width=32, device=0)
    o5 = fj.odf_rec_device(p5, d5, mask, separate_peaks=sep)
    for _ in range(3): fj.odf_rec_device(p5, d5, mask, out=o5, separate_peaks=sep)
    torch.cuda.synchronize()
    L.fib_profile_enable(1); L.fib_profile_reset()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): fj.odf_rec_device(p5, d5, mask, out=o5, separate_peaks=sep)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    L.fib_profile_enable(0)
    parts = {}
    for k in (b"odf_gemm", b"odf_peaks", b"odfmax_refine", b"zero_dead", b"mask_compact", b"qa_normalize"):
        ms, cnt = C.c_double(), C.c_int64()
        L.fib_profile_get(k, C.byref(ms), C.byref(cnt))
        if cnt.value: parts[k.decode()] = round(ms.value / cnt.value, 3)
    print(name, "step %.3f ms" % wall, parts, flush=True)
    res[name] = {k: (v.clone() if hasattr(v, "clone") else [t.clone() for t in v]) for k, v in o5.items()}
    del p5, o5
a, b = res["two_tiles"], res["three_tiles"]
nn = lambda t: torch.nan_to_num(t, nan=-7.0, posinf=-8.0, neginf=-9.0)
print("pdf equal", torch.equal(nn(a["pdf"]), nn(b["pdf"])))
od = (a["odf"] - b["odf"]).abs().amax(0) / b["odf"].abs().amax(0).clamp_min(1e-30)
print("odf max rel diff of voxel max", float(od.max()), "rows differing", int(((a["odf"] != b["odf"]).any(1)).sum()))
for k in range(3):
    print("peak", k, "identical frac", float((a["peak"][k] == b["peak"][k]).all(0).float().mean()), "qa max abs diff", float((a["qa"][k] - b["qa"][k]).abs().max()))
print("odfmax", a["odfmax"].tolist(), b["odfmax"].tolist())
