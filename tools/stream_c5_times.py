#!/usr/bin/env python3
"""C5 tracking step alone (3 DSI peaks per voxel, qa threshold, ball mask, nsub = 10: ~10 M lines, 1.25 G points): kernel times of
trace / scan / pack with whatever library FIBERS_HIP_LIB names."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
SHAPE = (140, 140, 140); dev = torch.device("cuda", 0); L = fj.lib()
b5, g5 = phantom.scheme_dsi()
d5, _ = phantom.make_dwi_torch(SHAPE, b5, g5, seed=5, device=dev)
p5 = fj.OdfPlan("dsi", b5, g5, fj.sphere_642, hann_width=32, device=0)
o5 = fj.odf_rec_device(p5, d5, torch.ones(140 ** 3, dtype=torch.uint8, device=dev))
del d5
bm = phantom.ball_mask_torch(SHAPE, dev)
f3, m3 = fj.stream_field_device(o5["peak"], f=o5["qa"], f_thresh=0.03, mask=bm)
seeds = torch.nonzero(m3).flatten()
sub = torch.from_numpy(fj.make_sublist(10, np.random.default_rng(5))).to(dev)
del o5
torch.cuda.empty_cache()
keep = {}
def xyz_out(n):
    if keep.get("t") is None or keep["t"].numel() < 3 * n: keep["t"] = torch.empty(int(3 * n * 1.05) + 16, dtype=torch.float32, device=dev)
    return keep["t"]
for _ in range(2): r = fj.stream_device(f3, SHAPE, seeds, sub, xyz_out=xyz_out)
torch.cuda.synchronize()
L.fib_profile_enable(1); L.fib_profile_reset()
for _ in range(4): r = fj.stream_device(f3, SHAPE, seeds, sub, xyz_out=xyz_out)
torch.cuda.synchronize(); L.fib_profile_enable(0)
out = []
for nm in (b"stream_trace", b"stream_pack", b"stream_scan"):
    ms, cnt = C.c_double(), C.c_int64(); L.fib_profile_get(nm, C.byref(ms), C.byref(cnt))
    out.append("%s %.3f ms" % (nm.decode(), ms.value / max(cnt.value, 1)))
print(os.path.basename(os.environ.get("FIBERS_HIP_LIB", "libfibers_hip.so")), "lines %d points %d |" % (r["npts"].numel(), r["xyz"].shape[0]), " | ".join(out))
