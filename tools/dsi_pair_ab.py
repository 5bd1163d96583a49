import os
"""DSI two-tile kernel: paired 16:16 workgroup split (default: samples fetched once) against the cost-balanced 17:15 split without pairing."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the switch this tool flips exists in the DIAGNOSTIC build only (csrc/common.h ab_env; make -C fibers.jl_amd/csrc stamp)
os.environ.setdefault("FIBERS_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fibers.jl_amd", "libfibers_hip_stamp.so"))
import torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0); L = fj.lib()
b5, g5 = phantom.scheme_dsi()
d5, _ = phantom.make_dwi_torch((140,140,140), b5, g5, seed=5, device=dev)
mask = torch.ones(140**3, dtype=torch.uint8, device=dev)
p5 = fj.OdfPlan("dsi", b5, g5, fj.sphere_642, hann_width=32, device=0)
o5 = fj.odf_rec_device(p5, d5, mask)
ref = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in o5.items()}
for rep in range(2):
    for pair in (None, "1"):
        if pair: os.environ.pop("FIBERS_DSI_NA", None)          # default: paired 16:16
        else: os.environ["FIBERS_DSI_NA"] = "17"                 # cost-balanced 17:15, no pairing
        for _ in range(3): fj.odf_rec_device(p5, d5, mask, out=o5)
        torch.cuda.synchronize()
        L.fib_profile_enable(1); L.fib_profile_reset()
        t0 = time.perf_counter()
        for _ in range(20): fj.odf_rec_device(p5, d5, mask, out=o5)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        L.fib_profile_enable(0)
        ms, cnt = C.c_double(), C.c_int64()
        L.fib_profile_get(b"odf_gemm", C.byref(ms), C.byref(cnt))
        same = all(torch.equal(torch.nan_to_num(o5[k]), torch.nan_to_num(ref[k])) for k in ("odf", "pdf"))
        print("pair", pair, "kernel %.3f ms step %.3f ms" % (ms.value / cnt.value, wall), "same", same, flush=True)
