#!/usr/bin/env python3
"""Per-kernel timing of one ODF reconstruction step (each configuration in a child process).
usage: gemm3_probe.py [gqi|dsi] ["T=1" (in-kernel stamps) | "G=f32" (f32-MFMA kernel) ...]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import ctypes as C
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import phantom
    kind = sys.argv[2]
    dev = torch.device("cuda", 0)
    SHAPE = (140, 140, 140)
    nvox = 140 ** 3
    bval, bvec = phantom.scheme_gqi() if kind == "gqi" else phantom.scheme_dsi()
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=3, device=dev)
    mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
    plan = fj.OdfPlan(kind, bval, bvec, fj.sphere_642, device=0)
    out = fj.odf_rec_device(plan, dwi, mask, normalize=False)
    L = fj.lib()
    for _ in range(3):
        fj.odf_rec_device(plan, dwi, mask, out=out, normalize=False)
    torch.cuda.synchronize()
    L.fib_profile_enable(1); L.fib_profile_reset()
    for _ in range(10):
        fj.odf_rec_device(plan, dwi, mask, out=out, normalize=False)
    torch.cuda.synchronize()
    ms, n = C.c_double(0), C.c_int64(0)
    r = []
    for k in ("odf_gemm", "odf_peaks", "dsi_fold"):
        L.fib_profile_get(k.encode(), C.byref(ms), C.byref(n))
        r.append("%s %.3f ms" % (k, ms.value / max(n.value, 1)))
    print("RESULT %s %s: %s" % (kind, sys.argv[3], "  ".join(r)), flush=True)
else:
    args = sys.argv[1:]
    kind = "gqi"
    if args and args[0] in ("gqi", "dsi"):
        kind = args.pop(0)
    for cfg in (args or ["T=0", "T=1", "G=f32"]):
        kv = dict(x.split("=") for x in cfg.split(","))
        env = dict(os.environ)
        env["FIBERS_GEMM3_STAMP"] = kv.get("T", "0")
        if "G" in kv: env["FIBERS_ODF_GEMM"] = kv["G"]
        subprocess.call([sys.executable, os.path.abspath(__file__), "child", kind, cfg], env=env)
