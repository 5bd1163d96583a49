#!/usr/bin/env python3
"""In-kernel clock of the contraction kernels (MI355X_MICROARCH.md "DVFS give-back" item 6).

Loads the DIAGNOSTIC build libfibers_hip_stamp.so (make -C fibers.jl_amd/csrc stamp: the product library with ONE
s_memtime / s_memrealtime pair around each workgroup's whole work loop, written to a buffer nothing else reads -- [r5] and nothing
else: the per-stage phase marks moved to the phase build, they made this kernel 14 % slower than the product's), runs each
kernel back to back for >= 2 s on the random phantom of the benchmark, and prints per kernel
    clock = d(s_memtime) / d(s_memrealtime) x 100 MHz   (median / min / max over the workgroups of the last launch)
next to the launch's wall time.  One JSON object on stdout.  The product library never executes a stamp.

usage: python tools/kernel_clock.py [--seconds 2.5] [--shape 140,140,140] [--kernels fused,unfused,dsi]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMP_LIB = os.path.join(ROOT, "fibers.jl_amd", "libfibers_hip_stamp.so")
os.environ["FIBERS_HIP_LIB"] = STAMP_LIB                      # before the package loads its library

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--shape", default="140,140,140")
    ap.add_argument("--kernels", default="fused,unfused,dsi")
    args = ap.parse_args()
    if not os.path.exists(STAMP_LIB):
        print(json.dumps(dict(error="libfibers_hip_stamp.so not built (make -C fibers.jl_amd/csrc stamp)")))
        return 1
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import phantom
    L = fj.lib()
    L.fib_debug_clock_stamps.restype = C.c_int
    L.fib_debug_clock_stamps.argtypes = [C.c_void_p, C.c_int]
    dev = torch.device("cuda", 0)
    shape = tuple(int(v) for v in args.shape.split(","))
    nvox = shape[0] * shape[1] * shape[2]
    sph = fj.sphere_642
    mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
    names = {1: "odf_gemm3_kernel<MB,NX,8> (unfused)", 2: "odf_gemm3_kernel<10,1,8,FUSE> (default GQI)", 3: "odf_gemm3_kernel<MB,NX,8,FOLD> (DSI)",
             8: "odf_dsi2_kernel (DSI: fused ODF tile + pdf tile)"}
    res = {}

    def measure(label, step):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        L.fib_debug_clock_clear()
        L.fib_profile_enable(1)
        L.fib_profile_reset()
        from fibers_jl_amd import energy as en          # the SMU's view of the same seconds: board energy, reported shader clock
        e0 = en.energy_joules()
        t0 = time.perf_counter()
        n = 0
        smu = []
        while time.perf_counter() - t0 < args.seconds:
            for _ in range(20):
                step()
            n += 20
            if n % 100 == 0:
                torch.cuda.synchronize()
                smu.append(en.sclk_mhz())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e1 = en.energy_joules()
        wall = (t1 - t0) / n
        smu = [v for v in smu if v]
        L.fib_profile_enable(0)
        ms, cnt = C.c_double(0), C.c_int64(0)
        L.fib_profile_get(b"odf_gemm", C.byref(ms), C.byref(cnt))
        buf = np.zeros((2048, 4), np.uint64)
        rc = L.fib_debug_clock_stamps(buf.ctypes.data, 2048)
        assert rc == 0
        live = buf[buf[:, 1] > 0]
        ghz = live[:, 0].astype(np.float64) / live[:, 1].astype(np.float64) * 0.1
        kid = int(np.bincount(live[:, 2].astype(np.int64)).argmax()) if len(live) else 0
        res[label] = dict(kernel=names.get(kid, "?"), workgroups=int(len(live)), clock_ghz_median=float(np.median(ghz)) if len(live) else None,
                          clock_ghz_min=float(ghz.min()) if len(live) else None, clock_ghz_max=float(ghz.max()) if len(live) else None,
                          loop_us_median=float(np.median(live[:, 1]) / 100.0) if len(live) else None,
                          kernel_ms_hipevent=ms.value / max(cnt.value, 1), step_ms_wall=wall * 1e3, steps=n,
                          smu_sclk_mhz_mean=float(np.mean(smu)) if smu else None, smu_sclk_mhz_min=float(np.min(smu)) if smu else None,
                          smu_sclk_mhz_max=float(np.max(smu)) if smu else None,
                          board_watts=(e1 - e0) / (t1 - t0) if e0 is not None and e1 is not None else None,
                          joules_per_step=(e1 - e0) / n if e0 is not None and e1 is not None else None)
        if kid == 8 and len(live):                      # odf_dsi2_kernel: workgroups with an even index on their XCD run the ODF tile, odd ones the pdf tile
            wg = np.nonzero(buf[:, 1] > 0)[0]
            par = (wg >> 3) & 1
            for t, nm in ((0, "odf_tile"), (1, "pdf_tile")):
                sel = live[par == t]
                if len(sel):
                    res[label][nm] = dict(workgroups=int(len(sel)), loop_us_median=float(np.median(sel[:, 1]) / 100.0), loop_us_max=float(sel[:, 1].max() / 100.0),
                                          clock_ghz_median=float(np.median(sel[:, 0].astype(np.float64) / sel[:, 1].astype(np.float64) * 0.1)),
                                          items_median=float(np.median(sel[:, 3])))

    kernels = args.kernels.split(",")
    if any(k in ("fused", "unfused") for k in kernels):
        bval, bvec = phantom.scheme_gqi()
        dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=3, device=dev)
        plan = fj.OdfPlan("gqi", bval, bvec, sph, sigma=1.25, device=0)
        out = fj.odf_rec_device(plan, dwi, mask, normalize=False)
        for k in kernels:
            if k not in ("fused", "unfused"):
                continue
            sep = k == "unfused"                        # FIB_ODF_SEPARATE_PEAKS: the unfused contraction + the separate peak kernel
            measure("gqi_" + k, lambda: fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True, separate_peaks=sep))
        del dwi, out, plan
        torch.cuda.empty_cache()
    if "dsi" in kernels:
        b5, g5 = phantom.scheme_dsi()
        d5, _ = phantom.make_dwi_torch(shape, b5, g5, seed=5, device=dev)
        p5 = fj.OdfPlan("dsi", b5, g5, sph, hann_width=32, device=0)
        o5 = fj.odf_rec_device(p5, d5, mask)
        measure("dsi", lambda: fj.odf_rec_device(p5, d5, mask, out=o5))
    res["note"] = ("diagnostic build (-DFIB_CLOCK_STAMP); clock = d(s_memtime)/d(s_memrealtime) x 100 MHz around each workgroup's work loop of the "
                   "last launch after >= %.1f s of back-to-back steps on the random phantom; nominal 2.4 GHz" % args.seconds)
    print(json.dumps(res))
    return 0


if __name__ == "__main__":
    sys.exit(main())
