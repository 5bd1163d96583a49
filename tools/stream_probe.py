#!/usr/bin/env python3
"""Timing of the C4 tracking step (DTI principal eigenvector, ball mask, 998 592 seeds) kernel by kernel.
Run on the GPU box: python tools/stream_probe.py [repeats]; environment switches of csrc/stream.hip apply."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import phantom, _lib
    rep = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    SHAPE = (140, 140, 140)
    nvox = 140 ** 3
    mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
    b2, g2 = phantom.scheme_dti(60, 4, 1000.0, seed=2)
    d2, _ = phantom.make_dwi_torch(SHAPE, b2, g2, seed=2, device=dev, nfib=1)
    p2 = fj.DtiPlan(b2, g2, device=0)
    o2 = fj.dti_fit_device(p2, d2, mask)
    bm = phantom.ball_mask_torch(SHAPE, dev)
    field, mout = fj.stream_field_device([o2["eigvec1"]], fa=o2["fa"], fa_thresh=0.1, mask=bm)
    seeds = torch.nonzero(mout).flatten().contiguous()
    sub = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)
    res = fj.stream_device(field, SHAPE, seeds, sub)
    torch.cuda.synchronize()
    L.fib_profile_enable(1); L.fib_profile_reset()
    t0 = time.perf_counter()
    for _ in range(rep):
        res = fj.stream_device(field, SHAPE, seeds, sub)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / rep
    L.fib_profile_enable(0)
    import ctypes as C
    npts = int(res["xyz"].shape[0])
    print("lines %d points %d  step %.3f ms  %.0f Mpoints/s  checksum %.6e" %
          (res["npts"].numel(), npts, dt * 1e3, npts / dt / 1e6, float(res["xyz"].double().sum().item())))
    for name in ("stream_trace", "stream_count", "stream_scan", "stream_pack", "stream_write"):
        ms, n = C.c_double(0), C.c_int64(0)
        if L.fib_profile_get(name.encode(), C.byref(ms), C.byref(n)) == 0 and n.value:
            print("  %-14s %.3f ms x %d" % (name, ms.value / n.value, n.value))


if __name__ == "__main__":
    main()
