#!/usr/bin/env python3
"""RUMBA-SD on the 140^3 x 270-frame phantom, ball mask, sphere_724: per-iteration kernel times and a result checksum.
Run on the GPU box: python tools/rumba_probe.py [iterations]."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import ctypes as C
    import torch
    import fibers_jl_amd as fj
    from fibers_jl_amd import phantom, _lib
    nit = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    SHAPE = (140, 140, 140)
    bm = phantom.ball_mask_torch(SHAPE, dev)
    b4, g4 = phantom.scheme_gqi()
    d4, _ = phantom.make_dwi_torch(SHAPE, b4, g4, seed=3, device=dev)
    rp = fj.RumbaPlan(b4, g4, fj.sphere_724, device=0)
    fj.rumba_rec_device(rp, d4, bm, SHAPE, niter=2)
    torch.cuda.synchronize()
    L.fib_profile_enable(1); L.fib_profile_reset()
    t0 = time.perf_counter()
    rr = fj.rumba_rec_device(rp, d4, bm, SHAPE, niter=nit)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.fib_profile_enable(0)
    print("%d iterations %.2f ms  snr_mean %.6f  sum(fodf) %.9e  sum(gfa) %.9e" %
          (nit, dt * 1e3, rr["snr_mean"], float(rr["fodf"].double().sum()), float(rr["gfa"].double().sum())))
    for name in ("matrix_gemm", "rumba_tv", "rumba_elementwise"):
        ms, n = C.c_double(0), C.c_int64(0)
        L.fib_profile_get(name.encode(), C.byref(ms), C.byref(n))
        print("  %-18s %.3f ms per iteration (%d launches)" % (name, ms.value / nit, n.value))


if __name__ == "__main__":
    main()
