#!/usr/bin/env python3
"""Shader clock and board power while a kernel of the hot path runs back to back: the step loops in this process, `rocm-smi` is sampled
from a child process (never exec'ed from here).  usage: python tools/clock_under_load.py [gqi|dsi|dti|stream] [seconds]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kind = sys.argv[1] if len(sys.argv) > 1 else "gqi"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0


def smi(tag):
    try:
        o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:                                   # noqa
        o = "rocm-smi failed: %r" % (e,)
    keep = [l.strip() for l in o.split("\n") if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "Temperature (Sensor junction)", "failed"))]
    print("[%s] %s" % (tag, " | ".join(keep)), flush=True)


smi("idle, before torch")
import torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
dev = torch.device("cuda", 0)
SHAPE = (140, 140, 140)
mask = torch.ones(140 ** 3, dtype=torch.uint8, device=dev)
if kind in ("gqi", "dsi"):
    bval, bvec = phantom.scheme_gqi() if kind == "gqi" else phantom.scheme_dsi()
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=3, device=dev)
    plan = fj.OdfPlan(kind, bval, bvec, fj.sphere_642, sigma=1.25, hann_width=32, device=0)
    out = fj.odf_rec_device(plan, dwi, mask, normalize=True)
    step = lambda: fj.odf_rec_device(plan, dwi, mask, out=out, normalize=True)   # noqa: E731
else:
    bval, bvec = phantom.scheme_dti(60, 4, 1000.0, 2)
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, 2, dev, nfib=1)
    plan = fj.DtiPlan(bval, bvec)
    o = fj.dti_fit_device(plan, dwi, mask)
    if kind == "dti":
        step = lambda: fj.dti_fit_device(plan, dwi, mask, out=o)                  # noqa: E731
    else:
        import numpy as np
        bm = phantom.ball_mask_torch(SHAPE, dev)
        field, mout = fj.stream_field_device([o["eigvec1"]], fa=o["fa"], fa_thresh=0.1, mask=bm)
        seeds = torch.nonzero(mout).flatten()
        sub = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)
        bufs = fj.StreamBuffers(dev)
        fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)
        step = lambda: fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)   # noqa: E731
step()
torch.cuda.synchronize()
stop = False


def sampler():
    k = 0
    while not stop:
        time.sleep(1.0)
        smi("%s running, %d s" % (kind, k + 1))
        k += 1


th = threading.Thread(target=sampler)
th.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    n += 50
el = time.perf_counter() - t0
stop = True
th.join()
print("%s: %d steps in %.2f s = %.3f ms/step" % (kind, n, el, el / n * 1e3))
