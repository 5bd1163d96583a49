#!/usr/bin/env python3
"""fibd_stream_run (batches, trace || pack on two streams) against trace + pack one after the other, C4 workload (998 592 seeds on
the DTI principal-eigenvector field, ball mask): wall time per call for a range of batch counts.  usage: stream_run_ab.py [nsub]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402

SHAPE = (140, 140, 140)
dev = torch.device("cuda", 0)
nsub = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bval, bvec = phantom.scheme_dti(60, 4, 1000.0, 2)
dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, 2, dev, nfib=1)
plan = fj.DtiPlan(bval, bvec)
o = fj.dti_fit_device(plan, dwi, torch.ones(140 ** 3, dtype=torch.uint8, device=dev))
bm = phantom.ball_mask_torch(SHAPE, dev)
field, mout = fj.stream_field_device([o["eigvec1"]], fa=o["fa"], fa_thresh=0.1, mask=bm)
seeds = torch.nonzero(mout).flatten()
sub = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev) if nsub == 1 else torch.from_numpy(fj.make_sublist(nsub, np.random.default_rng(5))).to(dev)
del dwi


def timeit(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


xyz = {}


def xyz_out(n):
    if xyz.get("t") is None or xyz["t"].numel() < 3 * n:
        xyz["t"] = torch.empty(int(3 * n * 1.05) + 16, dtype=torch.float32, device=dev)
    return xyz["t"]


r = fj.stream_device(field, SHAPE, seeds, sub, xyz_out=xyz_out)
print("lines %d points %d" % (r["npts"].numel(), r["xyz"].shape[0]))
print("trace + pack, two calls: %.3f ms" % timeit(lambda: fj.stream_device(field, SHAPE, seeds, sub, xyz_out=xyz_out)), flush=True)
bufs = fj.StreamBuffers(dev)
for nb in (1, 2, 3, 4, 6, 8, 12, 16):
    os.environ["FIBERS_STREAM_BATCHES"] = str(nb)
    fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)
    print("fibd_stream_run, %2d batches: %.3f ms" % (nb, timeit(lambda: fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs))), flush=True)
