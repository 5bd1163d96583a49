#!/usr/bin/env python3
"""Small driver for rocprofv3: runs N steps of one hot-path stage on resident synthetic data.
usage: prof_step.py {gqi|dti|stream|dsi|c5} [steps]   (c5: BASELINE config 5's tracking -- 3 peaks of a DSI fit, ~10 M lines, the fused kernel)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "gqi"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
SHAPE = (140, 140, 140)
nvox = 140 ** 3
dev = torch.device("cuda", 0)
mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
if os.environ.get("PROF_BALL_MASK"):
    mask = phantom.ball_mask_torch(SHAPE, dev)          # 36 % of the volume, like a brain mask
if what == "gqi":
    bval, bvec = phantom.scheme_gqi()
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, 3, dev)
    plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642)
    if os.environ.get("PROF_SLAB"):                         # rank 0 of 8's z-slab through the sharded driver (no collective with one rank)
        from fibers_jl_amd import dist as fd
        z0, z1 = fd.slab_bounds(140, 8, 0, 19600)
        ns = (z1 - z0) * 19600
        counts8 = [(b - a) * 19600 for a, b in (fd.slab_bounds(140, 8, r, 19600) for r in range(8))]
        dwi_s, mask_s = dwi[:, :ns].contiguous(), mask[:ns].contiguous()
        out = fj.odf_rec_device(plan, dwi_s, mask_s, normalize=False)
        for _ in range(steps):
            fd.odf_rec_sharded(plan, dwi_s, mask_s, out=out, counts=counts8, out_prezeroed=bool(os.environ.get("PROF_PREZEROED")))
        torch.cuda.synchronize()
        sys.exit(0)
    out = fj.odf_rec_device(plan, dwi, mask)
    for _ in range(steps):
        fj.odf_rec_device(plan, dwi, mask, out=out, out_prezeroed=bool(os.environ.get("PROF_PREZEROED")))
elif what == "dsi":
    bval, bvec = phantom.scheme_dsi()
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, 5, dev)
    plan = fj.OdfPlan("dsi", bval, bvec, fj.sphere_642)
    out = fj.odf_rec_device(plan, dwi, mask)
    for _ in range(steps):
        fj.odf_rec_device(plan, dwi, mask, out=out)
elif what == "dti":
    bval, bvec = phantom.scheme_dti(60, 4, 1000.0, 2)
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, 2, dev, nfib=1)
    plan = fj.DtiPlan(bval, bvec)
    out = fj.dti_fit_device(plan, dwi, mask)
    for _ in range(steps):
        fj.dti_fit_device(plan, dwi, mask, out=out)
elif what == "stream":
    bval, bvec = phantom.scheme_dti(60, 4, 1000.0, 2)
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, 2, dev, nfib=1)
    plan = fj.DtiPlan(bval, bvec)
    o = fj.dti_fit_device(plan, dwi, mask)
    bm = phantom.ball_mask_torch(SHAPE, dev)
    field, mout = fj.stream_field_device([o["eigvec1"]], fa=o["fa"], fa_thresh=0.1, mask=bm)
    seeds = torch.nonzero(mout).flatten()
    sub = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)
    import os
    if os.environ.get("PROF_STREAM_RUN"):                  # the one-call form into kept buffers (what bench.py's extra times)
        bufs = fj.StreamBuffers(dev)
        for _ in range(steps):
            fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)
    elif os.environ.get("PROF_STREAM_PREALLOC"):           # the two-call form into a kept output buffer (bench.py's tracking step)
        keep = {}

        def xyz_out(n):
            if keep.get("t") is None or keep["t"].numel() < 3 * n:
                keep["t"] = torch.empty(int(3 * n * 1.05) + 16, dtype=torch.float32, device=dev)
            return keep["t"]
        for _ in range(steps):
            fj.stream_device(field, SHAPE, seeds, sub, xyz_out=xyz_out)
    else:
        for _ in range(steps):
            fj.stream_device(field, SHAPE, seeds, sub)
elif what == "c5":
    import numpy as np
    bval, bvec = phantom.scheme_dsi()
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, 5, dev)
    plan = fj.OdfPlan("dsi", bval, bvec, fj.sphere_642)
    o5 = fj.odf_rec_device(plan, dwi, mask)
    del dwi
    bm = phantom.ball_mask_torch(SHAPE, dev)
    field, mout = fj.stream_field_device(o5["peak"], f=o5["qa"], f_thresh=0.03, mask=bm)
    seeds = torch.nonzero(mout).flatten()
    sub = torch.from_numpy(fj.make_sublist(10, np.random.default_rng(5))).to(dev)
    bufs = fj.StreamBuffers(dev)
    for _ in range(steps + 1):
        fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)
torch.cuda.synchronize()
print("done", what, steps)
