#!/usr/bin/env python3
"""A/B of the fused tracer (fibd_stream_run's default where it applies: the workgroup that traced 512 lines packs them behind a decoupled
look-back over the workgroups' totals -- VERDICT r4 item 4) against trace + scan + pack as three launches (DIAGNOSTIC build:
FIBERS_STREAM_UNFUSED=1), both through fibd_stream_run: results must be
bit-identical; wall time per call and the kernels' hipEvent times.  C4 (DTI field, 998 592 seeds, one offset) and, with `c5`, a three-vector
field with 10 offsets.  usage: stream_fused_ab.py [c4|c5] [reps]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FIBERS_HIP_LIB", os.path.join(ROOT, "fibers.jl_amd", "libfibers_hip_stamp.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "c4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SHAPE = (140, 140, 140)
dev = torch.device("cuda", 0)
L = fj.lib()
bval, bvec = phantom.scheme_dti(60, 4, 1000.0, 2)
dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, 2, dev, nfib=1)
plan = fj.DtiPlan(bval, bvec)
mask = torch.ones(140 ** 3, dtype=torch.uint8, device=dev)
o = fj.dti_fit_device(plan, dwi, mask)
bm = phantom.ball_mask_torch(SHAPE, dev)
if what == "c4":
    field, mout = fj.stream_field_device([o["eigvec1"]], fa=o["fa"], fa_thresh=0.1, mask=bm)
    sub = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)
else:
    field, mout = fj.stream_field_device([o["eigvec1"], o["eigvec2"], o["eigvec3"]], fa=o["fa"], fa_thresh=0.1, mask=bm)
    sub = torch.from_numpy(fj.make_sublist(10, np.random.default_rng(5))).to(dev)
seeds = torch.nonzero(mout).flatten()


def get(name):
    ms, n = C.c_double(0), C.c_int64(0)
    L.fib_profile_get(name.encode(), C.byref(ms), C.byref(n))
    return ms.value / max(n.value, 1)


res, rows = {}, []
for fused in (False, True, False, True):
    os.environ.pop("FIBERS_STREAM_UNFUSED", None); os.environ.pop("FIBERS_STREAM_FUSED", None)
    os.environ["FIBERS_STREAM_FUSED" if fused else "FIBERS_STREAM_UNFUSED"] = "1"
    bufs = fj.StreamBuffers(dev)
    for _ in range(3):
        r = fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    L.fib_profile_enable(1); L.fib_profile_reset()
    for _ in range(reps):
        r = fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)
    torch.cuda.synchronize()
    tr, pk, sc = get("stream_trace"), get("stream_pack"), get("stream_scan")
    L.fib_profile_enable(0)
    row = dict(workload=what, fused=fused, lines=int(r["npts"].numel()), points=int(r["xyz"].shape[0]), wall_ms=wall, trace_kernel_ms=tr, scan_ms=sc, pack_ms=pk,
               kernels_ms=tr + sc + pk)
    rows.append(row)
    print(json.dumps(row), flush=True)
    res[fused] = {k: r[k].clone() for k in ("npts", "seed_index", "xyz")}
same = all(torch.equal(res[False][k], res[True][k]) for k in ("npts", "seed_index", "xyz"))
print(json.dumps(dict(workload=what, bit_identical=same)))
sys.exit(0 if same else 1)
