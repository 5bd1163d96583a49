#!/usr/bin/env python3
"""Would a point scratch that stays in cache pay?  (the scratch round trip is 3.3 of the tracker's 5.2 GB of HBM
traffic.)  Trace + scan + pack on the first n lines of the C4 workload, n = 32 K ... 1 M: kernel time per million lines, with the scratch
written / read non-temporally (the product) and with the default cache policy (DIAGNOSTIC build: FIBERS_STREAM_SCRATCH_PLAIN=1 + FIBERS_STREAM_PACK_PLAIN=1).  If small
batches with a cacheable scratch (32 K lines = 56 MB: inside the 256-MB Infinity Cache) ran well below the 1-M-line figure, tracing the
lines in sequential batches that reuse ONE scratch region would beat the single pass.  usage: stream_batch_scaling.py"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FIBERS_HIP_LIB", os.path.join(ROOT, "fibers.jl_amd", "libfibers_hip_stamp.so"))
import torch  # noqa: E402

import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402

SHAPE = (140, 140, 140)
dev = torch.device("cuda", 0)
L = fj.lib()
bval, bvec = phantom.scheme_dti(60, 4, 1000.0, 2)
dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, 2, dev, nfib=1)
plan = fj.DtiPlan(bval, bvec)
mask = torch.ones(140 ** 3, dtype=torch.uint8, device=dev)
o = fj.dti_fit_device(plan, dwi, mask)
bm = phantom.ball_mask_torch(SHAPE, dev)
field, mout = fj.stream_field_device([o["eigvec1"]], fa=o["fa"], fa_thresh=0.1, mask=bm)
seeds_all = torch.nonzero(mout).flatten()
# (a random subset: the lines of a batch should be as long as the whole set's, wherever the batch comes from)
perm = torch.randperm(seeds_all.numel(), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
sub = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)


def get(name):
    ms, n = C.c_double(0), C.c_int64(0)
    L.fib_profile_get(name.encode(), C.byref(ms), C.byref(n))
    return ms.value / max(n.value, 1)


rows = []
for plain in (False, True):
    if plain:
        os.environ["FIBERS_STREAM_SCRATCH_PLAIN"] = "1"
        os.environ["FIBERS_STREAM_PACK_PLAIN"] = "1"
    else:
        os.environ.pop("FIBERS_STREAM_SCRATCH_PLAIN", None)
        os.environ.pop("FIBERS_STREAM_PACK_PLAIN", None)
    for n in (32768, 65536, 131072, 262144, 524288, int(seeds_all.numel())):
        seeds = seeds_all[perm[:n]].sort().values.contiguous()
        bufs = fj.StreamBuffers(dev)
        for _ in range(3):
            fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)
        torch.cuda.synchronize()
        L.fib_profile_enable(1); L.fib_profile_reset()
        for _ in range(10):
            r = fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)
        torch.cuda.synchronize()
        tr, pk, sc = get("stream_trace"), get("stream_pack"), get("stream_scan")
        L.fib_profile_enable(0)
        npnt = int(r["xyz"].shape[0])
        row = dict(scratch="default policy" if plain else "non-temporal", lines=n, points=npnt, scratch_mb=n * 144 * 12 / 1e6, trace_ms=tr, scan_ms=sc, pack_ms=pk,
                   ms_per_million_lines=(tr + sc + pk) / n * 1e6, trace_ms_per_million_lines=tr / n * 1e6, pack_ms_per_million_lines=pk / n * 1e6)
        rows.append(row)
        print(json.dumps(row), flush=True)
