#!/usr/bin/env python3
"""The FUSED tracer (trace + decoupled look-back + pack in one launch: fibd_stream_run from 2^21 lines on) against trace + scan + pack on
SMALL inputs, where only the DIAGNOSTIC build can force it (FIBERS_STREAM_FUSED=1; the product's full-size tests are the only other cover:
ADVICE r5).  For 1, 2 and 3 vectors per voxel: a line count that is not a multiple of the 512-line workgroup, lines dropped by len_min,
buffers that are too small (FIB_ERR_CAPACITY: the totals say what is needed, nothing is written past the capacities), and the enqueue
form whose counts stay on the device.  Lines must be bit-identical (stream.jl:625-690, 769-787).  Exit code 0 = all identical."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FIBERS_HIP_LIB", os.path.join(ROOT, "fibers.jl_amd", "libfibers_hip_stamp.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import _lib, phantom  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n = 30
    shape = (n, n, n)
    nvox = n ** 3
    g = torch.Generator(device=dev); g.manual_seed(3)
    base = torch.from_numpy(np.ascontiguousarray(phantom.fibre_field(n, n, n).astype(np.float32).reshape(nvox, 3, order="F").T)).to(dev)

    def vecs(k):
        v = base + 0.35 * k * torch.randn(base.shape, device=dev, generator=g)
        return (v / v.norm(dim=0, keepdim=True)).contiguous()
    mask = (torch.rand(nvox, device=dev, generator=g) < 0.93).to(torch.uint8)
    sub = torch.from_numpy(fj.make_sublist(2, np.random.default_rng(4))).to(dev)
    smod = sys.modules[fj.stream_device_run.__module__]
    bad = 0
    for nvec in (1, 2, 3):
        field, mout = fj.stream_field_device([vecs(k) for k in range(nvec)], mask=mask)
        seeds = torch.nonzero(mout).flatten()
        if (seeds.numel() * 2) % 512 == 0:
            seeds = seeds[:-1].contiguous()                                   # a partial last workgroup
        for len_min in (2, 25):                                               # 25: most lines of this field are dropped
            kw = dict(len_min=len_min, len_max=60, smooth_coeff=0.3)
            os.environ.pop("FIBERS_STREAM_FUSED", None)
            ref = fj.stream_device(field, shape, seeds, sub, **kw)            # trace + scan + pack (two calls)
            os.environ["FIBERS_STREAM_FUSED"] = "1"
            bufs = fj.StreamBuffers(dev)
            got = fj.stream_device_run(field, shape, seeds, sub, buffers=bufs, **kw)     # first call sizes the buffers (one retry)
            got2 = fj.stream_device_run(field, shape, seeds, sub, buffers=bufs, **kw)
            cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            fj.stream_device_run_enqueue(field, shape, seeds, sub, bufs, counts=cnt, **kw)
            torch.cuda.synchronize()
            same = all(torch.equal(got[k], ref[k]) and torch.equal(got2[k], ref[k]) for k in ("npts", "seed_index", "xyz"))
            same = same and cnt.tolist() == [int(ref["npts"].numel()), int(ref["xyz"].shape[0])]
            # too little room
            nl_ref, np_ref = int(ref["npts"].numel()), int(ref["xyz"].shape[0])
            capl, capp = max(1, nl_ref // 3), max(1, np_ref // 3)
            small_n = torch.full((capl,), -7, dtype=torch.int32, device=dev)
            small_s = torch.zeros(capl, dtype=torch.int64, device=dev)
            small_x = torch.full((capp + 8, 3), -7.0, device=dev)
            prm = smod._params(shape, nvec, len_min, 60, 45, 0.5, 0.3, 0, 10, smod.default_workspace(0))
            nl, npnt = C.c_int64(0), C.c_int64(0)
            rc = _lib.lib().fibd_stream_run(C.byref(prm), field.data_ptr(), seeds.data_ptr(), seeds.numel(), sub.data_ptr(), sub.shape[0],
                                            small_n.data_ptr(), small_s.data_ptr(), capl, small_x.data_ptr(), capp, C.byref(nl), C.byref(npnt), None)
            torch.cuda.synchronize()
            cap_ok = (rc == _lib.FIB_ERR_CAPACITY and nl.value == nl_ref and npnt.value == np_ref and bool((small_x[capp:] == -7.0).all()))
            kept = small_n != -7
            cap_ok = cap_ok and torch.equal(small_n[kept], ref["npts"][:capl][kept])
            os.environ.pop("FIBERS_STREAM_FUSED", None)
            ok = same and cap_ok and nl_ref > (200 if len_min > 2 else 5000)
            print("%d vector(s), len_min %2d: lines %6d of %6d, points %8d  fused == trace + pack: %s, capacity path: %s" % (
                nvec, len_min, nl_ref, seeds.numel() * 2, np_ref, same, cap_ok), flush=True)
            bad += 0 if ok else 1
    print("stream fused check:", "ok" if bad == 0 else "%d FAILURES" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
