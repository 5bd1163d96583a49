#!/usr/bin/env python3
"""What bounds the tracer?  (DIAGNOSTIC build; the switches below give WRONG lines -- they exist for timing only.)
C4 workload (998 592 lines, one vector per voxel), trace kernel ms per call:
  FIBERS_STREAM_DBG bit 1   no field gathers after the seed's (every step re-uses the seed voxel's vector)
  FIBERS_STREAM_DBG bit 2   no point stores
  smooth_coeff 0            no direction smoothing (normalise3: the Float64 square root and the three divisions)
  FIBERS_STREAM_SCRATCH_PLAIN = 1..6   the point stores with another cache policy (plain, sc1, sc0 sc1, sc1 nt, sc0 nt, sc0)
at three sizes (2 waves per SIMD; one full round of 8; the C4 size = two rounds).
Round 5's reading (profiles/r05/negative_results.txt): the stores bound the kernel -- without them it is 14 % shorter and THEN bound by
vector-ALU issue (smoothing off: another 35 %); without the gathers it is no shorter at all."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FIBERS_HIP_LIB", os.path.join(ROOT, "fibers.jl_amd", "libfibers_hip_stamp.so"))
import torch  # noqa: E402

import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    L = fj.lib()
    SHAPE = (140, 140, 140)
    b2, g2 = phantom.scheme_dti(60, 4, 1000.0, 2)
    d2, _ = phantom.make_dwi_torch(SHAPE, b2, g2, 2, dev, nfib=1)
    o2 = fj.dti_fit_device(fj.DtiPlan(b2, g2), d2, torch.ones(140 ** 3, dtype=torch.uint8, device=dev))
    bm = phantom.ball_mask_torch(SHAPE, dev)
    field, mout = fj.stream_field_device([o2["eigvec1"]], fa=o2["fa"], fa_thresh=0.1, mask=bm)
    seeds_all = torch.nonzero(mout).flatten()
    sub1 = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)
    bufs = fj.StreamBuffers(dev)

    def get(name):
        ms, k = C.c_double(0), C.c_int64(0)
        L.fib_profile_get(name, C.byref(ms), C.byref(k))
        return ms.value / max(k.value, 1)

    def t(seeds, **kw):
        for _ in range(3):
            fj.stream_device_run(field, SHAPE, seeds, sub1, buffers=bufs, **kw)
        torch.cuda.synchronize()
        L.fib_profile_enable(1); L.fib_profile_reset()
        for _ in range(10):
            fj.stream_device_run(field, SHAPE, seeds, sub1, buffers=bufs, **kw)
        torch.cuda.synchronize()
        out = (get(b"stream_trace"), get(b"stream_pack"))
        L.fib_profile_enable(0)
        return out

    for nl in (131072, 524288, 0):
        seeds = seeds_all[:nl].contiguous() if nl else seeds_all
        for dbg in (0, 1, 2, 3):
            for sm in (0.2, 0.0):
                os.environ["FIBERS_STREAM_DBG"] = str(dbg)
                print("lines %7d  gathers %-3s stores %-3s smoothing %-3s  trace %.4f ms" %
                      (int(seeds.numel()), "no" if dbg & 1 else "yes", "no" if dbg & 2 else "yes", "yes" if sm else "no", t(seeds, smooth_coeff=sm)[0]), flush=True)
    os.environ.pop("FIBERS_STREAM_DBG", None)
    names = {0: "nt (product)", 1: "plain", 2: "sc1", 3: "sc0 sc1", 4: "sc1 nt", 5: "sc0 nt", 6: "sc0"}
    for fl in range(7):
        os.environ["FIBERS_STREAM_SCRATCH_PLAIN"] = str(fl)
        tr, pa = t(seeds_all)
        print("point stores %-13s trace %.4f ms  pack %.4f ms" % (names[fl], tr, pa), flush=True)
    os.environ.pop("FIBERS_STREAM_SCRATCH_PLAIN", None)


if __name__ == "__main__":
    main()
