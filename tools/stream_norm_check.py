#!/usr/bin/env python3
"""normalise3 (csrc/stream.hip): the scaling-free square-root / shared-reciprocal division sequences the tracer takes for ordinary vectors
against hipcc's generic expansions (DIAGNOSTIC build, FIBERS_STREAM_NORM_GENERIC=1 forces them for every vector): bit-identical lines for
nearest-voxel tracking (1 / 3 vectors), trilinear, LCM-guided (2-D section: one component exactly zero) and the microscopy regime, over
several smoothing coefficients, plus fields whose vectors are scaled to the ends of the plain range (|w| ~ 2^-38, 2^38: no normalised
vectors -- the reference does not require them) and beyond it (2^-60, 2^60: the generic path on both sides).  With --time it also prints
the trace kernel's time per call for both forms on the C4 workload.  Exit code 0 = all identical."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FIBERS_HIP_LIB", os.path.join(ROOT, "fibers.jl_amd", "libfibers_hip_stamp.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402


def both(run):
    os.environ.pop("FIBERS_STREAM_NORM_GENERIC", None)
    a = run()
    os.environ["FIBERS_STREAM_NORM_GENERIC"] = "1"
    b = run()
    os.environ.pop("FIBERS_STREAM_NORM_GENERIC", None)
    torch.cuda.synchronize()
    same = all(torch.equal(a[k].view(torch.int32) if a[k].dtype == torch.float32 else a[k], b[k].view(torch.int32) if b[k].dtype == torch.float32 else b[k])
               for k in ("npts", "seed_index", "xyz")) and ("flags" not in a or torch.equal(a["flags"], b["flags"]))
    return a, same


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = 40
    shape = (n, n, n)
    nvox = n ** 3
    g = torch.Generator(device=dev); g.manual_seed(11)
    base = torch.from_numpy(np.ascontiguousarray(phantom.fibre_field(n, n, n).astype(np.float32).reshape(nvox, 3, order="F").T)).to(dev)

    def vecs(k, scale=1.0, unit=True):
        v = base + 0.3 * k * torch.randn(base.shape, device=dev, generator=g)
        if unit:
            v = v / v.norm(dim=0, keepdim=True)
        return (v * scale).contiguous()
    mask = (torch.rand(nvox, device=dev, generator=g) < 0.97).to(torch.uint8)
    sub = torch.from_numpy(fj.make_sublist(2, np.random.default_rng(4))).to(dev)
    cases = []
    for nvec in (1, 3):
        field, mout = fj.stream_field_device([vecs(k) for k in range(nvec)], mask=mask)
        seeds = torch.nonzero(mout).flatten()
        for sm in (0.2, 0.5, 0.95):
            cases.append(("nearest, %d vector(s), smooth %.2f" % (nvec, sm), lambda f=field, s=seeds, sm=sm: fj.stream_device(f, shape, s, sub, len_max=60, smooth_coeff=sm)))
        cases.append(("nearest, %d vector(s), 170 degrees" % nvec, lambda f=field, s=seeds: fj.stream_device(f, shape, s, sub, len_max=60, smooth_coeff=0.5, ang_thresh=170.0)))
    field1, mout1 = fj.stream_field_device([vecs(0)], mask=mask)
    seeds1 = torch.nonzero(mout1).flatten()
    cases.append(("trilinear, 1 vector", lambda: fj.stream_device(field1, shape, seeds1, sub, len_max=60, interp="trilinear")))
    # vectors that are not unit length (the reference normalises only the smoothed direction): the ends of the plain range, and past them
    for e in (-60, -41, -39, -20, 20, 38, 41, 60):
        fs, ms = fj.stream_field_device([vecs(0, scale=2.0 ** e, unit=False)], mask=mask)
        sd = torch.nonzero(ms).flatten()
        cases.append(("vectors x 2^%d" % e, lambda f=fs, s=sd, e=e: fj.stream_device(f, shape, s, sub, len_max=40, step_size=0.5 * 2.0 ** -e, ang_thresh=90.0, len_min=1, smooth_coeff=0.5)))
    # a field with tiny / zero / negative-zero components
    v = vecs(0)
    v[2] = torch.where(torch.rand(nvox, device=dev, generator=g) < 0.5, torch.zeros((), device=dev), v[2] * 1e-30)
    v[1] = torch.where(torch.rand(nvox, device=dev, generator=g) < 0.3, -torch.zeros((), device=dev), v[1])
    fz, mz = fj.stream_field_device([v.contiguous()], mask=mask)
    sz = torch.nonzero(mz).flatten()
    cases.append(("zero and 1e-30 components", lambda: fj.stream_device(fz, shape, sz, sub, len_max=60, smooth_coeff=0.3)))
    # LCM-guided, 2-D section
    n2 = 96
    ang = [((torch.rand(n2 * n2, device=dev, generator=g) - 0.5 + k * 3.14159265 / 3 + 1.5707963) % 3.14159265) - 1.5707963 for k in range(3)]
    ov2 = [fj.angles_to_vectors_device(a_.clamp(-1.5707963, 1.5707963), volres=(0.5, 0.5, 2.0))[0] for a_ in ang]
    lc = torch.rand((10, n2 * n2), device=dev, generator=g)
    fld, mo = fj.stream_field_device(ov2, mask=torch.ones(n2 * n2, dtype=torch.uint8, device=dev))
    sd2 = torch.nonzero(mo).flatten()
    s2 = torch.tensor([[0.1, -0.2, 0.0]], dtype=torch.float32, device=dev)
    cases.append(("LCM-guided, 2-D, 3 vectors", lambda: fj.stream_device(fld, (n2, n2, 1), sd2, s2, lcms=lc, lcm_thresh=0.099, strdims=(0, 1), rng_seed=7, len_max=80)))
    cases.append(("microscopy regime", lambda: fj.stream_device(field1, shape, seeds1[::16].contiguous(), sub[:1], len_max=40, search_dist=4, search_ang=20.0, ang_thresh=30.0, step_size=1.0)))
    bad = 0
    for name, run in cases:
        try:
            a, same = both(run)
        except Exception as e:                                     # noqa: BLE001
            print("%-40s ERROR %r" % (name, e), flush=True)
            bad += 1
            continue
        print("%-40s lines %7d points %9d  plain == generic: %s" % (name, int(a["npts"].numel()), int(a["xyz"].shape[0]), same), flush=True)
        bad += 0 if same and int(a["npts"].numel()) > 0 else 1
    if args.time:
        import ctypes as C
        L = fj.lib()
        SHAPE = (140, 140, 140)
        b2, g2 = phantom.scheme_dti(60, 4, 1000.0, 2)
        d2, _ = phantom.make_dwi_torch(SHAPE, b2, g2, 2, dev, nfib=1)
        o2 = fj.dti_fit_device(fj.DtiPlan(b2, g2), d2, torch.ones(140 ** 3, dtype=torch.uint8, device=dev))
        bm = phantom.ball_mask_torch(SHAPE, dev)
        field, mout = fj.stream_field_device([o2["eigvec1"]], fa=o2["fa"], fa_thresh=0.1, mask=bm)
        seeds = torch.nonzero(mout).flatten()
        sub1 = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)
        bufs = fj.StreamBuffers(dev)
        for rep in range(3):
            for gen in (0, 1):
                if gen:
                    os.environ["FIBERS_STREAM_NORM_GENERIC"] = "1"
                else:
                    os.environ.pop("FIBERS_STREAM_NORM_GENERIC", None)
                for _ in range(3):
                    fj.stream_device_run(field, SHAPE, seeds, sub1, buffers=bufs)
                torch.cuda.synchronize()
                L.fib_profile_enable(1); L.fib_profile_reset()
                for _ in range(10):
                    fj.stream_device_run(field, SHAPE, seeds, sub1, buffers=bufs)
                torch.cuda.synchronize()
                ms, k = C.c_double(0), C.c_int64(0)
                L.fib_profile_get(b"stream_trace", C.byref(ms), C.byref(k))
                L.fib_profile_enable(0)
                print("C4 trace kernel, %s: %.4f ms" % ("generic" if gen else "plain  ", ms.value / max(k.value, 1)), flush=True)
        os.environ.pop("FIBERS_STREAM_NORM_GENERIC", None)
    print("stream norm check:", "ok" if bad == 0 else "%d FAILURES" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
