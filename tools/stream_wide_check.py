#!/usr/bin/env python3
"""The tracer's WIDE form (64-bit voxel indices and gather offsets; the product takes it for fields of 2^28 vectors or more) against the 32-bit
form on SMALL fields, where the DIAGNOSTIC build can force it (FIBERS_STREAM_WIDE=1): nearest-voxel tracking with 1, 2 and 3 vectors per voxel,
the trilinear option and LCM-guided tracking must give bit-identical lines.  (That the 64-bit offsets reach past 4 GiB is what
tests/test_gpu_stream.py::test_stream_wide_field_past_the_32_bit_gather_limit checks on a real 4.6-GB field.)  Exit code 0 = all identical."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FIBERS_HIP_LIB", os.path.join(ROOT, "fibers.jl_amd", "libfibers_hip_stamp.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n = 36
    shape = (n, n, n)
    nvox = n ** 3
    g = torch.Generator(device=dev); g.manual_seed(3)
    base = torch.from_numpy(np.ascontiguousarray(phantom.fibre_field(n, n, n).astype(np.float32).reshape(nvox, 3, order="F").T)).to(dev)

    def vecs(k):
        v = base + 0.35 * k * torch.randn(base.shape, device=dev, generator=g)
        return (v / v.norm(dim=0, keepdim=True)).contiguous()
    mask = (torch.rand(nvox, device=dev, generator=g) < 0.95).to(torch.uint8)
    sub = torch.from_numpy(fj.make_sublist(2, np.random.default_rng(4))).to(dev)
    bad = 0
    cases = []
    for nvec in (1, 2, 3):
        field, mout = fj.stream_field_device([vecs(k) for k in range(nvec)], mask=mask)
        seeds = torch.nonzero(mout).flatten()
        cases.append(("nearest, %d vector(s)" % nvec, lambda f=field, s=seeds: fj.stream_device(f, shape, s, sub, len_max=60)))
        if nvec in (1, 2):
            cases.append(("trilinear, %d vector(s)" % nvec, lambda f=field, s=seeds: fj.stream_device(f, shape, s, sub, len_max=60, interp="trilinear")))
    # LCM-guided: a 2-D section, three in-plane orientations, one 10-element LCM per pixel
    n2 = 96
    ang = [((torch.rand(n2 * n2, device=dev, generator=g) - 0.5 + k * 3.14159265 / 3 + 1.5707963) % 3.14159265) - 1.5707963 for k in range(3)]
    ov2 = [fj.angles_to_vectors_device(a_.clamp(-1.5707963, 1.5707963), volres=(0.5, 0.5, 2.0))[0] for a_ in ang]
    lc = torch.rand((10, n2 * n2), device=dev, generator=g)
    fld, mo = fj.stream_field_device(ov2, mask=torch.ones(n2 * n2, dtype=torch.uint8, device=dev))
    sd2 = torch.nonzero(mo).flatten()
    s2 = torch.tensor([[0.1, -0.2, 0.0]], dtype=torch.float32, device=dev)
    cases.append(("LCM-guided, 3 vectors", lambda: fj.stream_device(fld, (n2, n2, 1), sd2, s2, lcms=lc, lcm_thresh=0.099, strdims=(0, 1), rng_seed=7, len_max=80)))
    for name, run in cases:
        os.environ.pop("FIBERS_STREAM_WIDE", None)
        a = run()
        os.environ["FIBERS_STREAM_WIDE"] = "1"
        b = run()
        os.environ.pop("FIBERS_STREAM_WIDE", None)
        torch.cuda.synchronize()
        same = all(torch.equal(a[k], b[k]) for k in ("npts", "seed_index", "xyz")) and ("flags" not in a or torch.equal(a["flags"], b["flags"]))
        print("%-28s lines %7d points %9d  wide == 32-bit: %s" % (name, int(a["npts"].numel()), int(a["xyz"].shape[0]), same), flush=True)
        bad += 0 if same and int(a["npts"].numel()) > 100 else 1
    print("stream wide check:", "ok" if bad == 0 else "%d FAILURES" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
