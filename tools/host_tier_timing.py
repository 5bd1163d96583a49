#!/usr/bin/env python3
"""PCIe-inclusive timing of the host-buffer (drop-in) entry points at benchmark size.
Everything a Julia caller would pay: device allocation, H2D of the DWI volume, kernels, D2H of the outputs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import fibers_jl_amd as fj  # noqa: E402
from fibers_jl_amd import phantom  # noqa: E402

SHAPE = (140, 140, 140)
nvox = 140 ** 3
dev = torch.device("cuda", 0)
mask = fj.MRI(np.ones(SHAPE, np.uint8))
for name, (bval, bvec), fn in (("dti_fit 140^3x64", phantom.scheme_dti(60, 4, 1000.0, 2), fj.dti_fit),
                               ("gqi_rec 140^3x270", phantom.scheme_gqi(), fj.gqi_rec)):
    d, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, 2, dev, nfib=1)
    host = np.asfortranarray(d.cpu().numpy().T.reshape(SHAPE + (len(bval),), order="F"))
    del d
    torch.cuda.empty_cache()
    dwi = fj.MRI(host, bval, bvec)
    fn(dwi, mask)                                   # warm-up (library load, first-touch of output pages)
    t0 = time.perf_counter()
    out = fn(dwi, mask)
    dt = time.perf_counter() - t0
    print("%-20s %.1f ms end-to-end incl. PCIe + host allocation  -> %.1f Mvoxels/s" % (name, dt * 1e3, nvox / dt / 1e6))
    del out, dwi, host
