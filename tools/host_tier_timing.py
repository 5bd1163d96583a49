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
    # the C call alone, outputs allocated and touched beforehand (what the library itself costs)
    import ctypes as C
    from fibers_jl_amd import _lib
    from fibers_jl_amd.dti import DTI_FIELDS, _check_tables, _mask_checked
    L = _lib.lib()
    bv, bg = _check_tables(dwi)
    m, mdt = _mask_checked(mask, SHAPE)
    if "dti" in name:
        outs = {k: np.ones(SHAPE + ((3,) if "vec" in k else (1,)), np.float32, order="F") for k in DTI_FIELDS}
        o = _lib.DtiOut(*[outs[k].ctypes.data for k in DTI_FIELDS])
        call = lambda: L.fib_dti_fit(0, host.ctypes.data, 140, 140, 140, len(bval), m.ctypes.data, mdt, bv.ctypes.data, bg.ctypes.data, C.byref(o))
        gb = (host.nbytes + sum(v.nbytes for v in outs.values())) / 1e9
    else:
        sph = fj.sphere_642
        v = np.asfortranarray(sph.vertices, np.float32); f = np.asfortranarray(sph.faces, np.int32)
        odf = np.ones(SHAPE + (sph.nvert,), np.float32, order="F")
        pk = [np.ones(SHAPE + (3,), np.float32, order="F") for _ in range(3)]
        qa = [np.ones(SHAPE + (1,), np.float32, order="F") for _ in range(3)]
        call = lambda: L.fib_gqi_rec(0, host.ctypes.data, 140, 140, 140, len(bval), m.ctypes.data, mdt, bv.ctypes.data, bg.ctypes.data,
                                     v.ctypes.data, v.shape[0], f.ctypes.data, f.shape[0], 1.25, odf.ctypes.data,
                                     _lib.P3(*[a.ctypes.data for a in pk]), _lib.P3(*[a.ctypes.data for a in qa]))
        gb = (host.nbytes + odf.nbytes + sum(a.nbytes for a in pk + qa)) / 1e9
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        assert call() == 0
        ts.append(time.perf_counter() - t0)
    print("%-20s C call alone (outputs pre-touched): %s ms; best %.1f ms = %.1f GB/s over the link (%.2f GB both ways)"
          % ("", " ".join("%.1f" % (t * 1e3) for t in ts), min(ts) * 1e3, gb / min(ts), gb))
    del out, dwi, host
