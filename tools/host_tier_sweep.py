#!/usr/bin/env python3
"""fib_gqi_rec on host arrays (140^3 x 270): the C call alone, outputs pre-touched, against the host tier's chunk size (FIBERS_HOST_CHUNK)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom, _lib
from fibers_jl_amd.dti import _check_tables, _mask_checked
SHAPE = (140, 140, 140); dev = torch.device("cuda", 0)
bval, bvec = phantom.scheme_gqi()
d, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, 2, dev, nfib=1)
host = np.asfortranarray(d.cpu().numpy().T.reshape(SHAPE + (len(bval),), order="F")); del d
torch.cuda.empty_cache()
mk = np.ones(SHAPE, np.uint8)
if os.environ.get("MASK") == "ball":
    mk = np.asfortranarray(phantom.ball_mask_torch(SHAPE, dev).reshape(SHAPE[::-1]).permute(2, 1, 0).cpu().numpy().astype(np.uint8))
dwi = fj.MRI(host, bval, bvec); mask = fj.MRI(mk)
print("mask keeps %.1f %% of the volume" % (100.0 * mk.mean()))
L = _lib.lib(); bv, bg = _check_tables(dwi); m, mdt = _mask_checked(mask, SHAPE)
sph = fj.sphere_642
v = np.asfortranarray(sph.vertices, np.float32); f = np.asfortranarray(sph.faces, np.int32)
odf = np.ones(SHAPE + (sph.nvert,), np.float32, order="F")
pk = [np.ones(SHAPE + (3,), np.float32, order="F") for _ in range(3)]
qa = [np.ones(SHAPE + (1,), np.float32, order="F") for _ in range(3)]
call = lambda: L.fib_gqi_rec(0, host.ctypes.data, 140, 140, 140, len(bval), m.ctypes.data, mdt, bv.ctypes.data, bg.ctypes.data,
                             v.ctypes.data, v.shape[0], f.ctypes.data, f.shape[0], 1.25, odf.ctypes.data,
                             _lib.P3(*[a.ctypes.data for a in pk]), _lib.P3(*[a.ctypes.data for a in qa]))
gb = (host.nbytes + odf.nbytes + sum(a.nbytes for a in pk + qa)) / 1e9
for chunk in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32768,65536,131072,262144,524288").split(",")]:
    os.environ["FIBERS_HOST_CHUNK"] = str(chunk)
    assert call() == 0
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); assert call() == 0; ts.append(time.perf_counter() - t0)
    print("chunk %7d voxels: %s ms; best %.1f ms = %.1f GB/s over the link" % (chunk, " ".join("%.1f" % (t * 1e3) for t in ts), min(ts) * 1e3, gb / min(ts)), flush=True)
