"""rumba_rec host mirror (reference: rusd.jl:7 `RUMBASD, rumba_rec, rumba_write`) -- SURVEY.md row N4."""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import _lib
from .dti import _check_tables, _chk_dev, _dwi_arg, _mask_checked, _stream_ptr
from .mri import MRI
from .odf import ODF, sphere_724


@dataclass
class RUMBASD:
    """Container for outputs of a RUMBA-SD fit (rusd.jl:11-20)"""
    fodf: MRI
    fgm: MRI
    fcsf: MRI
    peak: List[MRI]
    gfa: MRI
    var: MRI
    snr_mean: float
    snr_std: float


def _coil_mode(coil_combine: str) -> int:
    if coil_combine == "SoS-GRAPPA":
        return 1
    if coil_combine != "SMF-SENSE":
        raise ValueError("Unknown coil combine mode " + coil_combine)          # rusd.jl:433
    return 0


def rumba_rec(dwi: MRI, mask: MRI, odf_dirs: ODF = sphere_724, niter: int = 600, lam_para: float = 1.7e-3,
              lam_perp: float = 0.2e-3, lam_csf: float = 3.0e-3, lam_gm: float = 0.8e-4, ncoils: int = 1,
              coil_combine: str = "SMF-SENSE", ipat_factor: int = 1, use_tv: bool = True, device: int = 0) -> RUMBASD:
    """Robust and unbiased model-based spherical deconvolution (rusd.jl:419)."""
    bval, bvec = _check_tables(dwi)
    sos = _coil_mode(coil_combine)
    if ipat_factor < 1:
        raise ValueError("iPAT factor must be a positive integer")             # rusd.jl:437
    vol = _dwi_arg(dwi)
    nx, ny, nz, nvol = vol.shape
    m, mdt = _mask_checked(mask, vol.shape[:3])
    v = np.asfortranarray(odf_dirs.vertices, dtype=np.float32)
    ref = mask if isinstance(mask, MRI) else dwi
    fodf = MRI.like(ref, odf_dirs.nvert)
    sc = [MRI.like(ref, 1) for _ in range(4)]
    peak = [MRI.like(ref, 3) for _ in range(5)]
    out = _lib.RumbaOut(fodf.vol.ctypes.data, *[s.vol.ctypes.data for s in sc], (C.c_void_p * 5)(*[p.vol.ctypes.data for p in peak]))
    sm, ss = C.c_float(0), C.c_float(0)
    _lib.check(_lib.lib().fib_rumba_rec(device, vol.ctypes.data, nx, ny, nz, nvol, m.ctypes.data, mdt,
                                        bval.ctypes.data, bvec.ctypes.data, v.ctypes.data, v.shape[0], int(niter),
                                        float(lam_para), float(lam_perp), float(lam_csf), float(lam_gm), int(ncoils), sos,
                                        int(ipat_factor), 1 if use_tv else 0, C.byref(out), C.byref(sm), C.byref(ss)))
    return RUMBASD(fodf, sc[0], sc[1], peak, sc[2], sc[3], float(sm.value), float(ss.value))


class RumbaPlan:
    """kernel + contraction plans of rumba_rec resident on one GPU"""

    def __init__(self, bval, bvec, odf_dirs: ODF = sphere_724, lam_para=1.7e-3, lam_perp=0.2e-3, lam_csf=3.0e-3,
                 lam_gm=0.8e-4, device: int = 0):
        self._h = C.c_void_p()
        bval = np.ascontiguousarray(bval, np.float32)
        bvec = np.asfortranarray(np.asarray(bvec, np.float32).reshape(-1, 3))
        v = np.asfortranarray(odf_dirs.vertices, dtype=np.float32)
        self.nvert, self.nvol, self.device = odf_dirs.nvert, int(bval.shape[0]), device
        _lib.check(_lib.lib().fib_rumba_plan_create(device, bval.ctypes.data, bvec.ctypes.data, self.nvol, v.ctypes.data,
                                                    v.shape[0], float(lam_para), float(lam_perp), float(lam_csf),
                                                    float(lam_gm), C.byref(self._h)))

    def kernel(self):
        nd, nc = C.c_int(0), C.c_int(0)
        L = _lib.lib()
        _lib.check(L.fib_rumba_plan_kernel(self._h, None, C.byref(nd), C.byref(nc)))
        K = np.zeros((nd.value, nc.value), np.float32, order="F")
        _lib.check(L.fib_rumba_plan_kernel(self._h, K.ctypes.data, None, None))
        return K

    def close(self):
        if self._h:
            _lib.lib().fib_rumba_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rumba_rec_device(plan: RumbaPlan, dwi, mask, shape, niter=600, ncoils=1, coil_combine="SMF-SENSE", ipat_factor=1,
                     use_tv=True, stream=None):
    """dwi: float32 CUDA [nvol, nvox]; mask uint8 [nvox]; shape = (nx, ny, nz).  Returns dict(fodf [nvert,nvox], fgm, fcsf,
    gfa, var [nvox], peak [5][3,nvox], snr_mean, snr_std)."""
    import torch
    _chk_dev(dwi, torch.float32, "dwi")
    _chk_dev(mask, torch.uint8, "mask")
    nx, ny, nz = shape
    nvox = nx * ny * nz
    dev = dwi.device
    out = dict(fodf=torch.empty((plan.nvert, nvox), dtype=torch.float32, device=dev),
               fgm=torch.empty(nvox, dtype=torch.float32, device=dev), fcsf=torch.empty(nvox, dtype=torch.float32, device=dev),
               gfa=torch.empty(nvox, dtype=torch.float32, device=dev), var=torch.empty(nvox, dtype=torch.float32, device=dev),
               peak=[torch.empty((3, nvox), dtype=torch.float32, device=dev) for _ in range(5)])
    ro = _lib.RumbaOut(out["fodf"].data_ptr(), out["fgm"].data_ptr(), out["fcsf"].data_ptr(), out["gfa"].data_ptr(),
                       out["var"].data_ptr(), (C.c_void_p * 5)(*[t.data_ptr() for t in out["peak"]]))
    sm, ss = C.c_float(0), C.c_float(0)
    _lib.check(_lib.lib().fibd_rumba_rec(plan._h, dwi.data_ptr(), mask.data_ptr(), nx, ny, nz, int(niter), int(ncoils),
                                         _coil_mode(coil_combine), int(ipat_factor), 1 if use_tv else 0, C.byref(ro),
                                         C.byref(sm), C.byref(ss), _stream_ptr(stream)))
    out["snr_mean"], out["snr_std"] = float(sm.value), float(ss.value)
    return out
