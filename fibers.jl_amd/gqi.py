"""gqi_rec / dsi_rec / find_peaks host mirror (reference: gqi.jl:7 `GQI, gqi_rec, find_peaks!, gqi_write`,
dsi.jl:7 `DSI, dsi_rec, dsi_write`)."""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import _lib
from .dti import _check_tables, _chk_dev, _dwi_arg, _mask_arg, _mask_checked, _stream_ptr
from .mri import MRI
from .odf import ODF, sphere_642


@dataclass
class GQI:
    """Container for outputs of a GQI fit (gqi.jl:10-14)"""
    odf: MRI
    peak: List[MRI]
    qa: List[MRI]


@dataclass
class DSI:
    """Container for outputs of a DSI reconstruction (dsi.jl:10-15)"""
    pdf: MRI
    odf: MRI
    peak: List[MRI]
    qa: List[MRI]


def _odf_args(odf_dirs: ODF):
    v = np.asfortranarray(odf_dirs.vertices, dtype=np.float32)
    f = np.asfortranarray(odf_dirs.faces, dtype=np.int32)
    return v, f


def _p3(arrs):
    return _lib.P3(*[a.vol.ctypes.data if isinstance(a, MRI) else a for a in arrs])


def gqi_rec(dwi: MRI, mask: MRI, odf_dirs: ODF = sphere_642, sigma: float = 1.25, device: int = 0) -> GQI:
    """Generalized q-sampling imaging reconstruction (gqi.jl:109)."""
    bval, bvec = _check_tables(dwi)
    vol = _dwi_arg(dwi)
    nx, ny, nz, nvol = vol.shape
    m, mdt = _mask_checked(mask, (nx, ny, nz))
    v, f = _odf_args(odf_dirs)
    ref = mask if isinstance(mask, MRI) else dwi
    odf = MRI.like(ref, odf_dirs.nvert)
    peak = [MRI.like(ref, 3) for _ in range(3)]
    qa = [MRI.like(ref, 1) for _ in range(3)]
    _lib.check(_lib.lib().fib_gqi_rec(device, vol.ctypes.data, nx, ny, nz, nvol, m.ctypes.data, mdt | _lib.FIB_MASK_OUTPUTS_ZEROED,
                                      bval.ctypes.data, bvec.ctypes.data,
                                      v.ctypes.data, v.shape[0], f.ctypes.data, f.shape[0], float(sigma),
                                      odf.vol.ctypes.data, _p3(peak), _p3(qa)))
    return GQI(odf, peak, qa)


def dsi_rec(dwi: MRI, mask: MRI, odf_dirs: ODF = sphere_642, hann_width: int = 32, device: int = 0) -> DSI:
    """Diffusion spectrum imaging reconstruction (dsi.jl:171)."""
    bval, bvec = _check_tables(dwi)
    vol = _dwi_arg(dwi)
    nx, ny, nz, nvol = vol.shape
    m, mdt = _mask_checked(mask, (nx, ny, nz))
    v, f = _odf_args(odf_dirs)
    ref = mask if isinstance(mask, MRI) else dwi
    pdf = MRI.like(ref, nvol)
    odf = MRI.like(ref, odf_dirs.nvert)
    peak = [MRI.like(ref, 3) for _ in range(3)]
    qa = [MRI.like(ref, 1) for _ in range(3)]
    _lib.check(_lib.lib().fib_dsi_rec(device, vol.ctypes.data, nx, ny, nz, nvol, m.ctypes.data, mdt | _lib.FIB_MASK_OUTPUTS_ZEROED,
                                      bval.ctypes.data, bvec.ctypes.data,
                                      v.ctypes.data, v.shape[0], f.ctypes.data, f.shape[0], int(hann_width),
                                      pdf.vol.ctypes.data, odf.vol.ctypes.data, _p3(peak), _p3(qa)))
    return DSI(pdf, odf, peak, qa)


def find_peaks(odf, odf_dirs: ODF = sphere_642, device: int = 0):
    """find_peaks!(W) (gqi.jl:180-201) on host ODFs.  odf: [..., nvert] amplitudes on the half sphere (any leading
    shape, e.g. `GQI.odf.vol`); returns (isort_top int32 [..., 3]: the first three entries of `isort`, 0-based
    first-half vertex rows, -1 beyond the tessellation; nvalid int32 [...] = count(odf_peak .> 0))."""
    o = np.asarray(odf, dtype=np.float32)
    nvert = odf_dirs.nvert
    if o.shape[-1] != nvert:
        raise ValueError("last axis of odf must be the %d half-sphere vertices" % nvert)
    lead = o.shape[:-1]
    nvox = int(np.prod(lead)) if lead else 1
    planar = np.ascontiguousarray(o.reshape(nvox, nvert).T)             # [nvert, nvox]
    v, f = _odf_args(odf_dirs)
    top = np.empty((3, nvox), np.int32)
    nvalid = np.empty(nvox, np.int32)
    _lib.check(_lib.lib().fib_find_peaks(device, planar.ctypes.data, nvox, v.ctypes.data, v.shape[0], f.ctypes.data,
                                         f.shape[0], top.ctypes.data, nvalid.ctypes.data))
    return top.T.reshape(lead + (3,)), nvalid.reshape(lead)


def find_peaks_work(odf, odf_dirs: ODF = sphere_642, device: int = 0):
    """find_peaks!(W) (gqi.jl:180-201) with every output the reference's work struct receives.  odf: [..., nvert]; returns
    (odf_peak float32 [..., nvert]: W.odf_peak, the amplitudes of the local peaks and 0 elsewhere; isort int32 [..., nvert]:
    W.isort = sortperm(odf_peak, rev=true), 0-based; nvalid int32 [...]: the function's return value)."""
    o = np.asarray(odf, dtype=np.float32)
    nvert = odf_dirs.nvert
    if o.shape[-1] != nvert:
        raise ValueError("last axis of odf must be the %d half-sphere vertices" % nvert)
    lead = o.shape[:-1]
    nvox = int(np.prod(lead)) if lead else 1
    planar = np.ascontiguousarray(o.reshape(nvox, nvert).T)             # [nvert, nvox]
    v, f = _odf_args(odf_dirs)
    pk = np.empty((nvert, nvox), np.float32)
    isort = np.empty((nvert, nvox), np.int32)
    nvalid = np.empty(nvox, np.int32)
    _lib.check(_lib.lib().fib_find_peaks_work(device, planar.ctypes.data, nvox, v.ctypes.data, v.shape[0], f.ctypes.data,
                                              f.shape[0], pk.ctypes.data, isort.ctypes.data, nvalid.ctypes.data))
    return pk.T.reshape(lead + (nvert,)), isort.T.reshape(lead + (nvert,)), nvalid.reshape(lead)


# ---------------------------------------------------------------------------------------------
# device-resident form
# ---------------------------------------------------------------------------------------------
ODF_FORMATS = {"default": 0, "fp16x2": 1, "bf16x3": 2, "f32": 3}      # FIB_ODF_FORMAT_* (include/fibers_hip.h)


class OdfPlan:
    """GQIwork (gqi.jl:32-82) or DSIwork (dsi.jl:41-143) resident on one GPU.  `format`: the operand format of the contraction
    (include/fibers_hip.h FIB_ODF_FORMAT_*): "fp16x2" (two fp16 pieces per f32 operand, the default), "bf16x3" (three exact bf16
    pieces), "f32" (f32 MFMA chain) or "default" (the environment's choice); `plan.format` is what the kernels actually run."""

    def __init__(self, kind: str, bval, bvec, odf_dirs: ODF = sphere_642, sigma: float = 1.25,
                 hann_width: int = 32, device: int = 0, format: str = "default"):
        if format not in ODF_FORMATS:
            raise ValueError("format must be one of %s" % sorted(ODF_FORMATS))
        fmt = ODF_FORMATS[format]
        self._h = C.c_void_p()
        self.kind, self.device = kind, device
        bval = np.ascontiguousarray(bval, np.float32)
        bvec = np.asfortranarray(np.asarray(bvec, np.float32).reshape(-1, 3))
        v, f = _odf_args(odf_dirs)
        L = _lib.lib()
        self.nvol, self.nvert = int(bval.shape[0]), odf_dirs.nvert
        if kind == "gqi":
            _lib.check(L.fib_gqi_plan_create_fmt(device, bval.ctypes.data, bvec.ctypes.data, self.nvol, v.ctypes.data,
                                                 v.shape[0], f.ctypes.data, f.shape[0], float(sigma), fmt, C.byref(self._h)))
        elif kind == "dsi":
            _lib.check(L.fib_dsi_plan_create_fmt(device, bval.ctypes.data, bvec.ctypes.data, self.nvol, v.ctypes.data,
                                                 v.shape[0], f.ctypes.data, f.shape[0], int(hann_width), fmt, C.byref(self._h)))
        else:
            raise ValueError("kind must be 'gqi' or 'dsi'")

    @property
    def format(self) -> str:
        """the operand format the plan's contraction kernels run (a plan falls back to a wider one when its matrix needs it)"""
        code = _lib.lib().fib_odf_plan_format(self._h)
        if code <= 0:
            _lib.check(code if code < 0 else -1)
        return {v: k for k, v in ODF_FORMATS.items()}[code]

    def list_unit(self, stream=None) -> str:
        """diagnostic: the unit of the voxel list the next reconstruction call on this plan will use ("octets": aligned groups of 32
        voxels, the default; "quads": aligned groups of 4, chosen by the previous call for sparse masks).  Results do not depend on it."""
        code = _lib.lib().fib_odf_plan_list_unit(self._h, _stream_ptr(stream))
        if code < 0:
            _lib.check(code)
        return "octets" if code else "quads"

    def matrix(self):
        nr, nv, nt = C.c_int(0), C.c_int(0), C.c_int(0)
        L = _lib.lib()
        _lib.check(L.fib_odf_plan_matrix(self._h, None, C.byref(nr), C.byref(nv), C.byref(nt)))
        A = np.zeros((nr.value, nv.value), np.float32, order="F")
        _lib.check(L.fib_odf_plan_matrix(self._h, A.ctypes.data, None, None, None))
        return A

    def close(self):
        if self._h:
            _lib.lib().fib_odf_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def odf_rec_device(plan: OdfPlan, dwi, mask, out: Optional[dict] = None, normalize: bool = True, stream=None,
                   out_prezeroed: bool = False, separate_peaks: bool = False, raw_odfmax: bool = False):
    """dwi: float32 CUDA tensor [nvol, nvox]; mask uint8 [nvox].  Returns dict(odf [nvert,nvox],
    pdf [nvol,nvox] (DSI), peak [3][3,nvox], qa [3][nvox], odfmax float32[2] = {max, nan flag}).
    separate_peaks = FIB_ODF_SEPARATE_PEAKS: a caller that cuts ONE volume into pieces sets it for every piece when any piece's
    voxel count is not a multiple of 4 (such a piece cannot run the fused peak scan, and the two forms differ at rounding level).
    raw_odfmax = FIB_ODF_RAW_ODFMAX: odfmax = {maximum of the means that are not NaN, NaN flag}, the form a MAX all-reduce takes."""
    import torch
    _chk_dev(dwi, torch.float32, "dwi")
    _chk_dev(mask, torch.uint8, "mask")
    nvox = mask.numel()
    if dwi.numel() != nvox * plan.nvol:
        raise ValueError("dwi has %d elements, expected nvol*nvox = %d" % (dwi.numel(), nvox * plan.nvol))
    dev = dwi.device
    if out is None:
        out = dict(odf=torch.empty((plan.nvert, nvox), dtype=torch.float32, device=dev),
                   peak=[torch.empty((3, nvox), dtype=torch.float32, device=dev) for _ in range(3)],
                   qa=[torch.empty(nvox, dtype=torch.float32, device=dev) for _ in range(3)],
                   odfmax=torch.empty(2, dtype=torch.float32, device=dev))
        if plan.kind == "dsi":
            out["pdf"] = torch.empty((plan.nvol, nvox), dtype=torch.float32, device=dev)
    pdf_ptr = out["pdf"].data_ptr() if plan.kind == "dsi" else None
    _lib.check(_lib.lib().fibd_odf_rec(plan._h, dwi.data_ptr(), mask.data_ptr(), nvox, pdf_ptr, out["odf"].data_ptr(),
                                       _lib.P3(*[t.data_ptr() for t in out["peak"]]),
                                       _lib.P3(*[t.data_ptr() for t in out["qa"]]),
                                       out["odfmax"].data_ptr(),
                                       (1 if normalize else 0) | (2 if out_prezeroed else 0) | (4 if separate_peaks else 0) | (8 if raw_odfmax else 0),
                                       _stream_ptr(stream)))
    return out


def qa_normalize_device(qa, odfmax, stream=None, raw=False):
    """qa[k] ./= odfmax (gqi.jl:166-168).  odfmax: a float, or a float32 CUDA tensor whose first element is the divisor (e.g.
    the all-reduced `out["odfmax"]`: it never leaves the device).  raw: the tensor is the pair {maximum of the non-NaN means, NaN
    flag} (odf_rec_device(raw_odfmax=True), all-reduced): the divisor is NaN if the flag is set, and odfmax[0] becomes the divisor."""
    nvox = qa[0].numel()
    if raw:
        _lib.check(_lib.lib().fibd_qa_normalize_pair(_lib.P3(*[t.data_ptr() for t in qa]), nvox, odfmax.data_ptr(), _stream_ptr(stream)))
    elif hasattr(odfmax, "data_ptr"):
        _lib.check(_lib.lib().fibd_qa_normalize_dev(_lib.P3(*[t.data_ptr() for t in qa]), nvox, odfmax.data_ptr(), _stream_ptr(stream)))
    else:
        _lib.check(_lib.lib().fibd_qa_normalize(_lib.P3(*[t.data_ptr() for t in qa]), nvox, float(odfmax), _stream_ptr(stream)))


def find_peaks_device(plan: OdfPlan, odf, stream=None):
    """find_peaks!(W) (gqi.jl:180) over a planar ODF tensor [nvert, nvox] ->
    (isort_top int32 [3, nvox] 0-based, nvalid int32 [nvox])"""
    import torch
    _chk_dev(odf, torch.float32, "odf")
    nvox = odf.numel() // plan.nvert
    top = torch.empty((3, nvox), dtype=torch.int32, device=odf.device)
    nvalid = torch.empty(nvox, dtype=torch.int32, device=odf.device)
    _lib.check(_lib.lib().fibd_find_peaks(plan._h, odf.data_ptr(), nvox, top.data_ptr(), nvalid.data_ptr(), _stream_ptr(stream)))
    return top, nvalid
