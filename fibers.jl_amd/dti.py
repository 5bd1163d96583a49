"""dti_fit / adc_fit host mirror (reference: dti.jl:7 exports `DTI, adc_fit, dti_fit, dti_write`).

`dti_fit(dwi, mask)` and `adc_fit(dwi, mask)` take host `MRI`s and go through the host-buffer C ABI
(fib_dti_fit / fib_adc_fit).  `DtiPlan` + `dti_fit_device` / `adc_fit_device` are the device-resident
form on torch tensors (fibd_*), used for multi-GPU sharding and benchmarking."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .mri import MRI

DTI_FIELDS = ("s0", "eigval1", "eigval2", "eigval3", "eigvec1", "eigvec2", "eigvec3", "rd", "md", "fa")


@dataclass
class DTI:
    """Container for outputs of a DTI fit (dti.jl:11-22)"""
    s0: MRI
    eigval1: MRI
    eigval2: MRI
    eigval3: MRI
    eigvec1: MRI
    eigvec2: MRI
    eigvec3: MRI
    rd: MRI
    md: MRI
    fa: MRI


def _mask_arg(mask):
    m = mask.vol if isinstance(mask, MRI) else np.asarray(mask)
    if m.ndim == 4:
        m = m[..., 0]
    m = np.asfortranarray(m)
    name = m.dtype.name
    if name not in _lib.DTYPES:
        m = np.asfortranarray(m.astype(np.float64))
        name = "float64"
    return m, _lib.DTYPES[name]


def _dwi_arg(dwi):
    if dwi.vol.dtype != np.float32:
        # the reference's per-voxel methods dispatch on Vector{Float32} (dti.jl:286): other eltypes are a MethodError
        raise TypeError("dwi.vol must be float32 (the reference dispatches on Float32 only)")
    return dwi.vol if dwi.vol.flags.f_contiguous else np.asfortranarray(dwi.vol)


def _check_tables(dwi, need_bvec=True):
    """The reference's table checks (dti.jl:166-168, 223-229) plus the shapes the C ABI relies on.  Returns
    (bval float32 contiguous [nframes], bvec float32 Fortran-ordered [nframes, 3] or None): the arrays whose pointers go
    to the library — `dwi.bvec` itself may have been assigned in any memory order after the MRI was built."""
    if dwi.bval is None or len(dwi.bval) == 0:
        raise RuntimeError("Missing b-value table from input DWI structure")        # dti.jl:167,224
    if need_bvec and (dwi.bvec is None or len(dwi.bvec) == 0):
        raise RuntimeError("Missing gradient table from input DWI structure")       # dti.jl:228
    bval = np.ascontiguousarray(dwi.bval, np.float32).reshape(-1)
    if bval.shape[0] != dwi.nframes:
        raise ValueError("b-value table length %d does not match %d frames" % (bval.shape[0], dwi.nframes))
    bvec = None
    if need_bvec:
        bvec = np.asarray(dwi.bvec, np.float32)
        if bvec.shape != (dwi.nframes, 3):
            raise ValueError("gradient table must be [%d x 3], got %s" % (dwi.nframes, bvec.shape))
        bvec = np.asfortranarray(bvec)
    return bval, bvec


def _mask_checked(mask, shape3):
    """_mask_arg + the shape check every host wrapper needs (a mismatch would be an out-of-bounds read in the library)"""
    m, mdt = _mask_arg(mask)
    if m.shape != tuple(shape3):
        raise ValueError("mask shape %s does not match DWI volume %s" % (m.shape, tuple(shape3)))
    return m, mdt


def dti_fit(dwi: MRI, mask: MRI, device: int = 0) -> DTI:
    """Fit tensors to DWIs and return a `DTI` structure (dti.jl:221).  device: a GPU index, or _lib.DEVICE_ALL for the
    device set declared with fibers_jl_amd.init() (z-slab sharding, dti.jl:258)."""
    bval, bvec = _check_tables(dwi)
    L = _lib.lib()
    vol = _dwi_arg(dwi)
    nx, ny, nz, nvol = vol.shape
    m, mdt = _mask_checked(mask, (nx, ny, nz))
    outs = {k: MRI.like(mask if isinstance(mask, MRI) else dwi, 3 if "vec" in k else 1) for k in DTI_FIELDS}
    o = _lib.DtiOut(*[outs[k].vol.ctypes.data for k in DTI_FIELDS])
    _lib.check(L.fib_dti_fit(device, vol.ctypes.data, nx, ny, nz, nvol, m.ctypes.data, mdt | _lib.FIB_MASK_OUTPUTS_ZEROED,
                             bval.ctypes.data, bvec.ctypes.data, C.byref(o)))
    return DTI(**outs)


def adc_fit(dwi: MRI, mask: MRI, device: int = 0):
    """Fit the apparent diffusion coefficient; returns (adc, s0) (dti.jl:164)."""
    bval, _ = _check_tables(dwi, need_bvec=False)
    L = _lib.lib()
    vol = _dwi_arg(dwi)
    nx, ny, nz, nvol = vol.shape
    m, mdt = _mask_checked(mask, (nx, ny, nz))
    ref = mask if isinstance(mask, MRI) else dwi
    adc, s0 = MRI.like(ref, 1), MRI.like(ref, 1)
    _lib.check(L.fib_adc_fit(device, vol.ctypes.data, nx, ny, nz, nvol, m.ctypes.data, mdt | _lib.FIB_MASK_OUTPUTS_ZEROED,
                             bval.ctypes.data, adc.vol.ctypes.data, s0.vol.ctypes.data))
    return adc, s0


# ---------------------------------------------------------------------------------------------
# device-resident form (torch tensors are plumbing: device memory + streams)
# ---------------------------------------------------------------------------------------------
class DtiPlan:
    """DTIwork / ADCwork (dti.jl:39-155) resident on one GPU.  bvec=None builds the ADC plan."""

    def __init__(self, bval, bvec=None, device: int = 0):
        self._h = C.c_void_p()
        self.device = device
        bval = np.ascontiguousarray(bval, np.float32)
        self.nvol = int(bval.shape[0])
        bv = None if bvec is None else np.asfortranarray(np.asarray(bvec, np.float32).reshape(-1, 3))
        _lib.check(_lib.lib().fib_dti_plan_create(device, bval.ctypes.data, None if bv is None else bv.ctypes.data,
                                                  self.nvol, C.byref(self._h)))
        self.np = 7 if bv is not None else 2

    def tables(self):
        A = np.zeros((self.nvol, self.np), np.float32, order="F")
        pA = np.zeros((self.np, self.nvol), np.float32, order="F")
        n = C.c_int(0)
        _lib.check(_lib.lib().fib_dti_plan_tables(self._h, A.ctypes.data, pA.ctypes.data, C.byref(n)))
        return A, pA

    def last_partial_count(self, stream=None):
        c = C.c_int64(0)
        _lib.check(_lib.lib().fibd_dti_last_partial_count(self._h, _stream_ptr(stream), C.byref(c)))
        return int(c.value)

    def close(self):
        if self._h:
            _lib.lib().fib_dti_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stream_ptr(stream):
    if stream is None:
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(getattr(stream, "cuda_stream", stream))


def _sync(stream):
    """wait for `stream` (a torch stream, a raw hipStream_t handle, or None = the current stream)"""
    import torch
    if stream is None:
        torch.cuda.current_stream().synchronize()
    elif hasattr(stream, "synchronize"):
        stream.synchronize()
    else:                                               # a raw handle: wait for that stream, not for the current device
        torch.cuda.ExternalStream(int(getattr(stream, "value", stream) or 0)).synchronize()


def _chk_dev(t, dtype, what):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise TypeError("%s must be a contiguous CUDA tensor of dtype %s" % (what, dtype))
    return t


def dti_fit_device(plan: DtiPlan, dwi, mask, out=None, stream=None):
    """dwi: float32 CUDA tensor [nvol, nvox] (planar: frame slowest == MRI.vol memory order);
    mask: uint8 CUDA tensor [nvox].  Returns dict of tensors: scalars [nvox], eigvecs [3, nvox]."""
    import torch
    _chk_dev(dwi, torch.float32, "dwi")
    _chk_dev(mask, torch.uint8, "mask")
    nvox = mask.numel()
    if dwi.numel() != nvox * plan.nvol:
        raise ValueError("dwi has %d elements, expected nvol*nvox = %d" % (dwi.numel(), nvox * plan.nvol))
    if out is None:
        out = {k: torch.empty((3, nvox) if "vec" in k else (nvox,), dtype=torch.float32, device=dwi.device)
               for k in DTI_FIELDS}
    o = _lib.DtiOut(*[out[k].data_ptr() for k in DTI_FIELDS])
    _lib.check(_lib.lib().fibd_dti_fit(plan._h, dwi.data_ptr(), mask.data_ptr(), nvox, C.byref(o), _stream_ptr(stream)))
    return out


def adc_fit_device(plan: DtiPlan, dwi, mask, stream=None):
    import torch
    _chk_dev(dwi, torch.float32, "dwi")
    _chk_dev(mask, torch.uint8, "mask")
    nvox = mask.numel()
    adc = torch.empty(nvox, dtype=torch.float32, device=dwi.device)
    s0 = torch.empty(nvox, dtype=torch.float32, device=dwi.device)
    _lib.check(_lib.lib().fibd_adc_fit(plan._h, dwi.data_ptr(), mask.data_ptr(), nvox, adc.data_ptr(), s0.data_ptr(),
                                       _stream_ptr(stream)))
    return adc, s0
