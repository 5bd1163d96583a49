"""Structure-tensor eigen-decomposition (structens.jl:13-37) behind the C ABI: `st_eigen`.

The reference loops `eigen(Symmetric(S, :L))` over the voxels of six Float32 volumes; here one HIP kernel does it with
the 3x3 solver of the tensor fit (csrc/dti.hip).  `st_recon` (the Gaussian / Scharr filtering that produces the six
volumes, ImageFiltering.jl) is outside the hot path and not provided."""
import ctypes as C

import numpy as np

from . import _lib


def st_eigen(Sxx, Sxy, Sxz, Syy, Syz, Szz, device=0):
    """st_eigen(Sxx, Sxy, Sxz, Syy, Syz, Szz) -> (eigvec [nx,ny,nz,3,3], eigval [nx,ny,nz,3]), ascending eigenvalues,
    eigvec[..., :, j] the j-th eigenvector (structens.jl:13-37).  Float32 3-D arrays of one shape."""
    vols = [np.asfortranarray(v, dtype=np.float32) for v in (Sxx, Sxy, Sxz, Syy, Syz, Szz)]
    shape = vols[0].shape
    if len(shape) != 3 or any(v.shape != shape for v in vols):
        raise ValueError("st_eigen takes six 3-D arrays of one shape")
    nvox = int(np.prod(shape))
    eigvec = np.empty(shape + (3, 3), np.float32, order="F")
    eigval = np.empty(shape + (3,), np.float32, order="F")
    ptrs = (C.c_void_p * 6)(*[v.ctypes.data for v in vols])
    _lib.check(_lib.lib().fib_st_eigen(int(device), ptrs, nvox, eigvec.ctypes.data, eigval.ctypes.data))
    return eigvec, eigval


def st_eigen_device(S, stream=None):
    """Device tier: S = six float32 CUDA tensors [nvox]; returns (eigvec [9, nvox], eigval [3, nvox]) with
    eigvec[i + 3 j] = component i of eigenvector j."""
    import torch
    if len(S) != 6:
        raise ValueError("six volumes expected")
    for t in S:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("contiguous float32 CUDA tensors expected")
    nvox = S[0].numel()
    eigvec = torch.empty((9, nvox), dtype=torch.float32, device=S[0].device)
    eigval = torch.empty((3, nvox), dtype=torch.float32, device=S[0].device)
    ptrs = (C.c_void_p * 6)(*[t.data_ptr() for t in S])
    sp = None if stream is None else C.c_void_p(stream.cuda_stream)
    with torch.cuda.device(S[0].device):
        _lib.check(_lib.lib().fibd_st_eigen(ptrs, nvox, eigvec.data_ptr(), eigval.data_ptr(), sp))
    return eigvec, eigval
