"""Host-side mirror of the reference's volume container (`MRI`, mri.jl:80-130): only the fields the
hot path reads (`vol`, `bval`, `bvec`, geometry) — file I/O lives in nifti.py."""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np


@dataclass
class MRI:
    """`vol` is float32, Fortran-ordered [nx,ny,nz,nframes] exactly like `MRI.vol` (mri.jl:81)."""
    vol: np.ndarray
    bval: Optional[np.ndarray] = None            # [nframes]      (mri.jl:128)
    bvec: Optional[np.ndarray] = None            # [nframes, 3]   (mri.jl:129)
    volres: tuple = (1.0, 1.0, 1.0)              # voxel size, mm (mri.jl:93)
    vox2ras: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))   # vox2ras0 (mri.jl:104)
    tr: float = 0.0
    niftihdr: Optional[dict] = None              # header of the file this volume came from (mri.jl:126)

    def __post_init__(self):
        v = np.asanyarray(self.vol)                  # (asanyarray: an np.memmap stays one)
        if v.ndim == 3:
            v = v[..., None]
        if v.ndim != 4:
            raise ValueError("MRI.vol must be 3-D or 4-D")
        self.vol = v if v.flags.f_contiguous else np.asfortranarray(v)   # (a memory-mapped file stays a memory map)
        if self.bval is not None:
            self.bval = np.ascontiguousarray(self.bval, dtype=np.float32).reshape(-1)
        if self.bvec is not None:
            self.bvec = np.asfortranarray(np.asarray(self.bvec, dtype=np.float32).reshape(-1, 3))

    @property
    def volsize(self):
        return tuple(self.vol.shape[:3])

    @property
    def nframes(self):
        return self.vol.shape[3]

    @classmethod
    def like(cls, ref: "MRI", nframes: int = 1, dtype=np.float32) -> "MRI":
        """MRI(ref, nframes, T): zero-filled volume with ref's geometry (mri.jl:249-265)."""
        nx, ny, nz = ref.volsize
        return cls(np.zeros((nx, ny, nz, nframes), dtype=dtype, order="F"), volres=ref.volres,
                   vox2ras=ref.vox2ras.copy())
