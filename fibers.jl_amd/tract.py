"""Host-side mirror of the reference's tractogram container (`Tract`, trk.jl:11-42).  Streamlines are kept
packed (one [npoints,3] array + per-line counts) — `str` materialises the reference's
Vector{Matrix{Float32}} view ([3 x npts] per line) on demand."""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np


@dataclass
class Tract:
    xyz: np.ndarray                       # float32 [npoints, 3], 1-based voxel coordinates (stream.jl:660)
    npts: np.ndarray                      # int32 [nstr]                                   (trk.jl:39)
    seed_index: Optional[np.ndarray] = None   # int64 [nstr]: seed*nsub + sub in the reference's loop order
    volsize: tuple = (0, 0, 0)            # trk.jl:16  dim
    volres: tuple = (1.0, 1.0, 1.0)       # trk.jl:17  voxel_size
    vox2ras: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))   # trk.jl:28
    sublist: Optional[np.ndarray] = None
    scalars: Optional[np.ndarray] = None  # float32 [npoints] or [npoints, n_scalars] (trk.jl:19,41): LCM runs store the
                                          # method-difference indicator of every point here (stream.jl:538, 787)
    properties: Optional[np.ndarray] = None   # float32 [nstr] or [nstr, n_properties] (trk.jl:21,42)

    @property
    def n_scalars(self) -> int:
        if self.scalars is None:
            return 0
        return 1 if np.ndim(self.scalars) == 1 else int(np.shape(self.scalars)[1])

    @property
    def n_properties(self) -> int:
        if self.properties is None:
            return 0
        return 1 if np.ndim(self.properties) == 1 else int(np.shape(self.properties)[1])

    @property
    def nstr(self) -> int:
        return int(self.npts.shape[0])

    @property
    def offsets(self) -> np.ndarray:
        return np.concatenate([[0], np.cumsum(self.npts, dtype=np.int64)])

    @property
    def str(self):
        """Vector{Matrix{Float32}}: one [3 x npts] matrix per streamline (trk.jl:40)"""
        off = self.offsets
        return [self.xyz[off[i]:off[i + 1]].T for i in range(self.nstr)]

    def line(self, i: int) -> np.ndarray:
        off = self.offsets
        return self.xyz[off[i]:off[i + 1]]
