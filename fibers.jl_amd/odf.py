"""ODF tessellations (`ODF`, `sphere_362/642/724`, odf.jl:5-11).  The numeric tables are carried as
data files extracted from the reference by tools/extract_spheres.py."""
import os
from dataclasses import dataclass

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


@dataclass(frozen=True)
class ODF:
    vertices: np.ndarray    # float32 [nverts, 3]; second half = -first half
    faces: np.ndarray       # int32 [nfaces, 3], 1-based like the reference

    @property
    def nvert(self):
        """vertices on the half sphere (gqi.jl:48)"""
        return self.vertices.shape[0] // 2


def _load(name):
    return ODF(np.load(os.path.join(_DATA, name + "_vertices.npy")),
               np.load(os.path.join(_DATA, name + "_faces.npy")))


sphere_362 = _load("sphere_362")
sphere_642 = _load("sphere_642")     # default in gqi_rec / dsi_rec
sphere_724 = _load("sphere_724")
