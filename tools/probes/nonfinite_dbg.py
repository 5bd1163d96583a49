import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
sys.path.insert(0, "/root/repo/oracle")
import oracle as orc
shape = (8, 6, 5)
bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed=23, noise_frac=0.02, crossing=True)
mask = np.ones(shape, np.uint8); mask[0, 0, 1] = 0
dwi[1, 0, 0, 7] = np.inf; dwi[2, 0, 0, 9] = -np.inf; dwi[3, 0, 0, 11] = np.nan
dwi[4, 0, 0, 5] = np.inf; dwi[4, 0, 0, 40] = np.inf
dwi[5, 0, 0, :] = -1.0; dwi[5, 0, 0, 3] = np.nan; dwi[6, 0, 0, :] = -np.inf; dwi[0, 0, 1, 2] = np.inf
sph = fj.sphere_642
with np.errstate(all="ignore"):
    ref = orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=2)
    got = fj.gqi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph)
ro, go = ref["odf"], got.odf.vol
for v in [(1,0,0),(2,0,0),(3,0,0),(4,0,0),(5,0,0),(6,0,0),(0,0,1),(7,0,0)]:
    r, g = ro[v], go[v]
    print(v, "ref nan %d +inf %d -inf %d | got nan %d +inf %d -inf %d | first %s %s" % (np.isnan(r).sum(), np.isposinf(r).sum(), np.isneginf(r).sum(),
          np.isnan(g).sum(), np.isposinf(g).sum(), np.isneginf(g).sum(), r[:3], g[:3]))
d = np.argwhere(np.isnan(ro) != np.isnan(go)); print(len(d), d[:10])
