"""Parity: HIP DTI/ADC fit (through the C ABI) vs the CPU oracle.  Reference: dti.jl:164-335."""
import numpy as np
import pytest

from util import assert_dti_close

pytestmark = pytest.mark.gpu


def _case(fj, shape, ndir, nb0, seed, nonpos=0.0, noise=0.02):
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_dti(ndir, nb0, 1000.0, seed)
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed, noise_frac=noise, nonpositive_frac=nonpos)
    rng = np.random.default_rng(seed + 100)
    mask = (rng.random(shape) < 0.8).astype(np.uint8)
    return dwi, mask, bval, bvec


@pytest.mark.parametrize("shape,ndir,nb0", [((16, 16, 16), 6, 1), ((16, 16, 16), 30, 3), ((7, 5, 3), 6, 1),
                                            ((10, 9, 7), 12, 2), ((32, 32, 32), 6, 1)])
def test_dti_fit_matches_oracle(fj, orc, shape, ndir, nb0):
    dwi, mask, bval, bvec = _case(fj, shape, ndir, nb0, seed=1)
    ref = orc.dti_fit(dwi, mask, bval, bvec, nthreads=4)
    got = fj.dti_fit(fj.MRI(dwi, bval, bvec), fj.MRI(mask))
    assert_dti_close({k: getattr(got, k).vol for k in fj.dti.DTI_FIELDS}, ref, mask, label=str(shape))


def test_dti_fit_partial_branch(fj, orc):
    """voxels with zero / negative samples take the per-voxel pinv branch (dti.jl:297-303)"""
    dwi, mask, bval, bvec = _case(fj, (12, 12, 12), 30, 3, seed=7, nonpos=0.02)
    dwi[0, 0, 0, :] = 0                      # all non-positive -> zeros
    dwi[1, 0, 0, :3] = -1                    # no positive b0 -> zeros
    mask[:2, 0, 0] = 1
    ref = orc.dti_fit(dwi, mask, bval, bvec, nthreads=4)
    assert ref["_npartial"] > 50
    got = fj.dti_fit(fj.MRI(dwi, bval, bvec), fj.MRI(mask))
    g = {k: getattr(got, k).vol for k in fj.dti.DTI_FIELDS}
    # float32-SVD pinv (oracle, like Julia) vs float64 normal equations (GPU): cond(A)*eps32 ~ 1e-4
    assert_dti_close(g, ref, mask, label="partial", s0_rtol=2e-3, ev_rtol=5e-3, ev_atol=2e-6, fa_atol=5e-3, vec_tol=1e-3, gap=0.2)
    for k in ("s0", "fa"):
        assert g[k][0, 0, 0] == 0 and g[k][1, 0, 0] == 0


def test_adc_fit_matches_oracle(fj, orc):
    dwi, mask, bval, bvec = _case(fj, (16, 16, 16), 30, 3, seed=3, nonpos=0.01)
    radc, rs0 = orc.adc_fit(dwi, mask, bval, nthreads=4)
    adc, s0 = fj.adc_fit(fj.MRI(dwi, bval, bvec), fj.MRI(mask))
    np.testing.assert_allclose(adc.vol[..., 0], radc, rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(s0.vol[..., 0], rs0, rtol=2e-3)


def test_dti_device_tier_and_errors(fj, orc):
    import torch
    dwi, mask, bval, bvec = _case(fj, (16, 16, 16), 6, 1, seed=5)
    plan = fj.DtiPlan(bval, bvec)
    A, pA = plan.tables()
    W = orc.dti_work(bval, bvec)
    np.testing.assert_allclose(A, W["A"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(pA, W["pA"], rtol=2e-3, atol=2e-7 * np.abs(W["pA"]).max())
    nvox = mask.size
    d = torch.from_numpy(np.ascontiguousarray(dwi.reshape(nvox, -1, order="F").T)).cuda()
    m = torch.from_numpy(mask.reshape(-1, order="F").copy()).cuda()
    out = fj.dti_fit_device(plan, d, m)
    torch.cuda.synchronize()
    ref = orc.dti_fit(dwi, mask, bval, bvec, nthreads=2)
    got = {k: (v.cpu().numpy().T.reshape(mask.shape + (3,), order="F") if v.dim() == 2
               else v.cpu().numpy().reshape(mask.shape, order="F")) for k, v in out.items()}
    assert_dti_close(got, ref, mask, label="device")
    assert plan.last_partial_count() == 0
    with pytest.raises(RuntimeError, match="Missing b-value table"):
        fj.dti_fit(fj.MRI(dwi), fj.MRI(mask))
    with pytest.raises(RuntimeError, match="Missing gradient table"):
        fj.dti_fit(fj.MRI(dwi, bval), fj.MRI(mask))


def test_dti_nonfinite_samples(fj, orc):
    """dti.jl:291-303 on samples that are not finite: `s .> 0` is false for NaN and -Inf (they drop out like any
    non-positive sample: the row-subset fit, a finite tensor), true for +Inf (log(Inf) = Inf reaches the fit: nothing
    finite comes out and nothing is trapped)."""
    dwi, mask, bval, bvec = _case(fj, (8, 8, 6), 30, 3, seed=11)
    mask[:4, 0, 0] = 1
    dwi[0, 0, 0, 9] = np.nan
    dwi[1, 0, 0, 12] = -np.inf
    dwi[2, 0, 0, 15] = np.inf
    dwi[3, 0, 0, :] = np.nan                  # nothing positive: zeros
    with np.errstate(all="ignore"):
        ref = orc.dti_fit(dwi, mask, bval, bvec, nthreads=2)
        got = fj.dti_fit(fj.MRI(dwi, bval, bvec), fj.MRI(mask))
    g = {k: getattr(got, k).vol for k in fj.dti.DTI_FIELDS}
    for k in ("s0", "fa", "md", "eigval1"):
        assert np.isfinite(ref[k][:2, 0, 0]).all() and np.isfinite(g[k][:2, 0, 0]).all(), k
        assert g[k][3, 0, 0] == 0 and ref[k][3, 0, 0] == 0, k
    # d = pA * log.(s) has +-Inf entries: S0 = exp(d7) is 0 or Inf, the tensor's eigen-decomposition is not finite
    assert np.array_equal(np.ravel(g["s0"][2, 0, 0]), np.ravel(ref["s0"][2, 0, 0]))
    for k in ("fa", "md", "eigval1"):
        assert not np.isfinite(g[k][2, 0, 0]).any() and not np.isfinite(ref[k][2, 0, 0]).any(), k
    ok = mask.astype(bool).copy()
    ok[2, 0, 0] = False                        # (+Inf voxel: NaN / Inf pattern of the eigen-solver, not compared)
    m2 = mask * ok
    for k in g:
        g[k] = np.where(ok[..., None] if g[k].ndim == 4 else ok, g[k], 0)
        ref[k] = np.where(ok[..., None] if np.ndim(ref[k]) == 4 else ok, ref[k], 0) if k in ref and hasattr(ref[k], "ndim") else ref[k]
    assert_dti_close(g, ref, m2, label="nonfinite", s0_rtol=2e-3, ev_rtol=5e-3, ev_atol=2e-6, fa_atol=5e-3, vec_tol=1e-3, gap=0.2)


def test_st_eigen_matches_oracle(fj, orc):
    """st_eigen (structens.jl:13-37): eigen(Symmetric(S,:L)) of every voxel's structure tensor -- ascending eigenvalues,
    eigenvectors as columns -- against the oracle's StaticArrays closed form, host and device tier."""
    import torch
    rng = np.random.default_rng(41)
    shape = (9, 7, 5)
    g = rng.normal(size=shape + (3,)).astype(np.float32)
    sm = [g[..., i] * g[..., j] for i, j in ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))]     # rank-1 tensors (as st_recon without smoothing)
    sm = [np.asfortranarray(v + 0.05 * rng.normal(size=shape).astype(np.float32) * (k in (0, 3, 5))) for k, v in enumerate(sm)]
    sm[0][0, 0, 0] = sm[3][0, 0, 0] = sm[5][0, 0, 0] = 2.0; sm[1][0, 0, 0] = sm[2][0, 0, 0] = sm[4][0, 0, 0] = 0.0   # isotropic: degenerate
    for v in sm: v[1, 0, 0] = 0.0                                                                                      # zero tensor
    rvec, rval = orc.st_eigen(*sm)
    gvec, gval = fj.st_eigen(*sm)
    assert gvec.shape == shape + (3, 3) and gval.shape == shape + (3,)
    np.testing.assert_allclose(gval, rval, rtol=2e-5, atol=2e-6)
    assert (np.diff(gval, axis=3) >= 0).all()
    gap = np.minimum(np.abs(rval[..., 1] - rval[..., 0]), np.abs(rval[..., 2] - rval[..., 1])) / (np.abs(rval).max(axis=3) + 1e-30)
    well = gap > 1e-2
    for j in range(3):
        dots = np.abs((gvec[..., :, j] * rvec[..., :, j]).sum(axis=3))
        assert dots[well].min() > 1 - 1e-4, j
        np.testing.assert_allclose(np.linalg.norm(gvec[..., :, j], axis=3)[well], 1.0, atol=1e-5)
    # A v = lambda v where the spectrum is well separated
    A = np.zeros(shape + (3, 3), np.float64)
    for (i, j), v in zip(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)), sm):
        A[..., i, j] = v; A[..., j, i] = v
    for j in range(3):
        res = np.einsum("...ik,...k->...i", A, gvec[..., :, j].astype(np.float64)) - gval[..., j:j + 1] * gvec[..., :, j]
        assert np.abs(res)[well].max() < 5e-5 * max(1.0, np.abs(A).max())
    dvec, dval = fj.st_eigen_device([torch.from_numpy(v.reshape(-1, order="F").copy()).cuda() for v in sm])
    torch.cuda.synchronize()
    assert np.array_equal(dval.cpu().numpy().reshape((3,) + shape[::-1]).transpose(3, 2, 1, 0), gval)
    assert np.array_equal(dvec.cpu().numpy().reshape((3, 3) + shape[::-1]).transpose(4, 3, 2, 1, 0), gvec, equal_nan=True)


@pytest.mark.parametrize("case", range(10))
def test_dti_and_adc_randomised_configurations(fj, orc, case):
    """Seeded random draws: volume shape (odd sizes), 6-48 directions, 1-5 b0 frames in random positions, b-values, mask
    density and dtype, fraction of non-positive samples (row-subset fits and zero voxels), noise level."""
    from fibers_jl_amd import phantom
    rng = np.random.default_rng(500 + case)
    shape = tuple(int(x) for x in rng.integers(3, 13, 3))
    ndir, nb0 = int(rng.integers(6, 49)), int(rng.integers(1, 6))
    bval, bvec = phantom.scheme_dti(ndir, nb0, float(rng.choice([700.0, 1000.0, 2500.0])), 500 + case)
    perm = rng.permutation(len(bval))                              # b0 frames anywhere in the table
    bval, bvec = np.ascontiguousarray(bval[perm]), np.ascontiguousarray(bvec[perm])
    nonpos = float(rng.choice([0.0, 0.0, 0.01, 0.05]))
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, 500 + case, noise_frac=float(rng.choice([0.0, 0.02, 0.08])), nonpositive_frac=nonpos)
    mask = (rng.random(shape) < rng.uniform(0.4, 1.0)).astype(rng.choice([np.uint8, np.int16, np.float32]))
    with np.errstate(all="ignore"):
        ref = orc.dti_fit(dwi, mask, bval, bvec, nthreads=3)
        got = fj.dti_fit(fj.MRI(dwi, bval, bvec), fj.MRI(mask))
    g = {k: getattr(got, k).vol for k in fj.dti.DTI_FIELDS}
    loose = dict(s0_rtol=2e-3, ev_rtol=5e-3, ev_atol=2e-6, fa_atol=5e-3, vec_tol=1e-3, gap=0.2) if nonpos > 0 else {}
    assert_dti_close(g, ref, (mask != 0).astype(np.uint8), label="case %d %s" % (case, shape), **loose)
    radc, rs0 = orc.adc_fit(dwi, mask, bval, nthreads=3)
    adc, s0 = fj.adc_fit(fj.MRI(dwi, bval, bvec), fj.MRI(mask))
    np.testing.assert_allclose(adc.vol[..., 0], radc, rtol=3e-3, atol=2e-7)
    np.testing.assert_allclose(s0.vol[..., 0], rs0, rtol=3e-3)
