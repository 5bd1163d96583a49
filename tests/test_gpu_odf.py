"""Parity: HIP GQI / DSI reconstruction + ODF peak finder (through the C ABI) vs the CPU oracle.
Reference: gqi.jl:32-201, dsi.jl:41-270."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gqi_case(shape, seed, nb0=3, ndir=20, shells=(1000.0, 2000.0, 3000.0), nonpos=0.0, crossing=True):
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_gqi(nb0, ndir, shells, seed)
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed, noise_frac=0.02, nonpositive_frac=nonpos, crossing=crossing)
    rng = np.random.default_rng(seed + 11)
    mask = (rng.random(shape) < 0.85).astype(np.uint8)
    return dwi, mask, bval, bvec


def _check_odf_rec(got_odf, got_peak, got_qa, ref, mask, odf_rtol=2e-5, qa_atol=1e-5, label=""):
    m = mask.astype(bool)
    ro = ref["odf"]
    scale = np.abs(ro).max(axis=3, keepdims=True) + 1e-30
    err = np.abs(got_odf - ro) / scale
    assert err.max() <= odf_rtol, "%s odf rel err %g" % (label, err.max())
    assert (got_odf[~m] == 0).all()
    nbad = 0
    for k in range(3):
        rp, gp = ref["peak"][k], got_peak[k]
        same = np.all(rp == gp, axis=3)
        nbad += (~same).sum()
        np.testing.assert_allclose(got_qa[k][same], ref["qa"][k][same], atol=qa_atol, rtol=1e-5)
    # peak indices may legitimately differ only where two amplitudes are within rounding of each other
    assert nbad <= max(1, int(2e-3 * m.sum())), "%s: %d peak mismatches" % (label, nbad)
    return nbad


@pytest.mark.parametrize("shape,sphere", [((8, 8, 8), "sphere_642"), ((9, 7, 5), "sphere_362"),
                                          ((6, 6, 6), "sphere_724"), ((16, 16, 12), "sphere_642")])
def test_gqi_rec_matches_oracle(fj, orc, shape, sphere):
    dwi, mask, bval, bvec = _gqi_case(shape, seed=3)
    sph = getattr(fj, sphere)
    ref = orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=4)
    got = fj.gqi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph)
    _check_odf_rec(got.odf.vol, [p.vol for p in got.peak], [q.vol[..., 0] for q in got.qa], ref, mask, label=sphere)


def test_gqi_matrix_matches_oracle(fj, orc):
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_gqi()
    plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25)
    W = orc.gqi_work(bval, bvec, fj.sphere_642.vertices, fj.sphere_642.faces, 1.25)
    np.testing.assert_allclose(plan.matrix(), W["A"], atol=6e-7, rtol=0)   # 1 ulp of the sinc argument


def test_gqi_skips_and_nonpositive(fj, orc):
    dwi, mask, bval, bvec = _gqi_case((8, 8, 8), seed=9, nonpos=0.05)
    dwi[0, 0, 0, :] = -1.0          # max(s) == 0 after clamping -> voxel skipped (gqi.jl:142)
    dwi[1, 0, 0, :] = 0.0
    mask[:2, 0, 0] = 1
    sph = fj.sphere_642
    ref = orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=2)
    got = fj.gqi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph)
    _check_odf_rec(got.odf.vol, [p.vol for p in got.peak], [q.vol[..., 0] for q in got.qa], ref, mask)
    assert (got.odf.vol[:2, 0, 0] == 0).all()


def test_find_peaks_exact_ties(fj, orc):
    """strict '>' against all face neighbours; '>=' ties kill both; stable order (gqi.jl:185-198)"""
    import torch
    sph = fj.sphere_642
    nvert = sph.nvert
    rng = np.random.default_rng(5)
    nvox = 257
    odf = rng.integers(0, 6, size=(nvert, nvox)).astype(np.float32)      # many exact ties
    odf[:, 0] = 0                                                        # all-zero voxel
    odf[:, 1] = -odf[:, 1]                                               # all non-positive
    odf[:, 2] = 1.0                                                      # constant
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_gqi(2, 6, (1000.0,), 1)
    plan = fj.OdfPlan("gqi", bval, bvec, sph)
    top, nvalid = fj.find_peaks_device(plan, torch.from_numpy(odf).cuda())
    torch.cuda.synchronize()
    top, nvalid = top.cpu().numpy(), nvalid.cpu().numpy()
    faces0 = orc.fold_faces(sph.faces, nvert)
    for v in range(nvox):
        isort, nv, _ = orc.find_peaks(odf[:, v], faces0)
        assert nv == nvalid[v], v
        assert list(isort[:3]) == list(top[:, v]), (v, isort[:3], top[:, v])


def _dsi_case(shape, seed):
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_dsi()
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed, noise_frac=0.01, crossing=True)
    rng = np.random.default_rng(seed)
    mask = (rng.random(shape) < 0.9).astype(np.uint8)
    return dwi, mask, bval, bvec


def test_dsi_matrix_matches_fft_chain(fj, orc):
    """the dense maps equal the reference's scatter->Hanning->FFT->trilinear chain applied to unit samples"""
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_dsi()
    sph = fj.sphere_642
    plan = fj.OdfPlan("dsi", bval, bvec, sph, hann_width=32)
    A = plan.matrix()
    assert A.shape == (515 + 321, 515)
    # drive the oracle with one-hot signals (plus the b0 sample so that sum(p) = 4096*s_b0 is finite)
    cols = [1, 7, 100, 333, 514]
    nvox = len(cols)
    dwi = np.zeros((nvox, 1, 1, 515), np.float32, order="F")
    for i, c in enumerate(cols):
        dwi[i, 0, 0, 0] = 1.0
        dwi[i, 0, 0, c] = 1.0
    ref = orc.dsi_rec(dwi, np.ones((nvox, 1, 1)), bval, bvec, sph.vertices, sph.faces, 32)
    for i, c in enumerate(cols):
        want = np.concatenate([ref["pdf"][i, 0, 0], ref["odf"][i, 0, 0]]) * 4096.0
        have = A[:, 0] + A[:, c]
        np.testing.assert_allclose(have, want, atol=3e-5 * np.abs(want).max(), rtol=0)


def test_dsi_rec_matches_oracle(fj, orc):
    dwi, mask, bval, bvec = _dsi_case((5, 4, 3), seed=5)
    dwi[0, 0, 0, :] = 0
    mask[0, 0, 0] = 1
    sph = fj.sphere_642
    ref = orc.dsi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 32, nthreads=4)
    got = fj.dsi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph, 32)
    scale = np.abs(ref["pdf"]).max(axis=3, keepdims=True) + 1e-30
    assert (np.abs(got.pdf.vol - ref["pdf"]) / scale).max() < 5e-5
    _check_odf_rec(got.odf.vol, [p.vol for p in got.peak], [q.vol[..., 0] for q in got.qa], ref, mask,
                   odf_rtol=1e-4, qa_atol=1e-4, label="dsi")
    assert (got.pdf.vol[0, 0, 0] == 0).all()


def test_odf_device_tier_unnormalised(fj, orc):
    import torch
    dwi, mask, bval, bvec = _gqi_case((8, 8, 8), seed=21)
    sph = fj.sphere_642
    plan = fj.OdfPlan("gqi", bval, bvec, sph)
    nvox = mask.size
    d = torch.from_numpy(np.ascontiguousarray(dwi.reshape(nvox, -1, order="F").T)).cuda()
    m = torch.from_numpy(mask.reshape(-1, order="F").copy()).cuda()
    out = fj.odf_rec_device(plan, d, m, normalize=False)
    torch.cuda.synchronize()
    ref = orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=2)
    odfmax = float(out["odfmax"][0])
    assert abs(odfmax - ref["odfmax"]) <= 2e-6 * abs(ref["odfmax"])
    fj.qa_normalize_device(out["qa"], odfmax)
    torch.cuda.synchronize()
    qa0 = out["qa"][0].cpu().numpy().reshape(mask.shape, order="F")
    same = np.all(out["peak"][0].cpu().numpy().T.reshape(mask.shape + (3,), order="F") == ref["peak"][0], axis=3)
    np.testing.assert_allclose(qa0[same], ref["qa"][0][same], atol=1e-5, rtol=1e-5)
