"""CPU tests (no GPU): the oracle against analytic known answers and float64 closed forms.
The reference has no tests or golden vectors (test/runtests.jl:4-6), so these are what pins the oracle."""
import numpy as np
import pytest


def _rot(axis):
    a = np.asarray(axis, float); a /= np.linalg.norm(a)
    t = np.array([1.0, 0, 0]) if abs(a[0]) < 0.9 else np.array([0, 1.0, 0])
    b = np.cross(a, t); b /= np.linalg.norm(b)
    return np.stack([a, b, np.cross(a, b)], 1)


def test_sym3_eigen_matches_float64_eigh(orc):
    rng = np.random.default_rng(0)
    for _ in range(200):
        M = rng.normal(size=(3, 3)); M = (M + M.T) / 2 * 1e-3
        w, v = orc.sym3_eigen(M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2])
        w64, v64 = np.linalg.eigh(M)
        assert np.all(np.diff(w) >= 0)                               # ascending (dti.jl:313-314 relies on it)
        assert np.abs(w - w64).max() <= 2e-6 * np.abs(w64).max()
        assert np.abs(np.abs((v * v64).sum(0)) - 1).max() < 1e-4
        assert np.abs(v.T @ v - np.eye(3)).max() < 1e-5
    # diagonal input takes the sorting branch
    w, v = orc.sym3_eigen(3e-3, 0, 0, 1e-3, 0, 2e-3)
    assert np.allclose(w, [1e-3, 2e-3, 3e-3]) and np.array_equal(np.abs(v), np.eye(3)[:, [1, 2, 0]])


def test_pinv32_matches_numpy(orc):
    rng = np.random.default_rng(1)
    A = rng.normal(size=(30, 7)).astype(np.float32)
    assert np.abs(orc.pinv32(A) - np.linalg.pinv(A.astype(np.float64))).max() < 1e-5
    A[:, 3] = A[:, 2]                                                # rank deficient: cut-off path
    assert np.abs(orc.pinv32(A) - np.linalg.pinv(A.astype(np.float64), rcond=1e-6)).max() < 1e-4


def test_dti_noise_free_tensor_recovery(orc, fj):
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_dti(30, 3)
    R = _rot([0.6, 0.64, 0.48])
    D = R @ np.diag([1.7e-3, 0.4e-3, 0.2e-3]) @ R.T
    s = 1000 * np.exp(-bval.astype(float) * np.einsum("ij,jk,ik->i", bvec, D, bvec))
    dwi = np.broadcast_to(s.astype(np.float32), (3, 2, 2, len(s))).copy(order="F")
    out = orc.dti_fit(dwi, np.ones((3, 2, 2)), bval, bvec, nthreads=2)
    assert np.allclose(out["s0"], 1000, rtol=1e-4)
    for k, v in zip(("eigval1", "eigval2", "eigval3"), (1.7e-3, 0.4e-3, 0.2e-3)):
        assert np.allclose(out[k], v, atol=2e-6), (k, out[k].ravel()[0])
    md = (1.7e-3 + 0.4e-3 + 0.2e-3) / 3
    fa = np.sqrt(1.5 * ((1.7e-3 - md) ** 2 + (0.4e-3 - md) ** 2 + (0.2e-3 - md) ** 2) / (1.7e-3 ** 2 + 0.4e-3 ** 2 + 0.2e-3 ** 2))
    assert np.allclose(out["fa"], fa, atol=1e-3) and np.allclose(out["md"], md, atol=1e-6)
    assert np.allclose(out["rd"], 0.3e-3, atol=2e-6)
    assert np.abs(np.abs(out["eigvec1"][0, 0, 0] @ R[:, 0]) - 1) < 1e-4
    assert np.abs(np.abs(out["eigvec3"][0, 0, 0] @ R[:, 2]) - 1) < 1e-3


def test_dti_isotropic_and_degenerate_voxels(orc, fj):
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_dti(6, 1)
    s = (1000 * np.exp(-bval * 1e-3)).astype(np.float32)
    dwi = np.zeros((4, 1, 1, 7), np.float32, order="F")
    dwi[0, 0, 0] = s                    # isotropic -> FA ~ 0
    dwi[1, 0, 0] = 500.0                # constant signal -> D ~ 0 up to rounding of pA*log(s)
    dwi[2, 0, 0] = s; dwi[2, 0, 0, 3] = 0      # 6 positive of 7 -> npos > 6 fails -> zeros (dti.jl:297-303)
    dwi[3, 0, 0] = s                    # masked out
    mask = np.array([1, 1, 1, 0]).reshape(4, 1, 1)
    out = orc.dti_fit(dwi, mask, bval, bvec)
    assert out["fa"][0, 0, 0] < 2e-3 and abs(out["md"][0, 0, 0] - 1e-3) < 1e-6
    assert abs(out["eigval1"][1, 0, 0]) < 1e-7 and abs(out["s0"][1, 0, 0] - 500) < 0.1
    o = orc.dti_from_d(np.zeros(7, np.float32))     # D == 0 exactly -> 0/0 = NaN FA (dti.jl:331, not trapped)
    assert np.isnan(o[15]) and o[0] == 1 and np.all(o[1:4] == 0) and np.all(o[13:15] == 0)
    for k in ("s0", "eigval1", "fa", "md"):
        assert out[k][2, 0, 0] == 0 and out[k][3, 0, 0] == 0
    assert np.all(out["eigvec1"][2:, 0, 0] == 0)


def test_adc_known_answer(orc, fj):
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_dti(12, 2)
    s = (800 * np.exp(-bval * 0.9e-3)).astype(np.float32)
    dwi = np.broadcast_to(s, (2, 2, 2, len(s))).copy(order="F")
    adc, s0 = orc.adc_fit(dwi, np.ones((2, 2, 2)), bval)
    assert np.allclose(adc, 0.9e-3, rtol=1e-4) and np.allclose(s0, 800, rtol=1e-4)


def test_find_peaks_semantics(orc, fj):
    sph = fj.sphere_642
    n = sph.nvert
    f0 = orc.fold_faces(sph.faces, n)
    nbrs = [set() for _ in range(n)]
    for a, b, c in f0:
        nbrs[a] |= {b, c}; nbrs[b] |= {a, c}; nbrs[c] |= {a, b}
    assert max(len(s) for s in nbrs) == 6 and min(len(s) for s in nbrs) >= 5
    o = np.zeros(n, np.float32)
    o[10] = 2.0; o[200] = 3.0; o[300] = 1.0
    isort, nv, pk = orc.find_peaks(o, f0)
    assert nv == 3 and list(isort[:3]) == [200, 10, 300]
    u = next(iter(nbrs[10]))
    o[u] = 2.0                                   # '>=' tie kills BOTH neighbours (gqi.jl:185-196)
    isort, nv, pk = orc.find_peaks(o, f0)
    assert nv == 2 and list(isort[:2]) == [200, 300] and pk[10] == 0 and pk[u] == 0
    o[:] = 1.0                                   # constant ODF: no strict maximum at all
    assert orc.find_peaks(o, f0)[1] == 0
    o[:] = 0; o[5] = 1.0; o[50] = 1.0            # equal peaks: stable sort keeps the lower index first
    assert list(orc.find_peaks(o, f0)[0][:2]) == [5, 50]
    o[:] = -1.0; o[7] = -0.5                     # a negative local maximum survives but is not "valid" (gqi.jl:200)
    isort, nv, pk = orc.find_peaks(o, f0)
    assert nv == 0 and pk[7] == -0.5 and isort[-1] == 7


@pytest.mark.parametrize("sphere", ["sphere_362", "sphere_642", "sphere_724"])
def test_gqi_single_fibre_peak_is_nearest_vertex(orc, fj, sphere):
    from fibers_jl_amd import phantom
    sph = getattr(fj, sphere)
    bval, bvec = phantom.scheme_gqi()
    ax = np.array([0.6, 0.64, 0.48]); ax /= np.linalg.norm(ax)
    shape = (2, 2, 2)
    dwi = phantom.signal(bval, bvec, [np.broadcast_to(ax, shape + (3,))], [np.ones(shape)], np.full(shape, 1000.0))
    r = orc.gqi_rec(dwi, np.ones(shape), bval, bvec, sph.vertices, sph.faces, 1.25)
    V = sph.vertices[: sph.nvert]
    p = r["peak"][0][0, 0, 0]
    best = np.abs(V @ ax).max()
    assert abs(p @ ax) >= best - 0.02                      # within one vertex spacing of the optimum
    assert any(np.array_equal(p, v) for v in V)            # a FIRST-half vertex row (gqi.jl:155)
    o = r["odf"][0, 0, 0]
    assert np.isclose(r["qa"][0][0, 0, 0] * r["odfmax"], o.max() - o.min(), rtol=1e-5)
    assert np.isclose(r["odfmax"], o.mean(), rtol=1e-5)


def test_dsi_known_answers(orc, fj):
    from fibers_jl_amd import phantom
    sph = fj.sphere_642
    bval, bvec = phantom.scheme_dsi()
    assert len(bval) == 515
    W = orc.dsi_work(bval, bvec, sph.vertices, sph.faces, 32)
    assert W["nfft"] == 16 and W["iq"].min() == -5 and W["iq"].max() == 5 and len(np.unique(W["iq_ind"])) == 515
    assert np.isclose(W["H"][W["iq_ind"][0]], 1.0) and np.isclose(W["qr"][0], 2.1) and np.isclose(W["qr"][-1], 6.3)
    ax = np.array([0.6, 0.64, 0.48]); ax /= np.linalg.norm(ax)
    shape = (2, 1, 1)
    dwi = phantom.signal(bval, bvec, [np.broadcast_to(ax, shape + (3,))], [np.ones(shape)], np.full(shape, 1000.0))
    dwi[1] = 0                                             # max(X) == 0 -> skipped (dsi.jl:207)
    r = orc.dsi_rec(dwi, np.ones(shape), bval, bvec, sph.vertices, sph.faces, 32)
    assert abs(r["peak"][0][0, 0, 0] @ ax) > 0.99
    assert (r["pdf"][1] == 0).all() and (r["odf"][1] == 0).all()
    # the pdf is the real part of the DFT of the windowed signal: check two lattice points in float64
    X = np.zeros((16, 16, 16))
    iq = W["iq"]
    X[iq[:, 0] + 8, iq[:, 1] + 8, iq[:, 2] + 8] = dwi[0, 0, 0].astype(float) * W["H"][W["iq_ind"]]
    P = np.fft.fftshift(np.fft.fftn(np.fft.ifftshift(X))).real
    P /= P.sum()
    want = P[iq[:, 0] + 8, iq[:, 1] + 8, iq[:, 2] + 8]
    assert np.abs(r["pdf"][0, 0, 0] - want).max() < 2e-6 * np.abs(want).max() + 1e-9


def test_stream_uniform_field_known_lines(orc):
    n = 10
    ov = np.zeros((n, n, n, 3), np.float32); ov[..., 0] = 1
    sub = np.array([[0.25, 0.0, 0.0]], np.float32)
    mask = np.ones((n, n, n), np.uint8)
    res = orc.stream(ov, sub, mask=mask, len_max=100, return_all_npts=True)
    lines = orc.split_lines(res)
    # seed (1,1,1) at x=1.25: forward points x = 1.25, 1.75, ... while round(x+.5) <= 10; backward until round(x-.5) >= 1
    l0 = lines[0]
    fwd = [1.25 + 0.5 * i for i in range(100) if np.rint(1.25 + 0.5 * (i + 1)) <= n]
    bwd = [1.25 - 0.5 * i for i in range(100) if np.rint(1.25 - 0.5 * (i + 1)) >= 1]
    assert np.allclose(l0[:, 0], fwd[::-1] + bwd) and np.all(l0[:, 1:] == 1.0)
    assert l0[len(fwd) - 1, 0] == l0[len(fwd), 0] == 1.25          # seed emitted once per direction
    # round-half-to-even at the voxel boundary (stream.jl:514): 9.75+.5 = 10.25 -> 10 ok; a half-integer decides by parity
    res2 = orc.stream(ov, np.array([[0.0, 0.0, 0.0]], np.float32), mask=mask, len_max=100)
    l = orc.split_lines(res2)[0]
    assert l[0, 0] == 10.0      # 10.0+.5 = 10.5 -> rounds to 10 (even) -> still inside; 10.5+.5 = 11 -> out
    # len_max+2 cap and len_min filter
    res3 = orc.stream(ov, sub, mask=mask, len_max=4, len_min=7, return_all_npts=True)
    assert res3["all_npts"].max() == 6 and len(res3["npts"]) == 0
    # mask hole stops the line; zero vector inside the mask stops it the same way
    m2 = mask.copy(); m2[5, 0, 0] = 0
    a = orc.split_lines(orc.stream(ov, sub, mask=m2, len_max=100))[0]
    ov2 = ov.copy(); ov2[5, 0, 0] = 0
    b = orc.split_lines(orc.stream(ov2, sub, mask=mask, seed=m2, len_max=100))[0]
    assert np.array_equal(a, b) and a[:, 0].max() == 4.75


def test_stream_angle_threshold_and_vector_choice(orc):
    n = 8
    ov1 = np.zeros((n, n, n, 3), np.float32); ov1[..., 0] = 1
    ov2 = np.zeros((n, n, n, 3), np.float32); ov2[..., 1] = 1
    ov1[4:, :, :, :] = 0; ov1[4:, :, :, 1] = 1          # sharp 90 degree bend at x=5 -> exceeds 45 degrees
    res = orc.stream(ov1, np.zeros((1, 3), np.float32), mask=np.ones((n, n, n)), smooth_coeff=0.0, len_max=50)
    l0 = orc.split_lines(res)[0]
    assert l0[:, 0].max() == 4.5 and np.all(l0[:, 1] == 1)   # the point before the bend is saved, then the line stops
    # two vectors per voxel: the one closest to the current direction is followed, sign-corrected
    res = orc.stream([ov2, -np.roll(ov1, 0)], np.zeros((1, 3), np.float32), mask=np.ones((n, n, n)), len_max=50)
    l0 = orc.split_lines(res)[0]
    assert np.all(np.diff(l0[:, 1]) <= 0) and np.all(l0[:, 0] == 1) and l0[:, 1].max() >= n - 0.5


def test_micro_regime_uniform_field_known_path():
    """microscopy regime (stream.jl:547-619) on a uniform +x field: every cell of the search cone ties at |cos| = 1, and
    argmax takes the FIRST maximum in the cube's column-major order = the smallest x.  Going along +x that is the
    tentative voxel itself (one voxel per step); going along -x it is the farthest cone cell (search_dist + 1 voxels per
    step).  Positions snap to voxel centres."""
    from oracle import oracle as orc
    n, d = 20, 3
    ov = np.zeros((n, n, n, 3), np.float32, order="F")
    ov[..., 0] = 1
    mask = np.ones((n, n, n), np.uint8)
    seed = np.zeros((n, n, n), np.uint8)
    seed[9, 9, 9] = 1                                         # 1-based voxel (10, 10, 10)
    r = orc.stream(ov, np.zeros((1, 3), np.float32), mask=mask, seed=seed, ang_thresh=20, step_size=1.0,
                   smooth_coeff=0.0, search_dist=d, search_ang=10, len_max=40)
    fwd = [[x, 10, 10] for x in range(19, 9, -1)]             # forward points 10..19, prepended (stream.jl:652)
    bwd = [[10, 10, 10], [6, 10, 10], [2, 10, 10]]            # 10 -> 6 -> 2 -> (1) -> out of the volume
    assert r["npts"].tolist() == [13]
    assert np.array_equal(r["xyz"], np.array(fwd + bwd, np.float32))
    # the cone: with search_ang = 50 degrees off-axis cells tie too; the first in column-major order has the smallest z
    r2 = orc.stream(ov, np.zeros((1, 3), np.float32), mask=mask, seed=seed, ang_thresh=20, step_size=1.0,
                    smooth_coeff=0.0, search_dist=2, search_ang=50, len_max=40)
    step1 = r2["xyz"][r2["npts"][0] - 0 - 1] if False else None
    xyz = r2["xyz"]
    i10 = [i for i in range(len(xyz)) if tuple(xyz[i]) == (10.0, 10.0, 10.0)]
    assert len(i10) == 2                                       # the seed point opens both directions
    assert xyz[i10[0] - 1][2] < 10.0                           # first forward move drops in z (smallest z wins the tie)


def test_rumba_oracle_pieces():
    """RUMBA-SD oracle (rusd.jl): Perron's continued fraction vs the Bessel functions it approximates; the kernel's
    columns against the closed-form single-tensor signal; the TV term of a constant volume is 1; single-fibre recovery"""
    from scipy.special import ive
    from oracle import oracle as orc
    from fibers_jl_amd import phantom, sphere_724
    z = np.geomspace(1e-3, 200.0, 200).astype(np.float32)
    approx = orc._besseli_ratio(1, z)
    exact = ive(1, z.astype(np.float64)) / ive(0, z.astype(np.float64))
    assert np.abs(approx - exact).max() < 1e-2                   # the reference's 4-term truncation (rusd.jl:170-177) is this coarse (8.5e-3 near z ~ 1)
    bval, bvec = phantom.scheme_gqi(2, 40, (1500.0, 3000.0), 7)
    K, ib0 = orc.rumba_kernel(bval, bvec, sphere_724.vertices)
    assert K.shape == (int((~ib0).sum()) + 1, 362 + 2) and np.all(K[0] == 1.0)
    g = bvec[~ib0] / np.linalg.norm(bvec[~ib0], axis=1, keepdims=True)
    u = sphere_724.vertices[362 + 17].astype(np.float64)          # second-half vertex of column 17 (rusd.jl:502-504)
    want = np.exp(-bval[~ib0] * (0.2e-3 + (1.7e-3 - 0.2e-3) * (g @ u) ** 2))
    np.testing.assert_allclose(K[1:, 17], want, rtol=2e-5)
    np.testing.assert_allclose(K[1:, 362], np.exp(-bval[~ib0] * 3.0e-3), rtol=2e-6)
    tv = orc._rumba_tv(np.full((4, 5, 6), 0.3, np.float32), np.full((4, 5, 6), 0.01, np.float32))
    np.testing.assert_allclose(tv, 1.0, rtol=1e-6)
    ax = np.array([0.6, -0.48, 0.64])
    sig = (800.0 * np.exp(-bval * (0.2e-3 + 1.5e-3 * ((bvec / np.maximum(np.linalg.norm(bvec, axis=1, keepdims=True), 1e-12)) @ ax) ** 2)))
    dwi = np.asfortranarray(np.tile(sig.astype(np.float32), (2, 2, 2, 1)))
    r = orc.rumba_rec(dwi, np.ones((2, 2, 2), np.uint8), bval, bvec, sphere_724.vertices, niter=150, use_tv=False)
    pk = r["peak"][0][0, 0, 0]
    assert abs(pk @ ax) / np.linalg.norm(pk) > 0.985              # nearest vertex of a 362-point half sphere
    assert abs(r["fodf"][0, 0, 0].sum() - 1.0) < 1e-5
