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


def _verts_by_count():
    import fibers_jl_amd as fj
    return {s.nvert: np.asarray(s.vertices, np.float32) for s in (fj.sphere_362, fj.sphere_642, fj.sphere_724)}


class _Lazy(dict):
    def __missing__(self, k):
        self.update(_verts_by_count())
        return dict.__getitem__(self, k)


_VERTS = _Lazy()


class _LazyFaces(dict):
    def __missing__(self, k):
        import fibers_jl_amd as fj
        self.update({s.nvert: np.asarray(s.faces) for s in (fj.sphere_362, fj.sphere_642, fj.sphere_724)})
        return dict.__getitem__(self, k)


_FACES = _LazyFaces()


def _check_odf_rec(got_odf, got_peak, got_qa, ref, mask, odf_rtol=2e-5, qa_atol=1e-5, label=""):
    m = mask.astype(bool)
    ro = ref["odf"]
    scale = np.abs(ro).max(axis=3, keepdims=True) + 1e-30
    err = np.abs(got_odf - ro) / scale
    assert err.max() <= odf_rtol, "%s odf rel err %g" % (label, err.max())
    assert (got_odf[~m] == 0).all()
    for k in range(3):
        rp, gp = ref["peak"][k], got_peak[k]
        same = np.all(rp == gp, axis=3)
        np.testing.assert_allclose(got_qa[k][same], ref["qa"][k][same], atol=qa_atol, rtol=1e-5)
    # peak vertices may differ only where the oracle's amplitudes of the two vertices are within 1e-4 of the voxel maximum of each
    # other: every mismatch is checked, there is no allowance by count
    from util import peak_mismatches_are_ties
    nv = ref["odf"].shape[3]
    return peak_mismatches_are_ties(ref["odf"], ref["peak"], got_peak, _VERTS[nv][:nv], faces=_FACES[nv])


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


def test_sparse_mask_tile_skipping(fj, orc):
    """ball mask inside a larger volume: whole waves / workgroups / tiles lie outside the mask and are skipped"""
    from fibers_jl_amd import phantom
    shape = (40, 24, 20)
    bval, bvec = phantom.scheme_gqi(2, 12, (1000.0, 2500.0), 3)
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed=12, crossing=True)
    mask = phantom.ball_mask(*shape, radius=7.5)
    assert 0.02 < mask.mean() < 0.2
    sph = fj.sphere_642
    ref = orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=4)
    got = fj.gqi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph)
    _check_odf_rec(got.odf.vol, [p.vol for p in got.peak], [q.vol[..., 0] for q in got.qa], ref, mask, label="ball")
    assert all((q.vol[..., 0][mask == 0] == 0).all() for q in got.qa)
    b2, g2 = phantom.scheme_dti(12, 2)
    d2, _, _ = phantom.make_volume(shape, b2, g2, seed=13)
    r2 = orc.dti_fit(d2, mask, b2, g2, nthreads=4)
    t2 = fj.dti_fit(fj.MRI(d2, b2, g2), fj.MRI(mask))
    from util import assert_dti_close
    assert_dti_close({k: getattr(t2, k).vol for k in fj.dti.DTI_FIELDS}, r2, mask, label="ball")


@pytest.mark.gpu
def test_mask_compaction_device_tier(fj):
    """device tier: outputs outside the mask are zero-filled over stale buffers; FIB_ODF_PREZEROED skips that fill and
    gives the same volumes when the buffers are already zero there; an empty mask yields all zeros (odfmax 0);
    a scattered mask gives the same per-voxel results as the all-ones run (compaction only moves columns)"""
    import torch
    from fibers_jl_amd import phantom
    dev = torch.device("cuda", 0)
    shape = (37, 11, 9)                                    # odd voxel count: ragged tiles and unaligned rows
    nvox = shape[0] * shape[1] * shape[2]
    bval, bvec = phantom.scheme_gqi(2, 10, (1000.0, 2000.0), 5)
    dwi_h, _, _ = phantom.make_volume(shape, bval, bvec, seed=3, crossing=True)
    dwi = torch.from_numpy(np.ascontiguousarray(np.moveaxis(dwi_h, -1, 0).reshape(len(bval), -1))).to(dev)
    # same linearisation for every run, so any fixed order works for this comparison
    plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642)
    ones = torch.ones(nvox, dtype=torch.uint8, device=dev)
    full = fj.odf_rec_device(plan, dwi, ones, normalize=False)
    rng = np.random.default_rng(5)
    m_h = (rng.random(nvox) < 0.3).astype(np.uint8)
    m_h[:300] = 0                                          # whole tiles outside the mask
    m = torch.from_numpy(m_h).to(dev)
    out = fj.odf_rec_device(plan, dwi, m, normalize=False)
    stale = {k: ([t.clone().fill_(7.0) for t in v] if isinstance(v, list) else v.clone().fill_(7.0)) for k, v in out.items()}
    got = fj.odf_rec_device(plan, dwi, m, out=stale, normalize=False)
    live = m.bool()
    for a, b, c in [(got["odf"], out["odf"], full["odf"])] + [(got["peak"][k], out["peak"][k], full["peak"][k]) for k in range(3)] \
            + [(got["qa"][k], out["qa"][k], full["qa"][k]) for k in range(3)]:
        assert torch.equal(a, b)
        assert (a.reshape(-1, nvox)[:, ~live] == 0).all()
        assert torch.equal(a.reshape(-1, nvox)[:, live], c.reshape(-1, nvox)[:, live])
    pre = fj.odf_rec_device(plan, dwi, m, out=got, normalize=False, out_prezeroed=True)
    assert torch.equal(pre["odf"], out["odf"]) and all(torch.equal(pre["qa"][k], out["qa"][k]) for k in range(3))
    seq = torch.zeros(nvox, device=dev)
    for r in range(out["odf"].shape[0]):                   # the reference's mean: sequential f32 sum over the vertices, then ./ n
        seq = seq + out["odf"][r]
    assert float(got["odfmax"][0]) == float(torch.maximum(torch.div(seq, torch.tensor(float(out["odf"].shape[0]), device=dev)).max(),
                                                          torch.zeros((), device=dev)))
    empty = fj.odf_rec_device(plan, dwi, torch.zeros_like(m), out=stale, normalize=False)
    assert float(empty["odf"].abs().max()) == 0.0 and float(empty["odfmax"][0]) == 0.0
    assert all(float(q.abs().max()) == 0.0 for q in empty["qa"])


FORMATS = ["fp16x2", "bf16x3", "f32"]        # the operand formats of the contraction (include/fibers_hip.h FIB_ODF_FORMAT_*)


def _use_format(monkeypatch, fmt):
    """make `fmt` the default operand format of the plans built from here on (the host tier builds its plans with the default
    format and keys its plan cache on it): FIBERS_ODF_FORMAT = fp16x2 (or unset) | bf16x3 | f32"""
    import fibers_jl_amd as fj
    from fibers_jl_amd import _lib
    monkeypatch.delenv("FIBERS_ODF_FORMAT", raising=False)
    if fmt != "fp16x2":
        monkeypatch.setenv("FIBERS_ODF_FORMAT", fmt)
    assert _lib.lib().fib_odf_default_format() == fj.gqi.ODF_FORMATS[fmt]


@pytest.mark.parametrize("mode", FORMATS)
def test_gqi_all_operand_formats_full_mask(fj, orc, mode, monkeypatch):
    """the three operand formats of the contraction (two fp16 pieces, three exact bf16 pieces, f32 MFMA chain), each directly
    against the oracle on an all-ones mask: contiguous voxel runs take the LDS-transposed dwordx4 epilogue, the ragged last work
    item the scalar one"""
    _use_format(monkeypatch, mode)
    from fibers_jl_amd import phantom
    shape = (12, 10, 9)                                    # 1080 voxels: a multiple of 4, not of 256
    bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed=17, noise_frac=0.02, crossing=True)
    mask = np.ones(shape, np.uint8)
    sph = fj.sphere_642
    ref = orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=4)
    got = fj.gqi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph)
    _check_odf_rec(got.odf.vol, [p.vol for p in got.peak], [q.vol[..., 0] for q in got.qa], ref, mask, label=mode)


@pytest.mark.parametrize("mode", FORMATS)
def test_dsi_all_operand_formats_full_mask(fj, orc, mode, monkeypatch):
    """DSI (folded lattice: mapped pdf rows, ODF tile + pdf tile per voxel group) on an all-ones mask, every operand format
    directly against the oracle"""
    _use_format(monkeypatch, mode)
    dwi, _, bval, bvec = _dsi_case((8, 6, 2), seed=8)      # 96 voxels: one full 32-voxel wave run + ragged rest
    mask = np.ones(dwi.shape[:3], np.uint8)
    sph = fj.sphere_642
    ref = orc.dsi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 32, nthreads=4)
    got = fj.dsi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph, 32)
    scale = np.abs(ref["pdf"]).max(axis=3, keepdims=True) + 1e-30
    assert (np.abs(got.pdf.vol - ref["pdf"]) / scale).max() < 5e-5
    _check_odf_rec(got.odf.vol, [p.vol for p in got.peak], [q.vol[..., 0] for q in got.qa], ref, mask,
                   odf_rtol=1e-4, qa_atol=1e-4, label="dsi " + mode)


def test_split_formats_match_f32_kernel_large(fj):
    """40^3 x 63 frames: each piece-split contraction (three bf16 pieces, two fp16 pieces) agrees with the f32-MFMA chain to f32 rounding everywhere, with and
    without a mask (the compacted voxel list changes which lanes / work items a voxel lands in)"""
    import torch
    from fibers_jl_amd import phantom
    dev = torch.device("cuda", 0)
    shape = (40, 40, 40)
    nvox = 40 ** 3
    bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
    dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=4, device=dev)
    ball = phantom.ball_mask_torch(shape, dev, radius=17.3)
    outs = {}
    for mode in FORMATS:
        plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, format=mode)
        assert plan.format == mode
        for name, m in (("ones", torch.ones(nvox, dtype=torch.uint8, device=dev)), ("ball", ball)):
            o = fj.odf_rec_device(plan, dwi, m)
            outs[mode, name] = (o["odf"].clone(), o["peak"][0].clone())
        plan.close()
    live = ball.bool()
    for split in ("bf16x3", "fp16x2"):
        for name in ("ones", "ball"):
            a, b = outs["f32", name][0], outs[split, name][0]
            vmax = a.abs().max(0).values.clamp_min(1e-30)
            assert float(((a - b).abs() / vmax).max()) < 5e-6, split
            differ = (outs["f32", name][1] != outs[split, name][1]).any(0).float().mean()
            assert float(differ) < 1e-3                    # first-peak vertex: only amplitude ties at rounding level may flip
        assert torch.equal(outs[split, "ones"][0][:, live], outs[split, "ball"][0][:, live])
        assert (outs[split, "ball"][0][:, ~live] == 0).all()


@pytest.mark.parametrize("mode", FORMATS)
def test_gqi_nonfinite_samples_propagate_like_the_reference(fj, orc, mode, monkeypatch):
    """gqi.jl:139-144 on samples that are not finite: `s[s .< 0] .= 0` turns -Inf into 0 and keeps NaN; `maximum(s) == 0`
    does not skip a voxel with a NaN; mul!(o, A, s) makes a NaN sample a NaN column and a +Inf sample +-Inf rows (NaN where
    the coefficient is 0 or where both signs meet).  Nothing is trapped."""
    _use_format(monkeypatch, mode)
    from fibers_jl_amd import phantom
    shape = (8, 6, 5)
    bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed=23, noise_frac=0.02, crossing=True)
    mask = np.ones(shape, np.uint8)
    mask[0, 0, 1] = 0
    dwi[1, 0, 0, 7] = np.inf                                   # +Inf: +-Inf rows
    dwi[2, 0, 0, 9] = -np.inf                                  # -Inf: clamped to 0, finite column
    dwi[3, 0, 0, 11] = np.nan                                  # NaN: NaN column
    dwi[4, 0, 0, 5] = np.inf; dwi[4, 0, 0, 40] = np.inf        # two +Inf samples: NaN where their coefficients disagree in sign
    dwi[5, 0, 0, :] = -1.0; dwi[5, 0, 0, 3] = np.nan           # all samples <= 0 and one NaN: not skipped, NaN column
    dwi[6, 0, 0, :] = -np.inf                                  # everything clamps to 0: skipped, zeros
    dwi[0, 0, 1, 2] = np.inf                                   # outside the mask: zeros
    sph = fj.sphere_642
    with np.errstate(all="ignore"):
        ref = orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=2)
        got = fj.gqi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph)
    ro, go = ref["odf"], got.odf.vol
    special = np.zeros(shape, bool)
    special[1:7, 0, 0] = True
    assert np.array_equal(np.isnan(go), np.isnan(ro))
    assert np.array_equal(np.isposinf(go), np.isposinf(ro)) and np.array_equal(np.isneginf(go), np.isneginf(ro))
    assert np.isinf(ro[1, 0, 0]).sum() > 300 and np.isfinite(ro[2, 0, 0]).all() and np.isnan(ro[3, 0, 0]).all() and np.isnan(ro[5, 0, 0]).all()
    assert (go[6, 0, 0] == 0).all() and (go[0, 0, 1] == 0).all()
    fin = np.isfinite(ro)
    scale = np.abs(np.where(fin, ro, 0.0)).max(axis=3, keepdims=True) + 1e-30
    with np.errstate(all="ignore"):
        assert (np.abs(np.where(fin, go - ro, 0.0)) / scale).max() < 2e-5
    # the global odfmax is NaN (maximum of means, gqi.jl:164) -> every qa is NaN after the normalisation, peaks as the oracle's
    for k in range(3):
        assert np.array_equal(np.isnan(got.qa[k].vol[..., 0]), np.isnan(ref["qa"][k]))
        same = np.all(ref["peak"][k] == got.peak[k].vol, axis=3)
        assert same[special].all(), (mode, k)
        assert (~same).sum() <= 2


@pytest.mark.parametrize("mode", FORMATS)
def test_dsi_nonfinite_samples_propagate_like_the_reference(fj, orc, mode, monkeypatch):
    """dsi.jl:205-225 on samples that are not finite: max.(X, 0) turns -Inf into 0 and keeps NaN; the FFT smears a NaN or
    +Inf sample over the whole grid and p ./ sum(p) leaves NaN everywhere in that voxel's pdf and odf."""
    _use_format(monkeypatch, mode)
    dwi, mask, bval, bvec = _dsi_case((5, 4, 3), seed=6)
    mask[:, 0, 0] = 1
    dwi[0, 0, 0, 100] = np.inf
    dwi[1, 0, 0, 200] = -np.inf
    dwi[2, 0, 0, 300] = np.nan
    dwi[3, 0, 0, :] = -2.0; dwi[3, 0, 0, 17] = np.nan            # maximum(X) is NaN, not 0: the voxel is not skipped
    sph = fj.sphere_642
    with np.errstate(all="ignore"):
        ref = orc.dsi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, 32, nthreads=4)
        got = fj.dsi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph, 32)
    for name, g, r in (("pdf", got.pdf.vol, ref["pdf"]), ("odf", got.odf.vol, ref["odf"])):
        assert np.isnan(r[0, 0, 0]).all() and np.isnan(r[2, 0, 0]).all() and np.isnan(r[3, 0, 0]).all() and np.isfinite(r[1, 0, 0]).all(), name
        assert np.array_equal(np.isnan(g), np.isnan(r)), name
        fin = np.isfinite(r)
        assert np.isfinite(g[fin]).all(), name
        scale = np.abs(np.where(fin, r, 0.0)).max(axis=3, keepdims=True) + 1e-30
        with np.errstate(all="ignore"):
            assert (np.abs(np.where(fin, g - r, 0.0)) / scale).max() < 1e-4, name


@pytest.mark.parametrize("kind", ["gqi", "dsi"])
@pytest.mark.parametrize("shape,live", [((11, 9, 7), 0.9), ((11, 9, 7), 0.3), ((40, 40, 26), 0.5), ((40, 40, 26), 0.97), ((12, 12, 12), 0.0)])
def test_outputs_outside_the_mask_are_cleared_whatever_they_held(fj, kind, shape, live):
    """The reference's output volumes start zero-filled (mri.jl:251-255); the library writes into caller-owned memory, so every
    voxel outside the mask must be cleared whatever the buffers held: both clearing strategies (selective below a quarter
    of the volume outside the mask, everything above), voxel counts that are not multiples of 4 / 1024."""
    import torch
    from fibers_jl_amd import phantom
    dev = torch.device("cuda", 0)
    nvox = int(np.prod(shape))
    bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3) if kind == "gqi" else phantom.scheme_dsi()
    dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=8, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(5)
    mask = (torch.rand(nvox, device=dev, generator=g) < live).to(torch.uint8)
    plan = fj.OdfPlan(kind, bval, bvec, fj.sphere_642)

    def run(fill):
        out = dict(odf=torch.full((plan.nvert, nvox), fill, dtype=torch.float32, device=dev),
                   peak=[torch.full((3, nvox), fill, dtype=torch.float32, device=dev) for _ in range(3)],
                   qa=[torch.full((nvox,), fill, dtype=torch.float32, device=dev) for _ in range(3)],
                   odfmax=torch.empty(2, dtype=torch.float32, device=dev))
        if kind == "dsi":
            out["pdf"] = torch.full((plan.nvol, nvox), fill, dtype=torch.float32, device=dev)
        fj.odf_rec_device(plan, dwi, mask, out=out)
        torch.cuda.synchronize()
        return out

    a, b = run(0.0), run(float("nan"))
    dead = mask == 0
    for key in ("odf", "pdf") if kind == "dsi" else ("odf",):
        assert torch.equal(a[key], b[key]), key
        assert (b[key][:, dead] == 0).all(), key
    for k in range(3):
        assert torch.equal(a["peak"][k], b["peak"][k]) and torch.equal(a["qa"][k].nan_to_num(nan=-7.0), b["qa"][k].nan_to_num(nan=-7.0))
        assert (b["peak"][k][:, dead] == 0).all()
        if live > 0:
            assert (b["qa"][k][dead] == 0).all()
        else:                                                # odfmax = 0: qa ./ odfmax is 0/0 everywhere (gqi.jl:166-168), not trapped
            assert torch.isnan(b["qa"][k]).all()
    plan.close()


def test_dsi_fold_inside_the_kernel_and_fold_prepass_agree(fj):
    """The split kernels fold the antipodal pairs themselves (default format); a plan in the f32 format takes the separate fold pre-pass
    (dsi_fold_kernel) and the f32 MFMA chain.  Same folded numbers into two contractions: pdf / odf agree to the formats' rounding,
    non-finite samples and voxels outside the mask behave identically -- with a mask, a ragged voxel count and non-finite samples.
    (The pre-pass in front of the SPLIT kernels is what volumes past the fold-span limit run: tests/test_gpu_fullsize.py.)"""
    import torch
    from fibers_jl_amd import phantom
    dev = torch.device("cuda", 0)
    shape = (23, 21, 10)
    nvox = int(np.prod(shape))
    bval, bvec = phantom.scheme_dsi()
    dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=12, device=dev)
    dwi[100, 7] = float("nan"); dwi[400, 8] = float("inf"); dwi[33, 9] = -float("inf"); dwi[:, 10] = -1.0
    g = torch.Generator(device=dev); g.manual_seed(2)
    L = fj.lib()
    for mname, mask in (("ones", torch.ones(nvox, dtype=torch.uint8, device=dev)),
                        ("sparse", (torch.rand(nvox, device=dev, generator=g) < 0.6).to(torch.uint8))):
        res = {}
        for mode, fmt in (("fused", "fp16x2"), ("prepass", "f32")):
            plan = fj.OdfPlan("dsi", bval, bvec, fj.sphere_642, hann_width=32, format=fmt)
            L.fib_profile_enable(1); L.fib_profile_reset()
            o = fj.odf_rec_device(plan, dwi, mask)
            torch.cuda.synchronize()
            import ctypes as C
            ms, cnt = C.c_double(0), C.c_int64(0)
            L.fib_profile_get(b"dsi_fold", C.byref(ms), C.byref(cnt))
            L.fib_profile_enable(0)
            assert (cnt.value > 0) == (mode == "prepass"), (mode, cnt.value)
            res[mode] = {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in o.items()}
            plan.close()
        nn = lambda t: t.nan_to_num(nan=-7.0, posinf=-8.0, neginf=-9.0)
        for k in ("pdf", "odf"):
            a_, b_ = nn(res["fused"][k]), nn(res["prepass"][k])
            scale = b_.abs().amax(dim=0).clamp_min(1e-30)
            assert float(((a_ - b_).abs() / scale).max()) <= 4e-6, (mname, k)
            assert torch.equal(a_ == -7.0, b_ == -7.0) and torch.equal(a_ == -8.0, b_ == -8.0), (mname, k)     # NaN / Inf in the same places
        assert bool(torch.isnan(res["fused"]["odf"][:, 7]).all()) and bool(torch.isnan(res["fused"]["odf"][:, 8]).all()) or mname == "sparse"
        assert bool((res["fused"]["odf"][:, 10] == 0).all())


@pytest.mark.parametrize("sphere", ["sphere_362", "sphere_642", "sphere_724"])
def test_find_peaks_randomised_against_the_oracle(fj, orc, sphere):
    """find_peaks! (gqi.jl:180-201) on adversarial ODFs: heavy ties (values on a coarse grid), plateaus, negative values,
    all-equal columns, +-0, NaN and +-Inf entries.  sortperm(rev=true) order (ties: lower index first, NaN first) and the
    count of positive survivors must match the oracle exactly for every column."""
    sph = getattr(fj, sphere)
    rng = np.random.default_rng(31)
    n = 400
    odf = rng.random((n, sph.nvert)).astype(np.float32)
    odf[0:60] = np.round(odf[0:60] * 3) / 3                       # three levels: ties everywhere
    odf[60:90] = np.round(odf[60:90] * 16) / 16 - 0.5             # ties and negatives
    odf[90] = 0.25                                                # all equal: nobody is strictly greater
    odf[91] = 0.0; odf[92] = -0.0
    for i in range(100, 140):
        odf[i, rng.integers(0, sph.nvert, 3)] = np.nan
    for i in range(140, 160):
        odf[i, rng.integers(0, sph.nvert, 2)] = np.inf
        odf[i, rng.integers(0, sph.nvert, 2)] = -np.inf
    odf[160:170] = np.nan
    odf[170:200] = (odf[170:200] > 0.97).astype(np.float32)       # isolated spikes on a plateau of zeros
    top, nvalid = fj.find_peaks(odf.reshape(n, 1, sph.nvert), sph)
    faces0 = orc.fold_faces(sph.faces, sph.nvert)
    for i in range(n):
        with np.errstate(all="ignore"):
            isort, nv, _ = orc.find_peaks(odf[i], faces0)
        assert nv == nvalid[i, 0], (i, nv, nvalid[i, 0])
        assert list(isort[:3]) == list(top[i, 0]), (i, isort[:3], top[i, 0])


@pytest.mark.parametrize("case", range(8))
def test_gqi_and_dsi_randomised_configurations(fj, orc, case, monkeypatch):
    """Seeded random draws over what gqi_rec / dsi_rec take: tessellation, sigma, shell layout and frame order, Hanning
    width, lattice radius (257 / 389 / 515 frames, shuffled so that antipodal partners sit anywhere), mask density, fraction of
    non-positive samples, and which contraction kernel runs."""
    from fibers_jl_amd import phantom
    rng = np.random.default_rng(900 + case)
    _use_format(monkeypatch, FORMATS[[0, 1, 0, 2, 1, 0, 1, 2][case]])
    sph = getattr(fj, ["sphere_642", "sphere_362", "sphere_724"][case % 3])
    shape = tuple(int(x) for x in rng.integers(3, 9, 3))
    mask = (rng.random(shape) < rng.uniform(0.5, 1.0)).astype(np.uint8)
    # ---- GQI
    nb0, ndir = int(rng.integers(1, 5)), int(rng.integers(12, 40))
    shells = tuple(sorted(rng.choice([700.0, 1000.0, 1500.0, 2000.0, 3000.0, 5000.0], size=int(rng.integers(1, 4)), replace=False)))
    bval, bvec = phantom.scheme_gqi(nb0, ndir, shells, 900 + case)
    perm = rng.permutation(len(bval))
    bval, bvec = np.ascontiguousarray(bval[perm]), np.ascontiguousarray(bvec[perm])
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, 900 + case, noise_frac=0.03, nonpositive_frac=float(rng.choice([0.0, 0.03])), crossing=True)
    sigma = float(rng.choice([1.0, 1.25, 1.6]))
    ref = orc.gqi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, sigma, nthreads=3)
    got = fj.gqi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph, sigma)
    _check_odf_rec(got.odf.vol, [p.vol for p in got.peak], [q.vol[..., 0] for q in got.qa], ref, mask, label="gqi case %d" % case)
    # ---- DSI
    bval, bvec = phantom.scheme_dsi(bmax=float(rng.choice([5000.0, 7000.0])), r2max=int(rng.choice([16, 20, 25])))
    perm = rng.permutation(len(bval))
    bval, bvec = np.ascontiguousarray(bval[perm]), np.ascontiguousarray(bvec[perm])
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, 950 + case, noise_frac=0.01, crossing=True)
    hw = int(rng.choice([16, 32, 48]))
    ref = orc.dsi_rec(dwi, mask, bval, bvec, sph.vertices, sph.faces, hw, nthreads=3)
    got = fj.dsi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph, hw)
    scale = np.abs(ref["pdf"]).max(axis=3, keepdims=True) + 1e-30
    assert (np.abs(got.pdf.vol - ref["pdf"]) / scale).max() < 5e-5, "dsi pdf case %d" % case
    _check_odf_rec(got.odf.vol, [p.vol for p in got.peak], [q.vol[..., 0] for q in got.qa], ref, mask, odf_rtol=1e-4, qa_atol=1e-4,
                   label="dsi case %d" % case)


@pytest.mark.parametrize("kind", ["gqi", "dsi"])
def test_normalisation_inside_the_post_kernel_matches_the_separate_launch(fj, kind):
    """[r6] FIB_ODF_NORMALIZE runs inside odf_post_kernel (the last workgroup to arrive publishes the divisor, every workgroup divides
    its share of qa).  Voxels on the redo list (NaN / Inf samples, candidate-list overflow) get their qa from one workgroup and have it
    divided by another -- possibly on another XCD: bit-identical to `normalize=False` + fibd_qa_normalize_dev, over many volumes whose
    poisoned voxels sit in different places; and the drain list of the exact-mean refinement at and beyond its capacity (a volume of
    identical voxels: every voxel is a candidate for the maximum).  gqi.jl:164-168 / dsi.jl:263-267."""
    import torch
    from fibers_jl_amd import phantom
    dev = torch.device("cuda", 0)
    shape = (24, 20, 14)
    nvox = shape[0] * shape[1] * shape[2]
    bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3) if kind == "gqi" else phantom.scheme_dsi()
    plan = fj.OdfPlan(kind, bval, bvec, fj.sphere_642)
    base, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=6, device=dev, noise_frac=0.05)
    mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(11)
    for trial in range(12):
        d = base.clone()
        npo = [0, 1, 7, 40][trial % 4]
        if npo:
            vox = torch.randint(0, nvox, (npo,), device=dev, generator=g)
            frm = torch.randint(0, len(bval), (npo,), device=dev, generator=g)
            d[frm, vox] = float("inf") if trial % 3 else float("nan")
        if trial == 11:                                         # identical voxels: every mean is the maximum (list capacity 2048 < 6720 voxels)
            d = base[:, :1].repeat(1, nvox).contiguous()
        a = fj.odf_rec_device(plan, d, mask, normalize=True)
        torch.cuda.synchronize()
        a = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in a.items()}
        b = fj.odf_rec_device(plan, d, mask, normalize=False)
        fj.qa_normalize_device(b["qa"], b["odfmax"])
        torch.cuda.synchronize()
        nn = lambda t: t.nan_to_num(nan=-7.0, posinf=-8.0, neginf=-9.0)      # noqa: E731
        assert torch.equal(nn(a["odfmax"]), nn(b["odfmax"])), (trial, a["odfmax"], b["odfmax"])
        for k in range(3):
            assert torch.equal(nn(a["qa"][k]), nn(b["qa"][k])), (trial, k)
            assert torch.equal(nn(a["peak"][k]), nn(b["peak"][k])), (trial, k)
        assert torch.equal(nn(a["odf"]), nn(b["odf"]))
        if npo == 0 and trial != 11:
            assert float(a["odfmax"][1]) == 0.0 and float(a["qa"][0].max()) > 0
        if trial == 11:
            ref = b["odf"][:, 0].cpu().numpy()
            s = np.float32(0)
            for x in ref:
                s = np.float32(s + x)                          # mean(odf, dims=4): sequential float32 sum over the vertices, ./ n (gqi.jl:164)
            assert float(a["odfmax"][0]) == float(np.float32(s / np.float32(len(ref))))


def test_profile_filter_brackets_only_the_named_kernels(fj):
    """fib_profile_filter("odf_gemm"): only that kernel's launches are bracketed by events (bench.py's timed steps); NULL: all again"""
    import ctypes as C
    import torch
    from fibers_jl_amd import phantom
    L = fj.lib()
    dev = torch.device("cuda", 0)
    bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
    plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642)
    d, _ = phantom.make_dwi_torch((16, 16, 8), bval, bvec, seed=6, device=dev)
    mask = torch.ones(16 * 16 * 8, dtype=torch.uint8, device=dev)

    def counts():
        out = {}
        for k in ("odf_gemm", "mask_compact", "odf_post"):
            ms, n = C.c_double(0), C.c_int64(0)
            L.fib_profile_get(k.encode(), C.byref(ms), C.byref(n))
            out[k] = (n.value, ms.value)
        return out
    try:
        L.fib_profile_filter(b"odf_gemm")
        L.fib_profile_enable(1); L.fib_profile_reset()
        for _ in range(3):
            fj.odf_rec_device(plan, d, mask)
        torch.cuda.synchronize()
        c = counts()
        assert c["odf_gemm"][0] == 3 and c["odf_gemm"][1] > 0 and c["mask_compact"][0] == 0 and c["odf_post"][0] == 0
        L.fib_profile_filter(None)
        L.fib_profile_reset()
        for _ in range(2):
            fj.odf_rec_device(plan, d, mask)
        torch.cuda.synchronize()
        c = counts()
        assert c["odf_gemm"][0] == 2 and c["mask_compact"][0] == 2 and c["odf_post"][0] == 2
    finally:
        L.fib_profile_enable(0)
        L.fib_profile_filter(None)
        L.fib_profile_reset()


def test_qa_normalize_with_device_scalar(fj):
    """fibd_qa_normalize_dev: the divisor comes from device memory (the all-reduced odfmax of the multi-GPU flow)"""
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    qa = [torch.rand(1000, device="cuda", generator=g) for _ in range(3)]
    om = torch.tensor([3.5, 0.0], dtype=torch.float32, device="cuda")
    ref = [torch.div(q, om[0]) for q in qa]                       # (tensor divisor: a Python scalar makes torch multiply by 1/x)
    fj.qa_normalize_device(qa, om)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(qa, ref))
    fj.qa_normalize_device(qa, 2.0)
    torch.cuda.synchronize()
    two = torch.tensor(2.0, device="cuda")
    assert all(torch.equal(a, torch.div(b, two)) for a, b in zip(qa, ref))


@pytest.mark.parametrize("sphere", ["sphere_642", "sphere_362", "sphere_724"])
def test_find_peaks_work_fills_the_whole_work_struct(fj, sphere):
    """fib_find_peaks_work = find_peaks!(W) with all its outputs (gqi.jl:180-201): odf_peak, the complete isort and the count,
    against the independent NumPy restatement (the reference's face-mask formulation + sortperm by isless): exactly, on smooth
    ODFs, exact ties, zeros, negative lobes, -0.0, NaN and +-Inf amplitudes"""
    from oracle import oracle_np as onp
    sph = getattr(fj, sphere)
    nv = sph.nvert
    rng = np.random.default_rng(17)
    odf = rng.random((40, nv)).astype(np.float32)
    odf[1] = np.round(odf[1] * 4) / 4                        # many exact ties
    odf[2] = 0.0
    odf[3] -= 0.5                                            # negative lobes
    odf[4, ::7] = -0.0
    odf[5, 10] = np.nan
    odf[6, 3] = np.inf; odf[6, 40] = -np.inf
    odf[7] = 1.0                                             # all equal: no peak
    faces0 = onp.fold_faces(np.asarray(sph.faces), nv)
    pk, isort, nvalid = fj.find_peaks_work(odf, sph)
    for i in range(odf.shape[0]):
        want_isort, want_n, want_pk = onp.find_peaks(odf[i], faces0)
        assert np.array_equal(pk[i], want_pk, equal_nan=True), i
        assert int(nvalid[i]) == want_n, i
        assert np.array_equal(isort[i], want_isort), i
    top, nv3 = fj.find_peaks(odf, sph)                       # the three-entry form agrees with the complete one
    assert np.array_equal(top, isort[:, :3]) and np.array_equal(nv3, nvalid)


@pytest.mark.parametrize("case", ["full", "sparse", "poison"])
def test_dsi_two_tile_kernel_against_the_three_tile_path(fj, case):
    """odf_dsi2_kernel (default for folded lattices on sphere_642: fused ODF tile + pdf tile in one launch, find_peaks! on the
    accumulators) against the path other tessellations and FIB_ODF_SEPARATE_PEAKS take (three M tiles, then odf_peaks642_kernel on the stored
    ODF), same device buffers: peaks identical, qa / odfmax to rounding, ODF rows bit-identical except the three rows whose
    role differs (the fused layout's pole row and the three-tile layout's two extra rows are f32 VALU rows), pdf to rounding."""
    import torch
    from fibers_jl_amd import phantom
    dev = torch.device("cuda", 0)
    shape = (21, 17, 11)
    nvox = shape[0] * shape[1] * shape[2]
    bval, bvec = phantom.scheme_dsi()
    dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=9, device=dev, noise_frac=0.05)
    mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
    if case == "sparse":
        mask = (torch.rand(nvox, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) < 0.4).to(torch.uint8)
    if case == "poison":
        dwi[5, 100] = float("nan"); dwi[7, 2000] = float("inf"); dwi[:, 3000] = 0.0; dwi[:, 3001] = -1.0; dwi[0, 3002] = 0.0
    res = {}
    for name, sep in (("two", False), ("three", True)):
        plan = fj.OdfPlan("dsi", bval, bvec, fj.sphere_642, hann_width=32)
        o = fj.odf_rec_device(plan, dwi, mask, normalize=True, separate_peaks=sep)
        torch.cuda.synchronize()
        res[name] = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in o.items()}
    a, b = res["two"], res["three"]
    nn = lambda t: torch.nan_to_num(t, nan=-7.0, posinf=-8.0, neginf=-9.0)
    differing = (nn(a["odf"]) != nn(b["odf"])).any(1)
    assert int(differing.sum()) <= 3
    torch.testing.assert_close(nn(a["odf"]), nn(b["odf"]), rtol=2e-5, atol=1e-9)
    torch.testing.assert_close(nn(a["pdf"]), nn(b["pdf"]), rtol=2e-5, atol=1e-9)
    same = torch.ones(nvox, dtype=torch.bool, device=dev)
    for k in range(3):
        same &= (a["peak"][k] == b["peak"][k]).all(0)
    assert float(same.float().mean()) > 0.999                 # (a tie decided by one of the three rows may fall the other way)
    for k in range(3):
        torch.testing.assert_close(nn(a["qa"][k])[same], nn(b["qa"][k])[same], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(nn(a["odfmax"]), nn(b["odfmax"]), rtol=1e-5, atol=0)
    assert int((a["peak"][0] != 0).any(0).sum()) > int(mask.sum()) // 2


# ---- the two operand formats of the contraction: two fp16 pieces (default) / three exact bf16 pieces (format="bf16x3") ----------
def _rec_both_formats(fj, monkeypatch, kind, dwi, mask, bval, bvec, sph):
    """device tier with explicitly built plans (the host tier caches its plans; the format is chosen when a plan builds its matrix
    image): dict(odf [nvox, nvert], peak 3 x [nvox, 3], pdf) per format"""
    import torch
    nvol = dwi.shape[3]
    d = torch.from_numpy(np.ascontiguousarray(dwi.reshape(-1, nvol, order="F").T)).cuda()     # planar [nvol, nvox], x fastest
    m = torch.from_numpy(np.ascontiguousarray(mask.reshape(-1, order="F"))).cuda()
    res = {}
    for exact in (False, True):
        plan = fj.OdfPlan(kind, bval, bvec, sph, sigma=1.25, hann_width=32, format="bf16x3" if exact else "fp16x2")
        assert plan.format == ("bf16x3" if exact else "fp16x2")      # what the plan's kernels run, read back from the library
        o = fj.odf_rec_device(plan, d, m)
        torch.cuda.synchronize()
        res[exact] = dict(odf=o["odf"].cpu().numpy().T.copy(), peak=[p.cpu().numpy().T.copy() for p in o["peak"]],
                          pdf=o["pdf"].cpu().numpy().T.copy() if kind == "dsi" else None)
        plan.close()
    return res[False], res[True]


def _formats_agree(a, b, tol, nvert):
    """a: default (fp16 x 2), b: exact split (bf16 x 3).  ODF within tol of the voxel maximum (the exact split's own distance from a
    float64 contraction is ~1e-6: f32 accumulation), peaks identical except rounding-level ties"""
    from util import peak_mismatches_are_ties
    oa, ob = a["odf"], b["odf"]
    assert not np.array_equal(oa, ob), "the two operand formats gave identical bits: the switch did not reach the plan"
    scale = np.abs(ob).max(axis=1, keepdims=True) + 1e-30
    assert np.array_equal(np.isnan(oa), np.isnan(ob))
    err = np.nanmax(np.abs(oa - ob) / scale)
    assert err <= tol, "odf differs by %g of the voxel maximum between the two operand formats" % err
    return peak_mismatches_are_ties(ob, b["peak"], a["peak"], _VERTS[nvert][:nvert], faces=_FACES[nvert])


def test_fp16_pieces_agree_with_the_exact_split_gqi(fj, monkeypatch):
    dwi, mask, bval, bvec = _gqi_case((16, 16, 12), seed=21)
    a, b = _rec_both_formats(fj, monkeypatch, "gqi", dwi, mask, bval, bvec, fj.sphere_642)
    _formats_agree(a, b, 3e-6, fj.sphere_642.nvert)


def test_fp16_pieces_agree_with_the_exact_split_dsi(fj, monkeypatch):
    dwi, mask, bval, bvec = _dsi_case((8, 8, 8), seed=22)
    a, b = _rec_both_formats(fj, monkeypatch, "dsi", dwi, mask, bval, bvec, fj.sphere_642)
    _formats_agree(a, b, 3e-6, fj.sphere_642.nvert)
    pa, pb = a["pdf"], b["pdf"]
    assert np.nanmax(np.abs(pa - pb) / (np.abs(pb).max(axis=1, keepdims=True) + 1e-30)) <= 3e-6


@pytest.mark.parametrize("k", [-60, -17, 23, 60])
def test_scaling_the_samples_by_a_power_of_two_scales_the_odf_exactly(fj, k):
    """every voxel carries its own power-of-two sample scale (fp16 pieces have 5 exponent bits): the result must not depend on the
    magnitude of the data -- ODF x 2^k bit for bit, same peaks, same qa (which is a ratio)"""
    dwi, mask, bval, bvec = _gqi_case((8, 8, 8), seed=31)
    sph = fj.sphere_642
    a = fj.gqi_rec(fj.MRI(dwi, bval, bvec), fj.MRI(mask), sph)
    b = fj.gqi_rec(fj.MRI(np.ldexp(dwi, k).astype(np.float32), bval, bvec), fj.MRI(mask), sph)
    assert np.array_equal(np.ldexp(a.odf.vol.astype(np.float64), k), b.odf.vol.astype(np.float64))
    for p, q in zip(a.peak, b.peak):
        assert np.array_equal(p.vol, q.vol)
    for p, q in zip(a.qa, b.qa):
        assert np.array_equal(p.vol, q.vol)


def test_late_large_samples_lower_the_sample_scale(fj, orc):
    """the scale is chosen from the first 16 frames; frames that come later and are > 256 x larger make the kernel lower it and
    rescale the accumulators on the way: first stage tiny, first stage all zero, one huge frame at the very end, denormal samples"""
    dwi, mask, bval, bvec = _gqi_case((8, 8, 8), seed=41)
    sph = fj.sphere_642
    nf = dwi.shape[3]
    cases = {}
    d = dwi.copy(); d[..., :16] = np.ldexp(d[..., :16], -14); cases["tiny first stage"] = d
    d = dwi.copy(); d[..., :16] = 0.0; cases["zero first stage"] = d
    d = dwi.copy(); d[..., nf - 1] = np.ldexp(d[..., nf - 1], 20); cases["huge last frame"] = d
    d = dwi.copy(); d[..., :32] = 1e-41; d[..., 40] = np.ldexp(d[..., 40], 12); cases["denormal samples, then two steps up"] = d
    d = np.ldexp(dwi, -140).astype(np.float32); cases["all samples denormal"] = d
    for label, d in cases.items():
        ref = orc.gqi_rec(d, mask, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=2)
        got = fj.gqi_rec(fj.MRI(d, bval, bvec), fj.MRI(mask), sph)
        # (relative to the voxel maximum itself -- _check_odf_rec's 1e-30 floor would hide the denormal case)
        m = mask.astype(bool)
        top = np.abs(ref["odf"][m]).max(axis=1, keepdims=True)
        err = np.abs(got.odf.vol[m] - ref["odf"][m]) / top
        assert err.max() <= (2e-5 if label != "all samples denormal" else 2e-2), "%s: odf rel err %g" % (label, err.max())   # (a denormal sum carries a few bits)
        _check_odf_rec(got.odf.vol, [p.vol for p in got.peak], [q.vol[..., 0] for q in got.qa], ref, mask, label=label)


def _rec_dict(out):
    import torch
    torch.cuda.synchronize()
    return {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in out.items()}


@pytest.mark.parametrize("kind", ["gqi", "dsi"])
def test_list_unit_does_not_change_results_and_follows_the_mask(fj, kind):
    """The voxel list of the contraction kernels is made of aligned groups of 32 voxels by default (a wave's 128-byte row segments
    are whole cache lines whatever the mask's runs look like) and of aligned groups of 4 when the previous call's mask was sparse.
    Outputs are bit-identical either way (ragged ball mask, runs that start anywhere, isolated voxels, a volume that does not end on
    a group of 32); after a call with a sparse mask the plan switches to groups of 4, after a blob it switches back."""
    import torch
    from fibers_jl_amd import phantom
    dev = torch.device("cuda", 0)
    shape = (27, 22, 14)                                   # 8 316 voxels = 259 groups of 32 + 28
    bval, bvec = (phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3) if kind == "gqi" else phantom.scheme_dsi())
    dwi, _ = phantom.make_dwi_torch(shape, bval, bvec, seed=31, device=dev)
    nvox = dwi.shape[1]
    g = torch.Generator(device=dev); g.manual_seed(5)
    ball = phantom.ball_mask_torch(shape, dev).reshape(-1)
    sparse = (torch.rand(nvox, device=dev, generator=g) < 0.02).to(torch.uint8)
    runs = torch.zeros(nvox, dtype=torch.uint8, device=dev)
    for a, b in ((3, 41), (77, 78), (130, 389), (1001, 1033), (nvox - 9, nvox)):
        runs[a:b] = 1
    plan = fj.OdfPlan(kind, bval, bvec, fj.sphere_642, device=0)
    # the unit of a call is chosen by the mask of the call BEFORE it (no switch: the product library has none): a call on isolated voxels
    # puts the next one on groups of 4, a call on one long run puts it on groups of 32
    slab = torch.zeros(nvox, dtype=torch.uint8, device=dev)
    slab[37:5000] = 1
    prime = {"quads": sparse, "octets": slab}
    for name, mask in (("ball", ball), ("sparse", sparse), ("runs", runs)):
        res = {}
        for unit in ("quads", "octets"):
            fj.odf_rec_device(plan, dwi, prime[unit])
            assert plan.list_unit() == unit, (name, unit)
            res[unit] = _rec_dict(fj.odf_rec_device(plan, dwi, mask))
        a, b = res["quads"], res["octets"]
        for k in a:
            if isinstance(a[k], list):
                assert all(torch.equal(x, y) or torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)) for x, y in zip(a[k], b[k])), (name, k)
            else:
                assert torch.equal(torch.nan_to_num(a[k], nan=-7.0), torch.nan_to_num(b[k], nan=-7.0)), (name, k)
        dead = mask == 0
        assert float(a["odf"][:, dead].abs().max()) == 0.0, name
    # the plan follows the mask (one call behind): long runs -> groups of 32, isolated voxels -> groups of 4
    fj.odf_rec_device(plan, dwi, slab)
    assert plan.list_unit() == "octets"
    fj.odf_rec_device(plan, dwi, sparse)
    assert plan.list_unit() == "quads"
    got = _rec_dict(fj.odf_rec_device(plan, dwi, slab))     # this call runs on groups of 4 ..
    assert plan.list_unit() == "octets"                     # .. and puts the next one back on groups of 32
    want = _rec_dict(fj.odf_rec_device(plan, dwi, slab))    # (this one runs on groups of 32)
    assert torch.equal(got["odf"], want["odf"]) and all(torch.equal(x, y) for x, y in zip(got["peak"], want["peak"]))


def test_clearing_workgroups_whose_poll_times_out_cover_both_partitions():
    """mask_compact_kernel: a mixture of per-workgroup decisions (whole arrays | span by span) must not leave values
    outside the mask uncleared.  The forced time-out exists in the DIAGNOSTIC build only, so the check runs as a child process that
    loads libfibers_hip_stamp.so (tools/compact_mixture_check.py: NaN-filled outputs, three masks, every 2nd / 3rd / 7th workgroup
    forced 'unknown', GQI and DSI)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "fibers.jl_amd", "libfibers_hip_stamp.so")):
        pytest.skip("the diagnostic build is absent (make -C fibers.jl_amd/csrc stamp)")
    env = {k: v for k, v in os.environ.items() if k != "FIBERS_HIP_LIB"}
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "compact_mixture_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "compact mixture check: ok" in out.stdout
