"""BASELINE.json configurations C2..C5 at their full size (140^3) on the GPU, checked two ways:
  * parity with the CPU oracle on a random SAMPLE of voxels / seeds (the sampled voxel columns are gathered into a tiny
    volume the oracle finishes in seconds; tracking runs the oracle on the full field for a few hundred seeds);
  * size-independent properties over the WHOLE result: outputs outside the mask are exactly zero, the contraction is
    exactly homogeneous under power-of-two scaling (every bf16 piece, product and partial sum scales exactly), peak
    directions are sphere vertices, QA is scale invariant, streamline steps have the prescribed length, lines stay
    inside the mask, counts and offsets are consistent.
Inputs are the synthetic phantoms bench.py uses."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPE = (140, 140, 140)
NVOX = 140 ** 3


def _sample_volume(planar, idx):
    """gather voxel columns idx of a planar CUDA tensor [nframes, nvox] into an oracle volume [n,1,1,nframes]"""
    cols = planar[:, idx].T.contiguous().cpu().numpy()
    return np.asfortranarray(cols.reshape(len(idx), 1, 1, -1))


@pytest.fixture(scope="module")
def dev():
    import torch
    return torch.device("cuda", 0)


def test_c2_dti_140cubed(fj, orc, dev):
    import torch
    from fibers_jl_amd import phantom
    from util import assert_dti_close
    bval, bvec = phantom.scheme_dti(60, 4, 1000.0, seed=2)
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=2, device=dev, nfib=1)
    mask = phantom.ball_mask_torch(SHAPE, dev)
    plan = fj.DtiPlan(bval, bvec, device=0)
    out = fj.dti_fit_device(plan, dwi, mask)
    torch.cuda.synchronize()
    live = mask.bool()
    for k in fj.dti.DTI_FIELDS:                               # zero outside the mask (dti.jl:247-261)
        assert float(out[k].reshape(-1, NVOX)[:, ~live].abs().max()) == 0.0, k
    fa = out["fa"][live]
    assert bool(torch.isfinite(fa).all()) and float(fa.min()) >= 0.0 and float(fa.max()) <= 1.0 + 1e-6
    md = (out["eigval1"] + out["eigval2"] + out["eigval3"]) / 3.0          # dti_maps identities (dti.jl:325-335)
    torch.testing.assert_close(out["md"][live], md[live], rtol=2e-6, atol=1e-12)
    torch.testing.assert_close(out["rd"][live], ((out["eigval2"] + out["eigval3"]) / 2.0)[live], rtol=2e-6, atol=1e-12)
    nrm = (out["eigvec1"].reshape(3, NVOX) ** 2).sum(0)[live]
    assert float((nrm - 1).abs().max()) < 1e-5                # unit eigenvectors
    # oracle parity on 3000 sampled voxels (inside and outside the mask)
    rng = np.random.default_rng(7)
    idx = np.sort(rng.choice(NVOX, 3000, replace=False))
    tidx = torch.from_numpy(idx).to(dev)
    sub = _sample_volume(dwi, tidx)
    msub = mask[tidx].cpu().numpy().reshape(-1, 1, 1)
    ref = orc.dti_fit(sub, msub, bval, bvec, nthreads=4)
    got = {}
    for k in fj.dti.DTI_FIELDS:
        t = out[k].reshape(-1, NVOX)[:, tidx].T.contiguous().cpu().numpy()
        got[k] = t.reshape(len(idx), 1, 1, -1) if t.shape[1] == 3 else t.reshape(len(idx), 1, 1)
    assert_dti_close(got, ref, msub, label="C2 sample")


def _check_odf_sample(fj, orc, kind, out, dwi, mask, bval, bvec, dev, nsamp, odf_rtol):
    import torch
    rng = np.random.default_rng(11)
    idx = np.sort(rng.choice(int(mask.numel()), nsamp, replace=False))
    tidx = torch.from_numpy(idx).to(dev)
    sub = _sample_volume(dwi, tidx)
    msub = mask[tidx].cpu().numpy().reshape(-1, 1, 1)
    sph = fj.sphere_642
    if kind == "gqi":
        ref = orc.gqi_rec(sub, msub, bval, bvec, sph.vertices, sph.faces, 1.25, nthreads=4)
    else:
        ref = orc.dsi_rec(sub, msub, bval, bvec, sph.vertices, sph.faces, 32, nthreads=4)
    godf = out["odf"][:, tidx].T.cpu().numpy().reshape(nsamp, 1, 1, -1)
    scale = np.abs(ref["odf"]).max(axis=3, keepdims=True) + 1e-30
    assert (np.abs(godf - ref["odf"]) / scale).max() <= odf_rtol
    if kind == "dsi":
        gpdf = out["pdf"][:, tidx].T.cpu().numpy().reshape(nsamp, 1, 1, -1)
        ps = np.abs(ref["pdf"]).max(axis=3, keepdims=True) + 1e-30
        assert (np.abs(gpdf - ref["pdf"]) / ps).max() <= 5e-5
    # peaks: identical, or a tie at rounding level in the oracle's own ODF (margin <= 1e-4 of the maximum); no allowance by count
    from util import peak_mismatches_are_ties
    gps = [out["peak"][k][:, tidx].T.cpu().numpy().reshape(nsamp, 1, 1, 3) for k in range(3)]
    peak_mismatches_are_ties(ref["odf"], ref["peak"], gps, np.asarray(sph.vertices, np.float32)[:sph.nvert], faces=np.asarray(sph.faces))
    return ref


def _odf_properties(fj, out, mask, plan, dwi, dev, odf_gain):
    """whole-volume properties + exact homogeneity of the contraction under power-of-two scaling"""
    import torch
    live = mask.bool()
    assert float(out["odf"][:, ~live].abs().max()) == 0.0
    for k in range(3):
        assert float(out["peak"][k][:, ~live].abs().max()) == 0.0 and float(out["qa"][k][~live].abs().max()) == 0.0
    # every non-zero peak direction is a first-half vertex of the tessellation (gqi.jl:154-155)
    verts = torch.from_numpy(np.ascontiguousarray(fj.sphere_642.vertices[:fj.sphere_642.nvert])).to(dev)
    pk = out["peak"][0][:, live].T
    nz = pk.abs().sum(1) > 0
    sample = pk[nz][:: max(1, int(nz.sum()) // 20000)]
    d = torch.cdist(sample, verts, compute_mode="donot_use_mm_for_euclid_dist")
    assert float(d.min(1).values.max()) == 0.0
    # QA is sorted (descending peaks) and non-negative after normalisation
    q0, q1, q2 = (out["qa"][k][live] for k in range(3))
    assert bool((q0 >= q1).all()) and bool((q1 >= q2).all()) and float(q2.min()) >= 0.0
    # GQI: odf(4 s) == 4 odf(s) bit for bit; DSI: odf and pdf are normalised by sum(p) ~ s(q=0), so they do not change
    # at all; peaks identical; the normalised qa identical
    out4 = fj.odf_rec_device(plan, dwi * 4.0, mask)
    torch.cuda.synchronize()
    assert torch.equal(out4["odf"], out["odf"] * odf_gain)
    for k in range(3):
        assert torch.equal(out4["peak"][k], out["peak"][k])
        torch.testing.assert_close(out4["qa"][k], out["qa"][k], rtol=3e-7, atol=0)
    if "pdf" in out:                                          # the pdf is normalised by sum(p): scale invariant
        assert torch.equal(out4["pdf"], out["pdf"])


def test_c3_gqi_140cubed(fj, orc, dev):
    import torch
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_gqi()
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=3, device=dev)
    mask = phantom.ball_mask_torch(SHAPE, dev)
    plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642, sigma=1.25, device=0)
    out = fj.odf_rec_device(plan, dwi, mask)
    torch.cuda.synchronize()
    out = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in out.items()}
    _check_odf_sample(fj, orc, "gqi", out, dwi, mask, bval, bvec, dev, 1500, 2e-5)
    _odf_properties(fj, out, mask, plan, dwi, dev, 4.0)


@pytest.fixture(scope="module")
def dsi_result(fj, dev):
    import torch
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_dsi()
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=5, device=dev)
    mask = phantom.ball_mask_torch(SHAPE, dev)
    plan = fj.OdfPlan("dsi", bval, bvec, fj.sphere_642, hann_width=32, device=0)
    out = fj.odf_rec_device(plan, dwi, mask)
    torch.cuda.synchronize()
    out = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in out.items()}
    return dict(bval=bval, bvec=bvec, dwi=dwi, mask=mask, plan=plan, out=out)


def test_c5_dsi_140cubed(fj, orc, dev, dsi_result):
    r = dsi_result
    _check_odf_sample(fj, orc, "dsi", r["out"], r["dwi"], r["mask"], r["bval"], r["bvec"], dev, 300, 1e-4)
    _odf_properties(fj, r["out"], r["mask"], r["plan"], r["dwi"], dev, 1.0)


def test_dsi_fold_prepass_in_front_of_the_split_kernels_past_the_fold_span_limit(fj, orc, dev):
    """The split kernels fold the antipodal sample pairs inside their sample requests through ONE 32-bit buffer range per stage and
    side; a stage whose 16 folded samples span more than 0xE0000000 / (4 nvox) frames does not fit it, and the step runs the fold
    pre-pass (dsi_fold4_kernel) in front of the unfused split kernel + the separate peak finder instead.  A DSI table whose frames
    come in a scrambled order reaches that at 140^3 (span ~ 500 of 515 frames): the pre-pass must have run, and the result meets the
    oracle like every other path (dsi.jl:171-270; the reference does not care about the frame order either)."""
    import ctypes as C
    import torch
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_dsi()
    perm = np.random.default_rng(23).permutation(len(bval))
    bval, bvec = np.ascontiguousarray(bval[perm]), np.ascontiguousarray(bvec[perm])
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=6, device=dev)
    mask = phantom.ball_mask_torch(SHAPE, dev)
    plan = fj.OdfPlan("dsi", bval, bvec, fj.sphere_642, hann_width=32, device=0)
    L = fj.lib()
    L.fib_profile_enable(1); L.fib_profile_reset()
    out = fj.odf_rec_device(plan, dwi, mask)
    torch.cuda.synchronize()
    ms, cnt = C.c_double(0), C.c_int64(0)
    L.fib_profile_get(b"dsi_fold", C.byref(ms), C.byref(cnt))
    L.fib_profile_enable(0)
    assert cnt.value == 1, "the fold pre-pass did not run: the table's stages no longer exceed the span limit at this size"
    assert plan.format == "fp16x2"
    _check_odf_sample(fj, orc, "dsi", out, dwi, mask, bval, bvec, dev, 200, 1e-4)
    live = mask.bool()
    assert float(out["odf"][:, ~live].abs().max()) == 0.0 and float(out["pdf"][:, ~live].abs().max()) == 0.0


def _check_tract(res, nvec_field, mask, nseed, nsub, step=0.5, len_min=3, len_max=140):
    """size-independent properties of a packed tract (stream.jl:625-690, 761-787)"""
    import torch
    npts, sidx, xyz = res["npts"], res["seed_index"], res["xyz"]
    nl = int(npts.numel())
    assert int(npts.sum()) == xyz.shape[0]
    assert int(npts.min()) >= len_min and int(npts.max()) <= len_max + 2          # cumulative cap (stream.jl:674)
    assert bool((sidx[1:] > sidx[:-1]).all()) and int(sidx.max()) < nseed * nsub      # reference order, no duplicates
    assert bool(torch.isfinite(xyz).all())
    # every point lies in a voxel of the (tracking) mask: round-half-even to 1-based voxel indices (stream.jl:514-520)
    v = torch.round(xyz).long() - 1
    assert int(v.min()) >= 0 and int(v.max()) < 140
    lin = v[:, 0] + 140 * (v[:, 1] + 140 * v[:, 2])
    assert bool(mask[lin].bool().all())
    # consecutive points are one step apart, except the duplicated seed point where the two directions meet
    off = torch.cumsum(npts.long(), 0)
    d = (xyz[1:] - xyz[:-1]).norm(dim=1)
    inner = torch.ones(xyz.shape[0] - 1, dtype=torch.bool, device=xyz.device)
    inner[off[:-1] - 1] = False                                # gaps between lines
    dd = d[inner]
    is_step = (dd - step).abs() < 2e-5
    is_seed = dd == 0.0
    assert bool((is_step | is_seed).all())
    assert int(is_seed.sum()) <= nl                            # at most one seed junction per line
    return nl


def test_c4_stream_1m_seeds(fj, orc, dev):
    import torch
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_dti(60, 4, 1000.0, seed=2)
    dwi, _ = phantom.make_dwi_torch(SHAPE, bval, bvec, seed=2, device=dev, nfib=1)
    ones = torch.ones(NVOX, dtype=torch.uint8, device=dev)
    o = fj.dti_fit_device(fj.DtiPlan(bval, bvec, device=0), dwi, ones)
    bm = phantom.ball_mask_torch(SHAPE, dev)
    field, mout = fj.stream_field_device([o["eigvec1"]], fa=o["fa"], fa_thresh=0.1, mask=bm)
    seeds = torch.nonzero(mout).flatten()
    assert 9.0e5 < seeds.numel() <= 998592
    sub = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float32, device=dev)
    res = fj.stream_device(field, SHAPE, seeds, sub)
    torch.cuda.synchronize()
    nl = _check_tract(res, 1, mout, int(seeds.numel()), 1)
    assert nl > 0.9 * seeds.numel()
    # oracle on the same field for 400 sampled seeds: identical lines (bit exact)
    rng = np.random.default_rng(3)
    pick = np.sort(rng.choice(int(seeds.numel()), 400, replace=False))
    ev = o["eigvec1"].reshape(3, NVOX).T.contiguous().cpu().numpy().reshape(140, 140, 140, 3, order="F")
    fa = o["fa"].cpu().numpy().reshape(140, 140, 140, order="F")
    mk = bm.cpu().numpy().reshape(140, 140, 140, order="F")
    seedvol = np.zeros(NVOX, np.uint8)
    seedvol[seeds[torch.from_numpy(pick).to(dev)].cpu().numpy()] = 1
    ref = orc.stream(np.asfortranarray(ev), sub.cpu().numpy(), fa=fa, fa_thresh=0.1, mask=mk,
                     seed=seedvol.reshape(140, 140, 140, order="F"), nthreads=8)
    sidx = res["seed_index"].cpu().numpy()
    off = np.concatenate([[0], np.cumsum(res["npts"].cpu().numpy().astype(np.int64))])
    xyz = res["xyz"].cpu().numpy()
    pos = {int(s): i for i, s in enumerate(sidx)}
    roff = np.concatenate([[0], np.cumsum(ref["npts"].astype(np.int64))])
    assert len(ref["npts"]) > 300
    for j, rs in enumerate(ref["seed_index"]):                 # oracle numbers its 400 seeds 0..399 in findall order
        i = pos[int(pick[int(rs)])]
        assert np.array_equal(xyz[off[i]:off[i + 1]], ref["xyz"][roff[j]:roff[j + 1]])
    # [r5] the one-call form on the same field: 1 M lines take trace + scan + pack (below the fused kernel's 2^21 lines), 3 M lines (three
    # offsets) take the fused trace + look-back + pack kernel with ONE vector per voxel -- both bit-identical to the two-call form
    one = fj.stream_device_run(field, SHAPE, seeds, sub, buffers=fj.StreamBuffers(dev))
    for k in ("npts", "seed_index", "xyz"):
        assert torch.equal(one[k], res[k]), k
    sub3 = torch.tensor([[0.1, -0.2, 0.3], [-0.3, 0.25, 0.0], [0.0, 0.4, -0.45]], dtype=torch.float32, device=dev)
    two3 = fj.stream_device(field, SHAPE, seeds, sub3)
    import ctypes as C
    L = fj.lib()
    L.fib_profile_enable(1); L.fib_profile_reset()
    one3 = fj.stream_device_run(field, SHAPE, seeds, sub3, buffers=fj.StreamBuffers(dev))
    torch.cuda.synchronize()
    ms, cnt = C.c_double(0), C.c_int64(0)
    L.fib_profile_get(b"stream_pack", C.byref(ms), C.byref(cnt))
    L.fib_profile_enable(0)
    assert cnt.value == 0 and int(seeds.numel()) * 3 >= (1 << 21)          # (no separate pack launch: the fused kernel ran)
    for k in ("npts", "seed_index", "xyz"):
        assert torch.equal(one3[k], two3[k]), k


def test_c5_stream_three_peaks_10m(fj, dev, dsi_result):
    import torch
    r = dsi_result
    out = r["out"]
    field, mout = fj.stream_field_device(out["peak"], f=out["qa"], f_thresh=0.03, mask=r["mask"])
    seeds = torch.nonzero(mout).flatten()
    sub = torch.from_numpy(fj.make_sublist(10, np.random.default_rng(5))).to(dev)
    res = fj.stream_device(field, SHAPE, seeds, sub)
    torch.cuda.synchronize()
    nl = _check_tract(res, 3, mout, int(seeds.numel()), 10)
    assert nl > 5_000_000
    # oracle on the same three-peak field (f = qa, f_thresh = .03, stream.jl:135-139; first-max / NaN / sign rule of
    # stream_pick_by_angle!, stream.jl:340-374) for 300 sampled seeds x the 10 offsets: identical lines, bit for bit
    import fibers_jl_amd  # noqa: F401
    from oracle import oracle as orc
    rng = np.random.default_rng(7)
    pick = np.sort(rng.choice(int(seeds.numel()), 300, replace=False))
    vol = lambda t, c: t.reshape(c, NVOX).T.contiguous().cpu().numpy().reshape(140, 140, 140, c, order="F")
    ovs = [np.asfortranarray(vol(out["peak"][k], 3)) for k in range(3)]
    fs = [out["qa"][k].cpu().numpy().reshape(140, 140, 140, order="F") for k in range(3)]
    mk = r["mask"].cpu().numpy().reshape(140, 140, 140, order="F")
    seedvol = np.zeros(NVOX, np.uint8)
    seedvol[seeds[torch.from_numpy(pick).to(dev)].cpu().numpy()] = 1
    ref = orc.stream(ovs, sub.cpu().numpy(), f=fs, f_thresh=0.03, mask=mk, seed=seedvol.reshape(140, 140, 140, order="F"), nthreads=8)
    sidx = res["seed_index"].cpu().numpy()
    off = np.concatenate([[0], np.cumsum(res["npts"].cpu().numpy().astype(np.int64))])
    pos = {int(s_): i for i, s_ in enumerate(sidx)}
    roff = np.concatenate([[0], np.cumsum(ref["npts"].astype(np.int64))])
    assert len(ref["npts"]) > 1500                            # most of the 3000 (seed, offset) pairs give a line of >= 3 points
    xyz = res["xyz"]
    for j, rs in enumerate(ref["seed_index"]):               # the oracle numbers its 300 seeds 0..299 in findall order, x 10 offsets
        sd, so = divmod(int(rs), 10)
        i = pos[int(pick[sd]) * 10 + so]
        assert np.array_equal(xyz[off[i]:off[i + 1]].cpu().numpy(), ref["xyz"][roff[j]:roff[j + 1]]), (j, rs)
    assert len(ref["npts"]) == sum(1 for sd in pick for so in range(10) if int(sd) * 10 + so in pos)   # and no line more or less
    # [r5] the one-call form (fibd_stream_run): from 2^21 lines on the workgroup that traced 512 lines packs them itself behind a decoupled
    # look-back (fused_pack_block) -- the same lines, order, seed indices and points, bit for bit, as trace + scan + pack above; also into
    # buffers that are too small (the totals come back, nothing is written past the capacity, what is written is right)
    import ctypes as C
    L = fj.lib()
    L.fib_profile_enable(1); L.fib_profile_reset()
    bufs = fj.StreamBuffers(dev)
    one = fj.stream_device_run(field, SHAPE, seeds, sub, buffers=bufs)
    torch.cuda.synchronize()
    ms, cnt = C.c_double(0), C.c_int64(0)
    L.fib_profile_get(b"stream_pack", C.byref(ms), C.byref(cnt))
    L.fib_profile_enable(0)
    assert cnt.value == 0, "the fused kernel was expected to run (no separate pack launch)"
    for k in ("npts", "seed_index", "xyz"):
        assert torch.equal(one[k], res[k]), k
    small = fj.StreamBuffers(dev)
    small.reserve(1000, 100000)
    small.xyz.fill_(-7.0)
    import sys
    smod = sys.modules[fj.stream_device_run.__module__]
    from fibers_jl_amd import _lib
    prm = smod._params(SHAPE, 3, 3, None, 45, 0.5, 0.2, 0, 10, smod.default_workspace(0))
    nl2, np2 = C.c_int64(0), C.c_int64(0)
    rc = L.fibd_stream_run(C.byref(prm), field.data_ptr(), seeds.data_ptr(), seeds.numel(), sub.data_ptr(), sub.shape[0], small.npts.data_ptr(),
                           small.seed_index.data_ptr(), 1000, small.xyz.data_ptr(), 100000, C.byref(nl2), C.byref(np2), None)
    torch.cuda.synchronize()
    assert rc == _lib.FIB_ERR_CAPACITY and nl2.value == res["npts"].numel() and np2.value == res["xyz"].shape[0]
    nfit = int((torch.cumsum(res["npts"][:1000].long(), 0) <= 100000).sum())
    assert nfit > 100 and torch.equal(small.npts[:nfit], res["npts"][:nfit])
    npf = int(res["npts"][:nfit].sum())
    assert torch.equal(small.xyz[:npf], res["xyz"][:npf])
    assert bool((small.xyz[100000:] == -7.0).all())


# ---- the one-launch mask compaction at sizes where a chunk holds several sub-chunks and the helper workgroups take every path --------
@pytest.mark.parametrize("shape,dead", [((161, 163, 150), 0.10), ((161, 163, 150), 0.0), ((97, 101, 100), 0.55), ((203, 199, 204), 0.02)])
def test_mask_compaction_and_clearing_at_scale(fj, dev, shape, dead):
    """3.9 M / 1.0 M / 8.2 M voxels (not multiples of the 4 096-voxel sub-chunk; 4 to 9 sub-chunks per chunk): a random mask with 10 % /
    0 % / 55 % / 2 % of the voxels outside -- the helpers' span-by-span path, the nothing-to-do path, the clear-everything path and the
    voxel-by-voxel path.  A voxel's result depends on its own samples only, so the masked run must equal the all-ones run at the
    voxels inside, bit for bit, and be exactly zero outside -- in output buffers that start as NaN garbage."""
    import torch
    from fibers_jl_amd import phantom
    nvox = int(np.prod(shape))
    if nvox % 4:
        shape = (shape[0] + (4 - shape[0] % 4) % 4,) + shape[1:]          # (the fused scan wants whole quads; ragged volumes: test_gpu_odf)
        nvox = int(np.prod(shape))
    bval, bvec = phantom.scheme_gqi(2, 14, (1000.0, 2500.0), 7)            # 30 frames: a small contraction, the compaction dominates
    g = torch.Generator(device=dev); g.manual_seed(11)
    dwi = torch.rand((len(bval), nvox), device=dev, generator=g) * 1000.0 + 1.0
    plan = fj.OdfPlan("gqi", bval, bvec, fj.sphere_642)
    ones = torch.ones(nvox, dtype=torch.uint8, device=dev)
    full = fj.odf_rec_device(plan, dwi, ones, normalize=False)
    torch.cuda.synchronize()
    mask = (torch.rand(nvox, device=dev, generator=g) >= dead).to(torch.uint8) if dead > 0 else ones.clone()
    if dead > 0:                                                           # runs of voxels outside as well as scattered ones
        mask[nvox // 3: nvox // 3 + 70000] = 0
        mask[-5000:] = 0
    nan = float("nan")
    out = dict(odf=torch.full((plan.nvert, nvox), nan, device=dev), peak=[torch.full((3, nvox), nan, device=dev) for _ in range(3)],
               qa=[torch.full((nvox,), nan, device=dev) for _ in range(3)], odfmax=torch.empty(2, device=dev))
    got = fj.odf_rec_device(plan, dwi, mask, out=out, normalize=False)
    torch.cuda.synchronize()
    live = mask.bool()
    assert torch.equal(got["odf"][:, live], full["odf"][:, live])
    assert (got["odf"][:, ~live] == 0).all()
    for k in range(3):
        assert torch.equal(got["peak"][k][:, live], full["peak"][k][:, live]) and (got["peak"][k][:, ~live] == 0).all()
        assert torch.equal(got["qa"][k][live], full["qa"][k][live]) and (got["qa"][k][~live] == 0).all()
    assert not torch.isnan(got["odf"]).any()
    # the same buffers again, declared clean outside the mask: identical
    got2 = fj.odf_rec_device(plan, dwi, mask, out=out, normalize=False, out_prezeroed=True)
    torch.cuda.synchronize()
    assert (got2["odf"][:, ~live] == 0).all() and torch.equal(got2["odf"][:, live], full["odf"][:, live])
