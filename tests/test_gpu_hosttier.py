"""The host-buffer tier (fibers_hip.h: fib_init / fib_* with FIB_DEVICE_ALL): slab sharding over a device set, the chunk
pipeline, the plan cache and concurrent callers.  A 1-GPU box exercises the multi-worker paths with a device set that
names device 0 more than once (two or three pipelines on one GPU); results must not depend on the device set or on the
chunking (voxels are independent, gqi.jl:132-162; odfmax is a maximum, gqi.jl:164; seeds are independent,
stream.jl:764-767)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gqi_case(fj, shape=(22, 18, 14), seed=3):
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed=seed, crossing=True)
    rng = np.random.default_rng(seed)
    mask = (rng.random(shape) < 0.8).astype(np.uint8)
    return fj.MRI(dwi, bval, bvec), fj.MRI(np.asfortranarray(mask))


def _same_gqi(a, b):
    assert np.array_equal(a.odf.vol, b.odf.vol)
    for k in range(3):
        assert np.array_equal(a.peak[k].vol, b.peak[k].vol)
        assert np.array_equal(a.qa[k].vol, b.qa[k].vol, equal_nan=True)


def test_gqi_device_set_and_chunking_do_not_change_results(fj, orc, monkeypatch):
    dwi, mask = _gqi_case(fj)
    one = fj.gqi_rec(dwi, mask)                                  # one worker, one chunk
    ref = orc.gqi_rec(dwi.vol, mask.vol[..., 0], dwi.bval, dwi.bvec, fj.sphere_642.vertices, fj.sphere_642.faces, 1.25, nthreads=4)
    assert np.abs(one.odf.vol - ref["odf"]).max() <= 2e-5 * np.abs(ref["odf"]).max()
    try:
        monkeypatch.setenv("FIBERS_HOST_CHUNK", "1024")          # 6 chunks per call, ragged last chunk
        _same_gqi(fj.gqi_rec(dwi, mask), one)
        fj.init([0, 0, 0])                                       # three slabs, three host threads, one GPU
        _same_gqi(fj.gqi_rec(dwi, mask, device=fj.DEVICE_ALL), one)
        monkeypatch.delenv("FIBERS_HOST_CHUNK")
        fj.init([0, 0])
        _same_gqi(fj.gqi_rec(dwi, mask, device=fj.DEVICE_ALL), one)
    finally:
        fj.shutdown()


def test_dti_and_dsi_over_a_device_set(fj, monkeypatch):
    from fibers_jl_amd import phantom
    shape = (13, 11, 9)
    b2, g2 = phantom.scheme_dti(30, 3, 1000.0, seed=2)
    d2, _, _ = phantom.make_volume(shape, b2, g2, seed=5, nonpositive_frac=0.01)
    mask = fj.MRI(np.ones(shape, np.uint8))
    one = fj.dti_fit(fj.MRI(d2, b2, g2), mask)
    a1, s1 = fj.adc_fit(fj.MRI(d2, b2, g2), mask)
    b5, g5 = phantom.scheme_dsi()
    d5, _, _ = phantom.make_volume((6, 5, 4), b5, g5, seed=5)
    m5 = fj.MRI(np.ones((6, 5, 4), np.uint8))
    dsi1 = fj.dsi_rec(fj.MRI(d5, b5, g5), m5)
    try:
        monkeypatch.setenv("FIBERS_HOST_CHUNK", "1024")
        fj.init([0, 0])
        two = fj.dti_fit(fj.MRI(d2, b2, g2), mask, device=fj.DEVICE_ALL)
        for k in fj.dti.DTI_FIELDS:
            assert np.array_equal(getattr(one, k).vol, getattr(two, k).vol, equal_nan=True), k
        a2, s2 = fj.adc_fit(fj.MRI(d2, b2, g2), mask, device=fj.DEVICE_ALL)
        assert np.array_equal(a1.vol, a2.vol, equal_nan=True) and np.array_equal(s1.vol, s2.vol, equal_nan=True)
        dsi2 = fj.dsi_rec(fj.MRI(d5, b5, g5), m5, device=fj.DEVICE_ALL)
        assert np.array_equal(dsi1.pdf.vol, dsi2.pdf.vol) and np.array_equal(dsi1.odf.vol, dsi2.odf.vol)
        for k in range(3):
            assert np.array_equal(dsi1.peak[k].vol, dsi2.peak[k].vol) and np.array_equal(dsi1.qa[k].vol, dsi2.qa[k].vol, equal_nan=True)
    finally:
        fj.shutdown()


def test_stream_sharded_over_a_device_set_keeps_the_reference_order(fj, orc):
    from fibers_jl_amd import phantom
    n = 14
    ov = np.asfortranarray(phantom.fibre_field(n, n, n).astype(np.float32))
    m = np.asfortranarray(phantom.ball_mask(n, n, n, radius=5.5))
    sub = fj.make_sublist(3, np.random.default_rng(2))
    one = fj.stream(fj.MRI(ov), mask=fj.MRI(m), sublist=sub)
    ref = orc.stream(ov, sub, mask=m, nthreads=2)
    assert np.array_equal(one.npts, ref["npts"]) and np.array_equal(one.xyz, ref["xyz"])
    try:
        fj.init([0, 0, 0])
        three = fj.stream(fj.MRI(ov), mask=fj.MRI(m), sublist=sub, device=fj.DEVICE_ALL)
    finally:
        fj.shutdown()
    assert np.array_equal(three.npts, one.npts) and np.array_equal(three.seed_index, one.seed_index)
    assert np.array_equal(three.xyz, one.xyz)


def test_concurrent_callers_on_two_workers_and_on_one(fj):
    """two host threads, each with its own plan and pipeline (calls on different workers run concurrently; calls that
    share a worker are serialised by the library): every result equals the single-threaded one"""
    from fibers_jl_amd import _lib
    dwi_a, mask_a = _gqi_case(fj, seed=3)
    dwi_b, mask_b = _gqi_case(fj, shape=(16, 20, 12), seed=9)
    want_a, want_b = fj.gqi_rec(dwi_a, mask_a), fj.gqi_rec(dwi_b, mask_b)
    got, errs = {}, []

    def run(tag, dwi, mask, reps):
        try:
            for _ in range(reps):
                got[tag] = fj.gqi_rec(dwi, mask)
        except Exception as e:                       # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=("a", dwi_a, mask_a, 3)), threading.Thread(target=run, args=("b", dwi_b, mask_b, 3))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    _same_gqi(got["a"], want_a)
    _same_gqi(got["b"], want_b)
    # device-resident tier: two plans on two streams from two threads
    import torch
    from fibers_jl_amd import phantom
    dev = torch.device("cuda", 0)
    bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
    vols = [phantom.make_dwi_torch((20, 16, 12), bval, bvec, seed=s, device=dev)[0] for s in (1, 2)]
    nvox = vols[0].shape[1]
    mask = torch.ones(nvox, dtype=torch.uint8, device=dev)
    plans = [fj.OdfPlan("gqi", bval, bvec, fj.sphere_642) for _ in range(2)]
    want = [fj.odf_rec_device(plans[i], vols[i], mask) for i in range(2)]
    torch.cuda.synchronize()
    want = [{k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in w.items()} for w in want]
    res = [None, None]

    def run_dev(i):
        try:
            st = torch.cuda.Stream(device=dev)
            for _ in range(5):
                res[i] = fj.odf_rec_device(plans[i], vols[i], mask, stream=st)
            st.synchronize()
        except Exception as e:                       # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run_dev, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for i in range(2):
        assert torch.equal(res[i]["odf"], want[i]["odf"]) and torch.equal(res[i]["odfmax"], want[i]["odfmax"])
        for k in range(3):
            assert torch.equal(res[i]["peak"][k], want[i]["peak"][k]) and torch.equal(res[i]["qa"][k], want[i]["qa"][k])
    assert _lib.lib().fib_last_error() is not None


def test_many_concurrent_calls_of_different_sizes_on_one_worker(fj, monkeypatch):
    """Stress of the qa hand-over between the two passes of fib_gqi_rec (the worker's lock is released in between): four threads
    hammer ONE worker with volumes of different sizes and chunkings; a call must never see another call's qa (or a buffer the
    other call reallocated): every result equals the single-threaded one, bit for bit."""
    shapes = [(22, 18, 14), (9, 7, 5), (31, 12, 10), (16, 20, 12)]
    cases = [_gqi_case(fj, shape=sh, seed=20 + i) for i, sh in enumerate(shapes)]
    want = [fj.gqi_rec(d, m) for d, m in cases]
    monkeypatch.setenv("FIBERS_HOST_CHUNK", "1024")              # several chunks per call: the passes interleave more
    errs = []

    def run(i):
        try:
            d, m = cases[i]
            for _ in range(12):
                _same_gqi(fj.gqi_rec(d, m), want[i])
        except BaseException as e:                               # noqa: BLE001
            errs.append((i, e))
    th = [threading.Thread(target=run, args=(i,)) for i in range(len(cases))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def test_many_concurrent_dsi_calls_on_one_worker(fj):
    """The DSI pair kernel keeps sample requests in flight across stages, items and its epilogue with hand-counted waits (the fused
    GQI kernel's scheme: a hole in it once showed only under concurrent launches).  Three threads hammer one worker with
    volumes of different sizes (several work items per workgroup, ragged ends, a masked volume): every result equals the
    single-threaded one, bit for bit."""
    from fibers_jl_amd import phantom
    b5, g5 = phantom.scheme_dsi()
    shapes = [(40, 36, 10), (9, 7, 5), (33, 30, 9)]
    cases = []
    for i, sh in enumerate(shapes):
        d5, _, _ = phantom.make_volume(sh, b5, g5, seed=40 + i, crossing=True)
        m = np.ones(sh, np.uint8) if i != 2 else (np.random.default_rng(7).random(sh) < 0.7).astype(np.uint8)
        cases.append((fj.MRI(d5, b5, g5), fj.MRI(np.asfortranarray(m))))
    want = [fj.dsi_rec(d, m) for d, m in cases]

    def same(a, b):
        assert np.array_equal(a.pdf.vol, b.pdf.vol) and np.array_equal(a.odf.vol, b.odf.vol)
        for k in range(3):
            assert np.array_equal(a.peak[k].vol, b.peak[k].vol) and np.array_equal(a.qa[k].vol, b.qa[k].vol, equal_nan=True)
    errs = []

    def run(i):
        try:
            d, m = cases[i]
            for _ in range(6):
                same(fj.dsi_rec(d, m), want[i])
        except BaseException as e:                               # noqa: BLE001
            errs.append((i, e))
    th = [threading.Thread(target=run, args=(i,)) for i in range(len(cases))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def test_packing_the_voxels_inside_the_mask_does_not_change_results(fj, monkeypatch):
    """With a mask that leaves a good part of the volume out the host tier moves the voxels inside only (runs packed densely
    into the pinned ring, put back and the gaps zero-filled behind the device).  Same results bit for bit as the unpacked
    pipeline (FIBERS_HOST_PACK=0), for every fit, with several chunks per call: a ball, isolated voxels, a mask that keeps nothing,
    one that keeps the first and the last voxel only, runs across chunk boundaries."""
    from fibers_jl_amd import phantom
    shape = (23, 19, 11)
    nvox = int(np.prod(shape))
    rng = np.random.default_rng(77)
    ball = np.zeros(shape, np.uint8)
    x, y, z = np.meshgrid(*[np.arange(n) - (n - 1) / 2 for n in shape], indexing="ij")
    ball[(x / 10.0) ** 2 + (y / 8.0) ** 2 + (z / 4.5) ** 2 <= 1.0] = 1
    ends = np.zeros(nvox, np.uint8); ends[0] = 1; ends[-1] = 1
    masks = {"ball": ball, "sparse": (rng.random(shape) < 0.03).astype(np.uint8), "none": np.zeros(shape, np.uint8),
             "ends": ends.reshape(shape, order="F"), "half": (np.arange(nvox) % 700 < 300).astype(np.uint8).reshape(shape, order="F")}
    bg, gg = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
    dg, _, _ = phantom.make_volume(shape, bg, gg, seed=5, crossing=True)
    b2, g2 = phantom.scheme_dti(30, 3, 1000.0, seed=2)
    d2, _, _ = phantom.make_volume(shape, b2, g2, seed=6, nonpositive_frac=0.01)
    b5, g5 = phantom.scheme_dsi()
    d5, _, _ = phantom.make_volume((9, 8, 7), b5, g5, seed=7)
    monkeypatch.setenv("FIBERS_HOST_CHUNK", "1024")
    for name, m in masks.items():
        mk = fj.MRI(np.asfortranarray(m))
        res = {}
        for pack in ("0", "1"):
            monkeypatch.setenv("FIBERS_HOST_PACK", pack)
            gq = fj.gqi_rec(fj.MRI(dg, bg, gg), mk)
            dt = fj.dti_fit(fj.MRI(d2, b2, g2), mk)
            ad = fj.adc_fit(fj.MRI(d2, b2, g2), mk)
            res[pack] = (gq, dt, ad)
        (g0, t0, a0), (g1, t1, a1) = res["0"], res["1"]
        _same_gqi(g0, g1)
        for k in fj.dti.DTI_FIELDS:
            assert np.array_equal(getattr(t0, k).vol, getattr(t1, k).vol, equal_nan=True), (name, k)
        assert np.array_equal(a0[0].vol, a1[0].vol, equal_nan=True) and np.array_equal(a0[1].vol, a1[1].vol, equal_nan=True), name
        dead = m.reshape(-1, order="F") == 0
        assert not g1.odf.vol.reshape(nvox, -1, order="F")[dead].any(), name
    m5 = (rng.random((9, 8, 7)) < 0.4).astype(np.uint8)
    monkeypatch.setenv("FIBERS_HOST_CHUNK", "1024")
    out = {}
    for pack in ("0", "1"):
        monkeypatch.setenv("FIBERS_HOST_PACK", pack)
        out[pack] = fj.dsi_rec(fj.MRI(d5, b5, g5), fj.MRI(np.asfortranarray(m5)))
    assert np.array_equal(out["0"].pdf.vol, out["1"].pdf.vol) and np.array_equal(out["0"].odf.vol, out["1"].odf.vol)
    for k in range(3):
        assert np.array_equal(out["0"].peak[k].vol, out["1"].peak[k].vol) and np.array_equal(out["0"].qa[k].vol, out["1"].qa[k].vol, equal_nan=True)


def test_outputs_zeroed_flag_leaves_the_voxels_outside_the_mask_alone(fj, monkeypatch):
    """FIB_MASK_OUTPUTS_ZEROED in mask_dtype (what the Julia / Python wrappers pass: they allocate zeros, as the reference does): the scatter
    stage writes the runs inside the mask only.  Without the flag every output voxel is written -- arrays that held garbage read 0 outside
    the mask; with it the same arrays keep their garbage there (when the voxels inside travel packed -- the flag is a permission: a mask
    that keeps most of the volume, or in short runs, goes through whole and its rows are written whole), and the voxels inside are identical."""
    import ctypes as C
    from fibers_jl_amd import _lib, phantom
    shape = (64, 13, 9)                                              # (long rows: runs of >= 16 voxels on average, so the voxels inside travel packed)
    nx, ny, nz = shape
    nvox = nx * ny * nz
    x, y, z = np.meshgrid(*[np.arange(n) - (n - 1) / 2 for n in shape], indexing="ij")
    ball = ((x / 30.0) ** 2 + (y / 5.5) ** 2 + (z / 3.5) ** 2 <= 1.0)
    m8 = np.ascontiguousarray(ball.reshape(-1, order="F").astype(np.uint8))
    bg, gg = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
    dg, _, _ = phantom.make_volume(shape, bg, gg, seed=5, crossing=True)
    host = np.ascontiguousarray(np.asarray(dg, np.float32).reshape(nvox, -1, order="F").T)           # [nvol][nvox]
    sph = fj.sphere_642
    v = np.asfortranarray(sph.vertices, np.float32); f = np.asfortranarray(sph.faces, np.int32)
    bv = np.ascontiguousarray(bg, np.float32); bvec = np.asfortranarray(np.asarray(gg, np.float32))
    nvol, nvert = len(bg), sph.nvert
    monkeypatch.setenv("FIBERS_HOST_CHUNK", "1024")
    L = _lib.lib()

    def run(flag):
        odf = np.full((nvert, nvox), 7.5, np.float32)
        pk = [np.full((3, nvox), 7.5, np.float32) for _ in range(3)]
        qa = [np.full(nvox, 7.5, np.float32) for _ in range(3)]
        _lib.check(L.fib_gqi_rec(0, host.ctypes.data, nx, ny, nz, nvol, m8.ctypes.data, 0 | flag, bv.ctypes.data, bvec.ctypes.data, v.ctypes.data,
                                 v.shape[0], f.ctypes.data, f.shape[0], 1.25, odf.ctypes.data, _lib.P3(*[a.ctypes.data for a in pk]),
                                 _lib.P3(*[a.ctypes.data for a in qa])))
        return odf, pk, qa
    o0, p0, q0 = run(0)
    o1, p1, q1 = run(_lib.FIB_MASK_OUTPUTS_ZEROED)
    inside = m8 != 0
    assert inside.sum() > 500 and (~inside).sum() > 500
    assert not o0[:, ~inside].any() and not any(p[:, ~inside].any() for p in p0)                        # every voxel written
    assert np.array_equal(o0[:, inside], o1[:, inside]) and all(np.array_equal(a[:, inside], b[:, inside]) for a, b in zip(p0, p1))
    assert (o1[:, ~inside] == 7.5).all() and all((p[:, ~inside] == 7.5).all() for p in p1)             # left alone
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(q0, q1))                            # (the qa planes are written in full either way)


def test_gqi_unaligned_volume_does_not_depend_on_chunks_or_device_set(fj, orc, monkeypatch):
    """nvox % 4 != 0 (13 x 11 x 9 = 1287): the same kernel choice for every chunk, so chunk size and device set do not change
    a bit (the fused peak kernel needs 16-byte aligned rows; the choice is made from the whole volume, not per chunk)"""
    dwi, mask = _gqi_case(fj, shape=(13, 11, 9), seed=31)
    assert dwi.vol[..., 0].size % 4 != 0
    one = fj.gqi_rec(dwi, mask)
    ref = orc.gqi_rec(dwi.vol, mask.vol[..., 0], dwi.bval, dwi.bvec, fj.sphere_642.vertices, fj.sphere_642.faces, 1.25, nthreads=4)
    assert np.abs(one.odf.vol - ref["odf"]).max() <= 2e-5 * np.abs(ref["odf"]).max()
    try:
        for chunk in ("1024", "2048"):
            monkeypatch.setenv("FIBERS_HOST_CHUNK", chunk)
            _same_gqi(fj.gqi_rec(dwi, mask), one)
        fj.init([0, 0, 0])
        _same_gqi(fj.gqi_rec(dwi, mask, device=fj.DEVICE_ALL), one)
        monkeypatch.delenv("FIBERS_HOST_CHUNK")
        _same_gqi(fj.gqi_rec(dwi, mask, device=fj.DEVICE_ALL), one)
    finally:
        fj.shutdown()


def test_errors_come_back_as_codes_not_exceptions(fj):
    """invalid arguments through the multi-worker path: a status code and a message from the worker thread, the process
    lives on (no C++ exception crosses the ABI)"""
    import ctypes as C
    from fibers_jl_amd import _lib
    dwi, mask = _gqi_case(fj)
    L = _lib.lib()
    try:
        fj.init([0, 0])
        with pytest.raises(fj.FibersError) as ei:                # tessellation with an odd vertex count: plan creation fails in the workers
            odd = fj.ODF(fj.sphere_642.vertices[:-1], fj.sphere_642.faces)
            fj.gqi_rec(dwi, mask, odf_dirs=odd, device=fj.DEVICE_ALL)
        assert "tessellation" in str(ei.value)
        bad = np.array([7], np.int32)
        assert L.fib_init(1, bad.ctypes.data) == -2              # FIB_ERR_NO_DEVICE
        assert b"not available" in L.fib_last_error()
        rc = L.fib_gqi_rec(0, None, 2, 2, 2, 3, None, 0, None, None, None, 0, None, 0, C.c_float(1.25), None, _lib.P3(), _lib.P3())
        assert rc < 0
    finally:
        fj.shutdown()
    _same_gqi(fj.gqi_rec(dwi, mask), fj.gqi_rec(dwi, mask))      # still works afterwards


def test_fib_trim_returns_the_kept_buffers_and_the_next_call_is_unaffected(fj):
    """fib_trim (the host tier kept GBs of HBM between calls with no way to give them back): the pinned ring's device mirror,
    fib_stream's device buffers and the tracer's workspace go back to the driver, plans stay; the next calls re-allocate and return the
    same results (outputs allocated zero-filled by the caller, mri.jl:249-265)."""
    import torch
    from fibers_jl_amd import phantom
    bval, bvec = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
    dwi, _, _ = phantom.make_volume((16, 16, 12), bval, bvec, seed=3)
    big = np.asfortranarray(np.tile(dwi, (3, 3, 3, 1)))
    mask = np.ones(big.shape[:3], np.uint8)
    g1 = fj.gqi_rec(fj.MRI(big, bval, bvec), fj.MRI(mask))
    ov = np.asfortranarray(phantom.fibre_field(24, 24, 24).astype(np.float32))
    sub = np.array([[0.1, -0.2, 0.3]], np.float32)
    t1 = fj.stream(fj.MRI(ov), mask=fj.MRI(np.ones((24, 24, 24), np.uint8)), sublist=sub)
    torch.cuda.synchronize()
    held = torch.cuda.mem_get_info()[0]
    fj.trim()
    freed = torch.cuda.mem_get_info()[0] - held
    assert freed > 100 << 20, freed                                            # the ring's device mirror alone is several hundred MB
    fj.trim()                                                                  # (idempotent)
    g2 = fj.gqi_rec(fj.MRI(big, bval, bvec), fj.MRI(mask))
    t2 = fj.stream(fj.MRI(ov), mask=fj.MRI(np.ones((24, 24, 24), np.uint8)), sublist=sub)
    assert np.array_equal(g1.odf.vol, g2.odf.vol) and all(np.array_equal(a.vol, b.vol, equal_nan=True) for a, b in zip(g1.qa, g2.qa))
    assert np.array_equal(t1.npts, t2.npts) and np.array_equal(t1.xyz, t2.xyz)
