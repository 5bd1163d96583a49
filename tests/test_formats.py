"""Host-side formats either side of the hot path: NIfTI-1 + b-tables (N2), .trk (N1).
CPU tests check the wire formats against the published layouts and the reference's conventions
(mri.jl:1394-1672, 1695-1919, 2059-2266; trk.jl:88-144, 358-495); the GPU test checks that the device
serialiser emits byte-identical .trk bodies."""
import gzip
import os
import struct

import numpy as np
import pytest


def _affine():
    th = 0.3
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    M = np.eye(4)
    M[:3, :3] = R @ np.diag([1.5, 1.5, 2.0]) * np.array([-1, 1, 1])      # LAS-like, det < 0 -> qfac = -1
    M[:3, 3] = [90.0, -126.0, -72.0]
    return M.astype(np.float32)


@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
def test_nifti_roundtrip_and_header_layout(fj, tmp_path, ext):
    rng = np.random.default_rng(0)
    vol = np.asfortranarray(rng.normal(size=(5, 4, 3, 7)).astype(np.float32))
    bval = np.array([0, 1000, 1000, 2000, 2000, 3000, 3000], np.float32)
    bvec = rng.normal(size=(7, 3)).astype(np.float32); bvec[0] = 0
    m = fj.MRI(vol, bval, bvec, volres=(1.5, 1.5, 2.0), vox2ras=_affine())
    f = str(tmp_path / ("dwi" + ext))
    assert fj.mri_write(m, f) is False
    raw = (gzip.open(f, "rb") if ext.endswith("gz") else open(f, "rb")).read()
    assert struct.unpack("<i", raw[:4])[0] == 348 and raw[344:348] == b"n+1\0"
    assert struct.unpack("<8h", raw[40:56]) == (4, 5, 4, 3, 7, 1, 1, 1)
    assert struct.unpack("<hh", raw[70:74]) == (16, 32) and struct.unpack("<f", raw[108:112])[0] == 352.0
    assert struct.unpack("<hh", raw[252:256]) == (1, 1)                     # qform_code, sform_code
    assert np.allclose(struct.unpack("<4f", raw[280:296]), _affine()[0])
    assert struct.unpack("<f", raw[76:80])[0] == -1.0                       # qfac for a left-handed affine
    assert len(raw) == 352 + vol.size * 4
    r = fj.mri_read(f)
    assert r.vol.dtype == np.float32 and np.array_equal(r.vol, vol)
    assert np.allclose(r.vox2ras, _affine(), atol=1e-5) and np.allclose(r.volres, (1.5, 1.5, 2.0), atol=1e-5)
    assert np.allclose(r.niftihdr["qform"], _affine(), atol=1e-4)           # quaternion path reproduces the affine
    assert np.array_equal(r.bval, bval)
    nrm = np.linalg.norm(r.bvec, axis=1)
    assert nrm[0] == 0 and np.allclose(nrm[1:], 1, atol=1e-6)               # normalised, 0/0 -> 0 (mri.jl:711-712)


def test_nifti_int16_bigendian_and_scaling(fj, tmp_path):
    vol = np.arange(2 * 3 * 4, dtype=np.int16).reshape(2, 3, 4, order="F")
    m = fj.MRI(vol, vox2ras=_affine(), volres=(1.5, 1.5, 2.0))
    f = str(tmp_path / "a.nii")
    fj.mri_write(m, f)
    raw = bytearray(open(f, "rb").read())
    from fibers_jl_amd import nifti
    fields = list(nifti._HDR.unpack(bytes(raw[:348])))
    fields[31], fields[32] = 2.0, 1.0                                        # scl_slope, scl_inter
    be = struct.Struct(">" + nifti._HDR.format[1:]).pack(*fields) + b"\0" * 4 + vol.astype(">i2").tobytes(order="F")
    g = str(tmp_path / "b.nii")
    open(g, "wb").write(be)
    r = fj.mri_read(g)
    assert r.niftihdr["do_bswap"] and r.vol.dtype == np.int16
    assert np.array_equal(r.vol[..., 0], vol * 2 + 1)                        # dtype.(vol*slope + inter), mri.jl:1664-1668
    with pytest.raises(ValueError, match="Invalid header size"):
        nifti.load_nifti_hdr(b"\1\2\3\4" + bytes(raw[4:348]))


def test_bfiles_any_order_and_layout(fj, tmp_path):
    b = np.array([5, 1000, 2000, 3000], np.float32)
    g = np.array([[1, 0, 0], [0, 1, 0], [0, 0.6, 0.8], [0.6, 0, 0.8]], np.float32)   # (a 3x3 table is ambiguous: kept as is)
    fb, fg = str(tmp_path / "x.bval"), str(tmp_path / "x.bvec")
    np.savetxt(fb, b[None, :])                                               # single row
    np.savetxt(fg, g.T)                                                      # 3 rows (FSL layout)
    for a, c in ((fb, fg), (fg, fb)):
        bb, gg = fj.mri_read_bfiles(a, c)
        assert np.array_equal(bb, b) and np.allclose(gg, g)
    np.savetxt(fg, g[:2])
    with pytest.raises(ValueError):
        fj.mri_read_bfiles(fb, fg)


def test_result_struct_write_and_reload(fj, tmp_path):
    ref = fj.MRI(np.zeros((3, 3, 2), np.float32), volres=(2, 2, 2), vox2ras=_affine())
    rng = np.random.default_rng(1)
    mk = lambda n: fj.MRI(np.asfortranarray(rng.normal(size=(3, 3, 2, n)).astype(np.float32)), volres=ref.volres, vox2ras=ref.vox2ras)
    gqi = fj.GQI(mk(5), [mk(3) for _ in range(3)], [mk(1) for _ in range(3)])
    base = str(tmp_path / "sub01")
    fj.gqi_write(gqi, base)
    names = sorted(os.listdir(tmp_path))
    assert names == ["sub01_odf.nii.gz"] + ["sub01_peak%d.nii.gz" % k for k in (1, 2, 3)] + ["sub01_qa%d.nii.gz" % k for k in (1, 2, 3)]
    back = fj.read_struct(base, fj.GQI)
    assert np.array_equal(back.odf.vol, gqi.odf.vol) and all(np.array_equal(a.vol, b.vol) for a, b in zip(back.peak, gqi.peak))


def test_rumba_write_files(fj, tmp_path):
    """rumba_write (rusd.jl:645-663): <base>_<field>.nii.gz per volume, <base>_peak<k>.nii.gz, scalars as one-line .txt"""
    rng = np.random.default_rng(3)
    mk = lambda n: fj.MRI(np.asfortranarray(rng.normal(size=(3, 2, 2, n)).astype(np.float32)), volres=(2, 2, 2), vox2ras=_affine())
    r = fj.RUMBASD(fodf=mk(6), fgm=mk(1), fcsf=mk(1), peak=[mk(3) for _ in range(5)], gfa=mk(1), var=mk(1), snr_mean=12.539062, snr_std=0.1)
    base = str(tmp_path / "s")
    fj.rumba_write(r, base)
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(["s_%s.nii.gz" % k for k in ("fodf", "fgm", "fcsf", "gfa", "var")] + ["s_peak%d.nii.gz" % k for k in range(1, 6)] +
                           ["s_snr_mean.txt", "s_snr_std.txt"])
    assert open(base + "_snr_mean.txt").read() == "12.539062\n" and open(base + "_snr_std.txt").read() == "0.1\n"
    assert np.array_equal(fj.mri_read(base + "_peak4.nii.gz").vol, r.peak[3].vol)


def test_trk_header_body_and_roundtrip(fj, tmp_path):
    ref = fj.MRI(np.zeros((10, 12, 14), np.uint8), volres=(1.5, 1.5, 2.0), vox2ras=_affine())
    rng = np.random.default_rng(2)
    npts = np.array([3, 1, 5], np.int32)
    xyz = rng.uniform(1, 10, size=(9, 3)).astype(np.float32)
    tr = fj.Tract(xyz=xyz, npts=npts, volsize=ref.volsize, volres=ref.volres, vox2ras=ref.vox2ras)
    f = str(tmp_path / "t.trk")
    assert fj.trk_write(tr, f, ref) is False
    raw = open(f, "rb").read()
    assert len(raw) == 1000 + 4 * 3 + 12 * 9 and raw[:6] == b"TRACK\0"
    assert struct.unpack("<3h", raw[6:12]) == (10, 12, 14)
    assert np.allclose(struct.unpack("<3f", raw[12:24]), (1.5, 1.5, 2.0))
    assert np.allclose(np.array(struct.unpack("<16f", raw[440:504])).reshape(4, 4), _affine())
    assert raw[948:952] == b"LAS\0" and struct.unpack("<3i", raw[988:1000]) == (3, 2, 1000)
    body = np.frombuffer(raw, np.float32, offset=1000)
    assert body.view(np.int32)[0] == 3 and body.view(np.int32)[10] == 1
    want = ((xyz[0].astype(np.float64) + 0.5) * np.array([1.5, 1.5, 2.0])).astype(np.float32)   # trk.jl:475-476
    assert np.array_equal(body[1:4], want)
    back = fj.trk_read(f)
    assert np.array_equal(back.npts, npts) and np.allclose(back.xyz, xyz, atol=2e-6)
    assert len(back.str) == 3 and back.str[2].shape == (3, 5)
    fj.str_add(back, [np.ones((3, 4), np.float32)])
    assert back.nstr == 4 and back.npts[-1] == 4


def test_trk_scalars_and_properties_follow_the_reference_layout(fj, tmp_path):
    """trk_write interleaves the n_scalars values of a point after its xyz triple and appends the n_properties values of a
    line after its last point (trk.jl:471-482); trk_read returns them (trk.jl:404-416).  A Tract from an LCM run carries
    one scalar per point (stream.jl:787)."""
    ref = fj.MRI(np.zeros((4, 5, 6, 1), np.float32), volres=(2.0, 1.0, 0.5))
    rng = np.random.default_rng(1)
    npts = np.array([3, 1, 4], np.int32)
    xyz = rng.uniform(1, 4, (8, 3)).astype(np.float32)
    sc = rng.uniform(0, 1, (8, 2)).astype(np.float32)
    pr = rng.uniform(0, 1, (3, 3)).astype(np.float32)
    tr = fj.Tract(xyz=xyz, npts=npts, volsize=(4, 5, 6), volres=ref.volres, vox2ras=ref.vox2ras, scalars=sc, properties=pr)
    f = str(tmp_path / "s.trk")
    assert fj.trk_write(tr, f, ref) is False
    raw = open(f, "rb").read()
    assert len(raw) == 1000 + 4 * (3 * (1 + 3) + 8 * (3 + 2))
    hdr = struct.unpack("<6s3h3f3fh", raw[:38])
    assert hdr[-1] == 2 and struct.unpack("<h", raw[238:240])[0] == 3           # n_scalars @36, n_properties @238
    body = np.frombuffer(raw, np.float32, offset=1000)
    # first line by hand: npts, then (x, y, z, s1, s2) per point, then 3 properties
    assert body[:1].view(np.int32)[0] == 3
    rec = body[1:1 + 15].reshape(3, 5)
    want = ((xyz[:3].astype(np.float64) + 0.5) * np.array([2.0, 1.0, 0.5])).astype(np.float32)
    assert np.array_equal(rec[:, :3], want) and np.array_equal(rec[:, 3:], sc[:3])
    assert np.array_equal(body[16:19], pr[0])
    assert body[19:20].view(np.int32)[0] == 1                                   # second line starts right after
    back = fj.trk_read(f)
    assert np.array_equal(back.npts, npts) and np.array_equal(back.scalars, sc) and np.array_equal(back.properties, pr)
    np.testing.assert_allclose(back.xyz, xyz, atol=1e-6)
    # one scalar per point (LCM runs): 1-D in, 1-D out
    tr1 = fj.Tract(xyz=xyz, npts=npts, volsize=(4, 5, 6), volres=ref.volres, vox2ras=ref.vox2ras, scalars=sc[:, 0].copy())
    assert fj.trk_write(tr1, f, ref) is False
    b1 = fj.trk_read(f)
    assert b1.scalars.shape == (8,) and np.array_equal(b1.scalars, sc[:, 0]) and b1.properties is None


# ---- N2 against an INDEPENDENT implementation (tests/nifti_independent.py: written from the NIfTI-1 standard's offset table, shares
# no code with fibers.jl_amd/nifti.py) and a committed hand-assembled byte fixture -----------------------------------------------------
def test_mri_read_decodes_the_hand_assembled_fixture(fj):
    """tests/golden/nifti/hand_be_int16.nii: big-endian int16, scl_slope 2 / scl_inter -3, qform only (rotated, qfac -1), mm + s;
    b-table as rows.  What mri_read must return is computed here from the definitions, not from the module under test."""
    import nifti_independent as ni
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nifti")
    raw = open(os.path.join(d, "hand_be_int16.nii"), "rb").read()
    assert len(raw) == 352 + int(np.prod(ni.FIX_SHAPE)) * 2 and raw[:4] == b"\x00\x00\x01\x5c"      # 348, big-endian
    m = fj.mri_read(os.path.join(d, "hand_be_int16.nii"))
    want = (ni.fixture_raw_values().astype(np.float64) * ni.FIX_SLOPE + ni.FIX_INTER).astype(np.int16)   # dtype.(vol*slope + inter), mri.jl:1664-1668
    assert m.vol.dtype == np.int16 and m.vol.shape == ni.FIX_SHAPE and np.array_equal(m.vol, want)
    assert m.niftihdr["do_bswap"] and m.nframes == 5
    M = ni.quatern_to_affine(*ni.FIX_QUAT, *ni.FIX_QOFF, *ni.FIX_PIX[:3], -1.0)
    assert np.allclose(m.vox2ras, M, atol=2e-6) and np.linalg.det(M[:3, :3]) < 0
    assert np.allclose(m.volres, ni.FIX_PIX[:3], atol=1e-6)
    assert abs(m.tr - 1750.0) < 1e-3                                       # seconds -> ms (mri.jl:1452-1461)
    assert np.array_equal(m.bval, np.array([0, 1000, 1000, 2000, 3000], np.float32))
    g = np.array([[0, 0, 0], [2, 0, 0], [0, 3, 0], [1, 1, 1], [-3, 0, 4]], np.float64)
    gn = np.where(np.linalg.norm(g, axis=1, keepdims=True) > 0, g / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-30), 0)
    assert np.allclose(m.bvec, gn, atol=1e-7)                              # rows -> [n,3], normalised, 0/0 -> 0 (mri.jl:711-712)


@pytest.mark.parametrize("layout", ["rows", "columns"])
@pytest.mark.parametrize("endian", ["<", ">"])
def test_mri_read_against_independently_assembled_files(fj, tmp_path, layout, endian):
    """float32 / uint8 volumes, sform present (wins over the qform), millimetre / micron units, both byte orders, b-tables in both
    layouts: files assembled by the independent packer"""
    import nifti_independent as ni
    rng = np.random.default_rng(3)
    vol = np.asfortranarray(rng.normal(size=(3, 5, 4, 6)).astype(np.float32))
    srow = np.array([[-1.5, 0.1, 0.0, 80.0], [0.1, 1.5, 0.2, -100.0], [0.0, -0.2, 2.0, -60.0]], np.float32)
    raw = ni.assemble(vol, 16, endian=endian, pixdim=(1.5, 1.5, 2.0, 3.0, 0.0), quatern=(0.0, 0.0, 0.7), qoffset=(1.0, 2.0, 3.0),
                      qform_code=1, sform_code=1, srow=srow, xyzt_units=2 | 16)
    f = str(tmp_path / "a.nii")
    open(f, "wb").write(raw)
    bval = np.array([5, 1000, 2000, 1000, 3000, 5], np.float32)
    bvec = rng.normal(size=(6, 3)).astype(np.float32)
    if layout == "rows":
        np.savetxt(str(tmp_path / "a.bvals"), bval[None], fmt="%g"); np.savetxt(str(tmp_path / "a.bvecs"), bvec.T, fmt="%.7f")
    else:
        np.savetxt(str(tmp_path / "a.bvals"), bval[:, None], fmt="%g"); np.savetxt(str(tmp_path / "a.bvecs"), bvec, fmt="%.7f")
    m = fj.mri_read(f)
    assert np.array_equal(m.vol, vol) and m.vol.dtype == np.float32
    want = np.vstack([srow, [0, 0, 0, 1]])
    assert np.allclose(m.vox2ras, want, atol=1e-6)                          # the sform wins (mri.jl:1530-1545)
    assert np.allclose(m.volres, np.sqrt((srow[:, :3].astype(np.float64) ** 2).sum(axis=0)), atol=1e-6)
    assert abs(m.tr - 3.0) < 1e-6                                          # already ms
    assert np.array_equal(m.bval, bval) and np.allclose(m.bvec, bvec / np.linalg.norm(bvec, axis=1, keepdims=True), atol=2e-6)
    # microns: sizes and offsets scaled to mm
    rawu = ni.assemble(vol[..., 0].astype(np.float32), 16, endian=endian, pixdim=(10.0, 10.0, 40.0, 0.0, 0.0), qform_code=1, xyzt_units=3)
    fu = str(tmp_path / "u.nii")
    open(fu, "wb").write(rawu)
    mu = fj.mri_read(fu)
    assert np.allclose(mu.volres, (0.01, 0.01, 0.04), atol=1e-9) and min(mu.volres) <= 0.05      # the microscopy regime's test (stream.jl:85)


def test_mri_write_output_parsed_by_the_independent_reader(fj, tmp_path):
    import nifti_independent as ni
    rng = np.random.default_rng(4)
    for dt, code in ((np.float32, 16), (np.int16, 4), (np.uint8, 2)):
        vol = np.asfortranarray((rng.normal(size=(6, 5, 4, 3)) * 50).astype(dt))
        m = fj.MRI(vol, volres=(1.5, 1.5, 2.0), vox2ras=_affine())
        m.tr = 8.5
        f = str(tmp_path / ("w_%s.nii" % np.dtype(dt).name))
        assert fj.mri_write(m, f) is False
        raw = open(f, "rb").read()
        h = ni.parse(raw)
        assert h["endian"] == "<" and h["magic"] == b"n+1\0" and len(raw) == h["nbytes_expected"]
        assert h["dim"][:5] == [4, 6, 5, 4, 3] and h["datatype"] == code and h["bitpix"] == np.dtype(dt).itemsize * 8
        assert h["vox_offset"] == 352.0 and np.array_equal(h["data"], vol)
        assert h["sform_code"] == 1 and h["qform_code"] == 1 and h["xyzt_units"] == (2 | 16)
        assert np.allclose([h["srow_x"], h["srow_y"], h["srow_z"]], _affine()[:3], atol=1e-6)
        assert np.allclose(h["pixdim"][1:5], (1.5, 1.5, 2.0, 8.5), atol=1e-6) and h["pixdim"][0] == -1.0
        Mq = ni.quatern_to_affine(*h["quatern"], *h["qoffset"], *h["pixdim"][1:4], h["pixdim"][0])
        assert np.allclose(Mq, _affine(), atol=1e-4)                        # the quaternion the writer derived reproduces the affine


@pytest.mark.gpu
def test_gpu_trk_serialiser_matches_host(fj, tmp_path):
    import torch
    from fibers_jl_amd import phantom, trk
    n = 14
    ov = np.asfortranarray(phantom.fibre_field(n, n, n).astype(np.float32))
    ref = fj.MRI(np.ones((n, n, n), np.uint8), volres=(1.25, 1.5, 2.0))
    o = torch.from_numpy(np.ascontiguousarray(ov.reshape(n ** 3, 3, order="F").T)).cuda()
    field, mout = fj.stream_field_device([o], mask=torch.ones(n ** 3, dtype=torch.uint8, device="cuda"))
    seeds = torch.nonzero(mout).flatten()
    sub = torch.tensor([[0.1, -0.2, 0.3], [0.0, 0.4, -0.4]], dtype=torch.float32, device="cuda")
    res = fj.stream_device(field, (n, n, n), seeds, sub, len_min=4)
    tr = fj.Tract(xyz=res["xyz"].cpu().numpy(), npts=res["npts"].cpu().numpy(), volsize=(n, n, n), volres=ref.volres)
    f1, f2 = str(tmp_path / "host.trk"), str(tmp_path / "gpu.trk")
    fj.trk_write(tr, f1, ref)
    with open(f2, "wb") as fh:                                    # an existing, longer file at the target: the writer must end it at its own length
        fh.write(b"\xff" * (os.path.getsize(f1) + 4096))
    info = fj.stream_to_trk(f2, field, (n, n, n), seeds, sub, ref, len_min=4)
    assert info["nlines"] == tr.nstr and open(f1, "rb").read() == open(f2, "rb").read()


@pytest.mark.gpu
def test_fit_streams_a_memory_mapped_nifti_file(fj, orc, tmp_path):
    """N2 on the device path: mri_read(..., mmap=True) returns the .nii file itself as `vol`; fib_dti_fit / fib_gqi_rec gather
    their chunks from the mapping into the pinned ring (file -> pinned -> HBM).  The fits of the mapped file are compared with the
    ORACLE run on the arrays the file was written from (and, bit for bit, with the fits of an in-memory copy)."""
    from fibers_jl_amd import phantom
    shape = (20, 18, 16)
    bval, bvec = phantom.scheme_dti(30, 3, 1000.0, seed=2)
    dwi, _, _ = phantom.make_volume(shape, bval, bvec, seed=5, nonpositive_frac=0.005)
    src = fj.MRI(dwi, bval, bvec, volres=(1.25, 1.25, 1.25), vox2ras=_affine())
    f = str(tmp_path / "dwi.nii")
    assert fj.mri_write(src, f) is False
    np.savetxt(str(tmp_path / "dwi.bval"), bval[None], fmt="%g")
    np.savetxt(str(tmp_path / "dwi.bvec"), bvec.T, fmt="%.8f")
    m = fj.mri_read(f, mmap=True)
    assert isinstance(m.vol, np.memmap) and m.vol.flags.f_contiguous and m.vol.dtype == np.float32
    assert np.array_equal(np.asarray(m.vol), dwi) and np.array_equal(m.bval, bval)
    mask = fj.MRI(np.ones(shape, np.uint8))
    os.environ["FIBERS_HOST_CHUNK"] = "2048"
    try:
        a = fj.dti_fit(m, mask)
        b = fj.dti_fit(fj.MRI(dwi, m.bval, m.bvec), mask)
        for k in fj.dti.DTI_FIELDS:
            assert np.array_equal(getattr(a, k).vol, getattr(b, k).vol, equal_nan=True), k
        from util import assert_dti_close, peak_mismatches_are_ties
        ones = np.ones(shape, np.uint8)
        ref = orc.dti_fit(dwi, ones, bval, bvec, nthreads=4)                 # the oracle never sees the file
        # (0.5 % non-positive samples: those voxels take the per-voxel pinv branch, dti.jl:297-303 -> its looser tolerances)
        assert_dti_close({k: getattr(a, k).vol for k in fj.dti.DTI_FIELDS}, ref, ones, label="mmap dti",
                         s0_rtol=2e-3, ev_rtol=5e-3, ev_atol=2e-6, fa_atol=5e-3, vec_tol=1e-3, gap=0.2)
        ga, gb = fj.gqi_rec(m, mask), fj.gqi_rec(fj.MRI(dwi, m.bval, m.bvec), mask)
        rg = orc.gqi_rec(dwi, ones, bval, bvec, fj.sphere_642.vertices, fj.sphere_642.faces, 1.25, nthreads=4)
        assert (np.abs(ga.odf.vol - rg["odf"]) / (np.abs(rg["odf"]).max(axis=3, keepdims=True) + 1e-30)).max() < 2e-5
        nv = fj.sphere_642.nvert
        peak_mismatches_are_ties(rg["odf"], rg["peak"], [p.vol for p in ga.peak], np.asarray(fj.sphere_642.vertices, np.float32)[:nv],
                                 faces=np.asarray(fj.sphere_642.faces))
        assert np.array_equal(ga.odf.vol, gb.odf.vol) and all(np.array_equal(ga.qa[k].vol, gb.qa[k].vol, equal_nan=True) for k in range(3))
    finally:
        os.environ.pop("FIBERS_HOST_CHUNK", None)
    # .nii.gz: inflated in-process (no zcat), mmap request falls back to an array
    fz = str(tmp_path / "dwi2.nii.gz")
    assert fj.mri_write(src, fz) is False
    mz = fj.mri_read(fz, mmap=True)
    assert not isinstance(mz.vol, np.memmap) and np.array_equal(mz.vol, dwi)
