"""
oracle/oracle.py — NumPy/C CPU restatement of the Fibers.jl hot path (TEST INFRASTRUCTURE).

PARITY UNPINNED: the reference is Julia (not runnable here) and has no tests or golden
vectors (test/runtests.jl:4-6), so this restatement is checked only against analytic
known answers (tests/test_oracle_*.py).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product (fibers.jl_amd) never does.

The work-struct constructors (DTIwork, GQIwork, DSIwork, StreamWork) are restated here in
float32 NumPy following the Julia sources line by line; the per-voxel loops and the
streamline integrator live in fibers_oracle.c (OpenMP, the reference's z-slice / seed-chunk
decomposition).  Arrays are Fortran-ordered [nx,ny,nz,nframes] like `MRI.vol` (mri.jl:81).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32 = np.float32
_pf = np.ctypeslib.ndpointer(dtype=np.float32, flags="F_CONTIGUOUS")
_pfc = np.ctypeslib.ndpointer(dtype=np.float32)
_pu8 = np.ctypeslib.ndpointer(dtype=np.uint8)
_pi32 = np.ctypeslib.ndpointer(dtype=np.int32)
_pi64 = np.ctypeslib.ndpointer(dtype=np.int64)


def build():
    """Compile the C restatement (gcc); idempotent."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libfibers_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_gqi_rec.restype = C.c_float
        _LIB.orc_dsi_rec.restype = C.c_float
        _LIB.orc_stream.restype = C.c_int64
        _LIB.orc_stream_micro.restype = C.c_int64
        _LIB.orc_stream_full.restype = C.c_int64
        _LIB.orc_stream_flat.restype = C.c_int64
        _LIB.orc_uniform.restype = C.c_float
        _LIB.orc_find_peaks.restype = C.c_int
        _LIB.orc_max_threads.restype = C.c_int
    return _LIB


def max_threads():
    return int(lib().orc_max_threads())


def _fvol(a):
    return np.asfortranarray(a, dtype=np.float32)


def _mask_u8(mask):
    """`mask.vol[ix,iy,iz] == 0 && continue` (dti.jl:261): any non-zero value is in-mask."""
    m = np.asarray(mask)
    if m.ndim == 4:
        m = m[..., 0]
    return np.asfortranarray((m != 0).astype(np.uint8))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------------------
# pinv (LinearAlgebra.pinv on a Float32 matrix: LAPACK SVD, rtol = eps(Float32)*min(m,n))
# ----------------------------------------------------------------------------------------
def pinv32(A):
    A = np.asarray(A, dtype=np.float32)
    if A.size == 0:
        return np.zeros(A.shape[::-1], np.float32)
    U, S, Vt = np.linalg.svd(A, full_matrices=False)         # float32 LAPACK, like Julia
    tol = f32(np.finfo(np.float32).eps * min(A.shape)) * S.max()
    Sinv = np.zeros_like(S)
    keep = S > tol
    Sinv[keep] = f32(1) / S[keep]
    return ((Vt.T * Sinv) @ U.T).astype(np.float32)


# ----------------------------------------------------------------------------------------
# DTI / ADC   (dti.jl:39-84, 101-155, 164-213, 243-335)
# ----------------------------------------------------------------------------------------
def dti_work(bval, bvec):
    bval = np.asarray(bval, np.float32)
    bvec = np.asarray(bvec, np.float32)
    nvol = bval.shape[0]
    A = np.empty((nvol, 7), np.float32)
    A[:, 0] = bvec[:, 0] ** 2                                  # dti.jl:131-136
    A[:, 1] = f32(2) * bvec[:, 0] * bvec[:, 1]
    A[:, 2] = f32(2) * bvec[:, 0] * bvec[:, 2]
    A[:, 3] = bvec[:, 1] ** 2
    A[:, 4] = f32(2) * bvec[:, 1] * bvec[:, 2]
    A[:, 5] = bvec[:, 2] ** 2
    A[:, :6] *= -bval[:, None]                                 # dti.jl:138
    A[:, 6] = 1                                                # dti.jl:140
    return dict(nvol=nvol, A=A, pA=pinv32(A), ib0=(bval == bval.min()))   # dti.jl:117,143


def adc_work(bval):
    bval = np.asarray(bval, np.float32)
    A = np.stack([-bval, np.ones_like(bval)], axis=1).astype(np.float32)   # dti.jl:68-69
    return dict(nvol=bval.shape[0], A=A, pA=pinv32(A), ib0=(bval == bval.min()))


_DTI_FIELDS = ("s0", "eigval1", "eigval2", "eigval3", "eigvec1", "eigvec2", "eigvec3", "rd", "md", "fa")


def dti_from_d(d):
    out = np.zeros(16, np.float32)
    lib().orc_dti_from_d(_p(np.ascontiguousarray(d, np.float32)), _p(out))
    return out


def sym3_eigen(a11, a12, a13, a22, a23, a33):
    w = np.zeros(3, np.float32)
    v = np.zeros((3, 3), np.float32)
    L = lib()
    L.orc_sym3_eigen.argtypes = [C.c_float] * 6 + [C.c_void_p, C.c_void_p]
    L.orc_sym3_eigen(a11, a12, a13, a22, a23, a33, _p(w), _p(v))
    return w, v.T.copy()      # columns = eigenvectors, ascending eigenvalues


def st_eigen(sxx, sxy, sxz, syy, syz, szz):
    """st_eigen (structens.jl:13-37) -> (eigvec [nx,ny,nz,3,3], eigval [nx,ny,nz,3]), Fortran order."""
    vols = [np.asfortranarray(v, dtype=np.float32) for v in (sxx, sxy, sxz, syy, syz, szz)]
    shape = vols[0].shape
    nvox = int(np.prod(shape))
    eigvec = np.zeros(shape + (3, 3), np.float32, order="F")
    eigval = np.zeros(shape + (3,), np.float32, order="F")
    L = lib()
    L.orc_st_eigen.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_void_p, C.c_void_p]
    L.orc_st_eigen.restype = None
    L.orc_st_eigen(*[_p(v) for v in vols], nvox, _p(eigvec), _p(eigval))
    return eigvec, eigval


def dti_fit(dwi, mask, bval, bvec, nthreads=1):
    """dti_fit(dwi::MRI, mask::MRI) (dti.jl:221) -> dict of the 10 DTI volumes."""
    if bval is None or len(bval) == 0:
        raise ValueError("Missing b-value table from input DWI structure")       # dti.jl:224
    if bvec is None or len(bvec) == 0:
        raise ValueError("Missing gradient table from input DWI structure")      # dti.jl:228
    dwi = _fvol(dwi)
    nx, ny, nz, nvol = dwi.shape
    W = dti_work(bval, bvec)
    m = _mask_u8(mask)
    nvox = nx * ny * nz
    out = {k: np.zeros((nx, ny, nz, 3) if "vec" in k else (nx, ny, nz), np.float32, order="F") for k in _DTI_FIELDS}
    partial = np.zeros(nvox, np.int64)
    npart = C.c_int64(0)
    pA = np.asfortranarray(W["pA"])
    ib0 = W["ib0"].astype(np.uint8)
    lib().orc_dti_fit(_p(dwi), _p(m), nx, ny, nz, nvol, _p(pA), _p(ib0),
                      *[_p(out[k]) for k in _DTI_FIELDS], _p(partial), C.byref(npart), int(nthreads))
    # per-voxel pinv branch (dti.jl:297-298)
    flat = {k: out[k].reshape((nvox, -1) if "vec" in k else (nvox,), order="F") for k in _DTI_FIELDS}
    dflat = dwi.reshape((nvox, nvol), order="F")
    for vox in np.sort(partial[: npart.value]):
        s = dflat[vox]
        ipos = s > 0
        d = pinv32(W["A"][ipos]) @ np.log(s[ipos]).astype(np.float32)
        o = dti_from_d(d.astype(np.float32))
        flat["s0"][vox], flat["eigval1"][vox], flat["eigval2"][vox], flat["eigval3"][vox] = o[0:4]
        flat["eigvec1"][vox], flat["eigvec2"][vox], flat["eigvec3"][vox] = o[4:7], o[7:10], o[10:13]
        flat["rd"][vox], flat["md"][vox], flat["fa"][vox] = o[13:16]
    out["_npartial"] = int(npart.value)
    return out


def adc_fit(dwi, mask, bval, nthreads=1):
    """adc_fit(dwi::MRI, mask::MRI) (dti.jl:164) -> (adc, s0)."""
    if bval is None or len(bval) == 0:
        raise ValueError("Missing b-value table from input DWI structure")
    dwi = _fvol(dwi)
    nx, ny, nz, nvol = dwi.shape
    W = adc_work(bval)
    m = _mask_u8(mask)
    nvox = nx * ny * nz
    adc = np.zeros((nx, ny, nz), np.float32, order="F")
    s0 = np.zeros((nx, ny, nz), np.float32, order="F")
    partial = np.zeros(nvox, np.int64)
    npart = C.c_int64(0)
    lib().orc_adc_fit(_p(dwi), _p(m), nx, ny, nz, nvol, _p(np.asfortranarray(W["pA"])),
                      _p(W["ib0"].astype(np.uint8)), _p(adc), _p(s0), _p(partial), C.byref(npart), int(nthreads))
    dflat = dwi.reshape((nvox, nvol), order="F")
    a_flat, s_flat = adc.reshape(nvox, order="F"), s0.reshape(nvox, order="F")
    for vox in partial[: npart.value]:
        s = dflat[vox]
        ipos = s > 0
        d = pinv32(W["A"][ipos]) @ np.log(s[ipos]).astype(np.float32)
        a_flat[vox] = d[0]
        s_flat[vox] = np.exp(f32(d[1]))
    return adc, s0


# ----------------------------------------------------------------------------------------
# GQI   (gqi.jl:32-82, 109-201)
# ----------------------------------------------------------------------------------------
def fold_faces(faces, nvert):
    """faces[faces .> nvert] .-= nvert (gqi.jl:63-64); returns 0-based int32 [nfaces,3] F-order."""
    f = np.array(faces, dtype=np.int64, copy=True)
    f[f > nvert] -= nvert
    return np.asfortranarray((f - 1).astype(np.int32))


def _sinc32(x):
    """Base.sinc on Float32: sinpi(x)/(pi*x), sinc(0)=1; evaluated in float64, rounded once."""
    xd = x.astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        y = np.where(xd == 0, 1.0, np.sin(np.pi * xd) / (np.pi * xd))
    return y.astype(np.float32)


def gqi_work(bval, bvec, vertices, faces, sigma=1.25):
    bval = np.asarray(bval, np.float32)
    bvec = np.asarray(bvec, np.float32)
    V = np.asarray(vertices, np.float32)
    nvert = V.shape[0] // 2                                     # gqi.jl:48
    scale = np.sqrt(bval * f32(0.01506)) * (f32(sigma) / f32(np.pi))       # gqi.jl:68
    bq = bvec * scale[:, None]
    X = (V[nvert:, :] @ bq.T).astype(np.float32)                # gqi.jl:69
    return dict(nvol=bval.shape[0], nvert=nvert, A=np.asfortranarray(_sinc32(X)),
                faces=fold_faces(faces, nvert), vertices=np.asfortranarray(V))


def find_peaks(o, faces0):
    """find_peaks!(W) for one ODF (gqi.jl:180-201) -> (isort 0-based, nvalid, odf_peak)."""
    o = np.ascontiguousarray(o, np.float32)
    n = o.shape[0]
    pk = np.zeros(n, np.float32)
    isort = np.zeros(n, np.int32)
    tmp = np.zeros(n, np.int32)
    f = np.asfortranarray(faces0, dtype=np.int32)
    nv = lib().orc_find_peaks(_p(o), n, _p(f), f.shape[0], _p(pk), _p(isort), _p(tmp))
    return isort, int(nv), pk


def gqi_rec(dwi, mask, bval, bvec, vertices, faces, sigma=1.25, nthreads=1):
    """gqi_rec (gqi.jl:109) -> dict(odf, peak[3], qa[3], odfmax)."""
    if bval is None or len(bval) == 0:
        raise ValueError("Missing b-value table from input DWI structure")
    if bvec is None or len(bvec) == 0:
        raise ValueError("Missing gradient table from input DWI structure")
    dwi = _fvol(dwi)
    nx, ny, nz, nvol = dwi.shape
    W = gqi_work(bval, bvec, vertices, faces, sigma)
    m = _mask_u8(mask)
    odf = np.zeros((nx, ny, nz, W["nvert"]), np.float32, order="F")
    peak = [np.zeros((nx, ny, nz, 3), np.float32, order="F") for _ in range(3)]
    qa = [np.zeros((nx, ny, nz), np.float32, order="F") for _ in range(3)]
    with np.errstate(all="ignore"):
        odfmax = lib().orc_gqi_rec(_p(dwi), _p(m), nx, ny, nz, nvol, _p(W["A"]), W["nvert"],
                                   _p(W["faces"]), W["faces"].shape[0], _p(W["vertices"]), W["vertices"].shape[0],
                                   _p(odf), *[_p(x) for x in peak], *[_p(x) for x in qa], int(nthreads))
    return dict(odf=odf, peak=peak, qa=qa, odfmax=float(odfmax))


# ----------------------------------------------------------------------------------------
# DSI   (dsi.jl:41-143, 171-270)
# ----------------------------------------------------------------------------------------
def dsi_work(bval, bvec, vertices, faces, hann_width=32):
    bval = np.asarray(bval, np.float32)
    bvec = np.asarray(bvec, np.float32)
    V = np.asarray(vertices, np.float32)
    q = bvec * np.sqrt(bval)[:, None]                           # dsi.jl:62
    bmin = bval.min()
    dq = np.sqrt(bval[bval > bmin].min())                       # dsi.jl:66
    iq = np.rint(q / dq).astype(np.int32)                       # dsi.jl:67 (ties to even)
    nfft = int(iq.max() - iq.min() + 1)
    nfft = 2 ** int(np.ceil(np.log2(nfft)))                     # dsi.jl:70-71
    if nfft != 16:
        raise NotImplementedError("oracle supports the 16^3 q-grid only (nfft=%d)" % nfft)
    shift = nfft // 2 + 1                                       # dsi.jl:73 (1-based centre)
    sub = iq + shift                                            # 1-based subscripts
    iq_ind = (sub[:, 0] - 1) + nfft * (sub[:, 1] - 1) + nfft * nfft * (sub[:, 2] - 1)   # 0-based linear
    H = np.zeros(nfft ** 3, np.float32)
    if hann_width == 0:
        H[:] = 1
    else:                                                       # dsi.jl:83-84 (computed in Float64)
        r = np.sqrt((iq.astype(np.int64) ** 2).sum(axis=1).astype(np.float64))
        H[iq_ind] = ((1.0 + np.cos(r * (2 * np.pi / hann_width))) * 0.5).astype(np.float32)
    nvert = V.shape[0] // 2
    qr = f32(nfft / 2 - 1) * np.linspace(0.3, 0.9, 21).astype(np.float32)       # dsi.jl:104
    dqr = f32(qr[1] - qr[0])
    # iq_sub_interp[3, nrad, nvert] = v * qr' .+ iq_shift   (dsi.jl:106-109)
    interp = (V[nvert:, :].T[:, None, :] * qr[None, :, None] + f32(shift)).astype(np.float32)
    return dict(nvol=bval.shape[0], nfft=nfft, nvert=nvert, dqr=dqr, iq=iq, iq_ind=iq_ind.astype(np.int32),
                H=H, qr=qr, qr2=(qr ** 2).astype(np.float32), interp=np.asfortranarray(interp),
                faces=fold_faces(faces, nvert), vertices=np.asfortranarray(V), shift=shift)


def dsi_rec(dwi, mask, bval, bvec, vertices, faces, hann_width=32, nthreads=1):
    """dsi_rec (dsi.jl:171) -> dict(pdf, odf, peak[3], qa[3], odfmax)."""
    if bval is None or len(bval) == 0:
        raise ValueError("Missing b-value table from input DWI structure")
    if bvec is None or len(bvec) == 0:
        raise ValueError("Missing gradient table from input DWI structure")
    dwi = _fvol(dwi)
    nx, ny, nz, nvol = dwi.shape
    W = dsi_work(bval, bvec, vertices, faces, hann_width)
    m = _mask_u8(mask)
    pdf = np.zeros((nx, ny, nz, nvol), np.float32, order="F")
    odf = np.zeros((nx, ny, nz, W["nvert"]), np.float32, order="F")
    peak = [np.zeros((nx, ny, nz, 3), np.float32, order="F") for _ in range(3)]
    qa = [np.zeros((nx, ny, nz), np.float32, order="F") for _ in range(3)]
    L = lib()
    L.orc_dsi_rec.argtypes = None
    odfmax = L.orc_dsi_rec(_p(dwi), _p(m), nx, ny, nz, nvol, _p(W["iq_ind"]), _p(W["H"]), _p(W["interp"]),
                           W["qr"].shape[0], _p(W["qr2"]), C.c_float(float(W["dqr"])),
                           W["nvert"], _p(W["faces"]), W["faces"].shape[0], _p(W["vertices"]), W["vertices"].shape[0],
                           _p(pdf), _p(odf), *[_p(x) for x in peak], *[_p(x) for x in qa], int(nthreads))
    return dict(pdf=pdf, odf=odf, peak=peak, qa=qa, odfmax=float(odfmax))


# ----------------------------------------------------------------------------------------
# Streamlines   (stream.jl:74-193 non-LCM/non-micro, 340-374, 501-541, 625-690, 730-790)
# ----------------------------------------------------------------------------------------
def cosd32(deg):
    """cosd(Float32(deg)): exact-degree cosine, rounded to Float32."""
    deg = float(f32(deg))
    exact = {0.0: 1.0, 60.0: 0.5, 90.0: 0.0, 120.0: -0.5, 180.0: -1.0}
    if deg in exact:
        return f32(exact[deg])
    return f32(np.cos(np.deg2rad(np.float64(deg))))


def sind_cosd32(x):
    """sind.(x), cosd.(x) for Float32 angles in [-90, 90] (stream.jl:166-169).  Base's sind / cosd reduce the argument by
    quadrants in degrees -- exact at multiples of 90 -- and evaluate the kernels on an extended-precision deg2rad: restated as
    the Float64 function rounded once to Float32, with the exact values forced."""
    xd = np.asarray(x, np.float32).astype(np.float64)
    s_, c_ = np.sin(np.deg2rad(xd)), np.cos(np.deg2rad(xd))
    c_ = np.where(np.abs(xd) == 90.0, 0.0, c_)
    s_ = np.where(xd == 90.0, 1.0, np.where(xd == -90.0, -1.0, s_))
    return s_.astype(np.float32), c_.astype(np.float32)


def angles_to_vectors(vol, volres):
    """2-D orientation angles -> 3-D vectors, as StreamWork expands them (stream.jl:147-172): the through-plane dimension is the
    one with the largest voxel size (argmax: the first maximum), the angle lives in the other two; radians if every value is
    within [-pi/2 - eps, pi/2 + eps] (tested first), degrees if within [-90, 90], an error otherwise.  Returns
    (vectors [nx,ny,nz,3] float32, thrudim 0-based)."""
    a = np.asarray(vol, np.float32)
    if a.ndim == 4:
        a = a[..., 0]
    thru = int(np.argmax(np.asarray(volres, np.float32)))       # :149
    sd = [c for c in range(3) if c != thru]                     # :151
    eps32 = float(np.finfo(np.float32).eps)
    lo, hi = float(a.min()), float(a.max())
    out = np.zeros(a.shape + (3,), np.float32)
    if -np.pi / 2 - eps32 <= lo and hi <= np.pi / 2 + eps32:    # :157-158 in radians
        out[..., sd[0]] = np.cos(a)                             # Float32 cos / sin
        out[..., sd[1]] = np.sin(a)
    elif -90 <= lo and hi <= 90:                                # :163-164 in degrees
        s_, c_ = sind_cosd32(a)
        out[..., sd[0]] = c_
        out[..., sd[1]] = s_
    else:
        raise ValueError("Input orientations should be 3D vectors or angles in [-90, 90]")   # :170
    return out, thru


def stream_work(ovec, f=None, f_thresh=0.03, fa=None, fa_thresh=0.1, mask=None, volres=(1.0, 1.0, 1.0)):
    """StreamWork mask / vector repack (stream.jl:76-172). ovec: list of [nx,ny,nz,3] vectors or [nx,ny,nz(,1)] 2-D angles."""
    ovecs = [ovec] if isinstance(ovec, np.ndarray) else list(ovec)
    raw = ovecs
    ovecs = []
    thrus = []
    for v in raw:
        v = np.asarray(v)
        if v.ndim == 4 and v.shape[3] == 3:
            ovecs.append(v)
        else:
            if mask is None:
                raise ValueError("angle inputs need a mask here (stream.jl:96-100 would derive it from the angles)")
            e, t = angles_to_vectors(v, volres)
            ovecs.append(e)
            thrus.append(t)
    stream_work.last_thrudims = thrus
    fs = None if f is None else ([f] if isinstance(f, np.ndarray) else list(f))
    nvec = len(ovecs)
    nx, ny, nz = ovecs[0].shape[:3]
    if mask is None:                                            # stream.jl:95-100
        mk = np.zeros((nx, ny, nz), bool)
        for v in ovecs:
            mk |= np.any(np.asarray(v) != 0, axis=3)
    else:                                                       # stream.jl:102
        m = np.asarray(mask)
        mk = (m[..., 0] if m.ndim == 4 else m) > 0
    if fa is not None:                                          # stream.jl:116
        a = np.asarray(fa, np.float32)
        mk = mk & ((a[..., 0] if a.ndim == 4 else a) >= f32(fa_thresh))
    arr = np.zeros((3, nvec, nx, ny, nz), np.float32, order="F")
    for k, v in enumerate(ovecs):
        om = mk
        if fs is not None:                                      # stream.jl:138
            fk = np.asarray(fs[k], np.float32)
            om = mk & ((fk[..., 0] if fk.ndim == 4 else fk) >= f32(f_thresh))
        v = np.asarray(v, np.float32)
        for c in range(3):                                      # stream.jl:141-145 (Bool is a strong zero)
            arr[c, k] = np.where(om, v[..., c], f32(0))
    return np.asfortranarray(mk), arr


def seeds_from_mask(maskbool):
    """findall(mask) in column-major order -> [n,3] int32 1-based (stream.jl:744,754)."""
    lin = np.flatnonzero(np.asarray(maskbool).ravel(order="F"))
    nx, ny, nz = maskbool.shape
    return np.stack([lin % nx + 1, (lin // nx) % ny + 1, lin // (nx * ny) + 1], axis=1).astype(np.int32)


def stream(ovec, sublist, f=None, f_thresh=0.03, fa=None, fa_thresh=0.1, mask=None, seed=None,
           len_min=3, len_max=None, ang_thresh=45, step_size=0.5, smooth_coeff=0.2, nthreads=1,
           return_all_npts=False, search_dist=0, search_ang=10, lcms=None, lcm_thresh=0.099, rng_seed=0, volres=(1.0, 1.0, 1.0)):
    """stream (stream.jl:730) with an explicit `sublist` [nsub,3] instead of the global RNG
    (stream.jl:176-181).  Returns list of [npts,3] float32 arrays (1-based voxel coords) in
    reference order, plus seed_index (seed*nsub+sub) per kept line.
    search_dist > 0: the microscopy regime (stream.jl:83: minimum(volres) <= 0.05; 252-287, 547-619).
    lcms [nx,ny,nz,10]: LCM-guided tracking (stream.jl:200-236, 380-495); the uniforms behind `rand(Categorical(lcm))`
    come from the counter-based stream orc_uniform(rng_seed, line, k) (the random-number contract of this back end);
    the result then carries `flags` (one per point: the LCM and the angle pick disagreed, stream.jl:538).
    ovec volumes with one frame are 2-D orientation angles (stream.jl:147-172, `volres` picks the through-plane dimension); in
    the microscopy regime their through-plane search distance is 0 (stream.jl:153-155)."""
    mk, arr = stream_work(ovec, f, f_thresh, fa, fa_thresh, mask, volres)
    thrus = stream_work.last_thrudims
    search_flat = thrus[-1] if (thrus and search_dist > 0) else -1          # (every angle volume sets it; the last one stands)
    nx, ny, nz = mk.shape
    if len_max is None:
        len_max = max(nx, ny, nz)                               # stream.jl:74 default
    if seed is None:
        seeds = seeds_from_mask(mk)
    else:
        sd = np.asarray(seed)
        if mask is not None and sd.shape != np.asarray(mask).shape:
            raise ValueError("Dimension mismatch between seed mask %s and brain mask %s"
                             % (sd.shape, np.asarray(mask).shape))   # stream.jl:746-749
        seeds = seeds_from_mask((sd[..., 0] if sd.ndim == 4 else sd) > 0)
    sub = np.ascontiguousarray(sublist, np.float32).reshape(-1, 3)
    seeds = np.ascontiguousarray(seeds)
    p_npts, p_seed, p_xyz = C.POINTER(C.c_int32)(), C.POINTER(C.c_int64)(), C.POINTER(C.c_float)()
    total = C.c_int64(0)
    all_npts = np.zeros(max(1, seeds.shape[0] * sub.shape[0]), np.int32)
    L = lib()
    p_flags = C.POINTER(C.c_uint8)()
    lcm_arr, sd0, sd1 = None, 0, 1
    if lcms is not None:
        lv = np.asarray(lcms, np.float32)
        lcm_arr = np.asfortranarray(np.transpose(lv, (3, 0, 1, 2)))            # permutedims(lcms.vol, (4,1,2,3)), stream.jl:207
        lcm_arr = np.asfortranarray(np.where(lcm_arr >= f32(lcm_thresh), lcm_arr, f32(0)))   # .*= (. >= lcm_thresh), :217
        ov1 = np.asarray(ovec if isinstance(ovec, np.ndarray) else ovec[0])
        if ov1.ndim == 3:
            ov1 = ov1[..., None]
        # stream.jl:221 looks at the FRAMES of the first input volume: three for vectors, ONE for angle inputs (then the
        # through-plane dimension is "1" if every angle is 0, none otherwise, and the first two of what is left are in-plane)
        thru = [c for c in range(ov1.shape[3]) if np.all(ov1[..., c] == 0)]
        strd = [c for c in range(3) if c not in thru]                            # :223
        sd0, sd1 = strd[0], strd[1]
    nl = L.orc_stream_flat(_p(arr), _p(mk.astype(np.uint8, order="F")), nx, ny, nz, arr.shape[1],
                      _p(seeds), C.c_int64(seeds.shape[0]), _p(sub), sub.shape[0],
                      int(len_min), int(len_max), C.c_float(float(cosd32(ang_thresh))),
                      C.c_float(float(f32(step_size))), C.c_float(float(f32(smooth_coeff))),
                      int(search_dist), int(search_flat), C.c_float(float(cosd32(search_ang))),
                      None if lcm_arr is None else _p(lcm_arr), int(sd0), int(sd1), C.c_uint64(int(rng_seed)),
                      C.byref(p_npts), C.byref(p_seed), C.byref(p_xyz), C.byref(p_flags), C.byref(total),
                      _p(all_npts), int(nthreads))
    npts = np.ctypeslib.as_array(p_npts, shape=(max(nl, 1),))[:nl].copy()
    sidx = np.ctypeslib.as_array(p_seed, shape=(max(nl, 1),))[:nl].copy()
    xyz = np.ctypeslib.as_array(p_xyz, shape=(max(total.value, 1) * 3,))[: total.value * 3].copy().reshape(-1, 3)
    flags = None
    if lcm_arr is not None:
        flags = np.ctypeslib.as_array(p_flags, shape=(max(total.value, 1),))[: total.value].copy()
        L.orc_free(C.cast(p_flags, C.c_void_p))
    for ptr in (p_npts, p_seed, p_xyz):
        L.orc_free(C.cast(ptr, C.c_void_p))
    res = dict(npts=npts, seed_index=sidx, xyz=xyz, nseeds=seeds.shape[0], nsub=sub.shape[0], seeds=seeds)
    if flags is not None:
        res["flags"] = flags
    if return_all_npts:
        res["all_npts"] = all_npts[: seeds.shape[0] * sub.shape[0]]
    return res


def split_lines(res):
    off = np.concatenate([[0], np.cumsum(res["npts"])])
    return [res["xyz"][off[i]:off[i + 1]] for i in range(len(res["npts"]))]


# ---------------------------------------------------------------------------------------------
# RUMBA-SD (rusd.jl) -- row N4.  NumPy float32 restatement; every step cites the reference line.
# ---------------------------------------------------------------------------------------------
def _ang2rot(phi, theta):
    """util.jl:85-100"""
    c, s_, ct, st = np.cos(phi), np.sin(phi), np.cos(theta), np.sin(theta)
    Rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float64)
    Ry = np.array([[ct, 0, st], [0, 1, 0], [-st, 0, ct]], np.float64)
    return Rz @ Ry


def rumba_kernel(bval, bvec, vertices, lam_para=1.7e-3, lam_perp=0.2e-3, lam_csf=3.0e-3, lam_gm=0.8e-4):
    """reconstruction kernel of the multi-tensor model (rusd.jl:141-153, 466-469, 495-521): [ndir, nvert+2] float32,
    rows = (mean low-b, then the DWIs in acquisition order), columns = half-sphere vertices, CSF, GM.
    Built in float64 and rounded once (the reference computes it in Float32: differences at the 1e-7 level)."""
    bval = np.asarray(bval, np.float64)
    bvec = np.asarray(bvec, np.float64).reshape(-1, 3)
    ib0 = bval == bval.min()                                                # rusd.jl:449
    gd = bvec[~ib0]
    g = np.vstack([np.zeros((1, 3)), gd / np.sqrt((gd ** 2).sum(1, keepdims=True))])   # :466-467
    b = np.concatenate([[0.0], bval[~ib0]])                                 # :468
    V = np.asarray(vertices, np.float64)
    nvert = V.shape[0] // 2
    x, y, z = V[nvert:, 0], V[nvert:, 1], V[nvert:, 2]                      # second half (:502-504)
    hxy = np.hypot(x, y)
    theta = -np.arctan2(z, hxy)                                             # cart2sph, then θ .= -θ (:505)
    phi = np.arctan2(y, x)
    K = np.empty((len(b), nvert + 2), np.float64)
    for i in range(nvert):
        R = _ang2rot(phi[i], theta[i])
        D = R @ np.diag([lam_para, lam_perp, lam_perp]) @ R.T
        K[:, i] = np.exp(-b * np.einsum("ij,jk,ik->i", g, D, g))            # tensor_model, :150
    K[:, nvert] = np.exp(-b * lam_csf * (g ** 2).sum(1))                    # isotropic CSF (:515)
    K[:, nvert + 1] = np.exp(-b * lam_gm * (g ** 2).sum(1))                 # isotropic GM (:518)
    return K.astype(np.float32), ib0


def rumba_signal(dwi, mask, ib0):
    """signal matrix [ndir, nmask] (rusd.jl:444-464)"""
    dwi = np.asarray(dwi, np.float32)
    nxyz = int(np.prod(dwi.shape[:3]))
    ind = np.flatnonzero(np.asarray(mask).reshape(-1, order="F") > 0)       # :446
    vol = np.maximum(dwi, f32(0)).reshape(nxyz, -1, order="F")
    s0 = vol[:, ib0].mean(axis=1, dtype=np.float32)[ind]                    # :456-457 (mean in Float32)
    sig = np.empty((int((~ib0).sum()) + 1, ind.size), np.float32)
    sig[0] = s0
    with np.errstate(all="ignore"):
        sig[1:] = (vol[:, ~ib0][ind].T / s0[None, :]).astype(np.float32)    # :458-461
    sig[np.isnan(sig)] = 0                                                  # :462
    sig[0] = (sig[0] > 0).astype(np.float32)                                # :463
    sig[sig > 1] = 1                                                        # :464
    return sig, ind


def _besseli_ratio(nu, z):
    """Perron's continued fraction (rusd.jl:170-177), Float32"""
    nu = f32(nu)
    two = f32(2)
    with np.errstate(all="ignore"):
        return z / ((two * nu + z) - ((two * nu + 1) * z / (two * z + (two * nu + 1) -
                    ((two * nu + 3) * z / ((two * nu + 2) + two * z - ((two * nu + 5) * z / ((two * nu + 3) + two * z)))))))


def _rumba_tv(vol, lam):
    """rumba_tv! (rusd.jl:216-235) with sd_grad!/sd_div! (:183-207); vol, lam [nx,ny,nz] float32"""
    eps = np.finfo(np.float32).eps
    gx = np.concatenate([vol[1:], vol[-1:]], 0) - vol
    gy = np.concatenate([vol[:, 1:], vol[:, -1:]], 1) - vol
    gz = np.concatenate([vol[:, :, 1:], vol[:, :, -1:]], 2) - vol
    nrm = np.sqrt(gx ** 2 + gy ** 2 + gz ** 2 + eps)
    gx, gy, gz = gx / nrm, gy / nrm, gz / nrm
    div = np.zeros_like(vol)
    div[1:-1] = gx[1:-1] - gx[:-2]; div[0] = gx[0]; div[-1] = -gx[-2]
    t = np.zeros_like(vol); t[:, 1:-1] = gy[:, 1:-1] - gy[:, :-2]; t[:, 0] = gy[:, 0]; t[:, -1] = -gy[:, -2]; div += t
    t = np.zeros_like(vol); t[:, :, 1:-1] = gz[:, :, 1:-1] - gz[:, :, :-2]; t[:, :, 0] = gz[:, :, 0]; t[:, :, -1] = -gz[:, :, -2]; div += t
    return (f32(1) / (np.abs(f32(1) - lam * div) + eps)).astype(np.float32)


def rumba_neighbours(vertices):
    """idx_neig (rusd.jl:475-493): half-sphere vertices within ang_neig of each vertex (antipodally folded)"""
    V = np.asarray(vertices, np.float32)
    nvert = V.shape[0] // 2
    ang_neig = {362: 12.5, 321: 12.5, 181: 16.0}[nvert]
    H = V[:nvert]
    c = np.clip(H @ H.T, -1, 1)
    ang = np.degrees(np.arccos(c.astype(np.float64)))
    ang = np.minimum(ang, 180 - ang)
    isn = ang < ang_neig
    np.fill_diagonal(isn, False)
    return [np.flatnonzero(isn[i]) for i in range(nvert)]


def rumba_rec(dwi, mask, bval, bvec, vertices, niter=600, lam_para=1.7e-3, lam_perp=0.2e-3, lam_csf=3.0e-3, lam_gm=0.8e-4,
              ncoils=1, coil_combine="SMF-SENSE", ipat_factor=1, use_tv=True):
    """rumba_rec (rusd.jl:419-636).  Returns dict(fodf [nx,ny,nz,nvert], fgm, fcsf, peak[5] [nx,ny,nz,3], gfa, var,
    snr_mean, snr_std)."""
    n_order = 1
    if coil_combine == "SoS-GRAPPA":
        n_order = ncoils
    elif coil_combine != "SMF-SENSE":
        raise ValueError("Unknown coil combine mode " + coil_combine)
    if ipat_factor < 1:
        raise ValueError("iPAT factor must be a positive integer")
    dwi = np.asarray(dwi, np.float32)
    nx, ny, nz = dwi.shape[:3]
    nxyz = nx * ny * nz
    K, ib0 = rumba_kernel(bval, bvec, vertices, lam_para, lam_perp, lam_csf, lam_gm)
    sig, ind = rumba_signal(dwi, mask, ib0)
    ndir, ncomp = K.shape
    nvert = ncomp - 2
    nmask = ind.size
    eps = np.finfo(np.float32).eps
    fodf0 = np.ones(ncomp, np.float32) / f32(2 * nvert + 2)                 # :529-531
    fodf0 = fodf0 / fodf0.sum(dtype=np.float32)
    fodf = np.tile(fodf0[:, None], (1, nmask)).astype(np.float32)           # rumba_sd_initialize!, :241-259
    dodf = np.tile((K @ fodf0)[:, None], (1, nmask)).astype(np.float32)
    lam0 = f32(1 / 15) ** 2
    lam = np.full((nx, ny, nz), lam0, np.float32, order="F")
    s2 = np.full(nmask, lam0, np.float32)
    dsig = (sig * dodf) / s2[None, :]
    tv = np.ones((ncomp, nmask), np.float32)
    snr = np.zeros(nmask, np.float32)
    for _ in range(niter):                                                   # rumba_sd_iterate!, :266-345
        ir = _besseli_ratio(n_order, dsig)
        rl = K.T @ (sig * ir)
        rl2 = K.T @ dodf + eps
        rl = rl / rl2
        if use_tv:
            for ic in range(ncomp):
                vol = np.zeros(nxyz, np.float32)
                vol[ind] = fodf[ic]
                tv[ic] = _rumba_tv(vol.reshape(nx, ny, nz, order="F"), lam).reshape(-1, order="F")[ind]
        fodf = np.maximum(fodf * rl * tv, f32(0))
        dodf = (K @ fodf).astype(np.float32)
        dsig = (sig * dodf) / s2[None, :]
        ir = (sig ** 2 + dodf ** 2) / f32(2) - (s2[None, :] * dsig) * ir
        s2 = ir.sum(axis=0, dtype=np.float32) / f32(n_order * ndir)
        s2 = np.clip(s2, f32((1 / 80) ** 2), f32((1 / 8) ** 2))
        snr = f32(1) / np.sqrt(s2)
        if use_tv:
            if ipat_factor == 1:
                lam[...] = max(s2.mean(dtype=np.float32), f32((1 / 30) ** 2))
            else:
                lam[...] = 0
                lamv = lam.reshape(-1, order="F")                           # a view: lam is Fortran-contiguous
                assert np.shares_memory(lamv, lam)
                lamv[ind] = s2
    snr_mean = float(snr.mean(dtype=np.float32)) if niter > 0 else 0.0
    snr_std = float(np.sqrt(((snr - f32(snr_mean)) ** 2).sum(dtype=np.float32) / f32(max(nmask - 1, 1)))) if niter > 0 else 0.0
    fodf = fodf / (fodf.sum(axis=0, dtype=np.float32) + eps)                # :553
    out_fodf = np.zeros((nxyz, nvert), np.float32)
    out_fodf[ind] = fodf[:nvert].T
    fcsf = np.zeros(nxyz, np.float32); fcsf[ind] = fodf[nvert]
    fgm = np.zeros(nxyz, np.float32); fgm[ind] = fodf[nvert + 1]
    var = np.zeros(nxyz, np.float32); var[ind] = s2
    fiso = fgm + fcsf
    out_fodf = out_fodf + fiso[:, None]                                     # :578
    with np.errstate(all="ignore"):
        out_fodf = out_fodf / out_fodf.sum(axis=1, dtype=np.float32, keepdims=True)
    out_fodf[np.isnan(out_fodf)] = 0
    with np.errstate(all="ignore"):
        m = out_fodf.mean(axis=1, dtype=np.float32, keepdims=True)
        sd = np.sqrt(((out_fodf - m) ** 2).sum(axis=1, dtype=np.float32) / f32(nvert - 1))
        gfa = sd / np.sqrt((out_fodf ** 2).mean(axis=1, dtype=np.float32))  # :589
    gfa[np.isnan(gfa)] = 0
    neig = rumba_neighbours(vertices)
    H = np.asarray(vertices, np.float32)[:nvert]
    peaks = np.zeros((5, nxyz, 3), np.float32)
    mflat = np.asarray(mask).reshape(-1, order="F")
    for v in np.flatnonzero(mflat != 0):                                     # :605-631
        o = out_fodf[v]
        with np.errstate(all="ignore"):
            thr_abs = (f32(0.1) / (f32(1) - fiso[v])) * o.max()
        pk = o.copy()
        for iv in range(nvert):
            if o[iv] < thr_abs or o[iv] <= o[neig[iv]].max():
                pk[iv] = 0
        isort = np.argsort(-pk, kind="stable")
        nvalid = int((pk > 0).sum())
        n = min(nvalid, 5)
        with np.errstate(all="ignore"):
            fnorm = (f32(1) - fiso[v]) / o[isort[:n]].sum(dtype=np.float32)
        for k in range(n):
            peaks[k, v] = H[isort[k]] * (o[isort[k]] * fnorm)
    shp = (nx, ny, nz)
    return dict(fodf=out_fodf.reshape(shp + (nvert,), order="F"), fgm=fgm.reshape(shp, order="F"),
                fcsf=fcsf.reshape(shp, order="F"), gfa=gfa.reshape(shp, order="F"), var=var.reshape(shp, order="F"),
                peak=[peaks[k].reshape(shp + (3,), order="F") for k in range(5)], snr_mean=snr_mean, snr_std=snr_std,
                kernel=K)
