"""stream host mirror (reference: stream.jl:7-8 `StreamWork, stream, stream_new_line, ...`): deterministic
nearest-voxel / fixed-step Euler tractography: the angle-picking macro-scale path, the microscopy regime
(cone search) and LCM-guided tracking (stream.jl:380-495), on 3-D vector or 2-D angle inputs (stream.jl:147-172)."""
import ctypes as C
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib
from .dti import _chk_dev, _mask_arg, _stream_ptr, _sync
from .mri import MRI
from .tract import Tract

_EPS = float(np.finfo(np.float64).eps)


def cosd32(deg) -> np.float32:
    """cosd(Float32(ang)) (stream.jl:193): exact at multiples of 30/45 degrees like Julia's cosd."""
    d = float(np.float32(deg)) % 360.0
    table = {0.0: 1.0, 60.0: 0.5, 90.0: 0.0, 120.0: -0.5, 180.0: -1.0, 240.0: -0.5, 270.0: 0.0, 300.0: 0.5}
    if d in table:
        return np.float32(table[d])
    return np.float32(np.cos(np.deg2rad(np.float64(d))))


def sind_cosd32(x):
    """sind.(x), cosd.(x) on Float32 angles in [-90, 90] (stream.jl:166-169): Base reduces by quadrants in degrees (exact at
    multiples of 90) and evaluates on an extended-precision deg2rad -- here the Float64 function rounded once to Float32 with
    the exact values forced (a Julia wrapper simply calls cosd / sind: julia/FibersHIP.jl)."""
    xd = np.asarray(x, np.float32).astype(np.float64)
    sn, cs = np.sin(np.deg2rad(xd)), np.cos(np.deg2rad(xd))
    cs = np.where(np.abs(xd) == 90.0, 0.0, cs)
    sn = np.where(xd == 90.0, 1.0, np.where(xd == -90.0, -1.0, sn))
    return sn.astype(np.float32), cs.astype(np.float32)


def angles_to_vectors(vol, volres):
    """StreamWork's expansion of 2-D orientation angles (one frame) into 3-D vectors (stream.jl:147-172): the through-plane
    dimension is the one with the largest voxel size (`argmax(volres)`: the first maximum), the other two carry
    (cos, sin) -- radians when every value lies in [-pi/2 - eps(Float32), pi/2 + eps(Float32)] (tested first), degrees when in
    [-90, 90], the reference's error otherwise.  Returns ([nx,ny,nz,3] float32 Fortran-ordered, thrudim 0-based).
    (The `.* omask_array` of the reference is the field kernel's job: masked voxels get a zero vector either way.)"""
    a = np.asarray(vol, np.float32)
    if a.ndim == 4:
        a = a[..., 0]
    thru = int(np.argmax(np.asarray(volres, np.float32)))
    sd = [c for c in range(3) if c != thru]
    eps32 = float(np.finfo(np.float32).eps)
    lo, hi = float(a.min()), float(a.max())
    out = np.zeros(a.shape + (3,), np.float32, order="F")
    if -np.pi / 2 - eps32 <= lo and hi <= np.pi / 2 + eps32:
        out[..., sd[0]], out[..., sd[1]] = np.cos(a), np.sin(a)
    elif -90 <= lo and hi <= 90:
        sn, cs = sind_cosd32(a)
        out[..., sd[0]], out[..., sd[1]] = cs, sn
    else:
        raise ValueError("Input orientations should be 3D vectors or angles in [-90, 90]")       # stream.jl:170
    return out, thru


def make_sublist(nsub: int, rng=None) -> np.ndarray:
    """Sub-voxel sampling offsets (stream.jl:176-181): nsub uniform draws in (-.5+eps, .5-eps)^3 shared by
    all seeds; nsub == 0 -> a single zero offset.  The reference draws from Julia's global RNG, which
    cannot be reproduced: callers that need a specific realisation pass `sublist` explicitly."""
    if nsub <= 0:
        return np.zeros((1, 3), np.float32)
    rng = np.random.default_rng(rng)
    return rng.uniform(-0.5 + _EPS, 0.5 - _EPS, size=(nsub, 3)).astype(np.float32)


def _as_list(x):
    if x is None:
        return None
    return [x] if isinstance(x, (MRI, np.ndarray)) else list(x)


def _vol3(m, what):
    a = m.vol if isinstance(m, MRI) else np.asarray(m)
    if a.ndim == 4:
        a = a[..., 0]
    if a.ndim != 3:
        raise ValueError("%s must be a 3-D volume" % what)
    return a


class StreamWorkspace:
    """Caller-owned scratch arena of the tracer on one device (fibd_stream_ws_create): keeps the multi-GB point scratch
    between calls.  One job at a time takes it; concurrent jobs fall back to their own allocation."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        self.device = int(device)
        _lib.check(_lib.lib().fibd_stream_ws_create(self.device, C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                _lib.lib().fibd_stream_ws_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


_default_ws = {}


def default_workspace(device: int) -> StreamWorkspace:
    """the host mirror's workspace for `device` (what a StreamWork would own in the reference, stream.jl:43-60)"""
    ws = _default_ws.get(int(device))
    if ws is None:
        ws = _default_ws[int(device)] = StreamWorkspace(device)
    return ws


def _params(shape, nvec, len_min, len_max, ang_thresh, step_size, smooth_coeff, search_dist=0, search_ang=10, ws=None, interp="nearest",
            search_flat_axis=0):
    """search_dist > 0 selects the microscopy regime (stream.jl:83, 547-619); interp: "nearest" (the reference) or "trilinear"
    (fib_stream_params.interp in include/fibers_hip.h); search_flat_axis 1..3: that axis is searched over one voxel only
    (the through-plane axis of 2-D angle inputs, stream.jl:153-155)"""
    if interp not in ("nearest", "trilinear"):
        raise ValueError("interp must be 'nearest' or 'trilinear'")
    nx, ny, nz = shape
    return _lib.StreamParams(nx, ny, nz, nvec, int(len_min), int(len_max if len_max is not None else max(shape)),
                             float(cosd32(ang_thresh)), float(np.float32(step_size)), float(np.float32(smooth_coeff)),
                             int(search_dist), float(cosd32(search_ang)), ws._h if ws is not None else None, 1 if interp == "trilinear" else 0,
                             int(search_flat_axis))


def stream(ovec: Union[MRI, Sequence[MRI]], *, f=None, f_thresh: float = 0.03, fa: Optional[MRI] = None,
           fa_thresh: float = 0.1, mask: Optional[MRI] = None, seed: Optional[MRI] = None, nsub: Optional[int] = 3,
           len_min: int = 3, len_max: Optional[int] = None, ang_thresh: Optional[float] = 45,
           step_size: Optional[float] = 0.5, smooth_coeff: Optional[float] = 0.2, lcms=None, lcm_thresh: float = 0.099,
           search_dist: int = 15, search_ang: float = 10, sublist=None, rng=None, rng_seed: int = 0,
           device: int = 0, interp: str = "nearest") -> Tract:
    """Streamline tractography (stream.jl:730).  Returns a `Tract` whose lines are in the reference's order
    (seed voxels in column-major `findall` order, sub-voxel offsets innermost), points in 1-based voxel
    coordinates, each line ordered [forward reversed, backward] as stream.jl:652 builds it.
    Volumes with a voxel size of 0.05 mm or less (stream.jl:83) are tracked in the microscopy regime
    (stream_micro_new_point!, stream.jl:547-619: cone search of `search_dist` voxels / `search_ang` degrees).
    `lcms` [nx,ny,nz,10] switches to LCM-guided tracking (stream.jl:380-495); the reference samples from Julia's global
    RNG there, this back end from the counter-based stream defined in include/fibers_hip.h (`rng_seed`); the returned
    Tract then carries the per-point method-difference indicator as `scalars`."""
    ovecs = _as_list(ovec)
    fs = _as_list(f)
    if mask is None:
        raise ValueError("mask is required (the reference builds the Tract header from it, stream.jl:784)")
    vols = []
    flat_axis, angle_thru = 0, None
    for o in ovecs:
        v = o.vol if isinstance(o, MRI) else np.asarray(o)
        if v.ndim == 4 and v.shape[3] == 3:                     # orientation vectors (stream.jl:141)
            vols.append(np.asfortranarray(v, dtype=np.float32))
        elif v.ndim == 3 or (v.ndim == 4 and v.shape[3] == 1):  # 2-D orientation angles (stream.jl:147-172)
            e, thru = angles_to_vectors(v, o.volres if isinstance(o, MRI) else (1.0, 1.0, 1.0))
            vols.append(e)
            flat_axis, angle_thru = thru + 1, thru
        else:
            raise ValueError("Input orientations should be 3D vectors or angles in [-90, 90]")
    shape = vols[0].shape[:3]
    # Is this in the microscopy regime (min voxel size under 50 um)?  (stream.jl:83)
    domicro = min(ovecs[0].volres if isinstance(ovecs[0], MRI) else (1, 1, 1)) <= 0.05
    # scale-dependent defaults when `nothing` is passed (stream.jl:89-92)
    nsub = (0 if domicro else 3) if nsub is None else nsub
    ang_thresh = (20 if domicro else 45) if ang_thresh is None else ang_thresh
    step_size = (1 if domicro else 0.5) if step_size is None else step_size
    smooth_coeff = (0 if domicro else 0.2) if smooth_coeff is None else smooth_coeff
    nvec = len(vols)
    fvols = None
    if fs is not None:
        if len(fs) != nvec:
            raise ValueError("need one amplitude volume per orientation volume")
        fvols = [np.asfortranarray(_vol3(x, "f"), dtype=np.float32) for x in fs]
        _warn_thresh("f_thresh", f_thresh, fvols[0], _vol3(mask, "mask") > 0)           # stream.jl:121-127
    favol = None
    if fa is not None:
        favol = np.asfortranarray(_vol3(fa, "fa"), dtype=np.float32)
        _warn_thresh("fa_thresh", fa_thresh, favol, _vol3(mask, "mask") > 0)            # stream.jl:107-113
    m, mdt = _mask_arg(mask)
    if m.shape != shape:
        raise ValueError("mask shape %s does not match orientation volume %s" % (m.shape, shape))
    sptr, sdt = None, 0
    if seed is not None:
        sv = _vol3(seed, "seed")
        if sv.shape != m.shape:
            raise RuntimeError("Dimension mismatch between seed mask %s and brain mask %s"
                               % (sv.shape, m.shape))                                 # stream.jl:746-749
        sarr, sdt = _mask_arg(seed)
        sptr = sarr.ctypes.data
    sub = make_sublist(nsub, rng) if sublist is None else np.ascontiguousarray(sublist, np.float32).reshape(-1, 3)
    prm = _params(shape, nvec, len_min, len_max, ang_thresh, step_size, smooth_coeff,
                  int(search_dist) if domicro else 0, search_ang, interp=interp,
                  search_flat_axis=flat_axis if domicro else 0)                 # micro_search_dist[thrudim] = 0 (stream.jl:153-155)
    ov = (C.c_void_p * nvec)(*[v.ctypes.data for v in vols])
    fv = None if fvols is None else (C.c_void_p * nvec)(*[v.ctypes.data for v in fvols])
    out = _lib.TractOut()
    L = _lib.lib()
    scal = None
    if lcms is None:
        _lib.check(L.fib_stream(device, C.byref(prm), ov, fv, float(np.float32(f_thresh)),
                                None if favol is None else favol.ctypes.data, float(np.float32(fa_thresh)),
                                m.ctypes.data, mdt, sptr, sdt, sub.ctypes.data, sub.shape[0], C.byref(out)))
    else:
        if domicro:
            raise ValueError("LCM-guided tracking is a macro-scale mode (voxel size > 0.05 mm)")
        if angle_thru is not None and angle_thru != 2:
            # stream.jl:221 derives the LCM's in-plane dimensions from the FRAMES of the first input volume; with one-frame angle
            # inputs that yields dimensions (1, 2) whatever the slice orientation, which only matches the expanded vectors when
            # the through-plane dimension is the third
            raise ValueError("LCM-guided tracking on 2-D angle inputs needs the through-plane dimension to be the third "
                             "(largest voxel size along z): the reference pairs the LCM edges with dimensions (1, 2) there")
        lv = lcms.vol if isinstance(lcms, MRI) else np.asarray(lcms)
        if lv.shape != shape + (10,):
            raise ValueError("lcms must be [nx,ny,nz,10] (vectorised 4x4 symmetric local connection matrices)")
        lv = np.asfortranarray(lv, dtype=np.float32)
        if lcm_thresh > float(lv.max()):                                      # stream.jl:210-215
            print("WARNING: The value of lcm_thresh (%s) is greater than the maximum value in the lcms volume (%s)"
                  % (lcm_thresh, float(lv.max())))
        _lib.check(L.fib_stream_lcm(device, C.byref(prm), ov, fv, float(np.float32(f_thresh)),
                                    None if favol is None else favol.ctypes.data, float(np.float32(fa_thresh)),
                                    m.ctypes.data, mdt, sptr, sdt, sub.ctypes.data, sub.shape[0],
                                    lv.ctypes.data, float(np.float32(lcm_thresh)), int(rng_seed), C.byref(out)))
    try:
        nl, npnt = int(out.nlines), int(out.npoints)
        npts = np.ctypeslib.as_array(out.npts, shape=(max(nl, 1),))[:nl].copy()
        sidx = np.ctypeslib.as_array(out.seed_index, shape=(max(nl, 1),))[:nl].copy()
        xyz = np.ctypeslib.as_array(out.xyz, shape=(max(npnt, 1) * 3,))[: npnt * 3].copy().reshape(-1, 3)
        if lcms is not None:
            scal = np.ctypeslib.as_array(out.flags, shape=(max(npnt, 1),))[:npnt].astype(np.float32)
    finally:
        L.fib_tract_free(C.byref(out))
    ref = mask if isinstance(mask, MRI) else None
    return Tract(xyz=xyz, npts=npts, seed_index=sidx, volsize=shape,
                 volres=tuple(ref.volres) if ref is not None else (1.0, 1.0, 1.0),
                 vox2ras=ref.vox2ras.copy() if ref is not None else np.eye(4, dtype=np.float32),
                 sublist=sub, scalars=scal)


def _warn_thresh(name, thr, vol, maskbool):
    """`println("WARNING: ...")` on implausible thresholds (stream.jl:109-113, 123-127)"""
    vals = vol[maskbool]
    if vals.size == 0:
        return
    lo, hi = np.quantile(vals, 1e-5), np.quantile(vals, 0.9)
    if thr < lo or thr > hi:
        print("WARNING: The value of %s (%s) is outside the range of most values in the %s volume (%s, %s)"
              % (name, thr, name.split("_")[0], lo, hi))


# ---------------------------------------------------------------------------------------------
# device-resident form
# ---------------------------------------------------------------------------------------------
def angles_to_vectors_device(ang, volres=(1.0, 1.0, 1.0)):
    """angles_to_vectors for a device-resident angle volume (float32 CUDA tensor of nvox angles): planar [3, nvox] vectors by
    the same rules (stream.jl:147-172) -- radians if all values lie in [-pi/2, pi/2] (+- eps), degrees if in [-90, 90]."""
    import torch
    _chk_dev(ang, torch.float32, "angles")
    a = ang.reshape(-1)
    thru = int(np.argmax(np.asarray(volres, np.float32)))
    sd = [c for c in range(3) if c != thru]
    eps32 = float(np.finfo(np.float32).eps)
    lo, hi = float(a.min()), float(a.max())
    out = torch.zeros((3, a.numel()), dtype=torch.float32, device=a.device)
    if -np.pi / 2 - eps32 <= lo and hi <= np.pi / 2 + eps32:
        out[sd[0]], out[sd[1]] = torch.cos(a), torch.sin(a)
    elif -90 <= lo and hi <= 90:
        ad = a.double()
        cs = torch.where(ad.abs() == 90.0, torch.zeros_like(ad), torch.cos(torch.deg2rad(ad)))
        sn = torch.where(ad.abs() == 90.0, torch.sign(ad), torch.sin(torch.deg2rad(ad)))
        out[sd[0]], out[sd[1]] = cs.float(), sn.float()
    else:
        raise ValueError("Input orientations should be 3D vectors or angles in [-90, 90]")
    return out, thru


def stream_field_device(ovec: List, f: Optional[List] = None, f_thresh: float = 0.03, fa=None, fa_thresh: float = 0.1,
                        mask=None, stream=None):
    """StreamWork mask + repack on the GPU (stream.jl:95-145).  ovec[k]: float32 CUDA [3, nvox]; f[k], fa: [nvox];
    mask uint8 [nvox] (already `> 0`-tested) or None.  Returns (field float32 [nvox, nvec, 4], mask uint8 [nvox])."""
    import torch
    nvec = len(ovec)
    nvox = ovec[0].numel() // 3
    for t in ovec:
        _chk_dev(t, torch.float32, "ovec")
    dev = ovec[0].device
    field = torch.empty((nvox, nvec, 4), dtype=torch.float32, device=dev)
    mout = torch.empty(nvox, dtype=torch.uint8, device=dev)
    ov = (C.c_void_p * nvec)(*[t.data_ptr() for t in ovec])
    fv = None if f is None else (C.c_void_p * nvec)(*[_chk_dev(t, torch.float32, "f").data_ptr() for t in f])
    _lib.check(_lib.lib().fibd_stream_field(nvec, nvox, ov, fv, float(np.float32(f_thresh)),
                                            None if fa is None else _chk_dev(fa, torch.float32, "fa").data_ptr(),
                                            float(np.float32(fa_thresh)),
                                            None if mask is None else _chk_dev(mask, torch.uint8, "mask").data_ptr(),
                                            field.data_ptr(), mout.data_ptr(), _stream_ptr(stream)))
    return field, mout


def stream_device(field, shape, seeds, sublist, len_min=3, len_max=None, ang_thresh=45, step_size=0.5,
                  smooth_coeff=0.2, stream=None, want_all_npts=False, search_dist=0, search_ang=10,
                  lcms=None, lcm_thresh=0.099, strdims=(0, 1), rng_seed=0, xyz_out=None, workspace="default", interp="nearest"):
    """Trace + pack on the GPU.  field: [nvox, nvec, 4] from stream_field_device; seeds: int64 CUDA tensor of
    0-based column-major voxel indices (findall order); sublist: float32 CUDA [nsub, 3].
    search_dist > 0: microscopy regime (stream.jl:547-619; reference defaults there: search_dist 15, search_ang 10,
    ang_thresh 20, step_size 1, smooth_coeff 0, one zero sub-voxel offset).
    lcms (float32 CUDA [10, nvox], planar like MRI.vol[nx,ny,nz,10]): LCM-guided tracking (stream.jl:380-495) over the
    in-plane dimensions `strdims`, uniforms from the ABI's counter-based stream (`rng_seed`); adds `flags` uint8 [npoints].
    xyz_out: optional callable npoints -> float32 CUDA tensor of at least 3 * npoints elements to pack the points into (any
    4-byte alignment).  workspace: a StreamWorkspace, None (scratch allocated and freed by the job) or "default" (the
    host mirror's arena for the field's device).  interp="trilinear": the non-reference option of fib_stream_params.interp.
    Returns dict(npts int32 [nlines], seed_index int64 [nlines], xyz float32 [npoints, 3])."""
    import torch
    _chk_dev(field, torch.float32, "field")
    _chk_dev(seeds, torch.int64, "seeds")
    _chk_dev(sublist, torch.float32, "sublist")
    nvec = field.shape[1]
    ws = default_workspace(field.device.index or 0) if isinstance(workspace, str) else workspace
    prm = _params(shape, nvec, len_min, len_max, ang_thresh, step_size, smooth_coeff, search_dist, search_ang, ws, interp)
    job = C.c_void_p()
    nl, npnt = C.c_int64(0), C.c_int64(0)
    L = _lib.lib()
    sp = _stream_ptr(stream)
    if lcms is None:
        _lib.check(L.fibd_stream_trace(C.byref(prm), field.data_ptr(), seeds.data_ptr(), seeds.numel(),
                                       sublist.data_ptr(), sublist.shape[0], sp, C.byref(job), C.byref(nl), C.byref(npnt)))
    else:
        _chk_dev(lcms, torch.float32, "lcms")
        _lib.check(L.fibd_stream_trace_lcm(C.byref(prm), field.data_ptr(), lcms.data_ptr(), float(np.float32(lcm_thresh)),
                                           int(strdims[0]), int(strdims[1]), int(rng_seed), seeds.data_ptr(), seeds.numel(),
                                           sublist.data_ptr(), sublist.shape[0], sp, C.byref(job), C.byref(nl), C.byref(npnt)))
    try:
        dev = field.device
        out = dict(npts=torch.empty(nl.value, dtype=torch.int32, device=dev),
                   seed_index=torch.empty(nl.value, dtype=torch.int64, device=dev),
                   xyz=torch.empty((npnt.value, 3), dtype=torch.float32, device=dev) if xyz_out is None
                   else xyz_out(npnt.value)[:3 * npnt.value].view(npnt.value, 3))
        if lcms is None:
            _lib.check(L.fibd_stream_pack(job, out["npts"].data_ptr(), out["seed_index"].data_ptr(), out["xyz"].data_ptr(), sp))
        else:
            out["flags"] = torch.empty(npnt.value, dtype=torch.uint8, device=dev)
            _lib.check(L.fibd_stream_pack_flags(job, out["npts"].data_ptr(), out["seed_index"].data_ptr(),
                                                out["xyz"].data_ptr(), out["flags"].data_ptr(), sp))
        if want_all_npts:
            out["all_npts"] = torch.empty(seeds.numel() * sublist.shape[0], dtype=torch.int32, device=dev)
            _lib.check(L.fibd_stream_all_npts(job, out["all_npts"].data_ptr(), sp))
        _sync(stream)
    finally:
        L.fib_stream_job_destroy(job)
    return out


class StreamBuffers:
    """Caller-owned output buffers of `stream_device_run` (npts int32 [lines], seed_index int64 [lines], xyz float32 [points, 3]):
    kept between calls, grown when a call reports that it needs more (the steady state of a stream of volumes allocates nothing)."""

    def __init__(self, device, lines: int = 0, points: int = 0):
        self.device = device
        self.npts = self.seed_index = self.xyz = None
        self.reserve(lines, points)

    def reserve(self, lines: int, points: int):
        import torch
        if self.npts is None or self.npts.numel() < lines:
            self.npts = torch.empty(int(lines), dtype=torch.int32, device=self.device)
            self.seed_index = torch.empty(int(lines), dtype=torch.int64, device=self.device)
        if self.xyz is None or self.xyz.shape[0] < points:
            self.xyz = torch.empty((int(points), 3), dtype=torch.float32, device=self.device)


def stream_device_run(field, shape, seeds, sublist, buffers: StreamBuffers = None, len_min=3, len_max=None, ang_thresh=45, step_size=0.5,
                      smooth_coeff=0.2, stream=None, workspace="default", interp="nearest"):
    """stream_device in ONE library call (fibd_stream_run): trace, scan and pack without a host round trip in between -- from 2^21
    lines on (nearest-voxel tracking, 1 or 3 vectors per voxel) as ONE kernel in which the workgroup that traced 512 lines packs them
    behind a decoupled look-back; results go straight into `buffers` (grown and the call repeated when they are too small: a first
    call sizes them).  Same lines, order and layout as stream_device; macro-scale angle picking only.
    Returns dict(npts, seed_index, xyz) -- views of the buffers, valid until the next call with them."""
    import torch
    _chk_dev(field, torch.float32, "field")
    _chk_dev(seeds, torch.int64, "seeds")
    _chk_dev(sublist, torch.float32, "sublist")
    nvec = field.shape[1]
    ws = default_workspace(field.device.index or 0) if isinstance(workspace, str) else workspace
    prm = _params(shape, nvec, len_min, len_max, ang_thresh, step_size, smooth_coeff, 0, 10, ws, interp)
    nl_max = int(seeds.numel()) * int(sublist.shape[0])
    if buffers is None:
        buffers = StreamBuffers(field.device)
    if buffers.npts is None or buffers.npts.numel() == 0:
        buffers.reserve(nl_max, 32 * nl_max)                    # a first guess; the call below says what is needed
    L = _lib.lib()
    sp = _stream_ptr(stream)
    nl, npnt = C.c_int64(0), C.c_int64(0)
    for attempt in range(2):
        rc = L.fibd_stream_run(C.byref(prm), field.data_ptr(), seeds.data_ptr(), seeds.numel(), sublist.data_ptr(), sublist.shape[0],
                               buffers.npts.data_ptr(), buffers.seed_index.data_ptr(), buffers.npts.numel(),
                               buffers.xyz.data_ptr(), buffers.xyz.shape[0], C.byref(nl), C.byref(npnt), sp)
        if rc == _lib.FIB_ERR_CAPACITY and attempt == 0:
            buffers.reserve(int(nl.value * 1.02) + 16, int(npnt.value * 1.02) + 1024)
            continue
        _lib.check(rc)
        break
    return dict(npts=buffers.npts[: nl.value], seed_index=buffers.seed_index[: nl.value], xyz=buffers.xyz[: npnt.value], buffers=buffers)


def stream_device_run_enqueue(field, shape, seeds, sublist, buffers: StreamBuffers, counts=None, len_min=3, len_max=None, ang_thresh=45,
                              step_size=0.5, smooth_coeff=0.2, stream=None, workspace="default", interp="nearest"):
    """stream_device_run without the host round trip at its end (fibd_stream_run_enqueue): returns as soon as the work is enqueued.
    `buffers` must already be large enough (a stream_device_run call sizes them); `counts` (int64 CUDA tensor of 2 elements, made if
    None) receives {lines, points} from the stream -- read it after synchronising; values above the buffers' capacities mean that
    lines were dropped for lack of room.  Returns (buffers, counts)."""
    import torch
    _chk_dev(field, torch.float32, "field")
    _chk_dev(seeds, torch.int64, "seeds")
    _chk_dev(sublist, torch.float32, "sublist")
    if buffers is None or buffers.npts is None or buffers.npts.numel() == 0:
        raise ValueError("stream_device_run_enqueue needs sized buffers (call stream_device_run once)")
    if counts is None:
        counts = torch.zeros(2, dtype=torch.int64, device=field.device)
    _chk_dev(counts, torch.int64, "counts")
    nvec = field.shape[1]
    ws = default_workspace(field.device.index or 0) if isinstance(workspace, str) else workspace
    prm = _params(shape, nvec, len_min, len_max, ang_thresh, step_size, smooth_coeff, 0, 10, ws, interp)
    _lib.check(_lib.lib().fibd_stream_run_enqueue(C.byref(prm), field.data_ptr(), seeds.data_ptr(), seeds.numel(), sublist.data_ptr(),
                                                  sublist.shape[0], buffers.npts.data_ptr(), buffers.seed_index.data_ptr(), buffers.npts.numel(),
                                                  buffers.xyz.data_ptr(), buffers.xyz.shape[0], counts.data_ptr(), _stream_ptr(stream)))
    return buffers, counts
