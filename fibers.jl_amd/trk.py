""".trk (TrackVis v2) reader / writer and `Tract` assembly: the step right after the hot path
(SURVEY.md §8f N1).  Mirrors `Tract{T}(ref::MRI)` (trk.jl:88-144), `str_add!` (trk.jl:166-266),
`trk_read` (trk.jl:358-423) and `trk_write` (trk.jl:433-495).

The reference quirk is kept: `stream` emits 1-based voxel coordinates (stream.jl:649) while
`trk_write` treats xyz as 0-based and writes (xyz + .5) * voxel_size (trk.jl:475-476); the values
are reproduced, the offset is not "fixed".  `stream_to_trk` serialises straight from the GPU: the
pack kernel writes the file body (Int32 npts + npts x 3 Float32 per line) in HBM, so the host only
prepends the 1000-byte header."""
import ctypes as C
import struct

import numpy as np

from . import _lib
from .mri import MRI
from .tract import Tract

_HDR = struct.Struct("<6s3h3f3fh200sh200s16f444s4s4s6f2s6B3i")      # 1000 bytes, trk.jl:13-35
assert _HDR.size == 1000


def vox2ras_to_orient(M):
    """mri.jl:471-499"""
    out = ""
    for d in range(3):
        col = np.asarray(M)[:3, d]
        i = int(np.argmax(np.abs(col)))
        out += ("RL", "AP", "SI")[i][0 if col[i] > 0 else 1]
    return out


def tract_header(ref: MRI, n_count=0, n_scalars=0, n_properties=0) -> bytes:
    """Tract{T}(ref::MRI) header fields (trk.jl:88-144), serialised as trk_write does (trk.jl:441-466)"""
    M = np.asarray(ref.vox2ras, np.float32)
    orient = vox2ras_to_orient(M)
    res = np.asarray(ref.volres, np.float64)
    p2s = (np.diag([-1.0, -1.0, 1.0]) @ M[:3, :2].astype(np.float64)) @ np.diag(1.0 / res[:2])   # trk.jl:108-109
    vo = orient.encode() + b"\0"
    return _HDR.pack(b"TRACK\0", *[int(v) for v in ref.volsize], *[float(np.float32(v)) for v in res],
                     0.0, 0.0, 0.0, n_scalars, b"\0" * 200, n_properties, b"\0" * 200,
                     *[float(v) for v in M.reshape(-1)],            # row-major == permutedims(vox_to_ras) column-major
                     b"\0" * 444, vo, vo, *[float(np.float32(v)) for v in p2s.reshape(-1, order="F")], b"\0" * 2,
                     0, 0, 0, 0, 0, 0, int(n_count), 2, 1000)


def trk_body(tr: Tract, voxel_size) -> bytes:
    """per line: Int32 npts, then per point T.((xyz .+ .5) .* voxel_size) followed by the point's scalars, then the line's
    properties (trk.jl:471-482; Float64 arithmetic for the coordinates)"""
    vs = np.asarray(voxel_size, np.float32).astype(np.float64)
    pts = ((tr.xyz.astype(np.float64) + 0.5) * vs).astype(np.float32)
    ns, npr = tr.n_scalars, tr.n_properties
    npnt = pts.shape[0]
    rec = pts
    if ns:
        sc = np.asarray(tr.scalars, np.float32).reshape(npnt, ns)
        rec = np.concatenate([pts, sc], axis=1)
    w = 3 + ns
    off = tr.offsets
    out = np.empty(tr.nstr * (1 + npr) + w * npnt, np.float32)
    starts = np.arange(tr.nstr, dtype=np.int64) * (1 + npr) + w * off[:-1]           # position of each line's Int32 npts
    out.view(np.int32)[starts] = tr.npts
    keep = np.ones(out.shape[0], bool)
    keep[starts] = False
    if npr:
        pr = np.asarray(tr.properties, np.float32).reshape(tr.nstr, npr)
        pstart = starts + 1 + w * tr.npts.astype(np.int64)
        idx = (pstart[:, None] + np.arange(npr)[None, :]).reshape(-1)
        out[idx] = pr.reshape(-1)
        keep[idx] = False
    out[keep] = rec.reshape(-1)
    return out.tobytes()


def trk_write(tr: Tract, outfile: str, ref: MRI = None) -> bool:
    """trk_write(tr, outfile) (trk.jl:433-495).  Returns True if the byte count is not the expected one."""
    ref = ref if ref is not None else MRI(np.zeros(tuple(tr.volsize) + (1,), np.uint8), volres=tr.volres, vox2ras=tr.vox2ras)
    hdr = tract_header(ref, n_count=tr.nstr, n_scalars=tr.n_scalars, n_properties=tr.n_properties)
    body = trk_body(tr, np.asarray(ref.volres, np.float32))
    with open(outfile, "wb") as fh:
        nb = fh.write(hdr) + fh.write(body)
    npnt = int(tr.npts.sum())
    return nb != 1000 + 4 * tr.nstr * (1 + tr.n_properties) + 4 * (3 + tr.n_scalars) * npnt


def trk_read(infile: str) -> Tract:
    """trk_read (trk.jl:358-423): xyz = file ./ voxel_size .- .5; per-point scalars and per-line properties are kept
    (trk.jl:404-416)"""
    with open(infile, "rb") as fh:
        raw = fh.read()
    f = _HDR.unpack(raw[:1000])
    dim, vs = f[1:4], np.array(f[4:7], np.float32)
    n_scalars, n_props = f[10], f[12]
    M = np.array(f[14:30], np.float32).reshape(4, 4)
    n_count = f[-3]
    body = np.frombuffer(raw, np.float32, offset=1000)
    ibody = body.view(np.int32)
    npts = np.zeros(n_count, np.int32)
    chunks, schunks, props = [], [], []
    pos = 0
    for i in range(n_count):
        n = int(ibody[pos]); pos += 1
        rec = body[pos: pos + n * (3 + n_scalars)].reshape(n, 3 + n_scalars)
        chunks.append(rec[:, :3])
        schunks.append(rec[:, 3:])
        pos += n * (3 + n_scalars)
        props.append(body[pos: pos + n_props])
        pos += n_props
        npts[i] = n
    xyz = np.concatenate(chunks) if chunks else np.zeros((0, 3), np.float32)
    xyz = (xyz / vs - np.float32(0.5)).astype(np.float32)                       # trk.jl:410-411
    scalars = properties = None
    if n_scalars:
        scalars = np.concatenate(schunks).astype(np.float32) if schunks else np.zeros((0, n_scalars), np.float32)
        if n_scalars == 1:
            scalars = scalars[:, 0]
    if n_props:
        properties = np.stack(props).astype(np.float32) if props else np.zeros((0, n_props), np.float32)
    return Tract(xyz=xyz, npts=npts, volsize=tuple(int(d) for d in dim), volres=tuple(float(v) for v in vs), vox2ras=M,
                 scalars=scalars, properties=properties)


def str_add(tr: Tract, lines) -> Tract:
    """str_add!(tr, xyz) (trk.jl:166-266) for coordinate-only streamlines: `lines` is a list of [3 x npts]
    (reference shape) or [npts x 3] arrays appended after the existing ones."""
    arrs = []
    for a in lines:
        a = np.asarray(a, np.float32)
        if a.ndim != 2 or 3 not in a.shape:
            raise ValueError("Each streamline must be defined as a matrix with 3 rows")
        arrs.append(a.T if a.shape[0] == 3 and a.shape[1] != 3 else a)
    tr.xyz = np.concatenate([tr.xyz] + arrs) if arrs else tr.xyz
    tr.npts = np.concatenate([tr.npts, np.array([a.shape[0] for a in arrs], np.int32)])
    return tr


_STAGE = {}                      # device index -> two pinned staging buffers + a copy stream (kept: allocating pinned memory costs ~50 us per MB)


def _download_and_write(path, header: bytes, body, piece: int = 32 << 20):
    """header + a device tensor's bytes to `path`: the body comes down in pieces through two pinned staging buffers while the previous
    piece goes to the file (a box writes its page cache at ~10 GB/s from one thread and no faster from eight; the download runs at ~50)"""
    import os
    import torch
    dev = body.device
    st = _STAGE.get(dev.index)
    if st is None:
        st = _STAGE[dev.index] = dict(buf=[torch.empty(piece, dtype=torch.uint8, pin_memory=True) for _ in range(2)], stream=torch.cuda.Stream(dev),
                                      ev=[torch.cuda.Event() for _ in range(2)])
    raw = body.view(torch.uint8).reshape(-1)
    n = raw.numel()
    fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
    try:
        os.pwrite(fd, header, 0)
        st["stream"].wait_stream(torch.cuda.current_stream(dev))
        offs = list(range(0, n, piece))

        def issue(k):
            o = offs[k]
            with torch.cuda.stream(st["stream"]):
                st["buf"][k & 1][: min(piece, n - o)].copy_(raw[o: o + piece], non_blocking=True)
                st["ev"][k & 1].record(st["stream"])

        def drain(k):
            o = offs[k]
            m = min(piece, n - o)
            st["ev"][k & 1].synchronize()
            view = memoryview(st["buf"][k & 1].numpy())[:m]
            done = 0
            while done < m:
                done += os.pwrite(fd, view[done:], len(header) + o + done)
        for k in range(len(offs)):
            issue(k)
            if k:
                drain(k - 1)
        if offs:
            drain(len(offs) - 1)
        os.ftruncate(fd, len(header) + n)                # (an existing longer file; the same length is a no-op)
    finally:
        os.close(fd)


def stream_to_trk(outfile, field, shape, seeds, sublist, ref: MRI, stream=None, timings=None, **kw) -> dict:
    """GPU path: trace, then let the pack kernel emit the .trk body directly (device tier: fibd_stream_pack_trk, trk.jl:471-482), downloaded
    in pieces through pinned staging buffers while the previous piece is written.  timings (optional dict): perf_counter stamps `device_done` (trace + pack finished)
    and `file_done`."""
    import time
    import torch
    from .stream import _params, default_workspace
    from .dti import _stream_ptr, _sync
    nvec = field.shape[1]
    prm = _params(shape, nvec, kw.get("len_min", 3), kw.get("len_max"), kw.get("ang_thresh", 45),
                  kw.get("step_size", 0.5), kw.get("smooth_coeff", 0.2), ws=default_workspace(field.device.index or 0))
    job = C.c_void_p()
    nl, npnt = C.c_int64(0), C.c_int64(0)
    L = _lib.lib()
    sp = _stream_ptr(stream)
    _lib.check(L.fibd_stream_trace(C.byref(prm), field.data_ptr(), seeds.data_ptr(), seeds.numel(),
                                   sublist.data_ptr(), sublist.shape[0], sp, C.byref(job), C.byref(nl), C.byref(npnt)))
    try:
        body = torch.empty(nl.value + 3 * npnt.value, dtype=torch.float32, device=field.device)
        vs = (C.c_float * 3)(*[float(np.float32(v)) for v in ref.volres[:3]])
        _lib.check(L.fibd_stream_pack_trk(job, C.byref(vs), body.data_ptr(), sp))
        _sync(stream)                                   # the pack ran on `stream`: a copy only orders against the current one
        if timings is not None:
            timings["device_done"] = time.perf_counter()
        _download_and_write(outfile, tract_header(ref, n_count=nl.value), body)
    finally:
        L.fib_stream_job_destroy(job)
    if timings is not None:
        timings["file_done"] = time.perf_counter()
    return dict(nlines=nl.value, npoints=npnt.value)
