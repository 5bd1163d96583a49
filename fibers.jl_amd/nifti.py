"""NIfTI-1 reader / writer + DWI b-table files: the host-side formats either side of the hot path
(SURVEY.md §8f N2).  Mirrors the reference's `mri_read` / `mri_write` for `.nii` / `.nii.gz`
(mri.jl:611-733, 1394-1672, 1695-1919, 2059-2166), `mri_read_bfiles` (mri.jl:2179-2266), the
`<base>_<field>[k].nii.gz` naming of `dti_write` / `gqi_write` / `dsi_write` (dti.jl:344-349,
gqi.jl:210-225, dsi.jl:279-294) and the struct reload `mri_read(inbase, type)` (mri.jl:2276-2311).
Pure host code (NumPy); `.gz` goes through Python's gzip instead of shelling out to zcat/gzip
(mri.jl:1586-1591, 2160-2163).  MGH and Bruker inputs are not part of this back end."""
import glob
import gzip
import os
import re
import struct

import numpy as np

from .mri import MRI

_NIFTI_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64,
                 256: np.int8, 512: np.uint16, 768: np.uint32}                      # mri.jl:1601-1625
_DTYPE_CODES = {np.dtype(v): (k, np.dtype(v).itemsize * 8) for k, v in _NIFTI_DTYPES.items()}

# field layout of the 348-byte NIfTI-1 header (NIfTIheader, mri.jl:25-76)
_HDR = struct.Struct("<i10s18sihBB8h3fhhhh8f3fhbb4fii80s24shh6f4f4f4f16s4s")


def _open(fname, mode):
    return gzip.open(fname, mode) if fname.lower().endswith(".gz") else open(fname, mode)


def vox2ras_to_qform(M):
    """mri.jl:391-462 (mat44_to_quatern): returns b, c, d, x, y, z, qfac"""
    M = np.asarray(M, np.float64)
    x, y, z = M[0, 3], M[1, 3], M[2, 3]
    d = np.sqrt((M[:, :3] ** 2).sum(axis=0))
    R = M[:3, :3] / d
    det = np.linalg.det(R)
    if det == 0:
        raise ValueError("vox2ras determinant is 0")
    r11, r21, r31, r12, r22, r32, r13, r23, r33 = R[0, 0], R[1, 0], R[2, 0], R[0, 1], R[1, 1], R[2, 1], R[0, 2], R[1, 2], R[2, 2]
    qfac = 1.0
    if det < 0:
        r13, r23, r33, qfac = -r13, -r23, -r33, -1.0
    a = r11 + r22 + r33 + 1.0
    if a > 0.5:
        a = 0.5 * np.sqrt(a)
        b, c, dd = 0.25 * (r32 - r23) / a, 0.25 * (r13 - r31) / a, 0.25 * (r21 - r12) / a
    else:
        xd, yd, zd = 1.0 + r11 - (r22 + r33), 1.0 + r22 - (r11 + r33), 1.0 + r33 - (r11 + r22)
        if xd > 1:
            b = 0.5 * np.sqrt(xd); c = 0.25 * (r12 + r21) / b; dd = 0.25 * (r13 + r31) / b; a = 0.25 * (r32 - r23) / b
        elif yd > 1:
            c = 0.5 * np.sqrt(yd); b = 0.25 * (r12 + r21) / c; dd = 0.25 * (r23 + r32) / c; a = 0.25 * (r13 - r31) / c
        else:
            dd = 0.5 * np.sqrt(zd); b = 0.25 * (r13 + r31) / dd; c = 0.25 * (r23 + r32) / dd; a = 0.25 * (r21 - r12) / dd
        if a < 0:
            b, c, dd = -b, -c, -dd
    return b, c, dd, x, y, z, qfac


def load_nifti_hdr(buf):
    """load_nifti_hdr (mri.jl:1394-1560) on the first 348 bytes; returns dict incl. vox2ras"""
    (sz,) = struct.unpack("<i", buf[:4])
    if sz == 348:
        S, bswap = _HDR, False
    elif sz == struct.unpack(">i", struct.pack("<i", 348))[0]:
        S, bswap = struct.Struct(">" + _HDR.format[1:]), True
    else:
        raise ValueError("Invalid header size %d found in NIfTI header" % sz)
    f = S.unpack(buf[:348])
    h = dict(sizeof_hdr=f[0], dim_info=f[6], dim=list(f[7:15]), intent=(f[15], f[16], f[17], f[18]),
             datatype=f[19], bitpix=f[20], slice_start=f[21], pixdim=list(f[22:30]), vox_offset=f[30],
             scl_slope=f[31], scl_inter=f[32], slice_end=f[33], slice_code=f[34], xyzt_units=f[35],
             cal_max=f[36], cal_min=f[37], slice_duration=f[38], toffset=f[39], glmax=f[40], glmin=f[41],
             descrip=f[42], aux_file=f[43], qform_code=f[44], sform_code=f[45],
             quatern=list(f[46:52]), srow_x=np.array(f[52:56], np.float32), srow_y=np.array(f[56:60], np.float32),
             srow_z=np.array(f[60:64], np.float32), intent_name=f[64], magic=f[65], do_bswap=bswap)
    if h["dim"][1] < 0:                                           # > 32k columns, FreeSurfer (mri.jl:1431-1435)
        h["dim"][1], h["glmin"] = h["glmin"], 0
    xyzunits = h["xyzt_units"] & 7
    xyzscale = {1: 1000.0, 2: 1.0, 3: 0.001}.get(xyzunits)
    if xyzscale is None:
        print("WARNING: xyz units code %d is unrecognized, assuming mm" % xyzunits)
        xyzscale = 1.0
    tscale = {8: 1000.0, 16: 1.0, 32: 0.001}.get(h["xyzt_units"] & 56, 0.0)
    pd = h["pixdim"]
    h["pixdim"] = [pd[0]] + [np.float32(p * xyzscale) for p in pd[1:4]] + [np.float32(pd[4] * tscale)] + pd[5:]
    for k in ("srow_x", "srow_y", "srow_z"):
        h[k] = (h[k] * np.float32(xyzscale)).astype(np.float32)
    h["xyzt_units"] = 2 | 16
    sform = np.vstack([h["srow_x"], h["srow_y"], h["srow_z"], [0, 0, 0, 1]]).astype(np.float32)
    b, c, d, x, y, z = [np.float32(v) for v in h["quatern"]]
    a = np.float32(1) - (b * b + c * c + d * d)
    if abs(a) < 1.0e-7:
        a = np.float32(1) / np.sqrt(b * b + c * c + d * d)
        b, c, d, a = b * a, c * a, d * a, np.float32(0)
    else:
        a = np.sqrt(a)
    R = np.array([[a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c],
                  [2 * b * c + 2 * a * d, a * a + c * c - b * b - d * d, 2 * c * d - 2 * a * b],
                  [2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a + d * d - c * c - b * b]], np.float32)
    if h["pixdim"][0] < 0.0:
        R[:, 2] = -R[:, 2]
    qform = np.eye(4, dtype=np.float32)
    qform[:3, :3] = R * np.array(h["pixdim"][1:4], np.float32)
    qform[:3, 3] = [x, y, z]
    if h["sform_code"] != 0:
        vox2ras = sform
    elif h["qform_code"] != 0:
        vox2ras = qform
    else:
        print("WARNING: neither sform or qform are valid")
        vox2ras = np.diag(list(h["pixdim"][1:4]) + [1.0]).astype(np.float32)
    h.update(sform=sform, qform=qform, vox2ras=vox2ras)
    return h


def load_nifti(fname, headeronly=False, mmap=False):
    """load_nifti (mri.jl:1577-1672) -> (hdr dict, array in file order, x fastest == Fortran order).
    The reference shells out to `zcat` / `gunzip -c` for .gz files (mri.jl:1586-1591) and reads the whole volume into a Julia
    array; here .gz is inflated in-process, and with mmap=True an uncompressed, native-endian, unscaled .nii is not read at
    all: the returned array is a read-only memory map of the file, so the host tier of the fits gathers its chunks straight
    from the page cache into the pinned staging ring (file -> pinned -> HBM, no intermediate copy of the volume)."""
    if mmap and not fname.lower().endswith(".gz"):
        with open(fname, "rb") as fh:
            head = fh.read(352)
        hdr = load_nifti_hdr(head)
        last = max(i for i, v in enumerate(hdr["dim"]) if v != 0)
        dim = [int(v) for v in hdr["dim"][1:last + 1]]
        plain = hdr["scl_slope"] == 0 or (hdr["scl_inter"] == 0 and hdr["scl_slope"] == 1)
        if hdr["datatype"] in _NIFTI_DTYPES and not hdr["do_bswap"] and plain and not headeronly:
            dt = np.dtype(_NIFTI_DTYPES[hdr["datatype"]])
            off = int(round(hdr["vox_offset"]))
            if os.path.getsize(fname) != off + int(np.prod(dim)) * dt.itemsize:
                raise ValueError("%s, read a %s volume but did not reach end of file" % (fname, tuple(dim)))
            return hdr, np.memmap(fname, dtype=dt, mode="r", offset=off, shape=tuple(dim), order="F")
    # one pass: the header, then the volume read straight into the array it stays in (the reference reads the whole file into a Julia array,
    # mri.jl:1577-1672; round 5 read the file into a bytes object and copied it twice)
    with _open(fname, "rb") as fh:
        head = fh.read(352)
        hdr = load_nifti_hdr(head)
        last = max(i for i, v in enumerate(hdr["dim"]) if v != 0)
        dim = [int(v) for v in hdr["dim"][1:last + 1]]
        if hdr["datatype"] not in _NIFTI_DTYPES:
            raise ValueError("Data type %d not supported" % hdr["datatype"])
        dt = np.dtype(_NIFTI_DTYPES[hdr["datatype"]])
        if headeronly:
            return hdr, np.zeros([0] * len(dim), dt)
        off = int(round(hdr["vox_offset"]))
        n = int(np.prod(dim))
        if off >= len(head):
            skip = off - len(head)
            if skip and len(fh.read(skip)) != skip:
                raise ValueError("%s, read a %s volume but did not reach end of file" % (fname, tuple(dim)))
            vol = np.empty(n, dtype=dt.newbyteorder(">" if hdr["do_bswap"] else "<"))
            buf = memoryview(vol).cast("B")
            got = 0
            while got < len(buf):
                k = fh.readinto(buf[got:])
                if not k:
                    break
                got += k
            complete = got == len(buf) and fh.read(1) == b""
        else:                                                     # (a data offset inside the 352 bytes already read: not a file this writer makes)
            raw = head + fh.read()
            complete = len(raw) == off + n * dt.itemsize
            vol = np.frombuffer(raw, dtype=dt.newbyteorder(">" if hdr["do_bswap"] else "<"), count=n if complete else 0, offset=off)
    if not complete:
        raise ValueError("%s, read a %s volume but did not reach end of file" % (fname, tuple(dim)))
    if vol.dtype != dt or not vol.flags.writeable:
        vol = vol.astype(dt)                                      # (byte-swapped files; the frombuffer fallback)
    vol = vol.reshape(dim, order="F")
    if hdr["scl_slope"] != 0 and not (hdr["scl_inter"] == 0 and hdr["scl_slope"] == 1):
        vol = (vol * hdr["scl_slope"] + hdr["scl_inter"]).astype(dt)              # mri.jl:1664-1668
    return hdr, np.asfortranarray(vol)


def mri_read_bfiles(infile1, infile2):
    """mri_read_bfiles (mri.jl:2179-2232): files in any order -> (bval [n], bvec [n,3])"""
    tabs = []
    for f in (infile1, infile2):
        if not os.path.isfile(f):
            raise FileNotFoundError("Could not open " + f)
        tabs.append(np.atleast_2d(np.loadtxt(f, dtype=np.float32)))
    ival, ivec = (0, 1) if tabs[0].size < tabs[1].size else (1, 0)
    if tabs[ival].shape[1] != 1:
        if tabs[ival].shape[0] != 1:
            raise ValueError("Wrong format in table %s (should be single column or row)" % (infile1, infile2)[ival])
        tabs[ival] = tabs[ival].T
    if tabs[ivec].shape[1] != 3:
        if tabs[ivec].shape[0] != 3:
            raise ValueError("Wrong format in table %s (should be three columns or rows)" % (infile1, infile2)[ivec])
        tabs[ivec] = tabs[ivec].T
    if tabs[0].shape[0] != tabs[1].shape[0]:
        raise ValueError("Dimension mismatch between tables in %s %s and %s %s"
                         % (infile1, tabs[0].shape, infile2, tabs[1].shape))
    return tabs[ival][:, 0].copy(), tabs[ivec].copy()


def _normalise_bvec(g):
    with np.errstate(invalid="ignore", divide="ignore"):
        g = g / np.sqrt((g ** 2).sum(axis=1, keepdims=True))                      # mri.jl:711
    g[np.isnan(g)] = 0                                                            # mri.jl:712
    return g.astype(np.float32)


def mri_read(infile, headeronly=False, mmap=False):
    """mri_read for NIfTI inputs (mri.jl:611-733): volume + optional <stem>.bval[s]/.bvec[s] tables,
    gradient vectors normalised.  `vol` keeps the file's element type (the fits need Float32).
    mmap=True: see load_nifti (the volume stays in the file until a fit streams it to the GPU)."""
    low = infile.lower()
    if not (low.endswith(".nii") or low.endswith(".nii.gz")):
        raise ValueError("File extension not supported by this back end (NIfTI only): " + infile)
    hdr, vol = load_nifti(infile, headeronly, mmap=mmap)
    volsz = [int(v) for v in hdr["dim"][1:] if v > 0]
    if len(volsz) >= 5 and not headeronly:                                        # mri.jl:660-666
        vol = vol.reshape(volsz[0], volsz[1], volsz[2], -1, order="F")
    if headeronly:
        vol = np.zeros(tuple(volsz[:3]) + (volsz[3] if len(volsz) > 3 else 1,), vol.dtype, order="F")
    M = hdr["vox2ras"]
    mri = MRI(vol, volres=tuple(float(v) for v in np.sqrt((M[:3, :3].astype(np.float64) ** 2).sum(axis=0))),
              vox2ras=M.copy())
    mri.tr = float(hdr["pixdim"][4])
    mri.niftihdr = hdr
    stem = infile[: -7] if low.endswith(".nii.gz") else infile[: -4]
    bfile = next((stem + e for e in (".bvals", ".bval") if os.path.isfile(stem + e)), "")
    gfile = next((stem + e for e in (".bvecs", ".bvec") if os.path.isfile(stem + e)), "")
    if bfile and gfile:
        b, g = mri_read_bfiles(bfile, gfile)
        if len(b) == mri.nframes:                                                 # mri.jl:706
            mri.bval, mri.bvec = b, np.asfortranarray(_normalise_bvec(g))
    return mri


def mri_write(mri, outfile, datatype=None):
    """mri_write for NIfTI outputs (mri.jl:1695-1919 + save_nifti 2059-2166).  Returns True on error
    (byte count mismatch), like the reference."""
    low = outfile.lower()
    if not (low.endswith(".nii") or low.endswith(".nii.gz")):
        raise ValueError("File extension not supported by this back end (NIfTI only): " + outfile)
    vol = mri.vol
    dt = np.dtype(datatype if datatype is not None else vol.dtype)
    if dt not in _DTYPE_CODES:
        raise ValueError("Data type %s not supported" % dt)
    code, bitpix = _DTYPE_CODES[dt]
    nx, ny, nz, nf = vol.shape
    dim = [4 if nf > 1 else 3, nx, ny, nz, nf, 1, 1, 1]
    glmin = 0
    if dim[1] > 2 ** 15:
        glmin, dim[1] = dim[1], -1
    M = np.asarray(mri.vox2ras, np.float64)
    b, c, d, x, y, z, qfac = vox2ras_to_qform(M)
    pixdim = [qfac] + [float(v) for v in mri.volres[:3]] + [float(getattr(mri, "tr", 0.0)), 0.0, 0.0, 0.0]
    nh = getattr(mri, "niftihdr", None) or {}
    hdr = _HDR.pack(348, b"\0" * 10, b"\0" * 18, 0, 0, 0, 0, *dim, 0.0, 0.0, 0.0, 0, code, bitpix, 0, *pixdim,
                    352.0, float(nh.get("scl_slope", 0.0)), float(nh.get("scl_inter", 0.0)), 0, 0, 2 | 16,
                    float(vol.max()) if vol.size else 0.0, float(vol.min()) if vol.size else 0.0, 0.0, 0.0, 0, glmin,
                    ("%-80s" % "FreeSurfer julia").encode(), b"\0" * 24, 1, 1, b, c, d, x, y, z,
                    *[float(v) for v in M[0]], *[float(v) for v in M[1]], *[float(v) for v in M[2]],
                    b"huh?" + b"\0" * 12, b"n+1\0")
    data = np.asfortranarray(vol.astype(dt)).tobytes(order="F")
    with _open(outfile, "wb") as fh:
        nb = fh.write(hdr) + fh.write(b"\0" * 4) + fh.write(data)
    stem = outfile[: -7] if low.endswith(".nii.gz") else outfile[: -4]
    if mri.bval is not None and len(mri.bval):
        np.savetxt(stem + ".bvals", np.asarray(mri.bval, np.float32), fmt="%.9g", delimiter=" ")
    if mri.bvec is not None and len(mri.bvec):
        np.savetxt(stem + ".bvecs", np.asarray(mri.bvec, np.float32), fmt="%.9g", delimiter=" ")
    err = nb != 352 + vol.size * dt.itemsize
    if err:
        print("WARNING: Problem saving " + outfile)
    return err


def write_struct(obj, basename):
    """dti_write / gqi_write / dsi_write: every MRI field -> <base>_<field>.nii.gz, every list of MRIs ->
    <base>_<field><k>.nii.gz with k = 1.. (dti.jl:344-349, gqi.jl:210-225, dsi.jl:279-294)"""
    for name, val in vars(obj).items():
        if isinstance(val, MRI):
            mri_write(val, "%s_%s.nii.gz" % (basename, name))
        elif isinstance(val, (list, tuple)) and val and all(isinstance(v, MRI) for v in val):
            for k, v in enumerate(val, 1):
                mri_write(v, "%s_%s%d.nii.gz" % (basename, name, k))


dti_write = gqi_write = dsi_write = write_struct


def rumba_write(rumba, basename):
    """rumba_write (rusd.jl:645-663): the MRI fields like the other writers, every other field (snr_mean, snr_std) as
    `<base>_<field>.txt` (writedlm of one Float32: its shortest decimal form and a newline)."""
    write_struct(rumba, basename)
    for name, val in vars(rumba).items():
        if isinstance(val, MRI) or (isinstance(val, (list, tuple)) and val and all(isinstance(v, MRI) for v in val)):
            continue
        with open("%s_%s.txt" % (basename, name), "w") as f:
            f.write(" ".join(str(np.float32(x)) for x in np.atleast_1d(val)) + "\n")


def read_struct(inbase, cls):
    """mri_read(inbase, type) (mri.jl:2276-2311): reload a DTI / GQI / DSI result from its files"""
    import dataclasses
    import typing
    absbase = os.path.abspath(inbase)
    args = {}
    hints = typing.get_type_hints(cls)
    for f in dataclasses.fields(cls):
        if hints[f.name] is MRI:
            args[f.name] = mri_read("%s_%s.nii.gz" % (absbase, f.name))
        else:
            pat = re.compile("^" + re.escape(absbase) + "_" + f.name + r"([0-9]*)\.nii\.gz$")
            files = sorted((m for m in (pat.match(p) for p in glob.glob(absbase + "_" + f.name + "*.nii.gz")) if m),
                           key=lambda m: int(m.group(1) or 0))
            args[f.name] = [mri_read(m.group(0)) for m in files]
    return cls(**args)
