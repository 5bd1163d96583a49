"""A NIfTI-1 assembler and parser written from the NIfTI-1 standard's field table (nifti1.h: byte offsets below), sharing no code
with fibers.jl_amd/nifti.py: the independent side of the N2 checks (tests/test_formats.py) and the generator of the committed byte
fixture tests/golden/nifti/hand_be_int16.nii (python tests/nifti_independent.py writes it)."""
import os
import struct

import numpy as np

# (offset, struct code, count) of the fields used, from the standard's header table
OFF = dict(sizeof_hdr=(0, "i", 1), dim_info=(39, "B", 1), dim=(40, "h", 8), intent_p=(56, "f", 3), intent_code=(68, "h", 1),
           datatype=(70, "h", 1), bitpix=(72, "h", 1), slice_start=(74, "h", 1), pixdim=(76, "f", 8), vox_offset=(108, "f", 1),
           scl_slope=(112, "f", 1), scl_inter=(116, "f", 1), slice_end=(120, "h", 1), slice_code=(122, "b", 1), xyzt_units=(123, "B", 1),
           cal_max=(124, "f", 1), cal_min=(128, "f", 1), qform_code=(252, "h", 1), sform_code=(254, "h", 1),
           quatern=(256, "f", 3), qoffset=(268, "f", 3), srow_x=(280, "f", 4), srow_y=(296, "f", 4), srow_z=(312, "f", 4))
DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4"}


def quatern_to_affine(b, c, d, qx, qy, qz, dx, dy, dz, qfac):
    """the standard's "METHOD 2": rotation matrix of the unit quaternion (a, b, c, d), a = sqrt(1 - b^2 - c^2 - d^2) >= 0,
    columns scaled by the voxel sizes, the third negated when qfac = -1"""
    a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]], np.float64)
    M = np.eye(4)
    M[:3, :3] = R * np.array([dx, dy, dz * (-1.0 if qfac < 0 else 1.0)])
    M[:3, 3] = [qx, qy, qz]
    return M


def assemble(vol, code, endian="<", pixdim=(1.0, 1.0, 1.0, 1.0, 0.0), quatern=(0.0, 0.0, 0.0), qoffset=(0.0, 0.0, 0.0), qfac=1.0,
             qform_code=1, sform_code=0, srow=None, scl_slope=0.0, scl_inter=0.0, xyzt_units=2 | 8):
    """348-byte header + 4 pad bytes + the voxels in Fortran order, every field packed at its offset"""
    h = bytearray(352)

    def put(name, *vals):
        o, c, n = OFF[name]
        struct.pack_into(endian + c * n, h, o, *vals)
    shape = list(vol.shape) + [1] * (7 - vol.ndim)
    put("sizeof_hdr", 348)
    put("dim", vol.ndim, *shape)
    put("datatype", code)
    put("bitpix", np.dtype(DTYPES[code]).itemsize * 8)
    put("pixdim", qfac, *pixdim, 0.0, 0.0)
    put("vox_offset", 352.0)
    put("scl_slope", scl_slope)
    put("scl_inter", scl_inter)
    put("xyzt_units", xyzt_units)
    put("qform_code", qform_code)
    put("sform_code", sform_code)
    put("quatern", *quatern)
    put("qoffset", *qoffset)
    if srow is not None:
        for k, name in enumerate(("srow_x", "srow_y", "srow_z")):
            put(name, *[float(v) for v in srow[k]])
    h[344:348] = b"n+1\0"
    return bytes(h) + np.asarray(vol).astype(endian + DTYPES[code]).tobytes(order="F")


def parse(raw):
    """header fields by offset + the voxel array (file byte order -> native), nothing scaled or converted"""
    endian = "<" if struct.unpack_from("<i", raw, 0)[0] == 348 else ">"
    assert struct.unpack_from(endian + "i", raw, 0)[0] == 348, "not a NIfTI-1 header"
    out = {"endian": endian, "magic": bytes(raw[344:348])}
    for name, (o, c, n) in OFF.items():
        v = struct.unpack_from(endian + c * n, raw, o)
        out[name] = v[0] if n == 1 else list(v)
    nd = out["dim"][0]
    shape = out["dim"][1:1 + nd]
    dt = np.dtype(endian + DTYPES[out["datatype"]])
    off = int(out["vox_offset"])
    out["data"] = np.frombuffer(raw, dt, count=int(np.prod(shape)), offset=off).reshape(shape, order="F").astype(dt.newbyteorder("="))
    out["nbytes_expected"] = off + int(np.prod(shape)) * dt.itemsize
    return out


# ---- the committed fixture: big-endian int16, scaled, a rotated qform, mm + seconds ------------------------------------------------
FIX_SHAPE = (4, 3, 2, 5)
FIX_QUAT = (0.1, -0.2, 0.3)
FIX_QOFF = (-12.5, 30.0, 7.25)
FIX_PIX = (2.0, 2.5, 3.0, 1.75)                       # dx, dy, dz (mm), TR (s)
FIX_SLOPE, FIX_INTER = 2.0, -3.0


def fixture_raw_values():
    i, j, k, t = np.meshgrid(*[np.arange(n) for n in FIX_SHAPE], indexing="ij")
    return (100 * t + 10 * k + 3 * j + i - 7).astype(np.int16)


def write_fixture(dirname):
    os.makedirs(dirname, exist_ok=True)
    raw = assemble(fixture_raw_values(), 4, endian=">", pixdim=FIX_PIX + (0.0,), quatern=FIX_QUAT, qoffset=FIX_QOFF, qfac=-1.0,
                   qform_code=1, sform_code=0, scl_slope=FIX_SLOPE, scl_inter=FIX_INTER, xyzt_units=2 | 8)
    with open(os.path.join(dirname, "hand_be_int16.nii"), "wb") as f:
        f.write(raw)
    # b-table in the "FSL" layout: one ROW of b-values, three ROWS of gradient components (un-normalised on purpose)
    with open(os.path.join(dirname, "hand_be_int16.bval"), "w") as f:
        f.write("0 1000 1000 2000 3000\n")
    with open(os.path.join(dirname, "hand_be_int16.bvec"), "w") as f:
        f.write("0 2 0 1 -3\n0 0 3 1 0\n0 0 0 1 4\n")


if __name__ == "__main__":
    write_fixture(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nifti"))
    print("wrote tests/golden/nifti/hand_be_int16.{nii,bval,bvec}")
