"""Exchange format between this repository's fixtures and a machine that has Julia + the reference (julia/make_reference_fixtures.jl).

A case is a directory: every array is `<key>.bin` -- raw little-endian, COLUMN-MAJOR (Julia's `read!` into an `Array` of the listed
dims) -- and `meta.txt` lists them, one per line, `<key> <dtype> <ndim> <d1> <d2> ...` (dtype in float32 / float64 / int32 / int64 /
uint8), plus scalars as `<key> = <value>`.  No dependency on either side (no JSON / NPZ / ZIP reader needed in Julia)."""
import os

import numpy as np

_DT = {"float32": "<f4", "float64": "<f8", "int32": "<i4", "int64": "<i8", "uint8": "u1"}


def write_case(path, arrays, scalars=None):
    os.makedirs(path, exist_ok=True)
    lines = []
    for k, v in arrays.items():
        a = np.asarray(v)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        name = a.dtype.name
        if name not in _DT:
            raise TypeError("%s: dtype %s is not part of the exchange format" % (k, name))
        if a.ndim == 0:
            a = a.reshape(1)
        a.astype(_DT[name]).ravel(order="F").tofile(os.path.join(path, k + ".bin"))
        lines.append("%s %s %d %s" % (k, name, a.ndim, " ".join(str(d) for d in a.shape)))
    for k, v in (scalars or {}).items():
        lines.append("%s = %s" % (k, v))
    with open(os.path.join(path, "meta.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


def read_case(path):
    out = {}
    with open(os.path.join(path, "meta.txt")) as f:
        for ln in f:
            ln = ln.strip()
            if not ln:
                continue
            if " = " in ln:
                k, v = ln.split(" = ", 1)
                try:
                    out[k] = int(v)
                except ValueError:
                    try:
                        out[k] = float(v)
                    except ValueError:
                        out[k] = v
                continue
            p = ln.split()
            k, name, nd = p[0], p[1], int(p[2])
            dims = tuple(int(x) for x in p[3:3 + nd])
            a = np.fromfile(os.path.join(path, k + ".bin"), dtype=_DT[name])
            out[k] = a.reshape(dims, order="F")
    return out
