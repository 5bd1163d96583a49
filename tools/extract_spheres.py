#!/usr/bin/env python3
"""Extract the ODF tessellation tables (numbers only) from the reference.

The three spheres of the reference (`src/odf.jl:14-1100` sphere_362,
`:1104-3030` sphere_642, `:3034-5206` sphere_724) are pure data: rows of
decimal vertex coordinates converted with ``Float32.(...)`` and rows of 1-based
integer face indices.  This script parses the numeric rows and stores them as
``.npy`` arrays (float32 [nverts,3], int32 [nfaces,3], faces kept 1-based as in
the reference) under ``fibers.jl_amd/data``.  Run once in the build container
(where /root/reference exists); the GPU box only ever sees the .npy files.
"""
import os
import re
import sys
import numpy as np

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/odf.jl"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fibers.jl_amd", "data")

num = r"[-+]?\d+\.\d+(?:[eE][-+]?\d+)?"
vert_re = re.compile(rf"^\s*({num})\s+({num})\s+({num})\s*$")
face_re = re.compile(r"^\s*(\d+)\s+(\d+)\s+(\d+)\s*$")
name_re = re.compile(r"const\s+global\s+(sphere_\d+)\s*=\s*ODF\(")

tables = {}
cur = None
with open(SRC) as fh:
    for line in fh:
        m = name_re.search(line)
        if m:
            cur = m.group(1)
            tables[cur] = ([], [])
            continue
        if cur is None:
            continue
        m = vert_re.match(line)
        if m:
            tables[cur][0].append([float(m.group(i)) for i in (1, 2, 3)])
            continue
        m = face_re.match(line)
        if m:
            tables[cur][1].append([int(m.group(i)) for i in (1, 2, 3)])

os.makedirs(OUT, exist_ok=True)
for name, (v, f) in tables.items():
    v = np.asarray(v, dtype=np.float64).astype(np.float32)   # Float32.(...) odf.jl:15
    f = np.asarray(f, dtype=np.int32)
    n = v.shape[0]
    assert n % 2 == 0 and f.min() == 1 and f.max() == n, (name, n, f.min(), f.max())
    assert np.array_equal(v[n // 2:], -v[: n // 2]), name      # antipodal pairing
    np.save(os.path.join(OUT, f"{name}_vertices.npy"), v)
    np.save(os.path.join(OUT, f"{name}_faces.npy"), f)
    print(name, v.shape, f.shape)
