"""CPU tests (no GPU): the C-ABI library loads, exports every symbol include/fibers_hip.h declares, and
fails loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "fibers_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fibd?_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(fj):
    from fibers_jl_amd import _lib
    names = _header_functions()
    assert len(names) >= 30
    L = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, "declared in include/fibers_hip.h but not exported: %s" % missing
    assert sorted(_lib.exported_symbols()) == names, "ctypes prototype table out of sync with the header"


def test_version_and_device_count(fj):
    L = fj.lib()
    assert L.fib_version().startswith(b"fibers-hip")
    assert L.fib_device_count() >= 0


def test_no_gpu_is_a_loud_error_not_a_fallback(fj):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    bval = np.array([0, 1000, 1000, 1000, 1000, 1000, 1000], np.float32)
    with pytest.raises(fj.FibersError) as e:
        fj.DtiPlan(bval, np.eye(7, 3, dtype=np.float32))
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    dwi = fj.MRI(np.ones((2, 2, 2, 7), np.float32), bval, np.eye(7, 3, dtype=np.float32))
    with pytest.raises(fj.FibersError):
        fj.dti_fit(dwi, fj.MRI(np.ones((2, 2, 2), np.uint8)))
    with pytest.raises(fj.FibersError):
        fj.gqi_rec(dwi, fj.MRI(np.ones((2, 2, 2), np.uint8)))


def test_reference_error_strings(fj):
    dwi = fj.MRI(np.ones((2, 2, 2, 7), np.float32))
    m = fj.MRI(np.ones((2, 2, 2), np.uint8))
    with pytest.raises(RuntimeError, match="Missing b-value table from input DWI structure"):
        fj.dti_fit(dwi, m)
    with pytest.raises(RuntimeError, match="Missing b-value table from input DWI structure"):
        fj.adc_fit(dwi, m)
    dwi.bval = np.ones(7, np.float32)
    for fn in (fj.dti_fit, fj.gqi_rec, fj.dsi_rec):
        with pytest.raises(RuntimeError, match="Missing gradient table from input DWI structure"):
            fn(dwi, m)
    # the C entry points report the same conditions as status codes
    L = fj.lib()
    h = C.c_void_p()
    assert L.fib_dti_plan_create(0, None, None, 0, C.byref(h)) == -4
    assert b"Missing b-value table" in L.fib_last_error()
    v, f = fj.sphere_642.vertices, fj.sphere_642.faces
    assert L.fib_gqi_plan_create(0, dwi.bval.ctypes.data, None, 7, v.ctypes.data, 642, f.ctypes.data, 1280, 1.25, C.byref(h)) == -5
    assert b"Missing gradient table" in L.fib_last_error()


def test_sphere_tables(fj):
    for name, nv, nf in (("sphere_362", 362, 720), ("sphere_642", 642, 1280), ("sphere_724", 724, 1444)):
        s = getattr(fj, name)
        assert s.vertices.shape == (nv, 3) and s.faces.shape == (nf, 3) and s.vertices.dtype == np.float32
        assert np.array_equal(s.vertices[nv // 2:], -s.vertices[: nv // 2])
        assert s.faces.min() == 1 and s.faces.max() == nv


def test_struct_layouts_of_the_header_match_the_bindings(tmp_path):
    """The four structs that cross the C ABI by pointer have hand-written mirrors (ctypes in fibers.jl_amd/_lib.py, Julia structs in
    julia/FibersHIP.jl).  A C program compiled against include/fibers_hip.h prints sizeof and every offsetof; the ctypes mirrors
    must agree field by field, and so must the numbers quoted in FibersHIP.jl's comments."""
    import subprocess
    from fibers_jl_amd import _lib
    fields = {
        "fib_dti_out": ["s0", "eigval1", "eigval2", "eigval3", "eigvec1", "eigvec2", "eigvec3", "rd", "md", "fa"],
        "fib_rumba_out": ["fodf", "fgm", "fcsf", "gfa", "var", "peak"],
        "fib_stream_params": ["nx", "ny", "nz", "nvec", "len_min", "len_max", "cosang_thresh", "step_size", "smooth_coeff", "search_dist",
                              "search_cosang", "ws", "interp", "search_flat_axis"],
        "fib_tract_out": ["nlines", "npoints", "npts", "seed_index", "xyz", "flags"],
    }
    mirrors = {"fib_dti_out": _lib.DtiOut, "fib_rumba_out": _lib.RumbaOut, "fib_stream_params": _lib.StreamParams, "fib_tract_out": _lib.TractOut}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "fibers_hip.h"', 'int main(void) {']
    for st, fl in fields.items():
        src.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (st, st))
        for f in fl:
            src.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (st, f, st, f))
    src += ['  return 0;', '}']
    cfile = tmp_path / "layout.c"
    cfile.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(cfile), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        st, f, v = ln.split()
        got[(st, f)] = int(v)
    for st, fl in fields.items():
        m = mirrors[st]
        assert C.sizeof(m) == got[(st, "sizeof")], (st, C.sizeof(m), got[(st, "sizeof")])
        assert [n for n, _ in m._fields_] == fl, (st, [n for n, _ in m._fields_])
        for f in fl:
            assert getattr(m, f).offset == got[(st, f)], (st, f, getattr(m, f).offset, got[(st, f)])
    # the Julia mirrors state their layout in comments of the form `# layout: <struct> sizeof N: field@offset ...` -- keep them true
    jl = open(os.path.join(ROOT, "julia", "FibersHIP.jl")).read()
    for st, fl in fields.items():
        want = "# layout: %s sizeof %d: %s" % (st, got[(st, "sizeof")], " ".join("%s@%d" % (f, got[(st, f)]) for f in fl))
        assert want in jl, "julia/FibersHIP.jl lacks or misstates: " + want
