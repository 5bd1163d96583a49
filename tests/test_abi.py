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


# ---- the three statements of every prototype -- include/fibers_hip.h, the ctypes table, the Julia ccalls -- must agree ---------------
def _c_class(t):
    """C parameter / return type -> class: ptr, i32, i64, u64, f32, f64, void"""
    t = re.sub(r"\b(const|struct|volatile|restrict)\b", " ", t).strip()
    if "*" in t or "[" in t:
        return "ptr"
    t = " ".join(t.split())
    base = t.split(" ")[0] if t.split(" ")[0] not in ("unsigned",) else t
    table = {"void": "void", "int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "float": "f32", "double": "f64",
             "uint32_t": "u32", "size_t": "u64"}
    assert base in table, "unclassified C type %r" % t
    return table[base]


def _header_prototypes():
    src = open(os.path.join(ROOT, "include", "fibers_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"(?:^|[;}\n])\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*\b(fibd?_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", src):
        ret, name, params = m.group(1), m.group(2), m.group(3).strip()
        args = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                if "*" in p or "[" in p:
                    args.append("ptr")
                else:
                    args.append(_c_class(" ".join(p.split()[:-1])))          # drop the parameter name
        protos[name] = (_c_class(ret), args)
    return protos


def _ctypes_class(t):
    if t is None:
        return "void"
    if t in (C.c_void_p, C.c_char_p) or isinstance(t, type) and (issubclass(t, C._Pointer) or issubclass(t, C.Array)):
        return "ptr"
    return {C.c_int: "i32", C.c_int32: "i32", C.c_int64: "i64", C.c_uint64: "u64", C.c_float: "f32", C.c_double: "f64"}[t]


def _julia_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring",):
        return "ptr"
    return {"Cint": "i32", "Int32": "i32", "Int64": "i64", "UInt64": "u64", "Cfloat": "f32", "Float32": "f32", "Cdouble": "f64", "Cvoid": "void"}[t]


def _split_top(s):
    """split on commas that are not inside braces / parentheses"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out if x.strip()]


def _julia_ccalls(path):
    """every `ccall((:name, libfibers), Ret, (T...), args...)` of a Julia file -> [(name, ret class, [arg classes], number of values passed)]"""
    src = open(path).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*libfibers\),\s*(\w+),\s*\(", src):
        i, depth = m.end(), 1
        while depth:                                                   # the argument-type tuple
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        types = _split_top(src[m.end():i - 1])
        j, depth = i, 1                                                # the rest of the ccall: the values
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        vals = _split_top(src[i:j - 1].lstrip(", \n"))
        calls.append((m.group(1), _julia_class(m.group(2)), [_julia_class(t) for t in types], len(vals)))
    return calls


def test_ctypes_prototypes_match_the_header():
    """name, arity, and the class of the return value and of every argument (pointer / 32- / 64-bit integer / float)"""
    from fibers_jl_amd import _lib
    hdr = _header_prototypes()
    assert sorted(hdr) == sorted(_lib._PROTOS)
    for name, (res, args) in _lib._PROTOS.items():
        got = (_ctypes_class(res), [_ctypes_class(a) for a in args])
        assert got == hdr[name], "%s: ctypes %s, header %s" % (name, got, hdr[name])


def test_julia_ccalls_match_the_header_and_ctypes():
    """julia/FibersHIP.jl cannot be executed here (no Julia in the image), so its ccalls are parsed: every call names an exported
    function and states the header's return class, arity and per-argument class; as many values are passed as types are declared.
    Dropping or retyping an argument on either side fails this test."""
    from fibers_jl_amd import _lib
    hdr = _header_prototypes()
    calls = _julia_ccalls(os.path.join(ROOT, "julia", "FibersHIP.jl"))
    assert len(calls) >= 14
    seen = set()
    for name, ret, args, nvals in calls:
        assert name in hdr, "FibersHIP.jl calls %s, which include/fibers_hip.h does not declare" % name
        assert (ret, args) == hdr[name], "%s: Julia ccall %s, header %s" % (name, (ret, args), hdr[name])
        res_c, args_c = _lib._PROTOS[name]
        assert (ret, args) == (_ctypes_class(res_c), [_ctypes_class(a) for a in args_c])
        assert nvals == len(args), "%s: %d argument types, %d values" % (name, len(args), nvals)
        seen.add(name)
    # the reference's surface is bound
    for need in ("fib_init", "fib_dti_fit", "fib_adc_fit", "fib_gqi_rec", "fib_dsi_rec", "fib_find_peaks_work", "fib_stream", "fib_tract_free", "fib_last_error"):
        assert need in seen, need


def test_the_parser_notices_a_dropped_or_retyped_argument(tmp_path):
    """the check above is only worth something if it can fail"""
    hdr = _header_prototypes()
    src = open(os.path.join(ROOT, "julia", "FibersHIP.jl")).read()
    bad1 = tmp_path / "dropped.jl"
    bad1.write_text(src.replace("(Cint, Ptr{Float32}, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32}, Ref{FibDtiOut})",
                                "(Cint, Ptr{Float32}, Cint, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32}, Ref{FibDtiOut})"))
    got = {n: (r, a) for n, r, a, _ in _julia_ccalls(str(bad1))}
    assert got["fib_dti_fit"] != hdr["fib_dti_fit"]
    bad2 = tmp_path / "retyped.jl"
    bad2.write_text(src.replace("Ptr{Int32}, Cint, Cfloat, Ptr{Float32}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}})", "Ptr{Int32}, Cint, Cdouble, Ptr{Float32}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}})"))
    got = {n: (r, a) for n, r, a, _ in _julia_ccalls(str(bad2))}
    assert got["fib_gqi_rec"] != hdr["fib_gqi_rec"]


def test_reference_fixture_script_only_uses_names_the_reference_defines():
    """julia/make_reference_fixtures.jl calls into the reference as `F.<name>`; every such name must be defined by the reference's
    sources (function / struct / const).  Runs in the build container only: /root/reference does not exist on the GPU box."""
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        pytest.skip("the reference checkout is not present")
    used = set(re.findall(r"\bF\.([A-Za-z_][A-Za-z0-9_!]*)", open(os.path.join(ROOT, "julia", "make_reference_fixtures.jl")).read()))
    assert len(used) >= 10
    text = "\n".join(open(os.path.join(ref, f)).read() for f in os.listdir(ref) if f.endswith(".jl"))
    for name in sorted(used):
        n = re.escape(name)
        e = r"(?![A-Za-z0-9_!])"                                       # (a name may end in `!`: no \b there)
        defined = re.search(r"(?m)^\s*(?:function\s+%s%s|%s\s*\(.*\)\s*=|(?:mutable\s+)?struct\s+%s%s|const\s+(?:global\s+)?%s%s|%s\s*=)" % (n, e, n, n, e, n, e, n), text)
        assert defined, "make_reference_fixtures.jl uses F.%s, which the reference does not define" % name
    # and the drop-in module shadows exactly functions the reference exports
    exports = set(re.findall(r"[A-Za-z_][A-Za-z0-9_!]*", " ".join(re.findall(r"(?ms)^export\s+(.*?)(?=^\S|\Z)", text))))
    for fn in ("dti_fit", "adc_fit", "gqi_rec", "dsi_rec", "stream"):
        assert fn in exports, fn
