"""The host tier's pure host code (fibers.jl_amd/csrc/host_tier.h: mask element types, live maps and piece lists, chunk schedule, slabs,
and the three-stage ring pipeline fibh::run_chunks) built WITHOUT HIP under AddressSanitizer + UBSan and under ThreadSanitizer
(tests/host_tier_check.cpp: a device back end made of threads and memcpy).  Every advisor round found a host-tier ordering bug by
reading; here a tool reads.  The third build is a mutant (the scatter stage does not wait for its download): the harness must fail on
it, or it proves nothing.  Reference behaviour these helpers serve: caller-owned zero-filled outputs (mri.jl:249-265), the mask tests
dti.jl:261 / gqi.jl:134 / stream.jl:102."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_tier_check.cpp")


def _build(tmp_path, name, flags):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread"] + flags + [SRC, "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:]
    return exe


def _run(exe):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    # a sanitizer runtime that cannot start on this kernel (address-space layout it does not know) says nothing about the code under test
    if "unexpected memory mapping" in out.stderr or "Shadow memory range interleaves" in out.stderr:
        pytest.skip("the sanitizer runtime cannot start here: " + out.stderr.strip().splitlines()[0][:200])
    return out


def test_host_tier_under_asan_ubsan(tmp_path):
    out = _run(_build(tmp_path, "htc_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]))
    assert out.returncode == 0 and "host_tier_check ok" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr


def test_host_tier_ring_under_tsan(tmp_path):
    out = _run(_build(tmp_path, "htc_tsan", ["-fsanitize=thread"]))
    assert out.returncode == 0 and "host_tier_check ok" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
    assert "ThreadSanitizer" not in out.stderr


def test_the_harness_catches_a_missing_wait(tmp_path):
    """mutant: host_wait(E_OUT) returns at once -> the scatter stage reads the pinned buffer while the download writes it"""
    out = _run(_build(tmp_path, "htc_mutant", ["-fsanitize=thread", "-DFIBH_MUTATE=1"]))
    assert out.returncode != 0
    assert "ThreadSanitizer: data race" in out.stderr or "pipeline mismatch" in out.stderr


def test_plan_table_builders_under_asan_ubsan_match_the_oracle(tmp_path, orc):
    """csrc/setup.cpp (pinv / SVD, DTI design, GQI matrix, DSI dense maps, face folding) built with g++ -fsanitize=address,undefined and
    compared with the oracle's work structs: DTIwork dti.jl:101-155, ADCwork dti.jl:39-84, GQIwork gqi.jl:32-82, DSIwork dsi.jl:41-143"""
    import sys
    import numpy as np
    sys.path.insert(0, ROOT)
    import fibers_jl_amd as fj
    from fibers_jl_amd import phantom
    if not os.path.isdir("/opt/rocm/include/hip"):
        pytest.skip("no HIP headers")
    exe = str(tmp_path / "setup_check")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "fibers.jl_amd", "csrc"),
           os.path.join(ROOT, "tests", "setup_check.cpp"), "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    sph = fj.sphere_642
    V = np.asfortranarray(sph.vertices, np.float32)
    F = np.asfortranarray(sph.faces, np.int32)
    db, dg = phantom.scheme_dti(30, 3, 1000.0, seed=2)
    gb, gg = phantom.scheme_gqi(3, 20, (1000.0, 2000.0, 3000.0), 3)
    sb, sg = phantom.scheme_dsi()
    d = str(tmp_path)
    for name, arr in (("verts", V), ("faces", F), ("dti_bval", db), ("dti_bvec", np.asfortranarray(dg, np.float32)), ("gqi_bval", gb),
                      ("gqi_bvec", np.asfortranarray(gg, np.float32)), ("dsi_bval", sb), ("dsi_bvec", np.asfortranarray(sg, np.float32))):
        np.asarray(arr).ravel(order="F").tofile(os.path.join(d, name))
    run = subprocess.run([exe, d], capture_output=True, text=True, timeout=600,
                                                 env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"))
    assert run.returncode == 0 and "setup_check ok" in run.stdout, (run.stdout[-2000:], run.stderr[-4000:])
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
    rd = lambda name, dt=np.float32: np.fromfile(os.path.join(d, name), dt)      # noqa: E731
    W = orc.dti_work(db, dg)
    n = len(db)
    assert np.array_equal(rd("dti_A").reshape(n, 7, order="F"), W["A"])
    pA = rd("dti_pA").reshape(7, n, order="F")
    assert np.abs(pA - W["pA"]).max() <= 2e-5 * np.abs(W["pA"]).max()              # two SVD routines (float64 Jacobi here, LAPACK sgesdd in the oracle)
    Wa = orc.adc_work(db)
    assert np.abs(rd("adc_pA").reshape(2, n, order="F") - Wa["pA"]).max() <= 2e-5 * np.abs(Wa["pA"]).max()
    G = orc.gqi_work(gb, gg, V, F, 1.25)
    assert np.abs(rd("gqi_A").reshape(sph.nvert, len(gb), order="F") - G["A"]).max() <= 2e-6       # (the dot product's summation order: sequential here, BLAS in NumPy)
    nbr = rd("nbr", np.int32)
    md = int(nbr[-1])
    nbr = nbr[:-1].reshape(sph.nvert, md)
    ff = G["faces"]                                                                 # folded, 0-based
    for v in (0, 1, 77, 320):
        want = set()
        for a, b, c in ff[(ff == v).any(axis=1)]:
            want |= {int(a), int(b), int(c)}
        if not any((row == v).sum() > 1 for row in ff[(ff == v).any(axis=1)]):
            want.discard(v)
        assert set(int(x) for x in nbr[v] if x >= 0) == want
    D = orc.dsi_work(sb, sg, V, F, 32)
    meta = rd("dsi_meta")
    ns = len(sb)
    assert np.array_equal(meta[2:].reshape(ns, 3).astype(np.int32), D["iq"])
    zero = np.flatnonzero((D["iq"] == 0).all(axis=1))
    assert int(meta[0]) == int(zero[-1]) and meta[1] == np.float32(16 ** 3 * D["H"][D["iq_ind"][zero[-1]]])
    A = rd("dsi_A").reshape(ns + sph.nvert, ns, order="F")
    # pdf rows: H_j cos(2 pi iq_i . iq_j / 16) for the effective frames (dsi.jl:212-227)
    j = int(np.flatnonzero(np.abs(D["iq"]).sum(axis=1) == 3)[0])
    want = D["H"][D["iq_ind"][j]] * np.cos(2 * np.pi * (D["iq"].astype(np.float64) @ D["iq"][j].astype(np.float64)) / 16.0)
    assert np.abs(A[:ns, j] - want).max() <= 1e-6
