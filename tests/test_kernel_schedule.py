"""Compiler-output guard for the contraction kernels (no GPU: hipcc cross-compiles odf.hip to gfx950 assembly, ~90 s)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_stage_loops_do_not_wait_on_the_loads_they_have_just_issued():
    """every odf_gemm3_kernel / odf_dsi2_kernel instantiation: no `s_waitcnt vmcnt` between a stage's prefetch (>= 8 loads) and the
    MFMA block that is meant to cover it (tools/check_loop_waits.py; in round 3 a conditionally consumed load made hipcc put a
    vmcnt(0) there -- same instructions otherwise, 11 % slower)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_loop_waits.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(": ok") >= 20, r.stdout[-2000:]
    # the kernel whose samples travel through LDS: its hand-counted vmcnt(2) waits follow exactly the two sample requests, and the
    # DRAIN wait sits directly in front of the first row store (tools/check_loop_waits.py check_slds)
    assert r.stdout.count("slds ok") >= 1, r.stdout[-2000:]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_tracer_point_stores_and_pack_loads_keep_their_non_temporal_hint(tmp_path):
    """stream.hip (stream.jl:660: every emitted point): the tracer's scratch stores and the pack kernel's scratch loads are NON-TEMPORAL
    -- with the default policy 1.6 GB of points go through L2 and evict the orientation field (trace 0.53 -> 0.65 ms).  The hint is easy
    to lose without a trace in the source: in round 5 a run-time `plain ? store : nontemporal_store` made hipcc merge both arms into one
    plain store.  Compile the PRODUCT flags to assembly and look: every stream_trace_kernel instantiation stores its points as whole
    lines -- 16-byte stores, three per group of four trips, in the loop and once behind it -- every one with `nt`, no 12-byte store is
    left in it, and the tile pack kernel loads with `nt`."""
    import re
    out = str(tmp_path / "stream.s")
    src = os.path.join(ROOT, "fibers.jl_amd", "csrc", "stream.hip")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "-std=c++17", "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out, src], check=True, timeout=900,
                   capture_output=True)
    text = open(out).read()
    funcs = re.split(r"\n(?=_Z\w+:)", text)
    ntrace = 0
    for f in funcs:
        name = f.split(":", 1)[0]
        body = f.split(".Lfunc_end", 1)[0]
        if "stream_trace_kernel" in name:
            st = re.findall(r"global_store_dwordx4[^\n]*", body)
            plain = [x for x in st if " nt" not in x]
            fused = "ELb1EEEv" in name                          # (the fused kernel also stores its totals and the caller's counts: two 16-byte
            assert len(st) - len(plain) >= 6, (name, len(st))   #  records at scalar base addresses, plain)
            assert len(plain) <= (2 if fused else 0) and all(re.search(r", s\[\d+:\d+\]", x) for x in plain), (name, plain[:3])
            assert not re.findall(r"global_store_dwordx3[^\n]*", body), name
            ntrace += 1
        if "stream_trace_micro_kernel" in name:                 # (one lane of the wave stores the line's point: 12 bytes, nt)
            st = re.findall(r"global_store_dwordx3[^\n]*", body)
            assert st and all(" nt" in s for s in st), (name, st[:3])
        if "stream_pack_tile_kernel" in name:
            ld = re.findall(r"global_load_dword[^\n]*", body)
            assert sum(" nt" in s for s in ld) >= 3, (name, ld[:6])
    assert ntrace >= 9, ntrace            # {1, 3, runtime} vectors x {plain, LCM, trilinear} + the wide forms
