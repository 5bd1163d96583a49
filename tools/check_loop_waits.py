#!/usr/bin/env python3
"""Guards the contraction kernels' software pipeline against the compiler: in every instantiation of odf_gemm3_kernel /
odf_dsi2_kernel the stage loop issues the NEXT stage's loads (direct-to-LDS pieces + the lanes' samples) and then runs the
MFMA block; an `s_waitcnt vmcnt(N)` that hipcc puts between those loads and the first MFMA of the block waits for the loads
just issued and serialises memory latency with the matrix cores (round 3: a register-allocation change did exactly that and
cost 11 % with an otherwise identical instruction stream).

Compiles odf.hip to gfx950 assembly (no GPU needed) and, per kernel, lists every vmcnt wait that sits between a VMEM load and
the next MFMA with no barrier in between, after at least a stage's worth of loads (8).  Exit code 1 if any is found.

  python tools/check_loop_waits.py [path/to/odf.hip] [-D...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm(src, defs):
    if src.endswith(".s"):
        return open(src).read()
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out,
           "-I", os.path.dirname(src), src] + defs
    subprocess.run(cmd, check=True)
    text = open(out).read()
    os.unlink(out)
    return text


def kernels(text):
    """name -> instruction lines of every contraction kernel"""
    res, name, body = {}, None, []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w*(odf_gemm3_kernel|odf_dsi2_kernel)\w*):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            res[name] = body
            name = None
            continue
        s = line.split(";")[0].strip()
        if s and not s.startswith("."):
            body.append(s)
        elif s.startswith(".LBB"):
            body.append(s)
    return res


MIN_LOADS = 8      # a stage issues >= 8 sample loads + its pieces; fewer = the prologue's own waits


def check(body):
    """vmcnt waits with an un-waited VMEM load before them and an MFMA after them, in layout order with no barrier in between"""
    bad = []
    loads_since_sync = 0
    pending_wait = None
    for i, ins in enumerate(body):
        op = ins.split()[0]
        if op in ("s_barrier", "s_endpgm"):
            loads_since_sync, pending_wait = 0, None
            continue
        if ins.startswith(".LBB") or op.startswith("s_cbranch") or op == "s_branch":
            continue                                     # (the short skips around s_setprio / a dead stage's loads stay inside the run)
        if re.match(r"(buffer_load|global_load)", op):
            loads_since_sync += 1
            continue
        if op == "s_waitcnt" and "vmcnt" in ins:
            n = int(re.search(r"vmcnt\((\d+)\)", ins).group(1))
            if loads_since_sync > n and loads_since_sync >= MIN_LOADS:
                pending_wait = (i, ins, loads_since_sync)
            continue
        if op.startswith("v_mfma") and pending_wait is not None:
            bad.append(pending_wait)
            pending_wait = None
            loads_since_sync = 0
    return bad


def demangled(name):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        return name


def main():
    args = sys.argv[1:]
    defs = [a for a in args if a.startswith("-D")]
    srcs = [a for a in args if not a.startswith("-D")]
    src = srcs[0] if srcs else os.path.join(ROOT, "fibers.jl_amd", "csrc", "odf.hip")
    ks = kernels(device_asm(src, defs))
    nbad = 0
    for name, body in sorted(ks.items()):
        bad = check(body)
        nm = sum(1 for x in body if x.startswith("v_mfma"))
        short = re.sub(r"\(anonymous namespace\)::|\(\(anonymous namespace\)::GemmArgs\)", "", demangled(name))
        print("%-60s %5d instructions, %3d MFMAs: %s" % (short, len(body), nm, "ok" if not bad else "%d WAITS ON FRESH LOADS" % len(bad)))
        for i, ins, nl in bad:
            print("      line %d: %s  (after %d loads issued since the last barrier / branch)" % (i, ins, nl))
        nbad += len(bad)
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
