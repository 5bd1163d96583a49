#!/usr/bin/env python3
"""Generates fibers.jl_amd/csrc/sphere642_fused.inc: the register-level find_peaks! scan (gqi.jl:180-201) that the
split-bf16 contraction kernel runs on its accumulators for the default tessellation (sphere_642), so that the ODF is
never re-read from HBM for peak finding.

Why a generated layout.  After the contraction a wave holds 32 voxels x 321 ODF rows: lane l = (voxel l & 31, half
l >> 5), block m, register r holds row m*32 + (r&3) + 8*(r>>2) + 4*half; row 320 is the "extra" f32 row.  The two lane
halves execute ONE instruction stream, so a register-level neighbour test only works if the vertex held by half 1 at
(m, r) has its neighbours at the same register positions as the vertex held by half 0.  sphere_642 is an icosahedral
geodesic sphere; modulo the antipodal fold (projective plane) each of its 15 two-fold rotations g is a graph
automorphism and an involution.  Put v in half 0 and g(v) in half 1 of the same (m, r) ("slot"): then for a neighbour u
of v
  * u in half 0 at slot s      ->  g(u) is a neighbour of g(v), in half 1 at slot s: the lane's own register;
  * u = g(w) in half 1, w at s ->  g(u) = w is in half 0 at slot s: the OTHER half's register s for both halves
                                   (one cross-half move per such slot: the "foreign" slots along the cut);
  * u fixed by g               ->  the same vertex for both halves: replicated into both halves once.
g fixes 17 of the 321 vertices (its pole and the 16 on its polar line); 152 pairs fill 152 slots, 16 fixed vertices
fill the remaining 8 slots (tested redundantly by both halves from replicated values), the pole is the extra row.
The rotation and the cut of the fundamental domain are chosen to minimise the number of foreign slots (26).
The plan compares its own neighbour table with sphere642_scan.inc's before it uses this program."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_sphere_scan import table  # noqa: E402


def involution(H, adj, a):
    nv = H.shape[0]
    a = a / np.linalg.norm(a)
    G = 2 * np.outer(H @ a, a) - H
    d = np.abs(G @ H.T)
    idx = d.argmax(1)
    if d.max(1).min() < 1 - 1e-5 or len(set(idx)) != nv:
        return None
    if not all(set(idx[adj[v]]) == set(adj[idx[v]]) for v in range(nv)):
        return None
    if not all(idx[idx[v]] == v for v in range(nv)):
        return None
    return a, idx


def choose(H, adj):
    nv = H.shape[0]
    axes = []
    for i in range(nv):
        for j in adj[i]:
            if j > i:
                r = involution(H, adj, H[i] + H[j])
                if r:
                    axes.append(r)
        r = involution(H, adj, H[i])
        if r:
            axes.append(r)
    best = None
    for a, g in axes:
        X = H * np.sign(H @ a + 1e-12)[:, None]
        e1 = np.cross(a, [0.3, 0.5, 0.8]); e1 /= np.linalg.norm(e1)
        e2 = np.cross(a, e1)
        nonfixed = [v for v in range(nv) if g[v] != v]
        for th in np.linspace(0, np.pi, 721)[:-1]:
            d = np.cos(th) * e1 + np.sin(th) * e2
            s = X @ d
            if min(abs(s[v]) for v in nonfixed) < 1e-9:
                continue
            H0 = set(v for v in nonfixed if s[v] > 0)
            foreign = set()
            for v in H0:
                for u in adj[v]:
                    if u not in H0 and g[u] != u:
                        foreign.add(int(g[u]))
            if best is None or len(foreign) < best[0]:
                best = (len(foreign), a, g, s.copy(), H0)
    return best


def main():
    nv, adj = table("sphere_642")
    V = np.load(os.path.join(ROOT, "fibers.jl_amd", "data", "sphere_642_vertices.npy")).astype(np.float64)
    H = V[:nv]
    nfor, a, g, s, H0 = choose(H, adj)
    fixed = [v for v in range(nv) if g[v] == v]
    pole = int(np.abs(H @ a).argmax())
    assert pole in fixed and len(fixed) == 17 and len(H0) == 152
    line = [v for v in fixed if v != pole]
    # slot order: pairs by distance from the cut (foreign values die early), then the fixed pairs
    pairs = sorted(H0, key=lambda v: (abs(s[v]), v))
    slot_v = [[0] * 161, [0] * 161]
    for i, v in enumerate(pairs):
        slot_v[0][i] = int(v); slot_v[1][i] = int(g[v])
    for i in range(8):
        slot_v[0][152 + i] = int(line[2 * i]); slot_v[1][152 + i] = int(line[2 * i + 1])
    slot_v[0][160] = slot_v[1][160] = pole
    assert sorted(slot_v[0][:160] + slot_v[1][:160] + [pole]) == list(range(nv))
    slot_of = {}
    for h in range(2):
        for sl in range(160):
            slot_of[slot_v[h][sl]] = (sl, h)
    fix_id = {}                       # fixed vertex -> index into X(): 2i / 2i+1 = halves of slot 152+i, 16 = pole
    for i in range(8):
        fix_id[slot_v[0][152 + i]] = 2 * i; fix_id[slot_v[1][152 + i]] = 2 * i + 1
    fix_id[pole] = 16
    # position (row of the matrix image) -> vertex
    pos_v = [0] * 321
    for sl in range(160):
        m, r = divmod(sl, 16)
        for h in range(2):
            pos_v[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = slot_v[h][sl]
    pos_v[320] = pole
    assert sorted(pos_v) == list(range(nv))

    def own(sl):
        return "O(%d,%d)" % divmod(sl, 16)

    foreign_slots = []                # in order of first use
    prog = []
    for sl in range(152):
        v = slot_v[0][sl]
        ops = []
        for u in adj[v]:
            if g[u] == u:
                ops.append("X(%d)" % fix_id[u])
            elif slot_of[u][1] == 0:
                ops.append(own(slot_of[u][0]))
            else:
                fs = slot_of[u][0]
                if fs not in foreign_slots:
                    foreign_slots.append(fs)
                    prog.append("    FDEF(%d, %d, %d)" % ((len(foreign_slots) - 1,) + divmod(fs, 16)))
                ops.append("F(%d)" % foreign_slots.index(fs))
        # the same program must be right for half 1: check
        v1 = slot_v[1][sl]
        chk = set()
        for u in adj[v]:
            chk.add(int(g[u]))
        assert chk == set(adj[v1])
        ops += ["Z"] * (6 - len(ops))
        prog.append("    SLOT(%d, %d, %d, %s)" % ((sl,) + divmod(sl, 16) + (", ".join(ops),)))
    assert len(foreign_slots) == nfor
    xprog = []
    for i in range(8):
        xprog.append("    XDEF(%d, %d, %d, %d)" % ((2 * i, 2 * i + 1) + divmod(152 + i, 16)))

    def fixed_ops(v):
        owns, fixs = [], []
        for u in adj[v]:
            if g[u] == u:
                fixs.append("X(%d)" % fix_id[u])
            elif slot_of[u][1] == 0:
                assert int(g[u]) in adj[v]
                owns.append(own(slot_of[u][0]))
        assert len(owns) <= 3 and len(fixs) <= 3 and 2 * len(owns) + len(fixs) == len(adj[v])
        return ", ".join(owns + ["Z"] * (3 - len(owns)) + fixs + ["Z"] * (3 - len(fixs)))

    fprog = []
    for i in range(8):                # one line per fixed slot: half 0's vertex, half 1's vertex (tested by both halves, flagged by its own)
        va, vb = slot_v[0][152 + i], slot_v[1][152 + i]
        fprog.append("    FTEST2(%d, %d, %s, %d, %s)" % (152 + i, fix_id[va], fixed_ops(va), fix_id[vb], fixed_ops(vb)))
    fprog.append("    FPOLE(%d, %s)" % (fix_id[pole], fixed_ops(pole)))

    out = ["// generated by tools/gen_s642_fused.py from fibers.jl_amd/data/sphere_642_{vertices,faces}.npy -- do not edit",
           "// two-fold axis (%.6f, %.6f, %.6f): %d pair slots, 8 fixed-pair slots, pole = vertex %d, %d foreign slots" %
           (a[0], a[1], a[2], 152, pole, nfor),
           "#define FIB_F642_NFOREIGN %d" % nfor,
           "#define FIB_F642_POLE %d" % pole,
           "// [half][slot] -> vertex (0-based, first half-sphere); slot 160 = the extra row",
           "static const short fib_f642_slot_vertex[2][161] = {"]
    for h in range(2):
        out.append("    {" + ", ".join(str(x) for x in slot_v[h]) + "},")
    out.append("};")
    out.append("// row of the contraction's matrix image -> vertex")
    out.append("static const short fib_f642_pos_vertex[321] = {" + ", ".join(str(x) for x in pos_v) + "};")
    out.append("#define FIB_F642_POS_LIST(X) X(" + ", ".join(str(x) for x in pos_v) + ")")
    out.append("#define FIB_F642_SLOT_LIST(X) X(" + ", ".join(str(x) for x in slot_v[0] + slot_v[1]) + ")")
    out.append("// O(m,r): own accumulator; F(i): register (m,r) of the other lane half (FDEF); X(j): replicated fixed vertex; Z: no neighbour")
    out.append("#define FIB_F642_PAIRS(FDEF, SLOT) \\")
    out.append(" \\\n".join(prog))
    # the same program cut by accumulator block (16 slots each), every block self-contained: it (re)defines the foreign and the
    # replicated fixed values it uses, so that nothing but the accumulators and the flag strings lives across blocks
    # (odf_pipe_kernel runs one block per contraction stage of the NEXT work item)
    xdef_of = {}
    for i in range(8):
        xdef_of[2 * i] = xdef_of[2 * i + 1] = "    XDEF(%d, %d, %d, %d)" % ((2 * i, 2 * i + 1) + divmod(152 + i, 16))
    import re as _re
    for blk in range(10):
        lines, seenf, seenx = [], set(), set()
        for sl in range(16 * blk, min(16 * blk + 16, 152)):
            v = slot_v[0][sl]
            ops = []
            for u in adj[v]:
                if g[u] == u:
                    j = fix_id[u]
                    if j != 16 and xdef_of[j] not in seenx:
                        seenx.add(xdef_of[j]); lines.append(xdef_of[j])
                    ops.append("X(%d)" % j)
                elif slot_of[u][1] == 0:
                    ops.append(own(slot_of[u][0]))
                else:
                    fs = slot_of[u][0]
                    fi = foreign_slots.index(fs)
                    if fi not in seenf:
                        seenf.add(fi); lines.append("    FDEF(%d, %d, %d)" % ((fi,) + divmod(fs, 16)))
                    ops.append("F(%d)" % fi)
            ops += ["Z"] * (6 - len(ops))
            lines.append("    SLOT(%d, %d, %d, %s)" % ((sl,) + divmod(sl, 16) + (", ".join(ops),)))
        out.append("#define FIB_F642_BLOCK%d(FDEF, XDEF, SLOT) \\" % blk)
        out.append(" \\\n".join(lines))
    # the blocks once more, software-pipelined for odf_pipe_kernel: the cross-half fetch (FISSUE / XISSUE: one ds_bpermute) of a
    # value is emitted LOOKAHEAD slots before the slot that first uses it (or in the block's PRE list), the selection of a
    # replicated fixed value (XSEL) right before it.  Same slots, same operands, same order of the flags as FIB_F642_BLOCKn.
    LOOKAHEAD = 5
    for blk in range(10):
        slots = list(range(16 * blk, min(16 * blk + 16, 152)))
        need = []                         # per slot: (defs first used here, line)
        seenf, seenx = set(), set()
        for sl in slots:
            v = slot_v[0][sl]
            ops, defs = [], []
            for u in adj[v]:
                if g[u] == u:
                    j = fix_id[u]
                    if j != 16 and (j // 2) not in seenx:
                        seenx.add(j // 2); defs.append(("X", j // 2))
                    ops.append("X(%d)" % j)
                elif slot_of[u][1] == 0:
                    ops.append(own(slot_of[u][0]))
                else:
                    fs = slot_of[u][0]
                    fi = foreign_slots.index(fs)
                    if fi not in seenf:
                        seenf.add(fi); defs.append(("F", fi, fs))
                    ops.append("F(%d)" % fi)
            ops += ["Z"] * (6 - len(ops))
            need.append((defs, "    SLOT(%d, %d, %d, %s)" % ((sl,) + divmod(sl, 16) + (", ".join(ops),))))
        pre, body = [], [[] for _ in slots]
        for p_, (defs, line) in enumerate(need):
            for d in defs:
                if d[0] == "F":
                    txt = "    FISSUE(%d, %d, %d)" % ((d[1],) + divmod(d[2], 16))
                else:
                    txt = "    XISSUE(%d, %d, %d)" % ((d[1],) + divmod(152 + d[1], 16))
                (pre if p_ - LOOKAHEAD < 0 else body[p_ - LOOKAHEAD]).append(txt)
        lines = []
        for p_, (defs, line) in enumerate(need):
            lines += body[p_]
            for d in defs:
                if d[0] == "X":
                    lines.append("    XSEL(%d, %d, %d, %d, %d)" % ((d[1], 2 * d[1], 2 * d[1] + 1) + divmod(152 + d[1], 16)))
            lines.append(line)
        out.append("#define FIB_F642_PBLOCK%d_PRE(FISSUE, XISSUE) \\" % blk)
        out.append(" \\\n".join(pre) if pre else "    ")
        out.append("#define FIB_F642_PBLOCK%d_BODY(FISSUE, XISSUE, XSEL, SLOT) \\" % blk)
        out.append(" \\\n".join(lines))
    out.append("// fixed vertices: XDEF replicates the two vertices of a fixed slot into both halves; FTEST2 / FPOLE test them (own-pair")
    out.append("// neighbours O() x3 are combined with the other half's, then the fixed neighbours X() x3)")
    out.append("#define FIB_F642_XDEFS(XDEF) \\")
    out.append(" \\\n".join(xprog))
    out.append("#define FIB_F642_FTESTS(FTEST2, FPOLE) \\")
    out.append(" \\\n".join(fprog))
    path = os.path.join(ROOT, "fibers.jl_amd", "csrc", "sphere642_fused.inc")
    open(path, "w").write("\n".join(out) + "\n")
    print("wrote", path, "foreign", nfor, "pole", pole)


if __name__ == "__main__":
    main()
