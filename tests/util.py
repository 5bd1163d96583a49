"""shared comparison helpers for the parity tests (tolerances from SURVEY.md §8d)."""
import numpy as np


def solver_r(l1, l2, l3):
    """the closed-form 3x3 solver's r = det((A - q I) / p) / 2 (StaticArrays eigen, dti.jl:311), recomputed in Float64 from a tensor's
    eigenvalues: q = mean, p = sqrt(sum((l - q)^2) / 6), r = prod(l - q) / p^3 / 2; |r| -> 1 where two eigenvalues coincide (p == 0: 1)"""
    l = np.stack([np.asarray(x, np.float64) for x in (l1, l2, l3)])
    q = l.mean(0)
    p = np.sqrt(((l - q) ** 2).sum(0) / 6.0)
    with np.errstate(all="ignore"):
        r = np.prod(l - q, axis=0) / (p ** 3) / 2.0
    return np.where(p > 0, r, 1.0)


def assert_dti_close(got, ref, mask, label="", s0_rtol=1e-4, ev_atol=1e-7, ev_rtol=1e-4, fa_atol=1e-4,
                     vec_tol=1e-4, gap=1e-2, trig_rel=6e-4, trig_r=1e-3):
    """got/ref: dicts of arrays [nx,ny,nz(,3)].

    Eigenvalue tolerance: abs 1e-7 + rel 1e-4 -- PLUS trig_rel*|eigval1| ONLY in the voxels where the reference's
    closed-form solver (StaticArrays, dti.jl:311) is ill-conditioned: it takes acos(r), and for |r| > 1 - trig_r (prolate / oblate
    tensors: two eigenvalues nearly equal) a 1-ulp change of r moves the two near-degenerate eigenvalues by up to
    ~sqrt(eps32)*p ~ 3.5e-4*|eigval1| -- the reference algorithm's own conditioning (two libm's give two answers), not a kernel
    error.  r is recomputed here from the oracle's eigenvalues (solver_r); everywhere else SURVEY's tolerance applies as written."""
    m = np.asarray(mask).astype(bool)
    if m.ndim == 4:
        m = m[..., 0]
    with np.errstate(all="ignore"):
        illc = np.abs(solver_r(*(np.asarray(ref[k]).reshape(m.shape) for k in ("eigval1", "eigval2", "eigval3")))) > 1.0 - trig_r
    illc = illc | ~np.isfinite(np.asarray(ref["eigval1"]).reshape(m.shape))
    for k in ("s0", "eigval1", "eigval2", "eigval3", "rd", "md", "fa"):
        g, r = np.asarray(got[k]).reshape(m.shape), np.asarray(ref[k]).reshape(m.shape)
        assert np.array_equal(np.isnan(g), np.isnan(r)), "%s %s: NaN pattern differs" % (label, k)
        ok = ~np.isnan(r)
        if k == "s0":
            err = np.abs(g - r)[ok] <= s0_rtol * np.abs(r)[ok] + 1e-30
        elif k == "fa":
            err = np.abs(g - r)[ok] <= fa_atol
        else:
            lam1 = np.abs(np.asarray(ref["eigval1"]).reshape(m.shape))
            err = np.abs(g - r)[ok] <= ev_atol + ev_rtol * np.abs(r)[ok] + np.where(illc, trig_rel, 0.0)[ok] * lam1[ok]
        assert err.all(), "%s %s: %d voxels out of tolerance, max abs err %g" % (
            label, k, (~err).sum(), np.abs(g - r)[ok].max())
        assert (g[~m] == 0).all(), "%s %s: non-zero outside the mask" % (label, k)
    l1, l2, l3 = (np.asarray(ref[k]).reshape(m.shape) for k in ("eigval1", "eigval2", "eigval3"))
    with np.errstate(all="ignore"):
        sep1 = m & ((l1 - l2) > gap * np.abs(l1))
        sep3 = m & ((l2 - l3) > gap * np.abs(l1))
    for k, sel in (("eigvec1", sep1), ("eigvec2", sep1 & sep3), ("eigvec3", sep3)):
        g, r = np.asarray(got[k]).reshape(m.shape + (3,)), np.asarray(ref[k]).reshape(m.shape + (3,))
        dots = np.abs((g * r).sum(-1))[sel]
        if dots.size:
            assert dots.min() >= 1 - vec_tol, "%s %s: min |dot| = %g" % (label, k, dots.min())
        assert (g[~m] == 0).all()


def _half_sphere_neighbours(faces, nvert):
    """neighbour lists of the folded tessellation: faces[faces .> nvert] .-= nvert (gqi.jl:63-64); faces 1-based [nfaces, 3]"""
    f = np.array(faces, dtype=np.int64, copy=True)
    f[f > nvert] -= nvert
    f -= 1
    nb = [set() for _ in range(nvert)]
    for a, b, c in f:
        nb[a].update((b, c)); nb[b].update((a, c)); nb[c].update((a, b))
    return [np.array(sorted(x - {v}), dtype=np.int64) for v, x in enumerate(nb)]


def _lists_match_up_to_ties(col, want, got, slack):
    """two ranked vertex lists (-1 = no peak) agree if at every rank the amplitudes agree within slack (0 for "no peak")"""
    for w, g in zip(want, got):
        if w == g:
            continue
        aw = float(col[w]) if w >= 0 else 0.0
        ag = float(col[g]) if g >= 0 else 0.0
        if abs(aw - ag) > slack:
            return False
    return True


def peak_mismatches_are_ties(ref_odf, ref_peaks, got_peaks, verts_half, tol=1e-4, faces=None):
    """peak vertices are identical "where the deciding amplitude margin > 1e-4 * max".  A voxel whose peak vectors
    differ must be explained by margins of at most tol * the voxel's ODF maximum IN THE ORACLE'S ODF -- either
      * the amplitudes of the two vertices at the differing rank are that close (a tie in the ranking), or
      * (faces given) the peak test itself is that close for some vertices: find_peaks! keeps a vertex that is > all its neighbours
        and > 0 (gqi.jl:185-200); vertices whose margin over their largest neighbour (or over 0) is within tol may flip, and the
        kernel's list must be the top three of the certain peaks plus SOME subset of the marginal ones, again up to ranking ties.
    ref_odf [..., nvert]; ref_peaks / got_peaks: three arrays [..., 3]; verts_half [nvert, 3].  Returns the mismatch count."""
    import itertools
    vh = np.ascontiguousarray(verts_half, np.float32)
    nvert = vh.shape[0]
    index = {vh[i].tobytes(): i for i in range(nvert)}
    odf = np.asarray(ref_odf).reshape(-1, nvert)
    rp = [np.ascontiguousarray(np.asarray(ref_peaks[k], np.float32).reshape(-1, 3)) for k in range(3)]
    gp = [np.ascontiguousarray(np.asarray(got_peaks[k], np.float32).reshape(-1, 3)) for k in range(3)]
    differ = np.zeros(odf.shape[0], bool)
    for k in range(3):
        differ |= ~np.all(rp[k] == gp[k], axis=1)
    nbrs = _half_sphere_neighbours(faces, nvert) if faces is not None else None

    def vertex_of(vec, i, k):
        if not vec.any():
            return -1                                             # no peak at this rank
        assert vec.tobytes() in index, "peak %d of voxel %d is not a vertex of the tessellation" % (k, i)
        return index[vec.tobytes()]

    nbad = 0
    for i in np.flatnonzero(differ):
        nbad += 1
        col = odf[i].astype(np.float64)
        top = float(np.abs(col).max())
        slack = tol * top
        want = [vertex_of(rp[k][i], i, k) for k in range(3)]
        got = [vertex_of(gp[k][i], i, k) for k in range(3)]
        if _lists_match_up_to_ties(col, want, got, slack):
            continue
        msg = "voxel %d: peaks %s (oracle) vs %s differ by more than a tie of %.3g of the ODF maximum" % (i, want, got, tol)
        assert nbrs is not None, msg
        margin = np.array([min(col[v] - col[nbrs[v]].max(), col[v]) for v in range(nvert)])   # > 0: a peak (strictly above neighbours and 0)
        certain = [v for v in range(nvert) if margin[v] > slack]
        marginal = [v for v in range(nvert) if abs(margin[v]) <= slack]
        assert len(marginal) <= 12, msg + " (and %d vertices with a marginal peak test)" % len(marginal)
        ok = False
        for n in range(len(marginal) + 1):
            for sub in itertools.combinations(marginal, n):
                cand = sorted(certain + list(sub), key=lambda v: (-col[v], v))[:3]
                cand += [-1] * (3 - len(cand))
                if _lists_match_up_to_ties(col, cand, got, slack):
                    ok = True
                    break
            if ok:
                break
        assert ok, msg + "; marginal vertices %s do not explain it" % marginal
    return nbad
