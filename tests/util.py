"""shared comparison helpers for the parity tests (tolerances from SURVEY.md §8d)."""
import numpy as np


def assert_dti_close(got, ref, mask, label="", s0_rtol=1e-4, ev_atol=1e-7, ev_rtol=1e-4, fa_atol=1e-4,
                     vec_tol=1e-4, gap=1e-2, trig_rel=6e-4):
    """got/ref: dicts of arrays [nx,ny,nz(,3)].

    Eigenvalue tolerance: abs 1e-7 + rel 1e-4 PLUS trig_rel*|eigval1|: the reference's
    closed-form solver (StaticArrays, dti.jl:311) takes acos(r) with r -> +-1 for prolate/oblate tensors,
    so a 1-ulp change of r moves the two near-degenerate eigenvalues by ~sqrt(eps32)*p ~ 3.5e-4*|eigval1|.
    That is the reference algorithm's own conditioning (two libm's give two answers), not a kernel error."""
    m = np.asarray(mask).astype(bool)
    if m.ndim == 4:
        m = m[..., 0]
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
            err = np.abs(g - r)[ok] <= ev_atol + ev_rtol * np.abs(r)[ok] + trig_rel * lam1[ok]
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


def peak_mismatches_are_ties(ref_odf, ref_peaks, got_peaks, verts_half, tol=1e-4):
    """peak vertices are identical "where the deciding amplitude margin > 1e-4 * max".  For every voxel and rank at
    which the two peak vectors differ, the ORACLE's amplitudes of the two vertices involved (0 for "no peak at this rank") must
    be within tol * the voxel's ODF maximum of each other: a mismatch has to be a rounding-level tie, not an error.
    ref_odf [..., nvert]; ref_peaks / got_peaks: three arrays [..., 3]; verts_half [nvert, 3].  Returns the mismatch count."""
    vh = np.ascontiguousarray(verts_half, np.float32)
    index = {vh[i].tobytes(): i for i in range(vh.shape[0])}
    odf = np.asarray(ref_odf).reshape(-1, vh.shape[0])
    nbad = 0
    for k in range(3):
        rp = np.ascontiguousarray(np.asarray(ref_peaks[k], np.float32).reshape(-1, 3))
        gp = np.ascontiguousarray(np.asarray(got_peaks[k], np.float32).reshape(-1, 3))
        for i in np.flatnonzero(~np.all(rp == gp, axis=1)):
            nbad += 1
            col = odf[i]
            amp = []
            for vec in (rp[i], gp[i]):
                if not vec.any():
                    amp.append(0.0)                                   # no peak at this rank
                else:
                    assert vec.tobytes() in index, "peak %d of voxel %d is not a vertex of the tessellation" % (k, i)
                    amp.append(float(col[index[vec.tobytes()]]))
            gap, top = abs(amp[0] - amp[1]), float(np.abs(col).max())
            assert gap <= tol * top, "voxel %d rank %d: peaks differ with an amplitude gap of %g = %.3g of the ODF maximum (not a tie)" % (i, k, gap, gap / max(top, 1e-30))
    return nbad
