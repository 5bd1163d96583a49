"""oracle_np — a SECOND, independent CPU restatement of the hot path, written from the reference's .jl files only
(not from oracle/fibers_oracle.c): plain NumPy, Float32 arithmetic, one voxel / one streamline at a time, the
reference's own formulation of every step (e.g. find_peaks! as boolean masks over the folded FACE list, gqi.jl:185-196,
where the C oracle and the kernels use per-vertex neighbour lists; pinv through LAPACK's Float32 SVD like Julia, where
the C oracle uses a Float64 Jacobi; eigen-decomposition through LAPACK where the C oracle restates StaticArrays' closed
form).  TEST INFRASTRUCTURE ONLY: tests/ cross-check the C oracle and the GPU path against it on small cases (SURVEY.md
§4 / §8c asks for this three-way check because the reference ships no golden vectors and cannot be run here).
Parity with the real Julia package stays unpinned; this file narrows what the other two restatements could share.

Every function cites the reference lines it follows (paths under the reference's src/)."""
import numpy as np

f32 = np.float32


# ------------------------------------------------------------------------------------------------------------
# dti.jl
# ------------------------------------------------------------------------------------------------------------
def dti_work(bval, bvec):
    """DTIwork (dti.jl:110-143): ib0, A [nvol x 7], pA = pinv(A) (LinearAlgebra.pinv: SVD, rtol = eps(Float32)*min(m,n))"""
    bval = np.asarray(bval, f32)
    bvec = np.asarray(bvec, f32).reshape(-1, 3)
    ib0 = bval == bval.min()                                                    # :117
    A = np.empty((len(bval), 7), f32)
    A[:, 0] = bvec[:, 0] ** 2                                                   # :131-136 (Float32 elementwise)
    A[:, 1] = f32(2) * bvec[:, 0] * bvec[:, 1]
    A[:, 2] = f32(2) * bvec[:, 0] * bvec[:, 2]
    A[:, 3] = bvec[:, 1] ** 2
    A[:, 4] = f32(2) * bvec[:, 1] * bvec[:, 2]
    A[:, 5] = bvec[:, 2] ** 2
    A[:, :6] *= -bval[:, None]                                                  # :138
    A[:, 6] = 1                                                                 # :140
    return dict(ib0=ib0, A=A, pA=pinv32(A))


def pinv32(A):
    A = np.asarray(A, f32)
    return np.linalg.pinv(A, rcond=float(np.finfo(f32).eps) * min(A.shape)).astype(f32)


def dti_fit_voxel(s, W):
    """dti_fit_ls(dwi::Vector, W) (dti.jl:286-316) + dti_maps (dti.jl:325-335) -> 16 values or zeros"""
    s = np.asarray(s, f32)
    ipos = s > 0                                                                # :291
    npos = int(ipos.sum())
    with np.errstate(all="ignore"):
        if npos == len(s):
            d = W["pA"] @ np.log(s)                                             # :295-296
        elif npos > 6 and bool(ipos[W["ib0"]].any()):
            d = pinv32(W["A"][ipos]) @ np.log(s[ipos])                          # :298
        else:
            return None                                                         # :300-302: all outputs zero
        d = d.astype(f32)
        s0 = np.exp(d[6])                                                       # :305
        D = np.array([[d[0], d[1], d[2]], [d[1], d[3], d[4]], [d[2], d[4], d[5]]], np.float64)   # Symmetric(D, :L), :307-311
        if not np.isfinite(D).all():
            nan = f32(np.nan)
            return dict(s0=s0, eigval=np.full(3, nan), eigvec=np.full((3, 3), nan), rd=nan, md=nan, fa=nan)
        w, E = np.linalg.eigh(D)                                                # ascending, like eigen(Symmetric)
        l1, l2, l3 = f32(w[2]), f32(w[1]), f32(w[0])                            # :313
        rd = l2 + l3                                                            # :327-329
        md = (l1 + rd) / f32(3)
        rd = rd / f32(2)
        fa = np.sqrt(((l1 - md) ** 2 + (l2 - md) ** 2 + (l3 - md) ** 2) / (l1 ** 2 + l2 ** 2 + l3 ** 2) * f32(1.5))   # :331-332
    return dict(s0=f32(s0), eigval=np.array([l1, l2, l3], f32), eigvec=E[:, ::-1].T.astype(f32), rd=f32(rd), md=f32(md), fa=f32(fa))


# ------------------------------------------------------------------------------------------------------------
# gqi.jl
# ------------------------------------------------------------------------------------------------------------
def fold_faces(faces, nvert):
    """faces[faces .> nvert] .-= nvert (gqi.jl:63-64); 1-based in, 0-based out"""
    f = np.array(faces, np.int64).reshape(-1, 3).copy()
    f[f > nvert] -= nvert
    return f - 1


def gqi_work(bval, bvec, vertices, faces, sigma=1.25):
    """GQIwork (gqi.jl:42-69): A = sinc.(V[nvert+1:end,:] * bq') with bq = bvec .* (sqrt.(bval * 0.01506f0) * Float32(sigma / pi))"""
    bval = np.asarray(bval, f32)
    bvec = np.asarray(bvec, f32).reshape(-1, 3)
    V = np.asarray(vertices, f32)
    nvert = V.shape[0] // 2
    bq = bvec * (np.sqrt(bval * f32(0.01506)) * f32(f32(sigma) / np.pi))[:, None]      # :68  (sigma/pi in Float64, then T(...))
    X = (V[nvert:] @ bq.T).astype(f32)
    with np.errstate(all="ignore"):
        A = np.where(X == 0, f32(1), np.sin(f32(np.pi) * X) / (f32(np.pi) * X)).astype(f32)   # Base.sinc, :69
    return dict(nvert=nvert, A=A, faces=fold_faces(faces, nvert), V=V)


def isless_desc_order(v):
    """sortperm(v, rev=true) (gqi.jl:198): descending by Base.isless (NaN greatest, -0.0 < +0.0), stable -> ties keep ascending index"""
    v = np.asarray(v, f32)
    key = v.view(np.uint32).astype(np.int64)
    key = np.where(key & 0x80000000, -(key & 0x7fffffff) - 1, key)              # total order of the bit patterns: -0.0 (-1) < +0.0 (0)
    key = np.where(np.isnan(v), np.int64(1) << 40, key)                         # NaN above everything
    return np.lexsort((np.arange(len(v)), -key))


def find_peaks(o, faces0):
    """find_peaks!(W) (gqi.jl:180-201), the reference's own formulation over the folded face list"""
    o = np.asarray(o, f32)
    pk = o.copy()                                                               # :184
    a, b, c = faces0[:, 0], faces0[:, 1], faces0[:, 2]
    with np.errstate(invalid="ignore"):
        pk[a[(o[b] >= o[a]) | (o[c] >= o[a])]] = 0                              # :185-188
        pk[b[(o[a] >= o[b]) | (o[c] >= o[b])]] = 0                              # :189-192
        pk[c[(o[b] >= o[c]) | (o[a] >= o[c])]] = 0                              # :193-196
        nvalid = int((pk > 0).sum())                                            # :200
    return isless_desc_order(pk), nvalid, pk


def odf_peaks_qa(o, W, npeak=3):
    """gqi.jl:147-159 (same code in dsi.jl:244-258): odfmin, peaks, qa of one voxel"""
    with np.errstate(invalid="ignore"):
        odfmin = f32(np.nan) if np.isnan(o).any() else o.min()                  # minimum() propagates NaN
    isort, nvalid, _ = find_peaks(o, W["faces"])
    peak = np.zeros((npeak, 3), f32)
    qa = np.zeros(npeak, f32)
    for k in range(min(nvalid, npeak)):
        peak[k] = W["V"][isort[k]]                                              # first half of `vertices`, :154-155
        qa[k] = o[isort[k]] - odfmin
    return peak, qa


def gqi_voxel(s, W):
    """gqi_rec's loop body for one voxel (gqi.jl:139-159); None = voxel skipped (outputs stay zero)"""
    s = np.asarray(s, f32).copy()
    with np.errstate(invalid="ignore"):
        s[s < 0] = 0                                                            # :140
        smax = f32(np.nan) if np.isnan(s).any() else s.max()
    if smax == 0:                                                               # :142
        return None
    with np.errstate(all="ignore"):
        o = (W["A"].astype(np.float64) @ s.astype(np.float64)).astype(f32)      # mul!(o, A, s), :144 (BLAS order unknown: exact products, one rounding)
    peak, qa = odf_peaks_qa(o, W)
    return dict(odf=o, peak=peak, qa=qa)


def odfmax_of(odf_rows):
    """maximum(mean(odf.vol, dims=4)) (gqi.jl:164): Base's reduction over dim 4 runs sequentially over the vertices in Float32,
    mean divides by n, maximum propagates NaN.  odf_rows: [nvox, nvert]"""
    acc = np.zeros(odf_rows.shape[0], f32)
    with np.errstate(all="ignore"):
        for v in range(odf_rows.shape[1]):
            acc = acc + odf_rows[:, v]
        m = acc / f32(odf_rows.shape[1])
    return f32(np.nan) if np.isnan(m).any() else m.max()


# ------------------------------------------------------------------------------------------------------------
# dsi.jl
# ------------------------------------------------------------------------------------------------------------
def dsi_work(bval, bvec, vertices, faces, hann_width=32):
    """DSIwork (dsi.jl:59-143)"""
    bval = np.asarray(bval, f32)
    bvec = np.asarray(bvec, f32).reshape(-1, 3)
    V = np.asarray(vertices, f32)
    nvert = V.shape[0] // 2
    q = bvec * np.sqrt(bval)[:, None]                                           # :62
    dq = np.sqrt(bval[bval > bval.min()].min())                                 # :65-66
    iq = np.rint(q / dq).astype(np.int64)                                       # round(): ties to even, :67
    nfft = int(iq.max() - iq.min() + 1)
    nfft = 2 ** int(np.ceil(np.log2(nfft)))                                     # :70-71
    shift = nfft // 2 + 1                                                       # :73 (1-based)
    sub = iq + shift
    lin = (sub[:, 0] - 1) + nfft * ((sub[:, 1] - 1) + nfft * (sub[:, 2] - 1))   # LinearIndices, 0-based here, :74-77
    if hann_width == 0:
        H = np.ones(nfft ** 3, f32)
    else:
        H = np.zeros(nfft ** 3, f32)
        r = np.sqrt((iq.astype(np.float64) ** 2).sum(1))
        H[lin] = ((1 + np.cos(r * (2 * np.pi / hann_width))) * .5).astype(f32)  # later duplicates overwrite earlier, :84
    qr = f32(nfft / 2 - 1) * np.arange(0.3, 0.9 + 1e-9, 0.03).astype(f32)       # collect(T, .3:.03:.9), :104
    dqr = qr[1] - qr[0]
    interp = V[nvert:, :, None] * qr[None, None, :] + f32(shift)                # x * qr' .+ iq_shift (1-based coordinates), :106-109
    return dict(nfft=nfft, nvert=nvert, lin=lin, H=H, qr2=qr ** 2, dqr=dqr, interp=interp.astype(f32),
                faces=fold_faces(faces, nvert), V=V)


def dsi_voxel(s, W):
    """dsi_rec's loop body (dsi.jl:204-258); None = skipped"""
    n = W["nfft"]
    X = np.zeros(n ** 3, f32)
    X[W["lin"]] = np.asarray(s, f32)                                            # :205 (later frames overwrite earlier at equal points)
    with np.errstate(invalid="ignore"):
        xmax = f32(np.nan) if np.isnan(X).any() else X.max()
    if xmax == 0:                                                               # :207 (before the clamp)
        return None
    with np.errstate(all="ignore"):
        X = np.where(X < 0, f32(0), X) * W["H"]                                 # :209, :212  (max.(X, 0) keeps NaN)
        X3 = X.reshape(n, n, n, order="F")
        ns = n // 2
        x = np.roll(np.fft.fftn(np.roll(X3.astype(np.complex64), (ns, ns, ns), (0, 1, 2))), (ns, ns, ns), (0, 1, 2))   # :218-220
        p = x.real.astype(f32)
        p = (p / p.sum(dtype=f32)).astype(f32)                                  # :224-225
        pdf = p.reshape(-1, order="F")[W["lin"]]                                # :227
        # interpolate(p, BSpline(Linear())) at 1-based coordinates (:230-238), radial sum in Float32 (:233-242)
        c = W["interp"] - f32(1)                                                # [nvert, 3, nrad], 0-based
        i0 = np.floor(c).astype(np.int64)
        w = (c - i0).astype(f32)
        o = np.zeros(W["nvert"], f32)
        for r in range(c.shape[2]):
            val = np.zeros(W["nvert"], f32)
            for dx in (0, 1):
                for dy in (0, 1):
                    for dz in (0, 1):
                        wx = w[:, 0, r] if dx else f32(1) - w[:, 0, r]
                        wy = w[:, 1, r] if dy else f32(1) - w[:, 1, r]
                        wz = w[:, 2, r] if dz else f32(1) - w[:, 2, r]
                        val = val + wx * wy * wz * p[np.minimum(i0[:, 0, r] + dx, n - 1), np.minimum(i0[:, 1, r] + dy, n - 1),
                                                     np.minimum(i0[:, 2, r] + dz, n - 1)]
            o = o + val * W["qr2"][r]
        o = (o * W["dqr"]).astype(f32)
    peak, qa = odf_peaks_qa(o, W)
    return dict(pdf=pdf.astype(f32), odf=o, peak=peak, qa=qa)


# ------------------------------------------------------------------------------------------------------------
# stream.jl (macro scale, angle picking)
# ------------------------------------------------------------------------------------------------------------
def norm32(v):
    """LinearAlgebra.norm of a short Float32 vector (generic_norm2): squares in Float32, sum and sqrt in Float64, result Float32"""
    v = np.asarray(v, f32)
    return f32(np.sqrt(np.sum((v * v).astype(np.float64))))


def _pick_by_angle(vec, cands):
    """stream_pick_by_angle! (stream.jl:340-374) on cands [3, nvec]: (k, cos_k) with cos = -Inf for zero vectors"""
    nvec = cands.shape[1]
    cos = np.full(nvec, -np.inf, f32)
    cosabs = np.full(nvec, -np.inf, f32)
    for k in range(nvec):
        v = cands[:, k]
        if not (v == 0).all():
            cos[k] = (vec[0] * v[0] + vec[1] * v[1]) + vec[2] * v[2]
            cosabs[k] = np.abs(cos[k])
    nanpos = np.flatnonzero(np.isnan(cosabs))
    k = int(nanpos[0]) if len(nanpos) else int(np.argmax(cosabs))
    return k, cos[k]


def trilinear_direction(nxt, vec, ovecs):
    """The blend of fib_stream_params.interp = 1 (include/fibers_hip.h; NOT in the reference): corners of floor(nxt) + {0,1}^3
    inside the volume, x fastest; weight (ax * ay) * az; each corner's vector picked against `vec` by the angle rule and
    sign-aligned; Float32 products and sums in that order; LinearAlgebra.norm.  Returns None when the blend is zero / not finite."""
    shape = ovecs.shape[2:]
    g0 = np.floor(nxt).astype(f32)
    t = (nxt - g0).astype(f32)
    s = np.zeros(3, f32)
    one = f32(1)
    for c in range(8):
        cx, cy, cz = c & 1, (c >> 1) & 1, c >> 2
        g = g0 + np.array([cx, cy, cz], f32)
        if not all(1 <= g[d] <= shape[d] for d in range(3)):
            continue
        tc = ((t[0] if cx else one - t[0]) * (t[1] if cy else one - t[1])) * (t[2] if cz else one - t[2])
        gi = g.astype(np.int64) - 1
        cands = ovecs[:, :, gi[0], gi[1], gi[2]]
        k, ck = _pick_by_angle(vec, cands)
        with np.errstate(invalid="ignore"):
            if not np.isfinite(ck):
                continue
        sg = tc if ck > 0 else -tc
        u = cands[:, k]
        s = np.array([s[0] + sg * u[0], s[1] + sg * u[1], s[2] + sg * u[2]], f32)
    m = np.max(np.abs(s))
    if m == 0 or not np.isfinite(m):
        return None
    return (s / norm32(s)).astype(f32)


def stream_line(seed, sub, ovecs, mask, step=0.5, cosang_thresh=None, smooth=0.2, len_max=None, interp="nearest"):
    """stream_new_line (stream.jl:625-690) with stream_new_point! (:501-541) and stream_pick_by_angle! (:340-374).
    ovecs: [3, nvec, nx, ny, nz] Float32 (masked vectors zeroed, :141-145); seed: 1-based voxel (ix, iy, iz); returns [npts, 3].
    interp="trilinear": the library's non-reference option (trilinear_direction)."""
    nvec = ovecs.shape[1]
    shape = ovecs.shape[2:]
    step, smooth = f32(step), f32(smooth)
    cosang_thresh = f32(np.cos(np.deg2rad(45.0))) if cosang_thresh is None else f32(cosang_thresh)
    len_max = max(shape) if len_max is None else len_max
    line = []
    npts = 0
    ivec = 0                                                                    # W.ivec_next = 1, :645
    for fwd in (1, -1):
        pos = np.asarray(seed, f32) + np.asarray(sub, f32)                      # :648
        vec = ovecs[:, ivec, seed[0] - 1, seed[1] - 1, seed[2] - 1] * f32(fwd)  # :649
        while True:
            nxt = pos + vec * step                                              # :512
            with np.errstate(invalid="ignore"):
                if not np.isfinite(nxt).all():
                    break
            vox = np.rint(nxt).astype(np.int64)                                 # round(Int, .): ties to even, :514
            if not all(1 <= vox[d] <= shape[d] for d in range(3)):              # :517
                break
            if not mask[vox[0] - 1, vox[1] - 1, vox[2] - 1]:                    # :520
                break
            cos = np.full(nvec, -np.inf, f32)
            cosabs = np.full(nvec, -np.inf, f32)                                # a zero vector: cosang = cosangabs = -Inf, :354
            for k in range(nvec):
                v = ovecs[:, k, vox[0] - 1, vox[1] - 1, vox[2] - 1]
                if not (v == 0).all():                                          # iszero, :353
                    cos[k] = (vec[0] * v[0] + vec[1] * v[1]) + vec[2] * v[2]    # dot, :356
                    cosabs[k] = np.abs(cos[k])                                  # :357
            nanpos = np.flatnonzero(np.isnan(cosabs))
            k = int(nanpos[0]) if len(nanpos) else int(np.argmax(cosabs))       # argmax: first maximum, a NaN wins, :361
            if not np.isfinite(cos[k]):                                         # :363
                break
            v = ovecs[:, k, vox[0] - 1, vox[1] - 1, vox[2] - 1]
            vnext = v.copy() if cos[k] > 0 else -v                              # :365-369
            ivec = k                                                            # :371
            if interp == "trilinear":
                vnext = trilinear_direction(nxt.astype(f32), vec.astype(f32), ovecs)
                if vnext is None:
                    break
            if fwd == 1:
                line.insert(0, pos.copy())                                      # prepend!, :660
            else:
                line.append(pos.copy())
            npts += 1
            d = (vec[0] * vnext[0] + vec[1] * vnext[1]) + vec[2] * vnext[2]
            if d < cosang_thresh:                                               # :670
                break
            if npts > len_max:                                                  # :674
                break
            if smooth != 0:                                                     # :677-681
                vnext = smooth * vec + (f32(1) - smooth) * vnext
                vnext = vnext / norm32(vnext)
            pos, vec = nxt, vnext.astype(f32)                                   # :684-685
    return np.array(line, f32).reshape(-1, 3)
