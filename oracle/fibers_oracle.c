/*
 * oracle/fibers_oracle.c — CPU restatement of the Fibers.jl hot path (TEST INFRASTRUCTURE).
 *
 * PARITY UNPINNED: the reference (lincbrain/Fibers.jl) is Julia, cannot be run
 * in this image, and ships no tests or golden vectors (test/runtests.jl:4-6 is
 * empty).  This file restates the reference algorithms by hand from the Julia
 * sources; every function cites the file:line it follows.  It is checked only
 * against analytic known answers and float64 closed forms (tests/).
 *
 * This is the checker and the CPU baseline ("port") — never the product.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Conventions: float32 arithmetic throughout (the reference dispatches on
 * Float32), volumes are Julia column-major [nx,ny,nz,nframes] (x fastest),
 * threading is the reference's: static contiguous blocks of z-slices per
 * thread (Threads.@threads, dti.jl:258, gqi.jl:132, dsi.jl:197) and contiguous
 * seed chunks (stream.jl:757-761).  Build: see oracle/Makefile
 * (-O2 -mfma -ffp-contract=off -fopenmp; no fast-math).  The two BLAS sgemv calls of the
 * reference (dti.jl:296 `mul!(d, pA, logs)`, gqi.jl:144 `mul!(o, A, s)`) run in OpenBLAS with an
 * unknowable blocking / FMA usage; they are restated as a frame-ordered chain of explicit fmaf().
 * Nothing else is ever contracted (Julia never fuses a*b+c on its own).
 *
 * Third-party arithmetic restated from the published algorithms (packages are
 * not vendored under /root/reference; Project.toml compat pins in brackets):
 *   - StaticArrays [1.4.4] eigen(Symmetric(SMatrix{3,3})) closed form  -> sym3_eigen()
 *   - LinearAlgebra norm() of a short vector (generic_norm2: squares in T,
 *     sum and sqrt in Float64, result converted to T)                    -> norm3()
 *   - FFTW [1.5.0] plan_fft 16^3 complex forward, unnormalised            -> fft3_16()
 *   - Interpolations [0.13.6] BSpline(Linear()) on 1-based knots          -> trilinear()
 *   - Base sortperm!(rev=true): stable, ties keep ascending index         -> sortperm_desc()
 *   - Base round(Int, x): ties to even                                    -> rintf()
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

/* ------------------------------------------------------------------------ */
/* helpers                                                                   */
/* ------------------------------------------------------------------------ */

static void zslab(int nz, int nthreads, int tid, int *z0, int *z1)
{
    /* Threads.@threads :static — contiguous, near-equal blocks (Appendix A.1) */
    int len = nz / nthreads, rem = nz % nthreads;
    *z0 = tid * len + (tid < rem ? tid : rem);
    *z1 = *z0 + len + (tid < rem ? 1 : 0);
}

/* StaticArrays eigen for a real symmetric 3x3 (lower triangle a11 a21 a31 a22 a32 a33).
 * Returns eigenvalues ascending in w[3] and unit eigenvectors as columns V[:,k] = v[k][0..2].
 * Follows StaticArrays/src/eigen.jl `_eig(::Size{(3,3)}, A::RealHermSymComplexHerm, ...)`
 * (call site dti.jl:311).  All arithmetic in float32 like the reference. */
static void cross3(const float a[3], const float b[3], float c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

void orc_sym3_eigen(float a11, float a12, float a13, float a22, float a23, float a33,
                    float w[3], float v[3][3])
{
    float p1 = a12 * a12 + a13 * a13 + a23 * a23;
    if (p1 == 0.0f) { /* diagonal: sort the diagonal, unit axes as vectors */
        static const float e[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        int o0, o1, o2;
        if (a11 < a22) {
            if (a22 < a33)      { o0 = 0; o1 = 1; o2 = 2; }
            else if (a33 < a11) { o0 = 2; o1 = 0; o2 = 1; }
            else                { o0 = 0; o1 = 2; o2 = 1; }
        } else {
            if (a11 < a33)      { o0 = 1; o1 = 0; o2 = 2; }
            else if (a33 < a22) { o0 = 2; o1 = 1; o2 = 0; }
            else                { o0 = 1; o1 = 2; o2 = 0; }
        }
        const float d[3] = {a11, a22, a33};
        w[0] = d[o0]; w[1] = d[o1]; w[2] = d[o2];
        memcpy(v[0], e[o0], sizeof(float) * 3);
        memcpy(v[1], e[o1], sizeof(float) * 3);
        memcpy(v[2], e[o2], sizeof(float) * 3);
        return;
    }
    float q = (a11 + a22 + a33) / 3.0f;
    float p2 = (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) + (a33 - q) * (a33 - q) + 2.0f * p1;
    float p = sqrtf(p2 / 6.0f);
    float invp = 1.0f / p;
    float b11 = (a11 - q) * invp, b22 = (a22 - q) * invp, b33 = (a33 - q) * invp;
    float b12 = a12 * invp, b13 = a13 * invp, b23 = a23 * invp;
    /* det of the 3x3 (StaticArrays det: cofactor expansion along the first column) */
    float detB = b11 * (b22 * b33 - b23 * b23) - b12 * (b12 * b33 - b23 * b13)
               + b13 * (b12 * b23 - b22 * b13);
    float r = detB / 2.0f;
    float phi;
    const float PI_F = 3.14159274101257324f;
    if (r <= -1.0f)      phi = PI_F / 3.0f;
    else if (r >= 1.0f)  phi = 0.0f;
    else                 phi = acosf(r) / 3.0f;
    float eig3 = q + 2.0f * p * cosf(phi);
    float eig1 = q + 2.0f * p * cosf(phi + (2.0f * PI_F / 3.0f));
    float eig2 = 3.0f * q - eig1 - eig3;
    if (r > 0.0f) { float t = eig1; eig1 = eig3; eig3 = t; }

    /* first eigenvector: best cross product of rows of A - eig1 I */
    float r1[3] = {a11 - eig1, a12, a13};
    float r2[3] = {a12, a22 - eig1, a23};
    float r3[3] = {a13, a23, a33 - eig1};
    float n1 = r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2];
    float n2 = r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2];
    float n3 = r3[0] * r3[0] + r3[1] * r3[1] + r3[2] * r3[2];
    float r12[3], r23[3], r31[3];
    cross3(r1, r2, r12); cross3(r2, r3, r23); cross3(r3, r1, r31);
    float n12 = r12[0] * r12[0] + r12[1] * r12[1] + r12[2] * r12[2];
    float n23 = r23[0] * r23[0] + r23[1] * r23[1] + r23[2] * r23[2];
    float n31 = r31[0] * r31[0] + r31[1] * r31[1] + r31[2] * r31[2];
    float ev1[3];
    const float *best; float nb;
    if (n12 * n3 > n23 * n1) {
        if (n12 * n3 > n31 * n2) { best = r12; nb = n12; } else { best = r31; nb = n31; }
    } else {
        if (n23 * n1 > n31 * n2) { best = r23; nb = n23; } else { best = r31; nb = n31; }
    }
    { float s = sqrtf(nb); ev1[0] = best[0] / s; ev1[1] = best[1] / s; ev1[2] = best[2] / s; }

    /* second eigenvector: 2x2 problem in the plane orthogonal to ev1 */
    float o1[3], o2[3];
    if (fabsf(ev1[0]) < fabsf(ev1[1])) {
        float s = sqrtf(ev1[0] * ev1[0] + ev1[2] * ev1[2]);
        o1[0] = -ev1[2] / s; o1[1] = 0.0f / s; o1[2] = ev1[0] / s;
    } else {
        float s = sqrtf(ev1[1] * ev1[1] + ev1[2] * ev1[2]);
        o1[0] = 0.0f / s; o1[1] = ev1[2] / s; o1[2] = -ev1[1] / s;
    }
    cross3(ev1, o1, o2);
    float ao1[3] = {a11 * o1[0] + a12 * o1[1] + a13 * o1[2],
                    a12 * o1[0] + a22 * o1[1] + a23 * o1[2],
                    a13 * o1[0] + a23 * o1[1] + a33 * o1[2]};
    float ao2[3] = {a11 * o2[0] + a12 * o2[1] + a13 * o2[2],
                    a12 * o2[0] + a22 * o2[1] + a23 * o2[2],
                    a13 * o2[0] + a23 * o2[1] + a33 * o2[2]};
    float c11 = o1[0] * ao1[0] + o1[1] * ao1[1] + o1[2] * ao1[2] - eig2;
    float c12 = o1[0] * ao2[0] + o1[1] * ao2[1] + o1[2] * ao2[2];
    float c22 = o2[0] * ao2[0] + o2[1] * ao2[1] + o2[2] * ao2[2] - eig2;
    float c11s = c11 * c11, c12s = c12 * c12, c22s = c22 * c22;
    float ev2[3];
    float q1, q2; int have = 1;
    if (c11s >= c22s) {
        if (c11s > 0.0f || c12s > 0.0f) {
            if (c11s >= c12s) { float t = c12 / c11; q2 = 1.0f / sqrtf(1.0f + t * t); q1 = t * q2; }
            else              { float t = c11 / c12; q1 = 1.0f / sqrtf(1.0f + t * t); q2 = t * q1; }
        } else { have = 0; q1 = q2 = 0.0f; }
    } else {
        if (c22s >= c12s) { float t = c12 / c22; q1 = 1.0f / sqrtf(1.0f + t * t); q2 = t * q1; }
        else              { float t = c22 / c12; q2 = 1.0f / sqrtf(1.0f + t * t); q1 = t * q2; }
    }
    if (have) for (int i = 0; i < 3; i++) ev2[i] = q1 * o1[i] - q2 * o2[i];
    else      for (int i = 0; i < 3; i++) ev2[i] = o1[i];
    float ev3[3];
    cross3(ev1, ev2, ev3);
    if (r > 0.0f) {
        float t = eig1; eig1 = eig3; eig3 = t;
        for (int i = 0; i < 3; i++) { float u = ev1[i]; ev1[i] = ev3[i]; ev3[i] = u; }
    }
    w[0] = eig1; w[1] = eig2; w[2] = eig3;
    memcpy(v[0], ev1, sizeof ev1); memcpy(v[1], ev2, sizeof ev2); memcpy(v[2], ev3, sizeof ev3);
}

/* ------------------------------------------------------------------------ */
/* DTI / ADC  (dti.jl:164-213, 243-335)                                      */
/* ------------------------------------------------------------------------ */

/* dti_maps, dti.jl:325-335 */
static void dti_maps(float e1, float e2, float e3, float *rd, float *md, float *fa)
{
    float r = e2 + e3;
    float m = (e1 + r) / 3.0f;
    r = r / 2.0f;
    float num = (e1 - m) * (e1 - m) + (e2 - m) * (e2 - m) + (e3 - m) * (e3 - m);
    float den = e1 * e1 + e2 * e2 + e3 * e3;
    *fa = sqrtf(num / den * 1.5f);
    *rd = r; *md = m;
}

/* st_eigen (structens.jl:13-37): eigen(Symmetric(S, :L)) per voxel; eigvec[vox + nvox*(i + 3 j)] = component i of
 * eigenvector j (ascending eigenvalues), eigval[vox + nvox*k]. */
void orc_st_eigen(const float *sxx, const float *sxy, const float *sxz, const float *syy, const float *syz, const float *szz,
                  int64_t nvox, float *eigvec, float *eigval)
{
    for (int64_t vox = 0; vox < nvox; vox++) {
        float w[3], v[3][3];
        orc_sym3_eigen(sxx[vox], sxy[vox], sxz[vox], syy[vox], syz[vox], szz[vox], w, v);
        for (int j = 0; j < 3; j++) {
            eigval[(int64_t)j * nvox + vox] = w[j];
            for (int c = 0; c < 3; c++) eigvec[(int64_t)(c + 3 * j) * nvox + vox] = v[j][c];
        }
    }
}

/* Per-voxel tail shared by the full and partial branches: d[7] -> 16 outputs (dti.jl:305-315). */
void orc_dti_from_d(const float d[7], float out[16])
{
    float w[3], v[3][3];
    out[0] = expf(d[6]);
    /* D lower triangle: d1 d2 d3 / d4 d5 / d6 = Dxx Dxy Dxz Dyy Dyz Dzz (dti.jl:307-309) */
    orc_sym3_eigen(d[0], d[1], d[2], d[3], d[4], d[5], w, v);
    out[1] = w[2]; out[2] = w[1]; out[3] = w[0];
    for (int c = 0; c < 3; c++) { out[4 + c] = v[2][c]; out[7 + c] = v[1][c]; out[10 + c] = v[0][c]; }
    dti_maps(w[2], w[1], w[0], &out[13], &out[14], &out[15]);
}

/*
 * dti_fit_ls volume driver (dti.jl:243-278) + per-voxel fit (dti.jl:286-316).
 * pA is [7 x nvol] column-major.  Voxels that need the per-voxel pinv branch
 * (dti.jl:297-298) are NOT solved here: their linear index is appended to
 * `partial` (capacity nvox) and the Python side solves them with a float32 SVD
 * pinv (LAPACK, like Julia) and calls orc_dti_from_d.  Outputs are 10 planar
 * volumes, zero-filled for skipped voxels (MRI(mask,n,Float32), mri.jl:249-265).
 * status[vox]: 0 skipped/masked, 1 full, 2 partial (pending), 3 degenerate->zeros.
 */
void orc_dti_fit(const float *dwi, const uint8_t *mask, int nx, int ny, int nz, int nvol,
                 const float *pA, const uint8_t *ib0,
                 float *s0, float *l1, float *l2, float *l3,
                 float *e1, float *e2, float *e3, float *rd, float *md, float *fa,
                 int64_t *partial, int64_t *npartial, int nthreads)
{
    const int64_t nxy = (int64_t)nx * ny, nvox = nxy * nz;
    int64_t np = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        int z0, z1;
        zslab(nz, omp_get_num_threads(), omp_get_thread_num(), &z0, &z1);
        float *s = (float *)malloc(sizeof(float) * nvol);
        float *logs = (float *)malloc(sizeof(float) * nvol);
        for (int64_t vox = z0 * nxy; vox < z1 * nxy; vox++) {
            if (mask[vox] == 0) continue;                      /* dti.jl:261 */
            int npos = 0, b0pos = 0;
            for (int i = 0; i < nvol; i++) {                   /* strided gather, dti.jl:272 */
                s[i] = dwi[(int64_t)i * nvox + vox];
                if (s[i] > 0.0f) { npos++; if (ib0[i]) b0pos = 1; }   /* dti.jl:291-292 */
            }
            float d[7] = {0, 0, 0, 0, 0, 0, 0}, out[16];
            if (npos == nvol) {                                /* dti.jl:294-296 */
                for (int i = 0; i < nvol; i++) logs[i] = logf(s[i]);
                for (int i = 0; i < nvol; i++)
                    for (int j = 0; j < 7; j++) d[j] = fmaf(pA[j + 7 * i], logs[i], d[j]);
            } else if (npos > 6 && b0pos) {                    /* dti.jl:297-298 */
                int64_t slot;
#pragma omp atomic capture
                slot = np++;
                partial[slot] = vox;
                continue;
            } else {
                continue;                                      /* dti.jl:299-303: zeros */
            }
            orc_dti_from_d(d, out);
            s0[vox] = out[0]; l1[vox] = out[1]; l2[vox] = out[2]; l3[vox] = out[3];
            for (int c = 0; c < 3; c++) {
                e1[c * nvox + vox] = out[4 + c];
                e2[c * nvox + vox] = out[7 + c];
                e3[c * nvox + vox] = out[10 + c];
            }
            rd[vox] = out[13]; md[vox] = out[14]; fa[vox] = out[15];
        }
        free(s); free(logs);
    }
    *npartial = np;
}

/* adc_fit (dti.jl:164-213); pA is [2 x nvol] column-major. status as above. */
void orc_adc_fit(const float *dwi, const uint8_t *mask, int nx, int ny, int nz, int nvol,
                 const float *pA, const uint8_t *ib0, float *adc, float *s0,
                 int64_t *partial, int64_t *npartial, int nthreads)
{
    const int64_t nxy = (int64_t)nx * ny, nvox = nxy * nz;
    int64_t np = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        int z0, z1;
        zslab(nz, omp_get_num_threads(), omp_get_thread_num(), &z0, &z1);
        float *s = (float *)malloc(sizeof(float) * nvol);
        for (int64_t vox = z0 * nxy; vox < z1 * nxy; vox++) {
            if (mask[vox] == 0) continue;
            int npos = 0, b0pos = 0;
            for (int i = 0; i < nvol; i++) {
                s[i] = dwi[(int64_t)i * nvox + vox];
                if (s[i] > 0.0f) { npos++; if (ib0[i]) b0pos = 1; }
            }
            if (npos == nvol) {
                float d0 = 0, d1 = 0;
                for (int i = 0; i < nvol; i++) {
                    float l = logf(s[i]);
                    d0 = fmaf(pA[0 + 2 * i], l, d0); d1 = fmaf(pA[1 + 2 * i], l, d1);
                }
                adc[vox] = d0; s0[vox] = expf(d1);             /* dti.jl:212 */
            } else if (npos > 6 && b0pos) {                    /* dti.jl:206-207 */
                int64_t slot;
#pragma omp atomic capture
                slot = np++;
                partial[slot] = vox;
            }
        }
        free(s);
    }
    *npartial = np;
}

/* ------------------------------------------------------------------------ */
/* ODF peak finder (gqi.jl:180-201)                                          */
/* ------------------------------------------------------------------------ */

/* Julia isless on floats: NaN is greater than everything, -0.0 < +0.0 */
static int jl_isless(float a, float b)
{
    if (isnan(a)) return 0;
    if (isnan(b)) return 1;
    if (a == b) return signbit(a) && !signbit(b);
    return a < b;
}

/* sortperm!(ix, v, rev=true): descending, stable (ties -> ascending index). Bottom-up merge. */
static void sortperm_desc(const float *v, int n, int32_t *ix, int32_t *tmp)
{
    for (int i = 0; i < n; i++) ix[i] = i;
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                /* take right only if strictly "before" left in reverse order: v[right] > v[left] */
                if (jl_isless(v[ix[i]], v[ix[j]])) tmp[k++] = ix[j++]; else tmp[k++] = ix[i++];
            }
            while (i < mid) tmp[k++] = ix[i++];
            while (j < hi) tmp[k++] = ix[j++];
        }
        memcpy(ix, tmp, sizeof(int32_t) * n);
    }
}

/*
 * find_peaks!(W): o[nvert], folded faces [nfaces x 3] 0-based (column-major as in
 * Julia: faces[f + nfaces*c]).  Fills odf_peak, isort (0-based), returns nvalid.
 */
int orc_find_peaks(const float *o, int nvert, const int32_t *faces, int nfaces,
                   float *odf_peak, int32_t *isort, int32_t *tmp)
{
    memcpy(odf_peak, o, sizeof(float) * nvert);                 /* gqi.jl:184 */
    const int32_t *f1 = faces, *f2 = faces + nfaces, *f3 = faces + 2 * nfaces;
    /* NB each sweep tests the ORIGINAL o, not odf_peak (gqi.jl:185-196) */
    for (int f = 0; f < nfaces; f++)
        if (o[f2[f]] >= o[f1[f]] || o[f3[f]] >= o[f1[f]]) odf_peak[f1[f]] = 0.0f;
    for (int f = 0; f < nfaces; f++)
        if (o[f1[f]] >= o[f2[f]] || o[f3[f]] >= o[f2[f]]) odf_peak[f2[f]] = 0.0f;
    for (int f = 0; f < nfaces; f++)
        if (o[f2[f]] >= o[f3[f]] || o[f1[f]] >= o[f3[f]]) odf_peak[f3[f]] = 0.0f;
    sortperm_desc(odf_peak, nvert, isort, tmp);                 /* gqi.jl:198 */
    int nvalid = 0;
    for (int i = 0; i < nvert; i++) if (odf_peak[i] > 0.0f) nvalid++;   /* gqi.jl:200 */
    return nvalid;
}

/* batch form used by tests of the standalone fib_find_peaks entry: odf is planar [nvox, nvert] */
void orc_find_peaks_batch(const float *odf, int64_t nvox, int nvert, const int32_t *faces, int nfaces,
                          int32_t *isort_top3, int32_t *nvalid)
{
    float *o = (float *)malloc(sizeof(float) * nvert), *pk = (float *)malloc(sizeof(float) * nvert);
    int32_t *is = (int32_t *)malloc(sizeof(int32_t) * nvert), *tmp = (int32_t *)malloc(sizeof(int32_t) * nvert);
    for (int64_t vox = 0; vox < nvox; vox++) {
        for (int v = 0; v < nvert; v++) o[v] = odf[(int64_t)v * nvox + vox];
        nvalid[vox] = orc_find_peaks(o, nvert, faces, nfaces, pk, is, tmp);
        for (int k = 0; k < 3; k++) isort_top3[k * nvox + vox] = k < nvert ? is[k] : -1;
    }
    free(o); free(pk); free(is); free(tmp);
}

/* peaks + qa written for one voxel (gqi.jl:147-159 == dsi.jl:244-258) */
static void write_peaks(const float *o, int nvert, const int32_t *faces, int nfaces,
                        const float *verts, int nverts_full, int64_t vox, int64_t nvox,
                        float *peak[3], float *qa[3], float *pk, int32_t *is, int32_t *tmp)
{
    float odfmin = o[0];
    for (int v = 1; v < nvert; v++) if (o[v] < odfmin) odfmin = o[v];
    int nvalid = orc_find_peaks(o, nvert, faces, nfaces, pk, is, tmp);
    int n = nvalid < 3 ? nvalid : 3;
    for (int k = 0; k < n; k++) {
        int iv = is[k];      /* row of the FIRST half of vertices (gqi.jl:155) */
        for (int c = 0; c < 3; c++) peak[k][c * nvox + vox] = verts[iv + nverts_full * c];
        qa[k][vox] = o[iv] - odfmin;
    }
}

/* global QA normalisation (gqi.jl:164-168, dsi.jl:263-267): max over voxels of mean over vertices */
static float odf_max_of_means(const float *odf, int64_t nvox, int nvert, int nthreads)
{
    float best = -INFINITY;
    int anynan = 0;
#pragma omp parallel for num_threads(nthreads) reduction(max : best) reduction(| : anynan)
    for (int64_t vox = 0; vox < nvox; vox++) {
        float sum = 0.0f;
        for (int v = 0; v < nvert; v++) sum += odf[(int64_t)v * nvox + vox];
        float m = sum / (float)nvert;   /* Statistics.mean: sum(A, dims=4) ./ n; the sum runs sequentially over dim 4 (Base mapreducedim!) */
        if (isnan(m)) anynan = 1;
        else if (m > best) best = m;
    }
    return anynan ? NAN : best;   /* Julia maximum() propagates NaN */
}

static void qa_scale(float *qa[3], int64_t nvox, float odfmax, int nthreads)
{
    for (int k = 0; k < 3; k++) {
#pragma omp parallel for num_threads(nthreads)
        for (int64_t i = 0; i < nvox; i++) qa[k][i] = qa[k][i] / odfmax;
    }
}

/* ------------------------------------------------------------------------ */
/* GQI (gqi.jl:109-171)                                                      */
/* ------------------------------------------------------------------------ */

/*
 * A is [nvert x nvol] column-major (built by the Python side after gqi.jl:67-69),
 * faces are folded 0-based [nfaces x 3] column-major, verts [nverts_full x 3]
 * column-major.  odf [nvox*nvert], peak[k] [nvox*3], qa[k] [nvox], zero-filled by the caller.
 */
float orc_gqi_rec(const float *dwi, const uint8_t *mask, int nx, int ny, int nz, int nvol,
                  const float *A, int nvert, const int32_t *faces, int nfaces,
                  const float *verts, int nverts_full,
                  float *odf, float *peak0, float *peak1, float *peak2,
                  float *qa0, float *qa1, float *qa2, int nthreads)
{
    const int64_t nxy = (int64_t)nx * ny, nvox = nxy * nz;
    float *peak[3] = {peak0, peak1, peak2}, *qa[3] = {qa0, qa1, qa2};
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        int z0, z1;
        zslab(nz, omp_get_num_threads(), omp_get_thread_num(), &z0, &z1);
        float *s = (float *)malloc(sizeof(float) * nvol);
        float *o = (float *)malloc(sizeof(float) * nvert), *pk = (float *)malloc(sizeof(float) * nvert);
        int32_t *is = (int32_t *)malloc(sizeof(int32_t) * nvert), *tmp = (int32_t *)malloc(sizeof(int32_t) * nvert);
        for (int64_t vox = z0 * nxy; vox < z1 * nxy; vox++) {
            if (mask[vox] == 0) continue;                       /* gqi.jl:135 */
            float smax = -INFINITY;
            for (int i = 0; i < nvol; i++) {                    /* gqi.jl:139-140 */
                float x = dwi[(int64_t)i * nvox + vox];
                if (x < 0.0f) x = 0.0f;
                s[i] = x;
                if (x != x || (smax == smax && x > smax)) smax = x;   /* Base.maximum propagates NaN */
            }
            if (smax == 0.0f) continue;                         /* gqi.jl:142 (NaN == 0 is false: not skipped) */
            for (int v = 0; v < nvert; v++) o[v] = 0.0f;        /* sgemv, gqi.jl:144 */
            for (int i = 0; i < nvol; i++) {
                const float *col = A + (int64_t)i * nvert;
                float si = s[i];
                for (int v = 0; v < nvert; v++) o[v] = fmaf(col[v], si, o[v]);
            }
            for (int v = 0; v < nvert; v++) odf[(int64_t)v * nvox + vox] = o[v];   /* gqi.jl:145 */
            write_peaks(o, nvert, faces, nfaces, verts, nverts_full, vox, nvox, peak, qa, pk, is, tmp);
        }
        free(s); free(o); free(pk); free(is); free(tmp);
    }
    float odfmax = odf_max_of_means(odf, nvox, nvert, nthreads);   /* gqi.jl:164 */
    qa_scale(qa, nvox, odfmax, nthreads);                           /* gqi.jl:166-168 */
    return odfmax;
}

/* ------------------------------------------------------------------------ */
/* DSI (dsi.jl:171-270)                                                      */
/* ------------------------------------------------------------------------ */

typedef struct { float re, im; } cf32;

/* in-place radix-2 DIT forward DFT of length 16 with stride (FFTW forward sign, unnormalised) */
static void fft16(cf32 *x, int stride, const cf32 *tw /* tw[k] = exp(-2 pi i k / 16), k<8 */)
{
    static const int rev[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
    for (int i = 0; i < 16; i++) if (rev[i] > i) {
        cf32 t = x[i * stride]; x[i * stride] = x[rev[i] * stride]; x[rev[i] * stride] = t;
    }
    for (int len = 2; len <= 16; len *= 2) {
        int half = len / 2, step = 16 / len;
        for (int base = 0; base < 16; base += len)
            for (int k = 0; k < half; k++) {
                cf32 w = tw[k * step];
                cf32 *a = &x[(base + k) * stride], *b = &x[(base + k + half) * stride];
                cf32 t = {b->re * w.re - b->im * w.im, b->re * w.im + b->im * w.re};
                b->re = a->re - t.re; b->im = a->im - t.im;
                a->re = a->re + t.re; a->im = a->im + t.im;
            }
    }
}

static void fft3_16(cf32 *x, const cf32 *tw)
{
    for (int z = 0; z < 16; z++) for (int y = 0; y < 16; y++) fft16(x + 16 * y + 256 * z, 1, tw);
    for (int z = 0; z < 16; z++) for (int xx = 0; xx < 16; xx++) fft16(x + xx + 256 * z, 16, tw);
    for (int y = 0; y < 16; y++) for (int xx = 0; xx < 16; xx++) fft16(x + xx + 16 * y, 256, tw);
}

/* Julia sum(::Array{Float32}) is pairwise with a 1024-element base case */
static float pairwise_sum(const float *a, int n)
{
    if (n <= 1024) { float s = a[0]; for (int i = 1; i < n; i++) s += a[i]; return s; }
    int h = n / 2;
    return pairwise_sum(a, h) + pairwise_sum(a + h, n - h);
}

/* BSpline(Linear()) on 1-based knots of a 16^3 array p; coordinates are 1-based floats */
static float trilinear(const float *p, float x, float y, float z)
{
    int ix = (int)floorf(x), iy = (int)floorf(y), iz = (int)floorf(z);
    if (ix < 1) ix = 1; if (ix > 15) ix = 15;
    if (iy < 1) iy = 1; if (iy > 15) iy = 15;
    if (iz < 1) iz = 1; if (iz > 15) iz = 15;
    float fx = x - (float)ix, fy = y - (float)iy, fz = z - (float)iz;
    const float *q = p + (ix - 1) + 16 * (iy - 1) + 256 * (iz - 1);
    float c00 = (1.0f - fx) * q[0] + fx * q[1];
    float c10 = (1.0f - fx) * q[16] + fx * q[17];
    float c01 = (1.0f - fx) * q[256] + fx * q[257];
    float c11 = (1.0f - fx) * q[272] + fx * q[273];
    float c0 = (1.0f - fy) * c00 + fy * c10;
    float c1 = (1.0f - fy) * c01 + fy * c11;
    return (1.0f - fz) * c0 + fz * c1;
}

/*
 * iq_ind [nvol] 0-based linear index into the 16^3 grid, H [4096] window,
 * interp [3 x nrad x nvert] column-major 1-based coordinates (dsi.jl:106-109),
 * qr2 [nrad], dqr.  Only nfft == 16 is supported (the 515-point lattice; dsi.jl:70-71).
 */
float orc_dsi_rec(const float *dwi, const uint8_t *mask, int nx, int ny, int nz, int nvol,
                  const int32_t *iq_ind, const float *H, const float *interp, int nrad,
                  const float *qr2, float dqr,
                  int nvert, const int32_t *faces, int nfaces, const float *verts, int nverts_full,
                  float *pdf, float *odf, float *peak0, float *peak1, float *peak2,
                  float *qa0, float *qa1, float *qa2, int nthreads)
{
    const int64_t nxy = (int64_t)nx * ny, nvox = nxy * nz;
    float *peak[3] = {peak0, peak1, peak2}, *qa[3] = {qa0, qa1, qa2};
    cf32 tw[8];
    for (int k = 0; k < 8; k++) {
        tw[k].re = (float)cos(-2.0 * M_PI * k / 16.0);
        tw[k].im = (float)sin(-2.0 * M_PI * k / 16.0);
    }
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        int z0, z1;
        zslab(nz, omp_get_num_threads(), omp_get_thread_num(), &z0, &z1);
        float *X = (float *)calloc(4096, sizeof(float));       /* persists per thread, dsi.jl:88 */
        cf32 *x = (cf32 *)malloc(sizeof(cf32) * 4096), *xt = (cf32 *)malloc(sizeof(cf32) * 4096);
        float *p = (float *)malloc(sizeof(float) * 4096);
        float *o = (float *)malloc(sizeof(float) * nvert), *pk = (float *)malloc(sizeof(float) * nvert);
        int32_t *is = (int32_t *)malloc(sizeof(int32_t) * nvert), *tmp = (int32_t *)malloc(sizeof(int32_t) * nvert);
        for (int64_t vox = z0 * nxy; vox < z1 * nxy; vox++) {
            if (mask[vox] == 0) continue;                       /* dsi.jl:200 */
            for (int i = 0; i < nvol; i++) X[iq_ind[i]] = dwi[(int64_t)i * nvox + vox];   /* :205 */
            float xmax = X[0];
            for (int i = 1; i < 4096; i++) if (X[i] != X[i] || (xmax == xmax && X[i] > xmax)) xmax = X[i];   /* maximum propagates NaN */
            if (xmax == 0.0f) continue;                         /* dsi.jl:207 */
            for (int i = 0; i < 4096; i++) { float t = X[i] > 0.0f ? X[i] : (X[i] != X[i] ? X[i] : 0.0f); X[i] = t * H[i]; }   /* max(NaN, 0) is NaN in Julia */  /* :209-212 */
            /* circshift!(x, X, (8,8,8)); xtmp = F*x; circshift!(x, xtmp, (8,8,8))   dsi.jl:218-220 */
            for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) for (int i = 0; i < 16; i++) {
                int src = i + 16 * j + 256 * k;
                int dst = ((i + 8) & 15) + 16 * ((j + 8) & 15) + 256 * ((k + 8) & 15);
                xt[dst].re = X[src]; xt[dst].im = 0.0f;
            }
            fft3_16(xt, tw);
            for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) for (int i = 0; i < 16; i++) {
                int src = i + 16 * j + 256 * k;
                int dst = ((i + 8) & 15) + 16 * ((j + 8) & 15) + 256 * ((k + 8) & 15);
                x[dst] = xt[src];
            }
            for (int i = 0; i < 4096; i++) p[i] = x[i].re;      /* dsi.jl:224 */
            float psum = pairwise_sum(p, 4096);
            for (int i = 0; i < 4096; i++) p[i] = p[i] / psum;  /* dsi.jl:225 */
            for (int i = 0; i < nvol; i++) pdf[(int64_t)i * nvox + vox] = p[iq_ind[i]];   /* :227 */
            for (int v = 0; v < nvert; v++) {                   /* dsi.jl:233-242 */
                float acc = 0.0f;
                for (int r = 0; r < nrad; r++) {
                    const float *c = interp + 3 * (r + (int64_t)nrad * v);
                    acc += trilinear(p, c[0], c[1], c[2]) * qr2[r];
                }
                o[v] = acc * dqr;
            }
            for (int v = 0; v < nvert; v++) odf[(int64_t)v * nvox + vox] = o[v];   /* dsi.jl:246 */
            write_peaks(o, nvert, faces, nfaces, verts, nverts_full, vox, nvox, peak, qa, pk, is, tmp);
        }
        free(X); free(x); free(xt); free(p); free(o); free(pk); free(is); free(tmp);
    }
    float odfmax = odf_max_of_means(odf, nvox, nvert, nthreads);   /* dsi.jl:263 */
    qa_scale(qa, nvox, odfmax, nthreads);                           /* dsi.jl:265-267 */
    return odfmax;
}

/* ------------------------------------------------------------------------ */
/* Streamlines (stream.jl:340-374, 501-541, 625-690, 730-790)                */
/* ------------------------------------------------------------------------ */

typedef struct {
    int nx, ny, nz, nvec;
    const float *ovecs;     /* [3, nvec, nx, ny, nz] column-major, masked vectors zeroed (stream.jl:141-145) */
    const uint8_t *mask;    /* [nx,ny,nz] (stream.jl:95-116) */
    int len_min, len_max;
    float cosang_thresh, step_size, smooth_coeff;
    /* microscopy regime (stream.jl:252-287): search_dist > 0 */
    int search_dist;
    float search_cosang;        /* cosd(search_ang) */
    const float *search_area;   /* [3, Sx, Sy, Sz], S = 2*sd+1 per axis: unit vectors of the cells with rho < 1, else 0 */
    /* LCM-guided tracking (stream.jl:200-236, 380-495): lcms != NULL */
    const float *lcms;          /* [10, nx, ny, nz] column-major, already thresholded (stream.jl:217) */
    int strdims[2];             /* in-plane dimensions, 0-based (stream.jl:221-223) */
    uint64_t rng_seed;          /* uniform draws: orc_uniform(rng_seed, line, k) stands in for Julia's global RNG */
    int sd[3];                  /* micro_search_dist per axis: search_dist, or 0 along the through-plane axis of 2-D angle inputs (stream.jl:153-155) */
} stream_work;

/* The random-number contract of LCM-guided tracking.  The reference draws `rand(Categorical(lcm))` from Julia's global
 * RNG (stream.jl:455), which cannot be reproduced; this back end (HIP kernel and oracle alike) takes the k-th uniform
 * of streamline `line` from a counter-based generator: splitmix64 of (seed, line, k), top 24 bits -> [0,1). */
static uint64_t orc_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
float orc_uniform(uint64_t seed, uint64_t line, uint32_t k)
{
    uint64_t h = orc_splitmix64(seed ^ orc_splitmix64(line * 0xD1342543DE82EF95ull + (uint64_t)k));
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}

/* LinearAlgebra.norm of a 3-vector: squares in Float32, sum + sqrt in Float64 (generic_norm2) */
static float norm3(const float v[3])
{
    float m = fmaxf(fabsf(v[0]), fmaxf(fabsf(v[1]), fabsf(v[2])));
    if (m == 0.0f || isinf(m)) return m;
    double s = (double)(v[0] * v[0]);
    s += (double)(v[1] * v[1]);
    s += (double)(v[2] * v[2]);
    return (float)sqrt(s);
}

static float dot3(const float a[3], const float b[3])
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* stream_pick_by_angle! (stream.jl:340-374); voxel coordinates 1-based */
static int pick_by_angle(const stream_work *W, int ix, int iy, int iz, const float vec_now[3],
                         float vec_next[3], int *ivec_next)
{
    const float *v0 = W->ovecs + 3 * (int64_t)W->nvec *
                      ((ix - 1) + (int64_t)W->nx * ((iy - 1) + (int64_t)W->ny * (iz - 1)));
    int best = 0; float bestabs = 0.0f, bestcos = 0.0f;
    for (int k = 0; k < W->nvec; k++) {
        const float *v = v0 + 3 * k;
        float c, ca;
        if (v[0] == 0.0f && v[1] == 0.0f && v[2] == 0.0f) c = ca = -INFINITY;   /* :353-354 */
        else { c = dot3(vec_now, v); ca = fabsf(c); }                             /* :356-357 */
        /* argmax: first maximum; NaN beats everything (Base.argmax) */
        if (k == 0 || (!isnan(bestabs) && (isnan(ca) || ca > bestabs))) { best = k; bestabs = ca; bestcos = c; }
    }
    if (!isfinite(bestcos)) return 0;                                             /* :363 */
    const float *v = v0 + 3 * best;
    if (bestcos > 0.0f) { vec_next[0] = v[0]; vec_next[1] = v[1]; vec_next[2] = v[2]; }
    else { vec_next[0] = -v[0]; vec_next[1] = -v[1]; vec_next[2] = -v[2]; }        /* :365-369 */
    *ivec_next = best;                                                             /* :371 */
    return 1;
}

/* voxel edges connected by the i-th element of a vectorised LCM (stream.jl:228-229), 1-based edge types */
static const int lcm_edge[2][10] = {{1, 1, 1, 1, 2, 2, 2, 3, 3, 4}, {1, 2, 3, 4, 2, 3, 4, 3, 4, 4}};

/* coordinate increments for exiting through the j-th edge (stream.jl:224-226), j = 1..4 */
static void lcm_dxyz(const stream_work *W, int j, int d[3])
{
    static const int a[4] = {-1, 0, 1, 0}, b[4] = {0, -1, 0, 1};
    d[0] = d[1] = d[2] = 0;
    d[W->strdims[0]] = a[j - 1];
    d[W->strdims[1]] = b[j - 1];
}

static int lcm_match_edge(const stream_work *W, const int dv[3])
{
    for (int j = 1; j <= 4; j++) {
        int d[3];
        lcm_dxyz(W, j, d);
        if (d[0] == dv[0] && d[1] == dv[1] && d[2] == dv[2]) return j;
    }
    return 0;
}

/* stream_pick_by_lcm! (stream.jl:380-495).  *ivec_next is W.ivec_next (on entry: what stream_pick_by_angle! just chose,
 * stream.jl:530-531).  *ndraw counts the uniforms this line has consumed. */
static int pick_by_lcm(const stream_work *W, int ix, int iy, int iz, const float pos_now[3], const float pos_next[3],
                       const float vec_now[3], float vec_next[3], int *ivec_next, uint64_t line, uint32_t *ndraw)
{
    const int64_t vox = (ix - 1) + (int64_t)W->nx * ((iy - 1) + (int64_t)W->ny * (iz - 1));
    const float *v0 = W->ovecs + 3 * (int64_t)W->nvec * vox;
    int dv[3] = {(int)rintf(pos_now[0]) - ix, (int)rintf(pos_now[1]) - iy, (int)rintf(pos_now[2]) - iz};   /* :394-398 */
    if (dv[0] == 0 && dv[1] == 0 && dv[2] == 0) {                /* not entering a new voxel, :400-413 */
        const float *v = v0 + 3 * *ivec_next;
        if (dot3(vec_now, v) > 0.0f) { vec_next[0] = v[0]; vec_next[1] = v[1]; vec_next[2] = v[2]; }
        else { vec_next[0] = -v[0]; vec_next[1] = -v[1]; vec_next[2] = -v[2]; }
        return 1;
    }
    int entry = lcm_match_edge(W, dv);                           /* :416-422 */
    if (entry == 0) {                                            /* a diagonal jump: the dimension that changes faster, :424-438 */
        const int s1 = W->strdims[0], s2 = W->strdims[1];
        if (fabsf(pos_now[s1] - pos_next[s1]) < fabsf(pos_now[s2] - pos_next[s2])) dv[s2] = 0; else dv[s1] = 0;
        entry = lcm_match_edge(W, dv);
    }
    float lcm[10];
    int any = 0;
    for (int j = 0; j < 10; j++) {                               /* :441-446 */
        lcm[j] = W->lcms[j + 10 * vox];
        if (!(lcm_edge[0][j] == entry || lcm_edge[1][j] == entry)) lcm[j] = 0.0f;
        any |= lcm[j] != 0.0f;
    }
    if (!any) return 0;                                          /* while !iszero(lcm) ... iszero(lcm) && return false, :448,492 */
    float sum = lcm[0];
    for (int j = 1; j < 10; j++) sum += lcm[j];
    for (int j = 0; j < 10; j++) lcm[j] = lcm[j] / sum;          /* :450 */
    const float u = orc_uniform(W->rng_seed, line, (*ndraw)++);
    int il = 0;                                                  /* rand(Categorical(lcm)): first index whose running sum exceeds u */
    float cp = lcm[0];
    while (cp <= u && il < 9) cp += lcm[++il];
    const int exitedge = lcm_edge[0][il] == entry ? lcm_edge[1][il] : lcm_edge[0][il];   /* :454-456 */
    int d[3];
    lcm_dxyz(W, exitedge, d);
    const float df[3] = {(float)d[0], (float)d[1], (float)d[2]};
    int best = 0; float bestabs = 0.0f, bestcos = 0.0f;
    for (int k = 0; k < W->nvec; k++) {                          /* :462-472 */
        const float *v = v0 + 3 * k;
        float c, ca;
        if (v[0] == 0.0f && v[1] == 0.0f && v[2] == 0.0f) c = ca = -INFINITY;
        else { c = dot3(df, v); ca = fabsf(c); }
        if (k == 0 || (!isnan(bestabs) && (isnan(ca) || ca > bestabs))) { best = k; bestabs = ca; bestcos = c; }
    }
    if (!isfinite(bestcos)) return 0;                            /* :476 */
    const float *v = v0 + 3 * best;
    if (bestcos > 0.0f) { vec_next[0] = v[0]; vec_next[1] = v[1]; vec_next[2] = v[2]; }
    else { vec_next[0] = -v[0]; vec_next[1] = -v[1]; vec_next[2] = -v[2]; }      /* :480-484 */
    *ivec_next = best;                                           /* :486 */
    return 1;
}

/* stream_new_point! with LCMs (stream.jl:526-538): the angle pick first (it can end the line and it sets
 * W.ivec_next), then the LCM pick; *isdiff = the two methods chose different vectors */
static int new_point_lcm(const stream_work *W, const float pos_now[3], const float vec_now[3], float pos_next[3],
                         float vec_next[3], int *ivec_next, int *isdiff, uint64_t line, uint32_t *ndraw)
{
    for (int c = 0; c < 3; c++) pos_next[c] = pos_now[c] + vec_now[c] * W->step_size;
    float rx = rintf(pos_next[0]), ry = rintf(pos_next[1]), rz = rintf(pos_next[2]);
    if (!(rx >= 1.0f && rx <= (float)W->nx && ry >= 1.0f && ry <= (float)W->ny &&
          rz >= 1.0f && rz <= (float)W->nz)) return 0;
    int ix = (int)rx, iy = (int)ry, iz = (int)rz;
    if (!W->mask[(ix - 1) + (int64_t)W->nx * ((iy - 1) + (int64_t)W->ny * (iz - 1))]) return 0;
    if (!pick_by_angle(W, ix, iy, iz, vec_now, vec_next, ivec_next)) return 0;   /* :530 */
    const int ivec_ang = *ivec_next;                                               /* :531 */
    if (!pick_by_lcm(W, ix, iy, iz, pos_now, pos_next, vec_now, vec_next, ivec_next, line, ndraw)) return 0;   /* :535 */
    *isdiff = *ivec_next != ivec_ang;                                              /* :538 */
    return 1;
}

/* stream_new_point! (stream.jl:501-541), non-LCM */
static int new_point(const stream_work *W, const float pos_now[3], const float vec_now[3],
                     float pos_next[3], float vec_next[3], int *ivec_next)
{
    for (int c = 0; c < 3; c++) pos_next[c] = pos_now[c] + vec_now[c] * W->step_size;   /* :512 */
    float rx = rintf(pos_next[0]), ry = rintf(pos_next[1]), rz = rintf(pos_next[2]);    /* :514 ties-to-even */
    if (!(rx >= 1.0f && rx <= (float)W->nx && ry >= 1.0f && ry <= (float)W->ny &&
          rz >= 1.0f && rz <= (float)W->nz)) return 0;                                    /* :517 */
    int ix = (int)rx, iy = (int)ry, iz = (int)rz;
    if (!W->mask[(ix - 1) + (int64_t)W->nx * ((iy - 1) + (int64_t)W->ny * (iz - 1))]) return 0;  /* :520 */
    return pick_by_angle(W, ix, iy, iz, vec_now, vec_next, ivec_next);
}

/* search_area of the microscopy regime (stream.jl:255-277), all arithmetic in Float32 like the reference's T */
static float *micro_search_area(const int d[3])
{
    const int Sx = 2 * d[0] + 1, Sy = 2 * d[1] + 1, Sz = 2 * d[2] + 1;
    float *sa = (float *)calloc((size_t)3 * Sx * Sy * Sz, sizeof(float));
    for (int iz = 1; iz <= Sz; iz++)
        for (int iy = 1; iy <= Sy; iy++)
            for (int ix = 1; ix <= Sx; ix++) {
                const float rx = (float)(ix - d[0] - 1) / ((float)d[0] + 0.5f), ry = (float)(iy - d[1] - 1) / ((float)d[1] + 0.5f),
                            rz = (float)(iz - d[2] - 1) / ((float)d[2] + 0.5f);                       /* :262-267 */
                float q = rx * rx; q = q + ry * ry; q = q + rz * rz;
                const float r = sqrtf(q);
                float *v = sa + 3 * ((ix - 1) + (size_t)Sx * ((iy - 1) + (size_t)Sy * (iz - 1)));
                if (r < 1.0f) { v[0] = rx / r; v[1] = ry / r; v[2] = rz / r; }      /* centre: 0/0 = NaN, kept (see below) */
            }
    return sa;
}

/* stream_micro_new_point! (stream.jl:547-619): the next point is the voxel, within a cone of search_ang around the
 * current direction and search_dist voxels around the tentative position, whose (first) orientation vector is best
 * aligned with the current direction.  The centre cell's search vector is NaN (0/0), which passes both `iszero` and
 * the `<=` cone test, so the tentative voxel itself is always a candidate. */
static int micro_new_point(const stream_work *W, const float pos_now[3], const float vec_now[3],
                           float pos_next[3], float vec_next[3])
{
    for (int c = 0; c < 3; c++) pos_next[c] = pos_now[c] + vec_now[c] * W->step_size;   /* :561 */
    float rx = rintf(pos_next[0]), ry = rintf(pos_next[1]), rz = rintf(pos_next[2]);    /* :563 */
    if (!(rx >= 1.0f && rx <= (float)W->nx && ry >= 1.0f && ry <= (float)W->ny &&
          rz >= 1.0f && rz <= (float)W->nz)) return 0;                                    /* :566 */
    const int cx = (int)rx, cy = (int)ry, cz = (int)rz;
    const int dx = W->sd[0], dy = W->sd[1], dz = W->sd[2], Sx = 2 * dx + 1, Sy = 2 * dy + 1, Sz = 2 * dz + 1;
    if (!W->mask[(cx - 1) + (int64_t)W->nx * ((cy - 1) + (int64_t)W->ny * (cz - 1))]) return 0;   /* :569 */
    int have = 0, bx = 0, by = 0, bz = 0;
    float bestabs = -INFINITY, bestcos = -INFINITY;
    /* argmax over the Sx x Sy x Sz array in column-major order: first maximum, NaN wins (Base.argmax) */
    for (int kz = 1; kz <= Sz; kz++)
        for (int ky = 1; ky <= Sy; ky++)
            for (int kx = 1; kx <= Sx; kx++) {
                const int ix = cx - dx + kx - 1, iy = cy - dy + ky - 1, iz = cz - dz + kz - 1;
                float ca = -INFINITY, c = -INFINITY;
                if (ix >= 1 && ix <= W->nx && iy >= 1 && iy <= W->ny && iz >= 1 && iz <= W->nz) {   /* :586-588 */
                    const float *v = W->search_area + 3 * ((kx - 1) + (size_t)Sx * ((ky - 1) + (size_t)Sy * (kz - 1)));
                    const int64_t lin = (ix - 1) + (int64_t)W->nx * ((iy - 1) + (int64_t)W->ny * (iz - 1));
                    const int zero = v[0] == 0.0f && v[1] == 0.0f && v[2] == 0.0f;
                    if (W->mask[lin] && !zero && !(dot3(vec_now, v) <= W->search_cosang)) {           /* :596-598 */
                        c = dot3(vec_now, W->ovecs + 3 * (int64_t)W->nvec * lin);                      /* :600 (vector 1) */
                        ca = fabsf(c);
                    }
                }
                if (!have || (!isnan(bestabs) && (isnan(ca) || ca > bestabs))) {
                    have = 1; bestabs = ca; bestcos = c; bx = ix; by = iy; bz = iz;
                }
            }
    if (!isfinite(bestcos)) return 0;                                                     /* :609 */
    pos_next[0] = (float)bx; pos_next[1] = (float)by; pos_next[2] = (float)bz;            /* :612-614 */
    const float *v = W->ovecs + 3 * (int64_t)W->nvec * ((bx - 1) + (int64_t)W->nx * ((by - 1) + (int64_t)W->ny * (bz - 1)));
    if (bestcos > 0.0f) { vec_next[0] = v[0]; vec_next[1] = v[1]; vec_next[2] = v[2]; }
    else { vec_next[0] = -v[0]; vec_next[1] = -v[1]; vec_next[2] = -v[2]; }               /* :616-620 */
    return 1;
}

/*
 * stream_new_line (stream.jl:625-690). `line` has room for 3*(len_max+2) floats and is
 * filled in REFERENCE ORDER [fwd_N..fwd_1, bwd_1..bwd_M]; returns npts, *nfwd = N.
 */
static int new_line(const stream_work *W, const int seed[3], const float sub[3], float *line, int *nfwd,
                    float *fwdbuf, uint64_t lineno, uint8_t *flags)
{
    int npts = 0, ivec_next = 0, nf = 0, nb = 0;                 /* :638, :645 (0-based here) */
    uint32_t ndraw = 0;
    uint8_t *ffwd = flags ? flags + (W->len_max + 2) : NULL, *fbwd = flags ? flags + 2 * (W->len_max + 2) : NULL;
    float *bwd = line;                                           /* assembled after both passes */
    float *bwdbuf = fwdbuf + 3 * (W->len_max + 2);
    (void)bwd;
    for (int pass = 0; pass < 2; pass++) {
        float fwd = pass == 0 ? 1.0f : -1.0f;
        float pos_now[3], vec_now[3], pos_next[3], vec_next[3];
        const float *sv = W->ovecs + 3 * ((int64_t)ivec_next + (int64_t)W->nvec *
                          ((seed[0] - 1) + (int64_t)W->nx * ((seed[1] - 1) + (int64_t)W->ny * (seed[2] - 1))));
        for (int c = 0; c < 3; c++) {
            pos_now[c] = (float)seed[c] + sub[c];                /* :649 */
            vec_now[c] = sv[c] * fwd;                            /* :650 */
        }
        for (;;) {
            int isdiff = 0;
            if (!(W->lcms ? new_point_lcm(W, pos_now, vec_now, pos_next, vec_next, &ivec_next, &isdiff, lineno, &ndraw)
                  : W->search_dist > 0 ? micro_new_point(W, pos_now, vec_now, pos_next, vec_next)
                                       : new_point(W, pos_now, vec_now, pos_next, vec_next, &ivec_next))) break;   /* :655-657 */
            if (flags) { if (pass == 0) ffwd[nf] = (uint8_t)isdiff; else fbwd[nb] = (uint8_t)isdiff; }              /* :666 */
            float *dst = pass == 0 ? fwdbuf + 3 * nf++ : bwdbuf + 3 * nb++;                /* :660 */
            dst[0] = pos_now[0]; dst[1] = pos_now[1]; dst[2] = pos_now[2];
            npts++;                                              /* :661 */
            if (!W->lcms && dot3(vec_now, vec_next) < W->cosang_thresh) break;   /* :670 (not used with LCMs, :668) */
            if (npts > W->len_max) break;                        /* :674 */
            if (W->smooth_coeff != 0.0f) {                       /* :677-681 */
                float omc = 1.0f - W->smooth_coeff;
                for (int c = 0; c < 3; c++) vec_next[c] = W->smooth_coeff * vec_now[c] + omc * vec_next[c];
                float n = norm3(vec_next);
                for (int c = 0; c < 3; c++) vec_next[c] = vec_next[c] / n;
            }
            for (int c = 0; c < 3; c++) { pos_now[c] = pos_next[c]; vec_now[c] = vec_next[c]; }   /* :684-685 */
        }
    }
    /* prepend! for forward points reverses them; append! keeps backward order (:652) */
    for (int i = 0; i < nf; i++) memcpy(line + 3 * i, fwdbuf + 3 * (nf - 1 - i), 3 * sizeof(float));
    memcpy(line + 3 * nf, bwdbuf, 3 * sizeof(float) * nb);
    if (flags) {                                                 /* same order as the points */
        for (int i = 0; i < nf; i++) flags[i] = ffwd[nf - 1 - i];
        for (int i = 0; i < nb; i++) flags[nf + i] = fbwd[i];
    }
    *nfwd = nf;
    return npts;
}

/*
 * stream driver (stream.jl:730-790).  seeds [nseed x 3] 1-based voxel coordinates in
 * the reference's findall order, sublist [nsub x 3].  Results are returned in
 * (seed, sub) order == the reference's thread-chunk order under static scheduling.
 * Two-call protocol: the function allocates; caller frees with orc_free.
 * out_npts[nlines], out_seed[nlines] (index into seeds*nsub: seed*nsub+sub), out_xyz[3*total].
 * all_npts (optional, [nseed*nsub]) receives npts of every line incl. dropped ones.
 */
int64_t orc_stream_micro(const float *ovecs, const uint8_t *mask, int nx, int ny, int nz, int nvec,
                         const int32_t *seeds, int64_t nseed, const float *sublist, int nsub,
                         int len_min, int len_max, float cosang_thresh, float step_size, float smooth_coeff,
                         int search_dist, float search_cosang,
                         int32_t **out_npts, int64_t **out_seed, float **out_xyz, int64_t *out_total_pts,
                         int32_t *all_npts, int nthreads);

int64_t orc_stream(const float *ovecs, const uint8_t *mask, int nx, int ny, int nz, int nvec,
                   const int32_t *seeds, int64_t nseed, const float *sublist, int nsub,
                   int len_min, int len_max, float cosang_thresh, float step_size, float smooth_coeff,
                   int32_t **out_npts, int64_t **out_seed, float **out_xyz, int64_t *out_total_pts,
                   int32_t *all_npts, int nthreads)
{
    return orc_stream_micro(ovecs, mask, nx, ny, nz, nvec, seeds, nseed, sublist, nsub, len_min, len_max, cosang_thresh,
                            step_size, smooth_coeff, 0, 0.0f, out_npts, out_seed, out_xyz, out_total_pts, all_npts, nthreads);
}

int64_t orc_stream_full(const float *ovecs, const uint8_t *mask, int nx, int ny, int nz, int nvec,
                        const int32_t *seeds, int64_t nseed, const float *sublist, int nsub,
                        int len_min, int len_max, float cosang_thresh, float step_size, float smooth_coeff,
                        int search_dist, float search_cosang,
                        const float *lcms, int strdim0, int strdim1, uint64_t rng_seed,
                        int32_t **out_npts, int64_t **out_seed, float **out_xyz, uint8_t **out_flags, int64_t *out_total_pts,
                        int32_t *all_npts, int nthreads);

int64_t orc_stream_flat(const float *ovecs, const uint8_t *mask, int nx, int ny, int nz, int nvec,
                        const int32_t *seeds, int64_t nseed, const float *sublist, int nsub,
                        int len_min, int len_max, float cosang_thresh, float step_size, float smooth_coeff,
                        int search_dist, int search_flat, float search_cosang,
                        const float *lcms, int strdim0, int strdim1, uint64_t rng_seed,
                        int32_t **out_npts, int64_t **out_seed, float **out_xyz, uint8_t **out_flags, int64_t *out_total_pts,
                        int32_t *all_npts, int nthreads);

/* same driver; search_dist > 0 selects the microscopy regime (stream.jl:83, 252-287, 547-619) */
int64_t orc_stream_micro(const float *ovecs, const uint8_t *mask, int nx, int ny, int nz, int nvec,
                         const int32_t *seeds, int64_t nseed, const float *sublist, int nsub,
                         int len_min, int len_max, float cosang_thresh, float step_size, float smooth_coeff,
                         int search_dist, float search_cosang,
                         int32_t **out_npts, int64_t **out_seed, float **out_xyz, int64_t *out_total_pts,
                         int32_t *all_npts, int nthreads)
{
    return orc_stream_full(ovecs, mask, nx, ny, nz, nvec, seeds, nseed, sublist, nsub, len_min, len_max, cosang_thresh,
                           step_size, smooth_coeff, search_dist, search_cosang, NULL, 0, 1, 0, out_npts, out_seed, out_xyz,
                           NULL, out_total_pts, all_npts, nthreads);
}

/* the general driver.  lcms != NULL: LCM-guided tracking (stream.jl:200-236, 380-495): lcms [10,nx,ny,nz] already
 * thresholded, strdim0/1 the in-plane dimensions (0-based), rng_seed the seed of the uniform stream (orc_uniform);
 * out_flags (one byte per point, in point order) = "the LCM and the angle pick chose different vectors" (stream.jl:538) */
int64_t orc_stream_full(const float *ovecs, const uint8_t *mask, int nx, int ny, int nz, int nvec,
                        const int32_t *seeds, int64_t nseed, const float *sublist, int nsub,
                        int len_min, int len_max, float cosang_thresh, float step_size, float smooth_coeff,
                        int search_dist, float search_cosang,
                        const float *lcms, int strdim0, int strdim1, uint64_t rng_seed,
                        int32_t **out_npts, int64_t **out_seed, float **out_xyz, uint8_t **out_flags, int64_t *out_total_pts,
                        int32_t *all_npts, int nthreads)
{
    return orc_stream_flat(ovecs, mask, nx, ny, nz, nvec, seeds, nseed, sublist, nsub, len_min, len_max, cosang_thresh, step_size,
                           smooth_coeff, search_dist, -1, search_cosang, lcms, strdim0, strdim1, rng_seed, out_npts, out_seed, out_xyz,
                           out_flags, out_total_pts, all_npts, nthreads);
}

/* .. with the search distance of one axis set to 0 (search_flat = 0..2; -1: none): what StreamWork does to the through-plane
 * axis of 2-D orientation-angle inputs in the microscopy regime (stream.jl:153-155) */
int64_t orc_stream_flat(const float *ovecs, const uint8_t *mask, int nx, int ny, int nz, int nvec,
                        const int32_t *seeds, int64_t nseed, const float *sublist, int nsub,
                        int len_min, int len_max, float cosang_thresh, float step_size, float smooth_coeff,
                        int search_dist, int search_flat, float search_cosang,
                        const float *lcms, int strdim0, int strdim1, uint64_t rng_seed,
                        int32_t **out_npts, int64_t **out_seed, float **out_xyz, uint8_t **out_flags, int64_t *out_total_pts,
                        int32_t *all_npts, int nthreads)
{
    int sd3[3] = {search_dist, search_dist, search_dist};
    if (search_flat >= 0 && search_flat < 3) sd3[search_flat] = 0;
    float *sarea = search_dist > 0 ? micro_search_area(sd3) : NULL;
    stream_work W = {nx, ny, nz, nvec, ovecs, mask, len_min, len_max, cosang_thresh, step_size, smooth_coeff,
                     search_dist, search_cosang, sarea, lcms, {strdim0, strdim1}, rng_seed, {sd3[0], sd3[1], sd3[2]}};
    const int want_flags = lcms != NULL && out_flags != NULL;
    if (nthreads < 1) nthreads = 1;
    /* chunks of div(n, nthreads)+1 seeds (stream.jl:757-759) */
    int64_t per = nseed / nthreads + 1;
    int nchunks = (int)((nseed + per - 1) / per);
    float **cx = (float **)calloc(nchunks > 0 ? nchunks : 1, sizeof(float *));
    uint8_t **cf = (uint8_t **)calloc(nchunks > 0 ? nchunks : 1, sizeof(uint8_t *));
    int32_t **cn = (int32_t **)calloc(nchunks > 0 ? nchunks : 1, sizeof(int32_t *));
    int64_t **cs = (int64_t **)calloc(nchunks > 0 ? nchunks : 1, sizeof(int64_t *));
    int64_t *clines = (int64_t *)calloc(nchunks > 0 ? nchunks : 1, sizeof(int64_t));
    int64_t *cpts = (int64_t *)calloc(nchunks > 0 ? nchunks : 1, sizeof(int64_t));
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int ic = 0; ic < nchunks; ic++) {
        int64_t i0 = ic * per, i1 = i0 + per < nseed ? i0 + per : nseed;
        int64_t capl = 1024, capp = 1 << 16, nl = 0, np = 0;
        float *xyz = (float *)malloc(sizeof(float) * 3 * capp);
        uint8_t *fl = (uint8_t *)malloc(capp);
        int32_t *npl = (int32_t *)malloc(sizeof(int32_t) * capl);
        int64_t *sd = (int64_t *)malloc(sizeof(int64_t) * capl);
        float *line = (float *)malloc(sizeof(float) * 3 * (len_max + 2));
        float *scratch = (float *)malloc(sizeof(float) * 6 * (len_max + 2));
        uint8_t *lflags = (uint8_t *)malloc(3 * (size_t)(len_max + 2));
        for (int64_t is = i0; is < i1; is++) {
            for (int isub = 0; isub < nsub; isub++) {
                int nfwd;
                int n = new_line(&W, seeds + 3 * is, sublist + 3 * isub, line, &nfwd, scratch,
                                 (uint64_t)(is * nsub + isub), lcms ? lflags : NULL);
                if (all_npts) all_npts[is * nsub + isub] = n;
                if (n < len_min) continue;                       /* :769 */
                if (nl == capl) { capl *= 2; npl = realloc(npl, sizeof(int32_t) * capl); sd = realloc(sd, sizeof(int64_t) * capl); }
                while (np + n > capp) { capp *= 2; xyz = realloc(xyz, sizeof(float) * 3 * capp); fl = realloc(fl, capp); }
                memcpy(xyz + 3 * np, line, sizeof(float) * 3 * n);
                if (lcms) memcpy(fl + np, lflags, n);
                npl[nl] = n; sd[nl] = is * nsub + isub; nl++; np += n;
            }
        }
        free(line); free(scratch); free(lflags);
        cx[ic] = xyz; cf[ic] = fl; cn[ic] = npl; cs[ic] = sd; clines[ic] = nl; cpts[ic] = np;
    }
    int64_t nl = 0, np = 0;
    for (int ic = 0; ic < nchunks; ic++) { nl += clines[ic]; np += cpts[ic]; }
    *out_npts = (int32_t *)malloc(sizeof(int32_t) * (nl > 0 ? nl : 1));
    *out_seed = (int64_t *)malloc(sizeof(int64_t) * (nl > 0 ? nl : 1));
    *out_xyz = (float *)malloc(sizeof(float) * 3 * (np > 0 ? np : 1));
    if (want_flags) *out_flags = (uint8_t *)malloc(np > 0 ? np : 1);
    int64_t ol = 0, op = 0;
    for (int ic = 0; ic < nchunks; ic++) {                       /* reduce(vcat, W.str)  :787 */
        memcpy(*out_npts + ol, cn[ic], sizeof(int32_t) * clines[ic]);
        memcpy(*out_seed + ol, cs[ic], sizeof(int64_t) * clines[ic]);
        memcpy(*out_xyz + 3 * op, cx[ic], sizeof(float) * 3 * cpts[ic]);
        if (want_flags) memcpy(*out_flags + op, cf[ic], cpts[ic]);
        ol += clines[ic]; op += cpts[ic];
        free(cx[ic]); free(cf[ic]); free(cn[ic]); free(cs[ic]);
    }
    free(cx); free(cf); free(cn); free(cs); free(clines); free(cpts);
    free(sarea);
    *out_total_pts = np;
    return nl;
}

void orc_free(void *p) { free(p); }

int orc_max_threads(void) { return omp_get_max_threads(); }
