/* abi_caller.c -- the drop-in boundary driven from plain C: no Python, no torch, nothing but include/fibers_hip.h and libfibers_hip.so
 * (built and run by tests/test_gpu_abi_caller.py on the GPU box).  What the reference-side binding does through `ccall`, in C:
 *   fib_dti_fit  (dti_fit, dti.jl:221)   noise-free single-tensor signals s = S0 exp(-b g'Dg) -> the fit must return D's eigenvalues, FA, S0
 *   fib_adc_fit  (adc_fit, dti.jl:164)   mono-exponential decay -> ADC and S0
 *   fib_stream   (stream, stream.jl:730) a uniform field along x, every voxel a seed -> straight lines along x through the whole row
 *   errors       a missing b-table -> FIB_ERR_MISSING_BVAL with the reference's message (dti.jl:223-229)
 * Known answers only: the oracle is not involved. */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/fibers_hip.h"

#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "abi_caller: "); fprintf(stderr, __VA_ARGS__); fprintf(stderr, " (%s:%d)\n", __FILE__, __LINE__); return 1; } } while (0)

int main(void) {
    CHECK(fib_device_count() > 0, "no device");
    CHECK(strncmp(fib_version(), "fibers-hip", 10) == 0, "version %s", fib_version());
    enum { NX = 9, NY = 7, NZ = 5, NDIR = 12, NVOL = NDIR + 2 };
    const int64_t nvox = (int64_t)NX * NY * NZ;
    /* 12 directions: the icosahedron's vertices; two b = 0 frames in front */
    const double t = (1.0 + sqrt(5.0)) / 2.0, nrm = sqrt(1.0 + t * t);
    const double ico[NDIR][3] = {{0, 1, t}, {0, -1, t}, {0, 1, -t}, {0, -1, -t}, {1, t, 0}, {-1, t, 0}, {1, -t, 0}, {-1, -t, 0}, {t, 0, 1}, {t, 0, -1}, {-t, 0, 1}, {-t, 0, -1}};
    float bval[NVOL], bvec[3 * NVOL];                                   /* bvec [nvol x 3] column-major */
    for (int i = 0; i < NVOL; i++) {
        bval[i] = i < 2 ? 0.0f : 1000.0f;
        for (int c = 0; c < 3; c++) bvec[i + c * NVOL] = i < 2 ? (c == 0 ? 1.0f : 0.0f) : (float)(ico[i - 2][c] / nrm);
    }
    /* D = R diag(1.7, 0.6, 0.3) 1e-3 R', R a rotation by 30 degrees about z then 20 about x */
    const double lam[3] = {1.7e-3, 0.6e-3, 0.3e-3}, a = 30.0 * M_PI / 180.0, b = 20.0 * M_PI / 180.0;
    const double R[3][3] = {{cos(a), -sin(a), 0}, {cos(b) * sin(a), cos(b) * cos(a), -sin(b)}, {sin(b) * sin(a), sin(b) * cos(a), cos(b)}};
    double D[3][3] = {{0}};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) D[i][j] += R[i][k] * lam[k] * R[j][k];
    float *dwi = (float *)malloc(sizeof(float) * nvox * NVOL);
    uint8_t *mask = (uint8_t *)malloc((size_t)nvox);
    for (int64_t v = 0; v < nvox; v++) {
        mask[v] = (v % 11) != 3;                                        /* a few voxels outside */
        const double s0 = 800.0 + (double)(v % 37) * 10.0;
        for (int i = 0; i < NVOL; i++) {
            double q = 0;
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) q += bvec[i + r * NVOL] * D[r][c] * bvec[i + c * NVOL];
            dwi[(int64_t)i * nvox + v] = (float)(s0 * exp(-(double)bval[i] * q));     /* planar [nvol][nvox] == MRI.vol[nx,ny,nz,nvol] */
        }
    }
    float *o = (float *)calloc((size_t)nvox * 16, sizeof(float));
    fib_dti_out out = {o, o + nvox, o + 2 * nvox, o + 3 * nvox, o + 4 * nvox, o + 7 * nvox, o + 10 * nvox, o + 13 * nvox, o + 14 * nvox, o + 15 * nvox};
    int rc = fib_dti_fit(0, dwi, NX, NY, NZ, NVOL, mask, FIB_U8 | FIB_MASK_OUTPUTS_ZEROED, bval, bvec, &out);
    CHECK(rc == FIB_OK, "fib_dti_fit: %d %s", rc, fib_last_error());
    const double md = (lam[0] + lam[1] + lam[2]) / 3.0;
    const double fa = sqrt(1.5 * ((lam[0] - md) * (lam[0] - md) + (lam[1] - md) * (lam[1] - md) + (lam[2] - md) * (lam[2] - md)) / (lam[0] * lam[0] + lam[1] * lam[1] + lam[2] * lam[2]));
    for (int64_t v = 0; v < nvox; v++) {
        if (!mask[v]) { for (int k = 0; k < 16; k++) CHECK(o[k * nvox + v] == 0.0f, "voxel %ld outside the mask is not zero", (long)v); continue; }
        const double s0 = 800.0 + (double)(v % 37) * 10.0;
        CHECK(fabs(out.s0[v] - s0) <= 1e-4 * s0, "s0[%ld] = %g, expected %g", (long)v, out.s0[v], s0);
        CHECK(fabs(out.eigval1[v] - lam[0]) <= 1e-7 + 1e-4 * lam[0] && fabs(out.eigval2[v] - lam[1]) <= 1e-7 + 1e-4 * lam[1] && fabs(out.eigval3[v] - lam[2]) <= 1e-7 + 1e-4 * lam[2],
              "eigenvalues of voxel %ld: %g %g %g", (long)v, out.eigval1[v], out.eigval2[v], out.eigval3[v]);
        CHECK(fabs(out.fa[v] - fa) <= 1e-4 && fabs(out.md[v] - md) <= 1e-7 + 1e-4 * md, "fa / md of voxel %ld: %g %g (expected %g %g)", (long)v, out.fa[v], out.md[v], fa, md);
        double dot = 0;                                                   /* eigvec1 = +- R[:,0] (components planar: [3][nvox]) */
        for (int c = 0; c < 3; c++) dot += out.eigvec1[c * nvox + v] * R[c][0];
        CHECK(fabs(fabs(dot) - 1.0) <= 1e-4, "eigvec1 of voxel %ld: |dot| = %g", (long)v, fabs(dot));
    }
    /* adc_fit: the same signals along one direction are mono-exponential per frame pair only; use a clean decay instead */
    float bv2[4] = {0.0f, 500.0f, 1000.0f, 2000.0f};
    float *dw2 = (float *)malloc(sizeof(float) * nvox * 4), *adc = (float *)calloc((size_t)nvox, 4), *s02 = (float *)calloc((size_t)nvox, 4);
    for (int64_t v = 0; v < nvox; v++) for (int i = 0; i < 4; i++) dw2[(int64_t)i * nvox + v] = (float)(1000.0 * exp(-(double)bv2[i] * (0.5e-3 + 1e-6 * (double)(v % 100))));
    rc = fib_adc_fit(0, dw2, NX, NY, NZ, 4, mask, FIB_U8, bv2, adc, s02);
    CHECK(rc == FIB_OK, "fib_adc_fit: %d %s", rc, fib_last_error());
    for (int64_t v = 0; v < nvox; v++) {
        if (!mask[v]) { CHECK(adc[v] == 0.0f && s02[v] == 0.0f, "adc outside the mask"); continue; }
        const double want = 0.5e-3 + 1e-6 * (double)(v % 100);
        CHECK(fabs(adc[v] - want) <= 1e-4 * want && fabs(s02[v] - 1000.0) <= 0.1, "adc[%ld] = %g s0 = %g (expected %g, 1000)", (long)v, adc[v], s02[v], want);
    }
    /* the reference's error for a missing b-table (dti.jl:223-225) */
    rc = fib_dti_fit(0, dwi, NX, NY, NZ, NVOL, mask, FIB_U8, NULL, bvec, &out);
    CHECK(rc == FIB_ERR_MISSING_BVAL && strstr(fib_last_error(), "Missing b-value table") != NULL, "missing b-table: %d %s", rc, fib_last_error());
    /* stream: a uniform field along x, all voxels inside, every voxel a seed, offset 0 */
    float *ov = (float *)calloc((size_t)nvox * 3, sizeof(float));
    for (int64_t v = 0; v < nvox; v++) ov[v] = 1.0f;                      /* planar [3][nvox]: x component 1 */
    uint8_t *ones = (uint8_t *)malloc((size_t)nvox);
    memset(ones, 1, (size_t)nvox);
    fib_stream_params prm;
    memset(&prm, 0, sizeof prm);
    prm.nx = NX; prm.ny = NY; prm.nz = NZ; prm.nvec = 1; prm.len_min = 3; prm.len_max = NX;
    prm.cosang_thresh = 0.70710677f; prm.step_size = 0.5f; prm.smooth_coeff = 0.2f;
    const float *ovp[1] = {ov};
    const float sub[3] = {0.0f, 0.0f, 0.0f};
    fib_tract_out tr;
    memset(&tr, 0, sizeof tr);
    rc = fib_stream(0, &prm, ovp, NULL, 0.03f, NULL, 0.1f, ones, FIB_U8, NULL, 0, sub, 1, &tr);
    CHECK(rc == FIB_OK, "fib_stream: %d %s", rc, fib_last_error());
    CHECK(tr.nlines == nvox, "%ld lines for %ld seeds", (long)tr.nlines, (long)nvox);
    int64_t p = 0;
    for (int64_t l = 0; l < tr.nlines; l++) {
        const int64_t seed = tr.seed_index[l];
        const int sy = (int)((seed / NX) % NY) + 1, sz = (int)(seed / ((int64_t)NX * NY)) + 1;          /* 1-based voxel coordinates, like pos_now (stream.jl:660) */
        CHECK(seed == l && tr.npts[l] >= 3 && tr.npts[l] <= prm.len_max + 2, "line %ld: seed %ld, %d points", (long)l, (long)seed, tr.npts[l]);
        for (int k = 0; k < tr.npts[l]; k++, p++) {
            CHECK(tr.xyz[3 * p + 1] == (float)sy && tr.xyz[3 * p + 2] == (float)sz, "line %ld leaves its row", (long)l);
            if (k) CHECK(fabsf(fabsf(tr.xyz[3 * p] - tr.xyz[3 * (p - 1)]) - 0.5f) < 1e-6f || tr.xyz[3 * p] == tr.xyz[3 * (p - 1)], "line %ld: step %g", (long)l, tr.xyz[3 * p] - tr.xyz[3 * (p - 1)]);
        }
    }
    CHECK(p == tr.npoints, "point count");
    fib_tract_free(&tr);
    CHECK(fib_trim() == FIB_OK, "fib_trim");
    fib_shutdown();
    printf("abi_caller ok: dti_fit, adc_fit, stream and the error path through the C ABI from plain C (%ld voxels)\n", (long)nvox);
    free(dwi); free(mask); free(o); free(dw2); free(adc); free(s02); free(ov); free(ones);
    return 0;
}
