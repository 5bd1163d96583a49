// rumba.hip — row N4: RUMBA-SD (rusd.jl), robust and unbiased model-based spherical deconvolution.
//
// rumba_rec (rusd.jl:419-636) is a whole-volume Richardson-Lucy iteration under a Rician / noncentral-chi likelihood
// with a total-variation prior.  Per iteration (rumba_sd_iterate!, rusd.jl:266-345), on matrices [rows x nmask]:
//     Iratio = I_n / I_{n-1} (signal .* dodf ./ sigma2)                        elementwise (Perron's continued fraction)
//     rl     = K' (signal .* Iratio) ./ (K' dodf + eps)                        two contractions  [ncomp x ndir] x [ndir x nmask]
//     tv     = 1 ./ (|1 - lambda div(grad f / |grad f|)| + eps)                13-point stencil per compartment
//     fodf   = max(fodf .* rl .* tv, 0)
//     dodf   = K fodf                                                          one contraction   [ndir x ncomp] x [ncomp x nmask]
//     sigma2 = clamp(sum_dir((signal^2 + dodf^2)/2 - sigma2 .* dodf_sig .* Iratio) / (n ndir)), lambda = max(mean sigma2, 1/900)
// The three contractions have the shape of the GQI reconstruction (small M and K, the voxel axis huge and contiguous),
// so they run on the same split-bf16 MFMA kernel (odf.hip, fib::matrix_plan_run: exact f32 products); everything else
// is one lane per masked voxel with the voxel axis contiguous (coalesced).  Matrices are stored planar [row][npad] with
// npad = nmask rounded up to 256 (padding columns carry zero signal and stay zero).
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr float EPS32 = 1.1920929e-07f;

struct RumbaDims { int ndir, ncomp, nvert; int64_t nmask, npad; int nx, ny, nz; };

// signal matrix (rusd.jl:444-464): row 0 = (mean low-b > 0), rows 1.. = clamp(max(dwi,0) / mean low-b), NaN -> 0, > 1 -> 1
__global__ __launch_bounds__(256) void rumba_signal_kernel(const float *__restrict__ dwi, int64_t nvox, const int32_t *__restrict__ ind,
                                                          const int32_t *__restrict__ b0_frames, int nb0, const int32_t *__restrict__ dw_frames,
                                                          RumbaDims d, float *__restrict__ sig) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.npad) return;
    if (c >= d.nmask) { for (int r = 0; r < d.ndir; r++) sig[(int64_t)r * d.npad + c] = 0.0f; return; }
    const int64_t v = ind[c];
    float s0 = 0.0f;
    for (int i = 0; i < nb0; i++) { const float x = dwi[(int64_t)b0_frames[i] * nvox + v]; s0 += x > 0.0f ? x : (x != x ? x : 0.0f); }
    s0 = s0 / (float)nb0;                                         // mean(max.(dwi[..., ib0], 0), dims=4)
    sig[c] = s0 > 0.0f ? 1.0f : 0.0f;
    for (int r = 1; r < d.ndir; r++) {
        const float x = dwi[(int64_t)dw_frames[r - 1] * nvox + v];
        float q = (x > 0.0f ? x : (x != x ? x : 0.0f)) / s0;
        if (q != q) q = 0.0f;                                     // signal_mat[isnan.(signal_mat)] .= 0
        if (q > 1.0f) q = 1.0f;                                   // signal_mat[signal_mat .> 1] .= 1
        sig[(int64_t)r * d.npad + c] = q;
    }
}

// besseli_ratio (rusd.jl:170-177), Float32 like the reference
__device__ __forceinline__ float besseli_ratio(float nu, float z) {
    const float a = 2.0f * nu;
    return z / ((a + z) - ((a + 1.0f) * z / (2.0f * z + (a + 1.0f) - ((a + 3.0f) * z / ((a + 2.0f) + 2.0f * z - ((a + 5.0f) * z / ((a + 3.0f) + 2.0f * z)))))));
}

// rumba_sd_initialize! (rusd.jl:241-259).  dodf_sig = signal .* dodf ./ sigma2 only ever feeds the Bessel ratio of the next
// iteration (rusd.jl:275), so it is not stored: Iratio and the first contraction's operand x = signal .* Iratio are
// produced right where dodf_sig is known (here and in rumba_noise_kernel).
__global__ __launch_bounds__(256) void rumba_init_kernel(RumbaDims d, float nu, const float *__restrict__ fodf0, const float *__restrict__ dodf0, float lam0,
                                                        const float *__restrict__ sig, float *__restrict__ fodf, float *__restrict__ dodf,
                                                        float *__restrict__ ir, float *__restrict__ x, float *__restrict__ tv, float *__restrict__ s2) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.npad) return;
    const bool live = c < d.nmask;
    for (int k = 0; k < d.ncomp; k++) { fodf[(int64_t)k * d.npad + c] = live ? fodf0[k] : 0.0f; if (tv) tv[(int64_t)k * d.npad + c] = 1.0f; }
    for (int r = 0; r < d.ndir; r++) {
        const int64_t i = (int64_t)r * d.npad + c;
        const float dd = live ? dodf0[r] : 0.0f;
        dodf[i] = dd;
        const float q = besseli_ratio(nu, (sig[i] * dd) / lam0);
        ir[i] = q;
        x[i] = sig[i] * q;                                        // (0 * NaN = NaN reaches the contraction like in the reference)
    }
    s2[c] = lam0;
}

// TV term of one compartment block (rumba_tv!, sd_grad!, sd_div!: rusd.jl:183-235), straight from the masked matrix:
// the embedded volume is fodf at masked voxels and 0 elsewhere.  One lane per masked voxel; the 13 neighbour columns
// are looked up once and reused for every compartment.
struct TvArgs {
    const float *fodf; float *tv; const int32_t *ind; const int32_t *col_of; const float *lam;   // lam: [npad] per column
    RumbaDims d;
    // fused update (rusd.jl:281, 301): fodf_new = max(fodf .* (rl ./ (rl2 + eps)) .* tv, 0) -- the TV term is consumed where it is
    // produced and never stored; fodf is double buffered because neighbouring lanes still read the old values
    const float *rl, *rl2; float *fodf_new;
};
__device__ __forceinline__ int32_t rumba_col(const TvArgs &a, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= a.d.nx || y >= a.d.ny || z >= a.d.nz) return -2;   // outside the volume
    return a.col_of[(int64_t)x + (int64_t)a.d.nx * ((int64_t)y + (int64_t)a.d.ny * z)];     // -1: outside the mask
}
__global__ __launch_bounds__(256) void rumba_tv_kernel(const TvArgs a) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.d.nmask) return;
    const int64_t v = a.ind[c];
    const int x = (int)(v % a.d.nx), y = (int)((v / a.d.nx) % a.d.ny), z = (int)(v / ((int64_t)a.d.nx * a.d.ny));
    // points whose normalised gradient enters div(p): p, p-x, p-y, p-z; each needs f at itself and at its +x,+y,+z
    // neighbours (replicated at the far boundary: sd_grad! takes f[[2:end; end]])
    const int px[4] = {x, x - 1, x, x}, py[4] = {y, y, y - 1, y}, pz[4] = {z, z, z, z - 1};
    int32_t cc[4][4];                                             // [point][self, +x, +y, +z]; -1 = value 0, -2 = point outside
    for (int q = 0; q < 4; q++) {
        cc[q][0] = rumba_col(a, px[q], py[q], pz[q]);
        if (cc[q][0] == -2) { cc[q][1] = cc[q][2] = cc[q][3] = -2; continue; }
        cc[q][1] = px[q] + 1 < a.d.nx ? rumba_col(a, px[q] + 1, py[q], pz[q]) : cc[q][0];
        cc[q][2] = py[q] + 1 < a.d.ny ? rumba_col(a, px[q], py[q] + 1, pz[q]) : cc[q][0];
        cc[q][3] = pz[q] + 1 < a.d.nz ? rumba_col(a, px[q], py[q], pz[q] + 1) : cc[q][0];
    }
    const float lam = a.lam[c];
    // VALU-bound with IEEE sqrt and divisions (4 + 8 per voxel and compartment: 3.06 ms per sweep); the normalisation is
    // gx * rsq(s) and the two quotients are rcp-based: every factor within 1 ulp of the exact one, i.e. the same accuracy
    // against the real-number result as the reference's sqrt-then-divide (two roundings), at a third of the instructions.
    // Components that div(p) never uses (y, z of the -x point, ...) are not formed.
    const bool in1 = cc[1][0] != -2, in2 = cc[2][0] != -2, in3 = cc[3][0] != -2;
    // 13 distinct values enter div(p); six of them are what the neighbouring LANE holds when the masked columns run along x
    // (lane+1 = voxel x+1, lane-1 = voxel x-1): they come over by wave shuffles, not as further L2 reads (the sweep moved
    // 23 GB between L2 and the CUs for 5.8 GB of HBM traffic).  Lanes at a wave edge or a mask edge load them as before.
    const int lane = threadIdx.x & 63;
    const bool up = lane < 63 && cc[0][1] == (int32_t)c + 1;      // lane+1 holds (x+1, y, z)
    const bool dn = lane > 0 && cc[1][0] == (int32_t)c - 1;       // lane-1 holds (x-1, y, z)
    for (int k = 0; k < a.d.ncomp; k++) {
        const float *f = a.fodf + (int64_t)k * a.d.npad;
        auto at = [&](int32_t col) { return col >= 0 ? f[col] : 0.0f; };
        auto inv_norm = [&](float gx, float gy, float gz) {       // 1 ./ sqrt.(Gx.^2 .+ Gy.^2 .+ Gz.^2 .+ eps(T))
            return __builtin_amdgcn_rsqf(((gx * gx + gy * gy) + gz * gz) + EPS32);
        };
        const float fc = f[c];                                    // cc[0][0] == c (and cc[1][1], cc[2][2], cc[3][3] when those points exist)
        const float fB = at(cc[0][2]), fC = at(cc[0][3]);         // p+y, p+z
        const float fD = at(cc[2][0]), fE = at(cc[3][0]);         // p-y, p-z
        const float fF = at(cc[2][3]), fG = at(cc[3][2]);         // p-y+z, p-z+y
        const float sA_up = __shfl_down(fc, 1), sD_up = __shfl_down(fD, 1), sE_up = __shfl_down(fE, 1);
        const float sA_dn = __shfl_up(fc, 1), sB_dn = __shfl_up(fB, 1), sC_dn = __shfl_up(fC, 1);
        const float fX1 = up ? sA_up : at(cc[0][1]);              // p+x
        const float fD1 = up ? sD_up : at(cc[2][1]);              // p-y+x
        const float fE1 = up ? sE_up : at(cc[3][1]);              // p-z+x
        const float fM1 = dn ? sA_dn : at(cc[1][0]);              // p-x
        const float fM1y = dn ? sB_dn : at(cc[1][2]);             // p-x+y
        const float fM1z = dn ? sC_dn : at(cc[1][3]);             // p-x+z
        float g0x, g0y, g0z, g1x = 0.0f, g2y = 0.0f, g3z = 0.0f;
        {
            const float gx = fX1 - fc, gy = fB - fc, gz = fC - fc;
            const float iv = inv_norm(gx, gy, gz);
            g0x = gx * iv; g0y = gy * iv; g0z = gz * iv;
        }
        if (in1) { const float gx = fc - fM1, gy = fM1y - fM1, gz = fM1z - fM1; g1x = gx * inv_norm(gx, gy, gz); }
        if (in2) { const float gx = fD1 - fD, gy = fc - fD, gz = fF - fD; g2y = gy * inv_norm(gx, gy, gz); }
        if (in3) { const float gx = fE1 - fE, gy = fG - fE, gz = fc - fE; g3z = gz * inv_norm(gx, gy, gz); }
        // sd_div!: interior G[i] - G[i-1]; first G[1]; last -G[end-1]
        const float dx = a.d.nx == 1 ? g0x : (x == 0 ? g0x : (x == a.d.nx - 1 ? -g1x : g0x - g1x));
        const float dy = a.d.ny == 1 ? g0y : (y == 0 ? g0y : (y == a.d.ny - 1 ? -g2y : g0y - g2y));
        const float dz = a.d.nz == 1 ? g0z : (z == 0 ? g0z : (z == a.d.nz - 1 ? -g3z : g0z - g3z));
        const float div = (dx + dy) + dz;
        const float tvv = __builtin_amdgcn_rcpf(fabsf(1.0f - lam * div) + EPS32);
        const int64_t i = (int64_t)k * a.d.npad + c;
        const float r = a.rl[i] * __builtin_amdgcn_rcpf(a.rl2[i] + EPS32);
        const float fo = (fc * r) * tvv;
        a.fodf_new[i] = fo > 0.0f ? fo : (fo != fo ? fo : 0.0f);   // max.(x, 0): NaN stays NaN
    }
}

// rl = rl ./ (rl2 + eps); fodf = max(fodf .* rl .* tv, 0)   (rusd.jl:281, 301)
__global__ __launch_bounds__(256) void rumba_update_kernel(int64_t n, const float *__restrict__ rl, const float *__restrict__ rl2,
                                                          const float *__restrict__ tv, float *__restrict__ fodf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = rl[i] / (rl2[i] + EPS32);
    const float f = (fodf[i] * r) * tv[i];
    fodf[i] = f > 0.0f ? f : (f != f ? f : 0.0f);                 // max.(x, 0): NaN stays NaN
}

// dodf_sig = signal .* dodf ./ sigma2;  sigma2 <- clamp(sum((signal^2 + dodf^2)/2 - sigma2 .* dodf_sig .* Iratio) / (n ndir))
// (rusd.jl:313-326); one lane per column, rows walked in order (the reference's column sum)
__global__ __launch_bounds__(256) void rumba_noise_kernel(RumbaDims d, float n_order, const float *__restrict__ sig, const float *__restrict__ dodf,
                                                         float *__restrict__ ir, float *__restrict__ x, float *__restrict__ s2,
                                                         float *__restrict__ snr, double *__restrict__ s2sum) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float sn = 0.0f;
    if (c < d.npad) {
    const float so = s2[c];
    float acc = 0.0f;
    for (int r = 0; r < d.ndir; r++) {
        const int64_t i = (int64_t)r * d.npad + c;
        const float s = sig[i], dd = dodf[i];
        const float ds = (s * dd) / so;                           // dodf_sig with the OLD sigma2 (rusd.jl:314)
        acc += (s * s + dd * dd) / 2.0f - (so * ds) * ir[i];
        const float q = besseli_ratio(n_order, ds);               // next iteration's Iratio (rusd.jl:275) and contraction operand
        ir[i] = q;
        x[i] = s * q;
    }
    sn = acc / (n_order * (float)d.ndir);
    const float lo = (float)((1.0 / 80.0) * (1.0 / 80.0)), hi = (float)((1.0 / 8.0) * (1.0 / 8.0));
    sn = sn < lo ? lo : (sn > hi ? hi : sn);                      // clamp! (NaN stays NaN)
    s2[c] = sn;
    snr[c] = 1.0f / sqrtf(sn);
    }
    // sum of sigma2 over the masked columns for mean(W.sigma2_vec) (rusd.jl:334): wave reduction + one double atomic per wave
    double part = (c < d.nmask) ? (double)sn : 0.0;
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    if ((threadIdx.x & 63) == 0 && s2sum) atomicAdd(s2sum, part);
}

// sum / sum of squares of v[0..n) in double (mean(sigma2), mean / std of the SNR): one block
__global__ __launch_bounds__(1024) void rumba_stats_kernel(const float *__restrict__ v, int64_t n, double *__restrict__ out) {
    __shared__ double s1[1024], s2[1024];
    double a = 0.0, b = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) { const double x = v[i]; a += x; b += x * x; }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) { s1[threadIdx.x] += s1[threadIdx.x + off]; s2[threadIdx.x] += s2[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(&out[0], s1[0]); atomicAdd(&out[1], s2[0]); }     // out zeroed by the caller
}
// lambda (rusd.jl:331-345): ipat == 1: max(mean(sigma2), (1/30)^2) everywhere; ipat > 1: sigma2 of the voxel
__global__ __launch_bounds__(256) void rumba_lambda_kernel(int64_t nmask, int ipat, const double *__restrict__ stats, const float *__restrict__ s2,
                                                          float *__restrict__ lam) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nmask) return;
    if (ipat == 1) {
        const float m = (float)(stats[0] / (double)nmask);
        const float lo = (float)((1.0 / 30.0) * (1.0 / 30.0));
        lam[c] = m > lo ? m : lo;
    } else {
        lam[c] = s2[c];
    }
}

// after the iterations (rusd.jl:553-590): energy preservation, embedding, + f_iso, normalisation over the vertices, GFA
struct PostArgs {
    const float *fodf, *s2; const int32_t *ind; RumbaDims d; int64_t nvox;
    float *out_fodf, *fgm, *fcsf, *gfa, *var;                     // out_fodf planar [nvert][nvox]
};
__global__ __launch_bounds__(256) void rumba_post_kernel(const PostArgs a) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.d.nmask) return;
    const int64_t v = a.ind[c];
    float tot = 0.0f;
    for (int k = 0; k < a.d.ncomp; k++) tot += a.fodf[(int64_t)k * a.d.npad + c];
    tot += EPS32;                                                 // W.fodf_mat ./= (sum(W.fodf_mat, dims=1) .+ eps(T))
    const float fcsf = a.fodf[(int64_t)a.d.nvert * a.d.npad + c] / tot, fgm = a.fodf[(int64_t)(a.d.nvert + 1) * a.d.npad + c] / tot;
    const float fiso = fgm + fcsf;
    float s = 0.0f;
    for (int k = 0; k < a.d.nvert; k++) s += a.fodf[(int64_t)k * a.d.npad + c] / tot + fiso;      // sum(fodf.vol, dims=4) after + f_iso
    float mean = 0.0f, sq = 0.0f;
    for (int k = 0; k < a.d.nvert; k++) {
        float f = (a.fodf[(int64_t)k * a.d.npad + c] / tot + fiso) / s;
        if (f != f) f = 0.0f;                                     // fodf.vol[isnan.(fodf.vol)] .= 0
        a.out_fodf[(int64_t)k * a.nvox + v] = f;
        mean += f; sq += f * f;
    }
    mean /= (float)a.d.nvert;
    float var = 0.0f;
    for (int k = 0; k < a.d.nvert; k++) { const float dlt = a.out_fodf[(int64_t)k * a.nvox + v] - mean; var += dlt * dlt; }
    float g = sqrtf(var / (float)(a.d.nvert - 1)) / sqrtf(sq / (float)a.d.nvert);                   // std ./ sqrt.(mean(.^2)), rusd.jl:589
    if (g != g) g = 0.0f;
    a.fgm[v] = fgm; a.fcsf[v] = fcsf; a.gfa[v] = g; a.var[v] = a.s2[c];
}

// rumba_peaks! + peak extraction (rusd.jl:348-373, 595-631): one lane per masked voxel
struct PeakArgs5 {
    const float *fodf; const float *fgm, *fcsf; const int32_t *ind; const int32_t *nb_off, *nb_idx; const float *verts;   // verts [nvert][3]
    float *peak[5];                                               // planar [3][nvox]
    RumbaDims d; int64_t nvox;
};
__global__ __launch_bounds__(128) void rumba_peaks_kernel(const PeakArgs5 a) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.d.nmask) return;
    const int64_t v = a.ind[c];
    const float fiso = a.fgm[v] + a.fcsf[v];
    float omax = -INFINITY;
    bool anynan = false;
    for (int k = 0; k < a.d.nvert; k++) { const float f = a.fodf[(int64_t)k * a.nvox + v]; anynan |= f != f; omax = f > omax ? f : omax; }
    if (anynan) omax = NAN;                                       // maximum() propagates NaN
    const float thr_abs = (0.1f / (1.0f - fiso)) * omax;          // thr / (1 - f_iso) * maximum(fodf)
    // keep the five best survivors: descending amplitude, ties -> lower index (sortperm!(..., rev=true), stable)
    float bv[5]; int bi[5]; int nvalid = 0;
    for (int i = 0; i < 5; i++) { bv[i] = 0.0f; bi[i] = -1; }
    for (int k = 0; k < a.d.nvert; k++) {
        const float f = a.fodf[(int64_t)k * a.nvox + v];
        float nmax = -INFINITY;
        bool nnan = false;
        for (int j = a.nb_off[k]; j < a.nb_off[k + 1]; j++) { const float y = a.fodf[(int64_t)a.nb_idx[j] * a.nvox + v]; nnan |= y != y; nmax = y > nmax ? y : nmax; }
        if (nnan) nmax = NAN;
        if (f < thr_abs || f <= nmax) continue;                   // fodf_peak[ivert] = 0
        if (!(f > 0.0f)) continue;                                // only entries with fodf_peak > 0 are used (nvalid)
        nvalid++;
        int pos = 5;
        for (int i = 4; i >= 0; i--) if (bi[i] < 0 || f > bv[i]) pos = i;
        for (int i = 4; i > pos; i--) { bv[i] = bv[i - 1]; bi[i] = bi[i - 1]; }
        if (pos < 5) { bv[pos] = f; bi[pos] = k; }
    }
    const int n = nvalid < 5 ? nvalid : 5;
    float ssum = 0.0f;
    for (int i = 0; i < n; i++) ssum += bv[i];
    const float fnorm = (1.0f - fiso) / ssum;
    for (int i = 0; i < 5; i++) {
        float px = 0.0f, py = 0.0f, pz = 0.0f;
        if (i < n) { const float w = bv[i] * fnorm; px = a.verts[3 * bi[i]] * w; py = a.verts[3 * bi[i] + 1] * w; pz = a.verts[3 * bi[i] + 2] * w; }
        a.peak[i][v] = px; a.peak[i][a.nvox + v] = py; a.peak[i][2 * a.nvox + v] = pz;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
struct fib_rumba_plan {
    int device = 0, nvol = 0, ndir = 0, ncomp = 0, nvert = 0;
    std::vector<float> K;                                         // [ndir x ncomp] column-major
    std::vector<int32_t> b0_frames, dw_frames;
    fib_odf_plan *pT = nullptr, *pK = nullptr;                    // contractions with K' and with K
    fib::DevBuf<int32_t> d_b0, d_dw, d_nb_off, d_nb_idx;
    fib::DevBuf<float> d_verts, d_fodf0, d_dodf0;
    // work arrays of a call (grow-only, kept for the next call on this plan: ten fresh GB cost ~0.3 s of hipMalloc per call;
    // a plan serves one call at a time)
    mutable fib::DevBuf<float> w_sig, w_dodf, w_ir, w_x, w_fodf, w_fodf2, w_rl, w_rl2;
};

extern "C" void fib_rumba_plan_destroy(fib_rumba_plan *p) try {
    if (!p) return;
    fib::DeviceGuard guard;
    (void)hipSetDevice(p->device);
    fib_odf_plan_destroy(p->pT);
    fib_odf_plan_destroy(p->pK);
    delete p;
} FIB_API_CATCH_VOID

extern "C" int fib_rumba_plan_create(int device, const float *bval, const float *bvec, int nvol, const float *verts, int nverts,
                                     float lam_para, float lam_perp, float lam_csf, float lam_gm, fib_rumba_plan **plan) try {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan output pointer is NULL");
    *plan = nullptr;
    FIB_CHECK(bval != nullptr && nvol > 0, FIB_ERR_MISSING_BVAL, "Missing b-value table from input DWI structure");   // rusd.jl:421
    FIB_CHECK(bvec != nullptr, FIB_ERR_MISSING_BVEC, "Missing gradient table from input DWI structure");              // rusd.jl:425
    FIB_CHECK(verts && nverts >= 2 && nverts % 2 == 0, FIB_ERR_INVALID, "invalid ODF tessellation");
    const int nvert = nverts / 2;
    double ang_neig;
    if (nvert == 362 || nvert == 321) ang_neig = 12.5;            // sphere_724 / sphere_642 (rusd.jl:476-477)
    else if (nvert == 181) ang_neig = 16.0;                       // sphere_362 (:478-479)
    else return fib::fail(FIB_ERR_UNSUPPORTED, "rumba_rec defines its peak neighbourhood for sphere_362/642/724 only (rusd.jl:476-480)");
    fib::DeviceGuard guard;
    int rc = fib::use_device(device);
    if (rc != FIB_OK) return rc;
    fib_rumba_plan *p = new (std::nothrow) fib_rumba_plan();
    FIB_CHECK(p != nullptr, FIB_ERR_NOMEM, "out of host memory");
    struct Guard { fib_rumba_plan *p; bool ok = false; ~Guard() { if (!ok) fib_rumba_plan_destroy(p); } } gd{p};
    p->device = device; p->nvol = nvol; p->nvert = nvert; p->ncomp = nvert + 2;
    float bmin = bval[0];
    for (int i = 1; i < nvol; i++) bmin = bval[i] < bmin ? bval[i] : bmin;
    for (int i = 0; i < nvol; i++) (bval[i] == bmin ? p->b0_frames : p->dw_frames).push_back(i);   // ib0, rusd.jl:449
    p->ndir = (int)p->dw_frames.size() + 1;
    const int ndir = p->ndir, ncomp = p->ncomp;
    // kernel of the multi-tensor model (rusd.jl:141-153, 466-469, 495-521), float64 then rounded once
    std::vector<double> g((size_t)ndir * 3, 0.0), b((size_t)ndir, 0.0);
    for (int r = 1; r < ndir; r++) {
        const int f = p->dw_frames[r - 1];
        const double x = bvec[f], y = bvec[f + (size_t)nvol], z = bvec[f + 2 * (size_t)nvol];
        const double n = std::sqrt(x * x + y * y + z * z);
        g[3 * r] = x / n; g[3 * r + 1] = y / n; g[3 * r + 2] = z / n;
        b[r] = bval[f];
    }
    p->K.assign((size_t)ndir * ncomp, 0.0f);
    for (int iv = 0; iv < nvert; iv++) {
        const double x = verts[nvert + iv], y = verts[nvert + iv + (size_t)nverts], z = verts[nvert + iv + 2 * (size_t)nverts];
        const double hxy = std::hypot(x, y), th = -std::atan2(z, hxy), ph = std::atan2(y, x);     // cart2sph, theta .= -theta
        const double c = std::cos(ph), s = std::sin(ph), ct = std::cos(th), st = std::sin(th);
        const double R[3][3] = {{c * ct, -s, c * st}, {s * ct, c, s * st}, {-st, 0.0, ct}};       // Rz * Ry (util.jl:85-100)
        const double lam[3] = {lam_para, lam_perp, lam_perp};
        double D[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { D[i][j] = 0.0; for (int k = 0; k < 3; k++) D[i][j] += R[i][k] * lam[k] * R[j][k]; }
        for (int r = 0; r < ndir; r++) {
            double q = 0.0;
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) q += g[3 * r + i] * D[i][j] * g[3 * r + j];
            p->K[r + (size_t)ndir * iv] = (float)std::exp(-b[r] * q);
        }
    }
    for (int r = 0; r < ndir; r++) {
        const double gg = g[3 * r] * g[3 * r] + g[3 * r + 1] * g[3 * r + 1] + g[3 * r + 2] * g[3 * r + 2];
        p->K[r + (size_t)ndir * nvert] = (float)std::exp(-b[r] * (double)lam_csf * gg);
        p->K[r + (size_t)ndir * (nvert + 1)] = (float)std::exp(-b[r] * (double)lam_gm * gg);
    }
    std::vector<float> KT((size_t)ncomp * ndir);
    for (int r = 0; r < ndir; r++) for (int k = 0; k < ncomp; k++) KT[k + (size_t)ncomp * r] = p->K[r + (size_t)ndir * k];
    if ((rc = fib::matrix_plan_create(device, KT.data(), ncomp, ndir, &p->pT)) != FIB_OK) return rc;
    if ((rc = fib::matrix_plan_create(device, p->K.data(), ndir, ncomp, &p->pK)) != FIB_OK) return rc;
    // peak neighbourhoods (rusd.jl:475-493)
    std::vector<int32_t> off(nvert + 1, 0), idx;
    std::vector<float> v3((size_t)nvert * 3);
    for (int i = 0; i < nvert; i++) for (int c = 0; c < 3; c++) v3[3 * i + c] = verts[i + (size_t)nverts * c];
    for (int i = 0; i < nvert; i++) {
        for (int j = 0; j < nvert; j++) {
            if (i == j) continue;
            float ca = (v3[3 * i] * v3[3 * j] + v3[3 * i + 1] * v3[3 * j + 1]) + v3[3 * i + 2] * v3[3 * j + 2];   // half_vertices * half_vertices' (Float32)
            ca = ca > 1.0f ? 1.0f : (ca < -1.0f ? -1.0f : ca);
            double ang = std::acos((double)ca) * 180.0 / M_PI;
            ang = std::min(ang, 180.0 - ang);
            if (ang < ang_neig) idx.push_back(j);
        }
        off[i + 1] = (int32_t)idx.size();
    }
    // initial estimates (rusd.jl:529-531, 245-249)
    std::vector<float> f0((size_t)ncomp), d0((size_t)ndir, 0.0f);
    { float s = 0.0f; for (int k = 0; k < ncomp; k++) { f0[k] = 1.0f / (float)(2 * nvert + 2); s += f0[k]; } for (int k = 0; k < ncomp; k++) f0[k] = f0[k] / s; }
    for (int r = 0; r < ndir; r++) { float s = 0.0f; for (int k = 0; k < ncomp; k++) s += p->K[r + (size_t)ndir * k] * f0[k]; d0[r] = s; }
    auto up = [&](auto &buf, const auto &host) -> int {
        int r2 = buf.alloc(host.size());
        if (r2 != FIB_OK) return r2;
        FIB_HIP(hipMemcpy(buf.p, host.data(), host.size() * sizeof(host[0]), hipMemcpyHostToDevice));
        return FIB_OK;
    };
    if ((rc = up(p->d_b0, p->b0_frames)) != FIB_OK || (rc = up(p->d_dw, p->dw_frames.empty() ? std::vector<int32_t>{0} : p->dw_frames)) != FIB_OK ||
        (rc = up(p->d_nb_off, off)) != FIB_OK || (rc = up(p->d_nb_idx, idx.empty() ? std::vector<int32_t>{0} : idx)) != FIB_OK ||
        (rc = up(p->d_verts, v3)) != FIB_OK || (rc = up(p->d_fodf0, f0)) != FIB_OK || (rc = up(p->d_dodf0, d0)) != FIB_OK) return rc;
    gd.ok = true;
    *plan = p;
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fib_rumba_plan_kernel(const fib_rumba_plan *plan, float *K, int *ndir, int *ncomp) try {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan is NULL");
    if (ndir) *ndir = plan->ndir;
    if (ncomp) *ncomp = plan->ncomp;
    if (K) memcpy(K, plan->K.data(), plan->K.size() * sizeof(float));
    return FIB_OK;
} FIB_API_CATCH

// ------------------------------------------------------------------------------------------
// reconstruction, device tier
// ------------------------------------------------------------------------------------------
extern "C" int fibd_rumba_rec(const fib_rumba_plan *plan, const float *dwi, const uint8_t *mask, int nx, int ny, int nz,
                              int niter, int ncoils, int sos_grappa, int ipat_factor, int use_tv,
                              const fib_rumba_out *out, float *snr_mean, float *snr_std, void *stream) try {
    FIB_CHECK(plan && dwi && mask && out, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nx > 0 && ny > 0 && nz > 0 && niter >= 0, FIB_ERR_INVALID, "invalid sizes");
    FIB_CHECK(ipat_factor >= 1, FIB_ERR_INVALID, "iPAT factor must be a positive integer");                          // rusd.jl:437
    FIB_CHECK(out->fodf && out->fgm && out->fcsf && out->gfa && out->var, FIB_ERR_INVALID, "NULL output volume");
    for (int i = 0; i < 5; i++) FIB_CHECK(out->peak[i] != nullptr, FIB_ERR_INVALID, "NULL peak output volume");
    const float n_order = sos_grappa ? (float)ncoils : 1.0f;                                                          // rusd.jl:429-435
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t nvox = (int64_t)nx * ny * nz;
    FIB_CHECK(nvox < ((int64_t)1 << 31), FIB_ERR_UNSUPPORTED, "volume too large");
    // ind_mask = findall(vec(mask.vol) .> 0) and its inverse (host: once per call)
    std::vector<uint8_t> hm((size_t)nvox);
    FIB_HIP(hipMemcpyAsync(hm.data(), mask, (size_t)nvox, hipMemcpyDeviceToHost, st));
    FIB_HIP(hipStreamSynchronize(st));
    std::vector<int32_t> ind, col((size_t)nvox, -1);
    for (int64_t v = 0; v < nvox; v++) if (hm[v]) { col[v] = (int32_t)ind.size(); ind.push_back((int32_t)v); }
    const int64_t nmask = (int64_t)ind.size();
    // outputs start as zeros (MRI(mask, n, T) -> zeros, mri.jl:249-265)
    FIB_HIP(hipMemsetAsync(out->fodf, 0, sizeof(float) * nvox * plan->nvert, st));
    for (float *q : {out->fgm, out->fcsf, out->gfa, out->var}) FIB_HIP(hipMemsetAsync(q, 0, sizeof(float) * nvox, st));
    for (int i = 0; i < 5; i++) FIB_HIP(hipMemsetAsync(out->peak[i], 0, sizeof(float) * nvox * 3, st));
    if (snr_mean) *snr_mean = 0.0f;
    if (snr_std) *snr_std = 0.0f;
    if (nmask == 0) return FIB_OK;
    RumbaDims d{plan->ndir, plan->ncomp, plan->nvert, nmask, (nmask + 255) / 256 * 256, nx, ny, nz};
    const int64_t nD = (int64_t)d.ndir * d.npad, nC = (int64_t)d.ncomp * d.npad;
    fib::DevBuf<int32_t> d_ind, d_col;
    fib::DevBuf<float> tv, s2, snr, lam;
    fib::DevBuf<float> &sig = plan->w_sig, &dodf = plan->w_dodf, &ir = plan->w_ir, &x = plan->w_x, &fodf = plan->w_fodf, &fodf2 = plan->w_fodf2,
                       &rl = plan->w_rl, &rl2 = plan->w_rl2;
    fib::DevBuf<uint8_t> ones;
    fib::DevBuf<double> stats;
    int rc;
    if ((rc = d_ind.alloc((size_t)nmask)) != FIB_OK || (rc = d_col.alloc((size_t)nvox)) != FIB_OK ||
        (rc = sig.ensure((size_t)nD)) != FIB_OK || (rc = dodf.ensure((size_t)nD)) != FIB_OK ||
        (rc = ir.ensure((size_t)nD)) != FIB_OK || (rc = x.ensure((size_t)nD)) != FIB_OK || (rc = fodf.ensure((size_t)nC)) != FIB_OK ||
        (rc = fodf2.ensure((size_t)nC)) != FIB_OK || (rc = rl.ensure((size_t)nC)) != FIB_OK || (rc = rl2.ensure((size_t)nC)) != FIB_OK ||
        (rc = tv.alloc((size_t)(use_tv ? 1 : nC))) != FIB_OK ||
        (rc = s2.alloc((size_t)d.npad)) != FIB_OK || (rc = snr.alloc((size_t)d.npad)) != FIB_OK || (rc = lam.alloc((size_t)d.npad)) != FIB_OK ||
        (rc = ones.alloc((size_t)d.npad)) != FIB_OK || (rc = stats.alloc(2)) != FIB_OK) return rc;
    FIB_HIP(hipMemcpyAsync(d_ind.p, ind.data(), sizeof(int32_t) * nmask, hipMemcpyHostToDevice, st));
    FIB_HIP(hipMemcpyAsync(d_col.p, col.data(), sizeof(int32_t) * nvox, hipMemcpyHostToDevice, st));
    FIB_HIP(hipMemsetAsync(ones.p, 1, (size_t)d.npad, st));
    const unsigned gcol = (unsigned)fib::cdiv(d.npad, 256), gmask = (unsigned)fib::cdiv(nmask, 256);
    hipLaunchKernelGGL(rumba_signal_kernel, dim3(gcol), dim3(256), 0, st, dwi, nvox, d_ind.p, plan->d_b0.p, (int)plan->b0_frames.size(),
                       plan->d_dw.p, d, sig.p);
    const float lam0 = (1.0f / 15.0f) * (1.0f / 15.0f);                                                               // sigma0^2, rusd.jl:537-538
    float *f_cur = fodf.p, *f_new = fodf2.p;
    FIB_HIP(hipMemsetAsync(fodf2.p, 0, sizeof(float) * nC, st));   // (its padding columns are never written again)
    hipLaunchKernelGGL(rumba_init_kernel, dim3(gcol), dim3(256), 0, st, d, n_order, plan->d_fodf0.p, plan->d_dodf0.p, lam0, sig.p, f_cur, dodf.p,
                       ir.p, x.p, use_tv ? (float *)nullptr : tv.p, s2.p);          // without TV the term stays 1
    {   // lambda = lambda0 everywhere
        std::vector<float> l0((size_t)d.npad, lam0);
        FIB_HIP(hipMemcpyAsync(lam.p, l0.data(), sizeof(float) * d.npad, hipMemcpyHostToDevice, st));
        FIB_HIP(hipStreamSynchronize(st));
    }
    FIB_HIP(hipGetLastError());
    for (int it = 0; it < niter; it++) {                          // rumba_sd_iterate!, rusd.jl:266-345
        if ((rc = fib::matrix_plan_run(plan->pT, x.p, ones.p, d.npad, rl.p, it == 0, st)) != FIB_OK) return rc;        // K' (signal .* Iratio)
        if ((rc = fib::matrix_plan_run(plan->pT, dodf.p, ones.p, d.npad, rl2.p, false, st)) != FIB_OK) return rc;     // K' dodf
        if (use_tv) {                                             // TV term + multiplicative update in one pass
            fib::ProfScope prof("rumba_tv", st);
            TvArgs ta{f_cur, nullptr, d_ind.p, d_col.p, lam.p, d, rl.p, rl2.p, f_new};
            hipLaunchKernelGGL(rumba_tv_kernel, dim3(gmask), dim3(256), 0, st, ta);
            std::swap(f_cur, f_new);
        } else {
            fib::ProfScope prof("rumba_elementwise", st);
            hipLaunchKernelGGL(rumba_update_kernel, dim3((unsigned)fib::cdiv(nC, 256)), dim3(256), 0, st, nC, rl.p, rl2.p, tv.p, f_cur);
        }
        if ((rc = fib::matrix_plan_run(plan->pK, f_cur, ones.p, d.npad, dodf.p, it == 0, st)) != FIB_OK) return rc;    // K fodf
        { fib::ProfScope prof("rumba_elementwise", st);
        const bool need_mean = use_tv && ipat_factor == 1;
        if (need_mean) FIB_HIP(hipMemsetAsync(stats.p, 0, 2 * sizeof(double), st));
        hipLaunchKernelGGL(rumba_noise_kernel, dim3(gcol), dim3(256), 0, st, d, n_order, sig.p, dodf.p, ir.p, x.p, s2.p, snr.p,
                           need_mean ? stats.p : (double *)nullptr);
        if (use_tv) hipLaunchKernelGGL(rumba_lambda_kernel, dim3(gmask), dim3(256), 0, st, nmask, ipat_factor, stats.p, s2.p, lam.p); }
        FIB_HIP(hipGetLastError());
    }
    float *fodf_final = f_cur;
    if (niter > 0 && (snr_mean || snr_std)) {                     // mean / std (corrected) of the SNR estimates, rusd.jl:546-547
        FIB_HIP(hipMemsetAsync(stats.p, 0, 2 * sizeof(double), st));
        hipLaunchKernelGGL(rumba_stats_kernel, dim3(64), dim3(1024), 0, st, snr.p, nmask, stats.p);
        double hs[2];
        FIB_HIP(hipMemcpyAsync(hs, stats.p, sizeof hs, hipMemcpyDeviceToHost, st));
        FIB_HIP(hipStreamSynchronize(st));
        const double m = hs[0] / (double)nmask;
        const double var = nmask > 1 ? std::max(0.0, (hs[1] - (double)nmask * m * m) / (double)(nmask - 1)) : 0.0;
        if (snr_mean) *snr_mean = (float)m;
        if (snr_std) *snr_std = (float)std::sqrt(var);
    }
    PostArgs pa{fodf_final, s2.p, d_ind.p, d, nvox, out->fodf, out->fgm, out->fcsf, out->gfa, out->var};
    hipLaunchKernelGGL(rumba_post_kernel, dim3(gmask), dim3(256), 0, st, pa);
    PeakArgs5 pk{};
    pk.fodf = out->fodf; pk.fgm = out->fgm; pk.fcsf = out->fcsf; pk.ind = d_ind.p; pk.nb_off = plan->d_nb_off.p; pk.nb_idx = plan->d_nb_idx.p;
    pk.verts = plan->d_verts.p; pk.d = d; pk.nvox = nvox;
    for (int i = 0; i < 5; i++) pk.peak[i] = out->peak[i];
    hipLaunchKernelGGL(rumba_peaks_kernel, dim3((unsigned)fib::cdiv(nmask, 128)), dim3(128), 0, st, pk);
    FIB_HIP(hipGetLastError());
    FIB_HIP(hipStreamSynchronize(st));                            // the work buffers are locals
    return FIB_OK;
} FIB_API_CATCH
