// dti.hip — K1: per-voxel DTI / ADC least-squares fit on gfx950.
//
// Replaces the volume loops dti.jl:175-184 (adc_fit) and dti.jl:258-275 (dti_fit_ls) and the
// per-voxel bodies dti.jl:195-213 / 286-316 + dti_maps dti.jl:325-335.
//
// Design (HBM-bound: 4*nvol + 1 bytes in, 64 bytes out per voxel, ~20 VALU ops per sample):
//   - one thread owns V consecutive voxels; V = 1 with 16 frames in flight per lane and 7 waves/SIMD
//     measured fastest (5.3 TB/s), wider per-lane loads (V = 2, 4) cost occupancy and lose;
//   - the pseudo-inverse rows pA[:, i] (padded to 8 floats per frame, 8th = b0 flag) are read
//     with wave-uniform addresses, i.e. through the scalar cache into SGPRs — no LDS traffic,
//     no VGPRs; the frame loop is unrolled so ~8 independent loads are in flight per lane;
//   - log(s) feeds 7 (or 2) FMAs per sample; the 3x3 symmetric eigen-solve is StaticArrays'
//     closed form (the algorithm `eigen(Symmetric(D,:L))` dispatches to at dti.jl:311) in registers;
//   - voxels with non-positive samples (dti.jl:297-298 per-voxel pinv) are rare: the main kernel
//     only appends them to a list; a second small kernel solves their row-subset least-squares
//     problem via float64 normal equations + Jacobi eigen-decomposition (== pinv incl. rank cut-off).
#include "common.h"

// The reference (Julia) never contracts a*b+c; keep the eigen-solver's cancellation-prone
// cross products bit-compatible with the CPU restatement.  FMAs are written explicitly.
#pragma clang fp contract(off)

namespace {

struct DtiOutPtrs {
    float *s0, *l1, *l2, *l3, *e1, *e2, *e3, *rd, *md, *fa;
};

__device__ __forceinline__ void cross3(const float a[3], const float b[3], float c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// StaticArrays `_eig(::Size{(3,3)}, ::RealHermSymComplexHerm)`: trigonometric eigenvalues, eigenvectors
// from the best-conditioned cross product, 2x2 sub-problem for the second one.  Ascending eigenvalues
// w[0..2], eigenvectors ev[k][:].
__device__ __forceinline__ void sym3_eigen(float a11, float a12, float a13, float a22, float a23, float a33,
                           float w[3], float ev[3][3]) {
    const float p1 = a12 * a12 + a13 * a13 + a23 * a23;
    if (p1 == 0.0f) {  // diagonal matrix: sorted diagonal, unit axes
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
#define FIB_DIAG_ROW(k, o)                                   \
        w[k] = (o) == 0 ? a11 : ((o) == 1 ? a22 : a33);      \
        ev[k][0] = (o) == 0 ? 1.0f : 0.0f;                   \
        ev[k][1] = (o) == 1 ? 1.0f : 0.0f;                   \
        ev[k][2] = (o) == 2 ? 1.0f : 0.0f;
        FIB_DIAG_ROW(0, o0)
        FIB_DIAG_ROW(1, o1)
        FIB_DIAG_ROW(2, o2)
#undef FIB_DIAG_ROW
        return;
    }
    const float q = (a11 + a22 + a33) / 3.0f;
    const float p2 = (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) + (a33 - q) * (a33 - q) + 2.0f * p1;
    const float p = sqrtf(p2 / 6.0f);
    const float invp = 1.0f / p;
    const float b11 = (a11 - q) * invp, b22 = (a22 - q) * invp, b33 = (a33 - q) * invp;
    const float b12 = a12 * invp, b13 = a13 * invp, b23 = a23 * invp;
    const float detB = b11 * (b22 * b33 - b23 * b23) - b12 * (b12 * b33 - b23 * b13) + b13 * (b12 * b23 - b22 * b13);
    const float r = detB / 2.0f;
    const float PI_F = 3.14159274101257324f;
    float phi;
    if (r <= -1.0f)     phi = PI_F / 3.0f;
    else if (r >= 1.0f) phi = 0.0f;
    else                phi = acosf(r) / 3.0f;
    float eig3 = q + 2.0f * p * cosf(phi);
    float eig1 = q + 2.0f * p * cosf(phi + (2.0f * PI_F / 3.0f));
    const float eig2 = 3.0f * q - eig1 - eig3;
    if (r > 0.0f) { const float t = eig1; eig1 = eig3; eig3 = t; }

    const float r1[3] = {a11 - eig1, a12, a13};
    const float r2[3] = {a12, a22 - eig1, a23};
    const float r3[3] = {a13, a23, a33 - eig1};
    const float n1 = r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2];
    const float n2 = r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2];
    const float n3 = r3[0] * r3[0] + r3[1] * r3[1] + r3[2] * r3[2];
    float r12[3], r23[3], r31[3];
    cross3(r1, r2, r12); cross3(r2, r3, r23); cross3(r3, r1, r31);
    const float n12 = r12[0] * r12[0] + r12[1] * r12[1] + r12[2] * r12[2];
    const float n23 = r23[0] * r23[0] + r23[1] * r23[1] + r23[2] * r23[2];
    const float n31 = r31[0] * r31[0] + r31[1] * r31[1] + r31[2] * r31[2];
    int sel;  // 0: r12, 1: r23, 2: r31
    if (n12 * n3 > n23 * n1) sel = (n12 * n3 > n31 * n2) ? 0 : 2;
    else                     sel = (n23 * n1 > n31 * n2) ? 1 : 2;
    float v1[3];
    {
        const float nb = sel == 0 ? n12 : (sel == 1 ? n23 : n31);
        const float s = sqrtf(nb);
#pragma unroll
        for (int c = 0; c < 3; c++) v1[c] = (sel == 0 ? r12[c] : (sel == 1 ? r23[c] : r31[c])) / s;
    }
    float o1[3], o2[3];
    if (fabsf(v1[0]) < fabsf(v1[1])) {
        const float s = sqrtf(v1[0] * v1[0] + v1[2] * v1[2]);
        o1[0] = -v1[2] / s; o1[1] = 0.0f / s; o1[2] = v1[0] / s;
    } else {
        const float s = sqrtf(v1[1] * v1[1] + v1[2] * v1[2]);
        o1[0] = 0.0f / s; o1[1] = v1[2] / s; o1[2] = -v1[1] / s;
    }
    cross3(v1, o1, o2);
    const float ao1[3] = {a11 * o1[0] + a12 * o1[1] + a13 * o1[2],
                          a12 * o1[0] + a22 * o1[1] + a23 * o1[2],
                          a13 * o1[0] + a23 * o1[1] + a33 * o1[2]};
    const float ao2[3] = {a11 * o2[0] + a12 * o2[1] + a13 * o2[2],
                          a12 * o2[0] + a22 * o2[1] + a23 * o2[2],
                          a13 * o2[0] + a23 * o2[1] + a33 * o2[2]};
    const float c11 = o1[0] * ao1[0] + o1[1] * ao1[1] + o1[2] * ao1[2] - eig2;
    const float c12 = o1[0] * ao2[0] + o1[1] * ao2[1] + o1[2] * ao2[2];
    const float c22 = o2[0] * ao2[0] + o2[1] * ao2[1] + o2[2] * ao2[2] - eig2;
    const float c11s = c11 * c11, c12s = c12 * c12, c22s = c22 * c22;
    float q1 = 1.0f, q2 = 0.0f;   // eigvec2 = q1*o1 - q2*o2 (defaults: orthogonal1)
    if (c11s >= c22s) {
        if (c11s > 0.0f || c12s > 0.0f) {
            if (c11s >= c12s) { const float t = c12 / c11; q2 = 1.0f / sqrtf(1.0f + t * t); q1 = t * q2; }
            else              { const float t = c11 / c12; q1 = 1.0f / sqrtf(1.0f + t * t); q2 = t * q1; }
        }
    } else {
        if (c22s >= c12s) { const float t = c12 / c22; q1 = 1.0f / sqrtf(1.0f + t * t); q2 = t * q1; }
        else              { const float t = c22 / c12; q2 = 1.0f / sqrtf(1.0f + t * t); q1 = t * q2; }
    }
    float v2[3], v3[3];
    const bool degenerate = (c11s >= c22s) && !(c11s > 0.0f || c12s > 0.0f);
#pragma unroll
    for (int c = 0; c < 3; c++) v2[c] = degenerate ? o1[c] : q1 * o1[c] - q2 * o2[c];
    cross3(v1, v2, v3);
    if (r > 0.0f) {
        const float t = eig1; eig1 = eig3; eig3 = t;
#pragma unroll
        for (int c = 0; c < 3; c++) { const float u = v1[c]; v1[c] = v3[c]; v3[c] = u; }
    }
    w[0] = eig1; w[1] = eig2; w[2] = eig3;
#pragma unroll
    for (int c = 0; c < 3; c++) { ev[0][c] = v1[c]; ev[1][c] = v2[c]; ev[2][c] = v3[c]; }
}

// d[7] -> the 16 output scalars of one voxel (dti.jl:305-315 + dti_maps dti.jl:325-335)
struct Out16 { float o[16]; };
__device__ __forceinline__ void dti_finish_inl(const float d[7], float o[16]);
// out-of-line (one copy per kernel) with everything passed in registers
__device__ __noinline__ Out16 dti_finish_call(float d0, float d1, float d2, float d3, float d4, float d5, float d6) {
    const float d[7] = {d0, d1, d2, d3, d4, d5, d6};
    Out16 r;
    dti_finish_inl(d, r.o);
    return r;
}
__device__ __forceinline__ void dti_finish(const float d[7], float o[16]) {
    const Out16 r = dti_finish_call(d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
#pragma unroll
    for (int k = 0; k < 16; k++) o[k] = r.o[k];
}
__device__ __forceinline__ void dti_finish_inl(const float d[7], float o[16]) {
    float w[3], ev[3][3];
    o[0] = expf(d[6]);
    sym3_eigen(d[0], d[1], d[2], d[3], d[4], d[5], w, ev);
    const float e1 = w[2], e2 = w[1], e3 = w[0];
    o[1] = e1; o[2] = e2; o[3] = e3;
#pragma unroll
    for (int c = 0; c < 3; c++) { o[4 + c] = ev[2][c]; o[7 + c] = ev[1][c]; o[10 + c] = ev[0][c]; }
    float rd = e2 + e3;
    const float md = (e1 + rd) / 3.0f;
    rd = rd / 2.0f;
    const float num = (e1 - md) * (e1 - md) + (e2 - md) * (e2 - md) + (e3 - md) * (e3 - md);
    const float den = e1 * e1 + e2 * e2 + e3 * e3;
    o[13] = rd; o[14] = md; o[15] = sqrtf(num / den * 1.5f);
}

template <int V> struct VecT;
typedef float nt_f2 __attribute__((ext_vector_type(2)));
typedef float nt_f4 __attribute__((ext_vector_type(4)));
template <> struct VecT<1> { using F = float;  using M = uint8_t; };
template <> struct VecT<2> { using F = nt_f2; using M = uint16_t; };
template <> struct VecT<4> { using F = nt_f4; using M = uint32_t; };

template <int V> __device__ __forceinline__ void vload(const float *p, float (&x)[V]) {
    const typename VecT<V>::F t = __builtin_nontemporal_load(reinterpret_cast<const typename VecT<V>::F *>(p));
    __builtin_memcpy(x, &t, sizeof t);
}
template <int V> __device__ __forceinline__ void vstore(float *p, const float (&x)[V]) {
    typename VecT<V>::F t;
    __builtin_memcpy(&t, x, sizeof t);
    __builtin_nontemporal_store(t, reinterpret_cast<typename VecT<V>::F *>(p));
}

// NP = 7: DTI, NP = 2: ADC.  coef: [nvol][8] = pA[:, i] (NP floats), zero pad, [7] = (bval[i]==min) flag
template <int NP, int V, int UNR>
__global__ __launch_bounds__(256) void fit_kernel(const float *__restrict__ dwi, const uint8_t *__restrict__ mask,
                                                  const float *__restrict__ coef, int nvol, int64_t nvox,
                                                  DtiOutPtrs out, float *__restrict__ adc,
                                                  int64_t *__restrict__ partial_list, int *__restrict__ partial_count) {
    const int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (base >= nvox) return;
    // Fast path = every sample is a normal positive float (dti.jl:294).  The kernel is VALU-bound, not
    // HBM-bound, unless the per-sample work is tiny (PMC: 33 VALU/sample -> 77 % VALU busy), so the loop does
    // only: running minimum (1), log = v_log_f32 * ln2 (2), NP FMAs.  A non-positive / denormal / NaN sample
    // shows up as min < FLT_MIN or as a NaN in d; such voxels are listed for the complete (slow) kernel.
    uint8_t mk[V];
    {
        const typename VecT<V>::M t = *reinterpret_cast<const typename VecT<V>::M *>(mask + base);
        __builtin_memcpy(mk, &t, sizeof t);
    }
    bool anymask = false;
#pragma unroll
    for (int v = 0; v < V; v++) anymask |= mk[v] != 0;
    // a wave whose voxels are all outside the mask reads no frame at all (brain masks cover ~1/3 of a volume)
    const int nframes = __any(anymask) ? nvol : 0;
    // the NP dot products run two at a time on v_pk_fma_f32 (coefficient pairs straight from SGPRs, the logarithm broadcast
    // by op_sel): 4 instead of 7 FMA instructions per sample, each lane-result the same IEEE fma as before
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int NP2 = (NP + 1) / 2;
    f32x2 d2[V][NP2];
    float smin[V];
#pragma unroll
    for (int v = 0; v < V; v++) {
        smin[v] = INFINITY;
#pragma unroll
        for (int j = 0; j < NP2; j++) d2[v][j] = f32x2{0.0f, 0.0f};
    }
    const float *src = dwi + base;
#pragma unroll UNR
    for (int i = 0; i < nframes; i++) {
        float s[V];
        vload<V>(src + (int64_t)i * nvox, s);
        const f32x2 *c = reinterpret_cast<const f32x2 *>(coef + 8 * i);   // wave-uniform: scalar loads
#pragma unroll
        for (int v = 0; v < V; v++) {
            smin[v] = fminf(smin[v], s[v]);
            const float l = __builtin_amdgcn_logf(s[v]) * 0.693147182464599609375f;   // log.(dwi), dti.jl:295
            const f32x2 l2 = {l, l};
#pragma unroll
            for (int j = 0; j < NP2; j++) d2[v][j] = __builtin_elementwise_fma(c[j], l2, d2[v][j]);   // mul!(d, pA, logs), dti.jl:296
        }
    }
    float d[V][NP];
#pragma unroll
    for (int v = 0; v < V; v++)
#pragma unroll
        for (int j = 0; j < NP; j++) d[v][j] = d2[v][j >> 1][j & 1];
    bool fastok[V];
#pragma unroll
    for (int v = 0; v < V; v++) {
        float t = d[v][0];
#pragma unroll
        for (int j = 1; j < NP; j++) t += d[v][j];
        fastok[v] = smin[v] >= 1.17549435e-38f && t == t;
    }
    if constexpr (NP == 7) {
        float o[16][V];
#pragma unroll
        for (int v = 0; v < V; v++) {
            float r[16];
#pragma unroll
            for (int k = 0; k < 16; k++) r[k] = 0.0f;
            if (mk[v] != 0) {                                   // dti.jl:261
                if (fastok[v]) {                                // dti.jl:294
                    dti_finish(d[v], r);
                } else {                                        // dti.jl:297-303 -> complete kernel
                    const int slot = atomicAdd(partial_count, 1);
                    partial_list[slot] = base + v;
                }
            }
#pragma unroll
            for (int k = 0; k < 16; k++) o[k][v] = r[k];
        }
        vstore<V>(out.s0 + base, o[0]);
        vstore<V>(out.l1 + base, o[1]);
        vstore<V>(out.l2 + base, o[2]);
        vstore<V>(out.l3 + base, o[3]);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            vstore<V>(out.e1 + c * nvox + base, o[4 + c]);
            vstore<V>(out.e2 + c * nvox + base, o[7 + c]);
            vstore<V>(out.e3 + c * nvox + base, o[10 + c]);
        }
        vstore<V>(out.rd + base, o[13]);
        vstore<V>(out.md + base, o[14]);
        vstore<V>(out.fa + base, o[15]);
    } else {
        float a[V], s0[V];
#pragma unroll
        for (int v = 0; v < V; v++) {
            a[v] = 0.0f; s0[v] = 0.0f;
            if (mk[v] != 0) {
                if (fastok[v]) { a[v] = d[v][0]; s0[v] = expf(d[v][1]); }                // dti.jl:212
                else {                                                                    // dti.jl:206-210
                    const int slot = atomicAdd(partial_count, 1);
                    partial_list[slot] = base + v;
                }
            }
        }
        vstore<V>(adc + base, a);
        vstore<V>(out.s0 + base, s0);
    }
}

// cyclic Jacobi eigen-decomposition of a symmetric NxN float64 matrix (rare path; arrays live in scratch)
template <int N>
__device__ void jacobi_sym(double (&A)[N][N], double (&Q)[N][N]) {
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) Q[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < N; i++) {
            diag += A[i][i] * A[i][i];
            for (int j = i + 1; j < N; j++) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < N - 1; p++)
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p][q];
                if (apq == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < N; k++) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; k++) {
                    const double qkp = Q[k][p], qkq = Q[k][q];
                    Q[k][p] = c * qkp - s * qkq;
                    Q[k][q] = s * qkp + c * qkq;
                }
            }
    }
}

// The complete per-voxel fit (dti.jl:286-303) for the voxels the fast kernel could not finish: exact
// positive count, accurate logf; all positive -> d = pA*log(s); else npos > 6 with a positive b0 -> row-subset
// least squares (== pinv(A[ipos,:]) * log(s[ipos]), float64 normal equations + Jacobi, LinearAlgebra.pinv's
// rank cut-off); else zeros.  design/coef: [nvol][8].  One thread per listed voxel; count[1] += subset solves.
template <int NP>
__global__ __launch_bounds__(64) void fit_partial_kernel(const float *__restrict__ dwi, const float *__restrict__ design,
                                                         const float *__restrict__ coef,
                                                         int nvol, int64_t nvox, const int64_t *__restrict__ list,
                                                         int *__restrict__ count, DtiOutPtrs out, float *__restrict__ adc) {
  const int total = *count;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int64_t vox = list[t];
    int npos = 0;
    bool b0pos = false;
    for (int i = 0; i < nvol; i++) {
        const float s = dwi[(int64_t)i * nvox + vox];
        if (s > 0.0f) { npos++; b0pos |= coef[8 * i + 7] != 0.0f; }     // dti.jl:291-292, any(ipos[ib0])
    }
    float d[NP];
    bool solved = true;
    if (npos == nvol) {                                                  // dti.jl:294-296
        for (int j = 0; j < NP; j++) d[j] = 0.0f;
        for (int i = 0; i < nvol; i++) {
            const float l = logf(dwi[(int64_t)i * nvox + vox]);
            for (int j = 0; j < NP; j++) d[j] = __builtin_fmaf(coef[8 * i + j], l, d[j]);
        }
    } else if (npos > 6 && b0pos) {                                      // dti.jl:297-298
        atomicAdd(count + 1, 1);
        double N[NP][NP], Q[NP][NP], rhs[NP];
        for (int i = 0; i < NP; i++) { rhs[i] = 0.0; for (int j = 0; j < NP; j++) N[i][j] = 0.0; }
        for (int i = 0; i < nvol; i++) {
            const float s = dwi[(int64_t)i * nvox + vox];
            if (!(s > 0.0f)) continue;                                   // A[ipos, :]
            const double l = (double)logf(s);
            double a[NP];
            for (int j = 0; j < NP; j++) a[j] = (double)design[8 * i + j];
            for (int j = 0; j < NP; j++) {
                rhs[j] += a[j] * l;
                for (int k = 0; k < NP; k++) N[j][k] += a[j] * a[k];
            }
        }
        jacobi_sym<NP>(N, Q);
        // pinv(A_sub) b = V diag(1/lambda) V' A_sub' b; singular values sqrt(lambda) <= eps32*min(m,n)*smax dropped
        double lmax = 0.0;
        for (int j = 0; j < NP; j++) lmax = fmax(lmax, N[j][j]);
        const double rt = (double)1.1920929e-07f * (double)(npos < NP ? npos : NP);
        const double cut = rt * rt * lmax;
        for (int r = 0; r < NP; r++) {
            double acc = 0.0;
            for (int j = 0; j < NP; j++) {
                if (!(N[j][j] > cut)) continue;
                double proj = 0.0;
                for (int k = 0; k < NP; k++) proj += Q[k][j] * rhs[k];
                acc += Q[r][j] * proj / N[j][j];
            }
            d[r] = (float)acc;
        }
    } else {
        solved = false;                                                  // dti.jl:299-303: zeros (already written)
    }
    if (!solved) continue;
    if constexpr (NP == 7) {
        float o[16];
        dti_finish(d, o);
        out.s0[vox] = o[0]; out.l1[vox] = o[1]; out.l2[vox] = o[2]; out.l3[vox] = o[3];
        for (int c = 0; c < 3; c++) {
            out.e1[c * nvox + vox] = o[4 + c];
            out.e2[c * nvox + vox] = o[7 + c];
            out.e3[c * nvox + vox] = o[10 + c];
        }
        out.rd[vox] = o[13]; out.md[vox] = o[14]; out.fa[vox] = o[15];
    } else {
        adc[vox] = d[0];
        out.s0[vox] = expf(d[1]);
    }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
struct fib_dti_plan {
    int device = 0;
    int nvol = 0;
    int np = 7;                         // 7: DTIwork, 2: ADCwork
    std::vector<float> A, pA;           // host copies, column-major [nvol x np], [np x nvol]
    fib::DevBuf<float> coef;            // [nvol][8] pA columns + b0 flag
    fib::DevBuf<float> design;          // [nvol][8] A rows
    mutable fib::DevBuf<int64_t> partial_list;
    mutable fib::DevBuf<int> partial_count;
};

extern "C" int fib_dti_plan_create(int device, const float *bval, const float *bvec, int nvol, fib_dti_plan **plan) try {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan output pointer is NULL");
    *plan = nullptr;
    FIB_CHECK(bval != nullptr && nvol > 0, FIB_ERR_MISSING_BVAL, "Missing b-value table from input DWI structure");
    fib::DeviceGuard guard;
    int rc = fib::use_device(device);
    if (rc != FIB_OK) return rc;
    fib_dti_plan *p = new (std::nothrow) fib_dti_plan();
    FIB_CHECK(p != nullptr, FIB_ERR_NOMEM, "out of host memory");
    p->device = device;
    p->nvol = nvol;
    p->np = bvec ? 7 : 2;
    const int np = p->np;
    p->A.resize((size_t)nvol * np);
    p->pA.resize((size_t)nvol * np);
    fib::host_dti_design(bval, bvec, nvol, np, p->A.data());
    fib::host_pinv(p->A.data(), nvol, np, p->pA.data());
    float bmin = bval[0];
    for (int i = 1; i < nvol; i++) bmin = bval[i] < bmin ? bval[i] : bmin;
    std::vector<float> coef((size_t)nvol * 8, 0.0f), design((size_t)nvol * 8, 0.0f);
    for (int i = 0; i < nvol; i++) {
        for (int j = 0; j < np; j++) {
            coef[(size_t)8 * i + j] = p->pA[j + (size_t)np * i];
            design[(size_t)8 * i + j] = p->A[i + (size_t)nvol * j];
        }
        coef[(size_t)8 * i + 7] = (bval[i] == bmin) ? 1.0f : 0.0f;   // ib0, dti.jl:117
    }
    rc = p->coef.alloc(coef.size());
    if (rc == FIB_OK) rc = p->design.alloc(design.size());
    if (rc == FIB_OK) rc = p->partial_count.alloc(2);
    if (rc != FIB_OK) { delete p; return rc; }
    hipError_t e = hipMemcpy(p->coef.p, coef.data(), coef.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->design.p, design.data(), design.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(p->partial_count.p, 0, 2 * sizeof(int));
    if (e != hipSuccess) { delete p; return fib::fail(FIB_ERR_HIP, "plan upload failed: %s", hipGetErrorString(e)); }
    *plan = p;
    return FIB_OK;
} FIB_API_CATCH

extern "C" void fib_dti_plan_destroy(fib_dti_plan *plan) try {
    if (!plan) return;
    fib::DeviceGuard guard;
    (void)hipSetDevice(plan->device);
    delete plan;
} FIB_API_CATCH_VOID

extern "C" int fib_dti_plan_tables(const fib_dti_plan *plan, float *A, float *pA, int *np) try {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan is NULL");
    if (np) *np = plan->np;
    if (A) memcpy(A, plan->A.data(), plan->A.size() * sizeof(float));
    if (pA) memcpy(pA, plan->pA.data(), plan->pA.size() * sizeof(float));
    return FIB_OK;
} FIB_API_CATCH

namespace {

template <int NP>
int launch_fit(const fib_dti_plan *plan, const float *dwi, const uint8_t *mask, int64_t nvox,
               const DtiOutPtrs &o, float *adc, hipStream_t st) {
    FIB_CHECK(nvox > 0, FIB_ERR_INVALID, "nvox must be positive");
    FIB_CHECK(nvox < ((int64_t)1 << 31), FIB_ERR_UNSUPPORTED, "volumes of 2^31 voxels or more are not supported");
    int rc = plan->partial_list.ensure((size_t)nvox);
    if (rc != FIB_OK) return rc;
    FIB_HIP(hipMemsetAsync(plan->partial_count.p, 0, 2 * sizeof(int), st));
    // measured on MI355X (140^3 x 64): 4 voxels / lane (16-byte loads, 4 waves / SIMD) 0.185 ms; 2 -> 0.171 ms; 1 voxel / lane
    // (4-byte loads but 7 waves / SIMD and 16 frames in flight per lane) 0.165 ms, 0.155 with non-temporal accesses: the only
    // form that is built; block sizes 64-256 and 8-32 frames in flight made no difference
    { fib::ProfScope prof(NP == 7 ? "dti_fit" : "adc_fit", st);
    const int block = 256;
    const unsigned grid = (unsigned)fib::cdiv(nvox, block);
    hipLaunchKernelGGL((fit_kernel<NP, 1, 16>), dim3(grid), dim3(block), 0, st, dwi, mask, plan->coef.p, plan->nvol, nvox, o, adc, plan->partial_list.p, plan->partial_count.p);
    }
    FIB_HIP(hipGetLastError());
    fib::ProfScope prof2("fit_partial", st);
    // rare path: fixed small grid, grid-stride over the device-side list (length known only on device)
    const unsigned pgrid = (unsigned)std::min<int64_t>(fib::cdiv(nvox, 64), 2048);
    hipLaunchKernelGGL((fit_partial_kernel<NP>), dim3(pgrid), dim3(64), 0, st, dwi, plan->design.p, plan->coef.p,
                       plan->nvol, nvox, plan->partial_list.p, plan->partial_count.p, o, adc);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
}

}  // namespace

extern "C" int fibd_dti_fit(const fib_dti_plan *plan, const float *dwi, const uint8_t *mask, int64_t nvox,
                            const fib_dti_out *out, void *stream) try {
    FIB_CHECK(plan && dwi && mask && out, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(plan->np == 7, FIB_ERR_MISSING_BVEC, "Missing gradient table from input DWI structure");
    FIB_CHECK(out->s0 && out->eigval1 && out->eigval2 && out->eigval3 && out->eigvec1 && out->eigvec2 &&
              out->eigvec3 && out->rd && out->md && out->fa, FIB_ERR_INVALID, "NULL output volume");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    DtiOutPtrs o{out->s0, out->eigval1, out->eigval2, out->eigval3, out->eigvec1, out->eigvec2, out->eigvec3,
                 out->rd, out->md, out->fa};
    return launch_fit<7>(plan, dwi, mask, nvox, o, nullptr, (hipStream_t)stream);
} FIB_API_CATCH

extern "C" int fibd_adc_fit(const fib_dti_plan *plan, const float *dwi, const uint8_t *mask, int64_t nvox,
                            float *adc, float *s0, void *stream) try {
    FIB_CHECK(plan && dwi && mask && adc && s0, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(plan->np == 2, FIB_ERR_INVALID, "plan was not created as an ADC plan (bvec == NULL)");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    DtiOutPtrs o{};
    o.s0 = s0;
    return launch_fit<2>(plan, dwi, mask, nvox, o, adc, (hipStream_t)stream);
} FIB_API_CATCH

// st_eigen (structens.jl:13-37): eigen(Symmetric(S, :L)) of the structure tensor of every voxel, the same closed form as
// the diffusion tensor's (dti.jl:311).  eigval [nvox*3] ascending, eigvec [nvox*9]: component i of eigenvector j at
// (i + 3 j) * nvox + vox -- the column-major image of the reference's eigvec[ix, iy, iz, i, j].  HBM-bound: 24 B in, 48 B out.
namespace {
struct StIn { const float *s[6]; };
__global__ __launch_bounds__(256) void st_eigen_kernel(const StIn in, int64_t nvox, float *__restrict__ eigvec, float *__restrict__ eigval) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    float w[3], ev[3][3];
    sym3_eigen(in.s[0][i], in.s[1][i], in.s[2][i], in.s[3][i], in.s[4][i], in.s[5][i], w, ev);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        eigval[(int64_t)j * nvox + i] = w[j];
#pragma unroll
        for (int c = 0; c < 3; c++) eigvec[(int64_t)(c + 3 * j) * nvox + i] = ev[j][c];
    }
}
}  // namespace

extern "C" int fibd_st_eigen(const float *const S[6], int64_t nvox, float *eigvec, float *eigval, void *stream) try {
    FIB_CHECK(S && eigvec && eigval, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nvox > 0, FIB_ERR_INVALID, "nvox must be positive");
    StIn in{};
    for (int k = 0; k < 6; k++) { FIB_CHECK(S[k] != nullptr, FIB_ERR_INVALID, "NULL structure tensor volume %d", k); in.s[k] = S[k]; }
    fib::ProfScope prof("st_eigen", (hipStream_t)stream);
    hipLaunchKernelGGL(st_eigen_kernel, dim3((unsigned)fib::cdiv(nvox, 256)), dim3(256), 0, (hipStream_t)stream, in, nvox, eigvec, eigval);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fibd_dti_last_partial_count(const fib_dti_plan *plan, void *stream, int64_t *count) try {
    FIB_CHECK(plan && count, FIB_ERR_INVALID, "NULL argument");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    int c = 0;
    FIB_HIP(hipMemcpyAsync(&c, plan->partial_count.p + 1, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    FIB_HIP(hipStreamSynchronize((hipStream_t)stream));
    *count = c;
    return FIB_OK;
} FIB_API_CATCH
