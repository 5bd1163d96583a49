// odf.hip — K2..K5: GQI / DSI ODF reconstruction, ODF peak finder, global QA normalisation (gfx950).
//
//   K2/K5  odf_gemm_kernel   O[M x Nvox] = A[M x K] * clamp(S[K x Nvox])   on v_mfma_f32_32x32x2_f32
//                            (gqi.jl:139-145 `mul!(o, A, s)`; dsi.jl:204-246 recast as two dense maps)
//   K3     odf_peaks_kernel  find_peaks! + peak/qa extraction (gqi.jl:147-159,180-201; dsi.jl:244-258)
//   K4     max-of-means reduction + qa ./= odfmax (gqi.jl:164-168; dsi.jl:263-267)
//
// GEMM design.  M (ODF vertices, 321 for sphere_642) is small, K (frames, 270) is small, N (voxels,
// 2.7 M) is huge and contiguous in memory for both S (planar frames) and O (planar vertices).  So the
// voxel index sits on the MFMA lane (N = column): a wave owns 32 voxels and ALL rows of its M tile;
// accumulators stay in registers for the whole K loop (MB blocks of 32x32 = 16*MB VGPRs), S is read
// exactly once straight into VGPRs as the B operand (two 128-B segments per wave load), and the only
// shared operand, the matrix A (347 KB: larger than LDS), is streamed K-tile by K-tile through a
// double-buffered LDS ring with direct-to-LDS loads (global_load_lds), laid out K-major so that the
// A-fragment ds_read_b32 is bank-conflict free.  f32-input MFMA is exact f32 (k-ordered fma chain):
// the ODF matches a CPU sgemv to rounding, which the strict-inequality peak finder needs.
#include <algorithm>
#include <cmath>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KT = 16;        // frames per LDS stage
constexpr int WG_VOX = 128;   // voxels per workgroup (4 waves x 32)

struct GemmArgs {
    const float *At;          // [ntile_m][Kpad][MW]  K-major tiles (MW = MB*32 [+16 when extra rows]), zero padded
    const float *S;           // [K][nvox] planar DWI
    const uint8_t *mask;      // [nvox]
    const uint32_t *effbits;  // [Kpad/KT] bit j of word t: frame t*KT+j exists and takes part in the "any positive sample" test
    float *out0;              // rows [0, nrow0)        (DSI: pdf)
    float *out1;              // rows [nrow0, M)        (odf)
    int64_t nvox;
    int K, Kpad, M, nrow0, ntile_m;
    int scale_frame;          // DSI: frame whose clamped sample times scale_coef is sum(p); -1: no scaling
    float scale_coef;
    int has_ineff;
};

// scheduling hint: spread one k-step's fragment reads (ds_read2_b32 = 2 fragments) between the previous
// k-step's MFMAs instead of "read, wait, 2 MFMA" chains (hipcc otherwise minimises live registers)
template <int MB>
__device__ __forceinline__ void interleave_ds_mfma() {
#pragma unroll
    for (int i = 0; i < MB / 2; i++) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
    }
    if (MB & 1) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// On gfx950 the f32-input MFMA runs on the same FMA lanes as the vector ALU: every VALU instruction a wave
// issues costs the SIMD ~4 cycles of MFMA time whether it sits between MFMAs or after them (measured with
// tools/probes/mfma_probe.hip: +264 v_add per 88 MFMAs = +12 %).  So the K loop is written to issue almost
// no VALU work: clamp / running-max / non-finite tracking in 3 ops per sample, scalar-base + 32-bit-lane-
// offset addressing for every global access, fragment reads by immediate LDS offsets.
//
// Rows per tile = MB*32 + NX: MB 32-row MFMA blocks plus NX "extra" rows done as one v_fmac per k-step each
// (4 cycles instead of a 64-cycle MFMA block that would be 31/32 padding: sphere_642 has 321 = 10*32 + 1
// half-sphere vertices).  The extra rows sum even and odd frames in the two lane halves and add the halves
// at the end, so their rounding differs from the k-ordered MFMA chain by ~1 ulp.
template <int MB, int NX>
__global__ __launch_bounds__(256, 2) void odf_gemm_kernel(const GemmArgs a) {
    constexpr int MW = MB * 32 + (NX > 0 ? 16 : 0);    // LDS row stride (floats); multiple of 16 -> whole 1-KiB pieces
    constexpr int TILE = KT * MW;                       // floats per stage
    constexpr int NPIECE = TILE * 4 / 1024;
    constexpr int ROWS = MB * 32 + NX;                  // output rows per M tile
    static_assert((TILE * 4) % 1024 == 0, "stage must be a whole number of 1-KiB pieces");
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kh = lane >> 5;
    const int tile_m = blockIdx.x % a.ntile_m;
    const int64_t tile_n = blockIdx.x / a.ntile_m;
    const int64_t vox0 = tile_n * WG_VOX + wave * 32;   // wave-uniform
    const int64_t vox = vox0 + col;
    const bool inb = vox < a.nvox;
    const int ntiles = a.Kpad / KT;
    // per-lane 32-bit byte offsets; everything else in an address is wave-uniform (SGPR base)
    const uint32_t c_off = (uint32_t)((inb ? col : 0) * 4);
    const uint32_t s_off = (uint32_t)(((inb ? col : 0) + (int64_t)kh * a.nvox) * 4);   // frame kh of the pair, this voxel
    const char *Sbase = reinterpret_cast<const char *>(a.S + (vox0 < a.nvox ? vox0 : 0));
    const char *Abase = reinterpret_cast<const char *>(a.At + (size_t)tile_m * a.Kpad * MW);
    const uint32_t a_off = (uint32_t)lane * 16;
    const int64_t frame_pair_bytes = 2 * a.nvox * 4;
    const uint8_t mk = a.mask[inb ? vox : 0];           // used in the epilogue only: latency hidden

    auto stage_A = [&](int t, int buf) {                // one stage = TILE*4 contiguous bytes of At
        const char *g = Abase + (size_t)t * TILE * 4;
        char *l = reinterpret_cast<char *>(lds + buf * TILE);
        for (int p = wave; p < NPIECE; p += 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + p * 1024 + a_off),
                                             (__attribute__((address_space(3))) void *)(l + p * 1024), 16, 0, 0);
    };
    // B operand: KT/2 unconditional loads per stage (frame index clamped to K-1: the padded rows of At are zero)
    float braw[KT / 2];
    auto load_B = [&](int t) {
#pragma unroll
        for (int kk = 0; kk < KT / 2; kk++) {
            int kpair = t * (KT / 2) + kk;              // frames 2*kpair, 2*kpair+1 (lane half kh picks one)
            const int lastpair = (a.K - 1) / 2;
            kpair = kpair < lastpair ? kpair : lastpair;
            const uint32_t off = (2 * kpair + 1 >= a.K) ? c_off : s_off;   // odd K: the last pair has one frame only
            braw[kk] = *reinterpret_cast<const float *>(Sbase + (int64_t)kpair * frame_pair_bytes + off);
        }
    };
    auto load_A = [&](const float *L, int kk, float (&af)[MB]) {   // MB conflict-free ds_read_b32, immediate offsets
#pragma unroll
        for (int m = 0; m < MB; m++) af[m] = L[2 * kk * MW + m * 32];
    };

    f32x16 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
    float xacc[NX > 0 ? NX : 1];
#pragma unroll
    for (int x = 0; x < NX; x++) xacc[x] = 0.0f;
    float vmax = 0.0f;                                  // running max of the samples  -> "any sample > 0"
    float vnf = 0.0f;                                   // becomes NaN once a sample is NaN or +-Inf

    stage_A(0, 0);
    load_B(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ntiles; t++) {
        const int cur = t & 1;
        // clamp this stage's samples (gqi.jl:140, dsi.jl:209); track positivity (gqi.jl:142, dsi.jl:207) and NaN/Inf
        float bcur[KT / 2];
        if (a.has_ineff) {                              // rare: frames that never reach the model must not count
            const uint32_t eff = a.effbits[t] >> kh;
#pragma unroll
            for (int kk = 0; kk < KT / 2; kk++) {
                const float s = braw[kk];
                bcur[kk] = fmaxf(s, 0.0f);
                vmax = fmaxf(vmax, ((eff >> (2 * kk)) & 1u) ? s : 0.0f);
                vnf = __builtin_fmaf(s, 0.0f, vnf);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KT / 2; kk++) {
                const float s = braw[kk];
                bcur[kk] = fmaxf(s, 0.0f);
                vmax = fmaxf(vmax, s);
                vnf = __builtin_fmaf(s, 0.0f, vnf);
            }
        }
        if (t + 1 < ntiles) {
            stage_A(t + 1, cur ^ 1);
            load_B(t + 1);
        }
        const float *L = lds + cur * TILE + kh * MW + col;
        const float *LX = lds + cur * TILE + kh * MW + MB * 32;     // extra rows: same address in a lane half (broadcast)
        // software pipeline over the k-steps: fragments of step kk+1 are read while step kk's MFMAs issue
        float a0[MB], a1[MB];
        load_A(L, 0, a0);
        __builtin_amdgcn_sched_group_barrier(0x100, (MB + 1) / 2, 0);   // the first step's reads lead the block
#pragma unroll
        for (int kk = 0; kk < KT / 2; kk += 2) {
            load_A(L, kk + 1, a1);
#pragma unroll
            for (int m = 0; m < MB; m++) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], bcur[kk], acc[m], 0, 0, 0);
            interleave_ds_mfma<MB>();
            if (kk + 2 < KT / 2) load_A(L, kk + 2, a0);
#pragma unroll
            for (int m = 0; m < MB; m++) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], bcur[kk + 1], acc[m], 0, 0, 0);
            interleave_ds_mfma<MB>();
        }
        if (NX > 0) {
#pragma unroll
            for (int kk = 0; kk < KT / 2; kk++)
#pragma unroll
                for (int x = 0; x < NX; x++) xacc[x] = __builtin_fmaf(LX[2 * kk * MW + x], bcur[kk], xacc[x]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next stage's direct-to-LDS loads and samples have landed
        __syncthreads();
    }

    // ---- epilogue: the two k-halves of a voxel live in lanes l and l^32 -----------------------------------
    float pm = fmaxf(vmax, __shfl_xor(vmax, 32));
    float pn = vnf + __shfl_xor(vnf, 32);
    const bool nonfinite = pn != pn;                    // a NaN sample makes every output NaN (NaN * A[v,i] for all v)
    const bool valid = inb && (pm > 0.0f || nonfinite) && mk != 0;
    const bool do_scale = a.scale_frame >= 0;
    float scale = 1.0f;
    if (do_scale) {
        float s = *reinterpret_cast<const float *>(Sbase + (int64_t)a.scale_frame * a.nvox * 4 + c_off);
        s = s < 0.0f ? 0.0f : s;
        scale = 1.0f / (a.scale_coef * s);              // p ./ sum(p), dsi.jl:225 (0 -> Inf/NaN like the reference)
    }
    if (nonfinite) scale = __builtin_nanf("");
    const bool plain = __all(valid && !nonfinite) && !do_scale;   // wave-uniform: store the accumulators as they are
    const float mulv = valid ? scale : 0.0f;
#pragma unroll
    for (int x = 0; x < NX; x++) xacc[x] += __shfl_xor(xacc[x], 32);
    if (!inb) return;
    const uint32_t o_off = (uint32_t)((col + (int64_t)4 * kh * a.nvox) * 4);
    auto row_ptr = [&](int row) -> char * {             // wave-uniform row base for this wave's 32 voxels
        return reinterpret_cast<char *>(row >= a.nrow0 ? a.out1 + (int64_t)(row - a.nrow0) * a.nvox + vox0
                                                       : a.out0 + (int64_t)row * a.nvox + vox0);
    };
#pragma unroll
    for (int m = 0; m < MB; m++) {
        const int row0 = tile_m * ROWS + m * 32;        // wave-uniform
        if (row0 >= a.M) break;
        const bool whole = row0 + 32 <= a.M && (row0 >= a.nrow0 || row0 + 32 <= a.nrow0);   // uniform fast path
        char *base = row_ptr(row0);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int dr = (r & 3) + 8 * (r >> 2);      // row within the block, before the lane-half offset
            float v = acc[m][r];
            if (!plain) v = valid ? v * mulv : 0.0f;
            if (whole) {
                *reinterpret_cast<float *>(base + (int64_t)dr * a.nvox * 4 + o_off) = v;
            } else {
                const int row = row0 + dr + 4 * kh;
                if (row >= a.M) continue;
                *reinterpret_cast<float *>(row_ptr(row) + col * 4) = v;
            }
        }
    }
#pragma unroll
    for (int x = 0; x < NX; x++) {
        const int row = tile_m * ROWS + MB * 32 + x;    // wave-uniform
        if (row >= a.M) break;
        float v = xacc[x];
        if (!plain) v = valid ? v * mulv : 0.0f;
        if (kh == 0) *reinterpret_cast<float *>(row_ptr(row) + col * 4) = v;
    }
}

// ------------------------------------------------------------------------------------------
// peak finder
// ------------------------------------------------------------------------------------------
// Tile = 32 voxels x all vertices in LDS ([row][32] floats, 41 KB for sphere_642 -> 3 workgroups per CU so
// one group's HBM load overlaps the others' LDS-bound scans).  A wave scans TWO vertices at a time: lanes
// 0-31 hold the 32 voxels for vertex 2i, lanes 32-63 for vertex 2i+1, so the neighbour indices are
// wave-uniform per half (scalar loads + one select) and every ds_read_b32 is bank-conflict free.
// Unused neighbour slots point at a sentinel row of NaNs: `NaN >= x` is false, so no branch is needed.
constexpr int PV = 32;        // voxels per workgroup tile
constexpr int PW = 4;         // waves per workgroup
constexpr int PG = 2 * PW;    // vertex groups (wave, half)
constexpr int PREC = 10;      // floats per merge record

__device__ __forceinline__ bool jl_isless(float x, float y) {   // Base.isless on floats
    if (x != x) return false;
    if (y != y) return true;
    if (x == y) return (__float_as_uint(x) >> 31) && !(__float_as_uint(y) >> 31);
    return x < y;
}
// position in sortperm!(…, rev=true): descending value, ties keep ascending index (gqi.jl:198)
__device__ __forceinline__ bool sorts_before(float va, int ia, float vb, int ib) {
    return jl_isless(vb, va) || (!jl_isless(va, vb) && ia < ib);
}

struct Top3 { float v[3]; int i[3]; };
__device__ __forceinline__ void top3_insert(Top3 &t, float x, int idx) {
    // entries are kept sorted; empty slots (i < 0) only at the tail
    const bool b2 = t.i[2] >= 0 && !sorts_before(x, idx, t.v[2], t.i[2]);
    if (b2) return;
    const bool b1 = t.i[1] >= 0 && !sorts_before(x, idx, t.v[1], t.i[1]);
    const bool b0 = t.i[0] >= 0 && !sorts_before(x, idx, t.v[0], t.i[0]);
    if (b1) { t.v[2] = x; t.i[2] = idx; return; }
    t.v[2] = t.v[1]; t.i[2] = t.i[1];
    if (b0) { t.v[1] = x; t.i[1] = idx; return; }
    t.v[1] = t.v[0]; t.i[1] = t.i[0];
    t.v[0] = x; t.i[0] = idx;
}

struct PeakArgs {
    const float *odf;         // [nvert][nvox]
    const int32_t *nbr;       // [nvert_even][DEG]: row index of each neighbour, unused slots = sentinel row
    const float *verts;       // [nvert][3] first-half vertex coordinates (gqi.jl:155)
    float *peak[3];           // [3][nvox] each (or NULL in find-peaks mode)
    float *qa[3];             // [nvox] each
    int32_t *isort_top;       // [3][nvox] (find-peaks mode) or NULL
    int32_t *nvalid;          // [nvox]    (find-peaks mode) or NULL
    unsigned *maxenc;         // [2]: ordered-uint max of per-voxel means, NaN flag (may be NULL)
    int64_t nvox;
    int nvert, rows_pad;      // rows_pad = nvert rounded up to 8; sentinel row index = rows_pad
    int vec_ok;               // 1: every tile row is 16-byte aligned (nvox % 4 == 0 and aligned base)
};

__device__ __forceinline__ unsigned enc_ordered(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(unsigned e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

// DEG = padded neighbour count per vertex; EXACT: keep the full sortperm order (find_peaks! API) instead of
// only the entries gqi_rec/dsi_rec can use (positive or NaN survivors)
template <int DEG, bool EXACT>
__global__ __launch_bounds__(PW * 64) void odf_peaks_kernel(const PeakArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *o = smem;                                            // [rows_pad + 1][PV]
    float *mrg = o + (size_t)(a.rows_pad + 1) * PV;             // [PG][PV][PREC]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & (PV - 1), half = lane >> 5;
    const int64_t vox0 = (int64_t)blockIdx.x * PV;
    const int64_t vox = vox0 + j;
    const bool inb = vox < a.nvox;
    const bool full = vox0 + PV <= a.nvox;

    // ---- load the tile: 8 rows x 128 B per direct-to-LDS wave instruction --------------------------
    if (a.vec_ok && full) {
        const int npiece = a.rows_pad / 8;
        for (int p = wave; p < npiece; p += PW) {
            int row = 8 * p + (lane >> 3);
            row = row < a.nvert ? row : a.nvert - 1;            // padding rows: any valid address
            const float *g = a.odf + (int64_t)row * a.nvox + vox0 + 4 * (lane & 7);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(o + p * 256), 16, 0, 0);
        }
    } else {
        for (int r = wave * 2 + half; r < a.rows_pad; r += PG)
            o[r * PV + j] = (inb && r < a.nvert) ? a.odf[(int64_t)r * a.nvox + vox] : 0.0f;
    }
    if (tid < PV) o[a.rows_pad * PV + tid] = __builtin_nanf("");   // sentinel row
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- scan: this wave's vertex pairs -------------------------------------------------------------
    Top3 t;
#pragma unroll
    for (int k = 0; k < 3; k++) { t.v[k] = 0.0f; t.i[k] = -1; }
    int npos = 0;
    float vmin = INFINITY, vsum = 0.0f;
    bool hasnan = false;
    const int npair = (a.nvert + 1) / 2;
    for (int pi = wave; pi < npair; pi += PW) {
        const int v = 2 * pi + half;
        const bool live = v < a.nvert;
        const int32_t *nb = a.nbr + (size_t)2 * pi * DEG;       // wave-uniform: rows 2pi and 2pi+1
        const float x = o[(live ? v : a.rows_pad) * PV + j];
        float y[DEG];
#pragma unroll
        for (int d = 0; d < DEG; d++) {
            const int ua = nb[d], ub = nb[DEG + d];
            y[d] = o[(half ? ub : ua) * PV + j];
        }
        bool killed = false;
#pragma unroll
        for (int d = 0; d < DEG; d++) killed |= (y[d] >= x);    // gqi.jl:185-196: o[b] >= o[a] || o[c] >= o[a]
        if (live) {
            const float pk = killed ? 0.0f : x;                 // odf_peak (gqi.jl:184-196)
            if (pk > 0.0f) npos++;                              // gqi.jl:200
            if (EXACT || !(pk <= 0.0f)) top3_insert(t, pk, v);  // positive or NaN entries lead the sort order
            hasnan |= (x != x);
            vmin = fminf(vmin, x);
            vsum += x;
        }
    }
    // ---- merge the PG partial results of each voxel --------------------------------------------------
    {
        float *rec = mrg + (size_t)((wave * 2 + half) * PV + j) * PREC;
#pragma unroll
        for (int k = 0; k < 3; k++) { rec[k] = t.v[k]; rec[3 + k] = __int_as_float(t.i[k]); }
        rec[6] = __int_as_float(npos); rec[7] = vmin; rec[8] = vsum; rec[9] = hasnan ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (tid >= 64) return;
    float mean = 0.0f;
    bool mean_nan = false;
    const bool owner = half == 0;
    if (owner) {
        for (int gg = 1; gg < PG; gg++) {
            const float *r = mrg + (size_t)(gg * PV + j) * PREC;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int idx = __float_as_int(r[3 + k]);
                if (idx >= 0) top3_insert(t, r[k], idx);
            }
            npos += __float_as_int(r[6]);
            vmin = fminf(vmin, r[7]);
            vsum += r[8];
            hasnan |= r[9] != 0.0f;
        }
        if (hasnan) vmin = NAN;                                 // minimum() propagates NaN (gqi.jl:147)
        mean = vsum * (1.0f / (float)a.nvert);                  // mean(odf, dims=4), gqi.jl:164
        mean_nan = mean != mean;
        if (inb) {
            if (a.isort_top) {
#pragma unroll
                for (int k = 0; k < 3; k++) a.isort_top[(int64_t)k * a.nvox + vox] = t.i[k];
                a.nvalid[vox] = npos;
            } else {
                const int n = npos < 3 ? npos : 3;              // gqi.jl:151
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float px = 0.0f, py = 0.0f, pz = 0.0f, q = 0.0f;
                    if (k < n) {
                        const int iv = t.i[k];
                        px = a.verts[3 * iv]; py = a.verts[3 * iv + 1]; pz = a.verts[3 * iv + 2];
                        q = o[iv * PV + j] - vmin;              // gqi.jl:157-158
                    }
                    a.peak[k][vox] = px; a.peak[k][a.nvox + vox] = py; a.peak[k][2 * a.nvox + vox] = pz;
                    a.qa[k][vox] = q;
                }
            }
        }
    }
    if (a.maxenc) {
        const bool mine = owner && inb;
        unsigned e = mine && !mean_nan ? enc_ordered(mean) : 0u;
        const unsigned long long nanb = __ballot(mine && mean_nan);
        for (int off = 32; off >= 1; off >>= 1) { const unsigned oth = (unsigned)__shfl_xor((int)e, off); e = oth > e ? oth : e; }
        if (tid == 0) {
            if (e) atomicMax(&a.maxenc[0], e);
            if (nanb) atomicOr(&a.maxenc[1], 1u);
        }
    }
}

__global__ void odfmax_finalize_kernel(const unsigned *enc, float *out) {
    const bool nan = enc[1] != 0;
    const float m = enc[0] ? dec_ordered(enc[0]) : -INFINITY;
    out[0] = nan ? NAN : m;                                     // maximum() propagates NaN
    out[1] = nan ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void qa_normalize_kernel(float *q0, float *q1, float *q2, int64_t nvox,
                                                          const float *odfmax_dev, float odfmax_val) {
    const float d = odfmax_dev ? odfmax_dev[0] : odfmax_val;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (int64_t)gridDim.x * blockDim.x) {
        q0[i] = q0[i] / d;                                      // qa[ipeak].vol /= odfmax, gqi.jl:167
        q1[i] = q1[i] / d;
        q2[i] = q2[i] / d;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
struct fib_odf_plan {
    int device = 0;
    int nvol = 0, nvert = 0, nrows = 0, nrow0 = 0;   // nrow0 = rows that go to the pdf output (DSI), else 0
    int MB = 11, NX = 0, ntile_m = 1, Kpad = 0, maxdeg = 0;   // M tile = MB 32-row MFMA blocks + NX VALU rows
    int scale_frame = -1;
    float scale_coef = 0.0f;
    bool has_ineff = false;                          // some frame never reaches the model (DSI duplicates)
    std::vector<float> A;                            // host copy [nrows x nvol] column-major
    fib::DevBuf<float> At, verts;
    fib::DevBuf<uint32_t> effbits;
    fib::DevBuf<int32_t> nbr;        // [nvert_even][deg_pad] LDS row of each neighbour (sentinel-padded)
    int deg_pad = 6, rows_pad = 0;
    mutable fib::DevBuf<unsigned> maxenc;
    mutable fib::DevBuf<float> odfmax;
};

namespace {


int finish_plan(fib_odf_plan *p, const float *verts, int nverts, const int32_t *faces, int nfaces,
                const std::vector<float> &frame_eff) {
    const int M = p->nrows, K = p->nvol;
    // pick (MB, NX) minimising the per-k-step issue cost ntile*(64*MB + 4*NX) cycles (MFMA block = 64, v_fmac = 4)
    int best_cost = INT32_MAX;
    const int nxs[] = {0, 1, 2, 4};
    for (int mb = 11; mb >= 6; mb--)
        for (int nx : nxs) {
            if (nx > 0 && mb > 10) continue;            // register budget
            const int rows = mb * 32 + nx;
            const int nt = (M + rows - 1) / rows;
            const int cost = nt * (64 * mb + 4 * nx);
            if (cost < best_cost) { best_cost = cost; p->MB = mb; p->NX = nx; p->ntile_m = nt; }
        }
    p->Kpad = (K + KT - 1) / KT * KT;
    const int MW = p->MB * 32 + (p->NX > 0 ? 16 : 0), ROWS = p->MB * 32 + p->NX;
    std::vector<float> At((size_t)p->ntile_m * p->Kpad * MW, 0.0f);
    for (int k = 0; k < K; k++)
        for (int r = 0; r < M; r++) {
            const int tm = r / ROWS, rr = r % ROWS;
            At[((size_t)tm * p->Kpad + k) * MW + rr] = p->A[r + (size_t)M * k];
        }
    std::vector<int32_t> nbr32;
    int rc = fib::host_neighbours(faces, nfaces, nverts, nbr32, &p->maxdeg);
    if (rc != FIB_OK) return rc;
    p->deg_pad = p->maxdeg <= 6 ? 6 : (p->maxdeg <= 8 ? 8 : 16);
    p->rows_pad = (p->nvert + 7) / 8 * 8;
    const int nv_even = (p->nvert + 1) / 2 * 2;
    std::vector<int32_t> nbr((size_t)nv_even * p->deg_pad, p->rows_pad);          // default: sentinel row
    for (int v = 0; v < p->nvert; v++)
        for (int d = 0; d < p->maxdeg; d++) {
            const int32_t u = nbr32[(size_t)v * p->maxdeg + d];
            if (u >= 0) nbr[(size_t)v * p->deg_pad + d] = u;
        }
    std::vector<float> v3((size_t)p->nvert * 3);
    for (int v = 0; v < p->nvert; v++)
        for (int c = 0; c < 3; c++) v3[3 * v + c] = verts[v + (size_t)nverts * c];
    if ((rc = p->At.alloc(At.size())) != FIB_OK) return rc;
    std::vector<uint32_t> effbits((size_t)p->Kpad / KT, 0u);
    for (int k = 0; k < K; k++) { if (frame_eff[k] != 0.0f) effbits[k / KT] |= 1u << (k % KT); else p->has_ineff = true; }
    if ((rc = p->effbits.alloc(effbits.size())) != FIB_OK) return rc;
    if ((rc = p->verts.alloc(v3.size())) != FIB_OK) return rc;
    if ((rc = p->nbr.alloc(nbr.size())) != FIB_OK) return rc;
    if ((rc = p->maxenc.alloc(2)) != FIB_OK) return rc;
    if ((rc = p->odfmax.alloc(2)) != FIB_OK) return rc;
    FIB_HIP(hipMemcpy(p->At.p, At.data(), At.size() * sizeof(float), hipMemcpyHostToDevice));
    FIB_HIP(hipMemcpy(p->effbits.p, effbits.data(), effbits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    FIB_HIP(hipMemcpy(p->verts.p, v3.data(), v3.size() * sizeof(float), hipMemcpyHostToDevice));
    FIB_HIP(hipMemcpy(p->nbr.p, nbr.data(), nbr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return FIB_OK;
}

int check_plan_args(const float *bval, const float *bvec, int nvol, const float *verts, int nverts,
                    const int32_t *faces, int nfaces, fib_odf_plan **plan) {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan output pointer is NULL");
    *plan = nullptr;
    FIB_CHECK(bval != nullptr && nvol > 0, FIB_ERR_MISSING_BVAL, "Missing b-value table from input DWI structure");
    FIB_CHECK(bvec != nullptr, FIB_ERR_MISSING_BVEC, "Missing gradient table from input DWI structure");
    FIB_CHECK(verts && faces && nverts >= 2 && nverts % 2 == 0 && nfaces > 0, FIB_ERR_INVALID, "invalid ODF tessellation");
    FIB_CHECK(nverts / 2 < 32768, FIB_ERR_UNSUPPORTED, "too many ODF vertices");
    return FIB_OK;
}

}  // namespace

extern "C" int fib_gqi_plan_create(int device, const float *bval, const float *bvec, int nvol,
                                   const float *verts, int nverts, const int32_t *faces, int nfaces,
                                   float sigma, fib_odf_plan **plan) {
    int rc = check_plan_args(bval, bvec, nvol, verts, nverts, faces, nfaces, plan);
    if (rc != FIB_OK) return rc;
    fib::DeviceGuard guard;
    if ((rc = fib::use_device(device)) != FIB_OK) return rc;
    fib_odf_plan *p = new (std::nothrow) fib_odf_plan();
    FIB_CHECK(p != nullptr, FIB_ERR_NOMEM, "out of host memory");
    p->device = device; p->nvol = nvol; p->nvert = nverts / 2; p->nrows = p->nvert; p->nrow0 = 0;
    p->A.resize((size_t)p->nrows * nvol);
    fib::host_gqi_matrix(bval, bvec, nvol, verts, nverts, sigma, p->A.data());
    std::vector<float> eff((size_t)nvol, 1.0f);
    rc = finish_plan(p, verts, nverts, faces, nfaces, eff);
    if (rc != FIB_OK) { delete p; return rc; }
    *plan = p;
    return FIB_OK;
}

extern "C" int fib_dsi_plan_create(int device, const float *bval, const float *bvec, int nvol,
                                   const float *verts, int nverts, const int32_t *faces, int nfaces,
                                   int hann_width, fib_odf_plan **plan) {
    int rc = check_plan_args(bval, bvec, nvol, verts, nverts, faces, nfaces, plan);
    if (rc != FIB_OK) return rc;
    FIB_CHECK(hann_width >= 0, FIB_ERR_INVALID, "hann_width must be >= 0");
    fib::DeviceGuard guard;
    if ((rc = fib::use_device(device)) != FIB_OK) return rc;
    fib_odf_plan *p = new (std::nothrow) fib_odf_plan();
    FIB_CHECK(p != nullptr, FIB_ERR_NOMEM, "out of host memory");
    p->device = device; p->nvol = nvol; p->nvert = nverts / 2; p->nrows = nvol + p->nvert; p->nrow0 = nvol;
    p->A.resize((size_t)p->nrows * nvol);
    rc = fib::host_dsi_matrix(bval, bvec, nvol, verts, nverts, hann_width, p->A.data(), &p->scale_frame, &p->scale_coef);
    if (rc != FIB_OK) { delete p; return rc; }
    // a frame overwritten by a later one on the same lattice point never reaches X (dsi.jl:205): its column is 0
    std::vector<float> eff((size_t)nvol, 0.0f);
    for (int j = 0; j < nvol; j++)
        for (int r = 0; r < p->nrows; r++) if (p->A[r + (size_t)p->nrows * j] != 0.0f) { eff[j] = 1.0f; break; }
    if (p->scale_frame < 0) { p->scale_frame = 0; p->scale_coef = 0.0f; }   // no q=0 sample: sum(p) = 0 -> Inf/NaN
    rc = finish_plan(p, verts, nverts, faces, nfaces, eff);
    if (rc != FIB_OK) { delete p; return rc; }
    *plan = p;
    return FIB_OK;
}

extern "C" void fib_odf_plan_destroy(fib_odf_plan *plan) {
    if (!plan) return;
    fib::DeviceGuard guard;
    (void)hipSetDevice(plan->device);
    delete plan;
}

extern "C" int fib_odf_plan_matrix(const fib_odf_plan *plan, float *A, int *nrows, int *nvol, int *nvert) {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan is NULL");
    if (nrows) *nrows = plan->nrows;
    if (nvol) *nvol = plan->nvol;
    if (nvert) *nvert = plan->nvert;
    if (A) memcpy(A, plan->A.data(), plan->A.size() * sizeof(float));
    return FIB_OK;
}

// ------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------
namespace {

template <int MB, int NX>
void launch_gemm(const GemmArgs &ga, unsigned grid, hipStream_t st) {
    hipLaunchKernelGGL((odf_gemm_kernel<MB, NX>), dim3(grid), dim3(256), 0, st, ga);
}

size_t peaks_smem(const fib_odf_plan *p) {
    return ((size_t)(p->rows_pad + 1) * PV + (size_t)PG * PV * PREC) * sizeof(float);
}

template <int DEG, bool EXACT>
int launch_peaks_t(const PeakArgs &pa, size_t smem, unsigned grid, hipStream_t st) {
    FIB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(odf_peaks_kernel<DEG, EXACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((odf_peaks_kernel<DEG, EXACT>), dim3(grid), dim3(PW * 64), smem, st, pa);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
}

int launch_peaks(const fib_odf_plan *plan, const float *odf, int64_t nvox, float *const peak[3], float *const qa[3],
                 int32_t *isort_top, int32_t *nvalid, bool reduce, hipStream_t st) {
    PeakArgs pa{};
    pa.odf = odf; pa.nbr = plan->nbr.p; pa.verts = plan->verts.p;
    for (int k = 0; k < 3; k++) { pa.peak[k] = peak ? peak[k] : nullptr; pa.qa[k] = qa ? qa[k] : nullptr; }
    pa.isort_top = isort_top; pa.nvalid = nvalid;
    pa.maxenc = reduce ? plan->maxenc.p : nullptr;
    pa.nvox = nvox; pa.nvert = plan->nvert; pa.rows_pad = plan->rows_pad;
    pa.vec_ok = (nvox % 4 == 0 && ((uintptr_t)odf & 15) == 0) ? 1 : 0;
    const size_t smem = peaks_smem(plan);
    FIB_CHECK(smem <= 160 * 1024, FIB_ERR_UNSUPPORTED, "ODF with %d vertices does not fit the peak finder's LDS tile", plan->nvert);
    const unsigned grid = (unsigned)fib::cdiv(nvox, PV);
    const bool exact = isort_top != nullptr;
    fib::ProfScope prof("odf_peaks", st);
    switch (plan->deg_pad) {
        case 6:  return exact ? launch_peaks_t<6, true>(pa, smem, grid, st) : launch_peaks_t<6, false>(pa, smem, grid, st);
        case 8:  return exact ? launch_peaks_t<8, true>(pa, smem, grid, st) : launch_peaks_t<8, false>(pa, smem, grid, st);
        default: return exact ? launch_peaks_t<16, true>(pa, smem, grid, st) : launch_peaks_t<16, false>(pa, smem, grid, st);
    }
}

}  // namespace

extern "C" int fibd_odf_rec(const fib_odf_plan *plan, const float *dwi, const uint8_t *mask, int64_t nvox,
                            float *pdf, float *odf, float *const peak[3], float *const qa[3],
                            float *odfmax_dev, int normalize, void *stream) {
    FIB_CHECK(plan && dwi && mask && odf && peak && qa, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nvox > 0, FIB_ERR_INVALID, "nvox must be positive");
    FIB_CHECK(nvox < ((int64_t)1 << 28), FIB_ERR_UNSUPPORTED, "volumes of 2^28 voxels or more are not supported (32-bit lane offsets)");
    FIB_CHECK(plan->nrow0 == 0 || pdf != nullptr, FIB_ERR_INVALID, "DSI plans need a pdf output volume");
    for (int k = 0; k < 3; k++) FIB_CHECK(peak[k] && qa[k], FIB_ERR_INVALID, "NULL peak/qa output volume");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    hipStream_t st = (hipStream_t)stream;
    GemmArgs ga{};
    ga.At = plan->At.p; ga.S = dwi; ga.mask = mask; ga.effbits = plan->effbits.p;
    ga.out0 = pdf; ga.out1 = odf; ga.nvox = nvox;
    ga.K = plan->nvol; ga.Kpad = plan->Kpad; ga.M = plan->nrows; ga.nrow0 = plan->nrow0; ga.ntile_m = plan->ntile_m;
    ga.scale_frame = plan->nrow0 > 0 ? plan->scale_frame : -1;
    ga.scale_coef = plan->scale_coef;
    const int64_t nblk = fib::cdiv(nvox, WG_VOX) * plan->ntile_m;
    FIB_CHECK(nblk < ((int64_t)1 << 31), FIB_ERR_UNSUPPORTED, "volume too large for one launch");
    const unsigned grid = (unsigned)nblk;
    { fib::ProfScope prof("odf_gemm", st);
    ga.has_ineff = plan->has_ineff ? 1 : 0;
#define FIB_GEMM_CASE(MBV, NXV) if (plan->MB == MBV && plan->NX == NXV) { launch_gemm<MBV, NXV>(ga, grid, st); launched = true; }
    bool launched = false;
    FIB_GEMM_CASE(6, 0) FIB_GEMM_CASE(7, 0) FIB_GEMM_CASE(8, 0) FIB_GEMM_CASE(9, 0) FIB_GEMM_CASE(10, 0) FIB_GEMM_CASE(11, 0)
    FIB_GEMM_CASE(6, 1) FIB_GEMM_CASE(7, 1) FIB_GEMM_CASE(8, 1) FIB_GEMM_CASE(9, 1) FIB_GEMM_CASE(10, 1)
    FIB_GEMM_CASE(6, 2) FIB_GEMM_CASE(7, 2) FIB_GEMM_CASE(8, 2) FIB_GEMM_CASE(9, 2) FIB_GEMM_CASE(10, 2)
    FIB_GEMM_CASE(6, 4) FIB_GEMM_CASE(7, 4) FIB_GEMM_CASE(8, 4) FIB_GEMM_CASE(9, 4) FIB_GEMM_CASE(10, 4)
#undef FIB_GEMM_CASE
    if (!launched) return fib::fail(FIB_ERR_INVALID, "internal: no GEMM variant for MB=%d NX=%d", plan->MB, plan->NX);
    }
    FIB_HIP(hipGetLastError());
    FIB_HIP(hipMemsetAsync(plan->maxenc.p, 0, 2 * sizeof(unsigned), st));
    int rc = launch_peaks(plan, odf, nvox, peak, qa, nullptr, nullptr, true, st);
    if (rc != FIB_OK) return rc;
    float *om = odfmax_dev ? odfmax_dev : plan->odfmax.p;
    hipLaunchKernelGGL(odfmax_finalize_kernel, dim3(1), dim3(1), 0, st, plan->maxenc.p, om);
    FIB_HIP(hipGetLastError());
    if (normalize) {
        fib::ProfScope prof("qa_normalize", st);
        hipLaunchKernelGGL(qa_normalize_kernel, dim3(2048), dim3(256), 0, st, qa[0], qa[1], qa[2], nvox, om, 0.0f);
        FIB_HIP(hipGetLastError());
    }
    return FIB_OK;
}

extern "C" int fibd_qa_normalize(float *const qa[3], int64_t nvox, float odfmax, void *stream) {
    FIB_CHECK(qa && qa[0] && qa[1] && qa[2] && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    hipLaunchKernelGGL(qa_normalize_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, qa[0], qa[1], qa[2], nvox,
                       (const float *)nullptr, odfmax);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
}

extern "C" int fibd_find_peaks(const fib_odf_plan *plan, const float *odf, int64_t nvox,
                               int32_t *isort_top, int32_t *nvalid, void *stream) {
    FIB_CHECK(plan && odf && isort_top && nvalid && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    return launch_peaks(plan, odf, nvox, nullptr, nullptr, isort_top, nvalid, false, (hipStream_t)stream);
}
