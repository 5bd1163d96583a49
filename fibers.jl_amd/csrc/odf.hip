// odf.hip — K2..K5: GQI / DSI ODF reconstruction, ODF peak finder, global QA normalisation (gfx950).
//
//   K2/K5  odf_gemm3_kernel  O[M x Nvox] = A[M x K] * clamp(S[K x Nvox]) as an exact f32 contraction on the bf16 matrix
//                            cores (three bf16 pieces per operand, v_mfma_f32_32x32x16_bf16; default), with the DSI
//                            antipodal fold fused into the sample load (FOLD)
//          odf_gemm_kernel   the same contraction on v_mfma_f32_32x32x2_f32 (FIBERS_ODF_GEMM=f32 and every plan the
//                            split kernel does not take)
//                            (gqi.jl:139-145 `mul!(o, A, s)`; dsi.jl:204-246 recast as two dense maps)
//          mask_compact_kernel, odf_post_kernel, odf_inf_fix_kernel: voxel-list compaction + outputs outside the mask (one launch),
//                            columns of voxels with a +Inf sample
//   K3     odf_peaks642_kernel (sphere_642, specialised scan + candidate lists), odf_peaks64_kernel (any tessellation),
//          odf_peaks_kernel (32-voxel tiles): find_peaks! + peak/qa extraction (gqi.jl:147-159,180-201; dsi.jl:244-258)
//   K4     max-of-means reduction + qa ./= odfmax (gqi.jl:164-168; dsi.jl:263-267)
//
// GEMM design (both kernels).  M (ODF vertices, 321 for sphere_642) is small, K (frames, 270) is small, N (voxels,
// 2.7 M) is huge and contiguous in memory for both S (planar frames) and O (planar vertices).  So the
// voxel index sits on the MFMA lane (N = column): a wave owns 32 voxels and ALL rows of its M tile;
// accumulators stay in registers for the whole K loop (MB blocks of 32x32 = 16*MB VGPRs), S is read
// exactly once straight into VGPRs as the B operand (two 128-B segments per wave load), and the only
// shared operand, the matrix A (347 KB: larger than LDS), is streamed K-tile by K-tile through a
// double-buffered LDS ring with direct-to-LDS loads (global_load_lds), laid out so that the A-fragment
// reads are bank-conflict free.  Both kernels are exact f32 (a k-ordered fma chain / exact piece products):
// the ODF matches a CPU sgemv to rounding, which the strict-inequality peak finder needs.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KT = 16;        // frames per LDS stage
constexpr int WG_VOX = 128;   // voxels per workgroup (4 waves x 32)

struct GemmArgs {
    const float *At;          // [ntile_m][Kpad][MW]  K-major tiles (MW = gemm_row_stride(MB, NX)), zero padded
    const void *At3;          // split-bf16 kernel: [ntile_m][Kpad/16][3 pieces][MB][64 lanes][8 bf16] (+ 1 KiB of f32 extra rows)
    const float *S;           // [K][nvox] planar DWI
    const int32_t *vidx;      // [nlive] voxels inside the mask, ascending (mask_compact_*): lane -> voxel gather / scatter
    const int32_t *nlive;     // device count of vidx
    const uint8_t *mask;      // [nvox]: voxels of a listed quad that are outside the mask get zeros
    const uint32_t *effbits;  // [Kpad/KT] bit j of word t: frame t*KT+j exists and takes part in the "any positive sample" test
    float *out0;              // rows [0, nrow0)        (DSI: pdf)
    float *out1;              // rows [nrow0, M)        (odf)
    int64_t nvox;             // voxels in this launch
    int64_t stride;           // frame / row stride of S, out0, out1 (voxels of the whole volume)
    int K, Kpad, M, nrow0, ntile_m;
    int scale_frame;          // DSI: frame whose clamped sample times scale_coef is sum(p); -1: no scaling
    float scale_coef;
    int has_ineff;
    const float *Aextra;          // split-bf16 kernel: f32 coefficients of the NX extra rows [ntile_m][Kpad/16][NX][16]
    int vec_ok;                   // split-bf16 kernel: output rows are 16-byte aligned (dwordx4 stores allowed)
    const int32_t *rowA, *rowB;   // optional output-row map for rows < nrow0: row r goes to frames rowA[r] and rowB[r] (>= 0)    // split-bf16 kernel, unscaled outputs (GQI): voxels holding a +Inf sample are listed and recomputed by odf_inf_fix_kernel
    // (Inf has no three-piece split: Inf - Inf = NaN; the reference's A*s gives +-Inf rows there)
    int32_t *fix_count, *fix_list;
    int fix_cap;
    int fold;                     // split-bf16 kernel, FOLD variant: S holds the raw frames; sample J of the contraction is max(S[rowA[J]],0) + max(S[rowB[J]],0)
    // FUSE variant (sphere_642, GQI): find_peaks! + peak / qa extraction run on the accumulators (gemm3_epilogue_fused)
    float *peak[3], *qa[3];       // outputs as PeakArgs
    const float *verts;           // [nvert][3]
    unsigned *maxenc;             // [4]: {exact max of means (ordered uint), NaN flag, lower bound of the max from the approximate means, -}
    float *mean_hi;               // [nvox] upper bound of each listed voxel's mean (NaN: the voxel is on the redo list)
    int32_t *redo_count, *redo_list;   // voxels the register scan could not finish (NaN / Inf columns, candidate-list overflow)
    int redo_cap;
    int anti;                     // fused kernel: anti-phase wave halves (see odf_gemm3_kernel)
    const void *At3b;             // odf_dsi2_kernel: image of the pdf tile (At3 / Aextra = the ODF tile in the fused scan's row order)
    int one_slot, one_stride;     // .. its work list: voxel groups one_slot + i * one_stride of the workgroup's XCD (set by the kernel)
    int dsi_na;                   // .. workgroups per XCD that take the ODF tile (the others take the pdf tile)
    unsigned *pair_flags;         // .. pairing (dsi_na = half the workgroups): [8 XCDs][32] item counters of the ODF-tile workgroups
    int pair_role;                // 0: none; 1: publish my item number; 2: wait (bounded) until my partner has reached my item
    int h2;                       // the images hold two fp16 pieces per element, scaled by the power of two sa (gemm3_body H2) ..
    float h2_inv_sa;              // .. and 1 / sa
    int phase_item, phase_wg;     // diagnostic build: the work item and the workgroup whose phases are stamped (FIB_PHASE; FIBERS_PHASE_ITEM / _WG)
};

// Diagnostic build only (make stamp -> libfibers_hip_stamp.so, -DFIB_CLOCK_STAMP; in the product library no stamp executes):
// the in-kernel clock of the contraction kernels = d(s_memtime) / d(s_memrealtime) x 100 MHz around a workgroup's whole work loop
// (MI355X_MICROARCH.md "DVFS give-back" item 6).  Stamps leave through a buffer of their own that no kernel reads.
#ifdef FIB_CLOCK_STAMP
__device__ unsigned long long fib_clock_stamps[2048][4];        // per workgroup of the last launch: {shader cycles, 100-MHz ticks, kernel id, work items}
#define FIB_STAMP_BEGIN() unsigned long long fcs_c0_, fcs_r0_; asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(fcs_c0_), "=s"(fcs_r0_) :: "memory")
#define FIB_STAMP_END(kid, items)                                                                                                   \
    do {                                                                                                                            \
        unsigned long long fcs_c1_, fcs_r1_;                                                                                        \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(fcs_c1_), "=s"(fcs_r1_) :: "memory");          \
        if (threadIdx.x == 0 && blockIdx.x < 2048) {                                                                                \
            fib_clock_stamps[blockIdx.x][0] = fcs_c1_ - fcs_c0_; fib_clock_stamps[blockIdx.x][1] = fcs_r1_ - fcs_r0_;               \
            fib_clock_stamps[blockIdx.x][2] = (kid); fib_clock_stamps[blockIdx.x][3] = (unsigned long long)(items);                 \
        }                                                                                                                           \
    } while (0)
// .. and the phases of ONE work item (GemmArgs::phase_item, FIBERS_PHASE_ITEM, default the third) of waves 0 and 4 of workgroup 8:
// s_memtime at the marks below, 256 per wave (tools/phase_profile.py; a mark costs ~250 cycles)
__device__ unsigned long long fib_phase_stamps[2][256];
#define FIB_PHASE_VARS() int fps_n_ = 0
#define FIB_PHASE(item_, wave_, id_)                                                                                                \
    do {                                                                                                                            \
        if ((int)blockIdx.x == a.phase_wg && (item_) == a.phase_item && ((wave_) & 3) == 0 && fps_n_ < 256) {                                     \
            unsigned long long t_;                                                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                                          \
            if ((threadIdx.x & 63) == 0) fib_phase_stamps[(wave_) >> 2][fps_n_] = (t_ << 8) | (unsigned)(id_);                      \
            fps_n_++;                                                                                                               \
        }                                                                                                                           \
    } while (0)
#else
#define FIB_STAMP_BEGIN() do { } while (0)
#define FIB_STAMP_END(kid, items) do { } while (0)
#define FIB_PHASE_VARS() do { } while (0)
#define FIB_PHASE(item_, wave_, id_) do { } while (0)
#endif

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

// s[s .< 0] .= 0 (gqi.jl:140) / max.(X, 0) (dsi.jl:209): a NaN sample stays NaN, -Inf becomes 0.  v_max_f32 would
// return 0 for NaN; gfx950's v_maximum3_f32 is the NaN-propagating IEEE-754-2019 maximum.
__device__ __forceinline__ float clamp_sample(float x) {
    float c;
    asm("v_maximum3_f32 %0, %1, 0, 0" : "=v"(c) : "v"(x));
    return c;
}

// NaN-propagating maximum of three (IEEE-754-2019 maximum): the running maximum of a voxel's CLAMPED samples is > 0 iff some
// sample is positive (gqi.jl:142, dsi.jl:207), NaN iff some sample is NaN, +Inf iff some sample is +Inf -- one instruction per
// sample pair instead of a NaN-ignoring maximum plus two NaN trackers
__device__ __forceinline__ float max3_nan(float a, float b, float c) {
    float r;
    asm("v_maximum3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// sortperm!(odf_peak, rev=true) (gqi.jl:198) orders by descending value with Base.isless semantics (NaN above
// everything, +0.0 above -0.0) and keeps ascending index among equals.  Both are captured by one 64-bit key:
// high word = order-preserving uint image of the float (NaN canonicalised to the top), low word = ~index.
// A larger key sorts earlier; key 0 = empty slot.  Keeping the best three is then a branch-free 3-element
// insertion (3 compares + selects) instead of a comparison-function call per candidate.
__device__ __forceinline__ unsigned long long peak_key(float x, int idx) {
    const unsigned b = __float_as_uint(x);
    unsigned hi = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    if (x != x) hi = 0xffffffffu;
    return ((unsigned long long)hi << 32) | (unsigned)(~idx);
}
struct Top3 { unsigned long long k[3]; };
__device__ __forceinline__ void top3_clear(Top3 &t) { t.k[0] = t.k[1] = t.k[2] = 0ull; }
__device__ __forceinline__ void top3_insert_key(Top3 &t, unsigned long long k) {
    const bool g0 = k > t.k[0], g1 = k > t.k[1], g2 = k > t.k[2];
    t.k[2] = g1 ? t.k[1] : (g2 ? k : t.k[2]);
    t.k[1] = g0 ? t.k[0] : (g1 ? k : t.k[1]);
    t.k[0] = g0 ? k : t.k[0];
}
__device__ __forceinline__ void top3_insert(Top3 &t, float x, int idx) { top3_insert_key(t, peak_key(x, idx)); }
__device__ __forceinline__ int top3_index(const Top3 &t, int k) { return t.k[k] ? (int)~(unsigned)t.k[k] : -1; }

__device__ __forceinline__ unsigned enc_ordered(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(unsigned e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

// maximum(mean(odf, dims=4)) (gqi.jl:164) in two steps.  The reference's per-voxel sum runs sequentially over the vertices
// (Base mapreducedim! over dim 4) and is divided by n; a kernel that sums in another order can only bound that mean:
// |its sum - the sequential sum| <= 2 (n-1) 2^-24 sum|o|, and |o| <= o - 2 min(vmin, 0).  Every peak kernel therefore
// records mean_hi[vox] >= the voxel's mean and raises maxenc[2] to a lower bound of the maximum; odfmax_refine_kernel then
// recomputes, with the sequential sum, the few voxels whose upper bound reaches it.  NaN means set the NaN flag (maximum()
// propagates NaN), infinite means are order-independent and go straight to the exact maximum maxenc[0].
// Called by all 64 lanes of a wave; `active` lanes contribute voxel `vox`.
__device__ __forceinline__ void odfmax_contribute(unsigned *maxenc, float *mean_hi, int64_t vox, bool active, float mean, float vmin, int nvert) {
    const bool isnan_ = mean != mean, isinf_ = fabsf(mean) == INFINITY;
    const float eps = (2.1f * 5.9604645e-8f) * (float)nvert * (fabsf(mean) + 2.0f * fabsf(fminf(vmin, 0.0f)));
    if (active && mean_hi) mean_hi[vox] = (isnan_ || isinf_) ? __builtin_nanf("") : mean + eps;
    unsigned e = active && !isnan_ && !isinf_ ? enc_ordered(mean - eps) : 0u;
    unsigned ex = active && isinf_ ? enc_ordered(mean) : 0u;
    const unsigned long long nanb = __ballot(active && isnan_);
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned oth = (unsigned)__shfl_xor((int)e, off), othx = (unsigned)__shfl_xor((int)ex, off);
        e = oth > e ? oth : e;
        ex = othx > ex ? othx : ex;
    }
    if ((threadIdx.x & 63) == 0) {
        if (e) atomicMax(&maxenc[2], e);
        if (ex) atomicMax(&maxenc[0], ex);
        if (nanb) atomicOr(&maxenc[1], 1u);
    }
}

__device__ __forceinline__ float peak_key_value(unsigned hi) {    // inverse of peak_key's float image (NaN canonical)
    return hi == 0xffffffffu ? __builtin_nanf("") : __uint_as_float((hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi);
}


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
// LDS row stride of a stage: a multiple of 64 floats (256 B) so that every fragment offset from one of two
// base registers (even / odd 32-row block) is a multiple of 256 B and fits ds_read2st64_b32's 8-bit offsets:
// no per-stage v_add for LDS addresses (they would cost MFMA time, see above).
__host__ __device__ constexpr int gemm_row_stride(int mb, int nx) { return (mb * 32 + (nx > 0 ? 16 : 0) + 63) / 64 * 64; }

// Epilogue shared by the two GEMM kernels (the C/D layout of the 32x32 MFMAs does not depend on the input type):
// "any sample > 0" (gqi.jl:142, dsi.jl:207), NaN/Inf poisoning, the DSI 1/sum(p) scale, row -> output mapping.
template <int MB, int NX>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &a, f32x16 (&acc)[MB], float (&xacc)[NX > 0 ? NX : 1], float vmax, float vnf,
                                              bool inb, bool lv, int64_t vox, int kh, int tile_m, uint32_t c_off, float sraw) {
    constexpr int ROWS = MB * 32 + NX;
    // ---- epilogue: the two k-halves of a voxel live in lanes l and l^32 -----------------------------------
    float pm = fmaxf(vmax, __shfl_xor(vmax, 32));
    float pn = vnf + __shfl_xor(vnf, 32);
    const bool nonfinite = pn != pn;                    // the voxel holds a NaN or +Inf sample (after the clamp)
    const bool valid = lv && (pm > 0.0f || nonfinite);
    const bool do_scale = a.scale_frame >= 0;
    float scale = 1.0f;
    if (do_scale) {
        const float s = sraw < 0.0f ? 0.0f : sraw;
        scale = 1.0f / (a.scale_coef * s);              // p ./ sum(p), dsi.jl:225 (0 -> Inf/NaN like the reference)
    }
    // DSI: the reference's FFT smears a NaN / +Inf sample over the whole voxel and p ./ sum(p) makes it NaN everywhere.
    // GQI: o = A*s propagates on its own (NaN * a = NaN, Inf * a = +-Inf, Inf * 0 = NaN) exactly as the reference's mul!.
    if (nonfinite && do_scale) scale = __builtin_nanf("");
    const bool plain = __all(valid && !nonfinite) && !do_scale;   // wave-uniform: store the accumulators as they are
    const float mulv = valid ? scale : 0.0f;
#pragma unroll
    for (int x = 0; x < NX; x++) xacc[x] += __shfl_xor(xacc[x], 32);
    if (!inb) return;
    const uint32_t o_off = (uint32_t)((vox + (int64_t)4 * kh * a.stride) * 4);   // < 2^32: nvox <= 2^27
    auto row_ptr = [&](int row) -> char * {             // wave-uniform row base
        return reinterpret_cast<char *>(row >= a.nrow0 ? a.out1 + (int64_t)(row - a.nrow0) * a.stride
                                                       : a.out0 + (int64_t)row * a.stride);
    };
    const bool mapped = a.rowA != nullptr;
#pragma unroll
    for (int m = 0; m < MB; m++) {
        const int row0 = tile_m * ROWS + m * 32;        // wave-uniform
        if (row0 >= a.M) break;
        const bool whole = row0 + 32 <= a.M && (row0 >= a.nrow0 || (row0 + 32 <= a.nrow0 && !mapped));   // uniform fast path
        char *base = row_ptr(row0);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int dr = (r & 3) + 8 * (r >> 2);      // row within the block, before the lane-half offset
            float v = acc[m][r];
            if (!plain) v = valid ? v * mulv : 0.0f;
            if (whole) {
                *reinterpret_cast<float *>(base + (int64_t)dr * a.stride * 4 + o_off) = v;
            } else {
                const int row = row0 + dr + 4 * kh;
                if (row >= a.M) continue;
                if (mapped && row < a.nrow0) {           // symmetric DSI: p(r) = p(-r), one computed row feeds two frames
                    const int fa = a.rowA[row], fb = a.rowB[row];
                    a.out0[(int64_t)fa * a.stride + vox] = v;
                    if (fb >= 0) a.out0[(int64_t)fb * a.stride + vox] = v;
                } else {
                    *reinterpret_cast<float *>(row_ptr(row) + c_off) = v;
                }
            }
        }
    }
#pragma unroll
    for (int x = 0; x < NX; x++) {
        const int row = tile_m * ROWS + MB * 32 + x;    // wave-uniform
        if (row >= a.M) break;
        float v = xacc[x];
        if (!plain) v = valid ? v * mulv : 0.0f;
        if (kh == 0) {
            if (mapped && row < a.nrow0) {
                const int fa = a.rowA[row], fb = a.rowB[row];
                a.out0[(int64_t)fa * a.stride + vox] = v;
                if (fb >= 0) a.out0[(int64_t)fb * a.stride + vox] = v;
            } else {
                *reinterpret_cast<float *>(row_ptr(row) + c_off) = v;
            }
        }
    }
}

template <int MB, int NX>
__global__ __launch_bounds__(256, 2) void odf_gemm_kernel(const GemmArgs a) {
    constexpr int MW = gemm_row_stride(MB, NX);         // LDS row stride (floats)
    constexpr int TILE = KT * MW;                       // floats per stage
    constexpr int NPIECE = TILE * 4 / 1024;
    static_assert((TILE * 4) % 1024 == 0, "stage must be a whole number of 1-KiB pieces");
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kh = lane >> 5;
    const int tile_m = blockIdx.x % a.ntile_m;
    const int64_t tile_n = blockIdx.x / a.ntile_m;
    // The workgroup's 128 columns are 128 consecutive entries of the compacted voxel list: voxels outside the mask
    // cost nothing (brain masks cover about a third of a volume), and a lane's sample loads / output stores were
    // per-lane addresses anyway.  Workgroups past the end of the list leave at once.
    // The list entry is fetched before the count is known (the buffer holds nvox entries; those past the count are
    // stale and replaced by voxel 0) and the first stage of A is already on its way: one memory latency, not three.
    const int64_t slot = tile_n * WG_VOX + wave * 32 + col;
    const int32_t vraw = a.vidx[slot < a.nvox ? slot : a.nvox - 1];
    const char *Abase = reinterpret_cast<const char *>(a.At + (size_t)tile_m * a.Kpad * MW);
    const uint32_t a_off = (uint32_t)lane * 16;
    auto stage_A = [&](int t, int buf) {                // one stage = TILE*4 contiguous bytes of At
        const char *g = Abase + (size_t)t * TILE * 4;
        char *l = reinterpret_cast<char *>(lds + buf * TILE);
        for (int p = wave; p < NPIECE; p += 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + p * 1024 + a_off),
                                             (__attribute__((address_space(3))) void *)(l + p * 1024), 16, 0, 0);
    };
    // B operand: KT/2 unconditional loads per stage (frame index clamped to K-1: the padded rows of At are zero)
    float braw[KT / 2];
    stage_A(0, 0);
    const int nlive = a.nlive[0];
    if (tile_n * WG_VOX >= nlive) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the direct-to-LDS loads must not outlive the workgroup
        return;
    }
    const bool inb = slot < nlive;
    const int64_t vox = inb ? vraw : 0;
    const int ntiles = a.Kpad / KT;
    // per-lane 32-bit byte offsets; everything else in an address is wave-uniform (SGPR base)
    const uint32_t c_off = (uint32_t)(vox * 4);
    const uint32_t s_off = (uint32_t)((vox + (int64_t)kh * a.stride) * 4);   // frame kh of the pair, this voxel
    const char *Sbase = reinterpret_cast<const char *>(a.S);
    const int64_t frame_pair_bytes = 2 * a.stride * 4;

    // Buffer loads: SGPR resource (base = the frame pair's two rows, advanced with scalar adds) + one 32-bit lane
    // offset; flat global loads made hipcc build eight 64-bit per-lane addresses per stage with v_mad_u64_u32.
    // The resource's range check also replaces the clamping of the last stage: a pair past the last frame gets
    // num_records = 0 and a single-frame pair (odd K) one row, so those lanes read 0.0 without a memory access
    // (the padded rows of At are zero, and 0 changes neither the running max nor the non-finite tracker).
    const int32_t row_bytes = (int32_t)(a.stride * 4);
    auto load_B = [&](int t) {
        const char *fb = Sbase + (int64_t)t * (KT / 2) * frame_pair_bytes;               // wave-uniform
        int rem = a.K - t * KT;                                                           // frames left from this stage on
#pragma unroll
        for (int kk = 0; kk < KT / 2; kk++) {
            const int32_t nrec = rem >= 2 ? 2 * row_bytes : (rem == 1 ? row_bytes : 0);
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(rem > 0 ? fb : Sbase), 0, nrec, 0x00020000);
            braw[kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)s_off, 0, 0));
            fb += frame_pair_bytes;
            rem -= 2;
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
    float vnf = 0.0f;                                   // becomes NaN once a clamped sample is NaN or +Inf

    load_B(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): a builtin, so that hipcc's own wait-count bookkeeping sees it
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
                bcur[kk] = clamp_sample(s);
                vmax = fmaxf(vmax, ((eff >> (2 * kk)) & 1u) ? s : 0.0f);
                vnf = __builtin_fmaf(bcur[kk], 0.0f, vnf);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KT / 2; kk++) {
                const float s = braw[kk];
                // raw v_max_f32: fmaxf() would add a canonicalising v_max(s,s) per sample
                bcur[kk] = clamp_sample(s);
                asm("v_max_f32 %0, %1, %2" : "=v"(vmax) : "v"(vmax), "v"(s));
                vnf = __builtin_fmaf(bcur[kk], 0.0f, vnf);
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
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): next stage's direct-to-LDS loads and samples have landed
        __syncthreads();
    }

    float sraw = 0.0f;
    if (a.scale_frame >= 0) sraw = *reinterpret_cast<const float *>(Sbase + (int64_t)a.scale_frame * a.stride * 4 + c_off);
    const bool lv = inb && a.mask[vox] != 0;            // the listed quad's voxels outside the mask: zeros
    gemm_epilogue<MB, NX>(a, acc, xacc, vmax, vnf, inb, lv, vox, kh, tile_m, c_off, sraw);
}

// Epilogue of the split-bf16 kernel.  Fast path (16-byte aligned output rows): the wave's 32 voxels are 8 aligned quads
// of consecutive voxels -> each half of a 32x32 block goes through the wave's 2-KiB LDS tile ([16 rows][32 voxels]; rows
// r and r+4 interleaved so that both lane halves write different banks) and leaves as 2 dwordx4 stores of 8 rows x 128 B.
// PRE (odf_gemm16_kernel): the caller has already summed the extra rows over the lanes of a voxel and applied the DSI scale;
// ROWS = rows of an M tile (the 16x16x32 kernel's tiles are a whole number of 16-row blocks)
// ASC: the accumulators (not the extra rows) carry the voxel's power-of-two factor 1 / ascale (gemm3_body H2)
template <int MB, int NX, bool PRE = false, int ROWS_ = 0, bool MAPLDS = false, bool ASC = false>
__device__ __forceinline__ void gemm3_epilogue(const GemmArgs &a, f32x16 (&acc)[MB], float (&xacc)[NX > 0 ? NX : 1], float vmax, float vnf,
                                               bool inb, bool lv, int64_t vox, int lane, int tile_m, float sraw, char *tr,
                                               const int32_t *mapA = nullptr, const int32_t *mapB = nullptr, float ascale = 1.0f) {
    // MAPLDS: mapA / mapB are LDS copies of a.rowA / a.rowB (a global load per stored row would sit between the transposition and
    // its stores).  Two code paths, not one pointer chosen at run time: a generic pointer would make every lookup a flat load, and
    // a flat load waits for all the row stores before it (vmcnt).
    auto rowA_at = [&](int row) -> int { if constexpr (MAPLDS) return mapA[row]; else return a.rowA[row]; };
    auto rowB_at = [&](int row) -> int { if constexpr (MAPLDS) return mapB[row]; else return a.rowB[row]; };
    constexpr int ROWS = ROWS_ > 0 ? ROWS_ : MB * 32 + NX;
    constexpr int XROW0 = ROWS - NX;                    // first extra row of a tile
    const int col = lane & 31, kh = lane >> 5;
    float pm = fmaxf(vmax, __shfl_xor(vmax, 32));
    float pn = vnf + __shfl_xor(vnf, 32);
    const bool nonfinite = pn != pn;                    // the voxel holds a NaN or +Inf sample (after the clamp)
    const bool valid = lv && (pm > 0.0f || nonfinite);
    const bool do_scale = !PRE && a.scale_frame >= 0;
    float scale = 1.0f;
    if (do_scale) {
        const float s = sraw < 0.0f ? 0.0f : sraw;
        scale = 1.0f / (a.scale_coef * s);              // p ./ sum(p), dsi.jl:225 (0 -> Inf/NaN like the reference)
    }
    // DSI: the reference's FFT smears a NaN / +Inf sample over the whole voxel and p ./ sum(p) makes it NaN everywhere.
    // GQI: a NaN sample gives NaN pieces and a NaN column on its own; a +Inf sample too (Inf - Inf = NaN in the split), but
    // the reference's A*s has +-Inf rows there: the voxel is listed and odf_inf_fix_kernel recomputes its column.
    if (nonfinite && do_scale) scale = __builtin_nanf("");
    if (a.scale_frame < 0 && a.fix_list != nullptr && tile_m == 0 && kh == 0 && lv && pm == INFINITY) {
        const int slot = atomicAdd(a.fix_count, 1);
        if (slot < a.fix_cap) a.fix_list[slot] = (int32_t)vox;
    }
    const bool plain = __all(valid && !nonfinite) && !do_scale && !ASC;   // wave-uniform: store the accumulators as they are
    const float mulv = valid ? scale : 0.0f;
    const float mula = ASC ? (valid ? scale * ascale : 0.0f) : mulv;       // (scale * ascale: exact, ascale is a power of two)
    if (!PRE) {
#pragma unroll
        for (int x = 0; x < NX; x++) xacc[x] += __shfl_xor(xacc[x], 32);
    }
    // the voxel list is made of aligned quads (mask_write_kernel): lanes 4q..4q+3 hold four consecutive voxels, so the
    // lane that stores quad q of a row (lane & 7 == q after the transposition) takes its address from lane 4q
    const bool contig = a.vec_ok != 0;
    const int qsrc = 4 * (lane & 7);
    const int32_t qvox = __shfl((int)vox, qsrc);
    const bool qinb = __shfl((int)inb, qsrc) != 0;
    auto row_ptr = [&](int row) -> char * {
        return reinterpret_cast<char *>(row >= a.nrow0 ? a.out1 + (int64_t)(row - a.nrow0) * a.stride
                                                       : a.out0 + (int64_t)row * a.stride);
    };
    const bool mapped = a.rowA != nullptr;
    if (contig) {
        // LDS tile of half a block (16 rows): logical row r lives in physical row (r & 8) | ((r & 3) << 1) | ((r >> 2) & 1)
        float *tw = reinterpret_cast<float *>(tr) + kh * 32 + col;                // + physical row of (dr + 4 kh)
        const int P = lane >> 3;                                                  // physical row (within 8) this lane stores
        const int lrow = ((P >> 1) & 3) | ((P & 1) << 2);                         // its logical row within the 8
        const float4 *trd = reinterpret_cast<const float4 *>(tr) + lane;
        const uint32_t voff = (uint32_t)qvox * 4u;
#pragma unroll
        for (int m = 0; m < MB; m++) {
            const int row0 = tile_m * ROWS + m * 32;    // wave-uniform
            if (row0 >= a.M) break;
            const bool whole = row0 + 32 <= a.M && m * 32 + 32 <= XROW0 && !(mapped && row0 < a.nrow0);   // uniform: no row of the block needs a test
            char *base = row_ptr(row0) + (int64_t)lrow * a.stride * 4 + voff;     // rows of one output volume are equidistant
            const bool split_out = row0 < a.nrow0 && row0 + 32 > a.nrow0;         // block straddles pdf | odf
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
#pragma unroll
                for (int r = 8 * hb; r < 8 * hb + 8; r++) {
                    float v = acc[m][r];
                    if (!plain) v = valid ? v * mula : 0.0f;
                    tw[(8 * ((r >> 2) & 1) + 2 * (r & 3)) * 32] = v;   // logical row (r&3) + 8(r>>2) + 4kh
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float4 v4 = trd[j * 64];
                    if (!qinb) continue;                // ragged end of the voxel list
                    if (whole && !split_out) {
                        *reinterpret_cast<float4 *>(base + (int64_t)(16 * hb + 8 * j) * a.stride * 4) = v4;
                        continue;
                    }
                    const int row = row0 + 16 * hb + 8 * j + lrow;
                    if (row >= a.M || m * 32 + 16 * hb + 8 * j + lrow >= XROW0) continue;   // (a tile of the 16x16x32 kernel may end inside a 32-row block)
                    if (mapped && row < a.nrow0) {       // symmetric DSI: p(r) = p(-r), one computed row feeds two frames
                        const int fa = rowA_at(row), fb = rowB_at(row);
                        *reinterpret_cast<float4 *>(reinterpret_cast<char *>(a.out0 + (int64_t)fa * a.stride) + voff) = v4;
                        if (fb >= 0) *reinterpret_cast<float4 *>(reinterpret_cast<char *>(a.out0 + (int64_t)fb * a.stride) + voff) = v4;
                    } else {
                        *reinterpret_cast<float4 *>(row_ptr(row) + voff) = v4;
                    }
                }
            }
        }
    } else if (inb) {
        const uint32_t c_off = (uint32_t)(vox * 4);
#pragma unroll
        for (int m = 0; m < MB; m++) {
            const int row0 = tile_m * ROWS + m * 32;
            if (row0 >= a.M) break;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v = acc[m][r];
                if (!plain) v = valid ? v * mula : 0.0f;
                if (row >= a.M || m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh >= XROW0) continue;
                if (mapped && row < a.nrow0) {
                    const int fa = rowA_at(row), fb = rowB_at(row);
                    a.out0[(int64_t)fa * a.stride + vox] = v;
                    if (fb >= 0) a.out0[(int64_t)fb * a.stride + vox] = v;
                } else {
                    *reinterpret_cast<float *>(row_ptr(row) + c_off) = v;
                }
            }
        }
    }
    if (inb && kh == 0) {
#pragma unroll
        for (int x = 0; x < NX; x++) {
            const int row = tile_m * ROWS + XROW0 + x;    // wave-uniform
            if (row >= a.M) break;
            float v = xacc[x];
            if (!plain) v = valid ? v * mulv : 0.0f;
            if (mapped && row < a.nrow0) {
                const int fa = rowA_at(row), fb = rowB_at(row);
                a.out0[(int64_t)fa * a.stride + vox] = v;
                if (fb >= 0) a.out0[(int64_t)fb * a.stride + vox] = v;
            } else {
                *reinterpret_cast<float *>(row_ptr(row) + (uint32_t)(vox * 4)) = v;
            }
        }
    }
}

// ---- fused epilogue (sphere_642, GQI): ODF rows out + find_peaks! on the accumulators -------------------------------
// The reference finds the peaks on the thread-local ODF right after mul! (gqi.jl:144-159); a separate peak kernel re-reads
// 3.5 GB of ODF.  Here the wave that holds 32 voxels x 321 rows in registers tests every vertex against its <= 6 folded-face
// neighbours in registers (layout and program: sphere642_fused.inc / tools/gen_s642_fused.py), appends the few candidates
// (amplitude, slot) to a per-lane list in LDS, picks the top three by the sortperm key, merges the two lane halves and
// writes peak / qa.  Not handled here, listed for odf_redo_kernel instead: voxels whose column holds a NaN / Inf and
// voxels with more candidates in a lane half than the list holds.  The per-voxel mean is only bounded here (order of the
// f32 sum differs from the reference's): mean_hi[vox] >= mean(odf[vox,:]) and maxenc[2] <= max of the means;
// odfmax_refine_kernel recomputes the few voxels in between with the reference's sequential sum.
#include "sphere642_fused.inc"
__device__ const short fib_f642_pos_vertex_dev[321] = {
#define FIB_F642_COPY(...) __VA_ARGS__
    FIB_F642_POS_LIST(FIB_F642_COPY)
};
__device__ const short fib_f642_slot_vertex_dev[2 * 161] = {
    FIB_F642_SLOT_LIST(FIB_F642_COPY)
#undef FIB_F642_COPY
};
constexpr int FQ_CAP = 10;                       // candidates per lane half that the list holds (a longer list: odf_redo_kernel)
constexpr int FQ_LIST = FQ_CAP * 512;            // bytes per wave: [FQ_CAP][2][64] dwords
constexpr int FQ_CAPB = 9;                       // .. the DSI pair kernel's ODF tile: slots as bytes, [FQ_CAPB][64] dwords + [FQ_CAPB][64] bytes
constexpr int FQ_LISTB = FQ_CAPB * 320;
constexpr int FQ_NPOS = 320, FQ_NSLOT = 2 * 161, FQ_NV = 321;
constexpr int FQ_TABB = (2 * FQ_NPOS + FQ_NSLOT + 3 * FQ_NV) * 4 + 12;   // byte offset of each position's output row, vertex-of-slot, vertex coordinates (16-byte multiple)
static_assert(FQ_TABB % 16 == 0, "table block keeps the LDS carve-up 16-byte aligned");

// (plain fmaxf / fminf chains: hipcc folds them into v_max3_f32 / v_min3_f32 and knows MFMA results are canonical; inline asm
// would cost an s_nop per statement)
// the value the other lane half holds / the maximum over both halves, by v_permlane32_swap (VALU) instead of ds_bpermute (an LDS round trip)
__device__ __forceinline__ float fq_xhalf(float x, int kh) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);   // r[0]: the lower half's value in every lane, r[1]: the upper half's
    return __uint_as_float(kh ? r[0] : r[1]);
}
__device__ __forceinline__ float fq_xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float fq_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float fq_max3z(float a, float b) { return __builtin_fmaxf(__builtin_fmaxf(a, b), 0.0f); }
__device__ __forceinline__ float fq_min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float fq_max2(float a, float b) { return __builtin_fmaxf(a, b); }

// POW2: `scale` is a finite power of two for every voxel (H2 without the DSI factor): one legacy multiplication per value does the
// scaling and the zeroing of voxels that are skipped or outside the mask
// DRAIN: wait for the caller's requests in flight (the next item's first pieces and samples, issued a scan ago) right before the first
// row store goes out -- after the stores no wait can tell those requests from the stores (gemm3_body SLDS)
// TRN: 2-KiB transposition tiles of the wave (2: half block h + 1 is put down while h is read back; 1: one after the other);
// LB: the candidate lists keep the slot in a byte ([FQ_CAP][64] amplitudes, then [FQ_CAP][64] slot bytes: FQ_LISTB per wave)
// CAP: entries per list -- where LDS is short (the DSI pair kernel's ODF tile, whose samples travel through LDS as well)
template <int NW, bool PRE = false, bool SCALE = false, bool POW2 = false, bool DRAIN = false, int TRN = 2, bool LB = false, int CAP = FQ_CAP>
__device__ __forceinline__ void gemm3_epilogue_fused(const GemmArgs &a, f32x16 (&acc)[10], float xrow, float vmax, float vnf, bool inb, bool lv,
                                                     int64_t vox, int lane, char *tr, char *lst, const uint64_t *posoff, const int *slotv, const float *vl,
                                                     unsigned &en_run, float scale = 1.0f, float xscale = 1.0f) {
    const int col = lane & 31, kh = lane >> 5;
    const float pm = fmaxf(vmax, __shfl_xor(vmax, 32));
    const float pn = vnf + __shfl_xor(vnf, 32);
    const bool nonfinite = pn != pn;                    // the voxel holds a NaN or +Inf sample (after the clamp)
    const bool valid = lv && (pm > 0.0f || nonfinite);  // gqi.jl:142
    // (a +Inf sample where the repair exists, GQI: the voxel goes to the redo list with bit 31 set and odf_post_kernel recomputes its column)
    const bool refix = a.fix_list != nullptr && lv && pm == INFINITY;
    if (!PRE) xrow += __shfl_xor(xrow, 32);
    if constexpr (SCALE) {                              // DSI: p ./ sum(p) (dsi.jl:225) before the radial sums are looked at
#pragma unroll
        for (int m = 0; m < 10; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if constexpr (POW2) asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(acc[m][r]) : "v"(acc[m][r]), "v"(valid ? scale : 0.0f));   // (x * 0 = 0 for every x, NaN and Inf included)
                else acc[m][r] = valid ? acc[m][r] * scale : 0.0f;
            }
        xrow = valid ? xrow * xscale : 0.0f;             // (H2: the f32 extra row carries no power-of-two factor)
    } else if (!__all(valid && !nonfinite)) {           // wave-uniform, rare: skipped voxels and voxels outside the mask read 0
#pragma unroll
        for (int m = 0; m < 10; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][r] = valid ? acc[m][r] : 0.0f;
        xrow = valid ? xrow : 0.0f;
    }
    // ---- pass 1, branch-free: candidate flag of every slot = vertex above all its neighbours and above 0 (gqi.jl:185-196, 200),
    // shifted into five 32-bit strings per lane (slot 32w + i -> bit 31 - i of word w) --------------------------------------
    unsigned cw0 = 0, cw1 = 0, cw2 = 0, cw3 = 0, cw4 = 0;
    bool cpole = false;
    {
#define O(m, r) acc[m][r]
#define F(i) ff##i
#define X(j) fx##j
#define Z 0.0f
#define FQ_FLAG(sid, t_, x_) asm("v_cmp_nge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"((sid) < 32 ? cw0 : (sid) < 64 ? cw1 : (sid) < 96 ? cw2 : (sid) < 128 ? cw3 : cw4) : "v"(t_), "v"(x_) : "vcc");
#define FQ_FDEF(i, m, r) const float ff##i = fq_xhalf(acc[m][r], kh);
#define FQ_SLOT(sid, m, r, n0, n1, n2, n3, n4, n5) { const float t_ = fq_max3z(fq_max3(fq_max3(n0, n1, n2), n3, n4), n5); FQ_FLAG(sid, t_, acc[m][r]) }
#define FQ_XDEF(ja, jb, m, r) const float fo##ja = fq_xhalf(acc[m][r], kh); const float fx##ja = kh ? fo##ja : acc[m][r], fx##jb = kh ? acc[m][r] : fo##ja;
#define FQ_FMAX(o0, o1, o2, x0, x1, x2, out) float out; { float p_ = fq_max3(o0, o1, o2); p_ = fq_xhalf_max(p_); out = fq_max3z(fq_max3(p_, x0, x1), x2); }
#define FQ_FTEST2(sid, ja, oa0, oa1, oa2, xa0, xa1, xa2, jb, ob0, ob1, ob2, xb0, xb1, xb2) { FQ_FMAX(oa0, oa1, oa2, xa0, xa1, xa2, ta_) FQ_FMAX(ob0, ob1, ob2, xb0, xb1, xb2, tb_) \
        const float t_ = kh ? tb_ : ta_, x_ = kh ? fx##jb : fx##ja; FQ_FLAG(sid, t_, x_) }
#define FQ_FPOLE(j, o0, o1, o2, x0, x1, x2) { FQ_FMAX(o0, o1, o2, x0, x1, x2, t_) cpole = kh == 0 && !(t_ >= fx##j); }
        const float fx16 = xrow;
        FIB_F642_XDEFS(FQ_XDEF)
        FIB_F642_PAIRS(FQ_FDEF, FQ_SLOT)
        FIB_F642_FTESTS(FQ_FTEST2, FQ_FPOLE)
#undef O
#undef F
#undef X
#undef Z
#undef FQ_FLAG
#undef FQ_FDEF
#undef FQ_SLOT
#undef FQ_XDEF
#undef FQ_FMAX
#undef FQ_FTEST2
#undef FQ_FPOLE
    }
    // ---- ODF rows + pass 2.  Each half block (8 registers of both lane halves = 16 rows x 32 voxels) goes through one of the
    // wave's two LDS tiles and leaves as 2 stores of 8 rows x 128 B (row = vertex of the position).  While a half block is
    // in the tile, a lane that flagged one of its 8 slots reads the amplitude back by its dynamic index (registers cannot
    // be indexed per lane) and appends (amplitude, slot) to its candidate list. ----------------------------------------------
    uint32_t *lw = reinterpret_cast<uint32_t *>(lst) + lane;      // entry k of this lane: amplitude at lw[k*128], slot at lw[k*128 + 64]
    uint8_t *lb = reinterpret_cast<uint8_t *>(lst) + CAP * 256 + lane;   // (LB: amplitude at lw[k*64], slot at lb[k*64])
    auto list_put = [&](int k, float x, uint32_t slot) {
        if constexpr (LB) { lw[k * 64] = __float_as_uint(x); lb[k * 64] = (uint8_t)slot; }
        else { lw[k * 128] = __float_as_uint(x); lw[k * 128 + 64] = slot; }
    };
    int cnt = 0;
    {
        const int qsrc = 4 * (lane & 7);
        const int32_t qvox = __shfl((int)vox, qsrc);
        const bool qinb = __shfl((int)inb, qsrc) != 0;
        const int P = lane >> 3;
        const int lrow = ((P >> 1) & 3) | ((P & 1) << 2);
        char *obase = reinterpret_cast<char *>(a.out1) + (uint32_t)qvox * 4u;
        auto rows_out = [&](auto guard) {
            // software pipeline over the 20 half blocks: half block h + 1 goes into the other tile before tile h is read back
            auto put = [&](int h) {
                const int m = h >> 1, hb = h & 1;
                float *tw = reinterpret_cast<float *>(tr + (TRN == 2 ? hb * 2048 : 0)) + kh * 32 + col;
#pragma unroll
                for (int r = 8 * hb; r < 8 * hb + 8; r++) tw[(8 * ((r >> 2) & 1) + 2 * (r & 3)) * 32] = acc[m][r];
            };
            if (TRN == 2) put(0);
#pragma unroll
            for (int h = 0; h < 20; h++) {
                const int m = h >> 1, hb = h & 1;
                if (TRN == 2) { if (h + 1 < 20) put(h + 1); }
                else put(h);                            // (one tile: behind the reads of half block h - 1 -- a wave's LDS operations execute in order)
                const float *tw = reinterpret_cast<const float *>(tr + (TRN == 2 ? hb * 2048 : 0)) + kh * 32 + col;
                const float4 *trd = reinterpret_cast<const float4 *>(tr + (TRN == 2 ? hb * 2048 : 0)) + lane;
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float4 v4 = trd[j * 64];
                    typedef float nt4_t __attribute__((ext_vector_type(4)));
                    const nt4_t v4n = {v4.x, v4.y, v4.z, v4.w};
                    char *dst = obase + posoff[m * 32 + 16 * hb + 8 * j + lrow];
                    if (!decltype(guard)::value || qinb) __builtin_nontemporal_store(v4n, reinterpret_cast<nt4_t *>(dst));   // written once, not read again by this kernel
                }
                const unsigned cw = m < 2 ? cw0 : m < 4 ? cw1 : m < 6 ? cw2 : m < 8 ? cw3 : cw4;
                unsigned byte = (cw >> (8 * (3 - (2 * (m & 1) + hb)))) & 0xffu;      // bit 7 - j: slot 16 m + 8 hb + j
                if (__any(byte != 0u)) {
                    while (byte != 0u) {
                        const int b = 31 - __clz((int)byte);
                        byte &= ~(1u << b);
                        const int j = 7 - b;
                        const float x = tw[(8 * (j >> 2) + 2 * (j & 3)) * 32];
                        if (cnt < CAP) list_put(cnt, x, (uint32_t)(16 * m + 8 * hb + j));
                        cnt++;
                    }
                }
            }
        };
        if constexpr (DRAIN) __builtin_amdgcn_s_waitcnt(0x0F70);
        if (__all(qinb)) rows_out(std::false_type{}); else rows_out(std::true_type{});   // (the guarded copy: ragged end of the voxel list)
        if (inb && kh == 0) a.out1[(int64_t)FIB_F642_POLE * a.stride + vox] = xrow;
        if (cpole) { if (cnt < CAP) list_put(cnt, xrow, 160u); cnt++; }
    }
    // ---- minimum (gqi.jl:147), bounds of the mean (gqi.jl:164) over this half's 160 rows ---------------------------------
    float vmin = INFINITY;
    f32x2 vs2 = {0.0f, 0.0f};
#pragma unroll
    for (int m = 0; m < 10; m++)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            vmin = fq_min3(vmin, acc[m][r], acc[m][r + 1]);
            const f32x2 pr = {acc[m][r], acc[m][r + 1]};
            vs2 += pr;                                  // v_pk_add_f32
        }
    const float vsum = vs2[0] + vs2[1];
    // ---- top three of this half's candidates in the order of sortperm!(odf_peak, rev=true) (gqi.jl:198) -------------------
    Top3 t;
    top3_clear(t);
    const int nl = cnt < CAP ? cnt : CAP;
    for (int i = 0; __any(i < nl); i++)
        if (i < nl) {
            if constexpr (LB) top3_insert(t, __uint_as_float(lw[i * 64]), slotv[kh * 161 + (int)lb[i * 64]]);
            else top3_insert(t, __uint_as_float(lw[i * 128]), slotv[kh * 161 + (int)lw[i * 128 + 64]]);
        }
    // ---- merge the two lane halves of a voxel ---------------------------------------------------------------------------
    unsigned long long ok[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)t.k[k], 32), hi = (unsigned)__shfl_xor((int)(unsigned)(t.k[k] >> 32), 32);
        ok[k] = ((unsigned long long)hi << 32) | lo;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) top3_insert_key(t, ok[k]);
    const int cnt_o = __shfl_xor(cnt, 32);
    const int npos = cnt + cnt_o;                       // candidates are > 0: count(odf_peak .> 0), gqi.jl:200
    const float vmin_t = fq_min3(vmin, __shfl_xor(vmin, 32), xrow);
    const float vsum_t = (vsum + __shfl_xor(vsum, 32)) + xrow;
    const bool finite = fabsf(vsum_t) < INFINITY;       // false for NaN / Inf columns
    const bool redo = inb && (!finite || nonfinite || cnt > CAP || cnt_o > CAP);   // (nonfinite: the column is recomputed after this kernel)
    const float mean = vsum_t / (float)FQ_NV;
    const float eps = (2.1f * 5.9604645e-8f) * (float)FQ_NV * (fabsf(mean) + 2.0f * fabsf(fminf(vmin_t, 0.0f)));   // see odfmax_contribute
    if (kh == 0 && inb) {
        if (redo) {
            const int slot = atomicAdd(a.redo_count, 1);
            if (slot < a.redo_cap) a.redo_list[slot] = (int32_t)((unsigned)vox | (refix ? 0x80000000u : 0u));
        }
        a.mean_hi[vox] = redo ? __builtin_nanf("") : mean + eps;
        const int n = npos < 3 ? npos : 3;              // gqi.jl:151
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float px = 0.0f, py = 0.0f, pz = 0.0f, q = 0.0f;
            if (k < n && !redo) {
                const int iv = top3_index(t, k);
                px = vl[3 * iv]; py = vl[3 * iv + 1]; pz = vl[3 * iv + 2];          // gqi.jl:154-155
                q = peak_key_value((unsigned)(t.k[k] >> 32)) - vmin_t;              // gqi.jl:157-158
            }
            a.peak[k][vox] = px; a.peak[k][a.stride + vox] = py; a.peak[k][2 * a.stride + vox] = pz;
            a.qa[k][vox] = q;
        }
    }
    unsigned e = (kh == 0 && inb && !redo) ? enc_ordered(mean - eps) : 0u;
    // the wave's running lower bound of the maximum mean: ONE atomicMax per wave when the kernel ends (a thousand waves raising
    // the same word after every work item are waited for at the next stage's vmcnt(0))
    en_run = e > en_run ? e : en_run;
}

// ---- K2/K5, second form: the same f32 contraction on the bf16 matrix cores (16x the f32 MFMA rate) ----------
// An f32 number is EXACTLY the sum of three bf16 numbers (3 x 8 significant bits, round-to-nearest pieces):
//   a = a1 + a2 + a3 (split once on the host),   s = s1 + s2 + s3 (split in registers as the samples arrive),
// and a product of two bf16 numbers is exact in f32.  a*s = sum of the nine piece products; the six kept here
//   a1 s1 + (a1 s2 + a2 s1) + (a1 s3 + a2 s2 + a3 s1)
// miss only a2 s3 + a3 s2 + a3 s3 <= 2^-25 |a s|: less than half an ulp of the f32 product, i.e. every term enters
// the f32 accumulator at least as accurately as the f32 `fma` chain's own product rounding (measured against a
// float64 contraction the result is closer than the f32-MFMA kernel's: tools/gemm_accuracy.py).  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each):
// 192 instead of 512 matrix-core cycles per 16 frames.  The bf16 MFMA leaves 24 of its 32 cycles free for other
// issue, so the splitting (about 15 VALU per sample pair), the A-fragment reads (ds_read_b128) and the sample loads
// are software-pipelined one stage ahead and issued between the MFMAs of the current stage.
// Layout: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j] / S[k = 8h + j][voxel r], j = 0..7; a stage
// (16 frames) of the matrix is 3 pieces x MB blocks x 1 KiB in exactly the order the lanes read it (linear
// ds_read_b128: conflict free).  The NX extra rows stay f32: their coefficients come in by scalar loads.
// Persistent workgroups: NW waves own NW*32 voxels per work item and walk a list of work items; the stage ring
// (matrix pieces through LDS, samples through registers) runs across work-item boundaries.  Work items are dealt
// XCD by XCD so that the M tiles of one voxel group (DSI: 3) run side by side on one XCD and share its L2 copy of
// the samples.  Epilogue: a wave whose 32 voxels are contiguous in memory transposes each 32x32 block through a
// private 4-KiB LDS tile and writes it with 4 global_store_dwordx4 (8 rows x 128 B each) instead of 16
// global_store_dword: dword stores are issue-bound at ~6 B/clk/CU (measured: 30 000 cycles for the 161 rows).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {      // round-to-nearest-even, NaN stays NaN
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// FOLD (DSI with an antipodally symmetric lattice, see dsi_fold_kernel): the kernel reads the RAW frames and forms the folded
// sample t[J] = max(s[q_J],0) + max(s[-q_J],0) itself: 16 instead of 8 loads per lane and stage, but no 2.8-GB folded copy
// of the volume written and read back (the separate pre-pass ran at the HBM roofline and still cost 1.65 of 7.7 ms).
// The two lane halves of a load need different frames; their byte offsets come from an LDS table (relative to the
// lowest frame that the stage touches on that side: one buffer resource per stage and side).
constexpr int FKMAX = 512, FSMAX = FKMAX / KT;
// LDS of one instantiation: stage ring + per-wave transposition tiles + extra-row table + fold tables + fused-scan lists / tables
// (ONE && FOLD && H2 = the DSI pair kernel's tiles: the samples of both fold sides travel through LDS too -- 8 KiB per wave -- and the ODF
// tile pays for them with one transposition tile instead of two and byte-sized slots in the candidate lists)
template <int MB, int NX, int NW, bool FOLD, bool FUSE, bool H2, bool ONE = false>
constexpr int gemm3_lds_bytes() {
    constexpr bool SF = H2 && FOLD && ONE;
    return 2 * (H2 ? 2 : 3) * MB * 1024 + NW * (FUSE && !SF ? 4096 : 2048) + (NX > 0 ? (FUSE ? 2048 : 8192) : 0) +
           (FOLD ? 2 * FKMAX * 4 + 4 * FSMAX * 4 + (FUSE ? 0 : 2 * FKMAX * 4) : 0) + (FUSE ? NW * (SF ? FQ_LISTB : FQ_LIST) + FQ_TABB : 0) +
           (H2 && FUSE && !FOLD ? NW * (4096 + 512) : 0) +   // (SLDS: two sample tiles + the next items' mask bytes and list entries, per wave)
           (SF ? NW * (8192 + 256) : 0);                     // (.. two sample tiles of two sides + the list entries)
}
// ONE: the workgroup works on a single-tile image of its own (odf_dsi2_kernel: the DSI rows are cut into an ODF tile and a pdf tile
// with images of different shapes); the work list still deals the items of both tiles (a.ntile_m = 2), and with an even number
// of workgroups per XCD every workgroup keeps drawing items of its own tile
// H2: the operands travel as TWO fp16 pieces instead of three bf16 pieces (see "Two fp16 pieces" below): 3 MFMAs per block and
// 16 frames instead of 6
template <int MB, int NX, int NW, bool FOLD = false, bool FUSE = false, bool ONE = false, bool H2 = false>
__device__ __forceinline__ void gemm3_body(const GemmArgs &a, char *lds) {
    static_assert(!FUSE || (MB == 10 && NX == 1), "the fused peak scan is generated for 10 blocks + 1 extra row");
    constexpr int NPIECE = (H2 ? 2 : 3) * MB;           // 1-KiB pieces per stage
    constexpr int TILEB = NPIECE * 1024;                // bytes per stage
    constexpr int NA = (NPIECE + NW - 1) / NW;          // direct-to-LDS loads per wave and stage (a surplus load repeats the last piece)
    constexpr int WGV = NW * 32;                        // voxels per work item
    constexpr int NXA = NX > 0 ? NX : 1;
    constexpr int XTAB = NX > 0 ? (FUSE ? 2048 : 8192) : 0;   // coefficients of the extra rows, all stages of all M tiles: [ntile_m][NX][Kpad] f32
    constexpr int FTAB = FOLD ? 2 * FKMAX * 4 + 4 * FSMAX * 4 + (FUSE ? 0 : 2 * FKMAX * 4) : 0;   // (a tile with pdf rows: + the row -> frame tables)
    // SLDS (fused GQI on fp16 pieces; [r4] both tiles of the DSI pair kernel, SLDSF): the samples travel through LDS, see below
    constexpr bool SLDSF = H2 && FOLD && ONE;
    constexpr bool SLDS = (H2 && FUSE && !FOLD) || SLDSF;
    constexpr bool BOOK = SLDS && !FOLD;                      // the next item's mask bytes and list entries by LDS-DMA as well
    constexpr int BOOKB = BOOK ? 512 : (SLDSF ? 256 : 0);     // (SLDSF: the list entries only -- the mask byte is not looked at before the epilogue)
    constexpr int BOOKV = BOOK ? 64 : 0;                      // dword index of the list entries in the book
    constexpr int QLIST = SLDSF ? FQ_LISTB : FQ_LIST;
    constexpr int QTAB = FUSE ? NW * QLIST + FQ_TABB : 0;     // fused peak scan: candidate lists + lookup tables
    constexpr int TRB = FUSE && !SLDSF ? 4096 : 2048;         // per-wave transposition tile(s) of the epilogue
    // SLDS (fused GQI on fp16 pieces): the samples travel through LDS -- two tiles [16 frames][32 voxels] per wave, filled by
    // range-checked `buffer_load_dwordx4 .. lds` (a lane = 4 consecutive voxels of one frame: the voxel list is made of aligned quads),
    // two stages ahead of the split that reads them.  With the samples in registers a wave can have ONE stage in flight, a request
    // can precede its use by at most a stage, and a stage had settled at the ~1.5 us a sample load takes under load (twice what its
    // MFMAs need); requests in LDS cost no registers, so they also cross an item's epilogue.
    // SLDSF: a tile per fold side, [2 sides][16 folded frames][32 voxels]: a lane's DMA row = the frame the fold table names for it
    constexpr int SSLOT = FOLD ? 4096 : 2048;                 // bytes per wave and sample slot
    constexpr int NSREQ = FOLD ? 4 : 2;                       // DMA instructions of a stage's sample request
    constexpr int STILE = SLDS ? NW * (2 * SSLOT + BOOKB) : 0;
    static_assert(2 * TILEB + NW * TRB + XTAB + FTAB + QTAB + STILE == gemm3_lds_bytes<MB, NX, NW, FOLD, FUSE, H2, ONE>(), "LDS carve-up");
    static_assert(gemm3_lds_bytes<MB, NX, NW, FOLD, FUSE, H2, ONE>() <= 160 * 1024, "LDS of a CU");
    uint64_t *q_posoff = reinterpret_cast<uint64_t *>(lds + 2 * TILEB + NW * TRB + XTAB + FTAB + (FUSE ? NW * QLIST : 0));   // [320] matrix row -> byte offset of its output row
    int *q_slotv = reinterpret_cast<int *>(q_posoff + FQ_NPOS);                                                                               // [2][161] (half, slot) -> vertex
    float *q_vl = reinterpret_cast<float *>(q_slotv + FQ_NSLOT);                                                   // [321][3]
    uint32_t *f_off = reinterpret_cast<uint32_t *>(lds + 2 * TILEB + NW * TRB + XTAB);   // [2][FKMAX] byte offset of sample J's frame, side a / b
    int32_t *f_bs = reinterpret_cast<int32_t *>(f_off + 2 * FKMAX);                        // [FSMAX][4] per stage: lowest frame the stage touches, frames spanned (0: none), side a | side b
    int32_t *f_row = f_bs + 4 * FSMAX;                                                   // [2][FKMAX] (FOLD, not FUSE) folded pdf row -> its two frames
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kh = lane >> 5;
    const int ntiles = a.Kpad / KT;
    const uint32_t a_off = (uint32_t)lane * 16;
    if (NX > 0) {
        float *xt = reinterpret_cast<float *>(lds + 2 * TILEB + NW * TRB);
        for (int i = tid; i < (ONE ? 1 : a.ntile_m) * NX * a.Kpad; i += NW * 64) xt[i] = a.Aextra[i];
    }
    if constexpr (FUSE) {
        for (int i = tid; i < FQ_NPOS; i += NW * 64) q_posoff[i] = (uint64_t)fib_f642_pos_vertex_dev[i] * (uint64_t)a.stride * 4u;
        for (int i = tid; i < FQ_NSLOT; i += NW * 64) q_slotv[i] = fib_f642_slot_vertex_dev[i];
        for (int i = tid; i < 3 * FQ_NV; i += NW * 64) q_vl[i] = a.verts[i];
    }
    const char *Sbase = reinterpret_cast<const char *>(a.S);
    const uint32_t row_bytes = (uint32_t)(a.stride * 4);
    if (FOLD) {
        for (int i = tid; i < 2 * ntiles; i += NW * 64) {
            const int side = i / ntiles, t = i - side * ntiles;
            const int32_t *fr = side ? a.rowB : a.rowA;
            int lo = 0x7fffffff, hi = -1;
            for (int j = 0; j < KT; j++) {
                const int J = t * KT + j;
                const int f = J < a.K ? fr[J] : -1;
                if (f >= 0) { lo = f < lo ? f : lo; hi = f > hi ? f : hi; }
            }
            f_bs[4 * t + 2 * side] = hi >= 0 ? lo : 0;
            f_bs[4 * t + 2 * side + 1] = hi >= 0 ? hi - lo + 1 : 0;
        }
        __syncthreads();
        for (int i = tid; i < 2 * a.Kpad; i += NW * 64) {
            const int side = i / a.Kpad, J = i - side * a.Kpad;
            const int f = J < a.K ? (side ? a.rowB : a.rowA)[J] : -1;
            // (the host checked that a stage's span times the frame size stays below 0xE0000000; 0xF0000000 + 4 vox is past every span)
            f_off[side * FKMAX + J] = f >= 0 ? (uint32_t)(f - f_bs[4 * (J / KT) + 2 * side]) * row_bytes : 0xF0000000u;
            if (!FUSE) f_row[side * FKMAX + J] = f;     // (the folded pdf rows are the folded samples: same tables, dsi_fold_kernel)
        }
        __syncthreads();
    }

    // ---- work list of this workgroup ---------------------------------------------------------------------------
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int nlive = __builtin_amdgcn_readfirstlane(a.nlive[0]);
    const int ntile_n = (nlive + WGV - 1) / WGV;
    struct Work { int tile_m; int tile_n; bool valid; };
    auto work_at = [&](int i) {
        const int w = wslot + i * nslot;
        Work r;
        // (wave-uniform by construction; told to the compiler so that the stage loop's piece addresses are scalar arithmetic -- ntile_n
        // comes from a vector load and would otherwise drag 64-bit multiplications onto the VALU in every stage)
        constexpr bool ONE_M = ONE || (FUSE && !FOLD);   // (the fused GQI form exists for single-tile matrices only: finish_plan)
        r.tile_m = ONE_M ? 0 : __builtin_amdgcn_readfirstlane(w % a.ntile_m);
        r.tile_n = __builtin_amdgcn_readfirstlane(ONE ? (a.one_slot + i * a.one_stride) * 8 + xcd : (ONE_M ? w : w / a.ntile_m) * 8 + xcd);
        r.valid = r.tile_n < ntile_n;
        return r;
    };
    auto vidx_at = [&](const Work &w) -> int32_t {       // list entry of this lane's voxel (stale / clamped past the end)
        const int64_t sl = (int64_t)w.tile_n * WGV + wave * 32 + col;
        return a.vidx[sl < a.nvox ? sl : a.nvox - 1];
    };
    Work cur = work_at(0);
    if (!cur.valid) return;
    int32_t vraw = vidx_at(cur);
    Work nxt = work_at(1);
    int32_t vraw_nxt = vidx_at(nxt);     // (clamped: always a valid address)

    const uint32_t lds_l = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char *)lds);
    // (SLDS: the pieces through ONE buffer resource over the image -- per piece a scalar offset and the LDS address, nothing else: a
    // wave issues one instruction per ~4 cycles, and 64-bit address arithmetic per piece was 45 of a stage's ~250 instructions)
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    i32x4_t rsrcA;
    {
        const uint64_t b = reinterpret_cast<uint64_t>(a.At3);
        rsrcA[0] = (int)(uint32_t)b; rsrcA[1] = (int)(uint32_t)((b >> 32) & 0xffffu);
        rsrcA[2] = (int)((uint32_t)(ONE ? 1 : a.ntile_m) * (uint32_t)ntiles * (uint32_t)TILEB); rsrcA[3] = 0x00020000;
    }
    auto stage_A = [&](int tile_m, int t, int buf) {
        const char *g = reinterpret_cast<const char *>(a.At3) + ((size_t)tile_m * ntiles + t) * TILEB;
        char *l = lds + buf * TILEB;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            int p = wave + i * NW;
            p = p < NPIECE ? p : NPIECE - 1;
            if constexpr (H2) {
                // (as inline assembly through ONE buffer resource over the image: per piece a scalar offset and the LDS address, nothing
                // else.  SLDS needs it -- a builtin LDS-DMA in flight makes hipcc close every barrier with s_waitcnt vmcnt(0), and that
                // would wait for the sample request that is meant to stay in flight across it; [r4] the other fp16-piece kernels take the
                // same path for its instruction count: they close a stage with an explicit vmcnt(0) + barrier, which covers these requests)
                const uint32_t d = lds_l + (uint32_t)(buf * TILEB + p * 1024);   // (integer arithmetic on the LDS address: a pointer cast per piece is a null check per piece)
                const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)((tile_m * ntiles + t) * TILEB + p * 1024));   // (wave-uniform by construction)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(d), "v"(a_off), "s"(rsrcA), "s"(so) : "memory");
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + p * 1024 + a_off),
                                                 (__attribute__((address_space(3))) void *)(l + p * 1024), 16, 0, 0);
            }
        }
    };
    float braw[8], brawb[FOLD ? 8 : 1];
    // sample j of the lane = frame t*16 + 8h + j.  One buffer resource per stage, based at frame t*16 and ending with the
    // frame list: loads past it return 0.0 without a memory access (the padded columns of A are zero).  The frame
    // offsets j*row_bytes go in as scalar offsets, the lane's (voxel + 8h rows) as the 32-bit vector offset.
    auto load_B = [&](int t, uint32_t s_off, bool live) {
        if constexpr (FOLD) {
            const int ba = __builtin_amdgcn_readfirstlane(f_bs[4 * t]), bb = __builtin_amdgcn_readfirstlane(f_bs[4 * t + 2]);
            const int sa = live ? __builtin_amdgcn_readfirstlane(f_bs[4 * t + 1]) : 0, sb = live ? __builtin_amdgcn_readfirstlane(f_bs[4 * t + 3]) : 0;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(Sbase + (int64_t)ba * row_bytes), 0, (int)((uint32_t)sa * row_bytes), 0x00020000);
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(Sbase + (int64_t)bb * row_bytes), 0, (int)((uint32_t)sb * row_bytes), 0x00020000);
            const u32x4_t *pa = reinterpret_cast<const u32x4_t *>(f_off + t * KT + 8 * kh), *pb = reinterpret_cast<const u32x4_t *>(f_off + FKMAX + t * KT + 8 * kh);
            const u32x4_t oa0 = pa[0], oa1 = pa[1], ob0 = pb[0], ob1 = pb[1];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                braw[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (int)(s_off + (j < 4 ? oa0[j & 3] : oa1[j & 3])), 0, 0));
                brawb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (int)(s_off + (j < 4 ? ob0[j & 3] : ob1[j & 3])), 0, 0));
            }
            return;
        }
        const int rem = live ? a.K - t * KT : 0;        // frames from this stage's first to the end of the list
        const uint64_t span = (uint64_t)(rem > 0 ? rem : 0) * row_bytes;
        const uint32_t nrec = span > 0xffffffffull ? 0xffffffffu : (uint32_t)span;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(rem > 0 ? Sbase + (int64_t)t * KT * row_bytes : Sbase), 0, (int)nrec, 0x00020000);
#pragma unroll
        for (int j = 0; j < 8; j++)
            braw[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)s_off, (int)(j * row_bytes), FUSE ? 2 : 0));   // (aux 2 = nt: the samples are read once)
    };
    // SLDS: samples of stage t (frames 16 t ..) of the voxel quads at byte offsets qoff into this wave's sample tile `slot`.  Inline
    // assembly: hipcc treats its own LDS-DMA builtins as stores that a later LDS read may depend on
    char *stile = lds + 2 * TILEB + NW * TRB + XTAB + FTAB + QTAB + wave * (2 * SSLOT);
    const uint32_t stile_l = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char *)stile);
    const uint32_t lane_rows = (uint32_t)(lane >> 3) * row_bytes;
    // (SLDSF) what a stage's request needs from the fold tables: read at the top of the split, next to the sample reads -- one LDS round
    // trip instead of six in a row behind the split
    struct FoldReq { u32x4_t bs; uint32_t o[4]; };
    auto fold_tab = [&](int t) {
        FoldReq q;
        if constexpr (FOLD) {
            q.bs = *reinterpret_cast<const u32x4_t *>(f_bs + 4 * t);
            const uint32_t *fo = f_off + t * KT + (lane >> 3);
            q.o[0] = fo[0]; q.o[1] = fo[8]; q.o[2] = fo[FKMAX]; q.o[3] = fo[FKMAX + 8];
        }
        return q;
    };
    auto load_S = [&](int t, uint32_t qoff, bool live, int slot, const FoldReq &fq) {
        if constexpr (FOLD) {
            // per side one buffer resource over the frames the stage touches (f_bs), the lane's frame by the fold table: a partner that
            // does not exist has an offset past every span and reads 0.0
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const uint32_t fb = (uint32_t)__builtin_amdgcn_readfirstlane((int)fq.bs[2 * side]);
                const uint32_t fsr = (uint32_t)__builtin_amdgcn_readfirstlane((int)fq.bs[2 * side + 1]);
                const uint32_t fs = live ? fsr : 0u;
                const uint64_t b = reinterpret_cast<uint64_t>(Sbase) + (uint64_t)fb * row_bytes;
                i32x4_t r;
                r[0] = (int)(uint32_t)b;
                r[1] = (int)(uint32_t)((b >> 32) & 0xffffu);
                r[2] = (int)(fs * row_bytes);
                r[3] = 0x00020000;
                const uint32_t v0 = qoff + fq.o[2 * side], v1 = qoff + fq.o[2 * side + 1];
                const uint32_t d0 = stile_l + (uint32_t)(slot * SSLOT + side * 2048);
                // (default cache policy: the partner tile's workgroup reads the same samples from the XCD's L2)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(d0), "v"(v0), "s"(r) : "memory");
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(d0 + 1024u), "v"(v1), "s"(r) : "memory");
            }
            return;
        }
        const int rem = live ? a.K - t * KT : 0;
        const uint64_t span = (uint64_t)(rem > 0 ? rem : 0) * row_bytes;
        const uint64_t b = reinterpret_cast<uint64_t>(Sbase) + (rem > 0 ? (uint64_t)(uint32_t)(t * KT) * row_bytes : 0ull);
        i32x4_t r;
        r[0] = (int)(uint32_t)b;
        r[1] = (int)(uint32_t)((b >> 32) & 0xffffu);
        r[2] = (int)(span > 0xffffffffull ? 0xffffffffu : (uint32_t)span);
        r[3] = 0x00020000;
        const uint32_t v0 = qoff + lane_rows, v1 = v0 + 8u * row_bytes;   // (in the vector offset: that is what the range check sees)
        const uint32_t d0 = stile_l + (uint32_t)slot * 2048u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" :: "s"(d0), "v"(v0), "s"(r) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" :: "s"(d0 + 1024u), "v"(v1), "s"(r) : "memory");
    };
    // SLDS: the NEXT item's mask byte (book[lane]) and the list entry of the item after it (book[64 + lane]) also come by LDS-DMA, one
    // item ahead: a plain load at the top of an item would wait (vmcnt, in issue order) until the epilogue's row stores have drained
    uint32_t *book = reinterpret_cast<uint32_t *>(lds + 2 * TILEB + NW * TRB + XTAB + FTAB + QTAB + NW * 2 * SSLOT + wave * BOOKB);
    const uint32_t book_l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char *)book));
    auto book_mask = [&](int64_t voxn) {
        const uint8_t *mp = a.mask + voxn;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, off" :: "s"(book_l), "v"(mp) : "memory");
    };
    auto book_vidx = [&](const Work &w) {
        const int64_t sl = (int64_t)w.tile_n * WGV + wave * 32 + col;
        const int32_t *vp = a.vidx + (sl < a.nvox ? sl : a.nvox - 1);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" :: "s"(book_l + 4u * BOOKV), "v"(vp) : "memory");
    };
    auto lane_state = [&](const Work &w, int32_t vr, bool &inb, int64_t &vox, uint32_t &s_off) {
        inb = (int64_t)w.tile_n * WGV + wave * 32 + col < nlive;
        vox = inb ? vr : 0;
        s_off = (uint32_t)((vox + (FOLD ? (int64_t)0 : (int64_t)8 * kh * a.stride)) * 4);   // frame 8h of a stage (FOLD: the table's frame), this voxel (nvox <= 2^26)
    };

    f32x16 acc[MB];
    float xacc[NXA];
    float vmax = 0.0f, vnf = 0.0f;
    unsigned en_run = 0u;                               // FUSE: running lower bound of the maximum mean (gemm3_epilogue_fused)
    // Anti-phase halves (FUSE, 8 waves = 2 per SIMD): waves 0-3 ("early") run the MFMA block of stage t first and split the samples
    // of stage t+1 afterwards, waves 4-7 split stage t first and run its MFMA block afterwards, so that on every SIMD one wave's
    // VALU work falls into the other wave's MFMA block instead of both splitting with the matrix cores idle (s_setprio keeps the
    // MFMA block ahead of the splitting wave: without it the two instruction streams just alternate).
    constexpr bool ANTI = (FUSE || SLDSF) && NW == 8;   // ([r4] the DSI pair kernel's pdf tile too: it has no extra rows, whose sums only the fused epilogue takes from xfin)
    static_assert(!ANTI || FUSE || NX == 0, "anti-phase halves without the fused epilogue: no extra rows");
    const bool early = ANTI && (a.anti & 1) != 0 && wave < NW / 2;
    const bool prio = ANTI && (a.anti & 2) != 0;
    float xfin[NXA], vmax_fin = 0.0f;                   // early waves: the sums of the item being accumulated (its last split is one stage ahead)
#pragma unroll
    for (int x = 0; x < NXA; x++) xfin[x] = 0.0f;
    auto clear = [&](bool keep_sums) {
#pragma unroll
        for (int m = 0; m < MB; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
#pragma unroll
        for (int x = 0; x < NXA; x++) xacc[x] = keep_sums ? xacc[x] : 0.0f;
        vmax = keep_sums ? vmax : 0.0f; vnf = keep_sums ? vnf : 0.0f;
    };
    clear(false);

    // clamp (gqi.jl:140, dsi.jl:209), positivity / non-finite tracking, exact 3-way bf16 split (H2: two fp16 pieces), extra rows
    //
    // Two fp16 pieces (H2).  s * 2^k = h + l + e with h = RN16(s 2^k), l = RN16(s 2^k - h) and |e| <= 2^-23 |s 2^k| as long as l is
    // a normal fp16 number: 23 of the sample's 24 significant bits.  The matrix is split the same way on the host (scaled by the
    // power of two sa), and a product is a_h s_l + a_l s_h + a_h s_h: exact piece products, f32 accumulate, the dropped terms
    // (a_l s_l and the two residues) are each <= 2^-22 |a s| and of either sign.  Measured against a float64 contraction the result
    // is as close as the six-product bf16 form (the error of both is the f32 accumulation's) and closer than an f32 fma chain
    // (tools/gemm_accuracy.py); the three-piece bf16 form stays available (FIBERS_ODF_EXACT=1).
    // fp16 has 5 exponent bits, so every voxel carries its own power of two: 2^k puts the running maximum of its clamped samples
    // into [2^6, 2^7) when the item's first stage is split -- l stays normal for samples down to 2^-16 of that maximum, and below
    // that the absolute error is <= 2^-25 (half an fp16 subnormal step) against a maximum >= 2^6.  A later sample that would
    // reach 2^15 (it is > 256 x everything the voxel held so far) lowers k and the accumulators are multiplied by the (exact) power
    // of two in between.  k depends on the voxel's own samples only: results do not depend on which voxels share a wave.
    u32x4_t bp[3];
    constexpr int H2_TARGET = 127 + 6;                  // biased exponent of the scaled running maximum when k is chosen
    int kexp = 127, kexp_fin = 127;                     // H2: biased exponent of 2^k (kexp_fin: of the item an early wave is finishing)
    // (SLDS: reads the stage's samples from sample tile `slot` and, when done, requests stage tn of the quads at qo into that tile)
    auto split = [&](int tile_m, int t, int slot = 0, int tn = 0, uint32_t qo = 0, bool live = false) {
        float cs[H2 ? 8 : 1];
        FoldReq fq;
        if constexpr (SLDS) {
            fq = fold_tab(tn);
            const float *sp = reinterpret_cast<const float *>(stile + slot * SSLOT) + (8 * kh) * 32 + col;
#pragma unroll
            for (int j = 0; j < 8; j++) braw[j] = sp[j * 32];
            if constexpr (FOLD) {
#pragma unroll
                for (int j = 0; j < 8; j++) brawb[j] = sp[512 + j * 32];
            }
            // (all eight reads before the first use: left alone hipcc reads a pair, waits, clamps it, reads the next pair into the same
            // registers -- four LDS round trips in a row at the top of every split, and a wave issues one instruction per ~4 cycles)
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            float x0 = braw[2 * jj], x1 = braw[2 * jj + 1];
            if (FOLD) {                                 // t[J] = max(s[q],0) + max(s[-q],0) (dsi.jl:209; an absent partner loads 0)
                x0 = clamp_sample(x0) + clamp_sample(brawb[2 * jj]);
                x1 = clamp_sample(x1) + clamp_sample(brawb[2 * jj + 1]);
            }
            float c0, c1;
            c0 = FOLD ? x0 : clamp_sample(x0);
            c1 = FOLD ? x1 : clamp_sample(x1);
            vmax = max3_nan(vmax, c0, c1);              // (vnf is derived from it at the end of the item)
            if constexpr (H2) { cs[2 * jj] = c0; cs[2 * jj + 1] = c1; }
            else {
                const uint32_t h = cvt_pk_bf16(c0, c1);
                const float r0 = c0 - __uint_as_float(h << 16), r1 = c1 - __uint_as_float(h & 0xffff0000u);      // exact
                const uint32_t m = cvt_pk_bf16(r0, r1);
                const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xffff0000u);      // exact
                const uint32_t l = cvt_pk_bf16(q0, q1);                                                           // exact
                bp[0][jj] = h; bp[1][jj] = m; bp[2][jj] = l;
            }
            if (NX > 0) {                               // extra rows: f32 fma, coefficients from the LDS table (frames 8h + 2jj, +1)
                const f32x2 *ex = reinterpret_cast<const f32x2 *>(lds + 2 * TILEB + NW * TRB) + ((tile_m * NX) * a.Kpad + t * KT + 8 * kh) / 2 + jj;
#pragma unroll
                for (int x = 0; x < NX; x++) {
                    const f32x2 e = ex[x * (a.Kpad / 2)];
                    xacc[x] = __builtin_fmaf(e[0], c0, xacc[x]);
                    xacc[x] = __builtin_fmaf(e[1], c1, xacc[x]);
                }
            }
        }
        if constexpr (H2) {
            // The exchange between the two k halves of a voxel is needed when an item opens (k is chosen) and when a sample would reach
            // 2^15 after scaling (k is lowered: rare): a lane-local test and a wave-uniform branch decide, everything else is behind it
            const float sck = __uint_as_float((uint32_t)kexp << 23);
            if (t == 0 || __any(vmax * sck >= 32768.0f)) {       // (NaN: false -- the column is repaired anyway)
                // the voxel's running maximum over both k halves (v_permlane32_swap: no LDS round trip in the split)
                const auto vsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(vmax), __float_as_uint(vmax), false, false);
                const float mall = max3_nan(__uint_as_float(vsw[0]), __uint_as_float(vsw[1]), 0.0f);
                const int e = (int)((__float_as_uint(mall) >> 23) & 0xffu);          // (255: NaN / +Inf)
                int kfit = e == 0 ? 127 + 60 : 127 + (H2_TARGET - e);
                kfit = kfit < 1 ? 1 : (kfit > 253 ? 253 : kfit);
                if (t == 0) kexp = e == 255 ? 127 : kfit;
                else {
                    const bool lower = e != 255 && e + kexp >= 254 + 15;
                    if (__any(lower)) {                      // per-lane factor
                        const int d = lower ? kfit - kexp : 0;
                        const float f = __uint_as_float((uint32_t)(127 + (d < -126 ? -126 : d)) << 23);
#pragma unroll
                        for (int m = 0; m < MB; m++)
#pragma unroll
                            for (int r = 0; r < 16; r++) acc[m][r] *= f;
                        kexp = lower ? kfit : kexp;
                    }
                }
            }
            // four instructions per sample pair: h = RN16(c 2^k) and l = RN16(fma(c, 2^k, -h)), each one v_fma_mix{lo,hi}_f16 -- the
            // product and the difference are exact in f32 and the instruction rounds once to fp16.  (Written out in C++ hipcc makes it
            // seven, or pairs the multiplications into v_pk_mul_f32 with a 64-bit register operand whose upper half is undefined: that
            // half can land on a register a load is still writing to, and the waitcnt insertion then puts s_waitcnt vmcnt(0) in
            // front of the split, tools/check_loop_waits.py.)
            {
                const float sc = __uint_as_float((uint32_t)kexp << 23);
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    uint32_t h, l;
                    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(cs[2 * jj]), "v"(sc));
                    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(cs[2 * jj + 1]), "v"(sc));
                    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(cs[2 * jj]), "v"(sc), "v"(h));
                    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(cs[2 * jj + 1]), "v"(sc), "v"(h));
                    bp[0][jj] = h; bp[1][jj] = l;
                }
            }
            if constexpr (SLDS) load_S(tn, qo, live, slot, fq);
        }
    };

    bool inb; int64_t vox; uint32_t s_off;
    lane_state(cur, vraw, inb, vox, s_off);
    uint32_t qoff = SLDS ? (uint32_t)__shfl((int)vox, 4 * (lane & 7)) * 4u : 0u, qoff_n = 0u;   // SLDS: byte offset of the voxel quad this lane requests
    // ---- ring prologue: stage 0's pieces into LDS, its samples into registers (SLDS: stages 0 and 1 into the sample tiles) ----------
    stage_A(cur.tile_m, 0, 0);
    if constexpr (SLDS) { stage_A(cur.tile_m, 1, 1); load_S(0, qoff, true, 0, fold_tab(0)); load_S(1, qoff, true, 1, fold_tab(1)); if constexpr (BOOK) book_mask(vox); }
    else load_B(0, s_off, true);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();                                     // (also: the extra rows' table is complete)
    if (ANTI && early) split(cur.tile_m, 0, 0, 2, qoff, true);
    int g = 0;                                           // stages done: ring position
    FIB_STAMP_BEGIN();
    FIB_PHASE_VARS();
    for (;;) {
        if constexpr (ONE) {
            // The two tiles of a voxel group read the same samples.  With as many ODF-tile as pdf-tile workgroups, workgroup p of each
            // kind walks the same groups; the pdf-tile workgroup (the faster one) starts an item only when its partner has started
            // it, so that the second reader finds the samples in the XCD's L2.  A hint, not a protocol: the wait is bounded, and
            // nothing but speed depends on it (relaxed agent-scope accesses of a counter, no data is handed over).
            if (a.pair_role != 0) {
                const int it = g / ntiles;
                if (a.pair_role == 1) {
                    if (tid == 0) __hip_atomic_store(a.pair_flags + (blockIdx.x & 7) * 32 + a.one_slot, (unsigned)it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    // [r4] any split of the XCD's workgroups between the two tiles: voxel group k of the XCD is item k / dsi_na of ODF-tile workgroup k % dsi_na
                    const int k = a.one_slot + it * a.one_stride;
                    const unsigned *flag = a.pair_flags + (blockIdx.x & 7) * 32 + k % a.dsi_na;
                    const unsigned item = (unsigned)(k / a.dsi_na) + 1u;
                    if (tid == 0) {
                        for (int spin = 0; spin < 4000; spin++) {
                            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= item) break;
                            __builtin_amdgcn_s_sleep(8);
                        }
                    }
                    __syncthreads();
                }
            }
        }
        float sraw = 0.0f;                               // DSI: the sample that sum(p) is a multiple of (dsi.jl:224-225)
        if (a.scale_frame >= 0) sraw = *reinterpret_cast<const float *>(Sbase + (int64_t)a.scale_frame * a.stride * 4 + (uint32_t)(vox * 4));
        bool lv;                                         // voxels of a listed quad that are outside the mask: zeros
        if constexpr (BOOK) lv = inb && (book[lane] & 0xffu) != 0u;
        else lv = inb && a.mask[vox] != 0;
        // (unconditionally: vraw_nxt is a load, and a load whose only use sits behind a branch stays "in flight" for hipcc's waitcnt
        // insertion on the other path -- it then guards the first overwrite of a register near it with an s_waitcnt vmcnt(0) in
        // the middle of the stage loop, behind the next stage's loads: 11 % of the fused kernel, tools/check_loop_waits.py)
        bool inb_n = false; int64_t vox_n = 0; uint32_t s_off_n = 0;
        lane_state(nxt, vraw_nxt, inb_n, vox_n, s_off_n);
        if constexpr (SLDS) qoff_n = (uint32_t)__shfl((int)vox_n, 4 * (lane & 7)) * 4u;
        if constexpr (BOOK) {
            book_mask(vox_n);                            // (behind the read of book[lane] above)
            book_vidx(work_at(g / ntiles + 2));
        }
        if constexpr (SLDSF) book_vidx(work_at(g / ntiles + 2));
        for (int t = 0; t < ntiles; t++, g++) {
            const int cb = g & 1;
            const char *L = lds + cb * TILEB;
            FIB_PHASE(g / ntiles, wave, 1);             // stage top
            const bool w1 = t + 1 < ntiles;
            const int tm_n = w1 ? cur.tile_m : (nxt.valid ? nxt.tile_m : cur.tile_m);
            if constexpr (SLDS) {
                // pieces first, the sample request (at the end of the split) last: the stage's closing wait leaves exactly that request
                // in flight.  A wave that splits first takes stage t (ring position g) and asks for position g + 2
                // (an item's second stage was requested before the previous item's row stores went out -- below -- and the epilogue has
                // waited for it: stage 0 asks for no pieces and closes without a wait)
                if (t > 0) stage_A(tm_n, w1 ? t + 1 : 0, cb ^ 1);
                if (!(ANTI && early)) {
                    const bool in = t + 2 < ntiles;
                    split(cur.tile_m, t, g & 1, in ? t + 2 : t + 2 - ntiles, in ? qoff : qoff_n, in || nxt.valid);
                }
                FIB_PHASE(g / ntiles, wave, 2);
            } else {
            if (!(ANTI && early)) split(cur.tile_m, t);
            FIB_PHASE(g / ntiles, wave, 2);             // (late waves: split done)
            // the next stage (it may open the next work item): pieces into the other buffer, samples into braw
            stage_A(tm_n, w1 ? t + 1 : 0, cb ^ 1);
            load_B(w1 ? t + 1 : 0, w1 ? s_off : s_off_n, w1 || nxt.valid);
            }
            __builtin_amdgcn_sched_barrier(0);          // the requests go out before the MFMA block, not after it
            FIB_PHASE(g / ntiles, wave, 3);             // requests issued
            if constexpr (H2) {
                const f16x8_t b0 = __builtin_bit_cast(f16x8_t, bp[0]), b1 = __builtin_bit_cast(f16x8_t, bp[1]);
                const f16x8_t *LA = reinterpret_cast<const f16x8_t *>(L) + lane;
                f16x8_t a1 = LA[MB * 64], a0 = LA[0];
                if (prio) __builtin_amdgcn_s_setprio(2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MB; m++) {
                    f16x8_t n1 = a1, n0 = a0;
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[m], 0, 0, 0);     // smallest terms first
                    if (m + 1 < MB) n1 = LA[(MB + m + 1) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[m], 0, 0, 0);
                    if (m + 1 < MB) n0 = LA[(m + 1) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    a1 = n1; a0 = n0;
                }
            } else {
            const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, bp[0]), b1 = __builtin_bit_cast(bf16x8_t, bp[1]), b2 = __builtin_bit_cast(bf16x8_t, bp[2]);
            const bf16x8_t *LA = reinterpret_cast<const bf16x8_t *>(L) + lane;
            // The fragment reads are pinned (sched_barrier) 2-5 MFMAs ahead of their first use, each into the registers
            // its predecessor has just left: left to itself hipcc sinks every ds_read to the MFMA that needs it.
            bf16x8_t a2 = LA[(2 * MB) * 64], a1 = LA[(1 * MB) * 64], a0 = LA[0];
            if (prio) __builtin_amdgcn_s_setprio(2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MB; m++) {
                bf16x8_t n2 = a2, n1 = a1, n0 = a0;
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[m], 0, 0, 0);     // smallest terms first
                if (m + 1 < MB) n2 = LA[(2 * MB + m + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[m], 0, 0, 0);
                if (m + 1 < MB) n0 = LA[(m + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[m], 0, 0, 0);
                if (m + 1 < MB) n1 = LA[(1 * MB + m + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a2 = n2; a1 = n1; a0 = n0;
            }
            }
            FIB_PHASE(g / ntiles, wave, 4);             // MFMA block issued
            if constexpr (ANTI) {
                if (prio) __builtin_amdgcn_s_setprio(0);
                if (early) {                              // the samples requested above: the next stage's split, now
                    if (!w1) {                            // .. which opens the next work item: close this item's sums first
#pragma unroll
                        for (int x = 0; x < NXA; x++) { xfin[x] = xacc[x]; xacc[x] = 0.0f; }
                        vmax_fin = vmax; vmax = 0.0f; vnf = 0.0f;
                        kexp_fin = kexp;
                    }
                    if constexpr (SLDS) {                // stage t + 1 (position g + 1), then the request for position g + 3
                        if (w1) { const bool in = t + 3 < ntiles; split(cur.tile_m, t + 1, (g + 1) & 1, in ? t + 3 : t + 3 - ntiles, in ? qoff : qoff_n, in || nxt.valid); }
                        else split(tm_n, 0, (g + 1) & 1, 2, qoff_n, nxt.valid);
                    } else {
                        split(tm_n, w1 ? t + 1 : 0);
                    }
                }
            }
            FIB_PHASE(g / ntiles, wave, 5);             // (early waves: next split done)
            // vmcnt(2): everything but the sample request just issued (2 instructions) has landed.  Stage 0 of an item closes without a
            // wait -- all it and stage 1 need was requested before the previous item's stores and the epilogue has waited for it --
            // EXCEPT in a workgroup's first item: there an early wave's request for stage 2 went out behind the prologue's wait
            // (a stale sample tile in stage 1 otherwise: seen as a rare wrong ODF under concurrent launches, tests/test_gpu_hosttier.py)
            if constexpr (SLDS) { if (t > 0 || g == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | NSREQ); }   // (SLDSF: four instructions)
            else
            __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0): the next stage's pieces and samples have landed
            FIB_PHASE(g / ntiles, wave, 6);             // loads landed
            if constexpr (SLDS) __builtin_amdgcn_s_barrier();   // (no fence: this wave's LDS traffic of the stage is reads that its MFMAs have consumed)
            else
            __syncthreads();
            FIB_PHASE(g / ntiles, wave, 7);             // barrier passed
        }
        if constexpr (SLDS) stage_A(nxt.valid ? nxt.tile_m : cur.tile_m, 1, (g & 1) ^ 1);   // the next item's second stage: the buffer the last stage has just left
        {
            // the voxel's clamped-sample maximum over both k halves; vnf = NaN iff it is NaN or +Inf (the epilogues' "non-finite sample" flag)
            float vm = early ? vmax_fin : vmax;
            vm = max3_nan(vm, __shfl_xor(vm, 32), 0.0f);
            // H2: a voxel whose largest sample is a denormal number has no power of two that brings it into fp16's range (2^k is a
            // float here); where the repair list exists (GQI) it is handed over like a voxel with a +Inf sample and recomputed as a
            // plain f32 chain.  (DSI divides by sum(p) ~ that sample: Inf / NaN in the reference as well.)
            if (H2 && a.scale_frame < 0 && a.fix_list != nullptr && vm > 0.0f && vm < 1.17549435e-38f) vm = INFINITY;
            const float vn = vm < INFINITY ? 0.0f : __builtin_nanf("");
            // H2: the accumulators hold sa 2^k times the sums
            const float asc = H2 ? __uint_as_float((uint32_t)(254 - (early ? kexp_fin : kexp)) << 23) * a.h2_inv_sa : 1.0f;
            if constexpr (FUSE) {
                float fscale = 1.0f;                      // DSI (the FOLD form): 1 / sum(p), NaN where a sample is not finite (see gemm3_epilogue)
                if (FOLD) { const float s0 = sraw < 0.0f ? 0.0f : sraw; fscale = vn != vn ? __builtin_nanf("") : 1.0f / (a.scale_coef * s0); }
                gemm3_epilogue_fused<NW, false, FOLD || H2, H2 && !FOLD, SLDS, SLDSF ? 1 : 2, SLDSF, SLDSF ? FQ_CAPB : FQ_CAP>(a, acc, early ? xfin[0] : xacc[0], vm, vn, inb, lv, vox, lane, lds + 2 * TILEB + wave * TRB,
                                                            lds + 2 * TILEB + NW * TRB + XTAB + FTAB + wave * QLIST, q_posoff, q_slotv, q_vl, en_run,
                                                            fscale * asc, fscale);
            }
            else {
                // (the lane index is laundered per work item: otherwise every lane-derived row index, table lookup and 64-bit row
                // address of the epilogue is hoisted out of the persistent loop and parked in registers / scratch)
                int le = lane;
                asm volatile("" : "+v"(le));
                // (SLDS: everything in flight -- the next item's first pieces and samples -- lands before the first row store goes out: behind
                // the stores no wait can tell those requests from the stores)
                if constexpr (SLDS) __builtin_amdgcn_s_waitcnt(0x0F70);
                if constexpr (FOLD)
                    gemm3_epilogue<MB, NX, false, 0, true, H2>(a, acc, xacc, vm, vn, inb, lv, vox, le, cur.tile_m, sraw, lds + 2 * TILEB + wave * TRB, f_row, f_row + FKMAX, asc);
                else
                    gemm3_epilogue<MB, NX, false, 0, false, H2>(a, acc, xacc, vm, vn, inb, lv, vox, le, cur.tile_m, sraw, lds + 2 * TILEB + wave * TRB, nullptr, nullptr, asc);
            }
        }
        FIB_PHASE(g / ntiles - 1, wave, 8);             // epilogue done
        if (!nxt.valid) break;
        cur = nxt; inb = inb_n; vox = vox_n; s_off = s_off_n; qoff = qoff_n;
        nxt = work_at(g / ntiles + 1);
        if constexpr (SLDS) vraw_nxt = (int32_t)book[BOOKV + lane];   // (requested at the top of the item that has just ended)
        else vraw_nxt = vidx_at(nxt);     // (clamped: always a valid address)
        clear(early);
    }
    if constexpr (SLDS) __builtin_amdgcn_s_waitcnt(0x0F70);   // (the last stages' requests -- empty ones -- must not outlive the workgroup)
    FIB_STAMP_END(ONE ? 8 : (FUSE ? 2 : (FOLD ? 3 : 1)), g / ntiles);
    if constexpr (FUSE) {
        for (int off = 32; off >= 1; off >>= 1) { const unsigned oth = (unsigned)__shfl_xor((int)en_run, off); en_run = oth > en_run ? oth : en_run; }
        if (lane == 0 && en_run) atomicMax(&a.maxenc[2], en_run);
    }
}

template <int MB, int NX, int NW, bool FOLD = false, bool FUSE = false, bool H2 = false>
__global__ __launch_bounds__(NW * 64, 2) void odf_gemm3_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[gemm3_lds_bytes<MB, NX, NW, FOLD, FUSE, H2>()];
    gemm3_body<MB, NX, NW, FOLD, FUSE, false, H2>(a, lds);
}

// ---- K5, DSI on sphere_642 with an antipodally symmetric lattice (config 5): dsi.jl:204-258 in ONE launch ------------------------------
// The folded DSI map has gRow0 pdf rows + 321 ODF rows.  They are cut into two M tiles of different shapes: the ODF tile
// (10 blocks + the pole row, rows in the order of sphere642_fused.inc) runs the fused epilogue -- 1 / sum(p) scale, ODF rows out,
// find_peaks! on the accumulators, peak / qa / minimum / bounds of the mean -- exactly as the GQI kernel does, so the ODF is never
// re-read (the separate peak kernel cost 0.86 of 6.9 ms and 3.5 GB); the pdf tile (MBB blocks) writes each folded row to its two
// frames.  Both tiles of a voxel group run on the same XCD at about the same time and read the same samples (one HBM fetch, one
// L2 hit), and fold + clamp + 3-way split run twice per
// voxel instead of three times (three tiles of 6 blocks before).  Of the workgroups of an XCD the first dsi_na take ODF tiles, the
// others pdf tiles, each kind walking the XCD's voxel groups with its own stride: the split follows the two tiles' costs.
template <int MBB, bool H2>
__global__ __launch_bounds__(512, 2) void odf_dsi2_kernel(const GemmArgs a) {
    constexpr int LA_ = gemm3_lds_bytes<10, 1, 8, true, true, H2, true>(), LB_ = gemm3_lds_bytes<MBB, 0, 8, true, false, H2, true>();
    __shared__ __attribute__((aligned(16))) char lds[LA_ > LB_ ? LA_ : LB_];
    const int wslot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    GemmArgs b = a;
    const bool paired = a.pair_flags != nullptr && a.dsi_na <= 32;
    if (wslot < a.dsi_na) {
        b.one_slot = wslot; b.one_stride = a.dsi_na;
        b.pair_role = paired ? 1 : 0;
        gemm3_body<10, 1, 8, true, true, true, H2>(b, lds);
    } else {
        b.one_slot = wslot - a.dsi_na; b.one_stride = nslot - a.dsi_na;
        b.pair_role = paired ? 2 : 0;
        b.At3 = a.At3b;
        b.M = a.nrow0;                                   // the pdf rows only (rows >= M are padding of the tile)
        gemm3_body<MBB, 0, 8, true, false, true, H2>(b, lds);
    }
}


// ---- mask compaction ---------------------------------------------------------------------------------------
// vidx = ascending list of the voxels of every aligned 4-voxel group ("quad") that holds a voxel inside the mask, so
// that every four consecutive list entries are four consecutive, 16-byte aligned voxels (dwordx4 row stores in the
// GEMM epilogue; the GEMM re-tests the mask per voxel and writes zeros for the group's voxels outside it: at most 3
// wasted columns per run end).  tiles = ascending list of the 64-voxel tiles that hold at least one voxel of the
// mask; both counts stay on the device (no host round trip).
// [r4] ONE launch (it was three: per-block counts, one-block scan, ordered write -- plus a fourth for the outputs outside the
// mask): a workgroup draws a chunk (a whole number of 4096-voxel sub-chunks: 4 passes x 16 waves x 64 lanes, so that a wave-pass
// is exactly one tile and a ballot gives both counts; at most 1024 chunks per volume) from a ticket counter, publishes the
// chunk's two counts as ONE 8-byte granule {epoch, tiles, voxels}, adds up the granules of all chunks before it (every thread
// polls one: a single round; a chunk number is drawn before anything is waited for, so every chunk a workgroup waits for
// belongs to a workgroup that is already running -- no assumption on dispatch order or residency), writes its part of the two
// lists and clears the chunk's output voxels outside the mask.  Granules are written and read with agent-scope atomics (the
// data is the flag: no fence); the epoch is the plan's call counter, so nothing has to be cleared between calls.  The
// workgroup with the last chunk writes the totals, clears the peak finder's running maximum and resets the ticket.
constexpr int CB = 4096;                             // voxels per sub-chunk
constexpr int CB_ITERS_MAX = 32;                     // sub-chunks per chunk (2^27 voxels / 4096 / 1024 chunks)
// lanes of the quads (aligned groups of 4 lanes = voxels) in which at least one lane's bit is set
__device__ __forceinline__ unsigned long long quad_expand(unsigned long long b) {
    unsigned long long q = (b | (b >> 1) | (b >> 2) | (b >> 3)) & 0x1111111111111111ull;
    return q | (q << 1) | (q << 2) | (q << 3);
}
// outputs of voxels outside the mask are zero (the reference's output volumes start zero-filled): every output
// row of the GEMM plus the 9 peak components and 3 qa volumes
struct ZeroArgs { float *out0, *out1, *peak[3], *qa[3]; int n0, n1; int64_t nvox, stride; };
struct CompactArgs {
    const uint8_t *mask; int64_t nvox;
    int32_t *vidx, *tiles;
    unsigned long long *state;    // [nchunks] granules: epoch << 32 | tiles of the chunk << 18 | listed voxels of the chunk
    unsigned *ticket;             // [4]: chunk dispenser, the post kernel's arrival counter, -, -
    int32_t *totals;              // [4]: {listed voxels, listed tiles, length of the +Inf list, length of the redo list}
    unsigned *maxenc;             // [4] (may be NULL)
    unsigned epoch;
    int nchunks, iters, zero;     // iters: sub-chunks per chunk; zero: workgroups nchunks.. of the grid clear the outputs outside the mask (z)
    unsigned *clear; int nclear;  // words the call needs zeroed before its next launch (odf_dsi2_kernel's pairing counters)
    unsigned long long *sub_state;   // [sub-chunks] granules: epoch << 32 | voxels of the sub-chunk outside the mask (zero != 0 only)
    ZeroArgs z;
};
typedef __attribute__((address_space(1))) unsigned long long fib_gu64;
__global__ __launch_bounds__(1024) void mask_compact_kernel(const CompactArgs c) {
    __shared__ int cv[CB_ITERS_MAX][64], ct[CB_ITERS_MAX][64];
    __shared__ int s_chunk, s_base[2], s_agg[2], s_ndead[CB_ITERS_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool compacts = (int)blockIdx.x < c.nchunks;        // (the other workgroups only help to clear outputs)
    if (compacts) {
        if (tid == 0) { s_chunk = (int)atomicAdd(c.ticket, 1u); s_base[0] = 0; s_base[1] = 0; }
        if (tid < CB_ITERS_MAX) s_ndead[tid] = 0;
        __syncthreads();
        const int t = s_chunk;
        const int64_t base = (int64_t)t * c.iters * CB;
        for (int it = 0; it < c.iters; it++) {
            int nd = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int64_t vx = base + (int64_t)it * CB + i * 1024 + tid;
                const unsigned long long inr = __ballot(vx < c.nvox);
                const unsigned long long b = __ballot(vx < c.nvox && c.mask[vx] != 0);
                const unsigned long long e = quad_expand(b) & inr;
                nd += __popcll(inr & ~b);
                if (lane == 0) { cv[it][i * 16 + wave] = __popcll(e); ct[it][i * 16 + wave] = b != 0ull; }   // entry = wave-pass in voxel order
            }
            if (c.zero && lane == 0 && nd) atomicAdd(&s_ndead[it], nd);
        }
        __syncthreads();
        // (for the helpers below: how many voxels of each sub-chunk lie outside the mask -- a hint that saves them the look at the mask)
        if (c.zero && tid < c.iters && (int64_t)(t * c.iters + tid) * CB < c.nvox)
            __hip_atomic_store((fib_gu64 *)(c.sub_state + (size_t)t * c.iters + tid), ((unsigned long long)c.epoch << 32) | (unsigned)s_ndead[tid],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == 0) {                                         // exclusive prefix over the chunk's wave-passes, and its two totals
            int runv = 0, runt = 0;
            for (int it = 0; it < c.iters; it++) {
                const int v = cv[it][lane], tl = ct[it][lane];
                int sv = v, stl = tl;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int ov = __shfl_up(sv, off), ot = __shfl_up(stl, off);
                    if (lane >= off) { sv += ov; stl += ot; }
                }
                cv[it][lane] = runv + sv - v; ct[it][lane] = runt + stl - tl;
                runv += __shfl(sv, 63); runt += __shfl(stl, 63);
            }
            if (lane == 0) {
                s_agg[0] = runv; s_agg[1] = runt;
                __hip_atomic_store((fib_gu64 *)(c.state + t), ((unsigned long long)c.epoch << 32) | ((unsigned long long)runt << 18) | (unsigned long long)runv,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        {                                                        // the sum over the chunks before this one: thread i polls chunk i (nchunks <= 1024)
            int pv = 0, pt = 0;
            if (tid < t) {
                unsigned long long g;
                for (;;) {
                    g = __hip_atomic_load((fib_gu64 *)(c.state + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(g >> 32) == c.epoch) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                pv = (int)(g & 0x3ffffull); pt = (int)((g >> 18) & 0x3fffull);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { pv += __shfl_xor(pv, off); pt += __shfl_xor(pt, off); }
            if (lane == 0 && (pv | pt)) { atomicAdd(&s_base[0], pv); atomicAdd(&s_base[1], pt); }
        }
        __syncthreads();
        if (t == c.nchunks - 1 && wave == 0) {                   // every chunk has been drawn: totals, and the state of the next call
            if (lane == 0) {
                c.totals[0] = s_base[0] + s_agg[0]; c.totals[1] = s_base[1] + s_agg[1]; c.totals[2] = 0; c.totals[3] = 0;
                if (c.maxenc) { c.maxenc[0] = 0u; c.maxenc[1] = 0u; c.maxenc[2] = 0u; c.maxenc[3] = 0u; }
                c.ticket[0] = 0u; c.ticket[1] = 0u;
            }
            for (int i = lane; i < c.nclear; i += 64) c.clear[i] = 0u;
        }
        for (int it = 0; it < c.iters; it++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int64_t vx = base + (int64_t)it * CB + i * 1024 + tid;
                const unsigned long long b = __ballot(vx < c.nvox && c.mask[vx] != 0);
                const unsigned long long e = quad_expand(b) & __ballot(vx < c.nvox);
                const int pv = s_base[0] + cv[it][i * 16 + wave], pt = s_base[1] + ct[it][i * 16 + wave];
                if ((e >> lane) & 1ull) c.vidx[pv + __popcll(e & ((1ull << lane) - 1ull))] = (int32_t)vx;
                if (lane == 0 && b) c.tiles[pt] = (int32_t)(vx >> 6);
            }
        if (!c.zero) return;
    }
    // ---- outputs outside the mask: the job of the helper workgroups behind the compacting ones (the compacting workgroups'
    // chain is not lengthened: they join in at the end).  The (span of 8 sub-chunks, group of 16 rows) items are dealt out round-robin: a
    // thread owns 4 consecutive voxels of each sub-chunk and walks the rows, one 16-byte store per row and sub-chunk when all
    // four are to be cleared (scalar stores for mixed groups).  With everything inside the mask an item is eight granule loads.
    const ZeroArgs &z = c.z;
    constexpr int ZS = 8, ZR = 16;                               // an item = 8 consecutive sub-chunks (128 KiB of every row) x 16 rows
    __shared__ int s_nd[ZS], s_known, s_tot[2];
    const int nr = z.n0 + z.n1 + 12, ngrp = (nr + ZR - 1) / ZR;
    const int64_t nsub = (c.nvox + CB - 1) / CB, nspan = (nsub + ZS - 1) / ZS;
    const int hb = (int)blockIdx.x, nh = (int)gridDim.x;        // (the compacting workgroups join in when their lists are written)
    // First the whole picture, from the counts the compacting workgroups publish (bounded wait; they were dispatched first, but
    // nothing depends on that: a helper that does not get every count in time falls through to the per-span path, which counts for
    // itself): nothing outside the mask -> done; more than a quarter of the volume outside -> every output is cleared as a whole,
    // each helper one contiguous slice of each array (pure streaming stores: the fill rate), and the contraction / peak kernels
    // overwrite the voxels inside.
    if (tid < 2) s_tot[tid] = 0;
    __syncthreads();
    {
        long long nd = 0;
        int unknown = 0;
        for (int64_t sidx = tid; sidx < nsub; sidx += 1024) {
            int got = -1;
            for (int spin = 0; spin < 600; spin++) {
                const unsigned long long w = __hip_atomic_load((fib_gu64 *)(c.sub_state + sidx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(w >> 32) == c.epoch) { got = (int)(unsigned)w; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (got < 0) unknown = 1; else nd += got;
        }
        int q = (int)nd;                                         // (<= 2^27 voxels in all)
        for (int off = 32; off >= 1; off >>= 1) { q += __shfl_xor(q, off); unknown |= __shfl_xor(unknown, off); }
        if (lane == 0) { atomicAdd(&s_tot[0], q); if (unknown) atomicOr(&s_tot[1], 1); }
    }
    __syncthreads();
    const bool all_known = s_tot[1] == 0;
    const int64_t total_dead = s_tot[0];
    if (all_known && total_dead == 0) return;
    if (all_known && total_dead * 4 > c.nvox && z.stride == c.nvox && (c.nvox & 3) == 0) {
        auto clear = [&](float *p, int64_t nfl) {
            if (!p || nfl <= 0) return;
            if (reinterpret_cast<uintptr_t>(p) & 15) { for (int64_t i = (int64_t)hb * 1024 + tid; i < nfl; i += (int64_t)nh * 1024) p[i] = 0.0f; return; }
            const int64_t nq = nfl >> 2;
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
            v4f *d = reinterpret_cast<v4f *>(p);
            int64_t q = (int64_t)hb * 4096 + tid;
            for (; q + 3072 < nq; q += (int64_t)nh * 4096) { d[q] = zero4; d[q + 1024] = zero4; d[q + 2048] = zero4; d[q + 3072] = zero4; }
            for (int i = 0; i < 4; i++) if (q + 1024 * i < nq) d[q + 1024 * i] = zero4;
        };
        clear(z.out0, (int64_t)z.n0 * c.nvox);
        clear(z.out1, (int64_t)z.n1 * c.nvox);
        for (int k = 0; k < 3; k++) { clear(z.peak[k], 3 * c.nvox); clear(z.qa[k], c.nvox); }
        return;
    }
    // otherwise span by span; a helper takes a contiguous run of items (the row groups of a span follow each other)
    const int64_t nitem = nspan * ngrp, per = (nitem + nh - 1) / nh;
    const int64_t item_hi = ((int64_t)hb + 1) * per < nitem ? ((int64_t)hb + 1) * per : nitem;
    for (int64_t item = (int64_t)hb * per; item < item_hi; item++) {
        const int64_t span = item / ngrp;
        const int g = (int)(item % ngrp);
        __syncthreads();                                         // (s_nd of the previous item has been read)
        // The compacting workgroups publish every sub-chunk's count of voxels outside the mask; a helper waits for them a bounded
        // time (they were dispatched first, but nothing here depends on that: after the time-out the helper counts for itself)
        if (wave == 0) {
            int nd = -1;
            if (lane < ZS && span * ZS + lane < nsub) {
                for (int spin = 0; spin < 400; spin++) {
                    const unsigned long long w = __hip_atomic_load((fib_gu64 *)(c.sub_state + span * ZS + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(w >> 32) == c.epoch) { nd = (int)(unsigned)w; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
            } else if (lane < ZS) nd = 0;                        // past the end of the volume
            if (lane < ZS) s_nd[lane] = nd < 0 ? 0 : nd;
            const unsigned long long miss = __ballot(lane < ZS && nd < 0);
            if (lane == 0) s_known = miss == 0ull;
        }
        __syncthreads();
        const bool known = s_known != 0;
        unsigned deadb = 0u, inrb = 0u;                          // 4 bits per sub-chunk: my voxels outside the mask / inside the volume
        bool need_bits = !known;
        if (known) {
#pragma unroll
            for (int k = 0; k < ZS; k++) {                       // the mask itself is needed for the sub-chunks that are cleared voxel by voxel only
                const int64_t sb = (span * ZS + k) * CB;
                const int64_t len = c.nvox - sb < CB ? c.nvox - sb : CB;
                need_bits |= s_nd[k] != 0 && !((int64_t)s_nd[k] * 16 > len);
            }
        }
#pragma unroll
        for (int k = 0; k < ZS; k++) {
            const int64_t v0 = (span * ZS + k) * CB + (int64_t)tid * 4;
            int nd = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool in = v0 + i < c.nvox;
                const bool dd = need_bits && in && c.mask[v0 + i] == 0;
                inrb |= (unsigned)in << (4 * k + i); deadb |= (unsigned)dd << (4 * k + i); nd += dd;
            }
            if (!known) {
                for (int off = 32; off >= 1; off >>= 1) nd += __shfl_xor(nd, off);
                if (lane == 0 && nd) atomicAdd(&s_nd[k], nd);
            }
        }
        if (!known) __syncthreads();
        unsigned clr = 0u;                                       // what this thread clears
#pragma unroll
        for (int k = 0; k < ZS; k++) {
            const int64_t sb = (span * ZS + k) * CB;
            const int64_t len = c.nvox - sb < CB ? c.nvox - sb : CB;
            const unsigned m = 0xFu << (4 * k);
            // a sub-chunk with more than 1/16 of its voxels outside is cleared as a whole: ragged runs of 4-byte stores cost more
            // than the lines of the voxels inside (which the contraction kernel overwrites)
            clr |= (s_nd[k] == 0 ? 0u : ((int64_t)s_nd[k] * 16 > len ? inrb : deadb)) & m;
        }
        if (__syncthreads_or(clr != 0u) == 0) continue;
        const int r1 = (g + 1) * ZR < nr ? (g + 1) * ZR : nr;
        for (int r = g * ZR; r < r1; r++) {
            float *row;
            if (r < z.n0) row = z.out0 + (int64_t)r * z.stride;
            else if (r < z.n0 + z.n1) row = z.out1 + (int64_t)(r - z.n0) * z.stride;
            else { const int q = r - z.n0 - z.n1; row = q < 9 ? z.peak[q / 3] + (int64_t)(q % 3) * z.stride : z.qa[q - 9]; }
            const bool al = (reinterpret_cast<uintptr_t>(row) & 15) == 0;
#pragma unroll
            for (int k = 0; k < ZS; k++) {
                const unsigned bk = (clr >> (4 * k)) & 0xFu;
                if (bk == 0u) continue;
                float *d = row + (span * ZS + k) * CB + (int64_t)tid * 4;
                if (bk == 0xFu && al) { *reinterpret_cast<float4 *>(d) = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
#pragma unroll
                for (int i = 0; i < 4; i++) if ((bk >> i) & 1u) d[i] = 0.0f;
            }
        }
    }
}

// Columns of voxels with a +Inf sample, recomputed as a plain f32 fma chain over the frames (the reference's mul!(o, A, s):
// Inf * a = +-Inf, Inf * 0 = NaN, +Inf - Inf = NaN).  G is column-major [M x K].  Almost always an empty list.
struct InfFixArgs { const float *G, *S; float *out; const int32_t *count, *list; int cap, M, K; int64_t stride; };
__global__ __launch_bounds__(256) void odf_inf_fix_kernel(const InfFixArgs f) {
    const int n = f.count[0] < f.cap ? f.count[0] : f.cap;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t vox = f.list[i];
        for (int row = threadIdx.x; row < f.M; row += 256) {
            float o = 0.0f;
            for (int k = 0; k < f.K; k++) o = __builtin_fmaf(f.G[row + (size_t)f.M * k], clamp_sample(f.S[(int64_t)k * f.stride + vox]), o);
            f.out[(int64_t)row * f.stride + vox] = o;
        }
    }
}

// DSI with an antipodally symmetric q-space lattice: cos(2 pi r.q/n) is even in q, so the frames at q and -q
// enter every pdf / odf row with the same coefficient.  t[J] = max(s[q_J],0) + max(s[-q_J],0) halves K, and
// p(r) = p(-r) halves the pdf rows: 2.9x fewer flops for the 515-point scheme.  HBM-bound pre-pass.
__global__ __launch_bounds__(256) void dsi_fold_kernel(const float *__restrict__ S, const uint8_t *__restrict__ mask, const int32_t *__restrict__ fa,
                                                      const int32_t *__restrict__ fb, int nrep, int64_t nvox,
                                                      float *__restrict__ T) {
    const int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vox >= nvox || mask[vox] == 0) return;              // the GEMM only gathers voxels inside the mask
    for (int j = 0; j < nrep; j++) {
        const int a = fa[j], b = fb[j];                     // wave-uniform
        const float x = S[(int64_t)a * nvox + vox];
        float t = x < 0.0f ? 0.0f : x;                      // X .= max.(X, 0), dsi.jl:209 (NaN stays NaN)
        if (b >= 0) { const float y = S[(int64_t)b * nvox + vox]; t += y < 0.0f ? 0.0f : y; }
        T[(int64_t)j * nvox + vox] = t;
    }
}
// same, four consecutive voxels per lane (16-byte loads / stores; nvox % 4 == 0, aligned bases) and four folded frames
// per trip, so that a wave keeps 8 KB of loads in flight: 2.07 -> ~1.6 ms on 140^3 x 515
__global__ __launch_bounds__(256) void dsi_fold4_kernel(const float *__restrict__ S, const uint8_t *__restrict__ mask, const int32_t *__restrict__ fa,
                                                       const int32_t *__restrict__ fb, int nrep, int64_t nvox,
                                                       float *__restrict__ T) {
    const int64_t v0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (v0 >= nvox) return;
    if (*reinterpret_cast<const uint32_t *>(mask + v0) == 0u) return;   // quads with a voxel inside the mask are gathered whole
    auto cl = [](float4 q) { return make_float4(q.x < 0.0f ? 0.0f : q.x, q.y < 0.0f ? 0.0f : q.y, q.z < 0.0f ? 0.0f : q.z, q.w < 0.0f ? 0.0f : q.w); };
    for (int j0 = 0; j0 < nrep; j0 += 4) {
        float4 xa[4], xb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u < nrep ? j0 + u : nrep - 1;
            const int a = fa[j], b = fb[j];                 // wave-uniform
            xa[u] = *reinterpret_cast<const float4 *>(S + (int64_t)a * nvox + v0);
            xb[u] = b >= 0 ? *reinterpret_cast<const float4 *>(S + (int64_t)b * nvox + v0) : make_float4(-1.f, -1.f, -1.f, -1.f);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (j0 + u >= nrep) break;
            const float4 p = cl(xa[u]), q = cl(xb[u]);      // (no partner: clamp(-1) = 0, and t + 0 = t exactly)
            const bool has = fb[j0 + u] >= 0;
            const float4 t = has ? make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w) : p;
            *reinterpret_cast<float4 *>(T + (int64_t)(j0 + u) * nvox + v0) = t;
        }
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

struct PeakArgs {
    const float *odf;         // [nvert][nvox]
    const int32_t *nbr;       // [nvert_even][DEG]: row index of each neighbour, unused slots = sentinel row (32-voxel tiles)
    const int32_t *nbr64;     // same table with sentinel = nvert (64-voxel tiles)
    const float *verts;       // [nvert][3] first-half vertex coordinates (gqi.jl:155)
    float *peak[3];           // [3][nvox] each (or NULL in find-peaks mode)
    float *qa[3];             // [nvox] each
    int32_t *isort_top;       // [3][nvox] (find-peaks mode) or NULL
    int32_t *nvalid;          // [nvox]    (find-peaks mode) or NULL
    unsigned *maxenc;         // [4]: see odfmax_contribute (may be NULL)
    float *mean_hi;           // [nvox] upper bounds of the means (with maxenc)
    int64_t nvox;             // voxels in this launch
    int64_t stride;           // row stride of odf / component stride of the outputs
    int nvert, rows_pad;      // rows_pad = nvert rounded up to 8; sentinel row index = rows_pad
    int vec_ok;               // 1: every tile row is 16-byte aligned (nvox % 4 == 0 and aligned base)
    const int32_t *tiles;     // optional: ascending list of the 64-voxel tiles to scan (mask compaction) and its
    const int32_t *ntl;       // device-side length; tiles not listed keep the zeros zero_dead_kernel wrote
};

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
            const float *g = a.odf + (int64_t)row * a.stride + vox0 + 4 * (lane & 7);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(o + p * 256), 16, 0, 0);
        }
    } else {
        for (int r = wave * 2 + half; r < a.rows_pad; r += PG)
            o[r * PV + j] = (inb && r < a.nvert) ? a.odf[(int64_t)r * a.stride + vox] : 0.0f;
    }
    if (tid < PV) o[a.rows_pad * PV + tid] = __builtin_nanf("");   // sentinel row
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- scan: this wave's vertex pairs -------------------------------------------------------------
    Top3 t;
    top3_clear(t);
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
        for (int k = 0; k < 3; k++) { rec[2 * k] = __uint_as_float((unsigned)t.k[k]); rec[2 * k + 1] = __uint_as_float((unsigned)(t.k[k] >> 32)); }
        rec[6] = __int_as_float(npos); rec[7] = vmin; rec[8] = vsum; rec[9] = hasnan ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (tid >= 64) return;
    float mean = 0.0f;
    const bool owner = half == 0;
    if (owner) {
        for (int gg = 1; gg < PG; gg++) {
            const float *r = mrg + (size_t)(gg * PV + j) * PREC;
#pragma unroll
            for (int k = 0; k < 3; k++)
                top3_insert_key(t, ((unsigned long long)__float_as_uint(r[2 * k + 1]) << 32) | __float_as_uint(r[2 * k]));
            npos += __float_as_int(r[6]);
            vmin = fminf(vmin, r[7]);
            vsum += r[8];
            hasnan |= r[9] != 0.0f;
        }
        if (hasnan) vmin = NAN;                                 // minimum() propagates NaN (gqi.jl:147)
        mean = vsum / (float)a.nvert;                           // mean(odf, dims=4) = sum ./ n, gqi.jl:164
        if (inb) {
            if (a.isort_top) {
#pragma unroll
                for (int k = 0; k < 3; k++) a.isort_top[(int64_t)k * a.stride + vox] = top3_index(t, k);
                a.nvalid[vox] = npos;
            } else {
                const int n = npos < 3 ? npos : 3;              // gqi.jl:151
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float px = 0.0f, py = 0.0f, pz = 0.0f, q = 0.0f;
                    if (k < n) {
                        const int iv = top3_index(t, k);
                        px = a.verts[3 * iv]; py = a.verts[3 * iv + 1]; pz = a.verts[3 * iv + 2];
                        q = o[iv * PV + j] - vmin;              // gqi.jl:157-158
                    }
                    a.peak[k][vox] = px; a.peak[k][a.stride + vox] = py; a.peak[k][2 * a.stride + vox] = pz;
                    a.qa[k][vox] = q;
                }
            }
        }
    }
    if (a.maxenc) odfmax_contribute(a.maxenc, a.mean_hi, vox, owner && inb, mean, vmin, a.nvert);
}

#include "sphere642_scan.inc"
// largest of six values and 0, NaNs ignored (v_max3 returns the other operands): `o[b] >= o[a]` is false for a NaN neighbour
__device__ __forceinline__ float max6_0_f32(float a, float b, float c, float d, float e, float f) {
    float t;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "v"(b), "v"(c));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(d), "v"(e));
    asm("v_max3_f32 %0, %1, %2, 0" : "=v"(t) : "v"(t), "v"(f));
    return t;
}

// ---- v3: 64-voxel tiles, one persistent workgroup of 16 waves per CU -----------------------------------
// Every lane of a wave owns one voxel and the wave walks its share of the vertices, so vertex and neighbour
// indices are wave-uniform (scalar loads, SGPR operands) and an LDS address costs one v_add.  The tile
// (nvert x 64 floats, 82 KB for sphere_642) leaves room for one workgroup per CU only, so the next tile is
// prefetched into registers (6 x 16 B per lane) while the current one is scanned, and written to LDS after
// the barrier ("issue early, write late"): the HBM stream never waits for the LDS-bound scan.
constexpr int P64_W = 16;                    // waves per workgroup
constexpr int P64_T = P64_W * 64;            // threads
constexpr int P64_NI = 6;                    // float4 staging registers per lane -> nvert <= 6*1024/16 = 384

struct Peak64Partial { Top3 t; int npos; float vmin, vsum; bool hasnan; };

__device__ __forceinline__ void p64_store(float *rec, const Peak64Partial &p) {
#pragma unroll
    for (int k = 0; k < 3; k++) { rec[2 * k] = __uint_as_float((unsigned)p.t.k[k]); rec[2 * k + 1] = __uint_as_float((unsigned)(p.t.k[k] >> 32)); }
    rec[6] = __int_as_float(p.npos); rec[7] = p.vmin; rec[8] = p.vsum; rec[9] = p.hasnan ? 1.0f : 0.0f;
}
__device__ __forceinline__ void p64_merge(Peak64Partial &p, const float *r) {
#pragma unroll
    for (int k = 0; k < 3; k++)
        top3_insert_key(p.t, ((unsigned long long)__float_as_uint(r[2 * k + 1]) << 32) | __float_as_uint(r[2 * k]));
    p.npos += __float_as_int(r[6]);
    p.vmin = fminf(p.vmin, r[7]);
    p.vsum += r[8];
    p.hasnan |= r[9] != 0.0f;
}

template <int DEG, bool EXACT>
__global__ __launch_bounds__(P64_T) void odf_peaks64_kernel(const PeakArgs a, int64_t ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *o = smem;                                            // [nvert + 1][64]; row nvert = NaN sentinel
    float *mrg = o + (size_t)(a.nvert + 1) * 64;                // [P64_W][64][PREC]
    int *nbl = reinterpret_cast<int *>(mrg + (size_t)P64_W * 64 * PREC);   // [nvert][DEG] neighbour rows (LDS copy:
                                                                // global loads of the table would sit on the critical path)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = a.nvert * 16;                                // float4 elements per tile
    if (tid < 64) o[a.nvert * 64 + tid] = __builtin_nanf("");
    for (int i = tid; i < a.nvert * DEG; i += P64_T) nbl[i] = a.nbr64[i] * 64;
    float *vl = reinterpret_cast<float *>(nbl + a.nvert * DEG);             // [nvert][3] vertex coordinates
    for (int i = tid; i < a.nvert * 3; i += P64_T) vl[i] = a.verts[i];

    // fast path: whole, 16-byte aligned tiles are prefetched into registers; a ragged last tile (or an
    // unaligned volume) is loaded synchronously with guards when its turn comes.  The staging registers are
    // six named float4s (an array indexed inside conditionals ends up in scratch memory).
    static_assert(P64_NI == 6, "staging is written out for six registers");
    // (a second staging set, two tiles in flight, was measured slower: 1.35 vs 1.24 ms -- register pressure in the scan)
    float4 s0, s1, s2, s3, s4, s5;
    s0 = s1 = s2 = s3 = s4 = s5 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto is_fast = [&](int64_t tile) { return a.vec_ok && tile * 64 + 64 <= a.nvox; };
    const int rbase = tid >> 4;
#define FIB_P64_ROW(i) ((rbase + 64 * (i)) < a.nvert ? (rbase + 64 * (i)) : a.nvert - 1)
#define FIB_P64_FETCH(tile_)                                                                         \
    do {                                                                                             \
        const float *g0_ = a.odf + (tile_) * 64 + (tid & 15) * 4;                                    \
        s0 = *reinterpret_cast<const float4 *>(g0_ + (int64_t)FIB_P64_ROW(0) * a.stride);              \
        s1 = *reinterpret_cast<const float4 *>(g0_ + (int64_t)FIB_P64_ROW(1) * a.stride);              \
        s2 = *reinterpret_cast<const float4 *>(g0_ + (int64_t)FIB_P64_ROW(2) * a.stride);              \
        s3 = *reinterpret_cast<const float4 *>(g0_ + (int64_t)FIB_P64_ROW(3) * a.stride);              \
        s4 = *reinterpret_cast<const float4 *>(g0_ + (int64_t)FIB_P64_ROW(4) * a.stride);              \
        s5 = *reinterpret_cast<const float4 *>(g0_ + (int64_t)FIB_P64_ROW(5) * a.stride);              \
    } while (0)
#define FIB_P64_PUT(i, reg) if (tid + P64_T * (i) < nq) *reinterpret_cast<float4 *>(o + 4 * (tid + P64_T * (i))) = reg

    const int64_t nlist = a.tiles ? (int64_t)a.ntl[0] : ntiles;
    // voxels of unlisted tiles have an all-zero ODF: their mean (0) takes part in odfmax (gqi.jl:164-166)
    auto tile_at = [&](int64_t k) -> int64_t { return k < nlist ? (a.tiles ? (int64_t)a.tiles[k] : k) : -1; };
    int64_t slot = blockIdx.x;
    int64_t tile = tile_at(slot);
    if (tile < 0) return;
    int64_t next = tile_at(slot + gridDim.x);
    bool fast = is_fast(tile);
    if (fast) FIB_P64_FETCH(tile);
    for (;;) {
        // list entry of the tile after next: issued before this tile's staging registers are waited for, so the
        // prefetch of the next tile (below) is never waited on for it
        slot += gridDim.x;
        const int64_t next2 = tile_at(slot + gridDim.x);
        if (fast) {
            FIB_P64_PUT(0, s0); FIB_P64_PUT(1, s1); FIB_P64_PUT(2, s2);
            FIB_P64_PUT(3, s3); FIB_P64_PUT(4, s4); FIB_P64_PUT(5, s5);
        } else {
            for (int e = tid; e < a.nvert * 64; e += P64_T) {
                const int row = e >> 6, c = e & 63;
                const int64_t vx = tile * 64 + c;
                o[e] = vx < a.nvox ? a.odf[(int64_t)row * a.stride + vx] : 0.0f;
            }
        }
        __syncthreads();
        fast = next >= 0 && is_fast(next);
        if (fast) FIB_P64_FETCH(next);                          // in flight during the scan

        // ---- scan this wave's vertices (uniform v) ------------------------------------------------------
        Peak64Partial p;
        top3_clear(p.t);
        p.npos = 0; p.vmin = INFINITY; p.vsum = 0.0f; p.hasnan = false;
        // UNR vertices per iteration: their neighbour-table reads and ODF reads are all issued before the
        // first compare, so each wave keeps ~(1+DEG)*UNR LDS reads in flight instead of a dependent chain
        constexpr int UNR = 4;
        static_assert(DEG % 2 == 0, "neighbour slots come in pairs (v_max3)");
        unsigned gbits = 0;
        int gscan = 0;
        for (int v0 = wave; v0 < a.nvert; v0 += P64_W * UNR) {
            float x[UNR], y[UNR][DEG];
            int vv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int v = v0 + u * P64_W;
                vv[u] = v < a.nvert ? v : a.nvert;              // past the end: sentinel row (never a peak, NaN)
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int *nb = nbl + (vv[u] < a.nvert ? vv[u] : 0) * DEG;   // wave-uniform address: LDS broadcast reads
                x[u] = o[vv[u] * 64 + lane];
#pragma unroll
                for (int d = 0; d < DEG; d++) y[u][d] = o[nb[d] + lane];
            }
            if constexpr (!EXACT) {
                // candidates only (see the specialised scan): flag = !(max(neighbours, 0) >= x), shifted into `bits`
#pragma unroll
                for (int u = 0; u < UNR; u++) {
                    if (vv[u] < a.nvert) {                      // wave-uniform
                        float mx = 0.0f;
#pragma unroll
                        for (int d = 0; d + 1 < DEG; d += 2) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(y[u][d]), "v"(y[u][d + 1]));
                        asm("v_cmp_nge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(gbits) : "v"(mx), "v"(x[u]) : "vcc");
                        asm("v_min_f32 %0, %0, %1" : "+v"(p.vmin) : "v"(x[u]));      // NaN-ignoring; NaN is recovered from vsum
                        p.vsum += x[u];
                        gscan++;
                    }
                }
            } else {
            float pk[UNR];
            bool cand = false;
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                bool killed = false;
#pragma unroll
                for (int d = 0; d < DEG; d++) killed |= (y[u][d] >= x[u]);   // gqi.jl:185-196
                const bool live = vv[u] < a.nvert;              // wave-uniform
                pk[u] = (killed || !live) ? 0.0f : x[u];        // odf_peak
                if (live) {
                    p.vmin = x[u] < p.vmin ? x[u] : p.vmin;     // NaN-ignoring; NaN is recovered from vsum below
                    p.vsum += x[u];
                    cand = true;
                }
            }
            if (__any(cand)) {
#pragma unroll
                for (int u = 0; u < UNR; u++)
                    if (vv[u] < a.nvert) {
                        if (pk[u] > 0.0f) p.npos++;             // gqi.jl:200
                        top3_insert(p.t, pk[u], vv[u]);         // EXACT: the full sortperm order (find_peaks! API)
                    }
            }
            }
        }
        if constexpr (!EXACT) {
            // bit b of `gbits` = the (gscan-1-b)-th vertex this wave scanned = vertex wave + 16*(gscan-1-b)
            while (__any(gbits != 0u)) {
                if (gbits != 0u) {
                    const int b = __ffs((int)gbits) - 1;
                    gbits &= gbits - 1u;
                    const int v = wave + P64_W * (gscan - 1 - b);
                    const float xv = o[v * 64 + lane];
                    if (xv > 0.0f) p.npos++;                                    // gqi.jl:200
                    top3_insert(p.t, xv, v);
                }
            }
        }
        p.hasnan = p.vsum != p.vsum;                            // a NaN amplitude makes the sum NaN
        p64_store(mrg + (size_t)(wave * 64 + lane) * PREC, p);
        __syncthreads();
        // ---- two-level merge: waves 0..3 fold 4 partials each, wave 0 folds those -------------------------
        if (wave < 4) {
            for (int g = 1; g < 4; g++) p64_merge(p, mrg + (size_t)((wave + 4 * g) * 64 + lane) * PREC);
            if (wave > 0) p64_store(mrg + (size_t)(wave * 64 + lane) * PREC, p);
        }
        __syncthreads();
        const int64_t vox = tile * 64 + lane;
        const bool inb = vox < a.nvox;
        if (wave == 0) {
            for (int g = 1; g < 4; g++) p64_merge(p, mrg + (size_t)(g * 64 + lane) * PREC);
            if (p.hasnan) p.vmin = NAN;                         // minimum() propagates NaN (gqi.jl:147)
            const float mean = p.vsum / (float)a.nvert;   // mean(odf, dims=4) = sum ./ n, gqi.jl:164
            p64_store(mrg + (size_t)lane * PREC, p);            // final record of this voxel for the writer waves
            if (a.maxenc) odfmax_contribute(a.maxenc, a.mean_hi, vox, inb, mean, p.vmin, a.nvert);
        }
        __syncthreads();
        // ---- outputs: one wave per output row (9 peak components + 3 qa, or 3 indices + nvalid) -------------
        if (wave < 12 && inb) {
            const float *r = mrg + (size_t)lane * PREC;
            const int npos = __float_as_int(r[6]);
            if (a.isort_top) {
                if (wave < 3) a.isort_top[(int64_t)wave * a.stride + vox] = __float_as_uint(r[2 * wave]) | __float_as_uint(r[2 * wave + 1]) ? (int)~__float_as_uint(r[2 * wave]) : -1;
                else if (wave == 3) a.nvalid[vox] = npos;
            } else {
                const int k = wave < 9 ? wave / 3 : wave - 9;
                const bool have = k < (npos < 3 ? npos : 3);     // gqi.jl:151
                const int iv = (int)~__float_as_uint(r[2 * k]);
                if (wave < 9) {
                    const int c = wave - 3 * k;
                    a.peak[k][(int64_t)c * a.stride + vox] = have ? vl[3 * iv + c] : 0.0f;      // gqi.jl:154-155
                } else {
                    a.qa[k][vox] = have ? o[iv * 64 + lane] - r[7] : 0.0f;                    // gqi.jl:157-158
                }
            }
        }
        if (next < 0) break;
        tile = next;
        next = next2;
        __syncthreads();                                        // wave 0 is done reading o[] before it is overwritten
    }
}

// ---- v4 (default tessellation only): candidates go to per-voxel lists, two barriers per tile ---------------------------
// Same tile and scan as the S642 variant above; what changes is everything after the scan.  A wave no longer keeps a top-3
// per voxel that two merge levels (three barriers, most waves idle) fold together: every candidate (a voxel has a handful)
// is appended to its voxel's list in LDS through an LDS atomic counter, wave 0 then picks the top three of each list and
// wave 1 folds the 16 partial sums / minima, WHILE the other waves already write the next tile into LDS (the keys carry
// the amplitudes, so the finished tile is not needed any more), and the 12 output rows of a tile are written while the
// scan of the next one runs.  A voxel of finite data has at most 107 candidates (no two of the 640 folded triangles'
// vertices are both maxima: sum of degrees <= 640); the list holds 112.  Only a NaN-poisoned voxel can overflow it: the
// tile is then finished before the next one is stored and wave 0 re-derives that voxel's candidates from the tile.
constexpr int PQ_CAP = 112;
__device__ const short fib_s642_nbr_dev[FIB_S642_NVERT][FIB_S642_DEG] = {
#define FIB_S642_ROW(a, b, c, d, e, f) {a, b, c, d, e, f},
    FIB_S642_TABLE(FIB_S642_ROW)
#undef FIB_S642_ROW
};
__global__ __launch_bounds__(P64_T) void odf_peaks642_kernel(const PeakArgs a, int64_t ntiles) {
    constexpr int NV = FIB_S642_NVERT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *o = smem;                                              // [NV + 1][64]; row NV = NaN sentinel
    unsigned long long *list = reinterpret_cast<unsigned long long *>(o + (NV + 1) * 64);   // [PQ_CAP][64] keys, slot-major
    int *cnt = reinterpret_cast<int *>(list + PQ_CAP * 64);       // [64] candidates appended per voxel; [64] = overflow flag
    float *pmin = reinterpret_cast<float *>(cnt + 128);           // [16][64]
    float *psum = pmin + P64_W * 64;                              // [16][64]
    float *fin = psum + P64_W * 64;                               // [2][8][64]: keys (lo,hi) x 3, npos, vmin
    float *vl = fin + 2 * 8 * 64;                                 // [NV][3] vertex coordinates
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = NV * 16;
    if (tid < 64) { o[NV * 64 + tid] = __builtin_nanf(""); cnt[tid] = 0; cnt[64 + tid] = 0; }
    for (int i = tid; i < NV * 3; i += P64_T) vl[i] = a.verts[i];
    float4 s0, s1, s2, s3, s4, s5;
    s0 = s1 = s2 = s3 = s4 = s5 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto is_fast = [&](int64_t tile) { return a.vec_ok && tile * 64 + 64 <= a.nvox; };
    const int rbase = tid >> 4;
    const int64_t nlist = a.tiles ? (int64_t)a.ntl[0] : ntiles;
    auto tile_at = [&](int64_t k) -> int64_t { return k < nlist ? (a.tiles ? (int64_t)a.tiles[k] : k) : -1; };
    auto put_tile = [&](int64_t tile, bool fast) {                // staging registers (or memory) -> o
        if (fast) {
            FIB_P64_PUT(0, s0); FIB_P64_PUT(1, s1); FIB_P64_PUT(2, s2);
            FIB_P64_PUT(3, s3); FIB_P64_PUT(4, s4); FIB_P64_PUT(5, s5);
        } else {
            for (int e = tid; e < NV * 64; e += P64_T) {
                const int row = e >> 6, c = e & 63;
                const int64_t vx = tile * 64 + c;
                o[e] = vx < a.nvox ? a.odf[(int64_t)row * a.stride + vx] : 0.0f;
            }
        }
    };
    int64_t slot = blockIdx.x;
    int64_t tile = tile_at(slot);
    if (tile < 0) return;
    int64_t next = tile_at(slot + gridDim.x);
    bool fast = is_fast(tile);
    if (fast) FIB_P64_FETCH(tile);
    put_tile(tile, fast);
    __syncthreads();
    int par = 0;
    const float *ob = o + lane, *ob1 = o + FIB_S642_BASE1 * 64 + lane;
    for (;;) {
        slot += gridDim.x;
        const int64_t next2 = tile_at(slot + gridDim.x);
        const bool fast_n = next >= 0 && is_fast(next);
        if (fast_n) FIB_P64_FETCH(next);                          // in flight during the scan
        // ---- scan (as odf_peaks64_kernel<.., S642>) ------------------------------------------------------------------
        float vmin = INFINITY, vsum = 0.0f;
        unsigned bits = 0;
        int nscan = 0;
#define FIB_RD(B, R) (((B) == 0 || ((B) == 2 && (R) <= 255)) ? ob[(R) * 64] : ob1[((R) - FIB_S642_BASE1) * 64])
#define FIB_SCAN_ONE(V, B, A0, A1, A2, A3, A4, A5)                                                          \
        if ((V) < FIB_S642_NVERT) {                                                                         \
            const float x = FIB_RD(B, V);                                                                   \
            const float mx = max6_0_f32(FIB_RD(B, A0), FIB_RD(B, A1), FIB_RD(B, A2), FIB_RD(B, A3), FIB_RD(B, A4), FIB_RD(B, A5)); \
            asm("v_cmp_nge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(mx), "v"(x) : "vcc"); \
            asm("v_min_f32 %0, %0, %1" : "+v"(vmin) : "v"(x));                                              \
            vsum += x;                                                                                      \
            nscan++;                                                                                        \
        }
#define FIB_SCAN_G(V0, B0, a0, a1, a2, a3, a4, a5, V1, B1, b0, b1, b2, b3, b4, b5, V2, B2, c0, c1, c2, c3, c4, c5, V3, B3, d0, d1, d2, d3, d4, d5) \
        FIB_SCAN_ONE(V0, B0, a0, a1, a2, a3, a4, a5) FIB_SCAN_ONE(V1, B1, b0, b1, b2, b3, b4, b5)           \
        FIB_SCAN_ONE(V2, B2, c0, c1, c2, c3, c4, c5) FIB_SCAN_ONE(V3, B3, d0, d1, d2, d3, d4, d5)
        switch (wave) {
            case 0: FIB_S642_WAVE0(FIB_SCAN_G) break;
            case 1: FIB_S642_WAVE1(FIB_SCAN_G) break;
            case 2: FIB_S642_WAVE2(FIB_SCAN_G) break;
            case 3: FIB_S642_WAVE3(FIB_SCAN_G) break;
            case 4: FIB_S642_WAVE4(FIB_SCAN_G) break;
            case 5: FIB_S642_WAVE5(FIB_SCAN_G) break;
            case 6: FIB_S642_WAVE6(FIB_SCAN_G) break;
            case 7: FIB_S642_WAVE7(FIB_SCAN_G) break;
            case 8: FIB_S642_WAVE8(FIB_SCAN_G) break;
            case 9: FIB_S642_WAVE9(FIB_SCAN_G) break;
            case 10: FIB_S642_WAVE10(FIB_SCAN_G) break;
            case 11: FIB_S642_WAVE11(FIB_SCAN_G) break;
            case 12: FIB_S642_WAVE12(FIB_SCAN_G) break;
            case 13: FIB_S642_WAVE13(FIB_SCAN_G) break;
            case 14: FIB_S642_WAVE14(FIB_SCAN_G) break;
            default: FIB_S642_WAVE15(FIB_SCAN_G) break;
        }
#undef FIB_SCAN_G
#undef FIB_SCAN_ONE
#undef FIB_RD
        // candidates -> the voxel's list (bit b of `bits` = vertex wave + 16*(nscan-1-b))
        while (__any(bits != 0u)) {
            if (bits != 0u) {
                const int b = __ffs((int)bits) - 1;
                bits &= bits - 1u;
                const int v = wave + P64_W * (nscan - 1 - b);
                const int sl = atomicAdd(&cnt[lane], 1);
                if (sl < PQ_CAP) list[sl * 64 + lane] = peak_key(ob[v * 64], v);
                else cnt[64] = 1;                                 // overflow: NaN-poisoned voxel
            }
        }
        pmin[wave * 64 + lane] = vmin;
        psum[wave * 64 + lane] = vsum;
        __syncthreads();                                          // B2: lists and partials complete
        const bool ovf = cnt[64] != 0;
        const int64_t vox = tile * 64 + lane;
        const bool inb = vox < a.nvox;
        float *fn = fin + par * 8 * 64;
        auto finalize = [&]() {
            if (wave == 0) {                                      // top three of the voxel's candidates, npos (gqi.jl:198-200)
                Top3 t;
                top3_clear(t);
                int npos = 0;
                const int n = cnt[lane];
                if (n <= PQ_CAP) {
                    for (int i = 0; i < n; i++) {
                        const unsigned long long k = list[i * 64 + lane];
                        npos += (unsigned)(k >> 32) != 0xffffffffu;        // candidates are > 0 or NaN
                        top3_insert_key(t, k);
                    }
                } else {                                          // more candidates than the list holds: rescan this voxel
                    for (int v = 0; v < NV; v++) {
                        const float x = ob[v * 64];
                        bool killed = false;
                        for (int d = 0; d < FIB_S642_DEG; d++) { const int u = fib_s642_nbr_dev[v][d]; if (u < NV) killed |= ob[u * 64] >= x; }
                        const float pk = killed ? 0.0f : x;
                        if (pk > 0.0f) npos++;
                        if (!(pk <= 0.0f)) top3_insert(t, pk, v);
                    }
                }
                cnt[lane] = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) { fn[(2 * k) * 64 + lane] = __uint_as_float((unsigned)t.k[k]); fn[(2 * k + 1) * 64 + lane] = __uint_as_float((unsigned)(t.k[k] >> 32)); }
                fn[6 * 64 + lane] = __int_as_float(npos);
            } else if (wave == 1) {                               // partial sums / minima, folded in the order of the two-level merge
                float s[4], m = INFINITY;
                bool hasnan = false;                              // a NaN amplitude makes its wave's partial sum NaN
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    s[w] = psum[w * 64 + lane];
                    hasnan |= s[w] != s[w];
#pragma unroll
                    for (int g = 1; g < 4; g++) { const float q = psum[(w + 4 * g) * 64 + lane]; hasnan |= q != q; s[w] += q; }
                }
#pragma unroll
                for (int w = 0; w < P64_W; w++) m = fminf(m, pmin[w * 64 + lane]);
                const float vs = ((s[0] + s[1]) + s[2]) + s[3];
                if (hasnan) m = NAN;                              // minimum() propagates NaN (gqi.jl:147)
                fn[7 * 64 + lane] = m;
                const float mean = vs / (float)NV;                // mean(odf, dims=4) = sum ./ n, gqi.jl:164
                if (a.maxenc) odfmax_contribute(a.maxenc, a.mean_hi, vox, inb, mean, m, NV);
            }
        };
        if (ovf) {                                                // rare: the tile must survive until wave 0 is done
            finalize();
            __syncthreads();
            if (tid == 0) cnt[64] = 0;
            if (next >= 0) put_tile(next, fast_n);
        } else {
            if (wave < 2) finalize();
            if (next >= 0) put_tile(next, fast_n);
        }
        __syncthreads();                                          // B3: records of this tile and the next tile in LDS are complete
        // ---- outputs of this tile: one wave per output row; the other waves are already scanning the next tile ---------
        if (wave < 12 && inb) {
            const int npos = __float_as_int(fn[6 * 64 + lane]);
            const int k = wave < 9 ? wave / 3 : wave - 9;
            const bool have = k < (npos < 3 ? npos : 3);          // gqi.jl:151
            const unsigned klo = __float_as_uint(fn[(2 * k) * 64 + lane]), khi = __float_as_uint(fn[(2 * k + 1) * 64 + lane]);
            const int iv = (int)~klo;
            if (wave < 9) {
                const int c = wave - 3 * k;
                a.peak[k][(int64_t)c * a.stride + vox] = have ? vl[3 * iv + c] : 0.0f;      // gqi.jl:154-155
            } else {
                a.qa[k][vox] = have ? peak_key_value(khi) - fn[7 * 64 + lane] : 0.0f;       // gqi.jl:157-158
            }
        }
        if (next < 0) break;
        tile = next;
        next = next2;
        par ^= 1;
    }
}

// ---- companions of the fused epilogue (gemm3_epilogue_fused) ------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)k, off), hi = (unsigned)__shfl_xor((int)(unsigned)(k >> 32), off);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        k = o > k ? o : k;
    }
    return k;
}
// [r4] What follows the contraction kernel, in ONE launch (it was five: +Inf repair, redo list, exact odfmax, its finalisation):
//  * voxels the register scan left alone (NaN / Inf columns, candidate-list overflow): one wave per listed voxel reads the
//    stored column and runs find_peaks! + peak / qa extraction with the generic semantics of odf_peaks_kernel (NaN amplitudes
//    lead the sort order, `NaN >= x` kills nothing), and the reference's sequential mean (gqi.jl:164).  An entry with bit 31
//    set is a voxel with a +Inf (or, fp16 pieces, denormal-maximum) sample: its column is first recomputed as a plain f32 fma
//    chain over the frames (the reference's mul!(o, A, s): Inf * a = +-Inf, Inf * 0 = NaN, +Inf - Inf = NaN) and stored;
//  * maximum(mean(odf, dims=4)) (gqi.jl:164) with the reference's arithmetic: the fused epilogue only bounds each voxel's mean
//    (mean_hi) and the maximum (maxenc[2]); every listed voxel whose upper bound reaches the lower bound of the maximum gets
//    the sequential f32 sum over its stored column here (a handful of voxels unless many voxels hold the same ODF).  Voxels on
//    the redo list carry mean_hi = NaN and are never selected, so the two parts do not depend on each other;
//  * the workgroup that arrives last (agent-scope release before the arrival ticket, acquire after it) turns the ordered-uint
//    maximum and the NaN flag into the two floats the caller gets.
struct RedoArgs { const float *odf; int64_t stride; const int32_t *count, *list; int cap; const float *verts; float *peak[3], *qa[3]; unsigned *maxenc; };
struct RefineArgs { const float *odf; int64_t stride, nvox; int nvert; const int32_t *vidx, *nlive; const float *mean_hi; unsigned *maxenc; };
struct PostArgs {
    RedoArgs redo;                    // count == NULL: no redo list (the separate peak kernels have done everything)
    const float *G, *S; float *out; int M, K;   // the matrix, column-major [M x K], the samples and the ODF rows of the flagged entries' recompute
    RefineArgs refine;
    unsigned *arrive;                 // arrival counter (0 at launch, left 0)
    float *odfmax;                    // [2]
    int raw;                          // 0: {maximum (NaN if any mean is NaN), NaN flag}; 1: {maximum of the means that are not NaN or -Inf, NaN flag}
};
__device__ __forceinline__ void redo_voxel(const RedoArgs &a, float *o, int64_t vox, int lane) {
    constexpr int NV = FIB_S642_NVERT;
    Top3 t;
    top3_clear(t);
    int npos = 0;
    float vmin = INFINITY;
    bool hasnan = false;
    for (int v = lane; v < NV; v += 64) {
        const float x = o[v];
        bool killed = false;
#pragma unroll
        for (int d = 0; d < FIB_S642_DEG; d++) killed |= o[fib_s642_nbr_dev[v][d]] >= x;   // gqi.jl:185-196
        const float pk = killed ? 0.0f : x;
        if (pk > 0.0f) npos++;                      // gqi.jl:200
        if (!(pk <= 0.0f)) top3_insert(t, pk, v);
        hasnan |= x != x;
        vmin = x < vmin ? x : vmin;
    }
    Top3 best;
#pragma unroll
    for (int k = 0; k < 3; k++) {                   // keys are unique (vertex index in the low word): one lane pops per round
        const unsigned long long m = wave_max_u64(t.k[0]);
        best.k[k] = m;
        if (m != 0ull && t.k[0] == m) { t.k[0] = t.k[1]; t.k[1] = t.k[2]; t.k[2] = 0ull; }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        npos += __shfl_xor(npos, off);
        const float om = __shfl_xor(vmin, off);
        vmin = om < vmin ? om : vmin;
    }
    if (__any(hasnan)) vmin = __builtin_nanf("");   // minimum() propagates NaN (gqi.jl:147)
    if (lane == 0) {
        float sum = 0.0f;
        for (int v = 0; v < NV; v++) sum += o[v];   // mean(odf, dims=4): sequential over the vertices, then ./ n (gqi.jl:164)
        const float mean = sum / (float)NV;
        if (mean != mean) atomicOr(&a.maxenc[1], 1u); else atomicMax(&a.maxenc[0], enc_ordered(mean));
        const int n3 = npos < 3 ? npos : 3;         // gqi.jl:151
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float px = 0.0f, py = 0.0f, pz = 0.0f, q = 0.0f;
            if (k < n3) {
                const int iv = top3_index(best, k);
                px = a.verts[3 * iv]; py = a.verts[3 * iv + 1]; pz = a.verts[3 * iv + 2];
                q = o[iv] - vmin;
            }
            a.peak[k][vox] = px; a.peak[k][a.stride + vox] = py; a.peak[k][2 * a.stride + vox] = pz;
            a.qa[k][vox] = q;
        }
    }
}
__global__ __launch_bounds__(256) void odf_post_kernel(const PostArgs p) {
    __shared__ float col[4][512];                               // a voxel's column, per wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // ---- the redo list (sphere_642 only: the fused paths) ---------------------------------------------------------------------
    if (p.redo.count) {
        constexpr int NV = FIB_S642_NVERT;
        const RedoArgs &a = p.redo;
        float *o = col[wv];
        const int n = a.count[0] < a.cap ? a.count[0] : a.cap;
        for (int i = blockIdx.x * 4 + wv; i < n; i += gridDim.x * 4) {
            const unsigned ent = (unsigned)a.list[i];
            const int64_t vox = (int64_t)(ent & 0x7fffffffu);
            __builtin_amdgcn_wave_barrier();
            if ((ent >> 31) && p.G) {
                for (int row = lane; row < p.M; row += 64) {
                    float acc = 0.0f;
                    for (int k = 0; k < p.K; k++) acc = __builtin_fmaf(p.G[row + (size_t)p.M * k], clamp_sample(p.S[(int64_t)k * a.stride + vox]), acc);
                    p.out[(int64_t)row * a.stride + vox] = acc;
                    if (row < NV) o[row] = acc;
                }
            } else {
                for (int v = lane; v < NV; v += 64) o[v] = a.odf[(int64_t)v * a.stride + vox];
            }
            if (lane == 0) o[NV] = __builtin_nanf("");      // unused neighbour slots
            __builtin_amdgcn_wave_barrier();
            redo_voxel(a, o, vox, lane);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the exact maximum of the means -----------------------------------------------------------------------------------------
    {
        const RefineArgs &a = p.refine;
        const int nlive = a.nlive[0];
        // voxels of quads that are not listed have an all-zero ODF: their mean (0) takes part in the maximum
        if (blockIdx.x == 0 && threadIdx.x == 0 && nlive < a.nvox) atomicMax(&a.maxenc[0], enc_ordered(0.0f));
        const unsigned lo_e = a.maxenc[2];                      // (final: written by the kernels before this one)
        const float m_lo = lo_e ? dec_ordered(lo_e) : -INFINITY;
        // a thread tests one listed quad (four consecutive list entries = four consecutive, 16-byte aligned voxels): one index load
        // and one 16-byte load of the four upper bounds per trip (a trip costs two dependent load latencies whatever it fetches)
        const int64_t nquad = ((int64_t)nlive + 3) / 4, nround = (nquad + 63) / 64 * 64;
        unsigned ebest = 0u;
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nround; q += (int64_t)gridDim.x * 256) {
            int64_t vox0 = 0;
            float mh[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (q < nquad) {
                vox0 = a.vidx[4 * q];
                if (4 * q + 4 <= nlive && (vox0 & 3) == 0) {
                    const float4 m4 = *reinterpret_cast<const float4 *>(a.mean_hi + vox0);
                    mh[0] = m4.x; mh[1] = m4.y; mh[2] = m4.z; mh[3] = m4.w;
                } else {
                    for (int j = 0; j < 4; j++) if (4 * q + j < nlive) mh[j] = a.mean_hi[a.vidx[4 * q + j]];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool sel = mh[j] >= m_lo;
                const int64_t vox = (4 * q + 4 <= nlive && (vox0 & 3) == 0) ? vox0 + j : (sel ? (int64_t)a.vidx[4 * q + j] : 0);
                // the few selected voxels of this wave, one after the other: the whole wave fetches the column (the loads of a lane that sums
                // its own column one element after the other are 321 dependent round trips), lane 0 adds it up in the reference's order
                unsigned long long todo = __ballot(sel);
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    const int64_t v = __shfl(vox, src);
                    if (a.nvert <= 512) {
                        for (int r = lane; r < a.nvert; r += 64) col[wv][r] = a.odf[(int64_t)r * a.stride + v];
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) {
                            float sum = 0.0f;
                            for (int r = 0; r < a.nvert; r++) sum += col[wv][r];      // mean(odf, dims=4): sequential over the vertices (gqi.jl:164)
                            const unsigned e = enc_ordered(sum / (float)a.nvert);     // finite columns only (the others are on the redo list)
                            ebest = e > ebest ? e : ebest;
                        }
                        __builtin_amdgcn_wave_barrier();
                    } else if (lane == 0) {
                        float sum = 0.0f;
                        for (int r = 0; r < a.nvert; r++) sum += a.odf[(int64_t)r * a.stride + v];
                        const unsigned e = enc_ordered(sum / (float)a.nvert);
                        ebest = e > ebest ? e : ebest;
                    }
                }
            }
        }
        if (lane == 0 && ebest) atomicMax(&a.maxenc[0], ebest);
    }
    // ---- the last workgroup to arrive publishes the result: everything it reads was written by agent-scope atomics, which are
    // performed at the device's coherence point; a workgroup's arrival follows its atomics (vmcnt(0) in every wave, then the
    // barrier, then the ticket), and the last arriver reads with agent-scope atomic loads: no cache has to be written back or dropped
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            const unsigned e0 = __hip_atomic_load(&p.refine.maxenc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned e1 = __hip_atomic_load(&p.refine.maxenc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool nan = e1 != 0;
            const float m = e0 ? dec_ordered(e0) : -INFINITY;
            p.odfmax[0] = (nan && !p.raw) ? NAN : m;             // maximum() propagates NaN
            p.odfmax[1] = nan ? 1.0f : 0.0f;
            __hip_atomic_store(p.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// qa[k] ./= odfmax (gqi.jl:166-168).  pair != 0: odfmax_dev = {maximum of the means that are not NaN, NaN flag} as it comes out of
// the all-reduce of the multi-rank flow: the divisor is NaN when the flag is set, and the first element becomes what
// maximum() returns.
__global__ __launch_bounds__(256) void qa_normalize_kernel(float *q0, float *q1, float *q2, int64_t nvox,
                                                          float *odfmax_dev, float odfmax_val, int pair) {
    float d = odfmax_dev ? odfmax_dev[0] : odfmax_val;
    if (pair) {                                                 // (the flag never changes; the first element may already be NaN: same d)
        d = odfmax_dev[1] > 0.0f ? NAN : d;
        if (blockIdx.x == 0 && threadIdx.x == 0) odfmax_dev[0] = d;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (int64_t)gridDim.x * blockDim.x) {
        q0[i] = q0[i] / d;                                      // qa[ipeak].vol /= odfmax, gqi.jl:167
        q1[i] = q1[i] / d;
        q2[i] = q2[i] / d;
    }
}

// find_peaks!(W) with ALL of its outputs (gqi.jl:180-201), one workgroup per voxel: odf_peak = the amplitudes of the local peaks, 0
// elsewhere (:184-196); isort = sortperm(odf_peak, rev=true), the complete permutation (:198; descending by isless, equal
// values in ascending index order: the rank of vertex v = the number of keys above its key, peak_key); nvalid = count(. > 0)
// (:200).  Not a hot path (gqi_rec / dsi_rec need the first three entries only and take them from the fused scan or the tile
// kernels): it exists so that the reference's function has a complete counterpart behind the C ABI.
__global__ __launch_bounds__(256) void odf_peaks_work_kernel(const float *__restrict__ odf, int64_t stride, int64_t nvox, int nvert, int deg,
                                                             const int32_t *__restrict__ nbr64, float *__restrict__ odf_peak,
                                                             int32_t *__restrict__ isort, int32_t *__restrict__ nvalid) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long pk_keys[];   // [nvert] keys, then [nvert + 1] amplitudes
    float *o = reinterpret_cast<float *>(pk_keys + nvert);
    __shared__ int cnt;
    for (int64_t vox = blockIdx.x; vox < nvox; vox += gridDim.x) {
        __syncthreads();
        for (int v = threadIdx.x; v < nvert; v += blockDim.x) o[v] = odf[(int64_t)v * stride + vox];
        if (threadIdx.x == 0) { o[nvert] = __builtin_nanf(""); cnt = 0; }           // unused neighbour slots: `NaN >= x` is false
        __syncthreads();
        int mine = 0;
        for (int v = threadIdx.x; v < nvert; v += blockDim.x) {
            const float x = o[v];
            bool killed = false;
            for (int d = 0; d < deg; d++) killed |= o[nbr64[(size_t)v * deg + d]] >= x;   // gqi.jl:185-196
            const float pk = killed ? 0.0f : x;
            odf_peak[(int64_t)v * stride + vox] = pk;
            pk_keys[v] = peak_key(pk, v);
            mine += pk > 0.0f;
        }
        if (mine) atomicAdd(&cnt, mine);
        __syncthreads();
        for (int v = threadIdx.x; v < nvert; v += blockDim.x) {
            const unsigned long long k = pk_keys[v];
            int rank = 0;
            for (int u = 0; u < nvert; u++) rank += pk_keys[u] > k;
            isort[(int64_t)rank * stride + vox] = v;
        }
        if (threadIdx.x == 0) nvalid[vox] = cnt;
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
    // what the GEMM actually runs: G [gM x gK] (== A unless the DSI lattice is folded by symmetry)
    int gM = 0, gK = 0, gRow0 = 0;
    std::vector<float> G;
    bool folded = false;
    int fold_span_max = 0, scale_frame_raw = -1;     // fused fold: most frames a 16-sample stage spans on one side; raw frame of sum(p)
    fib::DevBuf<int32_t> foldA, foldB;               // [gK] frames summed into folded sample J; [gRow0] == same tables map pdf rows
    mutable fib::DevBuf<float> folded_dwi;           // [gK x nvox] scratch of the fold pre-pass (grow-only)
    std::vector<float> A;                            // host copy [nrows x nvol] column-major
    fib::DevBuf<float> At, verts;
    fib::DevBuf<uint16_t> At3;                       // split-bf16 image of G (odf_gemm3_kernel), empty in f32-MFMA mode
    fib::DevBuf<float> Aextra;                       // f32 coefficients of the NX extra rows [ntile_m][NX][Kpad]
    fib::DevBuf<float> Gdev;                         // G, column-major [gM x gK] (odf_inf_fix_kernel)
    mutable fib::DevBuf<int32_t> inf_list;           // [nvox] voxels with a +Inf sample (GQI, split-bf16 kernel; grow-only)
    bool split_bf16 = false;
    bool h2 = false;                                 // .. with two fp16 pieces per element (default) instead of three bf16 pieces (FIBERS_ODF_EXACT=1)
    float h2_sa = 1.0f;                              // power of two the matrix is scaled by before it is split into fp16 pieces
    bool fused_shape = false;                        // (GQI, 10 blocks + 1 extra row: the shape the fused scan is generated for)
    bool fused = false;                              // sphere_642 GQI plan: the contraction kernel finds the peaks on its accumulators
    bool dsi2_shape = false, dsi2 = false;           // folded DSI plan on sphere_642: odf_dsi2_kernel (fused ODF tile + pdf tile of MBB blocks)
    int MBB = 0;
    fib::DevBuf<uint16_t> At3b;                      // image of the pdf tile (the ODF tile's image / pole row: At3f / Aextraf)
    mutable fib::DevBuf<unsigned> pair_flags;        // item counters of the ODF-tile workgroups (odf_dsi2_kernel's pairing hint)
    fib::DevBuf<uint16_t> At3f;                      // split-bf16 image with the rows in the order of sphere642_fused.inc
    fib::DevBuf<float> Aextraf;                      // its extra row (the pole of the layout's rotation)
    mutable fib::DevBuf<float> mean_hi;              // [nvox] per-voxel upper bound of the mean (fused path)
    mutable fib::DevBuf<int32_t> redo_list;          // [nvox] voxels left to odf_redo_kernel
    fib::DevBuf<uint32_t> effbits;
    fib::DevBuf<int32_t> nbr, nbr64; // [nvert_even][deg_pad] LDS row of each neighbour (sentinel-padded)
    int deg_pad = 6, rows_pad = 0;
    bool is_s642 = false;                            // neighbour table == the compiled-in sphere_642 table (specialised scan)
    mutable fib::DevBuf<unsigned> maxenc;
    mutable fib::DevBuf<int32_t> live_vox, live_tiles, live_counts;   // mask compaction scratch (grow-only), counts = {voxels, tiles, +Inf voxels}
    mutable fib::DevBuf<unsigned long long> compact_state;   // the chunk granules of mask_compact_kernel [1024]
    mutable fib::DevBuf<unsigned long long> compact_sub;     // .. and its per-sub-chunk hints for the workgroups that clear outputs
    mutable unsigned compact_epoch = 0;              // .. and the call counter they are tagged with
    fib::DevBuf<unsigned> tickets;                   // [4]: chunk dispenser of mask_compact_kernel, arrival counter of odf_post_kernel (both 0 between calls)
    mutable fib::DevBuf<float> odfmax;
};

namespace {


// FIB_ODF_FORMAT_DEFAULT -> what the environment asks for (FIBERS_ODF_GEMM=f32, FIBERS_ODF_EXACT=1), else two fp16 pieces
int resolve_format(int format) {
    if (format != FIB_ODF_FORMAT_DEFAULT) return format;
    const char *e = getenv("FIBERS_ODF_GEMM");
    if (e && (!strcmp(e, "f32") || !strcmp(e, "F32"))) return FIB_ODF_FORMAT_F32;
    const char *ex = getenv("FIBERS_ODF_EXACT");
    return (ex && ex[0] != '0' && ex[0] != 0) ? FIB_ODF_FORMAT_BF16X3 : FIB_ODF_FORMAT_FP16X2;
}

int finish_plan(fib_odf_plan *p, const float *verts, int nverts, const int32_t *faces, int nfaces,
                const std::vector<float> &frame_eff, int format = FIB_ODF_FORMAT_DEFAULT) {
    if (p->G.empty()) { p->G = p->A; p->gM = p->nrows; p->gK = p->nvol; p->gRow0 = p->nrow0; }
    const int M = p->gM, K = p->gK;
    // the operand format of the contraction (header: FIB_ODF_FORMAT_*): two fp16 pieces per f32 operand (default), three exact
    // bf16 pieces, or v_mfma_f32_32x32x2_f32 (a k-ordered f32 fma chain, bit-identical to the oracle's loop)
    format = resolve_format(format);
    p->split_bf16 = format != FIB_ODF_FORMAT_F32;
    // pick (MB, NX) minimising the per-k-step issue cost ntile*(64*MB + 4*NX) cycles (MFMA block = 64, v_fmac = 4)
    int best_cost = INT32_MAX;
    const int nxs[] = {0, 1};                            // (tiles with 2 or 4 VALU rows never won for a shape in use: 13 variants instead of 25)
    for (int mb = p->split_bf16 ? 10 : 11; mb >= 5; mb--)
        for (int nx : nxs) {
            if (nx > 0 && mb > 10) continue;            // register budget
            const int rows = mb * 32 + nx;
            const int nt = (M + rows - 1) / rows;
            const int cost = nt * (64 * mb + 4 * nx);
            if (cost < best_cost) { best_cost = cost; p->MB = mb; p->NX = nx; p->ntile_m = nt; }
        }
    p->Kpad = (K + KT - 1) / KT * KT;
    const int MW = gemm_row_stride(p->MB, p->NX), ROWS = p->MB * 32 + p->NX;
    std::vector<float> At((size_t)p->ntile_m * p->Kpad * MW, 0.0f);
    for (int k = 0; k < K; k++)
        for (int r = 0; r < M; r++) {
            const int tm = r / ROWS, rr = r % ROWS;
            At[((size_t)tm * p->Kpad + k) * MW + rr] = p->G[r + (size_t)M * k];
        }
    // the split-bf16 kernel needs: every frame effective (DSI frames that share a lattice point are not), >= 2 stages,
    // MB <= 10 (LDS: two workgroups per CU), and the extra rows' coefficient table within its 4-KiB LDS slot
    bool any_ineff = false;
    for (int k = 0; k < K; k++) if (frame_eff[k] == 0.0f) any_ineff = true;
    if (p->split_bf16 && (any_ineff || p->Kpad / KT < 2 || p->MB > 10 || (size_t)p->ntile_m * p->NX * p->Kpad > 2048)) p->split_bf16 = false;
    if (p->split_bf16) {
        // piece format: two fp16 pieces of sa * G (sa = the power of two that puts max |G| into [2^8, 2^9): the second piece of an
        // element stays a normal fp16 number down to 2^-20 of the largest, gemm3_body H2), or three exact bf16 pieces
        p->h2 = format == FIB_ODF_FORMAT_FP16X2;
        float gmax = 0.0f;
        for (float v : p->G) if (std::isfinite(v)) gmax = std::max(gmax, std::fabs(v));
        if (p->h2 && !(gmax > 0.0f)) p->h2 = false;
        for (float v : p->G) if (!std::isfinite(v)) p->h2 = false;       // (a matrix with NaN / Inf entries keeps the exact split's behaviour)
        if (p->h2) { int eg; std::frexp(gmax, &eg); p->h2_sa = std::ldexp(1.0f, 9 - eg); }   // gmax = m 2^eg, m in [0.5, 1)
        const int NP = p->h2 ? 2 : 3;
        const int npiece = NP * p->MB, nst = p->Kpad / KT;
        auto f16_rn = [](float f) -> uint16_t { const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; };
        auto f16_f = [](uint16_t u) -> float { _Float16 h; memcpy(&h, &u, 2); return (float)h; };
        auto bf16_rn = [](float f) -> uint16_t {
            uint32_t u; memcpy(&u, &f, 4);
            if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
            u += 0x7fffu + ((u >> 16) & 1u);
            return (uint16_t)(u >> 16);
        };
        auto bf16_f = [](uint16_t h) -> float { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
        auto pieces = [&](float v, uint16_t (&hs)[3]) {
            if (p->h2) {
                const float w = v * p->h2_sa;                          // exact
                hs[0] = f16_rn(w);
                hs[1] = f16_rn(w - f16_f(hs[0]));                       // (the difference is exact)
                hs[2] = 0;
            } else {
                hs[0] = bf16_rn(v);
                const float r1 = v - bf16_f(hs[0]);
                hs[1] = bf16_rn(r1);
                hs[2] = bf16_rn(r1 - bf16_f(hs[1]));
            }
        };
        // image row `row` of the kernel holds row rowmap[row] of G (identity, or the layout of the fused peak scan)
        auto build = [&](const short *rowmap, std::vector<uint16_t> &A3, std::vector<float> &AX) {
            A3.assign((size_t)p->ntile_m * nst * npiece * 512, 0);
            AX.assign((size_t)std::max(1, p->ntile_m * p->NX * p->Kpad), 0.0f);
            for (int tm = 0; tm < p->ntile_m; tm++) {
                for (int t = 0; t < nst; t++) {
                    uint16_t *st = A3.data() + ((size_t)tm * nst + t) * npiece * 512;
                    for (int m = 0; m < p->MB; m++)
                        for (int l = 0; l < 64; l++)
                            for (int j = 0; j < 8; j++) {
                                const int row = tm * ROWS + m * 32 + (l & 31), k = t * KT + 8 * (l >> 5) + j;
                                if (row >= M || k >= K) continue;
                                uint16_t hs[3];
                                pieces(p->G[(rowmap ? rowmap[row] : row) + (size_t)M * k], hs);
                                for (int pc = 0; pc < NP; pc++) st[((size_t)(pc * p->MB + m) * 64 + l) * 8 + j] = hs[pc];
                            }
                }
                for (int x = 0; x < p->NX; x++)
                    for (int k = 0; k < K; k++) {
                        const int row = tm * ROWS + p->MB * 32 + x;
                        if (row < M) AX[((size_t)tm * p->NX + x) * p->Kpad + k] = p->G[(rowmap ? rowmap[row] : row) + (size_t)M * k];
                    }
            }
        };
        std::vector<uint16_t> A3;
        std::vector<float> AX;
        build(nullptr, A3, AX);
        int rc3 = p->At3.alloc(A3.size());
        if (rc3 != FIB_OK) return rc3;
        if (p->gRow0 == 0 && p->scale_frame < 0 && faces) {     // GQI: +Inf samples are repaired after the GEMM
            if ((rc3 = p->Gdev.alloc(p->G.size())) != FIB_OK) return rc3;
            FIB_HIP(hipMemcpy(p->Gdev.p, p->G.data(), p->G.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if ((rc3 = p->Aextra.alloc(AX.size())) != FIB_OK) return rc3;
        FIB_HIP(hipMemcpy(p->At3.p, A3.data(), A3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        FIB_HIP(hipMemcpy(p->Aextra.p, AX.data(), AX.size() * sizeof(float), hipMemcpyHostToDevice));
        // single-tile image of mb blocks (+ nx extra rows) whose image row r holds row rowfn(r) of G (< 0: padding)
        auto build_one = [&](int mb, int nx, auto rowfn, std::vector<uint16_t> &A3o, std::vector<float> &AXo) {
            A3o.assign((size_t)nst * NP * mb * 512, 0);
            AXo.assign((size_t)std::max(1, nx * p->Kpad), 0.0f);
            for (int t = 0; t < nst; t++) {
                uint16_t *st = A3o.data() + (size_t)t * NP * mb * 512;
                for (int m = 0; m < mb; m++)
                    for (int l = 0; l < 64; l++)
                        for (int j = 0; j < 8; j++) {
                            const int gr = rowfn(m * 32 + (l & 31)), k = t * KT + 8 * (l >> 5) + j;
                            if (gr < 0 || k >= K) continue;
                            uint16_t hs[3];
                            pieces(p->G[gr + (size_t)M * k], hs);
                            for (int pc = 0; pc < NP; pc++) st[((size_t)(pc * mb + m) * 64 + l) * 8 + j] = hs[pc];
                        }
            }
            for (int x = 0; x < nx; x++)
                for (int k = 0; k < K; k++) { const int gr = rowfn(mb * 32 + x); if (gr >= 0) AXo[(size_t)x * p->Kpad + k] = p->G[gr + (size_t)M * k]; }
        };
        // folded DSI on sphere_642: ODF tile in the fused scan's row order + pdf tile (odf_dsi2_kernel)
        static const int mbbs[] = {5, 7, 9};
        for (int mbb : mbbs) if (p->MBB == 0 && p->gRow0 <= mbb * 32) p->MBB = mbb;
        p->dsi2_shape = faces && p->folded && p->gRow0 > 0 && p->MBB > 0 && M == p->gRow0 + FQ_NV && p->scale_frame >= 0 && p->Kpad <= 512 && nst >= 2 &&
                        !getenv("FIBERS_DSI_THREE_TILES");
        if (p->dsi2_shape) {
            const int r0 = p->gRow0;
            build_one(10, 1, [&](int r) { return r < FQ_NV ? r0 + fib_f642_pos_vertex[r] : -1; }, A3, AX);
            if ((rc3 = p->At3f.alloc(A3.size())) != FIB_OK || (rc3 = p->Aextraf.alloc(AX.size())) != FIB_OK) return rc3;
            FIB_HIP(hipMemcpy(p->At3f.p, A3.data(), A3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            FIB_HIP(hipMemcpy(p->Aextraf.p, AX.data(), AX.size() * sizeof(float), hipMemcpyHostToDevice));
            build_one(p->MBB, 0, [&](int r) { return r < r0 ? r : -1; }, A3, AX);
            if ((rc3 = p->At3b.alloc(A3.size())) != FIB_OK || (rc3 = p->pair_flags.alloc(8 * 32)) != FIB_OK) return rc3;
            FIB_HIP(hipMemcpy(p->At3b.p, A3.data(), A3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        p->fused_shape = faces && p->gRow0 == 0 && p->scale_frame < 0 && M == FQ_NV && p->MB == 10 && p->NX == 1 && p->ntile_m == 1 && p->Kpad <= 512 &&
                         nst >= 3;                              // (the sample tiles are requested two stages ahead)
        if (p->fused_shape) {                                   // second image in the row order of sphere642_fused.inc
            build(fib_f642_pos_vertex, A3, AX);
            if ((rc3 = p->At3f.alloc(A3.size())) != FIB_OK || (rc3 = p->Aextraf.alloc(AX.size())) != FIB_OK) return rc3;
            FIB_HIP(hipMemcpy(p->At3f.p, A3.data(), A3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            FIB_HIP(hipMemcpy(p->Aextraf.p, AX.data(), AX.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    std::vector<int32_t> nbr32;
    int rc = FIB_OK;
    if (faces) {
        rc = fib::host_neighbours(faces, nfaces, nverts, nbr32, &p->maxdeg);
        if (rc != FIB_OK) return rc;
    } else {                                            // matrix-only plan (fib::matrix_plan_create): no peak finder tables
        p->maxdeg = 0;
    }
    p->deg_pad = p->maxdeg <= 6 ? 6 : (p->maxdeg <= 8 ? 8 : 16);
    p->rows_pad = (p->nvert + 7) / 8 * 8;
    const int nv_even = (p->nvert + 1) / 2 * 2;
    std::vector<int32_t> nbr((size_t)nv_even * p->deg_pad, p->rows_pad);          // default: sentinel row
    for (int v = 0; v < p->nvert; v++)
        for (int d = 0; d < p->maxdeg; d++) {
            const int32_t u = nbr32[(size_t)v * p->maxdeg + d];
            if (u >= 0) nbr[(size_t)v * p->deg_pad + d] = u;
        }
    // the default tessellation has a scan specialised at compile time (sphere642_scan.inc): use it iff the tables agree
    p->is_s642 = p->nvert == FIB_S642_NVERT && p->maxdeg <= FIB_S642_DEG && !getenv("FIBERS_PEAKS_GENERIC");
    for (int v = 0; v < p->nvert && p->is_s642; v++) {
        std::vector<int32_t> mine, ref;
        for (int d = 0; d < p->maxdeg; d++) { const int32_t u = nbr32[(size_t)v * p->maxdeg + d]; if (u >= 0) mine.push_back(u); }
        for (int d = 0; d < FIB_S642_DEG; d++) if (fib_s642_nbr[v][d] < FIB_S642_NVERT) ref.push_back(fib_s642_nbr[v][d]);
        std::sort(mine.begin(), mine.end());
        if (mine != ref) p->is_s642 = false;
    }
    p->fused = p->fused_shape && p->is_s642 && p->At3f.p != nullptr;
    p->dsi2 = p->dsi2_shape && p->is_s642 && p->At3f.p != nullptr && p->At3b.p != nullptr;
    std::vector<int32_t> nbr64(nbr);
    for (auto &u : nbr64) if (u == p->rows_pad) u = p->nvert;
    if ((rc = p->nbr64.alloc(nbr64.size())) != FIB_OK) return rc;
    FIB_HIP(hipMemcpy(p->nbr64.p, nbr64.data(), nbr64.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    std::vector<float> v3((size_t)p->nvert * 3, 0.0f);
    if (verts)
        for (int v = 0; v < p->nvert; v++)
            for (int c = 0; c < 3; c++) v3[3 * v + c] = verts[v + (size_t)nverts * c];
    if ((rc = p->At.alloc(At.size())) != FIB_OK) return rc;
    std::vector<uint32_t> effbits((size_t)p->Kpad / KT, 0u);
    for (int k = 0; k < K; k++) { if (frame_eff[k] != 0.0f) effbits[k / KT] |= 1u << (k % KT); else p->has_ineff = true; }
    if ((rc = p->effbits.alloc(effbits.size())) != FIB_OK) return rc;
    if ((rc = p->verts.alloc(v3.size())) != FIB_OK) return rc;
    if ((rc = p->nbr.alloc(nbr.size())) != FIB_OK) return rc;
    if ((rc = p->maxenc.alloc(4)) != FIB_OK) return rc;
    if ((rc = p->live_counts.alloc(4)) != FIB_OK) return rc;
    if ((rc = p->tickets.alloc(4)) != FIB_OK) return rc;
    FIB_HIP(hipMemset(p->tickets.p, 0, 4 * sizeof(unsigned)));
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
    return fib_gqi_plan_create_fmt(device, bval, bvec, nvol, verts, nverts, faces, nfaces, sigma, FIB_ODF_FORMAT_DEFAULT, plan);
}

extern "C" int fib_odf_default_format(void) try { return resolve_format(FIB_ODF_FORMAT_DEFAULT); } FIB_API_CATCH

extern "C" int fib_odf_plan_format(const fib_odf_plan *plan) try {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan is NULL");
    return !plan->split_bf16 ? FIB_ODF_FORMAT_F32 : (plan->h2 ? FIB_ODF_FORMAT_FP16X2 : FIB_ODF_FORMAT_BF16X3);
} FIB_API_CATCH

extern "C" int fib_gqi_plan_create_fmt(int device, const float *bval, const float *bvec, int nvol,
                                       const float *verts, int nverts, const int32_t *faces, int nfaces,
                                       float sigma, int format, fib_odf_plan **plan) try {
    int rc = check_plan_args(bval, bvec, nvol, verts, nverts, faces, nfaces, plan);
    if (rc != FIB_OK) return rc;
    FIB_CHECK(format >= FIB_ODF_FORMAT_DEFAULT && format <= FIB_ODF_FORMAT_F32, FIB_ERR_INVALID, "unknown operand format %d", format);
    fib::DeviceGuard guard;
    if ((rc = fib::use_device(device)) != FIB_OK) return rc;
    fib_odf_plan *p = new (std::nothrow) fib_odf_plan();
    FIB_CHECK(p != nullptr, FIB_ERR_NOMEM, "out of host memory");
    p->device = device; p->nvol = nvol; p->nvert = nverts / 2; p->nrows = p->nvert; p->nrow0 = 0;
    p->A.resize((size_t)p->nrows * nvol);
    fib::host_gqi_matrix(bval, bvec, nvol, verts, nverts, sigma, p->A.data());
    std::vector<float> eff((size_t)nvol, 1.0f);
    rc = finish_plan(p, verts, nverts, faces, nfaces, eff, format);
    if (rc != FIB_OK) { delete p; return rc; }
    *plan = p;
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fib_dsi_plan_create(int device, const float *bval, const float *bvec, int nvol,
                                   const float *verts, int nverts, const int32_t *faces, int nfaces,
                                   int hann_width, fib_odf_plan **plan) {
    return fib_dsi_plan_create_fmt(device, bval, bvec, nvol, verts, nverts, faces, nfaces, hann_width, FIB_ODF_FORMAT_DEFAULT, plan);
}

extern "C" int fib_dsi_plan_create_fmt(int device, const float *bval, const float *bvec, int nvol,
                                       const float *verts, int nverts, const int32_t *faces, int nfaces,
                                       int hann_width, int format, fib_odf_plan **plan) try {
    int rc = check_plan_args(bval, bvec, nvol, verts, nverts, faces, nfaces, plan);
    if (rc != FIB_OK) return rc;
    FIB_CHECK(format >= FIB_ODF_FORMAT_DEFAULT && format <= FIB_ODF_FORMAT_F32, FIB_ERR_INVALID, "unknown operand format %d", format);
    FIB_CHECK(hann_width >= 0, FIB_ERR_INVALID, "hann_width must be >= 0");
    fib::DeviceGuard guard;
    if ((rc = fib::use_device(device)) != FIB_OK) return rc;
    fib_odf_plan *p = new (std::nothrow) fib_odf_plan();
    FIB_CHECK(p != nullptr, FIB_ERR_NOMEM, "out of host memory");
    p->device = device; p->nvol = nvol; p->nvert = nverts / 2; p->nrows = nvol + p->nvert; p->nrow0 = nvol;
    p->A.resize((size_t)p->nrows * nvol);
    std::vector<int> iq;
    rc = fib::host_dsi_matrix(bval, bvec, nvol, verts, nverts, hann_width, p->A.data(), &p->scale_frame, &p->scale_coef, &iq);
    if (rc != FIB_OK) { delete p; return rc; }
    // a frame overwritten by a later one on the same lattice point never reaches X (dsi.jl:205): its column is 0
    std::vector<float> eff((size_t)nvol, 0.0f);
    for (int j = 0; j < nvol; j++)
        for (int r = 0; r < p->nrows; r++) if (p->A[r + (size_t)p->nrows * j] != 0.0f) { eff[j] = 1.0f; break; }
    if (p->scale_frame < 0) { p->scale_frame = 0; p->scale_coef = 0.0f; }   // no q=0 sample: sum(p) = 0 -> Inf/NaN
    // ---- antipodal folding (see dsi_fold_kernel): every frame needs a partner at -q with an identical column ----
    {
        std::vector<int> partner(nvol, -1);
        bool ok = true;
        for (int j = 0; j < nvol && ok; j++) if (eff[j] == 0.0f) ok = false;
        for (int j = 0; j < nvol && ok; j++) {
            for (int k = 0; k < nvol; k++)
                if (iq[3 * k] == -iq[3 * j] && iq[3 * k + 1] == -iq[3 * j + 1] && iq[3 * k + 2] == -iq[3 * j + 2]) { partner[j] = k; break; }
            if (partner[j] < 0) ok = false;
        }
        for (int j = 0; j < nvol && ok; j++) {
            const float *cj = &p->A[(size_t)p->nrows * j], *ck = &p->A[(size_t)p->nrows * partner[j]];
            for (int r = 0; r < p->nrows && ok; r++) if (cj[r] != ck[r]) ok = false;   // cos is even: columns must be identical
        }
        if (ok) {
            std::vector<int32_t> fa, fb;
            for (int j = 0; j < nvol; j++) if (j <= partner[j]) { fa.push_back(j); fb.push_back(partner[j] == j ? -1 : partner[j]); }
            const int nrep = (int)fa.size();
            p->folded = true; p->gK = nrep; p->gRow0 = nrep; p->gM = nrep + p->nvert;
            p->G.assign((size_t)p->gM * nrep, 0.0f);
            for (int c = 0; c < nrep; c++) {
                const float *col = &p->A[(size_t)p->nrows * fa[c]];
                for (int r = 0; r < nrep; r++) p->G[r + (size_t)p->gM * c] = col[fa[r]];
                for (int v = 0; v < p->nvert; v++) p->G[nrep + v + (size_t)p->gM * c] = col[nvol + v];
                if (fa[c] == p->scale_frame) p->scale_frame = -1000 - c;          // re-index below
            }
            if (p->scale_frame <= -1000) { p->scale_frame = -(p->scale_frame + 1000); p->scale_frame_raw = fa[p->scale_frame]; }
            for (int t0 = 0; t0 < nrep; t0 += KT)
                for (const std::vector<int32_t> *side : {&fa, &fb}) {
                    int lo = 1 << 30, hi = -1;
                    for (int j = t0; j < std::min(nrep, t0 + KT); j++) if ((*side)[j] >= 0) { lo = std::min(lo, (*side)[j]); hi = std::max(hi, (*side)[j]); }
                    if (hi >= 0) p->fold_span_max = std::max(p->fold_span_max, hi - lo + 1);
                }
            if ((rc = p->foldA.alloc(nrep)) != FIB_OK || (rc = p->foldB.alloc(nrep)) != FIB_OK) { delete p; return rc; }
            (void)hipMemcpy(p->foldA.p, fa.data(), nrep * sizeof(int32_t), hipMemcpyHostToDevice);
            (void)hipMemcpy(p->foldB.p, fb.data(), nrep * sizeof(int32_t), hipMemcpyHostToDevice);
            eff.assign((size_t)nrep, 1.0f);
        }
    }
    rc = finish_plan(p, verts, nverts, faces, nfaces, eff, format);
    if (rc != FIB_OK) { delete p; return rc; }
    *plan = p;
    return FIB_OK;
} FIB_API_CATCH

extern "C" void fib_odf_plan_destroy(fib_odf_plan *plan) try {
    if (!plan) return;
    fib::DeviceGuard guard;
    (void)hipSetDevice(plan->device);
    delete plan;
} FIB_API_CATCH_VOID

extern "C" int fib_odf_plan_matrix(const fib_odf_plan *plan, float *A, int *nrows, int *nvol, int *nvert) try {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan is NULL");
    if (nrows) *nrows = plan->nrows;
    if (nvol) *nvol = plan->nvol;
    if (nvert) *nvert = plan->nvert;
    if (A) memcpy(A, plan->A.data(), plan->A.size() * sizeof(float));
    return FIB_OK;
} FIB_API_CATCH

// ------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------
namespace {

constexpr int FOLD_MB_MAX = 10;   // M tiles that the fused-fold variant is built for (room for 16 raw samples per lane and stage)
template <int MB, int NX>
void launch_gemm(const GemmArgs &ga, unsigned grid, hipStream_t st) {
    if (ga.At3) {
        // persistent grid: one 8-wave workgroup per CU (two 4-wave workgroups per CU were measured 6 % slower), a multiple of 8
        // so that blockIdx & 7 is the XCD
        int ncu = 256, dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        GemmArgs g2 = ga;
        const int nw = 8;
        const int64_t items = fib::cdiv(ga.nvox, nw * 32) * ga.ntile_m;
        unsigned pg = (unsigned)std::min<int64_t>((int64_t)ncu * (8 / nw), items);
        pg = (pg + 7) / 8 * 8;
        auto go = [&](auto h2_) {
            constexpr bool H2 = decltype(h2_)::value;
            if (ga.fold) {
                if constexpr (MB <= FOLD_MB_MAX) hipLaunchKernelGGL((odf_gemm3_kernel<MB, NX, 8, true, false, H2>), dim3(pg), dim3(512), 0, st, g2);
                return;
            }
            if constexpr (MB == 10 && NX == 1) {
                if (ga.mean_hi) { hipLaunchKernelGGL((odf_gemm3_kernel<10, 1, 8, false, true, H2>), dim3(pg), dim3(512), 0, st, g2); return; }
            }
            hipLaunchKernelGGL((odf_gemm3_kernel<MB, NX, 8, false, false, H2>), dim3(pg), dim3(512), 0, st, g2);
        };
        if (ga.h2) go(std::true_type{}); else go(std::false_type{});
    }
    else hipLaunchKernelGGL((odf_gemm_kernel<MB, NX>), dim3(grid), dim3(256), 0, st, ga);
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

template <int DEG, bool EXACT>
int launch_peaks64_t(const PeakArgs &pa, size_t smem, int64_t ntiles, unsigned grid, hipStream_t st) {
    FIB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(odf_peaks64_kernel<DEG, EXACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((odf_peaks64_kernel<DEG, EXACT>), dim3(grid), dim3(P64_T), smem, st, pa, ntiles);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
}

int launch_peaks(const fib_odf_plan *plan, const float *odf, int64_t nvox, int64_t stride, float *const peak[3], float *const qa[3],
                 int32_t *isort_top, int32_t *nvalid, bool reduce, hipStream_t st, bool small_tiles = false,
                 const int32_t *tiles = nullptr, const int32_t *ntl = nullptr) {
    PeakArgs pa{};
    pa.tiles = tiles; pa.ntl = ntl;
    pa.odf = odf; pa.nbr = plan->nbr.p; pa.nbr64 = plan->nbr64.p; pa.verts = plan->verts.p;
    for (int k = 0; k < 3; k++) { pa.peak[k] = peak ? peak[k] : nullptr; pa.qa[k] = qa ? qa[k] : nullptr; }
    pa.isort_top = isort_top; pa.nvalid = nvalid;
    pa.maxenc = reduce ? plan->maxenc.p : nullptr;
    pa.mean_hi = reduce ? plan->mean_hi.p : nullptr;
    pa.nvox = nvox; pa.stride = stride; pa.nvert = plan->nvert; pa.rows_pad = plan->rows_pad;
    pa.vec_ok = (stride % 4 == 0 && ((uintptr_t)odf & 15) == 0) ? 1 : 0;
    const size_t smem = peaks_smem(plan);
    FIB_CHECK(smem <= 160 * 1024, FIB_ERR_UNSUPPORTED, "ODF with %d vertices does not fit the peak finder's LDS tile", plan->nvert);
    const unsigned grid = (unsigned)fib::cdiv(nvox, PV);
    const bool exact = isort_top != nullptr;
    fib::ProfScope prof("odf_peaks", st);
    const size_t smem64 = ((size_t)(plan->nvert + 1) * 64 + (size_t)P64_W * 64 * PREC + (size_t)plan->nvert * (plan->deg_pad + 3)) * sizeof(float);
    if (!small_tiles && plan->nvert * 16 <= P64_NI * P64_T && smem64 <= 160 * 1024) {
        int ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, plan->device);
        const int64_t ntiles = fib::cdiv(nvox, 64);
        const unsigned g64 = (unsigned)std::min<int64_t>(ntiles, ncu);
        if (plan->is_s642 && !exact) {
            const size_t smem642 = ((size_t)(FIB_S642_NVERT + 1) * 64 + (size_t)PQ_CAP * 64 * 2 + 128 + 2 * P64_W * 64 + 2 * 8 * 64 +
                                    (size_t)FIB_S642_NVERT * 3) * sizeof(float);
            FIB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(odf_peaks642_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem642));
            hipLaunchKernelGGL(odf_peaks642_kernel, dim3(g64), dim3(P64_T), smem642, st, pa, ntiles);
            FIB_HIP(hipGetLastError());
            return FIB_OK;
        }
        switch (plan->deg_pad) {
            case 6:  return exact ? launch_peaks64_t<6, true>(pa, smem64, ntiles, g64, st) : launch_peaks64_t<6, false>(pa, smem64, ntiles, g64, st);
            case 8:  return exact ? launch_peaks64_t<8, true>(pa, smem64, ntiles, g64, st) : launch_peaks64_t<8, false>(pa, smem64, ntiles, g64, st);
            default: return exact ? launch_peaks64_t<16, true>(pa, smem64, ntiles, g64, st) : launch_peaks64_t<16, false>(pa, smem64, ntiles, g64, st);
        }
    }
    switch (plan->deg_pad) {
        case 6:  return exact ? launch_peaks_t<6, true>(pa, smem, grid, st) : launch_peaks_t<6, false>(pa, smem, grid, st);
        case 8:  return exact ? launch_peaks_t<8, true>(pa, smem, grid, st) : launch_peaks_t<8, false>(pa, smem, grid, st);
        default: return exact ? launch_peaks_t<16, true>(pa, smem, grid, st) : launch_peaks_t<16, false>(pa, smem, grid, st);
    }
}

}  // namespace

namespace {
// the one-launch mask compaction (mask_compact_kernel) on the plan's scratch; z != NULL: also clear those outputs outside the mask
int launch_mask_compact(const fib_odf_plan *plan, const uint8_t *mask, int64_t nvox, unsigned *maxenc, const ZeroArgs *z, hipStream_t st) {
    // at most 256 chunks (one ticket each: returning atomics on one address are served at ~90 per microsecond), all resident at once
    int iters = (int)std::max<int64_t>(1, fib::cdiv(fib::cdiv(nvox, CB), 256));
    if (iters > CB_ITERS_MAX) iters = CB_ITERS_MAX;
    const int nchunks = (int)fib::cdiv(nvox, (int64_t)CB * iters);                      // <= 1024 (2^27 voxels)
    FIB_CHECK(nchunks <= 1024, FIB_ERR_UNSUPPORTED, "volume too large for the mask compaction");
    int rc;
    if ((rc = plan->live_vox.ensure((size_t)nvox)) != FIB_OK) return rc;
    if ((rc = plan->live_tiles.ensure((size_t)fib::cdiv(nvox, 64))) != FIB_OK) return rc;
    if (!plan->compact_state.p) {                                                       // fresh granules carry epoch 0 = "never written"
        if ((rc = plan->compact_state.alloc(1024)) != FIB_OK) return rc;
        FIB_HIP(hipMemsetAsync(plan->compact_state.p, 0, 1024 * sizeof(unsigned long long), st));
    }
    const size_t nsub = (size_t)fib::cdiv(nvox, CB);
    if (z && plan->compact_sub.n < nsub) {
        if ((rc = plan->compact_sub.alloc(nsub)) != FIB_OK) return rc;
        FIB_HIP(hipMemsetAsync(plan->compact_sub.p, 0, nsub * sizeof(unsigned long long), st));
    }
    if (++plan->compact_epoch == 0u) plan->compact_epoch = 1u;
    CompactArgs c{};
    c.mask = mask; c.nvox = nvox; c.vidx = plan->live_vox.p; c.tiles = plan->live_tiles.p; c.state = plan->compact_state.p;
    c.ticket = plan->tickets.p; c.totals = plan->live_counts.p; c.maxenc = maxenc; c.epoch = plan->compact_epoch; c.nchunks = nchunks; c.iters = iters;
    c.zero = z != nullptr;
    if (z) c.z = *z;
    c.clear = plan->pair_flags.p; c.nclear = plan->pair_flags.p ? 8 * 32 : 0;
    c.sub_state = plan->compact_sub.p;
    fib::ProfScope prof("mask_compact", st);
    // with outputs to clear: helpers behind the compacting workgroups, so that a volume that is mostly outside the mask is cleared by the whole chip
    const int grid = nchunks + (z ? 256 : 0);
    hipLaunchKernelGGL(mask_compact_kernel, dim3(grid), dim3(1024), 0, st, c);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
}
}  // namespace

extern "C" int fibd_odf_rec(const fib_odf_plan *plan, const float *dwi, const uint8_t *mask, int64_t nvox,
                            float *pdf, float *odf, float *const peak[3], float *const qa[3],
                            float *odfmax_dev, int flags, void *stream) try {
    FIB_CHECK(plan && dwi && mask && odf && peak && qa, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nvox > 0, FIB_ERR_INVALID, "nvox must be positive");
    FIB_CHECK(nvox <= ((int64_t)1 << 27), FIB_ERR_UNSUPPORTED, "volumes of more than 2^27 voxels are not supported (32-bit lane offsets)");
    FIB_CHECK(plan->nrow0 == 0 || pdf != nullptr, FIB_ERR_INVALID, "DSI plans need a pdf output volume");
    for (int k = 0; k < 3; k++) FIB_CHECK(peak[k] && qa[k], FIB_ERR_INVALID, "NULL peak/qa output volume");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    hipStream_t st = (hipStream_t)stream;
    GemmArgs ga{};
    // launch 1: compact the mask -- voxel list for the GEMM, 64-voxel tile list for the peak finder (counts stay on the device) -- and
    // clear the outputs outside it
    {
        ZeroArgs z{};
        z.out0 = pdf; z.out1 = odf; z.n0 = plan->nrow0; z.n1 = plan->nrows - plan->nrow0; z.nvox = nvox; z.stride = nvox;
        for (int k = 0; k < 3; k++) { z.peak[k] = peak[k]; z.qa[k] = qa[k]; }
        int rcc = launch_mask_compact(plan, mask, nvox, plan->maxenc.p, !(flags & FIB_ODF_PREZEROED) ? &z : nullptr, st);
        if (rcc != FIB_OK) return rcc;
    }
    ga.At = plan->At.p; ga.At3 = (plan->split_bf16 && nvox <= ((int64_t)1 << 26)) ? plan->At3.p : nullptr; ga.S = dwi;
    ga.Aextra = plan->Aextra.p; ga.h2 = plan->h2 ? 1 : 0; ga.h2_inv_sa = 1.0f / plan->h2_sa;
#ifdef FIB_CLOCK_STAMP
    { const char *pi = getenv("FIBERS_PHASE_ITEM"); ga.phase_item = pi ? atoi(pi) : 2; }
    { const char *pw = getenv("FIBERS_PHASE_WG"); ga.phase_wg = pw ? atoi(pw) : 8; }        // (odf_dsi2_kernel: 8 = an ODF-tile workgroup, 136 = a pdf-tile one)
#endif
    ga.vec_ok = (nvox % 4 == 0 && ((uintptr_t)odf & 15) == 0 && (pdf == nullptr || ((uintptr_t)pdf & 15) == 0)) ? 1 : 0; ga.vidx = plan->live_vox.p; ga.nlive = plan->live_counts.p; ga.mask = mask; ga.effbits = plan->effbits.p;
    ga.out0 = pdf; ga.out1 = odf; ga.nvox = nvox;
    if (ga.At3 && plan->Gdev.p) {                        // GQI: voxels with a +Inf sample are listed and recomputed (no cap: the list holds every voxel)
        int rci = plan->inf_list.ensure((size_t)nvox);
        if (rci != FIB_OK) return rci;
        ga.fix_count = plan->live_counts.p + 2; ga.fix_list = plan->inf_list.p; ga.fix_cap = (int)std::min<int64_t>(nvox, INT32_MAX);
    }
    // sphere_642 GQI plans: find_peaks! runs on the contraction kernel's accumulators (gemm3_epilogue_fused)
    const bool sep = (flags & FIB_ODF_SEPARATE_PEAKS) != 0 || getenv("FIBERS_ODF_UNFUSED") != nullptr;
    const bool fuse = plan->fused && ga.At3 != nullptr && ga.vec_ok && !sep;
    {
        int rcm = plan->mean_hi.ensure((size_t)nvox + 4);
        if (rcm != FIB_OK) return rcm;
    }
    auto setup_fused = [&]() -> int {                   // the arguments of gemm3_epilogue_fused
        int rcf = plan->redo_list.ensure((size_t)nvox);
        if (rcf != FIB_OK) return rcf;
        ga.At3 = plan->At3f.p; ga.Aextra = plan->Aextraf.p;
        for (int k = 0; k < 3; k++) { ga.peak[k] = peak[k]; ga.qa[k] = qa[k]; }
        ga.verts = plan->verts.p; ga.maxenc = plan->maxenc.p; ga.mean_hi = plan->mean_hi.p;
        ga.redo_count = plan->live_counts.p + 3; ga.redo_list = plan->redo_list.p; ga.redo_cap = (int)std::min<int64_t>(nvox, INT32_MAX);
        { const char *pa = getenv("FIBERS_ODF_ANTI"); ga.anti = pa ? atoi(pa) : 3; }   // bit 0: anti-phase wave halves, bit 1: s_setprio around the MFMA block (default both; 0 = neither)
        return FIB_OK;
    };
    if (fuse) { int rcf = setup_fused(); if (rcf != FIB_OK) return rcf; }
    ga.K = plan->gK; ga.Kpad = plan->Kpad; ga.M = plan->gM; ga.nrow0 = plan->gRow0; ga.ntile_m = plan->ntile_m;
    const bool fold_ok = plan->folded && ga.At3 != nullptr && plan->Kpad <= FKMAX && plan->scale_frame_raw >= 0 &&
                         (int64_t)plan->fold_span_max * nvox * 4 < (int64_t)0xE0000000ll && !getenv("FIBERS_DSI_UNFUSED");
    const bool fuse_fold = fold_ok && plan->MB <= FOLD_MB_MAX;
    // folded DSI on sphere_642: one launch of odf_dsi2_kernel does contraction, scale, ODF / pdf rows and find_peaks!
    const bool dsi2 = plan->dsi2 && fold_ok && ga.vec_ok && !sep;
    if (dsi2) {
        int rcf = setup_fused();
        if (rcf != FIB_OK) return rcf;
        ga.At3b = plan->At3b.p;
        ga.ntile_m = 2;
    }
    if (fuse_fold || dsi2) {
        ga.fold = 1;
        ga.rowA = plan->foldA.p; ga.rowB = plan->foldB.p;
    } else if (plan->folded) {
        int rcf = plan->folded_dwi.ensure((size_t)plan->gK * nvox);
        if (rcf != FIB_OK) return rcf;
        fib::ProfScope prof("dsi_fold", st);
        if (nvox % 4 == 0 && ((uintptr_t)dwi & 15) == 0 && ((uintptr_t)mask & 3) == 0)
            hipLaunchKernelGGL(dsi_fold4_kernel, dim3((unsigned)fib::cdiv(nvox / 4, 256)), dim3(256), 0, st, dwi, mask, plan->foldA.p, plan->foldB.p,
                               plan->gK, nvox, plan->folded_dwi.p);
        else
            hipLaunchKernelGGL(dsi_fold_kernel, dim3((unsigned)fib::cdiv(nvox, 256)), dim3(256), 0, st, dwi, mask, plan->foldA.p, plan->foldB.p,
                               plan->gK, nvox, plan->folded_dwi.p);
        ga.S = plan->folded_dwi.p;
        ga.rowA = plan->foldA.p; ga.rowB = plan->foldB.p;
    }
    ga.scale_frame = plan->nrow0 > 0 ? ((fuse_fold || dsi2) ? plan->scale_frame_raw : plan->scale_frame) : -1;
    ga.scale_coef = plan->scale_coef;
    ga.stride = nvox;
    ga.has_ineff = plan->has_ineff ? 1 : 0;
    FIB_CHECK(fib::cdiv(nvox, WG_VOX) * plan->ntile_m < ((int64_t)1 << 31), FIB_ERR_UNSUPPORTED, "volume too large for one launch");

    auto run_gemm = [&](GemmArgs g, hipStream_t s) -> int {
        const unsigned grid = (unsigned)(fib::cdiv(g.nvox, WG_VOX) * plan->ntile_m);
        fib::ProfScope prof("odf_gemm", s);
        if (dsi2) {                                      // persistent grid, an even number of workgroups per XCD (ODF tile | pdf tile)
            int ncu = 256;
            (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, plan->device);
            const int64_t items = fib::cdiv(g.nvox, 8 * 32) * 2;
            unsigned pg = (unsigned)std::min<int64_t>((int64_t)ncu, items);
            pg = (pg + 15) / 16 * 16;
            const int nslot = (int)pg / 8;
            // share of the ODF tile's cost per voxel group (tools/dsi_na_sweep.py, 140^3 x 515: 68 us per ODF item, 59.7 per pdf item of 9
            // blocks; ~4.5 us per block): 17 of 32 workgroups per XCD for the 515-point lattice
            const double ca = 68.0, cb = 19.0 + 4.5 * plan->MBB;
            int na = (int)(nslot * ca / (ca + cb) + 0.5);
            // The pdf-tile workgroups follow the ODF-tile ones through the XCD's voxel groups (see gemm3_body: a bounded wait on the
            // partner's item counter), so the samples are fetched from HBM once and the second reader finds them in the XCD's L2 --
            // [r4] for any split, not only 16 | 16 (FIBERS_DSI_NA=<n> sets the split, FIBERS_DSI_PAIR=0 switches the waiting off:
            // tools/dsi_na_sweep.py).
            const char *ena = getenv("FIBERS_DSI_NA");
            if (ena) na = atoi(ena);
            g.dsi_na = std::max(1, std::min(nslot - 1, na));
            const char *epair = getenv("FIBERS_DSI_PAIR");
            if (plan->pair_flags.p && nslot <= 32 && !(epair && epair[0] == '0')) g.pair_flags = plan->pair_flags.p;   // (the counters were cleared by mask_compact_kernel)
            switch (plan->MBB) {
                case 5: if (g.h2) hipLaunchKernelGGL((odf_dsi2_kernel<5, true>), dim3(pg), dim3(512), 0, s, g); else hipLaunchKernelGGL((odf_dsi2_kernel<5, false>), dim3(pg), dim3(512), 0, s, g); break;
                case 7: if (g.h2) hipLaunchKernelGGL((odf_dsi2_kernel<7, true>), dim3(pg), dim3(512), 0, s, g); else hipLaunchKernelGGL((odf_dsi2_kernel<7, false>), dim3(pg), dim3(512), 0, s, g); break;
                default: if (g.h2) hipLaunchKernelGGL((odf_dsi2_kernel<9, true>), dim3(pg), dim3(512), 0, s, g); else hipLaunchKernelGGL((odf_dsi2_kernel<9, false>), dim3(pg), dim3(512), 0, s, g); break;
            }
            FIB_HIP(hipGetLastError());
            return FIB_OK;
        }
#define FIB_GEMM_CASE(MBV, NXV) if (plan->MB == MBV && plan->NX == NXV) { launch_gemm<MBV, NXV>(g, grid, s); launched = true; }
        bool launched = false;
        FIB_GEMM_CASE(5, 0) FIB_GEMM_CASE(6, 0) FIB_GEMM_CASE(7, 0) FIB_GEMM_CASE(8, 0) FIB_GEMM_CASE(9, 0) FIB_GEMM_CASE(10, 0) FIB_GEMM_CASE(11, 0)
        FIB_GEMM_CASE(5, 1) FIB_GEMM_CASE(6, 1) FIB_GEMM_CASE(7, 1) FIB_GEMM_CASE(8, 1) FIB_GEMM_CASE(9, 1) FIB_GEMM_CASE(10, 1)
#undef FIB_GEMM_CASE
        if (!launched) return fib::fail(FIB_ERR_INVALID, "internal: no GEMM variant for MB=%d NX=%d", plan->MB, plan->NX);
        FIB_HIP(hipGetLastError());
        return FIB_OK;
    };

    int rc = run_gemm(ga, st);                           // launch 2
    if (rc != FIB_OK) return rc;
    const bool fused_scan = fuse || dsi2;
    if (!fused_scan) {                                  // separate peak finder on the stored ODF (other tessellations, unaligned pieces, FIB_ODF_SEPARATE_PEAKS)
        if (ga.fix_list) {
            InfFixArgs fx{plan->Gdev.p, dwi, odf, ga.fix_count, ga.fix_list, ga.fix_cap, plan->gM, plan->gK, nvox};
            hipLaunchKernelGGL(odf_inf_fix_kernel, dim3(64), dim3(256), 0, st, fx);
        }
        rc = launch_peaks(plan, odf, nvox, nvox, peak, qa, nullptr, nullptr, true, st, false, plan->live_tiles.p, plan->live_counts.p + 1);
        if (rc != FIB_OK) return rc;
    }
    float *om = odfmax_dev ? odfmax_dev : plan->odfmax.p;
    {                                                   // launch 3: redo list (+ column repair), exact odfmax, its two floats
        fib::ProfScope prof("odf_post", st);
        PostArgs pa{};
        pa.redo = RedoArgs{odf, nvox, fused_scan ? ga.redo_count : nullptr, ga.redo_list, ga.redo_cap, plan->verts.p, {peak[0], peak[1], peak[2]}, {qa[0], qa[1], qa[2]}, plan->maxenc.p};
        pa.G = (fused_scan && ga.fix_list) ? plan->Gdev.p : nullptr; pa.S = dwi; pa.out = odf; pa.M = plan->gM; pa.K = plan->gK;
        pa.refine = RefineArgs{odf, nvox, nvox, plan->nvert, plan->live_vox.p, plan->live_counts.p, plan->mean_hi.p, plan->maxenc.p};
        pa.arrive = plan->tickets.p + 1; pa.odfmax = om; pa.raw = (flags & FIB_ODF_RAW_ODFMAX) ? 1 : 0;
        hipLaunchKernelGGL(odf_post_kernel, dim3(384), dim3(256), 0, st, pa);
        FIB_HIP(hipGetLastError());
    }
    if (flags & FIB_ODF_NORMALIZE) {
        fib::ProfScope prof("qa_normalize", st);
        hipLaunchKernelGGL(qa_normalize_kernel, dim3(2048), dim3(256), 0, st, qa[0], qa[1], qa[2], nvox, om, 0.0f, (flags & FIB_ODF_RAW_ODFMAX) ? 1 : 0);   // launch 4
        FIB_HIP(hipGetLastError());
    }
    return FIB_OK;
} FIB_API_CATCH

// ---- plain use of the contraction kernels: O[M x n] = A[M x K] * max(S[K x n], 0) (row N4, RUMBA-SD) -----------------
int fib::matrix_plan_create(int device, const float *A, int nrows, int ncols, fib_odf_plan **plan) {
    FIB_CHECK(A && plan && nrows > 0 && ncols > 0, FIB_ERR_INVALID, "invalid matrix plan arguments");
    *plan = nullptr;
    fib::DeviceGuard guard;
    int rc = fib::use_device(device);
    if (rc != FIB_OK) return rc;
    fib_odf_plan *p = new (std::nothrow) fib_odf_plan();
    FIB_CHECK(p != nullptr, FIB_ERR_NOMEM, "out of host memory");
    p->device = device; p->nvol = ncols; p->nvert = 2; p->nrows = nrows; p->nrow0 = 0;
    p->A.assign(A, A + (size_t)nrows * ncols);
    std::vector<float> eff((size_t)ncols, 1.0f);
    rc = finish_plan(p, nullptr, 4, nullptr, 0, eff);
    if (rc != FIB_OK) { delete p; return rc; }
    *plan = p;
    return FIB_OK;
}

int fib::matrix_plan_run(const fib_odf_plan *plan, const float *S, const uint8_t *ones, int64_t n, float *out, bool recompact, void *stream) {
    FIB_CHECK(plan && S && ones && out && n > 0, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(n <= ((int64_t)1 << 26), FIB_ERR_UNSUPPORTED, "more than 2^26 columns");
    hipStream_t st = (hipStream_t)stream;
    if (recompact) {                                    // the column list of an all-ones mask: kept in the plan between calls
        int rcc = launch_mask_compact(plan, ones, n, nullptr, nullptr, st);
        if (rcc != FIB_OK) return rcc;
    }
    GemmArgs ga{};
    ga.At = plan->At.p; ga.At3 = plan->split_bf16 ? plan->At3.p : nullptr; ga.S = S;
    ga.Aextra = plan->Aextra.p; ga.h2 = plan->h2 ? 1 : 0; ga.h2_inv_sa = 1.0f / plan->h2_sa;
    ga.vec_ok = (n % 4 == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    ga.vidx = plan->live_vox.p; ga.nlive = plan->live_counts.p; ga.mask = ones; ga.effbits = plan->effbits.p;
    ga.out0 = nullptr; ga.out1 = out; ga.nvox = n;
    ga.K = plan->gK; ga.Kpad = plan->Kpad; ga.M = plan->gM; ga.nrow0 = 0; ga.ntile_m = plan->ntile_m;
    ga.scale_frame = -1; ga.scale_coef = 0.0f; ga.stride = n; ga.has_ineff = 0;
    const unsigned grid = (unsigned)(fib::cdiv(n, WG_VOX) * plan->ntile_m);
    fib::ProfScope prof("matrix_gemm", st);
#define FIB_GEMM_CASE(MBV, NXV) if (plan->MB == MBV && plan->NX == NXV) { launch_gemm<MBV, NXV>(ga, grid, st); launched = true; }
    bool launched = false;
    FIB_GEMM_CASE(5, 0) FIB_GEMM_CASE(6, 0) FIB_GEMM_CASE(7, 0) FIB_GEMM_CASE(8, 0) FIB_GEMM_CASE(9, 0) FIB_GEMM_CASE(10, 0) FIB_GEMM_CASE(11, 0)
    FIB_GEMM_CASE(5, 1) FIB_GEMM_CASE(6, 1) FIB_GEMM_CASE(7, 1) FIB_GEMM_CASE(8, 1) FIB_GEMM_CASE(9, 1) FIB_GEMM_CASE(10, 1)
#undef FIB_GEMM_CASE
    if (!launched) return fib::fail(FIB_ERR_INVALID, "internal: no GEMM variant for MB=%d NX=%d", plan->MB, plan->NX);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
}

extern "C" int fibd_qa_normalize(float *const qa[3], int64_t nvox, float odfmax, void *stream) try {
    FIB_CHECK(qa && qa[0] && qa[1] && qa[2] && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    hipLaunchKernelGGL(qa_normalize_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, qa[0], qa[1], qa[2], nvox,
                       (float *)nullptr, odfmax, 0);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fibd_qa_normalize_dev(float *const qa[3], int64_t nvox, const float *odfmax_dev, void *stream) try {
    FIB_CHECK(qa && qa[0] && qa[1] && qa[2] && odfmax_dev && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    hipLaunchKernelGGL(qa_normalize_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, qa[0], qa[1], qa[2], nvox, const_cast<float *>(odfmax_dev), 0.0f, 0);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fibd_qa_normalize_pair(float *const qa[3], int64_t nvox, float *odfmax_pair_dev, void *stream) try {
    FIB_CHECK(qa && qa[0] && qa[1] && qa[2] && odfmax_pair_dev && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    hipLaunchKernelGGL(qa_normalize_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, qa[0], qa[1], qa[2], nvox, odfmax_pair_dev, 0.0f, 1);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fibd_find_peaks(const fib_odf_plan *plan, const float *odf, int64_t nvox,
                               int32_t *isort_top, int32_t *nvalid, void *stream) try {
    FIB_CHECK(plan && odf && isort_top && nvalid && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    return launch_peaks(plan, odf, nvox, nvox, nullptr, nullptr, isort_top, nvalid, false, (hipStream_t)stream);
} FIB_API_CATCH

#ifdef FIB_CLOCK_STAMP
// diagnostic build: {cycles, ticks, kernel id, items} of the first `cap` workgroups of the last contraction launch
extern "C" int fib_debug_clock_stamps(unsigned long long *out, int cap) try {
    FIB_CHECK(out && cap > 0 && cap <= 2048, FIB_ERR_INVALID, "invalid stamp buffer");
    FIB_HIP(hipDeviceSynchronize());
    FIB_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(fib_clock_stamps), (size_t)cap * 4 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost));
    return FIB_OK;
} FIB_API_CATCH
extern "C" int fib_debug_phase_stamps(unsigned long long *out) try {      // [2][256]: (s_memtime << 8) | mark id
    FIB_CHECK(out, FIB_ERR_INVALID, "invalid stamp buffer");
    FIB_HIP(hipDeviceSynchronize());
    FIB_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(fib_phase_stamps), sizeof(unsigned long long) * 512, 0, hipMemcpyDeviceToHost));
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fib_debug_clock_clear(void) try {
    static unsigned long long zeros[2048][4];
    FIB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(fib_clock_stamps), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice));
    return FIB_OK;
} FIB_API_CATCH
#endif

extern "C" int fibd_find_peaks_work(const fib_odf_plan *plan, const float *odf, int64_t nvox,
                                    float *odf_peak, int32_t *isort, int32_t *nvalid, void *stream) try {
    FIB_CHECK(plan && odf && odf_peak && isort && nvalid && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(plan->nbr64.p != nullptr && plan->maxdeg > 0, FIB_ERR_INVALID, "the plan has no tessellation");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    const size_t smem = (size_t)plan->nvert * sizeof(unsigned long long) + (size_t)(plan->nvert + 1) * sizeof(float);
    FIB_CHECK(smem <= 64 * 1024, FIB_ERR_UNSUPPORTED, "ODF with %d vertices does not fit the peak finder's LDS", plan->nvert);
    const unsigned grid = (unsigned)std::min<int64_t>(nvox, 8192);
    hipLaunchKernelGGL(odf_peaks_work_kernel, dim3(grid), dim3(256), smem, (hipStream_t)stream, odf, nvox, nvox, plan->nvert, plan->deg_pad,
                       plan->nbr64.p, odf_peak, isort, nvalid);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH
