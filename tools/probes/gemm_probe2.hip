// gemm_probe.hip — bisects the cost of the pieces of odf_gemm_kernel (generated from odf.hip by tools/probes/make_gemm_probe.py)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KT = 16;        // frames per LDS stage
constexpr int WG_VOX = 128;   // voxels per workgroup (4 waves x 32)

struct GemmArgs {
    const float *At;          // [ntile_m][Kpad][MB*32]  K-major tiles, zero padded
    const float *S;           // [K][nvox] planar DWI
    const uint8_t *mask;      // [nvox]
    const uint32_t *effbits;  // [Kpad/KT] bit j of word t: frame t*KT+j exists and takes part in the "any positive sample" test
    float *out0;              // rows [0, nrow0)        (DSI: pdf)
    float *out1;              // rows [nrow0, M)        (odf)
    int64_t nvox;
    int K, Kpad, M, nrow0, ntile_m;
    int scale_frame;          // DSI: frame whose clamped sample times scale_coef is sum(p); -1: no scaling
    float scale_coef;
};

// scheduling hint: spread one k-step's fragment reads (ds_read2_b32 = 2 fragments) between the previous
// k-step's MFMAs instead of "read, wait, 2 MFMA" chains (hipcc otherwise minimises live registers)
template <int MB>
__device__ __forceinline__ void interleave_ds_mfma() {
#ifdef NO_SCHED
    return;
#endif
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

template <int MB, bool BL, bool EP, bool CL>
__global__ __launch_bounds__(256, 2) void odf_gemm_kernel(const GemmArgs a) {
    constexpr int MW = MB * 32;
    constexpr int TILE = KT * MW;                       // floats per stage
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int tile_m = blockIdx.x % a.ntile_m;
    const int64_t tile_n = blockIdx.x / a.ntile_m;
    const int64_t vox = tile_n * WG_VOX + wave * 32 + col;
    const bool inb = vox < a.nvox;
    const float *Sp = a.S + (inb ? vox : 0);
    const float *Atile = a.At + (size_t)tile_m * a.Kpad * MW;
    const int ntiles = a.Kpad / KT;

    f32x16 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;

    // one stage = TILE*4 bytes contiguous in global memory; each wave-instruction moves 1 KiB
    constexpr int NPIECE = TILE * 4 / 1024;             // MB*32*16*4/1024 = 2*MB
    auto stage_A = [&](int t, int buf) {
        const char *g = reinterpret_cast<const char *>(Atile + (size_t)t * TILE);
        char *l = reinterpret_cast<char *>(lds + buf * TILE);
        for (int p = wave; p < NPIECE; p += 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + p * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(l + p * 1024), 16, 0, 0);
    };
    float bcur[KT / 2], bnext[KT / 2];
    bool anypos = false;
    // B operand: unconditional loads (clamped frame index) so that all KT/2 loads of a stage are in flight at once
    auto load_B = [&](int t, float (&b)[KT / 2]) {
#pragma unroll
        for (int kk = 0; kk < KT / 2; kk++) {
            const int k = t * KT + 2 * kk + kh;
            const int kc = k < a.K ? k : a.K - 1;
            b[kk] = BL ? Sp[(int64_t)kc * a.nvox] : 1.0f + kc;
        }
    };
    auto clamp_B = [&](int t, float (&b)[KT / 2]) {
        if (!CL) return;
        const uint32_t eff = a.effbits[t];                  // wave-uniform: one scalar load per stage
#pragma unroll
        for (int kk = 0; kk < KT / 2; kk++) {
            const int k = t * KT + 2 * kk + kh;
            const bool live = inb && k < a.K;
            const float s = live ? b[kk] : 0.0f;
            if (live && !(s <= 0.0f) && ((eff >> (2 * kk + kh)) & 1u)) anypos = true;   // positive or NaN (gqi.jl:142, dsi.jl:207)
            b[kk] = s < 0.0f ? 0.0f : s;                                                 // gqi.jl:140, dsi.jl:209
        }
    };
    // A fragments of one k-step: MB conflict-free ds_read_b32 (lane -> row col of block m, frame kh)
    auto load_A = [&](const float *L, int kk, float (&af)[MB]) {
#pragma unroll
        for (int m = 0; m < MB; m++) af[m] = L[2 * kk * MW + m * 32];
    };

    stage_A(0, 0);
    load_B(0, bcur);
    clamp_B(0, bcur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ntiles; t++) {
        const int cur = t & 1;
        if (t + 1 < ntiles) {
            stage_A(t + 1, cur ^ 1);
            load_B(t + 1, bnext);
        }
        const float *L = lds + cur * TILE + kh * MW + col;
        // software pipeline over the k-steps: fragments of step kk+1 are read while step kk's MFMAs issue
        float a0[MB], a1[MB];
        load_A(L, 0, a0);
#ifndef NO_SCHED
        __builtin_amdgcn_sched_group_barrier(0x100, (MB + 1) / 2, 0);
#endif
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
        if (t + 1 < ntiles) {
            clamp_B(t + 1, bnext);
#pragma unroll
            for (int kk = 0; kk < KT / 2; kk++) bcur[kk] = bnext[kk];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stage's direct-to-LDS loads have landed
        __syncthreads();
    }

    // the two k-halves of a voxel live in lanes l and l^32
    const bool valid_half = anypos;
    const bool other = __shfl_xor((int)valid_half, 32) != 0;
    bool valid = inb && (valid_half || other) && a.mask[inb ? vox : 0] != 0;
    float scale = 1.0f;
    if (a.scale_frame >= 0 && inb) {
        float s = Sp[(int64_t)a.scale_frame * a.nvox];
        s = s < 0.0f ? 0.0f : s;
        scale = 1.0f / (a.scale_coef * s);              // p ./ sum(p), dsi.jl:225 (0 -> Inf/NaN like the reference)
    }
    if (!inb) return;
    const bool do_scale = a.scale_frame >= 0;
#pragma unroll
    for (int m = 0; m < MB; m++) {
        const int row0 = tile_m * MW + m * 32;               // wave-uniform
        if (row0 >= a.M) break;
        const bool whole = row0 + 32 <= a.M && (row0 >= a.nrow0 || row0 + 32 <= a.nrow0);   // uniform fast path
        float *base = row0 >= a.nrow0 ? a.out1 + (int64_t)(row0 - a.nrow0 + 4 * kh) * a.nvox + vox
                                      : a.out0 + (int64_t)(row0 + 4 * kh) * a.nvox + vox;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int dr = (r & 3) + 8 * (r >> 2);          // row within the block, before the lane-half offset
            float v = valid ? acc[m][r] : 0.0f;
            if (do_scale) v = valid ? v * scale : 0.0f;
            if (!EP && v != 123.456f) continue;
            if (whole) {
                base[(int64_t)dr * a.nvox] = v;
            } else {
                const int row = row0 + dr + 4 * kh;
                if (row >= a.M) continue;
                if (row < a.nrow0) a.out0[(int64_t)row * a.nvox + vox] = v;
                else               a.out1[(int64_t)(row - a.nrow0) * a.nvox + vox] = v;
            }
        }
    }
}


}
template <bool BL, bool EP, bool CL>
void run(const char *name, GemmArgs ga, unsigned grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((odf_gemm_kernel<11, BL, EP, CL>), dim3(grid), dim3(256), 0, 0, ga);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms;
    }
    printf("%-40s %8.3f ms  %7.1f TFLOP/s (algorithmic)\n", name, best, 2.0 * 321 * 270 * (double)ga.nvox / best / 1e9);
}
int main() {
    const int64_t nvox = 140 * 140 * 140; const int K = 270, Kpad = 272, M = 321, MW = 352;
    float *At, *S, *out; uint8_t *mask; uint32_t *eff;
    hipMalloc(&At, sizeof(float) * Kpad * MW); hipMalloc(&S, sizeof(float) * K * nvox); hipMalloc(&out, sizeof(float) * M * nvox);
    hipMalloc(&mask, nvox); hipMalloc(&eff, 4 * 17);
    hipMemset(mask, 1, nvox);
    std::vector<float> h((size_t)Kpad * MW, 0.25f); hipMemcpy(At, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> e(17, 0xffffu); hipMemcpy(eff, e.data(), 68, hipMemcpyHostToDevice);
    std::vector<float> hs((size_t)nvox); for (size_t i = 0; i < hs.size(); i++) hs[i] = 1.0f + (i % 97) * 0.01f;
    for (int k = 0; k < K; k++) hipMemcpy(S + (size_t)k * nvox, hs.data(), nvox * 4, hipMemcpyHostToDevice);
    GemmArgs ga{}; ga.At = At; ga.S = S; ga.mask = mask; ga.effbits = eff; ga.out0 = nullptr; ga.out1 = out; ga.nvox = nvox;
    ga.K = K; ga.Kpad = Kpad; ga.M = M; ga.nrow0 = 0; ga.ntile_m = 1; ga.scale_frame = -1; ga.scale_coef = 0;
    const unsigned grid = (unsigned)((nvox + 127) / 128);
    run<true, true, true>("full kernel", ga, grid);
    run<true, false, true>("no epilogue stores", ga, grid);
    run<false, true, true>("no B loads", ga, grid);
    run<false, false, true>("no B loads, no stores", ga, grid);
    run<false, false, false>("no B loads, no stores, no clamp", ga, grid);
    run<true, true, false>("B loads + stores, no clamp", ga, grid);
    return 0;
}
