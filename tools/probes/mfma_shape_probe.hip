// mfma_shape_probe.hip -- which bf16 MFMA shape does the chip hold a higher clock on, on REAL operands?
//
// MI355X_MICROARCH.md "DVFS give-back" items 6 and 7: the test is the in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz)
// of a loop on random data, not rocm-smi, and constant / zero operands hide the effect.  This probe runs the contraction's
// instruction mix -- 6 piece products per f32 product, operands = exact 3-way bf16 splits of random f32 numbers, every matrix
// fragment re-read from LDS by ds_read_b128, the same 320 x 32 output tile per wave -- once with v_mfma_f32_32x32x16_bf16
// (10 accumulators of 16 registers, 60 MFMAs per 16 frames: the shape odf_gemm3_kernel uses) and once with
// v_mfma_f32_16x16x32_bf16 (40 accumulators of 4 registers, 240 MFMAs per 32 frames), at one and at two waves per SIMD, with
// the matrix operand from LDS or held in registers, and on all-zero data for the cycle ranking.
// Prints one JSON object per variant: executed dense-bf16 TFLOP/s, in-kernel clock (median over workgroups), cycles per MFMA.
// build: hipcc -O3 --offload-arch=gfx950 mfma_shape_probe.hip -o mfma_shape_probe ; run: ./mfma_shape_probe [seconds per variant]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr int STAGE0 = 30 * 1024;     // 32x32x16: 3 pieces x 10 blocks x 1 KiB per 16 frames
constexpr int STAGE1 = 60 * 1024;     // 16x16x32: 3 pieces x 20 blocks x 1 KiB per 32 frames

// SHAPE 0: 32x32x16, SHAPE 1: 16x16x32.  LDSA: matrix fragments re-read from LDS every stage (else: 3 fragments held in registers).
template <int SHAPE, int NW, bool LDSA>
__global__ __launch_bounds__(NW * 64) void probe(int iters, const uint32_t *__restrict__ amat, const uint32_t *__restrict__ bmat,
                                                  unsigned long long *__restrict__ stamps, float *__restrict__ sink) {
    constexpr int STAGE = SHAPE == 0 ? STAGE0 : STAGE1;
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * STAGE / 16; i += NW * 64)
        reinterpret_cast<u32x4_t *>(lds)[i] = reinterpret_cast<const u32x4_t *>(amat)[i];
    // sample pieces of this lane: [col block][piece] (32x32x16 has one column block of 32 voxels, 16x16x32 two of 16)
    constexpr int NCB = SHAPE == 0 ? 1 : 2;
    bf16x8_t b[NCB][3];
    for (int c = 0; c < NCB; c++)
        for (int p = 0; p < 3; p++)
            b[c][p] = __builtin_bit_cast(bf16x8_t, reinterpret_cast<const u32x4_t *>(bmat)[((blockIdx.x * NW + (tid >> 6)) * 6 + c * 3 + p) * 64 + lane]);
    __syncthreads();
    float s = 0.0f;
    unsigned long long c0, r0, c1, r1;
    if constexpr (SHAPE == 0) {
        f32x16 acc[10];
        for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
        bf16x8_t f0 = reinterpret_cast<const bf16x8_t *>(lds)[lane], f1 = reinterpret_cast<const bf16x8_t *>(lds)[640 + lane], f2 = reinterpret_cast<const bf16x8_t *>(lds)[1280 + lane];
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
        for (int t = 0; t < iters; t++) {
            const bf16x8_t *LA = reinterpret_cast<const bf16x8_t *>(lds + (t & 1) * STAGE) + lane;
#pragma unroll
            for (int m = 0; m < 10; m++) {
                bf16x8_t a0 = f0, a1 = f1, a2 = f2;
                if (LDSA) { a0 = LA[m * 64]; a1 = LA[(10 + m) * 64]; a2 = LA[(20 + m) * 64]; }
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[0][0], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[0][1], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[0][2], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[0][0], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[0][1], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[0][0], acc[m], 0, 0, 0);
            }
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
        for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) s += acc[m][r];
    } else {
        f32x4 acc[20][2];
        for (int m = 0; m < 20; m++) for (int c = 0; c < 2; c++) for (int r = 0; r < 4; r++) acc[m][c][r] = 0.0f;
        bf16x8_t f0 = reinterpret_cast<const bf16x8_t *>(lds)[lane], f1 = reinterpret_cast<const bf16x8_t *>(lds)[1280 + lane], f2 = reinterpret_cast<const bf16x8_t *>(lds)[2560 + lane];
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
        for (int t = 0; t < iters; t++) {
            const bf16x8_t *LA = reinterpret_cast<const bf16x8_t *>(lds + (t & 1) * STAGE) + lane;
#pragma unroll
            for (int m = 0; m < 20; m++) {
                bf16x8_t a0 = f0, a1 = f1, a2 = f2;
                if (LDSA) { a0 = LA[m * 64]; a1 = LA[(20 + m) * 64]; a2 = LA[(40 + m) * 64]; }
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b[c][0], acc[m][c], 0, 0, 0);
                    acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[c][1], acc[m][c], 0, 0, 0);
                    acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[c][2], acc[m][c], 0, 0, 0);
                    acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[c][0], acc[m][c], 0, 0, 0);
                    acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[c][1], acc[m][c], 0, 0, 0);
                    acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[c][0], acc[m][c], 0, 0, 0);
                }
            }
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
        for (int m = 0; m < 20; m++) for (int c = 0; c < 2; c++) for (int r = 0; r < 4; r++) s += acc[m][c][r];
    }
    if (tid == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }   // a buffer of their own: nothing reads it
    if (s == 123.456f) sink[blockIdx.x * blockDim.x + tid] = s;
}

static uint16_t bf16_rn(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// n f32 numbers in (-1, 1) -> three bf16 piece images of n elements each (piece p of element i at [p][i])
static void split3(std::mt19937 &rng, size_t n, uint16_t *p0, uint16_t *p1, uint16_t *p2, bool zero) {
    std::uniform_real_distribution<float> d(-1.0f, 1.0f);
    for (size_t i = 0; i < n; i++) {
        const float v = zero ? 0.0f : d(rng);
        const uint16_t h1 = bf16_rn(v); const float r1 = v - bf16_f(h1);
        const uint16_t h2 = bf16_rn(r1); const float r2 = r1 - bf16_f(h2);
        p0[i] = h1; p1[i] = h2; p2[i] = bf16_rn(r2);
    }
}

template <int SHAPE, int NW, bool LDSA>
void run(const char *label, double seconds, bool zero, uint32_t *d_a, uint32_t *d_b, unsigned long long *d_st, float *d_sink) {
    const int nblk = 256;
    constexpr int STAGE = SHAPE == 0 ? STAGE0 : STAGE1;
    // matrix image: 2 stage buffers, inside a stage [piece][block][64 lanes][8 bf16]
    std::mt19937 rng(12345);
    {
        const size_t nfrag = (size_t)2 * STAGE / 16 / 3 / 64;      // fragments (blocks) per piece over both buffers... per buffer: STAGE/3/1024
        (void)nfrag;
        std::vector<uint16_t> img((size_t)2 * STAGE / 2);
        const size_t per_piece = (size_t)STAGE / 3 / 2;              // bf16 elements of one piece in one stage
        for (int buf = 0; buf < 2; buf++) {
            uint16_t *base = img.data() + (size_t)buf * STAGE / 2;
            split3(rng, per_piece, base, base + per_piece, base + 2 * per_piece, zero);
        }
        (void)hipMemcpy(d_a, img.data(), img.size() * 2, hipMemcpyHostToDevice);
        std::vector<uint16_t> bim((size_t)nblk * 8 * 6 * 64 * 8);
        // per wave: [col block 2][piece 3][64 lanes][8]
        for (size_t w = 0; w < (size_t)nblk * 8; w++)
            for (int c = 0; c < 2; c++) {
                uint16_t *q = bim.data() + (w * 6 + c * 3) * 512;
                split3(rng, 512, q, q + 512, q + 1024, zero);
            }
        (void)hipMemcpy(d_b, bim.data(), bim.size() * 2, hipMemcpyHostToDevice);
    }
    const int iters = SHAPE == 0 ? 16000 : 8000;                     // the same frames per launch: 256 000
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE, NW, LDSA>), dim3(nblk), dim3(NW * 64), 0, 0, 100, d_a, d_b, d_st, d_sink);
    (void)hipDeviceSynchronize();
    double total_ms = 0; int n = 0;
    while (total_ms < seconds * 1e3) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<SHAPE, NW, LDSA>), dim3(nblk), dim3(NW * 64), 0, 0, iters, d_a, d_b, d_st, d_sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms; n++;
    }
    std::vector<unsigned long long> st(2 * nblk);
    (void)hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz(nblk), cyc(nblk);
    for (int i = 0; i < nblk; i++) { ghz[i] = (double)st[2 * i] / (double)st[2 * i + 1] * 0.1; cyc[i] = (double)st[2 * i]; }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double mfma_per_wave = (double)iters * (SHAPE == 0 ? 60.0 : 240.0);
    const double flop_per_mfma = SHAPE == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    const double flops = (double)n * mfma_per_wave * flop_per_mfma * nblk * NW;
    const double waves_per_simd = NW / 4.0;
    printf("{\"variant\": \"%s\", \"shape\": \"%s\", \"waves_per_simd\": %d, \"matrix_from\": \"%s\", \"data\": \"%s\", \"tflops_dense_bf16\": %.1f, "
           "\"clock_ghz_median\": %.3f, \"clock_ghz_min\": %.3f, \"clock_ghz_max\": %.3f, \"cycles_per_mfma_per_simd\": %.2f, \"launches\": %d, \"seconds\": %.1f}\n",
           label, SHAPE == 0 ? "32x32x16" : "16x16x32", NW / 4, LDSA ? "lds ds_read_b128" : "registers", zero ? "zeros" : "random f32 split in 3 bf16 pieces",
           flops / (total_ms * 1e-3) / 1e12, ghz[nblk / 2], ghz[0], ghz[nblk - 1], cyc[nblk / 2] / (mfma_per_wave * waves_per_simd), n, total_ms * 1e-3);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const double sec = argc > 1 ? atof(argv[1]) : 2.5;
    uint32_t *d_a, *d_b; unsigned long long *d_st; float *d_sink;
    (void)hipMalloc(&d_a, 2 * STAGE1); (void)hipMalloc(&d_b, (size_t)256 * 8 * 6 * 64 * 16);
    (void)hipMalloc(&d_st, 512 * 8); (void)hipMalloc(&d_sink, 256 * 512 * 4);
    for (int rep = 0; rep < 2; rep++) {                              // interleaved rounds in one process (rule 24)
        run<0, 8, true>("32x32x16 lds 2w", sec, false, d_a, d_b, d_st, d_sink);
        run<1, 8, true>("16x16x32 lds 2w", sec, false, d_a, d_b, d_st, d_sink);
        run<0, 4, true>("32x32x16 lds 1w", sec, false, d_a, d_b, d_st, d_sink);
        run<1, 4, true>("16x16x32 lds 1w", sec, false, d_a, d_b, d_st, d_sink);
    }
    run<0, 4, false>("32x32x16 regs 1w", sec, false, d_a, d_b, d_st, d_sink);
    run<1, 4, false>("16x16x32 regs 1w", sec, false, d_a, d_b, d_st, d_sink);
    run<0, 8, true>("32x32x16 lds 2w zeros", sec, true, d_a, d_b, d_st, d_sink);
    run<1, 8, true>("16x16x32 lds 2w zeros", sec, true, d_a, d_b, d_st, d_sink);
    return 0;
}
