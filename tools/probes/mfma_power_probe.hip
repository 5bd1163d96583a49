// mfma_power_probe.hip -- dense bf16 MFMA streams at full rate, 4 waves per CU (one per SIMD), for a few seconds each: which shape
// does the board's power limit let run faster?  Prints TFLOP/s; run tools/clock_watch-style sampling of rocm-smi beside it.
// build: hipcc -O3 --offload-arch=gfx950 mfma_power_probe.hip -o mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int SHAPE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(int iters, float *out) {
    const int lane = threadIdx.x & 63;
    bf16x8_t a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(1.0f + lane * 0.001f); b[i] = (__bf16)(0.5f); }
    float s = 0;
    if (SHAPE == 0) {                                   // 32x32x16: 10 accumulators, 60 MFMAs per trip
        f32x16 acc[10];
        for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
        for (int t = 0; t < iters; t++)
#pragma unroll
            for (int k = 0; k < 60; k++) acc[k % 10] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k % 10], 0, 0, 0);
        for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) s += acc[m][r];
    } else if (SHAPE >= 2) {                            // 32x32x16 with dependent MFMAs DIST apart: 6 products per accumulator as in the contraction
        constexpr int DIST = SHAPE - 1;                  // 1, 2, 3
        f32x16 acc[12];
        for (int m = 0; m < 12; m++) for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
        for (int t = 0; t < iters; t++)
#pragma unroll
            for (int k = 0; k < 60; k++) {
                const int grp = k / (6 * DIST), idx = grp * DIST + (k % DIST);      // DIST accumulators take turns, 6 MFMAs each
                acc[idx % 12] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[idx % 12], 0, 0, 0);
            }
        for (int m = 0; m < 12; m++) for (int r = 0; r < 16; r++) s += acc[m][r];
    } else {                                            // 16x16x32: 40 accumulators, 120 MFMAs per trip (same flops per trip)
        f32x4 acc[40];
        for (int m = 0; m < 40; m++) for (int r = 0; r < 4; r++) acc[m][r] = 0.0f;
        for (int t = 0; t < iters; t++)
#pragma unroll
            for (int k = 0; k < 120; k++) acc[k % 40] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k % 40], 0, 0, 0);
        for (int m = 0; m < 40; m++) for (int r = 0; r < 4; r++) s += acc[m][r];
    }
    if (s == 123.456f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int WAVES>
void run(float *out, double seconds) {
    const int nblk = 256, iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE, WAVES>), dim3(nblk), dim3(WAVES * 64), 0, 0, 100, out);
    (void)hipDeviceSynchronize();
    double total_ms = 0; int n = 0;
    while (total_ms < seconds * 1e3) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<SHAPE, WAVES>), dim3(nblk), dim3(WAVES * 64), 0, 0, iters, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms; n++;
    }
    const double flops = (double)n * iters * 60.0 * 2.0 * 32 * 32 * 16 * nblk * WAVES;
    printf("shape %s, %d waves/CU: %.1f TFLOP/s dense bf16 over %.1f s\n", SHAPE == 0 ? "32x32x16" : SHAPE == 1 ? "16x16x32" : SHAPE == 2 ? "32x32x16 dependent, distance 1" : SHAPE == 3 ? "32x32x16 dependent, distance 2" : "32x32x16 dependent, distance 3", WAVES, flops / (total_ms * 1e-3) / 1e12, total_ms * 1e-3);
    fflush(stdout);
}

int main(int argc, char **argv) {
    float *out; (void)hipMalloc(&out, 256 * 512 * 4);
    const double sec = argc > 1 ? atof(argv[1]) : 4.0;
    run<0, 4>(out, sec);
    run<2, 4>(out, sec);
    run<3, 4>(out, sec);
    run<4, 4>(out, sec);
    run<2, 8>(out, sec);
    run<3, 8>(out, sec);
    run<1, 4>(out, sec);
    run<1, 8>(out, sec);
    return 0;
}
