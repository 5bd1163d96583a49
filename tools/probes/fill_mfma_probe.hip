// Do a CU's LDS-DMA fills overlap with its MFMA blocks?  The fused GQI kernel's stage = 36 KB of fills (20 KB matrix pieces from L2,
// 16 KB samples from HBM, requested one / two stages ahead) + two MFMA blocks of 30 v_mfma_f32_32x32x16_f16 per SIMD; timing-only
// builds of the kernel say the two ADD UP.  This probe has nothing but the two in it:
//   8 waves per CU, persistent; per stage every wave issues NA piece requests (global_load_lds_dwordx4, 1 KB each, from a 340-KB
//   image that lives in L2) into the other ring buffer and 2 sample requests (buffer-less global_load_lds_dwordx4 from a streamed
//   1-GB buffer) into its own tile two stages ahead, runs its MFMA block on fragments read from the current ring buffer, waits for
//   everything but the sample request (s_waitcnt vmcnt(2)) and meets the others at s_barrier.
//   modes: fills (bit 0), MFMA (bit 1), samples (bit 2), MFMA only on waves 0-3 (bit 3), wait vmcnt(0) instead of (2) (bit 4),
//   a stand-in for the sample split (bit 5: 8 tile reads + ~100 VALU per wave and stage; waves 0-3 behind their MFMA block, waves 4-7 in
//   front of it, as the kernel's anti-phase halves), s_setprio 2 around the MFMA block (bit 6), the late waves' sample request in front
//   of the MFMA block (bit 7)
// Output: one JSON line per mode with cycles per stage (s_memtime) and ns per stage.
// hipcc --offload-arch=gfx950 -O3 -o fill_mfma_probe fill_mfma_probe.hip && ./fill_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TILEB = 20 * 1024, NSTAGE_IMG = 17;
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const char *__restrict__ img, const char *__restrict__ samples, size_t sample_bytes, int nstage,
                                                float *sink, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) char lds[2 * TILEB + 8 * 4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ring_l = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char *)lds);
    const uint32_t tile_l = ring_l + 2 * TILEB + wave * 4096;
    f32x16 acc[10];
    for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
    f16x8_t b0, b1;
    for (int j = 0; j < 8; j++) { b0[j] = (_Float16)(0.001f * (lane + j)); b1[j] = (_Float16)(1e-5f * (lane - j)); }
    // zero the LDS so that the MFMAs see finite numbers
    for (int i = tid; i < (2 * TILEB + 8 * 4096) / 4; i += 512) reinterpret_cast<float *>(lds)[i] = 0.0f;
    __syncthreads();
    const size_t wg_stride = (size_t)8 * 2048;                                   // bytes of samples per workgroup and stage
    auto fill_pieces = [&](int t, int buf) {
        if (!(MODE & 1)) return;
        const char *g = img + (size_t)(t % NSTAGE_IMG) * TILEB;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            int p = wave + 8 * i; p = p < 20 ? p : 19;
            const uint32_t d = __builtin_amdgcn_readfirstlane(ring_l + buf * TILEB + p * 1024);
            const char *src = g + p * 1024 + lane * 16;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(d), "v"(src) : "memory");
        }
    };
    auto fill_samples = [&](int t, int slot) {
        if (!(MODE & 4)) return;
        size_t off = ((size_t)t * gridDim.x + blockIdx.x) * wg_stride + (size_t)wave * 2048;
        off %= (sample_bytes - 4096);
        off &= ~(size_t)15;
        const uint32_t d = __builtin_amdgcn_readfirstlane(tile_l + slot * 2048);
        const char *src = samples + off + lane * 16;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" :: "s"(d), "v"(src) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" :: "s"(d + 1024u), "v"(src + 1024) : "memory");
    };
    float xs = 0.0f;
    uint32_t bp0[4] = {0, 0, 0, 0}, bp1[4] = {0, 0, 0, 0};
    auto split = [&](int slot) {                                                 // what the kernel's split costs, roughly
        if (!(MODE & 32)) return;
        const float *sp = reinterpret_cast<const float *>(lds + 2 * TILEB + wave * 4096 + slot * 2048) + (8 * (lane >> 5)) * 32 + (lane & 31);
        float c[8];
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = sp[j * 32];
        float vmax = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++) { c[j] = fmaxf(c[j], 0.0f); vmax = fmaxf(vmax, c[j]); xs = __builtin_fmaf(c[j], 0.37f + j, xs); }
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(vmax), __float_as_uint(vmax), false, false);
        const float mall = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        const int kx = 6 - (int)((__float_as_uint(mall) >> 23) & 0xffu) + 127;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const float p0 = __builtin_amdgcn_ldexpf(c[2 * jj], kx), p1 = __builtin_amdgcn_ldexpf(c[2 * jj + 1], kx);
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 h = {(_Float16)p0, (_Float16)p1};
            const float r0 = p0 - (float)h[0], r1 = p1 - (float)h[1];
            const h2 l = {(_Float16)r0, (_Float16)r1};
            bp0[jj] = __builtin_bit_cast(uint32_t, h); bp1[jj] = __builtin_bit_cast(uint32_t, l);
        }
    };
    fill_pieces(0, 0); fill_samples(0, 0); fill_samples(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int t = 0; t < nstage; t++) {
        const int cb = t & 1;
        fill_pieces(t + 1, cb ^ 1);
        if (wave >= 4) { split(t & 1); if (MODE & 128) fill_samples(t + 2, t & 1); }
        if (MODE & 32) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); const u4 q0 = {bp0[0], bp0[1], bp0[2], bp0[3]}, q1 = {bp1[0], bp1[1], bp1[2], bp1[3]};
                         b0 = __builtin_bit_cast(f16x8_t, q0); b1 = __builtin_bit_cast(f16x8_t, q1); }
        if ((MODE & 2) && (!(MODE & 8) || wave < 4)) {
            if (MODE & 64) __builtin_amdgcn_s_setprio(2);
            const f16x8_t *LA = reinterpret_cast<const f16x8_t *>(lds + cb * TILEB) + lane;
            f16x8_t a1 = LA[10 * 64], a0 = LA[0];
#pragma unroll
            for (int m = 0; m < 10; m++) {
                f16x8_t n1 = a1, n0 = a0;
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[m], 0, 0, 0);
                if (m + 1 < 10) n1 = LA[(10 + m + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[m], 0, 0, 0);
                if (m + 1 < 10) n0 = LA[(m + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a1 = n1; a0 = n0;
            }
            if (MODE & 64) __builtin_amdgcn_s_setprio(0);
        }
        if (wave < 4) split((t + 1) & 1);
        if (!(MODE & 128) || wave < 4) fill_samples(t + 2, t & 1);
        if (MODE & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (MODE & 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.0f;
    for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) s += acc[m][r];
    if (s + xs + (float)bp1[3] == 12345.678f) sink[0] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char *img, const char *samples, size_t sb, int nstage, float *sink, unsigned long long *cyc, int ncu, const char *what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(probe<MODE>, dim3(ncu), dim3(512), 0, 0, img, samples, sb, nstage, sink, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int NL = 5;
    for (int rep = 0; rep < NL; rep++) hipLaunchKernelGGL(probe<MODE>, dim3(ncu), dim3(512), 0, 0, img, samples, sb, nstage, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(ncu);
    hipMemcpy(h.data(), cyc, ncu * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("{\"mode\": %d, \"what\": \"%s\", \"cycles_per_stage_median\": %.0f, \"ns_per_stage\": %.1f, \"fill_TBps\": %.2f}\n", MODE, what,
           (double)h[ncu / 2] / nstage, ms / NL * 1e6 / nstage, ((MODE & 1 ? 20480.0 : 0) + (MODE & 4 ? 16384.0 : 0)) * ncu / (ms / NL * 1e-3 / nstage) / 1e12);
}
int main() {
    int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t sb = (size_t)2 << 30;
    char *img, *samples; float *sink; unsigned long long *cyc;
    hipMalloc(&img, NSTAGE_IMG * TILEB); hipMalloc(&samples, sb); hipMalloc(&sink, 64); hipMalloc(&cyc, 4096 * 8);
    hipMemset(img, 0, NSTAGE_IMG * TILEB); hipMemset(samples, 0, sb);
    const int nstage = 17 * 42;
    run<2>(img, samples, sb, nstage, sink, cyc, ncu, "MFMA blocks only (60 per SIMD and stage)");
    run<1>(img, samples, sb, nstage, sink, cyc, ncu, "piece fills only (20 KB per CU and stage, from L2)");
    run<5>(img, samples, sb, nstage, sink, cyc, ncu, "piece + sample fills (36 KB per CU and stage)");
    run<3>(img, samples, sb, nstage, sink, cyc, ncu, "MFMA + piece fills");
    run<7>(img, samples, sb, nstage, sink, cyc, ncu, "MFMA + piece + sample fills");
    run<15>(img, samples, sb, nstage, sink, cyc, ncu, "MFMA on waves 0-3 only + all fills");
    run<23>(img, samples, sb, nstage, sink, cyc, ncu, "MFMA + all fills, vmcnt(0) at the stage end");
    run<39>(img, samples, sb, nstage, sink, cyc, ncu, "MFMA + all fills + split stand-in (anti-phase halves)");
    run<103>(img, samples, sb, nstage, sink, cyc, ncu, "same + s_setprio 2 around the MFMA block");
    run<231>(img, samples, sb, nstage, sink, cyc, ncu, "same + late waves request their samples in front of the MFMA block");
    run<37>(img, samples, sb, nstage, sink, cyc, ncu, "fills + split stand-in, no MFMA");
    return 0;
}
