// overlap_probe.hip -- do VALU instructions of one wave overlap with bf16 MFMAs of ANOTHER wave on the same SIMD (gfx950)?
// 8 waves per workgroup = 2 per SIMD: waves 0-3 issue MFMAs, waves 4-7 issue VALU (or VMEM) work.  Timed alone and together.
// build: hipcc -O3 --offload-arch=gfx950 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int KIND>   // second role: 0 = v_fma_f32 chain x8 accumulators, 1 = v_cvt_pk_bf16 + sub (the split's mix), 2 = global loads
__global__ __launch_bounds__(512, 2) void probe(int do_mfma, int do_other, int iters, const float *src, float *out, int prio) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.0f;
    if (wave < 4) {
        if (!do_mfma) return;
        f32x16 acc[4];
        for (int m = 0; m < 4; m++) for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
        bf16x8_t a, b;
        for (int i = 0; i < 8; i++) { a[i] = (__bf16)(1.0f + lane * 0.001f); b[i] = (__bf16)(0.5f); }
        for (int t = 0; t < iters; t++) {
#pragma unroll
            for (int u = 0; u < 15; u++)                   // 60 MFMAs per trip, like one stage of the contraction
#pragma unroll
                for (int m = 0; m < 4; m++) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
        }
        for (int m = 0; m < 4; m++) for (int r = 0; r < 16; r++) s += acc[m][r];
    } else {
        if (!do_other) return;
        if (prio) __builtin_amdgcn_s_setprio(3);
        float x[8];
        for (int i = 0; i < 8; i++) x[i] = 1.0f + lane * 1e-3f + i;
        for (int t = 0; t < iters; t++) {
            if (KIND == 0) {
#pragma unroll
                for (int u = 0; u < 60; u++)               // 480 VALU per trip
#pragma unroll
                    for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
            } else if (KIND == 1) {
#pragma unroll
                for (int u = 0; u < 20; u++)
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        unsigned h;
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(x[i]), "v"(x[i + 1]));
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(__uint_as_float(h << 16)));
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i + 1]) : "v"(__uint_as_float(h & 0xffff0000u)));
                    }
            } else {
#pragma unroll
                for (int u = 0; u < 12; u++) x[u & 7] += src[(size_t)((t * 12 + u) & 1023) * 65536 + blockIdx.x * 512 + tid];
            }
        }
        for (int i = 0; i < 8; i++) s += x[i];
    }
    if (s == 123.456f) out[blockIdx.x * 512 + tid] = s;
}

template <int KIND>
void run(const char *name, const float *src, float *out, int prio) {
    const int nblk = 256, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms[3];
    const int cfg[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int c = 0; c < 3; c++) {
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(probe<KIND>, dim3(nblk), dim3(512), 0, 0, cfg[c][0], cfg[c][1], iters, src, out, prio);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        (void)hipEventElapsedTime(&ms[c], e0, e1);
    }
    printf("%-34s MFMA alone %.3f ms, other alone %.3f ms, together %.3f ms  (sum %.3f, max %.3f)\n", name, ms[0], ms[1], ms[2], ms[0] + ms[1],
           ms[0] > ms[1] ? ms[0] : ms[1]);
}

int main() {
    float *src, *out;
    (void)hipMalloc(&src, (size_t)1024 * 65536 * 4 + (1 << 20));
    (void)hipMalloc(&out, 256 * 512 * 4);
    for (int prio = 0; prio < 2; prio++) {
        printf("-- the non-MFMA waves run at s_setprio %d\n", prio ? 3 : 0);
        run<0>("v_fma_f32 (480 per 60 MFMAs)", src, out, prio);
        run<1>("cvt_pk_bf16 + 2 sub (240 per 60)", src, out, prio);
        run<2>("global loads (12 per 60)", src, out, prio);
    }
    return 0;
}
