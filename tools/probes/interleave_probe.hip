// interleave_probe.hip -- ONE wave per SIMD (gfx950): how many independent VALU / LDS instructions of the SAME wave fit behind a
// v_mfma_f32_32x32x16_bf16 without lengthening the MFMA stream?  Each trip issues 60 MFMAs (10 accumulators x 6, dependent in
// sixes like the contraction) with NV VALU (+ ND LDS reads) instructions after every MFMA.
// build: hipcc -O3 --offload-arch=gfx950 interleave_probe.hip -o interleave_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int NV, int ND, int KIND>   // KIND 0: v_fma_f32, 1: v_max3_f32, 2: v_pk_add_f32, 3: v_cmp + v_addc, 4: 4 v_fma + one buffer_load..lds (1 KB) every NV-th MFMA
__global__ __launch_bounds__(256, 1) void probe(int iters, float *out, const float *src) {
    __shared__ __attribute__((aligned(16))) float lds[4096 + 8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    f32x16 acc[10];
    for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
    bf16x8_t a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(1.0f + lane * 0.001f); b[i] = (__bf16)(0.5f); }
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = 1.0f + lane * 1e-3f + i;
    float d[4] = {0, 0, 0, 0};
    const float *lp = lds + lane;
    i32x4 rs;
    { const unsigned long long b = (unsigned long long)src; rs[0] = (int)(unsigned)b; rs[1] = (int)((b >> 32) & 0xffff); rs[2] = 1 << 20; rs[3] = 0x00020000; }
    const unsigned ldsa = (unsigned)(unsigned long)((__attribute__((address_space(3))) char *)(lds + 4096)) + __builtin_amdgcn_readfirstlane(tid >> 6) * 8192;
    for (int t = 0; t < iters; t++) {
#pragma unroll
        for (int k = 0; k < 60; k++) {
            acc[k / 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k / 6], 0, 0, 0);
            if (KIND == 4) {
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(x[(i + 3) & 7]));
                if (k % NV == 0 && k / NV < 10)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(ldsa + (unsigned)((k / NV) & 7) * 1024u), "v"(lane * 16), "s"(rs), "s"(((t * 10 + k / NV) & 511) * 1024) : "memory");
            }
#pragma unroll
            for (int i = 0; i < (KIND == 4 ? 0 : NV); i++) {
                const int j = (k * NV + i) & 7;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[j]) : "v"(x[(j + 3) & 7]));
                if (KIND == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(x[(j + 3) & 7]), "v"(x[(j + 5) & 7]));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&x[j & 6])) : "v"(*reinterpret_cast<double *>(&x[(j + 2) & 6])));
                if (KIND == 3) asm volatile("v_cmp_nge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x[j]) : "v"(x[(j + 3) & 7]), "v"(x[(j + 5) & 7]) : "vcc");
            }
#pragma unroll
            for (int i = 0; i < ND; i++) {
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d[(k + i) & 3]) : "v"((unsigned)(lane * 4)), "n"(256 * 4));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
        if (KIND == 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    }
    float s = 0;
    for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) s += acc[m][r];
    for (int i = 0; i < 8; i++) s += x[i];
    for (int i = 0; i < 4; i++) s += d[i];
    if (s == 123.456f) out[blockIdx.x * 256 + tid] = s;
}

template <int NV, int ND, int KIND>
void run(float *out, const float *src = nullptr) {
    const int nblk = 256, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NV, ND, KIND>), dim3(nblk), dim3(256), 0, 0, 10, out, src);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NV, ND, KIND>), dim3(nblk), dim3(256), 0, 0, iters, out, src);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    // cycles per MFMA slot at the measured time, assuming 2.4 GHz is NOT sustained: report ns per slot instead
    printf("kind %d  valu/mfma %d  lds/mfma %d : %.3f ms  = %.2f ns per MFMA slot\n", KIND, NV, ND, ms, ms * 1e6 / (iters * 60.0));
}

int main() {
    float *out; (void)hipMalloc(&out, 256 * 256 * 4);
    run<0, 0, 0>(out);
    run<1, 0, 0>(out); run<2, 0, 0>(out); run<3, 0, 0>(out); run<4, 0, 0>(out); run<5, 0, 0>(out); run<6, 0, 0>(out); run<7, 0, 0>(out); run<8, 0, 0>(out); run<10, 0, 0>(out);
    run<4, 0, 1>(out); run<6, 0, 1>(out);
    run<4, 0, 2>(out); run<6, 0, 2>(out);
    run<2, 0, 3>(out); run<3, 0, 3>(out);
    { float *src; (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 0, 1 << 20);
      run<4, 0, 0>(out); run<1, 0, 4>(out, src); run<2, 0, 4>(out, src); run<3, 0, 4>(out, src); run<6, 0, 4>(out, src); }
    run<0, 1, 0>(out); run<0, 2, 0>(out); run<4, 1, 0>(out); run<5, 1, 0>(out); run<4, 2, 0>(out);
    return 0;
}
