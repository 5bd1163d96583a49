// Issue cost (cycles per wave-instruction, one wave per SIMD, independent instructions) of the vector instructions the tracer's
// normalise3 is made of: f64 add / mul / fma / rsq / conversions beside their f32 counterparts.
// hipcc -O3 --offload-arch=gfx950 valu_cost_probe.hip -o valu_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, float seed) {
    double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7;
    float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3, f4 = seed + 4, f5 = seed + 5, f6 = seed + 6, f7 = seed + 7;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; it++) {
        if (MODE == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));) }
        if (MODE == 1) { REP16(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n v_fma_f64 %6, %6, %6, %6\n v_fma_f64 %7, %7, %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));) }
        if (MODE == 2) { REP16(asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3\n v_add_f64 %4, %4, %4\n v_add_f64 %5, %5, %5\n v_add_f64 %6, %6, %6\n v_add_f64 %7, %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));) }
        if (MODE == 3) { REP16(asm volatile("v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3\n v_mul_f64 %4, %4, %4\n v_mul_f64 %5, %5, %5\n v_mul_f64 %6, %6, %6\n v_mul_f64 %7, %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));) }
        if (MODE == 4) { REP16(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));) }
        if (MODE == 5) { REP16(asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n v_cvt_f64_f32 %4, %12\n v_cvt_f64_f32 %5, %13\n v_cvt_f64_f32 %6, %14\n v_cvt_f64_f32 %7, %15" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));) }
        if (MODE == 6) { REP16(asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));) }
        if (MODE == 7) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));) }
        if (MODE == 8) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));) }
        if (MODE == 9) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %0\n v_mul_lo_u32 %1, %1, %1\n v_mul_lo_u32 %2, %2, %2\n v_mul_lo_u32 %3, %3, %3\n v_mul_lo_u32 %4, %4, %4\n v_mul_lo_u32 %5, %5, %5\n v_mul_lo_u32 %6, %6, %6\n v_mul_lo_u32 %7, %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));) }
        if (MODE == 10) { REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));) }
        if (MODE == 11) { REP16(asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0\n v_div_scale_f32 %2, vcc, %2, %3, %2\n v_div_fmas_f32 %4, %4, %5, %4\n v_div_fixup_f32 %6, %6, %7, %6\n v_div_scale_f32 %0, vcc, %0, %1, %0\n v_div_scale_f32 %2, vcc, %2, %3, %2\n v_div_fmas_f32 %4, %4, %5, %4\n v_div_fixup_f32 %6, %6, %7, %6" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) :: "vcc");) }
        if (MODE == 12) { REP16(asm volatile("v_rndne_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_rndne_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_rndne_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_rndne_f32 %6, %6\n v_cvt_i32_f32 %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));) }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) == 12345.678f) out[1] = 1;
}
int main() {
    uint64_t *d; hipMalloc(&d, 64);
    const char *names[] = {"v_fma_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_rsq_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_rcp_f32", "v_pk_fma_f32", "v_mul_lo_u32", "v_sqrt_f32", "v_div_scale/fmas/fixup_f32", "v_rndne_f32 / v_cvt_i32_f32"};
    for (int waves = 1; waves <= 1; waves++) {
        for (int m = 0; m < 13; m++) {
            uint64_t h[2] = {0, 0};
            for (int rep = 0; rep < 2; rep++) {
                dim3 g(1), b(256 * waves);
#define L(M) case M: hipLaunchKernelGGL(k<M>, g, b, 0, 0, d, 1.5f); break;
                switch (m) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) }
                hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            }
            printf("%d wave(s) per SIMD  %-28s %6.2f cycles per instruction and wave\n", waves, names[m], (double)h[0] / (64.0 * 16 * 8) / waves);
        }
    }
    return 0;
}
