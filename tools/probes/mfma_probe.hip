// mfma_probe.hip — standalone experiment: what limits the GQI GEMM main loop on gfx950?
// build: hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MB = 11, KT = 16, MW = MB * 32, TILE = KT * MW;

// MODE 0: LDS A fragments + barriers + glds staging (no global B)   1: no LDS reads (A const)   2: LDS reads, no staging/barriers
template <int MODE, int WAVES_PER_SIMD, int NT = 0, bool SPREAD = false>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void probe(const float *At, float *out, int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kh = lane >> 5;
    f32x16 acc[MB];
    for (int m = 0; m < MB; m++) for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
    for (int i = tid; i < 2 * TILE; i += 256) lds[i] = At[i % TILE];
    __syncthreads();
    float b = 1.0f + lane * 1e-6f;
    float junk[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    for (int t = 0; t < ntiles; t++) {
        const int cur = t & 1;
        if (MODE == 0) {
            const char *g = reinterpret_cast<const char *>(At + (size_t)(t % 16) * TILE);
            char *l = reinterpret_cast<char *>(lds + (cur ^ 1) * TILE);
            for (int p = wave; p < 2 * MB; p += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + p * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(l + p * 1024), 16, 0, 0);
        }
        const float *L = lds + cur * TILE + kh * MW + col;
#pragma unroll
        for (int kk = 0; kk < KT / 2; kk++) {
#pragma unroll
            for (int m = 0; m < MB; m++) {
                const float av = (MODE == 1) ? b : L[2 * kk * MW + m * 32];
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[m], 0, 0, 0);
                if (NT > 0 && SPREAD) {
#pragma unroll
                    for (int i = 0; i < (NT + 87) / 88; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(junk[(m + i) & 7]) : "v"(b));
                }
            }
        }
        if (NT > 0 && !SPREAD) {
#pragma unroll
            for (int i = 0; i < NT; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(junk[i & 7]) : "v"(b));
        }
        if (MODE == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    float s = junk[0] + junk[1] + junk[2] + junk[3] + junk[4] + junk[5] + junk[6] + junk[7];
    for (int m = 0; m < MB; m++) for (int r = 0; r < 16; r++) s += acc[m][r];
    if (s == 123.456f) out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int W, int NT = 0, bool SP = false>
void run(const char *name, const float *At, float *out, int nblk) {
    const int ntiles = 17;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<MODE, W, NT, SP>), dim3(nblk), dim3(256), 0, 0, At, out, ntiles);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    double flops = (double)nblk * 4 * ntiles * (KT / 2) * MB * 4096.0;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}

int main() {
    float *At, *out;
    hipMalloc(&At, sizeof(float) * TILE * 17);
    hipMalloc(&out, sizeof(float) * 256 * 30000);
    std::vector<float> h(TILE * 17, 0.5f);
    hipMemcpy(At, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int nblk = 21438;
    run<1, 2>("no LDS, 2 waves/SIMD", At, out, nblk);
    run<1, 1>("no LDS, 1 wave/SIMD", At, out, nblk);
    run<2, 2>("LDS reads only, 2 w/SIMD", At, out, nblk);
    run<0, 2>("LDS+glds+barrier, 2 w/SIMD", At, out, nblk);
    run<0, 1>("LDS+glds+barrier, 1 w/SIMD", At, out, nblk);
    run<0, 2, 88>("  + 88 VALU tail/tile", At, out, nblk);
    run<0, 2, 264>("  + 264 VALU tail/tile", At, out, nblk);
    run<0, 2, 264, true>("  + 264 VALU spread (3/MFMA)", At, out, nblk);
    run<0, 2, 528, true>("  + 528 VALU spread (6/MFMA)", At, out, nblk);
    run<0, 1, 264>("1w: + 264 VALU tail/tile", At, out, nblk);
    run<0, 1, 264, true>("1w: + 264 VALU spread", At, out, nblk);
    return 0;
}
