// HBM read ceiling on the box: a grid-stride sum over 3.5 GB with 16-byte loads, at several grid sizes / unrolls.
// hipcc --offload-arch=gfx950 -O3 -o read_probe read_probe.hip && ./read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int UNR>
__global__ __launch_bounds__(256) void rd(const float4 *__restrict__ p, size_t n, float *out) {
    float acc = 0.f;
    size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256 * UNR;
    for (; i + (UNR - 1) * 256 < n; i += step) {
        float4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNR; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 1234.5f) out[0] = acc;
}
int main() {
    const size_t bytes = 3523296000ull, n = bytes / 16;
    float4 *p; float *o;
    hipMalloc(&p, bytes); hipMalloc(&o, 4); hipMemset(p, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 4096, 8192, 16384, 65536}) {
        for (int unr : {1, 4, 8}) {
            float best = 1e9f;
            for (int r = 0; r < 5; r++) {
                hipEventRecord(e0);
                if (unr == 1) hipLaunchKernelGGL(rd<1>, dim3(grid), dim3(256), 0, 0, p, n, o);
                else if (unr == 4) hipLaunchKernelGGL(rd<4>, dim3(grid), dim3(256), 0, 0, p, n, o);
                else hipLaunchKernelGGL(rd<8>, dim3(grid), dim3(256), 0, 0, p, n, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("grid %6d unroll %d: %.3f ms  %.0f GB/s\n", grid, unr, best, bytes / best / 1e6);
        }
    }
    return 0;
}
