// fill_probe.hip — pure-write HBM bandwidth on MI355X for a few store shapes (build: hipcc -O3 --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void fill_plain(float4 *p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(256) void fill_nt(float4 *p, size_t n4) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(z, reinterpret_cast<v4 *>(p) + i);
}
// one block = one contiguous chunk of `chunk4` float4 (like a GEMM epilogue writing a tile row by row)
__global__ __launch_bounds__(256) void fill_chunk(float4 *p, size_t n4, int chunk4) {
    size_t base = (size_t)blockIdx.x * chunk4;
    for (int i = threadIdx.x; i < chunk4 && base + i < n4; i += 256) p[base + i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(256) void fill_dword(float *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}
// strided rows: each wave writes 128 B per row (32 lanes x 4 B) for many rows — the GEMM's store shape
__global__ __launch_bounds__(256) void fill_rows(float *p, size_t stride, int nrows, size_t ncol) {
    size_t col = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncol) return;
    for (int r = 0; r < nrows; r++) p[(size_t)r * stride + col] = 0.f;
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    float *d;
    CK(hipMalloc(&d, bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto report = [&](const char *name, float ms) { printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); };
    for (int rep = 0; rep < 2; rep++) {
        float ms;
        CK(hipEventRecord(a)); CK(hipMemsetAsync(d, 0, bytes)); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); report("hipMemsetAsync", ms);
        for (int grid : {1024, 4096, 16384, 65536}) {
            CK(hipEventRecord(a)); hipLaunchKernelGGL(fill_plain, dim3(grid), dim3(256), 0, 0, (float4 *)d, bytes / 16); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            char nm[64]; snprintf(nm, 64, "plain x4 grid=%d", grid); report(nm, ms);
        }
        for (int grid : {1024, 16384}) {
            CK(hipEventRecord(a)); hipLaunchKernelGGL(fill_nt, dim3(grid), dim3(256), 0, 0, (float4 *)d, bytes / 16); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            char nm[64]; snprintf(nm, 64, "nontemporal x4 grid=%d", grid); report(nm, ms);
        }
        for (int chunk4 : {256, 2048, 16384}) {
            size_t n4 = bytes / 16;
            CK(hipEventRecord(a)); hipLaunchKernelGGL(fill_chunk, dim3((unsigned)((n4 + chunk4 - 1) / chunk4)), dim3(256), 0, 0, (float4 *)d, n4, chunk4); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            char nm[64]; snprintf(nm, 64, "chunk %d KB per block", chunk4 * 16 / 1024); report(nm, ms);
        }
        CK(hipEventRecord(a)); hipLaunchKernelGGL(fill_dword, dim3(16384), dim3(256), 0, 0, d, bytes / 4); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); report("plain dword grid=16384", ms);
        {   // 373 rows x 2744000 columns (~4 GB): one thread per column walks the rows
            const size_t ncol = 2744000; const int nrows = (int)(bytes / 4 / ncol);
            CK(hipEventRecord(a)); hipLaunchKernelGGL(fill_rows, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, 0, d, ncol, nrows, ncol); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            printf("%-28s %8.3f ms  %7.1f GB/s\n", "rows: thread per column", ms, (double)nrows * ncol * 4 / ms / 1e6);
        }
    }
    return 0;
}
