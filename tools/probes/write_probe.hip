// Sustained store bandwidth to a buffer far larger than the caches (4 GiB), by store flavour and shape -- what bounds the tracker's
// point scratch and the pack kernel's output.  hipcc -O3 --offload-arch=gfx950 write_probe.hip -o write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x3_t __attribute__((ext_vector_type(3)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NT>   // 0 plain, 1 nt, 2 sc1 (write-through), 3 sc0 sc1 nt
__global__ __launch_bounds__(256) void k_store16(u32x4_t *dst, size_t n16, uint32_t seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4_t v = {seed, (uint32_t)threadIdx.x, (uint32_t)blockIdx.x, 7u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        v[0] += (uint32_t)i;
        if (NT == 0) dst[i] = v;
        else if (NT == 1) __builtin_nontemporal_store(v, dst + i);
        else if (NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(dst + i), "v"(v) : "memory");
    }
}
// each workgroup owns a contiguous chunk (blocked instead of grid-strided)
template <int NT>
__global__ __launch_bounds__(256) void k_store16_blocked(u32x4_t *dst, size_t n16, uint32_t seed) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t b = per * blockIdx.x, e = b + per < n16 ? b + per : n16;
    u32x4_t v = {seed, (uint32_t)threadIdx.x, (uint32_t)blockIdx.x, 7u};
    for (size_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        v[0] += (uint32_t)i;
        if (NT == 0) dst[i] = v; else __builtin_nontemporal_store(v, dst + i);
    }
}
// the tracer's shape: 12 bytes per lane, TILE lines share a row of TILE x 12 B per trip (16 lines: 192-B rows, a wave writes 4 runs that
// end in the middle of a 128-byte line; 32 lines: 384-B rows = 3 whole lines), rows of a tile 144 trips deep
template <int NT, int TILE>
__global__ __launch_bounds__(256) void k_store12_tiles(float *dst, size_t ntiles, int nslots, uint32_t seed) {
    const size_t li = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li / TILE >= ntiles) return;
    float *d = dst + (li / TILE) * ((size_t)nslots * TILE * 3) + (li % TILE) * 3;
    f32x3_t v = {(float)seed, (float)threadIdx.x, 1.0f};
    for (int t = 0; t < nslots; t++) {
        v[0] += 1.0f;
        if (NT == 0) *reinterpret_cast<f32x3_t *>(d) = v; else __builtin_nontemporal_store(v, reinterpret_cast<f32x3_t *>(d));
        d += TILE * 3;
    }
}
// the same bytes as three planes per row: [slot][component][TILE lines] -- one dword per lane and store
template <int NT, int TILE>
__global__ __launch_bounds__(256) void k_store4x3_tiles(float *dst, size_t ntiles, int nslots, uint32_t seed) {
    const size_t li = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li / TILE >= ntiles) return;
    float *d = dst + (li / TILE) * ((size_t)nslots * TILE * 3) + (li % TILE);
    float v = (float)seed + threadIdx.x;
    for (int t = 0; t < nslots; t++) {
        v += 1.0f;
        if (NT == 0) { d[0] = v; d[TILE] = v; d[2 * TILE] = v; }
        else { __builtin_nontemporal_store(v, d); __builtin_nontemporal_store(v, d + TILE); __builtin_nontemporal_store(v, d + 2 * TILE); }
        d += TILE * 3;
    }
}

// 16-line tiles, but a lane keeps the point of an even trip and stores it together with the odd trip's: the two stores of a wave cover
// 4 x 384 contiguous bytes = whole 128-byte lines, issued back to back
template <int NT, int TILE, int GROUP>
__global__ __launch_bounds__(256) void k_store12_paired(float *dst, size_t ntiles, int nslots, uint32_t seed) {
    const size_t li = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li / TILE >= ntiles) return;
    float *d = dst + (li / TILE) * ((size_t)nslots * TILE * 3) + (li % TILE) * 3;
    f32x3_t v = {(float)seed, (float)threadIdx.x, 1.0f};
    f32x3_t keep[GROUP];
    for (int t = 0; t < nslots; t += GROUP) {
#pragma unroll
        for (int g = 0; g < GROUP; g++) { v[0] += 1.0f; keep[g] = v; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int g = 0; g < GROUP; g++) {
            if (NT == 0) *reinterpret_cast<f32x3_t *>(d + g * TILE * 3) = keep[g];
            else __builtin_nontemporal_store(keep[g], reinterpret_cast<f32x3_t *>(d + g * TILE * 3));
        }
        d += GROUP * TILE * 3;
    }
}

// 16-line tiles, 192-byte rows, but the wave parks four trips' points in LDS and stores them as whole lines: per tile 4 slots = 768 B =
// 48 float4, lane j of the tile's 16 stores float4 j, j + 16, j + 32 -> every store instruction writes 4 runs of 256 contiguous bytes
template <int NT>
__global__ __launch_bounds__(256) void k_store12_grouped(float *dst, size_t ntiles, int nslots, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) float stage[4][4][192];           // [wave][tile of the wave][4 slots x 16 lines x 3]
    const size_t li = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li / 16 >= ntiles) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tl = lane >> 4, j = lane & 15;
    float *tile = dst + (li / 16) * ((size_t)nslots * 48);
    float *st = &stage[wave][tl][0];
    float x = (float)seed, y = (float)threadIdx.x, z = 1.0f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int t = 0; t < nslots; t++) {
        x += 1.0f;
        float *p = st + (t & 3) * 48 + j * 3;
        p[0] = x; p[1] = y; p[2] = z;
        if ((t & 3) == 3) {
            __builtin_amdgcn_wave_barrier();
            f4 *g = reinterpret_cast<f4 *>(tile + (size_t)(t - 3) * 48);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const f4 q = reinterpret_cast<const f4 *>(st)[j + 16 * k];
                if (NT) __builtin_nontemporal_store(q, g + j + 16 * k); else g[j + 16 * k] = q;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

int main() {
    const size_t NB = (size_t)4 << 30;
    void *buf;
    CK(hipMalloc(&buf, NB));
    CK(hipMemset(buf, 0, NB));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, size_t bytes, auto launch) {
        for (int i = 0; i < 2; i++) launch();
        std::vector<float> ms;
        for (int r = 0; r < 7; r++) {
            hipEventRecord(e0, 0);
            launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-52s %8.3f ms  %6.2f TB/s (median of 7; best %.2f)\n", name, ms[3], bytes / ms[3] * 1e-9, bytes / ms[0] * 1e-9);
        fflush(stdout);
    };
    u32x4_t *d16 = (u32x4_t *)buf;
    for (int grid : {1024, 2048, 4096, 16384}) {
        char nm[128];
        snprintf(nm, sizeof nm, "dwordx4 plain, grid-stride, %d x 256", grid);
        time(nm, NB, [&] { hipLaunchKernelGGL(k_store16<0>, dim3(grid), dim3(256), 0, 0, d16, NB / 16, 1u); });
        snprintf(nm, sizeof nm, "dwordx4 nt, grid-stride, %d x 256", grid);
        time(nm, NB, [&] { hipLaunchKernelGGL(k_store16<1>, dim3(grid), dim3(256), 0, 0, d16, NB / 16, 1u); });
    }
    time("dwordx4 sc1, grid-stride, 4096 x 256", NB, [&] { hipLaunchKernelGGL(k_store16<2>, dim3(4096), dim3(256), 0, 0, d16, NB / 16, 1u); });
    time("dwordx4 sc0 sc1 nt, grid-stride, 4096 x 256", NB, [&] { hipLaunchKernelGGL(k_store16<3>, dim3(4096), dim3(256), 0, 0, d16, NB / 16, 1u); });
    time("dwordx4 plain, blocked, 4096 x 256", NB, [&] { hipLaunchKernelGGL(k_store16_blocked<0>, dim3(4096), dim3(256), 0, 0, d16, NB / 16, 1u); });
    time("dwordx4 nt, blocked, 4096 x 256", NB, [&] { hipLaunchKernelGGL(k_store16_blocked<1>, dim3(4096), dim3(256), 0, 0, d16, NB / 16, 1u); });
    time("dwordx4 nt, blocked, 65536 x 256", NB, [&] { hipLaunchKernelGGL(k_store16_blocked<1>, dim3(65536), dim3(256), 0, 0, d16, NB / 16, 1u); });
    time("hipMemsetAsync", NB, [&] { hipMemsetAsync(buf, 1, NB, 0); });
    {   // the tracer's scratch: 998 592 lines x 144 slots x 12 B = 1.73 GB
        const size_t nl = 998592; const int nslots = 144;
        const size_t bytes = nl * (size_t)nslots * 12;
        const dim3 g((nl + 255) / 256), b(256);
        float *f = (float *)buf;
        time("12 B per lane, 16-line tiles (192-B rows), nt", bytes, [&] { hipLaunchKernelGGL((k_store12_tiles<1, 16>), g, b, 0, 0, f, nl / 16, nslots, 1u); });
        time("12 B per lane, 16-line tiles (192-B rows), plain", bytes, [&] { hipLaunchKernelGGL((k_store12_tiles<0, 16>), g, b, 0, 0, f, nl / 16, nslots, 1u); });
        time("12 B per lane, 32-line tiles (384-B rows), nt", bytes, [&] { hipLaunchKernelGGL((k_store12_tiles<1, 32>), g, b, 0, 0, f, nl / 32, nslots, 1u); });
        time("12 B per lane, 32-line tiles (384-B rows), plain", bytes, [&] { hipLaunchKernelGGL((k_store12_tiles<0, 32>), g, b, 0, 0, f, nl / 32, nslots, 1u); });
        time("12 B per lane, 64-line tiles (768-B rows), nt", bytes, [&] { hipLaunchKernelGGL((k_store12_tiles<1, 64>), g, b, 0, 0, f, nl / 64, nslots, 1u); });
        time("12 B per lane, 64-line tiles (768-B rows), plain", bytes, [&] { hipLaunchKernelGGL((k_store12_tiles<0, 64>), g, b, 0, 0, f, nl / 64, nslots, 1u); });
        time("12 B per lane, 16-line tiles, trips stored in pairs, nt", bytes, [&] { hipLaunchKernelGGL((k_store12_paired<1, 16, 2>), g, b, 0, 0, f, nl / 16, nslots, 1u); });
        time("12 B per lane, 16-line tiles, trips stored in fours, nt", bytes, [&] { hipLaunchKernelGGL((k_store12_paired<1, 16, 4>), g, b, 0, 0, f, nl / 16, nslots, 1u); });
        time("12 B per lane, 32-line tiles, trips stored in pairs, nt", bytes, [&] { hipLaunchKernelGGL((k_store12_paired<1, 32, 2>), g, b, 0, 0, f, nl / 32, nslots, 1u); });
        time("16-line tiles, four trips parked in LDS, whole-line dwordx4 stores, nt", bytes, [&] { hipLaunchKernelGGL((k_store12_grouped<1>), g, b, 0, 0, f, nl / 16, nslots, 1u); });
        time("16-line tiles, four trips parked in LDS, whole-line dwordx4 stores, plain", bytes, [&] { hipLaunchKernelGGL((k_store12_grouped<0>), g, b, 0, 0, f, nl / 16, nslots, 1u); });
        time("3 x 4 B per lane, 16-line tiles, planes, nt", bytes, [&] { hipLaunchKernelGGL((k_store4x3_tiles<1, 16>), g, b, 0, 0, f, nl / 16, nslots, 1u); });
        time("3 x 4 B per lane, 32-line tiles, planes, nt", bytes, [&] { hipLaunchKernelGGL((k_store4x3_tiles<1, 32>), g, b, 0, 0, f, nl / 32, nslots, 1u); });
        time("3 x 4 B per lane, 64-line tiles, planes, nt", bytes, [&] { hipLaunchKernelGGL((k_store4x3_tiles<1, 64>), g, b, 0, 0, f, nl / 64, nslots, 1u); });
        time("3 x 4 B per lane, 64-line tiles, planes, plain", bytes, [&] { hipLaunchKernelGGL((k_store4x3_tiles<0, 64>), g, b, 0, 0, f, nl / 64, nslots, 1u); });
    }
    {   // smaller targets: does a 1.7-GB / 256-MB / 64-MB region take writes faster than 4 GiB?
        for (size_t mb : {64, 256, 1024}) {
            char nm[128]; snprintf(nm, sizeof nm, "dwordx4 nt, grid-stride 4096 x 256, %zu MiB region x (4096/MiB) passes", mb);
            const size_t nb = mb << 20; const int passes = (int)(NB / nb);
            time(nm, nb * passes, [&] { for (int p = 0; p < passes; p++) hipLaunchKernelGGL(k_store16<1>, dim3(4096), dim3(256), 0, 0, d16, nb / 16, 1u); });
        }
    }
    return 0;
}
