// How fast does ONE CU get a burst of row stores acknowledged?  The contraction kernels' epilogue issues, per wave, 40
// global_store_dwordx4 of 8 rows x 128 B each (rows 11 MB apart), all 8 waves of the CU at once = 328 KB, and a wave's later
// loads return behind its own older stores (vmcnt counts both, in issue order).  This probe times such a burst from the first
// store to s_waitcnt vmcnt(0) with s_memtime, per wave:
//   pattern 0: 8 rows x 128 B per instruction (the epilogue's shape), rows `rowstride` bytes apart
//   pattern 1: 1 KB contiguous per instruction (one row), successive instructions 1 KB on
//   nt 0/1: default policy / non-temporal
//   active: number of workgroups (CUs) that store; the others exit at once
// hipcc --offload-arch=gfx950 -O3 -o store_drain_probe store_drain_probe.hip && ./store_drain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PAT, bool NT>
__global__ __launch_bounds__(512) void burst(char *out, long rowstride, int nstore, int active, int rounds, unsigned long long *cyc) {
    if ((int)blockIdx.x >= active) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long tot = 0;
    for (int r = 0; r < rounds; r++) {
        // every round writes a fresh 256-voxel column group: base advances by 1 KB per workgroup and round
        char *base = out + ((long)(r * gridDim.x + blockIdx.x)) * 1024;
        __syncthreads();
        unsigned long long t0, t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll 4
        for (int i = 0; i < nstore; i++) {
            char *p;
            if (PAT == 0) p = base + (long)(8 * i + (lane >> 3)) * rowstride + wave * 128 + (lane & 7) * 16;
            else p = base + (long)(i * 8 + wave) * rowstride + lane * 16;      // one row, 1 KB, per instruction (8 waves = 8 rows)
            const f4 v = {(float)i, (float)lane, (float)r, 1.0f};
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
            else *reinterpret_cast<f4 *>(p) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        tot += t1 - t0;
        __builtin_amdgcn_s_sleep(64);
    }
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = tot / rounds;
}
int main() {
    const long rowstride = 140L * 140 * 140 * 4;
    const int nrows = 321, nstore = 40, rounds = 20;
    char *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)rowstride * (nrows + 8));
    hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long));
    int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<unsigned long long> h(256 * 8);
    for (int pat = 0; pat < 2; pat++) for (int nt = 0; nt < 2; nt++) for (int active : {1, 8, 64, ncu}) {
        hipMemset(cyc, 0, 256 * 8 * 8);
        for (int rep = 0; rep < 2; rep++) {
            if (pat == 0 && nt == 0) hipLaunchKernelGGL((burst<0, false>), dim3(ncu), dim3(512), 0, 0, out, rowstride, nstore, active, rounds, cyc);
            if (pat == 0 && nt == 1) hipLaunchKernelGGL((burst<0, true>), dim3(ncu), dim3(512), 0, 0, out, rowstride, nstore, active, rounds, cyc);
            if (pat == 1 && nt == 0) hipLaunchKernelGGL((burst<1, false>), dim3(ncu), dim3(512), 0, 0, out, rowstride, nstore, active, rounds, cyc);
            if (pat == 1 && nt == 1) hipLaunchKernelGGL((burst<1, true>), dim3(ncu), dim3(512), 0, 0, out, rowstride, nstore, active, rounds, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> v(h.begin(), h.begin() + active * 8);
        std::sort(v.begin(), v.end());
        const double med = (double)v[v.size() / 2];
        printf("{\"pattern\": \"%s\", \"nt\": %d, \"active_cus\": %d, \"cycles_burst_median\": %.0f, \"cycles_max\": %llu, \"bytes_per_cu\": %d, \"bytes_per_clk_per_cu\": %.1f}\n",
               pat == 0 ? "8 rows x 128 B" : "1 row x 1 KB", nt, active, med, v.back(), nstore * 8 * 1024, nstore * 8 * 1024 / med);
    }
    return 0;
}
