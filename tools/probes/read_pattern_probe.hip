// Does the SHAPE of the contraction kernels' sample requests bound the read rate?  The fused GQI kernel reads its samples by
// `buffer_load_dwordx4 .. lds`: a wave instruction covers 8 frames (rows, 11 MB apart) x 128 B (the wave's 32 voxels).  A timing
// build of the kernel without MFMAs read at 4.2 TB/s; whole-line streaming reads reach 6.5 TB/s on this chip.  This probe issues
// nothing but the requests of the stage loop -- 8 waves per CU, 2 requests per wave and 16-frame stage, 17 stages per 256-voxel
// item, a barrier per stage, the requests of a stage waited for one stage later (vmcnt(2)) -- in two shapes:
//   A  lane = (frame, voxel quad of the wave): 8 rows x 128 B per instruction        (what the kernel does)
//   B  lane = voxel quad of the ITEM, one frame per instruction: 1 row x 1 KB          (the workgroup's 16 rows, 2 per wave)
// and prints GB/s.  hipcc --offload-arch=gfx950 -O3 -o read_pattern_probe read_pattern_probe.hip && ./read_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4_t __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void reads(const float *S, long nvox, int K, int sleep_cycles, unsigned *sink) {
    __shared__ __attribute__((aligned(16))) char tile[2][16 * 1024];      // two stage slots: [16 frames][256 voxels] floats
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const long nitems = nvox / 256;
    const int nst = (K + 15) / 16;
    const unsigned row_bytes = (unsigned)(nvox * 4);
    const unsigned lds0 = (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char *)&tile[0][0]);
    unsigned acc = 0;
    int g = 0;
    for (long it = wslot; ; it += nslot) {
        const long item = it * 8 + xcd;                 // items dealt XCD by XCD, like the kernel
        if (item >= nitems) break;
        for (int t = 0; t < nst; t++, g++) {
            const int rem = K - t * 16;
            const unsigned long b = (unsigned long)S + (unsigned long)t * 16 * row_bytes;
            i32x4_t r;
            r[0] = (int)(unsigned)b; r[1] = (int)(unsigned)((b >> 32) & 0xffffu);
            const unsigned long span = (unsigned long)(rem > 16 ? 16 : rem) * row_bytes;
            r[2] = (int)(span > 0xffffffffUL ? 0xffffffffu : (unsigned)span); r[3] = 0x00020000;
            unsigned v0, v1, d0, d1;
            if (SHAPE == 0) {                            // 8 rows x 128 B
                const unsigned q = (unsigned)(item * 1024 + wave * 128 + (lane & 7) * 16);
                v0 = q + (unsigned)(lane >> 3) * row_bytes; v1 = v0 + 8u * row_bytes;
                d0 = lds0 + (unsigned)(g & 1) * 16384u + (unsigned)wave * 2048u; d1 = d0 + 1024u;
            } else {                                     // 1 row x 1 KB
                const unsigned q = (unsigned)(item * 1024 + lane * 16);
                v0 = q + (unsigned)(2 * wave) * row_bytes; v1 = v0 + row_bytes;
                d0 = lds0 + (unsigned)(g & 1) * 16384u + (unsigned)(2 * wave) * 1024u; d1 = d0 + 1024u;
            }
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" :: "s"(d0), "v"(v0), "s"(r) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" :: "s"(d1), "v"(v1), "s"(r) : "memory");
            for (int k = 0; k < sleep_cycles / 64; k++) __builtin_amdgcn_s_sleep(1);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // the previous stage's requests have landed
            __builtin_amdgcn_s_barrier();
            acc += reinterpret_cast<const unsigned *>(tile[(g + 1) & 1])[lane + 64 * wave];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const long nvox = 140L * 140 * 140;
    const int K = 270;
    float *S; unsigned *sink;
    hipMalloc(&S, (size_t)nvox * K * 4);
    hipMalloc(&sink, 64);
    hipMemset(S, 0, (size_t)nvox * K * 4);
    for (int sleep : {0, 1024, 2048, 3072}) for (int shape = 0; shape < 2; shape++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0, 0);
            if (shape == 0) hipLaunchKernelGGL((reads<0>), dim3(256), dim3(512), 0, 0, S, nvox, K, sleep, sink);
            else hipLaunchKernelGGL((reads<1>), dim3(256), dim3(512), 0, 0, S, nvox, K, sleep, sink);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("{\"shape\": \"%s\", \"stage_filler_cycles\": %d, \"ms\": %.3f, \"read_TBps\": %.2f}\n", shape == 0 ? "8 rows x 128 B" : "1 row x 1 KB", sleep, best,
               (double)nvox * K * 4 / (best * 1e-3) / 1e12);
    }
    return 0;
}
