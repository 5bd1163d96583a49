// energy_probe.hip -- Joules per unit of work for the ingredients of the contraction kernels, on one MI355X, read from the board's
// accumulated-energy counter (rocm_smi: rsmi_dev_energy_count_get, 15.3 uJ per count) around >= S seconds of back-to-back launches
// of ONE ingredient at a time on random data:
//   idle / spin                       what the board draws with nothing / with every CU occupied by sleeping waves
//   hbm_read / hbm_write / hbm_rw     non-temporal streams (rw: in the GQI step's 1084 : 1332 byte proportion)
//   mfma [duty]                       v_mfma_f32_32x32x16_f16 from registers, 2 waves per SIMD (the product kernels' occupancy)
//   lds_read [duty]                   ds_read_b128, the fragment pattern of the stage loop (a lane reads 16 B at lane*16 + piece*1024)
//   mfma_lds                          the stage loop's MFMA block: per block 2 fragment reads + 3 MFMAs (additivity check)
//   valu [duty]                       the split / epilogue instruction mix (fma_mix, max3, cmp + addc, ldexp, cvt) on random registers
//   ldsdma                            buffer_load_dwordx4 .. lds of an L2-resident image: the piece fetch L2 -> LDS
//   lds_write_read                    ds_write_b32 + ds_read_b128 of a private tile: the epilogue's transposition
// Every kernel stamps s_memtime / s_memrealtime around its loop (probe only; the product never does): the in-kernel clock the
// ingredient ran at.  Output: one JSON object per mode on stdout.  build: see tools/energy_model.py
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { uint64_t c0, c1, r0, r1; };
__device__ __forceinline__ void stamp_begin(Stamp &s) { s.c0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void stamp_end(Stamp &s, Stamp *out) {
    s.c1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__device__ __forceinline__ uint32_t rnd(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
// duty: after every trip sleep `slp` x 64 clocks (0: none)
#define DUTY_SLEEP(slp) do { for (int s_ = 0; s_ < (slp); s_++) __builtin_amdgcn_s_sleep(16); } while (0)

__global__ __launch_bounds__(512, 2) void k_spin(int iters, Stamp *st) {
    Stamp s; stamp_begin(s);
    for (int i = 0; i < iters; i++) __builtin_amdgcn_s_sleep(64);
    stamp_end(s, st);
}

// ---- HBM streams ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void k_hbm_read(const u32x4_t *src, size_t n16, uint32_t *sink, Stamp *st) {
    Stamp s; stamp_begin(s);
    u32x4_t acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {      // four requests in flight per lane
        const u32x4_t v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + stride);
        const u32x4_t v2 = __builtin_nontemporal_load(src + i + 2 * stride), v3 = __builtin_nontemporal_load(src + i + 3 * stride);
        acc ^= v0 ^ v1 ^ v2 ^ v3;
    }
    for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(src + i);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[threadIdx.x] = 1;
    stamp_end(s, st);
}
__global__ __launch_bounds__(512, 2) void k_hbm_write(u32x4_t *dst, size_t n16, uint32_t seed, Stamp *st) {
    Stamp s; stamp_begin(s);
    uint32_t x = rnd(seed + blockIdx.x * 512u + threadIdx.x + 1u);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        x = rnd(x);
        u32x4_t v = {x, x * 0x9E3779B9u, x ^ 0x5bd1e995u, x + 0x7f4a7c15u};
        __builtin_nontemporal_store(v, dst + i);
    }
    stamp_end(s, st);
}
// read nr16 and write nw16 16-byte units (interleaved per workgroup the way a work item reads its frames and writes its rows)
__global__ __launch_bounds__(512, 2) void k_hbm_rw(const u32x4_t *src, size_t nr16, u32x4_t *dst, size_t nw16, Stamp *st) {
    Stamp s; stamp_begin(s);
    u32x4_t acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nmax = nr16 > nw16 ? nr16 : nw16;
    for (size_t i = i0; i < nmax; i += stride) {
        if (i < nr16) acc ^= __builtin_nontemporal_load(src + i);
        if (i < nw16) { u32x4_t v = acc; v[0] += (uint32_t)i; __builtin_nontemporal_store(v, dst + i); }
    }
    stamp_end(s, st);
}

// ---- matrix cores from registers --------------------------------------------------------------------------------------------------
__device__ __forceinline__ f16x8_t rnd_f16x8(uint32_t &x) {
    f16x8_t v;
    for (int i = 0; i < 8; i++) { x = rnd(x); v[i] = (_Float16)(((int)(x & 0xffffu) - 32768) * (1.0f / 32768.0f)); }
    return v;
}
__global__ __launch_bounds__(512, 2) void k_mfma(int iters, int slp, float *sink, Stamp *st) {
    uint32_t x = rnd(threadIdx.x * 7919u + blockIdx.x * 104729u + 17u);
    f16x8_t a[4], b[2];
    for (int i = 0; i < 4; i++) a[i] = rnd_f16x8(x);
    for (int i = 0; i < 2; i++) b[i] = rnd_f16x8(x);
    f32x16 acc[10];
    for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
    Stamp s; stamp_begin(s);
    for (int t = 0; t < iters; t++) {
#pragma unroll
        for (int m = 0; m < 10; m++) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(m + 1) & 3], b[0], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[1], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[0], acc[m], 0, 0, 0);
        }
        DUTY_SLEEP(slp);
    }
    stamp_end(s, st);
    float z = 0;
    for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) z += acc[m][r];
    if (z == 123.456f) sink[threadIdx.x] = z;
}

// ---- LDS fragment reads -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void k_lds_read(int iters, int slp, uint32_t *sink, Stamp *st) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[20 * 256 * 2];          // two stage buffers of 20 KiB
    uint32_t x = rnd(threadIdx.x * 7919u + blockIdx.x * 104729u + 17u);
    for (int i = threadIdx.x; i < 20 * 256 * 2; i += 512) { x = rnd(x); lds[i] = x; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4_t acc = {0, 0, 0, 0};
    Stamp s; stamp_begin(s);
    for (int t = 0; t < iters; t++) {
        const u32x4_t *L = reinterpret_cast<const u32x4_t *>(lds + (t & 1) * 20 * 256) + lane;
#pragma unroll
        for (int p = 0; p < 20; p++) acc ^= L[p * 64];
        asm volatile("" : "+v"(acc));
        DUTY_SLEEP(slp);
    }
    stamp_end(s, st);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[threadIdx.x] = 1;
}

// ---- the stage loop's MFMA block: fragments from LDS ------------------------------------------------------------------------------
template <int NREAD>   // NREAD fragment reads per block (2 = the product's 32-voxel wave; 1 = a wave that uses every fragment for two voxel tiles)
__global__ __launch_bounds__(512, 2) void k_mfma_lds(int iters, int slp, float *sink, Stamp *st) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[20 * 512 * 2];
    uint32_t x = rnd(threadIdx.x * 7919u + blockIdx.x * 104729u + 17u);
    for (int i = threadIdx.x; i < 20 * 512 * 2; i += 512) { x = rnd(x); lds[i] = (_Float16)(((int)(x & 0xffffu) - 32768) * (1.0f / 32768.0f)); }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f16x8_t b[4];
    for (int i = 0; i < 4; i++) b[i] = rnd_f16x8(x);
    f32x16 acc[10];
    for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
    Stamp s; stamp_begin(s);
    for (int t = 0; t < iters; t++) {
        const f16x8_t *LA = reinterpret_cast<const f16x8_t *>(lds + (t & 1) * 20 * 512) + lane;
        if (NREAD == 2) {
#pragma unroll
            for (int m = 0; m < 10; m++) {
                const f16x8_t a1 = LA[(10 + m) * 64], a0 = LA[m * 64];
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[0], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[1], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[0], acc[m], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 5; m++) {                // 5 blocks x 2 voxel tiles: the same 30 MFMAs, half the fragment reads
                const f16x8_t a1 = LA[(10 + m) * 64], a0 = LA[m * 64];
                acc[2 * m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[0], acc[2 * m], 0, 0, 0);
                acc[2 * m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[1], acc[2 * m], 0, 0, 0);
                acc[2 * m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[0], acc[2 * m], 0, 0, 0);
                acc[2 * m + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[2], acc[2 * m + 1], 0, 0, 0);
                acc[2 * m + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[3], acc[2 * m + 1], 0, 0, 0);
                acc[2 * m + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[2], acc[2 * m + 1], 0, 0, 0);
            }
        }
        DUTY_SLEEP(slp);
    }
    stamp_end(s, st);
    float z = 0;
    for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) z += acc[m][r];
    if (z == 123.456f) sink[threadIdx.x] = z;
}

// ---- vector ALU: the split / epilogue mix ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void k_valu(int iters, int slp, float *sink, Stamp *st) {
    uint32_t x = rnd(threadIdx.x * 7919u + blockIdx.x * 104729u + 17u);
    float v[16];
    for (int i = 0; i < 16; i++) { x = rnd(x); v[i] = ((int)(x & 0xffffu) - 32768) * (1.0f / 32768.0f); }
    uint32_t flags = 0;
    Stamp s; stamp_begin(s);
    for (int t = 0; t < iters; t++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {                     // 8 x 32 = 256 vector instructions per trip
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                uint32_t h;
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(v[i]), "v"(v[i + 1]));
                const float m3 = __builtin_fmaxf(__builtin_fmaxf(v[i], v[(i + 3) & 15]), v[(i + 5) & 15]);
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(v[i + 1]) : "v"(v[i + 1]), "v"(m3), "v"(v[(i + 7) & 15]));
                flags = flags * 2u + (v[i] > m3 ? 1u : 0u);
                v[i] = __builtin_fmaf(v[i], 0.999f, __uint_as_float((h & 0x007fffffu) | 0x3c000000u) - 0.0078125f);
            }
        }
        DUTY_SLEEP(slp);
    }
    stamp_end(s, st);
    float z = (float)flags;
    for (int i = 0; i < 16; i++) z += v[i];
    if (z == 123.456f) sink[threadIdx.x] = z;
}

// ---- L2 -> LDS by LDS-DMA (the matrix pieces: every workgroup fetches the same 340-KB image stage by stage) -------------------------
__global__ __launch_bounds__(512, 2) void k_ldsdma(const char *img, uint32_t img_bytes, int iters, uint32_t *sink, Stamp *st) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 20 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds_l = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char *)lds);
    i32x4_t rs;
    const uint64_t b = reinterpret_cast<uint64_t>(img);
    rs[0] = (int)(uint32_t)b; rs[1] = (int)(uint32_t)((b >> 32) & 0xffffu); rs[2] = (int)img_bytes; rs[3] = 0x00020000;
    const uint32_t a_off = (uint32_t)lane * 16;
    const uint32_t nst = img_bytes / (20 * 1024);
    Stamp s; stamp_begin(s);
    for (int t = 0; t < iters; t++) {
        const uint32_t stg = (uint32_t)t % nst, buf = (uint32_t)t & 1u;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            int p = wave + i * 8; p = p < 20 ? p : 19;
            const uint32_t d = lds_l + buf * 20480u + (uint32_t)p * 1024u;
            const uint32_t so = __builtin_amdgcn_readfirstlane(stg * 20480u + (uint32_t)p * 1024u);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(d), "v"(a_off), "s"(rs), "s"(so) : "memory");
        }
        if ((t & 3) == 3) { __builtin_amdgcn_s_waitcnt(0x0F70); }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    stamp_end(s, st);
    if (reinterpret_cast<uint32_t *>(lds)[threadIdx.x] == 0x12345u) sink[threadIdx.x] = 1;
}

// ---- the epilogue's transposition: ds_write_b32 x 16 then ds_read_b128 x 4 of a private 2-KiB tile ----------------------------------
__global__ __launch_bounds__(512, 2) void k_lds_wr(int iters, int slp, uint32_t *sink, Stamp *st) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[8 * 512];
    uint32_t x = rnd(threadIdx.x * 7919u + blockIdx.x * 104729u + 17u);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *T = lds + wave * 512;
    uint32_t v[8];
    for (int i = 0; i < 8; i++) { x = rnd(x); v[i] = x; }
    u32x4_t acc = {0, 0, 0, 0};
    Stamp s; stamp_begin(s);
    for (int t = 0; t < iters; t++) {
#pragma unroll
        for (int i = 0; i < 8; i++) T[i * 64 + ((lane + i) & 63)] = v[i] + (uint32_t)t;
#pragma unroll
        for (int i = 0; i < 2; i++) acc ^= reinterpret_cast<const u32x4_t *>(T)[i * 64 + lane];
        asm volatile("" : "+v"(acc));
        DUTY_SLEEP(slp);
    }
    stamp_end(s, st);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[threadIdx.x] = 1;
}

// ---- the step's unavoidable work and nothing else: an empirical floor ------------------------------------------------------------------
// One launch = one GQI step's algorithmic HBM bytes (2.975 GB read + 3.655 GB written, non-temporal) AND its executed matrix-core
// work (three fp16 piece products: 1.433 PFLOP of v_mfma_f32_32x32x16_f16), side by side on every CU (two workgroups each): waves 0-3 of a workgroup issue
// the MFMAs from registers (FRAG: fragments re-read from LDS as the stage loop does, 20 KiB per 30 MFMAs), waves 4-7 stream.  No
// sample split, no epilogue, no lists: what a kernel made of nothing but the unavoidable ingredients costs under the board's cap.
template <bool FRAG>
__global__ __launch_bounds__(512, 2) void k_essential(const u32x4_t *src, size_t nr16, u32x4_t *dst, size_t nw16, int mfma_trips, float *sink, Stamp *st) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[20 * 512];
    uint32_t x = rnd(threadIdx.x * 7919u + blockIdx.x * 104729u + 17u);
    for (int i = threadIdx.x; i < 20 * 512; i += 512) { x = rnd(x); lds[i] = (_Float16)(((int)(x & 0xffffu) - 32768) * (1.0f / 32768.0f)); }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    Stamp s; stamp_begin(s);
    if (wave < 4) {
        f16x8_t a[4], b[2];
        for (int i = 0; i < 4; i++) a[i] = rnd_f16x8(x);
        for (int i = 0; i < 2; i++) b[i] = rnd_f16x8(x);
        f32x16 acc[10];
        for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
        const f16x8_t *LA = reinterpret_cast<const f16x8_t *>(lds) + lane;
        for (int t = 0; t < mfma_trips; t++) {
#pragma unroll
            for (int m = 0; m < 10; m++) {
                const f16x8_t a1 = FRAG ? LA[(10 + m) * 64] : a[(m + 1) & 3], a0 = FRAG ? LA[m * 64] : a[m & 3];
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[0], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[1], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[0], acc[m], 0, 0, 0);
            }
        }
        float z = 0;
        for (int m = 0; m < 10; m++) for (int r = 0; r < 16; r++) z += acc[m][r];
        if (z == 123.456f) sink[threadIdx.x] = z;
    } else {
        // eight 16-byte requests in flight per lane (a stream needs ~12 MB in flight chip-wide: latency x rate)
        u32x4_t acc = {0, 0, 0, 0};
        const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + (threadIdx.x - 256);
        const size_t nmax = nr16 > nw16 ? nr16 : nw16;
        for (size_t i = i0; i < nmax; i += 8 * stride) {
            u32x4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { v[u] = u32x4_t{0, 0, 0, 0}; if (i + u * stride < nr16) v[u] = __builtin_nontemporal_load(src + i + u * stride); }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                acc ^= v[u];
                if (i + u * stride < nw16) { u32x4_t w = v[u]; w[0] += (uint32_t)i; __builtin_nontemporal_store(w, dst + i + u * stride); }
            }
        }
        if ((acc[0] ^ acc[1]) == 0x12345u) sink[threadIdx.x] = 1;
    }
    stamp_end(s, st);
}

// ---- host ---------------------------------------------------------------------------------------------------------------------------
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double energy_j() {
    uint64_t c = 0, ts = 0; float res = 0;
    if (rsmi_dev_energy_count_get(0, &c, &res, &ts) != RSMI_STATUS_SUCCESS) return -1.0;
    return (double)c * (double)res * 1e-6;
}
static double power_w() { uint64_t p = 0; RSMI_POWER_TYPE ty; if (rsmi_dev_power_get(0, &p, &ty) != RSMI_STATUS_SUCCESS) return -1; return p * 1e-6; }
static double sclk_mhz() { rsmi_frequencies_t f; if (rsmi_dev_gpu_clk_freq_get(0, RSMI_CLK_TYPE_SYS, &f) != RSMI_STATUS_SUCCESS) return -1; return f.frequency[f.current] * 1e-6; }

struct Result { double secs, joules, gpu_ms, work, clk_ghz, p_mid, sclk_mid; int launches; };

template <class F>
static Result run_mode(double seconds, Stamp *d_st, int nblk, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // warm: 0.5 s untimed (the chip leaves its idle state, the firmware settles the clock)
    double t0 = now_s();
    while (now_s() - t0 < 0.5) { for (int i = 0; i < 4; i++) launch(); CK(hipDeviceSynchronize()); }
    Result r{}; r.p_mid = r.sclk_mid = 0; int nmid = 0;
    const double ej0 = energy_j(); t0 = now_s();
    CK(hipEventRecord(e0));
    double next_sample = t0 + 0.25;
    while (now_s() - t0 < seconds) {
        for (int i = 0; i < 4; i++) { launch(); r.launches++; }
        CK(hipStreamSynchronize(0));
        if (now_s() >= next_sample) { r.p_mid += power_w(); r.sclk_mid += sclk_mhz(); nmid++; next_sample += 0.25; }
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    const double t1 = now_s(), ej1 = energy_j();
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    r.secs = t1 - t0; r.joules = ej1 - ej0; r.gpu_ms = ms;
    if (nmid) { r.p_mid /= nmid; r.sclk_mid /= nmid; }
    std::vector<Stamp> st(nblk);
    CK(hipMemcpy(st.data(), d_st, nblk * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> g;
    for (auto &s : st) if (s.r1 > s.r0) g.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 0.1);
    std::sort(g.begin(), g.end());
    r.clk_ghz = g.empty() ? 0.0 : g[g.size() / 2];
    return r;
}

static void report(const char *mode, const char *unit, double work_per_launch, const Result &r, double p_idle, const char *note) {
    const double work = work_per_launch * r.launches, p = r.joules / r.secs;
    printf("{\"mode\": \"%s\", \"seconds\": %.3f, \"joules\": %.2f, \"watts\": %.1f, \"watts_smi_mean\": %.1f, \"sclk_smi_mhz\": %.0f, "
           "\"in_kernel_clock_ghz\": %.3f, \"launches\": %d, \"gpu_busy_frac\": %.3f, \"unit\": \"%s\", \"work\": %.6g, \"rate_per_s\": %.6g, "
           "\"pj_per_unit_total\": %.4g, \"pj_per_unit_above_idle\": %.4g, \"note\": \"%s\"}\n",
           mode, r.secs, r.joules, p, r.p_mid, r.sclk_mid, r.clk_ghz, r.launches, r.gpu_ms * 1e-3 / r.secs, unit, work, work / r.secs,
           work > 0 ? r.joules / work * 1e12 : 0.0, work > 0 ? (r.joules - p_idle * r.secs) / work * 1e12 : 0.0, note);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const double sec = argc > 1 ? atof(argv[1]) : 3.0;
    const char *only = argc > 2 ? argv[2] : "";
    auto want = [&](const char *m) { return !only[0] || strstr(only, m) != nullptr; };
    if (rsmi_init(0) != RSMI_STATUS_SUCCESS) { fprintf(stderr, "rsmi_init failed\n"); return 2; }
    if (energy_j() < 0) { fprintf(stderr, "no energy counter\n"); return 2; }
    CK(hipSetDevice(0));
    const int nblk = 512;                                  // 2 workgroups of 8 waves per CU = 4 waves per SIMD .. see OCC below
    Stamp *d_st; CK(hipMalloc(&d_st, 4096 * sizeof(Stamp)));
    uint32_t *d_sink; CK(hipMalloc(&d_sink, 4096));
    // (the product kernels run ONE 8-wave workgroup per CU = 2 waves per SIMD: grid 256 for the on-chip ingredients)
    const int OCC = 256;

    // idle: nothing runs (the context exists)
    double p_idle = 0;
    {
        std::this_thread::sleep_for(std::chrono::milliseconds(1500));
        const double e0 = energy_j(), t0 = now_s();
        std::this_thread::sleep_for(std::chrono::milliseconds((int)(sec * 1000)));
        const double e1 = energy_j(), t1 = now_s();
        p_idle = (e1 - e0) / (t1 - t0);
        printf("{\"mode\": \"idle\", \"seconds\": %.3f, \"joules\": %.2f, \"watts\": %.1f, \"watts_smi\": %.1f, \"sclk_smi_mhz\": %.0f}\n", t1 - t0, e1 - e0, p_idle, power_w(), sclk_mhz());
        fflush(stdout);
    }
    if (want("spin")) {
        Result r = run_mode(sec, d_st, OCC, [&] { hipLaunchKernelGGL(k_spin, dim3(OCC), dim3(512), 0, 0, 20000, d_st); });
        report("spin", "s", 0.0, r, p_idle, "every CU holds 8 waves that sleep (s_sleep 64): the board with its clocks up and nothing switching");
    }
    const size_t NB = (size_t)4 << 30;                     // 4 GiB stream buffers
    u32x4_t *d_a = nullptr, *d_b = nullptr;
    if (want("hbm")) {
        CK(hipMalloc(&d_a, NB)); CK(hipMalloc(&d_b, NB));
        hipLaunchKernelGGL(k_hbm_write, dim3(2048), dim3(512), 0, 0, d_a, NB / 16, 1u, d_st);
        hipLaunchKernelGGL(k_hbm_write, dim3(2048), dim3(512), 0, 0, d_b, NB / 16, 2u, d_st);
        CK(hipDeviceSynchronize());
        Result r = run_mode(sec, d_st, 2048, [&] { hipLaunchKernelGGL(k_hbm_read, dim3(2048), dim3(512), 0, 0, d_a, NB / 16, d_sink, d_st); });
        report("hbm_read", "B", (double)NB, r, p_idle, "non-temporal dwordx4 loads of 4 GiB of random words, grid-stride");
        r = run_mode(sec, d_st, 2048, [&] { hipLaunchKernelGGL(k_hbm_write, dim3(2048), dim3(512), 0, 0, d_b, NB / 16, 3u, d_st); });
        report("hbm_write", "B", (double)NB, r, p_idle, "non-temporal dwordx4 stores of 4 GiB of random words");
        const size_t nr = (size_t)(2.975e9 / 16), nw = (size_t)(3.655e9 / 16);
        r = run_mode(sec, d_st, 2048, [&] { hipLaunchKernelGGL(k_hbm_rw, dim3(2048), dim3(512), 0, 0, d_a, nr, d_b, nw, d_st); });
        report("hbm_rw_gqi", "B", (double)(nr + nw) * 16.0, r, p_idle, "the GQI step's bytes: 2.975 GB read + 3.655 GB written per launch, non-temporal");
        CK(hipFree(d_a)); CK(hipFree(d_b));
    }
    if (want("essential")) {
        CK(hipMalloc(&d_a, NB)); CK(hipMalloc(&d_b, NB));
        hipLaunchKernelGGL(k_hbm_write, dim3(2048), dim3(512), 0, 0, d_a, NB / 16, 1u, d_st);
        CK(hipDeviceSynchronize());
        const size_t nr = (size_t)(2.975e9 / 16), nw = (size_t)(3.655e9 / 16);
        const double flops = 3.0 * 2.0 * 320 * 272 * 2744000.0;                     // executed per step
        const int EG = 2 * OCC;                                                      // two workgroups per CU: 8 MFMA waves (2 per SIMD, the product's occupancy) + 8 streaming waves
        const int trips = (int)(flops / (30.0 * 2.0 * 32 * 32 * 16) / (EG * 4.0) + 0.5);
        for (int frag = 0; frag < 2; frag++) {
            Result r = frag ? run_mode(sec, d_st, EG, [&] { hipLaunchKernelGGL(k_essential<true>, dim3(EG), dim3(512), 0, 0, d_a, nr, d_b, nw, trips, (float *)d_sink, d_st); })
                            : run_mode(sec, d_st, EG, [&] { hipLaunchKernelGGL(k_essential<false>, dim3(EG), dim3(512), 0, 0, d_a, nr, d_b, nw, trips, (float *)d_sink, d_st); });
            report(frag ? "essential_gqi_step_lds_fragments" : "essential_gqi_step", "step", 1.0, r, p_idle,
                   frag ? "one launch = the GQI step's HBM bytes + its executed MFMAs with the fragments re-read from LDS (20 KiB per 30 MFMAs), nothing else"
                        : "one launch = the GQI step's algorithmic HBM bytes (2.975 GB in, 3.655 GB out) + its executed MFMAs (1.433 PFLOP, operands in registers), nothing else");
        }
        CK(hipFree(d_a)); CK(hipFree(d_b));
    }
    const int slps[4] = {0, 1, 3, 8};
    if (want("mfma_reg")) {
        for (int s : slps) {
            const int it = 40000 / (1 + s);
            Result r = run_mode(sec, d_st, OCC, [&] { hipLaunchKernelGGL(k_mfma, dim3(OCC), dim3(512), 0, 0, it, s, (float *)d_sink, d_st); });
            char nm[64]; snprintf(nm, sizeof nm, "mfma_reg_sleep%d", s);
            report(nm, "flop", (double)it * 30.0 * 2.0 * 32 * 32 * 16 * OCC * 8, r, p_idle, "v_mfma_f32_32x32x16_f16 on random operands in registers, 8 waves per CU, s_sleep between trips of 30");
        }
    }
    if (want("lds_read")) {
        for (int s : slps) {
            const int it = 100000 / (1 + s);
            Result r = run_mode(sec, d_st, OCC, [&] { hipLaunchKernelGGL(k_lds_read, dim3(OCC), dim3(512), 0, 0, it, s, d_sink, d_st); });
            char nm[64]; snprintf(nm, sizeof nm, "lds_read_sleep%d", s);
            report(nm, "B", (double)it * 20.0 * 1024 * OCC * 8, r, p_idle, "ds_read_b128 of random data, the stage loop's fragment pattern: 20 x 1 KiB per wave and trip");
        }
    }
    if (want("mfma_lds")) {
        for (int s : {0, 3}) {
            const int it = 40000 / (1 + s);
            Result r = run_mode(sec, d_st, OCC, [&] { hipLaunchKernelGGL(k_mfma_lds<2>, dim3(OCC), dim3(512), 0, 0, it, s, (float *)d_sink, d_st); });
            char nm[64]; snprintf(nm, sizeof nm, "mfma_lds2_sleep%d", s);
            report(nm, "flop", (double)it * 30.0 * 2.0 * 32 * 32 * 16 * OCC * 8, r, p_idle, "the stage loop's MFMA block: 30 MFMAs + 20 fragment reads (20 KiB) per wave and trip");
            r = run_mode(sec, d_st, OCC, [&] { hipLaunchKernelGGL(k_mfma_lds<1>, dim3(OCC), dim3(512), 0, 0, it, s, (float *)d_sink, d_st); });
            snprintf(nm, sizeof nm, "mfma_lds1_sleep%d", s);
            report(nm, "flop", (double)it * 30.0 * 2.0 * 32 * 32 * 16 * OCC * 8, r, p_idle, "the same 30 MFMAs with every fragment used for two voxel tiles: 10 fragment reads (10 KiB) per wave and trip");
        }
    }
    if (want("valu")) {
        for (int s : slps) {
            const int it = 20000 / (1 + s);
            Result r = run_mode(sec, d_st, OCC, [&] { hipLaunchKernelGGL(k_valu, dim3(OCC), dim3(512), 0, 0, it, s, (float *)d_sink, d_st); });
            char nm[64]; snprintf(nm, sizeof nm, "valu_sleep%d", s);
            report(nm, "wave-instruction", (double)it * 360.0 * OCC * 8, r, p_idle, "360 vector instructions per wave and trip on random registers (ISA: 128 v_max3_f32, 64 v_fma_mixlo_f16, 112 v_and / v_or, 28 v_pk_add_f32, 28 v_pk_fma_f32)");
        }
    }
    if (want("ldsdma")) {
        char *img; const uint32_t IB = 17 * 20 * 1024;
        CK(hipMalloc(&img, IB));
        hipLaunchKernelGGL(k_hbm_write, dim3(64), dim3(512), 0, 0, (u32x4_t *)img, (size_t)IB / 16, 5u, d_st);
        CK(hipDeviceSynchronize());
        const int it = 20000;
        Result r = run_mode(sec, d_st, OCC, [&] { hipLaunchKernelGGL(k_ldsdma, dim3(OCC), dim3(512), 0, 0, img, IB, it, d_sink, d_st); });
        report("ldsdma_l2", "B", (double)it * 24.0 * 1024 * OCC, r, p_idle, "buffer_load_dwordx4 .. lds: every workgroup streams the same 340-KB image (L2 resident) into LDS, 24 KiB per trip");
        CK(hipFree(img));
    }
    if (want("lds_wr")) {
        const int it = 100000;
        Result r = run_mode(sec, d_st, OCC, [&] { hipLaunchKernelGGL(k_lds_wr, dim3(OCC), dim3(512), 0, 0, it, 0, d_sink, d_st); });
        report("lds_write_read", "B", (double)it * (8.0 * 256 + 2.0 * 1024) * OCC * 8, r, p_idle, "8 ds_write_b32 + 2 ds_read_b128 per wave and trip (the epilogue's transposition tile): bytes written + read");
    }
    return 0;
}
