// stream.hip — K6/K7: deterministic streamline tractography (nearest-voxel lookup + fixed-step Euler).
//
// Replaces StreamWork's mask/vector repack (stream.jl:95-145), stream_new_line (stream.jl:625-690),
// stream_new_point! (stream.jl:501-541), stream_pick_by_angle! (stream.jl:340-374) and the seed loop /
// len_min filter / concatenation of `stream` (stream.jl:761-787), for the non-LCM macro-scale path.
//
// Layout.  The orientation field is repacked to float4 [nvox][nvec] (xyz0): one aligned 16-byte gather
// per candidate vector per step; vectors of masked-out voxels are zero, which makes the reference's
// separate mask lookup (stream.jl:520) redundant (an all-zero voxel fails stream_pick_by_angle! the same
// way).  One lane integrates one (seed, sub-voxel offset) line, forward then backward.  Points go to a
// slot-major scratch [2*(len_max+2) slots][nlines][3]: forward step i -> slot i, backward step j -> slot
// L+j, so that at every step the 64 lanes of a wave (consecutive lines, same step) store one contiguous
// 768-byte run.  (Line-major rows cost 2.6x: 12-byte stores to 64 different cache lines per step, partial
// lines evicted to HBM.)  Worst-case rows are affordable with 288 GB: 1 M lines x 284 slots x 12 B = 3.4 GB.
// A scan over the per-line counts gives every kept line its offset; the pack kernel transposes 16-slot x
// 64-line blocks through LDS and emits the reference's point order [fwd_N .. fwd_1, bwd_1 .. bwd_M]
// (prepend!/append!, stream.jl:652) in 48-byte pieces.
//
// Arithmetic.  No a*b+c contraction anywhere in this file: positions are compared after round-to-nearest-
// even (stream.jl:514) so one ulp moves a line to another voxel; with contraction off every operation is
// an IEEE single/double operation in the reference's order and the CPU restatement is matched bit for bit.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "common.h"

#pragma clang fp contract(off)

namespace {

struct FieldArgs {
    const float *ovec[8];
    const float *f[8];
    const float *fa;
    const uint8_t *mask;
    float4 *field;
    uint8_t *mask_out;
    int64_t nvox;
    int nvec;
    int has_f;
    float f_thresh, fa_thresh;
};

__global__ __launch_bounds__(256) void stream_field_kernel(const FieldArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nvox) return;
    float v[8][3];
    bool any = false;
    for (int k = 0; k < a.nvec; k++)
        for (int c = 0; c < 3; c++) {
            v[k][c] = a.ovec[k][(int64_t)c * a.nvox + i];
            any |= v[k][c] != 0.0f;                               // any(x -> x != 0, ...), stream.jl:99
        }
    bool m = a.mask ? a.mask[i] != 0 : any;                       // stream.jl:95-103 (mask already '> 0'-tested)
    if (a.fa) m = m && (a.fa[i] >= a.fa_thresh);                  // stream.jl:116
    for (int k = 0; k < a.nvec; k++) {
        bool om = m;
        if (a.has_f) om = m && (a.f[k][i] >= a.f_thresh);         // stream.jl:138
        // w = the voxel mask itself (the microscopy regime tests W.mask apart from the vectors, stream.jl:596)
        const float w = m ? 1.0f : 0.0f;
        a.field[i * a.nvec + k] = om ? make_float4(v[k][0], v[k][1], v[k][2], w) : make_float4(0.0f, 0.0f, 0.0f, w);
    }
    if (a.mask_out) a.mask_out[i] = m ? 1 : 0;
}

struct Pair { int64_t pts, lines; };
constexpr int SCR_TILE = 16;         // lines per pack tile: the unit whose output range a pack workgroup assembles in LDS
constexpr int FUSED_TILE = 16;       // .. in the fused trace + pack kernel
constexpr int FUSED_BLOCK = 512;     // its workgroup: 8 waves, so that 4 workgroups of 35 KB of LDS keep the CU's 32 wave slots full
// Lines that share a ROW of the point scratch (a row = one slot of SCR_ROW lines, SCR_ROW x 12 bytes).  [r5] The tracer is bound by these
// stores (tools/stream_bound_probe.py: without them the trace takes 0.43 of 0.50 ms, without the field gathers nothing less), and
// non-temporal stores that end in the middle of a 128-byte line are slow: 192-byte rows (16 lines) reach 4.1 TB/s alone, 384-byte rows
// (32 lines = three whole lines) 5.1 (tools/probes/write_probe.hip), and the trace kernel went 0.54 -> 0.49 ms with them.  But then the
// pack side pays the same back: 16-line tiles that read 192 of every 384 bytes 0.54 -> 0.63 ms, 32-line tiles (55 KB of LDS, two
// workgroups per CU) 0.60 (1 024 threads) / 0.64 (512), the fused kernel 10.6 -> 11.7 ms (half rows) / 13.3 (32-line tiles, 1 024
// threads): profiles/r05/negative_results.txt.  So the row stays the pack tile, and the trace kernel gets its whole lines another way:
// it parks four trips' points in LDS and stores the four rows together (see there).  (8-line rows, 96 B: the trace 3 x slower.)
constexpr int SCR_ROW = 16;
static_assert(SCR_ROW % SCR_TILE == 0 && SCR_ROW % FUSED_TILE == 0, "pack tiles are whole fractions of a scratch row");
constexpr int SCR_SLOT_FLOATS = SCR_ROW * 3;                       // from one slot of a line to its next
// first float of line li's slot 0:  [row = li / SCR_ROW][slot][li % SCR_ROW][3]
__host__ __device__ __forceinline__ int64_t scratch_line_base(int64_t li, int nslots) {
    return (li / SCR_ROW) * ((int64_t)nslots * SCR_SLOT_FLOATS) + (li % SCR_ROW) * 3;
}
constexpr int TRACE_SCAN_B = 2048;   // = SCAN_B (lines per block of the scan)
struct TraceArgs {
    const float4 *field;        // [nvox][nvec]
    const int64_t *seeds;       // [nseed] 0-based linear voxel index
    const float *sublist;       // [nsub][3]
    float *scratch;             // [nlines/16 rows][nslots][16 lines][3]: the point a line emits at loop trip t -> slot t: forward point i in slot i,
                                // backward point j in slot nf + gap + j (a row's slots are one contiguous run that the pack kernel streams; the
                                // trace kernel stores four trips' rows of a tile together, as whole 128-byte lines)
    int32_t *npts, *nfwd;       // [nlines]; nfwd = forward points | gap << 30 (gap = 1: the forward pass ended on a trip that emitted nothing)
    int64_t line0, nlines;      // this batch covers global lines [line0, line0+nlines)
    int nx, ny, nz, nvec, nsub, len_max, stride, nslots;
    int scratch_plain;          // diagnostic build only (FIBERS_STREAM_SCRATCH_PLAIN): the point scratch with the default cache policy instead of non-temporal
    int norm_generic;           // diagnostic build only (FIBERS_STREAM_NORM_GENERIC): normalise3's generic expansion for every vector
    int lcm_plain;              // LCM runs: lcm_thresh >= 2^-40 (every entry of the thresholded matrices is 0 or at least that)
    int dbg;                    // diagnostic build only (FIBERS_STREAM_DBG bit mask; WRONG RESULTS, timing experiments): 1 = no gather after the seed's, 2 = no point stores
    // FUSED (fibd_stream_run): the block that traced 256 lines also packs them -- a decoupled look-back over the blocks' kept-line / point
    // totals gives it its place in the output
    unsigned long long *fstate; // [blocks] granules: status << 62 | kept lines << 36 | points (zeroed before the launch)
    int32_t *out_npts;
    int64_t *out_seed;
    float *out_xyz;
    Pair *ftotal;               // the last block's inclusive prefix = the call's totals
    int64_t *fcounts;           // .. and, fibd_stream_run_enqueue, {lines, points} for the caller
    int64_t lines_cap, points_cap;
    int len_min;
    float cosang, step, smooth;
    // microscopy regime (stream.jl:252-287, 547-619)
    // LCM-guided tracking (stream.jl:200-236, 380-495)
    const float *lcm;           // [nvox][10] thresholded local connection matrices (lcm_prepare_kernel)
    int sd0, sd1;               // in-plane dimensions (0-based)
    unsigned long long rng_seed;
    const float4 *search;       // half of the search cube's cells with rho < 1, column-major order: {unit vector, bits of (kx | ky<<8 | kz<<16)} (0-based cell)
    int nsearch, search_dist;
    int sdx, sdy, sdz;          // .. per axis: search_dist, or 0 along the through-plane axis of 2-D angle inputs (stream.jl:153-155)
    float search_cosang;
    const int32_t *cell_start;  // [G^3 + 1]: the table is sorted by direction cell (x fastest); entries of cell i = [cell_start[i], cell_start[i+1])
    int G;                      // direction grid: cell (floor((v + 1) / h)) per axis, h = 2 / G
};

// iszero(v) on a 3-view (stream.jl:353): all three components == 0, -0.0 included.  One OR of the three bit patterns and one class
// test (a pattern with no magnitude bit set is +-0) instead of three compares.
__device__ __forceinline__ bool is_zero3(float x, float y, float z) {
    const unsigned u = __float_as_uint(x) | __float_as_uint(y) | __float_as_uint(z);
    return __builtin_amdgcn_classf(__uint_as_float(u), 0x60);     // (v_cmp_class: -0 | +0)
}
// v_cvt_i32_f32 itself (NaN -> 0, out of range -> INT_MIN / INT_MAX): what the bounds test below wants, and not what C++ promises
__device__ __forceinline__ int cvt_i32_sat(float x) {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return ax * bx + ay * by + az * bz;                           // (x+y)+z, no fma
}
// A point on its way out: written once, read back by the pack kernel (or by nobody on the device).  NON-TEMPORAL -- [r4] with the
// default policy the 1.6 GB of scratch points of a million lines go through the XCDs' L2 and push the orientation field out of it:
// the trace kernel took 0.69 ms, 0.55 with this store (and 0.50 with no store at all); the pack kernel gains 4 % too.
typedef float f32x3_t __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void store_point(float *d, float x, float y, float z) {   // (the microscopy tracer: one lane per line stores)
    const f32x3_t v = {x, y, z};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x3_t *>(d));
}
#ifdef FIB_AB_VARIANTS
typedef float fib_f4_t __attribute__((ext_vector_type(4)));
// (diagnostic build, FIBERS_STREAM_SCRATCH_PLAIN = 1..6: the scratch stores with another cache policy, as inline assembly -- written as a second
// C++ store hipcc once merged the two branches into ONE store without the non-temporal hint; tests/test_kernel_schedule.py asserts the hint)
__device__ __forceinline__ void store_x4_variant(fib_f4_t *d, fib_f4_t v, int flavour) {
    if (flavour == 1) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(d), "v"(v) : "memory");
    else if (flavour == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(d), "v"(v) : "memory");
    else if (flavour == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(d), "v"(v) : "memory");
    else if (flavour == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(d), "v"(v) : "memory");
    else if (flavour == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(d), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(d), "v"(v) : "memory");
}
#endif

// w ./= norm(w)  (stream.jl:680).  LinearAlgebra.norm on a 3-vector is generic_norm2: the largest magnitude first (zero / Inf / NaN return
// it), then the squares in Float32, their sum and the square root in Float64, the result converted to Float32; then three IEEE divisions.
// [r5] Written out as `(float)sqrt(acc)` and `w / n` this is ~60 of the tracer's ~125 vector instructions per step (which bound the
// three-vector tracer and, once its stores leave as whole lines, the one-vector tracer too): hipcc expands the f64 square root with a range-scaling prologue / special-value epilogue and each division with two
// v_div_scale, a reciprocal, its refinement, v_div_fmas and v_div_fixup.  Where every component is 0 or within 2^-40 .. 2^40 -- any
// orientation field -- none of the scaling or fix-up can trigger, and the SAME instruction sequences without them give the same bits:
// the Goldschmidt square root of llvm's f64 lowering, ONE refined reciprocal for the three divisions, two residual corrections per
// quotient.  Anything outside that range takes the generic expansion.  tests/test_gpu_stream.py compares the two over random and edge inputs.
__device__ __forceinline__ void normalise3(float &wx, float &wy, float &wz, bool generic = false) {
    const float m = fmaxf(fabsf(wx), fmaxf(fabsf(wy), fabsf(wz)));
    const float lo = fminf(fabsf(wx), fminf(fabsf(wy), fabsf(wz)));
    bool plain = lo >= 0x1p-40f && m <= 0x1p40f;                   // (NaN fails)
#ifdef FIB_AB_VARIANTS
    if (generic) plain = false;                                    // diagnostic build, FIBERS_STREAM_NORM_GENERIC: hipcc's expansions throughout
    else
#endif
    if (!plain && m <= 0x1p40f && m >= 0x1p-40f)                   // components that are exactly zero (2-D sections) are fine too
        plain = (wx == 0.0f || fabsf(wx) >= 0x1p-40f) && (wy == 0.0f || fabsf(wy) >= 0x1p-40f) && (wz == 0.0f || fabsf(wz) >= 0x1p-40f);
    if (plain) {
        double acc = (double)(wx * wx);
        acc += (double)(wy * wy);
        acc += (double)(wz * wz);
        const double y = __builtin_amdgcn_rsq(acc);
        double g = acc * y, h = y * 0.5;
        const double r = fma(-h, g, 0.5);
        g = fma(g, r, g);
        double d = fma(-g, g, acc);
        h = fma(h, r, h);
        g = fma(d, h, g);
        d = fma(-g, g, acc);
        g = fma(d, h, g);
        const float n = (float)g;
        float rc = __builtin_amdgcn_rcpf(n);
        rc = fmaf(fmaf(-n, rc, 1.0f), rc, rc);
        auto quot = [&](float x) {
            float q = x * rc;
            q = fmaf(fmaf(-n, q, x), rc, q);
            q = fmaf(fmaf(-n, q, x), rc, q);
            return __builtin_copysignf(q, x);                     // (-0 / n = -0; the residual steps give +0)
        };
        wx = quot(wx); wy = quot(wy); wz = quot(wz);
        return;
    }
    float n;
    if (m == 0.0f || !(m < INFINITY)) n = m;
    else {
        double acc = (double)(wx * wx);
        acc += (double)(wy * wy);
        acc += (double)(wz * wz);
        n = (float)sqrt(acc);
    }
    wx = wx / n; wy = wy / n; wz = wz / n;
}

// The random-number contract of LCM-guided tracking (include/fibers_hip.h): the k-th uniform of streamline `line` is
// splitmix64 of (seed, line, k), top 24 bits -> [0,1).  Same function in the oracle (orc_uniform).
__host__ __device__ __forceinline__ unsigned long long fib_splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ float fib_uniform(unsigned long long seed, unsigned long long line, unsigned k) {
    const unsigned long long h = fib_splitmix64(seed ^ fib_splitmix64(line * 0xD1342543DE82EF95ull + (unsigned long long)k));
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}

// lcm_array = permutedims(lcms.vol, (4,1,2,3)) .* (. >= lcm_thresh)  (stream.jl:207,217): [nvox][10] from planar [10][nvox]
__global__ __launch_bounds__(256) void lcm_prepare_kernel(const float *__restrict__ lcms, float thresh, int64_t nvox, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    for (int j = 0; j < 10; j++) {
        const float x = lcms[(int64_t)j * nvox + i];
        out[i * 10 + j] = x >= thresh ? x : 0.0f;                  // `x * false` is a strong zero in Julia (NaN -> 0)
    }
}

// coordinate increments for exiting through edge j = 1..4 (stream.jl:224-226) along dimension c
__device__ __forceinline__ int lcm_dxyz(int j, int c, int sd0, int sd1) {
    const int a4 = (j == 1) ? -1 : (j == 3 ? 1 : 0), b4 = (j == 2) ? -1 : (j == 4 ? 1 : 0);
    return c == sd0 ? a4 : (c == sd1 ? b4 : 0);
}
__device__ __forceinline__ int lcm_match_edge(int dx, int dy, int dz, int sd0, int sd1) {
    for (int j = 1; j <= 4; j++)
        if (lcm_dxyz(j, 0, sd0, sd1) == dx && lcm_dxyz(j, 1, sd0, sd1) == dy && lcm_dxyz(j, 2, sd0, sd1) == dz) return j;
    return 0;
}

// The points go into the slot-major scratch and the per-line counts beside them; the scan + pack kernels below give every kept
// line its place.  ([r5] the two-pass form -- count, scan, trace again straight into the output -- was as slow as the copy it
// saved (a trace without stores takes 0.50 of the 0.53 ms) and has been removed: profiles/r04/negative_results.txt.)
// TRI (params.interp = 1, not in the reference): the direction followed is the trilinear blend of the eight voxels around the
// tentative position instead of the nearest voxel's vector -- see the TRI block below for the exact definition.
// WIDE: 64-bit voxel indices and gather offsets, for orientation fields of 2^28 vectors (4 GiB) or more -- the microscopy
// regime's whole-slide sections (stream.jl:83,147-172); chosen at launch, bit-identical to the 32-bit form on small fields.
struct FuseLds;   // (defined with the pack kernels below)
template <int NVEC> __device__ void fused_pack_block(const TraceArgs &a, int64_t b, int64_t li, bool live, int npts, int nf, int gap);

template <int NVEC, bool LCM = false, bool TRI = false, bool WIDE = false, bool FUSED = false>   // NVEC > 0: compile-time vector count; 0: runtime
__global__ __launch_bounds__(FUSED ? FUSED_BLOCK : 256) void stream_trace_kernel(const TraceArgs a) {
    // FUSED: the workgroup's place in the line order is a TICKET, not blockIdx: whatever order workgroups are dispatched in, everything a
    // workgroup later waits for in the look-back (tickets below its own) was drawn by a workgroup that is already running
    int64_t wg = blockIdx.x;
    if constexpr (FUSED) {
        __shared__ unsigned s_ticket;
        if (threadIdx.x == 0) s_ticket = atomicAdd(reinterpret_cast<unsigned *>(a.fstate + gridDim.x), 1u);
        __syncthreads();
        wg = s_ticket;
    }
    const int64_t li = wg * blockDim.x + threadIdx.x;
    const bool live_line = li < a.nlines;              // (every thread takes part in the scratch stores and, FUSED, in the pack)
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type vox_t;
    const int nvec = NVEC > 0 ? NVEC : a.nvec;
    const int64_t line = a.line0 + (!live_line ? a.nlines - 1 : li);   // (a thread past the end borrows the last line's seed and traces nothing)
    const int64_t iseed = line / a.nsub;
    const int isub = (int)(line - iseed * a.nsub);
    const int64_t lin = a.seeds[iseed];
    const int sx = (int)(lin % a.nx), sy = (int)((lin / a.nx) % a.ny), sz = (int)(lin / ((int64_t)a.nx * a.ny));
    const float p0x = (float)(sx + 1) + a.sublist[3 * isub];      // pos_now .= seed_vox .+ sub_vox, stream.jl:649
    const float p0y = (float)(sy + 1) + a.sublist[3 * isub + 1];
    const float p0z = (float)(sz + 1) + a.sublist[3 * isub + 2];
    // [r5] The points leave in GROUPS OF FOUR TRIPS as whole 128-byte lines.  A trip's points of a tile are a 192-byte row, and a
    // non-temporal store instruction that ends in the middle of a line is slow (such stores alone: 4.1 TB/s; whole lines: 5.1 --
    // tools/probes/write_probe.hip), and these stores bound the kernel (tools/stream_bound_probe.py).  So a lane parks its point in LDS
    // (768 B per tile: 4 slots x 16 lines x 12 B) and after every fourth trip the tile's 16 lanes store the 48 float4 of those four rows --
    // 3 instructions per wave, each 4 runs of 256 contiguous bytes -- instead of 4 instructions of 4 x 192 B.  Same scratch layout, same
    // bytes; lanes whose line has ended stay in the loop (switched off) to do their share of the stores.
    static_assert(SCR_ROW == 16, "the grouped scratch stores map a tile's 16 lanes to its 48 float4 per four slots");
    float *stage;                                                 // this lane's tile: [4 slots][16 lines][3]
    if constexpr (FUSED) {
        extern __shared__ __attribute__((aligned(16))) float f_obuf[];   // (the pack phase's tile buffer: free until the trace loop is over)
        stage = f_obuf + (threadIdx.x >> 4) * (4 * SCR_SLOT_FLOATS);
    } else {
        __shared__ __attribute__((aligned(16))) float s_stage[256 / 16][4 * SCR_SLOT_FLOATS];
        stage = s_stage[threadIdx.x >> 4];
    }
    const int tj = threadIdx.x & 15;                              // this line's place in its tile
    float *const tile0 = a.scratch + scratch_line_base(li - tj, a.nslots);   // the tile's slot 0
    const bool tile_ok = li - tj < a.nlines;                      // (a tile wholly past the end has no scratch)
    int trip = 0;                                                 // wave-uniform: the slot the current trip writes
    const char *fbase = reinterpret_cast<const char *>(a.field);   // wave-uniform base; per-lane offsets are 32-bit
    const float fnx = (float)a.nx, fny = (float)a.ny, fnz = (float)a.nz;
    const float omc = 1.0f - a.smooth;
    int ivec = 0, npts = 0, nf = 0, pass = 0, gap = 0;
    unsigned ndraw = 0;                                           // uniforms consumed by this line (LCM)
    // [r4] The voxel's vectors stay in registers while the line stays in the voxel (a step is half a voxel: 35-50 % of the steps do): a
    // lane that does not need the gather is switched off for it.  Same data, same arithmetic.  Three vectors per voxel (C5): trace
    // 6.4 -> 6.0 ms; one vector (C4): nothing (tools/stream_c5_times.py, stream_kernel_times.py).
    constexpr int NCV = NVEC > 0 ? NVEC : 1;
    constexpr int NTRI = (TRI && NVEC == 1) ? 8 : 1;             // the trilinear option's cell of corner vectors (one vector per voxel)
    float tri_x[NTRI], tri_y[NTRI], tri_z[NTRI];
    float tri_gx = __builtin_nanf(""), tri_gy = 0.0f, tri_gz = 0.0f;   // (NaN: no cell yet)
    float4 cvec[NCV];
    vox_t cvox = ~(vox_t)0;
    float px = p0x, py = p0y, pz = p0z;
    float vx, vy, vz;
    {
        const float4 s = a.field[lin * nvec];                     // view(W.ovecs, :, ivec_next, seed...), stream.jl:650 (ivec_next = 1)
        vx = s.x * 1.0f; vy = s.y * 1.0f; vz = s.z * 1.0f;
    }
    // One trip of the loop = one step of whichever pass the lane is in.  true: the pass ended at this trip (stream.jl:657, :670, :674).
    bool emitted = false;
    auto step = [&]() -> bool {
        emitted = false;
        {
            const float nxp = px + vx * a.step, nyp = py + vy * a.step, nzp = pz + vz * a.step;   // stream.jl:512
            const float rx = rintf(nxp), ry = rintf(nyp), rz = rintf(nzp);                        // stream.jl:514
            // x in axes(mask, 1) ... (:517) on the 0-based integers: one unsigned compare per axis (NaN -> 0 - 1, +-huge -> saturated: outside)
            const unsigned ux = (unsigned)cvt_i32_sat(rx) - 1u, uy = (unsigned)cvt_i32_sat(ry) - 1u, uz = (unsigned)cvt_i32_sat(rz) - 1u;   // (unsigned: no overflow to reason about)
            if (!((ux < (unsigned)a.nx) & (uy < (unsigned)a.ny) & (uz < (unsigned)a.nz))) return true;
            const int ix = (int)ux, iy = (int)uy, iz = (int)uz;
            vox_t vox;
            const float4 *cand;
            if constexpr (WIDE) {
                vox = (uint64_t)((int64_t)ix + (int64_t)a.nx * ((int64_t)iy + (int64_t)a.ny * (int64_t)iz));
                cand = reinterpret_cast<const float4 *>(fbase + vox * (uint64_t)(nvec * 16));
            } else {
                vox = (uint32_t)(ix + a.nx * (iy + a.ny * iz));   // nvox < 2^28 / nvec: the byte offset fits 32 bits
                cand = reinterpret_cast<const float4 *>(fbase + (size_t)(vox * (uint32_t)(nvec * 16)));
            }
            float bx = 0.0f, by = 0.0f, bz = 0.0f, bestc = 0.0f, besta = 0.0f;
            int best = 0;
            if constexpr (NVEC > 0) {
#ifdef FIB_AB_VARIANTS
                if (a.dbg & 1) { if (cvox == ~(vox_t)0) { for (int k = 0; k < NVEC; k++) cvec[k] = cand[k]; } cvox = vox; }
#endif
                if (vox != cvox) {
#pragma unroll
                    for (int k = 0; k < NVEC; k++) cvec[k] = cand[k];
                    cvox = vox;
                }
            }
#pragma unroll
            for (int k = 0; k < nvec; k++) {                      // stream_pick_by_angle!, stream.jl:350-361
                const float4 w = NVEC > 0 ? cvec[k < NCV ? k : 0] : cand[k];
                float c, ca;
                if (is_zero3(w.x, w.y, w.z)) { c = -INFINITY; ca = -INFINITY; }
                else { c = dot3(vx, vy, vz, w.x, w.y, w.z); ca = fabsf(c); }
                // argmax: first maximum, NaN wins over everything -- as ONE signed compare of the bit patterns: ca is -Inf, >= +0 or a NaN
                // with its sign cleared, and as integers -Inf < +0 <= .. <= +Inf < every such NaN.  (A later NaN with a larger payload
                // replaces an earlier one where the reference keeps the first: both end the line, below, before `best` is used.)
                if (k == 0 || __float_as_int(ca) > __float_as_int(besta)) {
                    best = k; besta = ca; bestc = c; bx = w.x; by = w.y; bz = w.z;
                }
            }
            if (!(fabsf(bestc) < INFINITY)) return true;          // !isfinite -> false, stream.jl:363
            float wx, wy, wz;
            if (bestc > 0.0f) { wx = bx; wy = by; wz = bz; } else { wx = -bx; wy = -by; wz = -bz; }   // :365-369
            ivec = best;                                          // stream.jl:371
            if (TRI) {
                // Trilinear option.  Everything above still decides whether the line goes on (bounds, the nearest voxel's pick must
                // exist) and which vector index the backward pass starts from; the direction becomes
                //   w = normalise( sum over the 8 corners c of floor(p') + {0,1}^3 inside the volume of  t_c * s_c * u_c ),
                // t_c = (ax * ay) * az with a = frac or 1 - frac per axis, u_c = the corner's vector picked by the same angle rule
                // against the current direction (corners without a vector contribute nothing), s_c = sign of its cosine; products
                // and sums in Float32 in corner order x fastest, the norm as LinearAlgebra.norm does it (below).  A zero or
                // non-finite blend ends the line.  The 8 x nvec float4 loads of a step hit L2 / the vector cache (the stencil moves
                // by half a voxel per step); an LDS copy per lane (128 B x nvec x 256 lanes) was not worth its LDS traffic.  [r5] With one
                // vector per voxel the cell's eight vectors are kept in REGISTERS between the steps that stay in it (trace 2.39 -> 1.82 ms).
                const float gx0 = floorf(nxp), gy0 = floorf(nyp), gz0 = floorf(nzp);
                const float tx = nxp - gx0, ty = nyp - gy0, tz = nzp - gz0;
                float sx = 0.0f, sy = 0.0f, sz = 0.0f;
                // per-axis validity of the low / high corner and the linear index of the low corner: no per-corner bounds test
                const bool xl = gx0 >= 1.0f && gx0 <= fnx, xh = gx0 + 1.0f >= 1.0f && gx0 + 1.0f <= fnx;
                const bool yl = gy0 >= 1.0f && gy0 <= fny, yh = gy0 + 1.0f >= 1.0f && gy0 + 1.0f <= fny;
                const bool zl = gz0 >= 1.0f && gz0 <= fnz, zh = gz0 + 1.0f >= 1.0f && gz0 + 1.0f <= fnz;
                const int64_t cbase = ((int64_t)gx0 - 1) + (int64_t)a.nx * (((int64_t)gy0 - 1) + (int64_t)a.ny * ((int64_t)gz0 - 1));
                const float ax0 = 1.0f - tx, ay0 = 1.0f - ty, az0 = 1.0f - tz;
                if constexpr (NVEC == 1) {
                    // [r5] one vector per voxel: the eight corner vectors stay in registers while the position stays in the cell (a step is
                    // half a voxel: ~40 % of the steps do) -- the lanes that changed cell reload, corners outside the volume count as
                    // "no vector".  Same data, same arithmetic.
                    if (gx0 != tri_gx || gy0 != tri_gy || gz0 != tri_gz) {
#pragma unroll
                        for (int c = 0; c < 8; c++) {
                            const int cx = c & 1, cy = (c >> 1) & 1, cz = c >> 2;
                            float4 w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            if ((cx ? xh : xl) && (cy ? yh : yl) && (cz ? zh : zl)) {
                                if constexpr (WIDE) w = *reinterpret_cast<const float4 *>(fbase + (uint64_t)(cbase + cx + (int64_t)a.nx * (cy + a.ny * cz)) * (uint64_t)16);
                                else w = *reinterpret_cast<const float4 *>(fbase + (size_t)((uint32_t)((int)cbase + cx + a.nx * (cy + a.ny * cz)) * 16u));
                            }
                            tri_x[c] = w.x; tri_y[c] = w.y; tri_z[c] = w.z;
                        }
                        tri_gx = gx0; tri_gy = gy0; tri_gz = gz0;
                    }
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const int cx = c & 1, cy = (c >> 1) & 1, cz = c >> 2;
                        const float ux = tri_x[c], uy = tri_y[c], uz = tri_z[c];
                        if (is_zero3(ux, uy, uz)) continue;           // (no vector there, or a corner outside the volume)
                        const float tc = ((cx ? tx : ax0) * (cy ? ty : ay0)) * (cz ? tz : az0);
                        const float uc = dot3(vx, vy, vz, ux, uy, uz);
                        if (!(fabsf(uc) < INFINITY)) continue;
                        const float sg = uc > 0.0f ? tc : -tc;
                        sx = sx + sg * ux; sy = sy + sg * uy; sz = sz + sg * uz;
                    }
                } else
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int cx = c & 1, cy = (c >> 1) & 1, cz = c >> 2;
                    if (!((cx ? xh : xl) && (cy ? yh : yl) && (cz ? zh : zl))) continue;
                    const float tc = ((cx ? tx : ax0) * (cy ? ty : ay0)) * (cz ? tz : az0);
                    const float4 *cc;
                    if constexpr (WIDE) cc = reinterpret_cast<const float4 *>(fbase + (uint64_t)(cbase + cx + (int64_t)a.nx * (cy + a.ny * cz)) * (uint64_t)(nvec * 16));
                    else {
                        const uint32_t cv = (uint32_t)((int)cbase + cx + a.nx * (cy + a.ny * cz));
                        cc = reinterpret_cast<const float4 *>(fbase + (size_t)(cv * (uint32_t)(nvec * 16)));
                    }
                    float ux = 0.0f, uy = 0.0f, uz = 0.0f, uc = 0.0f, ua = 0.0f;
#pragma unroll
                    for (int k = 0; k < nvec; k++) {
                        const float4 w = cc[k];
                        float cs, ca;
                        if (is_zero3(w.x, w.y, w.z)) { cs = -INFINITY; ca = -INFINITY; }
                        else { cs = dot3(vx, vy, vz, w.x, w.y, w.z); ca = fabsf(cs); }
                        if (k == 0 || __float_as_int(ca) > __float_as_int(ua)) { ua = ca; uc = cs; ux = w.x; uy = w.y; uz = w.z; }
                    }
                    if (!(fabsf(uc) < INFINITY)) continue;
                    const float sg = uc > 0.0f ? tc : -tc;
                    sx = sx + sg * ux; sy = sy + sg * uy; sz = sz + sg * uz;
                }
                const float m = fmaxf(fabsf(sx), fmaxf(fabsf(sy), fabsf(sz)));
                if (m == 0.0f || !(m < INFINITY)) return true;
                wx = sx; wy = sy; wz = sz;
                normalise3(wx, wy, wz, a.norm_generic != 0);
            }
            bool isdiff = false;
            if (LCM) {
                // stream_pick_by_lcm! (stream.jl:380-495), after the angle pick above set W.ivec_next (stream.jl:530-531)
                const int ivec_ang = ivec;
                int dvx = (int)rintf(px) - (ix + 1), dvy = (int)rintf(py) - (iy + 1), dvz = (int)rintf(pz) - (iz + 1);   // :394-398
                if (dvx == 0 && dvy == 0 && dvz == 0) {           // not entering a new voxel, :400-413
                    float4 w;                                     // (the voxel's vectors are in registers when their count is a template constant)
                    if constexpr (NVEC > 0) { w = cvec[0]; for (int k = 1; k < NVEC; k++) if (ivec == k) w = cvec[k]; }
                    else w = cand[ivec];
                    if (dot3(vx, vy, vz, w.x, w.y, w.z) > 0.0f) { wx = w.x; wy = w.y; wz = w.z; } else { wx = -w.x; wy = -w.y; wz = -w.z; }
                } else {
                    int entry = lcm_match_edge(dvx, dvy, dvz, a.sd0, a.sd1);                         // :416-422
                    if (entry == 0) {                             // diagonal jump: keep the dimension that changes faster, :424-438
                        const float p1 = a.sd0 == 0 ? px : (a.sd0 == 1 ? py : pz), q1 = a.sd0 == 0 ? nxp : (a.sd0 == 1 ? nyp : nzp);
                        const float p2 = a.sd1 == 0 ? px : (a.sd1 == 1 ? py : pz), q2 = a.sd1 == 0 ? nxp : (a.sd1 == 1 ? nyp : nzp);
                        const int zap = fabsf(p1 - q1) < fabsf(p2 - q2) ? a.sd1 : a.sd0;
                        if (zap == 0) dvx = 0; else if (zap == 1) dvy = 0; else dvz = 0;
                        entry = lcm_match_edge(dvx, dvy, dvz, a.sd0, a.sd1);
                    }
                    const float *lp = a.lcm + (size_t)vox * 10;
                    float lcm[10];
                    bool any = false;
#pragma unroll
                    for (int j = 0; j < 10; j++) {                // :441-446
                        const int e0 = j < 4 ? 1 : (j < 7 ? 2 : (j < 9 ? 3 : 4));
                        const int e1 = j < 4 ? j + 1 : (j < 7 ? j - 2 : (j < 9 ? j - 4 : 4));
                        lcm[j] = (e0 == entry || e1 == entry) ? lp[j] : 0.0f;
                        any |= lcm[j] != 0.0f;
                    }
                    if (!any) return true;                        // :448, :492
                    float sum = lcm[0];
#pragma unroll
                    for (int j = 1; j < 10; j++) sum += lcm[j];
                    // lcm ./ sum (:450): ten IEEE divisions by one divisor.  Every entry is 0 or >= lcm_thresh (lcm_prepare_kernel), so with a threshold
                    // >= 2^-40 and a sum <= 2^40 neither operand nor quotient comes near the ends of the exponent range and the
                    // division's expansion without scaling and fix-up gives the same bits (as normalise3): ONE refined reciprocal, two
                    // residual corrections per quotient -- 5 instructions each instead of 11.
                    if (a.lcm_plain && sum <= 0x1p40f) {
                        float rc = __builtin_amdgcn_rcpf(sum);
                        rc = fmaf(fmaf(-sum, rc, 1.0f), rc, rc);
#pragma unroll
                        for (int j = 0; j < 10; j++) {
                            float q = lcm[j] * rc;
                            q = fmaf(fmaf(-sum, q, lcm[j]), rc, q);
                            lcm[j] = fmaf(fmaf(-sum, q, lcm[j]), rc, q);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 10; j++) lcm[j] = lcm[j] / sum;
                    }
                    const float u = fib_uniform(a.rng_seed, (unsigned long long)line, ndraw++);
                    int il = 0;                                   // rand(Categorical(lcm)): first index whose running sum exceeds u
                    float cp = lcm[0];
#pragma unroll
                    for (int j = 1; j < 10; j++) { const bool go = cp <= u && il == j - 1; if (go) { cp += lcm[j]; il = j; } }
                    const int e0 = il < 4 ? 1 : (il < 7 ? 2 : (il < 9 ? 3 : 4));
                    const int e1 = il < 4 ? il + 1 : (il < 7 ? il - 2 : (il < 9 ? il - 4 : 4));
                    const int exitedge = e0 == entry ? e1 : e0;  // :454-456
                    const float ex = (float)lcm_dxyz(exitedge, 0, a.sd0, a.sd1), ey = (float)lcm_dxyz(exitedge, 1, a.sd0, a.sd1),
                                ez = (float)lcm_dxyz(exitedge, 2, a.sd0, a.sd1);
                    float lx = 0.0f, ly = 0.0f, lz = 0.0f, lc = 0.0f, la = 0.0f;
                    int lb = 0;
#pragma unroll
                    for (int k = 0; k < nvec; k++) {              // :462-472
                        const float4 w = NVEC > 0 ? cvec[k < NCV ? k : 0] : cand[k];
                        float c, ca;
                        if (is_zero3(w.x, w.y, w.z)) { c = -INFINITY; ca = -INFINITY; }
                        else { c = dot3(ex, ey, ez, w.x, w.y, w.z); ca = fabsf(c); }
                        if (k == 0 || __float_as_int(ca) > __float_as_int(la)) { lb = k; la = ca; lc = c; lx = w.x; ly = w.y; lz = w.z; }
                    }
                    if (!(fabsf(lc) < INFINITY)) return true;     // :476
                    if (lc > 0.0f) { wx = lx; wy = ly; wz = lz; } else { wx = -lx; wy = -ly; wz = -lz; }   // :480-484
                    ivec = lb;                                    // :486
                }
                isdiff = ivec != ivec_ang;                        // :538
            }
            // push!/prepend! of pos_now (stream.jl:660): the slot of this trip
            // LCM runs: the method-difference flag of the point (stream.jl:666) rides in the sign bit of x (x > 0)
            {
                float *sp = stage + (trip & 3) * SCR_SLOT_FLOATS + tj * 3;
                sp[0] = (LCM && isdiff) ? -px : px; sp[1] = py; sp[2] = pz;
            }
            emitted = true;
            npts++;                                               // (nf = the count when the forward pass ends: below)
            if (!LCM && dot3(vx, vy, vz, wx, wy, wz) < a.cosang) return true;   // stream.jl:670 (not used with LCMs, :668)
            if (npts > a.len_max) return true;                    // stream.jl:674
            if (a.smooth != 0.0f) {                               // stream.jl:677-681
                wx = a.smooth * vx + omc * wx;
                wy = a.smooth * vy + omc * wy;
                wz = a.smooth * vz + omc * wz;
                normalise3(wx, wy, wz, a.norm_generic != 0);
            }
            px = nxp; py = nyp; pz = nzp;                         // stream.jl:684-685
            vx = wx; vy = wy; vz = wz;
        }
        return false;
    };
    // [r4] ONE loop for both passes: a lane whose forward pass ends turns round on the spot (back to the seed, opposite direction,
    // stream.jl:649-650 with dir = -1) while its neighbours go on, so a wave runs max(forward + backward trips) over its lanes --
    // with one loop per pass it ran max(forward) + max(backward): 1.5 x the trips on the benchmark field, whose lines all use up
    // len_max but split it differently between the two directions.  The trip counter is wave-uniform, and so is the scratch slot
    // a trip writes (the 64 lanes' points of a trip are 4 x 192-byte rows): forward point i sits in slot i, backward
    // point j in slot nf + gap + j, gap = 1 if the forward pass ended on a trip that emitted nothing (a failed step).
    auto flush = [&](int t0) {                                   // slots t0 .. t0 + 3 of the wave's four tiles: whole lines
        typedef float nt4_t __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_wave_barrier();                          // (LDS is in order within a wave: the lanes' points are there)
#ifdef FIB_AB_VARIANTS
        if (!(a.dbg & 2))
#endif
        if (tile_ok) {
            nt4_t *g = reinterpret_cast<nt4_t *>(tile0 + (int64_t)t0 * SCR_SLOT_FLOATS);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const nt4_t q = reinterpret_cast<const nt4_t *>(stage)[tj + 16 * k];
#ifdef FIB_AB_VARIANTS
                if (a.scratch_plain) { store_x4_variant(g + tj + 16 * k, q, a.scratch_plain); continue; }
#endif
                __builtin_nontemporal_store(q, g + tj + 16 * k);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    bool dead = !live_line;
    for (;;) {
        if (!dead) {
            const bool ended = step();
            if (ended) {
                if (pass == 1) dead = true;
                else {
                    pass = 1;
                    nf = npts;                                    // the forward pass's points
                    gap = emitted ? 0 : 1;
                    px = p0x; py = p0y; pz = p0z;
                    const float4 s = a.field[lin * nvec + ivec];  // view(W.ovecs, :, ivec_next, seed...), stream.jl:650
                    vx = s.x * -1.0f; vy = s.y * -1.0f; vz = s.z * -1.0f;
                }
            }
        }
        if ((trip & 3) == 3) flush(trip - 3);
        trip++;
        if (__ballot(!dead) == 0ull) break;                       // (wave-uniform)
    }
    if (trip & 3) flush(trip & ~3);                               // (the last, partial group: its unused slots carry stale points nobody reads)
    if (live_line) { a.npts[li] = npts; a.nfwd[li] = nf | (gap << 30); }
    if constexpr (FUSED) fused_pack_block<NVEC>(a, wg, li, live_line, npts, nf, gap);
}

// Divergent termination (lines of a wave end at different steps: 30-53 % of the lane-steps idle on a phantom with a broad length
// distribution, 8 % on the benchmark field).  Four remedies were built, each bit-identical to this kernel, and each measured slower
// on every workload (tools/trace_divergence.py; profiles/r03/trace_compaction.log): a per-lane state machine over one
// flat step loop that refills finished lanes from a queue (round 2, two forms); rounds of launches with the surviving lines
// compacted into a dense list in between (one returning atomic per hand-over on one word: ~88 / us); persistent waves that stash
// their surviving lines in LDS and refill from their own share of the lines, with the two passes as two launches of a flat loop
// (static shares: tail imbalance; a chunked global queue: the atomic rate again).  One lane per line is final; the source of the
// others is in the repository's history.
// ---- microscopy regime: stream_micro_new_point! (stream.jl:547-619) -------------------------------------------
// The next point is the voxel, among those within search_dist voxels of the tentative position and inside a cone of
// search_ang around the current direction, whose first orientation vector is best aligned with the current direction
// (first maximum of |cos| in the column-major order of the (2d+1)^3 search cube; the position snaps to that voxel).
// One wave per line: the 64 lanes share out the cells of the search ball each step.  The unit vectors of the cells
// (search_area, stream.jl:255-277) sit in LDS, one entry per antipodal pair (v(-cell) = -v(cell) exactly, and so is
// its dot product).  The centre cell's vector is 0/0 = NaN, which passes `iszero` and the `<=` cone test in the
// reference: the tentative voxel itself is always a candidate (lane 0 adds it).
__device__ __forceinline__ unsigned long long micro_key(float c, unsigned lin) {
    const float ca = fabsf(c);
    unsigned hi = __float_as_uint(ca);                            // |c| >= 0: the bit pattern is order preserving
    if (ca != ca) hi = 0xffffffffu;                               // NaN wins (Base.argmax)
    return ((unsigned long long)hi << 32) | (unsigned)(~lin);     // ties: the smaller linear index
}

__global__ __launch_bounds__(1024) void stream_trace_micro_kernel(const TraceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 stab[];  // [nsearch] entries, then [G^3 + 1] cell offsets
    int32_t *cst = reinterpret_cast<int32_t *>(stab + a.nsearch);
    for (int i = threadIdx.x; i < a.nsearch; i += blockDim.x) stab[i] = a.search[i];
    for (int i = threadIdx.x; i <= a.G * a.G * a.G; i += blockDim.x) cst[i] = a.cell_start[i];
    __syncthreads();
    const float gh = 2.0f / (float)a.G;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int dx = a.sdx, dy = a.sdy, dz = a.sdz, Sx = 2 * dx + 1, Sy = 2 * dy + 1, Sz = 2 * dz + 1;
    const unsigned lin_centre = (unsigned)(dx + Sx * (dy + Sy * dz)), lin_last = (unsigned)(Sx * Sy * Sz - 1);
    constexpr int64_t slot_floats = SCR_SLOT_FLOATS;
    const float fnx = (float)a.nx, fny = (float)a.ny, fnz = (float)a.nz;
    const float omc = 1.0f - a.smooth;
    for (int64_t li = wave0; li < a.nlines; li += nwaves) {
        const int64_t line = a.line0 + li;
        const int64_t iseed = line / a.nsub;
        const int isub = (int)(line - iseed * a.nsub);
        const int64_t lin = a.seeds[iseed];
        const int sx = (int)(lin % a.nx), sy = (int)((lin / a.nx) % a.ny), sz = (int)(lin / ((int64_t)a.nx * a.ny));
        const float p0x = (float)(sx + 1) + a.sublist[3 * isub];
        const float p0y = (float)(sy + 1) + a.sublist[3 * isub + 1];
        const float p0z = (float)(sz + 1) + a.sublist[3 * isub + 2];
        float *dcur = a.scratch + scratch_line_base(li, a.nslots);   // next slot: the line's points one after the other (gap 0)
        int npts = 0, nf = 0;
        for (int pass = 0; pass < 2; pass++) {
            const float fwd = pass == 0 ? 1.0f : -1.0f;
            float px = p0x, py = p0y, pz = p0z;
            const float4 s = a.field[lin * a.nvec];               // the first vector of the seed voxel (W.ivec_next stays 1)
            float vx = s.x * fwd, vy = s.y * fwd, vz = s.z * fwd;
            for (;;) {
                const float nxp = px + vx * a.step, nyp = py + vy * a.step, nzp = pz + vz * a.step;   // stream.jl:561
                const float rx = rintf(nxp), ry = rintf(nyp), rz = rintf(nzp);                        // :563
                if (!(rx >= 1.0f && rx <= fnx && ry >= 1.0f && ry <= fny && rz >= 1.0f && rz <= fnz)) break;   // :566
                const int cx = (int)rx - 1, cy = (int)ry - 1, cz = (int)rz - 1;                       // 0-based
                const float4 fc = a.field[((int64_t)cx + a.nx * ((int64_t)cy + (int64_t)a.ny * cz)) * a.nvec];
                if (fc.w == 0.0f) break;                                                              // :569
                unsigned long long best = 0ull;                  // 0 = nothing found (every real key has |c| bits or ~lin != 0)
                if (lane == 0) best = micro_key(dot3(vx, vy, vz, fc.x, fc.y, fc.z), lin_centre);     // the centre cell
                // Entries that can pass dot(u, v) > c lie within |v - u/|u|| < sqrt(2 - 2c/|u|) of the (normalised) direction:
                // only the direction cells that box touches are visited (the test itself stays exact); sg = 1 visits the
                // antipodes (dot(u, -v) > c  <=>  v near -u).  A degenerate direction falls back to the whole table.
                const float un = sqrtf(dot3(vx, vy, vz, vx, vy, vz));
                const float r2 = 2.0f - 2.0f * a.search_cosang / un;
                const bool boxed = un > 0.0f && r2 < 1.0f;           // NaN / zero / very short vectors: everything
                const float rr = boxed ? sqrtf(r2 > 0.0f ? r2 : 0.0f) + 2e-3f : 3.0f;
                auto test_entry = [&](int e, int sg) {
                    const float4 t = stab[e];
                    const float sdot = dot3(vx, vy, vz, t.x, t.y, t.z);
                    const float dv = sg ? -sdot : sdot;
                    if (dv <= a.search_cosang) return;                                                // :597-598
                    const unsigned cell = __float_as_uint(t.w);
                    const int kx = (int)(cell & 255u), ky = (int)((cell >> 8) & 255u), kz = (int)(cell >> 16);
                    const int ox = sg ? dx - kx : kx - dx, oy = sg ? dy - ky : ky - dy, oz = sg ? dz - kz : kz - dz;
                    const int ix = cx + ox, iy = cy + oy, iz = cz + oz;
                    if (ix < 0 || ix >= a.nx || iy < 0 || iy >= a.ny || iz < 0 || iz >= a.nz) return;   // :586-588
                    const float4 f = a.field[((int64_t)ix + a.nx * ((int64_t)iy + (int64_t)a.ny * iz)) * a.nvec];
                    if (f.w == 0.0f) return;                                                          // :596
                    const unsigned l0 = (unsigned)(kx + Sx * (ky + Sy * kz));
                    const unsigned long long k = micro_key(dot3(vx, vy, vz, f.x, f.y, f.z), sg ? lin_last - l0 : l0);   // :600-601
                    best = k > best ? k : best;
                };
                if (!boxed) {
                    for (int e = lane; e < a.nsearch; e += 64) { test_entry(e, 0); test_entry(e, 1); }
                } else {
#pragma unroll 1
                    for (int sg = 0; sg < 2; sg++) {
                        const float qx = (sg ? -vx : vx) / un, qy = (sg ? -vy : vy) / un, qz = (sg ? -vz : vz) / un;
                        auto cl = [&](float x) { int c = (int)floorf((x + 1.0f) / gh); return c < 0 ? 0 : (c >= a.G ? a.G - 1 : c); };
                        const int x0 = cl(qx - rr), x1 = cl(qx + rr), y0 = cl(qy - rr), y1 = cl(qy + rr), z0 = cl(qz - rr), z1 = cl(qz + rr);
                        for (int gz = z0; gz <= z1; gz++)
                            for (int gy = y0; gy <= y1; gy++) {
                                const int rowc = a.G * (gy + a.G * gz);
                                const int e0 = cst[rowc + x0], e1 = cst[rowc + x1 + 1];   // cells x0..x1 of this row are contiguous
                                for (int e = e0 + lane; e < e1; e += 64) test_entry(e, sg);
                            }
                    }
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const unsigned long long o = ((unsigned long long)(unsigned)__shfl_xor((int)(best >> 32), off) << 32) |
                                                 (unsigned)__shfl_xor((int)(unsigned)best, off);
                    best = o > best ? o : best;
                }
                if (best == 0ull) break;                          // (cannot happen: the centre is always a candidate)
                const unsigned bl = ~(unsigned)best;              // argmax cell, column-major in the cube
                const int bx = cx + (int)(bl % (unsigned)Sx) - dx, by = cy + (int)((bl / (unsigned)Sx) % (unsigned)Sy) - dy,
                          bz = cz + (int)(bl / (unsigned)(Sx * Sy)) - dz;
                const float4 fb = a.field[((int64_t)bx + a.nx * ((int64_t)by + (int64_t)a.ny * bz)) * a.nvec];
                const float bc = dot3(vx, vy, vz, fb.x, fb.y, fb.z);
                if (!(fabsf(bc) < INFINITY)) break;               // !isfinite, :609
                float wx, wy, wz;
                if (bc > 0.0f) { wx = fb.x; wy = fb.y; wz = fb.z; } else { wx = -fb.x; wy = -fb.y; wz = -fb.z; }   // :616-620
                if (lane == 0) store_point(dcur, px, py, pz);                    // addpt!(strline, pos_now), stream.jl:660
                dcur += slot_floats;
                npts++;
                if (pass == 0) nf++;
                if (dot3(vx, vy, vz, wx, wy, wz) < a.cosang) break;   // :670
                if (npts > a.len_max) break;                          // :674
                if (a.smooth != 0.0f) {                               // :677-681
                    wx = a.smooth * vx + omc * wx;
                    wy = a.smooth * vy + omc * wy;
                    wz = a.smooth * vz + omc * wz;
                    normalise3(wx, wy, wz, a.norm_generic != 0);
                }
                px = (float)(bx + 1); py = (float)(by + 1); pz = (float)(bz + 1);   // the position snaps to the voxel found, :612-614
                vx = wx; vy = wy; vz = wz;
            }
        }
        if (lane == 0) { a.npts[li] = npts; a.nfwd[li] = nf; }
    }
}

// ---- exclusive scan of (kept ? npts : 0, kept ? 1 : 0) over the lines, int64 pairs --------------------
constexpr int SCAN_T = 256, SCAN_E = 8, SCAN_B = SCAN_T * SCAN_E;
static_assert(SCAN_B == TRACE_SCAN_B, "scan block size");

__global__ __launch_bounds__(SCAN_T) void scan_block_kernel(const int32_t *npts, int64_t n, int len_min,
                                                            Pair *excl, Pair *block_tot) {
    __shared__ Pair sh[SCAN_T];
    const int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_E;
    Pair loc[SCAN_E], run{0, 0};
#pragma unroll
    for (int e = 0; e < SCAN_E; e++) {
        const int64_t i = base + e;
        const int np = i < n ? npts[i] : 0;
        const bool keep = i < n && np >= len_min;                 // size(strline, 2) < W.len_min && continue, stream.jl:769
        loc[e] = run;
        run.pts += keep ? np : 0;
        run.lines += keep ? 1 : 0;
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (int off = 1; off < SCAN_T; off <<= 1) {
        Pair t{0, 0};
        if ((int)threadIdx.x >= off) t = sh[threadIdx.x - off];
        __syncthreads();
        sh[threadIdx.x].pts += t.pts; sh[threadIdx.x].lines += t.lines;
        __syncthreads();
    }
    const Pair before = threadIdx.x ? sh[threadIdx.x - 1] : Pair{0, 0};
#pragma unroll
    for (int e = 0; e < SCAN_E; e++) {
        const int64_t i = base + e;
        if (i < n) excl[i] = Pair{loc[e].pts + before.pts, loc[e].lines + before.lines};
    }
    if (threadIdx.x == SCAN_T - 1) block_tot[blockIdx.x] = sh[SCAN_T - 1];
}

// exclusive scan of the per-block totals in place (single workgroup, chunks of SCAN_T with a running carry)
// base (optional): the totals of everything before this batch of lines -- the offsets start there and *total continues from it
__global__ __launch_bounds__(SCAN_T) void scan_totals_kernel(Pair *block_tot, int nblocks, Pair *total, const Pair *base = nullptr, int64_t *counts = nullptr) {
    __shared__ Pair sh[SCAN_T];
    Pair carry = base ? *base : Pair{0, 0};
    for (int base = 0; base < nblocks; base += SCAN_T) {
        const int i = base + (int)threadIdx.x;
        const Pair mine = i < nblocks ? block_tot[i] : Pair{0, 0};
        sh[threadIdx.x] = mine;
        __syncthreads();
        for (int off = 1; off < SCAN_T; off <<= 1) {
            Pair t{0, 0};
            if ((int)threadIdx.x >= off) t = sh[threadIdx.x - off];
            __syncthreads();
            sh[threadIdx.x].pts += t.pts; sh[threadIdx.x].lines += t.lines;
            __syncthreads();
        }
        const Pair incl = sh[threadIdx.x], last = sh[SCAN_T - 1];
        if (i < nblocks) block_tot[i] = Pair{carry.pts + incl.pts - mine.pts, carry.lines + incl.lines - mine.lines};
        carry.pts += last.pts; carry.lines += last.lines;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total = carry;
        if (counts) { counts[0] = carry.lines; counts[1] = carry.pts; }   // (fibd_stream_run_enqueue: {lines, points} for the caller)
    }
}

struct PackArgs {
    const float *scratch;
    const int32_t *npts, *nfwd;
    const Pair *excl, *block_off;
    int32_t *out_npts;
    int64_t *out_seed;
    float *out_xyz;
    int64_t nlines, line0, out_line0, out_pt0;
    int stride, nslots, len_min, scratch_plain;
    int64_t lines_cap, points_cap;   // capped: a line whose place lies beyond the caller's buffers is dropped (fibd_stream_run reports the need)
    int capped;
    int trk;                    // 1: out_xyz is a .trk body: [Int32 npts, npts x 3 Float32 ((xyz+.5)*voxel_size)] per line
    float vs[3];
    // LCM runs (the tile kernel; not with trk): the method-difference flag of a point rides in the sign bit of its x (stream.jl:666) -- the pack
    // strips it and, if flags != NULL, writes it out as one byte per point ([r5] a second pass over the packed points did this: 3.0 of
    // the 17 ms of the 2048^2 benchmark section)
    int lcm;
    uint8_t *out_flags;
};

// One wave per scratch tile (16 lines), four independent waves per workgroup: no workgroup barriers.  A chunk = 16 slots of
// the tile = 3 KiB of contiguous scratch: three 16-byte loads per lane, parked in a wave-private LDS tile (rows padded
// to 52 dwords), then written out line by line: 16 lanes per line, one point each -> a wave instruction emits 4 runs
// of 192 contiguous bytes.  The next chunk's loads fly during the write-out.  Forward slots are reversed on the way out,
// backward slots follow (stream.jl:652).
constexpr int PK_LINES = SCR_TILE, PK_SLOTS = 16, PK_ROW = 52, PK_WAVES = 4;
static_assert(PK_LINES == 16, "the pack kernel maps 16 lanes to the 16 lines of a tile");
__global__ __launch_bounds__(PK_WAVES * 64) void stream_pack_kernel(const PackArgs a) {
    __shared__ __attribute__((aligned(16))) float tile[PK_WAVES][2][PK_SLOTS * PK_ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tix = (int64_t)blockIdx.x * PK_WAVES + wave;   // scratch tile of this wave
    const int64_t line0 = tix * PK_LINES;
    if (line0 >= a.nlines) return;
    // lanes 0..15: the tile's lines
    int nf = 0, nb = 0, bs = 0;
    int64_t p0 = 0;
    {
        const int64_t li = line0 + (lane & 15);
        if (lane < PK_LINES && li < a.nlines) {
            const int n = a.npts[li];
            if (n >= a.len_min) {                               // stream.jl:769
                const int nfr = a.nfwd[li];
                nf = nfr & 0x3fffffff; nb = n - nf; bs = nf + (nfr >> 30);
                const Pair e = a.excl[li], bo = a.block_off[li / SCAN_B];
                const int64_t pt = a.out_pt0 + e.pts + bo.pts, l0 = a.out_line0 + e.lines + bo.lines;
                if (a.capped && (l0 >= a.lines_cap || pt + n > a.points_cap)) { nf = 0; nb = 0; bs = 0; }   // no room: dropped
                else if (a.trk) {
                    reinterpret_cast<int32_t *>(a.out_xyz)[l0 + 3 * pt] = n;          // write(io, Int32(npts)), trk.jl:472
                    p0 = pt * 3 + l0 + 1;
                } else {
                    a.out_npts[l0] = n; a.out_seed[l0] = a.line0 + li;
                    p0 = pt * 3;
                }
            }
        }
    }
    int cnt = nb > 0 ? bs + nb : nf;                            // slots of the tile that hold points of kept lines
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) cnt = max(cnt, __shfl_xor(cnt, o));
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    const int nch = (cnt + PK_SLOTS - 1) / PK_SLOTS;
    // write-out mapping: lane = (line group lg, point pt); line tl = lg + 4 j
    const int lg = lane >> 4, pt = lane & 15;
    int wnf[4], wnb[4], wbs[4];
    int64_t wp0[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        wnf[j] = __shfl(nf, lg + 4 * j); wnb[j] = __shfl(nb, lg + 4 * j); wbs[j] = __shfl(bs, lg + 4 * j);
        wp0[j] = ((int64_t)__shfl((int)(p0 >> 32), lg + 4 * j) << 32) | (uint32_t)__shfl((int)(uint32_t)p0, lg + 4 * j);
    }
    const float4 *tbase = reinterpret_cast<const float4 *>(a.scratch + scratch_line_base(line0, a.nslots));
    float4 v[3];
    auto fetch = [&](int c) {                                   // 16 slots of the tile
        const int s0 = c * PK_SLOTS;
        const int live = cnt - s0;                              // live slots of the chunk
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int f = lane + 64 * i;                        // float4 index within the chunk: slot f / 12 (rows of 16 lines: one contiguous 3-KiB run)
            v[i] = (f < live * 12) ? tbase[(int64_t)(s0 + f / 12) * (SCR_SLOT_FLOATS / 4) + f % 12] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (nch > 0) fetch(0);
    for (int c = 0; c < nch; c++) {
        float *T = tile[wave][c & 1];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int f = lane + 64 * i;
            *reinterpret_cast<float4 *>(T + (f / 12) * PK_ROW + (f % 12) * 4) = v[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (c + 1 < nch) fetch(c + 1);                          // next chunk's loads fly during the write-out
        const int sidx = c * PK_SLOTS + pt;                     // the slot = the loop trip that emitted the point
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int tl = lg + 4 * j;
            int64_t p = -1;                                     // forward points reversed, backward points behind them (stream.jl:652)
            if (sidx < wnf[j]) p = (int64_t)(wnf[j] - 1 - sidx);
            else if (sidx >= wbs[j] && sidx - wbs[j] < wnb[j]) p = (int64_t)wnf[j] + (sidx - wbs[j]);
            if (p >= 0) {
                struct P3 { float x, y, z; };
                const float *t = T + pt * PK_ROW + tl * 3;
                float *d = a.out_xyz + wp0[j] + p * 3;
                if (a.trk)                                      // T.((xyz .+ .5) .* voxel_size), Float64 arithmetic (trk.jl:475-476)
                    *reinterpret_cast<P3 *>(d) =
                        P3{(float)(((double)t[0] + 0.5) * (double)a.vs[0]), (float)(((double)t[1] + 0.5) * (double)a.vs[1]),
                           (float)(((double)t[2] + 0.5) * (double)a.vs[2])};
                else
                    *reinterpret_cast<P3 *>(d) = P3{t[0], t[1], t[2]};
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Default pack kernel: one workgroup per scratch tile.  The kept lines of a tile occupy ONE contiguous range of the packed
// output (or of the .trk body), so the workgroup assembles that whole range in LDS -- thread = (slot, line) reads its
// 12-byte point from the contiguous tile (coalesced) and drops it at its final position -- and then streams the range
// out with 16-byte stores aligned to the output address (fill-like: 12-byte stores in 192-byte runs reached 2.7 TB/s,
// aligned 16-byte stores of whole lines reach twice that).  LDS = 16 (len_max + 2) points; longer lines than the LDS
// holds go through stream_pack_kernel above.
template <int TL, int SPC>   // TL lines per tile (a whole fraction of a scratch row); SPC slots per pass of the workgroup's SPC x TL threads
__global__ __launch_bounds__(SPC * TL) void stream_pack_tile_kernel(const PackArgs a) {
    extern __shared__ __attribute__((aligned(16))) float obuf[];
    __shared__ int s_nf[TL], s_nb[TL], s_bs[TL], s_o[TL];   // per line: forward / backward counts (0 if dropped), first backward slot, offset in obuf
    __shared__ int s_cnt[4];                                        // slots in use, -, range length (floats), misalignment
    __shared__ int64_t s_g0;                                        // first float of the range in out_xyz
    const int tid = threadIdx.x;
    const int64_t line0 = (int64_t)blockIdx.x * TL;
    if (tid < 64) {
        const int64_t li = line0 + tid;
        int nf = 0, nb = 0, bs = 0, n = 0;
        int64_t p0 = 0, gs = INT64_MAX, ge = -1;
        if (tid < TL && li < a.nlines) {
            n = a.npts[li];
            if (n >= a.len_min) {                               // stream.jl:769
                const int nfr = a.nfwd[li];
                nf = nfr & 0x3fffffff; nb = n - nf; bs = nf + (nfr >> 30);
                const Pair e = a.excl[li], bo = a.block_off[li / SCAN_B];
                const int64_t pt = a.out_pt0 + e.pts + bo.pts, l0 = a.out_line0 + e.lines + bo.lines;
                if (a.capped && (l0 >= a.lines_cap || pt + n > a.points_cap)) { n = 0; nf = 0; nb = 0; bs = 0; }   // no room: dropped
                else {
                    if (a.trk) { p0 = pt * 3 + l0 + 1; gs = p0 - 1; }   // the Int32 point count precedes the points (trk.jl:472)
                    else { a.out_npts[l0] = n; a.out_seed[l0] = a.line0 + li; p0 = pt * 3; gs = p0; }
                    ge = p0 + (int64_t)n * 3;
                }
            } else n = 0;
        }
        int mf = nb > 0 ? bs + nb : nf;                        // slots of the tile that hold points of kept lines
        int64_t g0 = gs, g1 = ge;
#pragma unroll
        for (int o = 1; o < TL; o <<= 1) {
            mf = max(mf, __shfl_xor(mf, o));
            const int64_t og0 = ((int64_t)__shfl_xor((int)(g0 >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)g0, o);
            const int64_t og1 = ((int64_t)__shfl_xor((int)(g1 >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)g1, o);
            g0 = og0 < g0 ? og0 : g0; g1 = og1 > g1 ? og1 : g1;
        }
        const int mis = g1 >= 0 ? (int)((reinterpret_cast<uintptr_t>(a.out_xyz + g0) >> 2) & 3) : 0;
        if (tid < TL) {
            s_nf[tid] = nf; s_nb[tid] = nb; s_bs[tid] = bs; s_o[tid] = n > 0 ? (int)(p0 - g0) + mis : 0;
            if (a.trk && n > 0) obuf[(int)(p0 - g0) + mis - 1] = __int_as_float(n);
        }
        if (tid == 0) { s_cnt[0] = mf; s_cnt[1] = 0; s_cnt[2] = g1 >= 0 ? (int)(g1 - g0) : 0; s_cnt[3] = mis; s_g0 = g0; }
    }
    __syncthreads();
    const int cnt = s_cnt[0], len = s_cnt[2], mis = s_cnt[3];
    if (len == 0) return;
    uint8_t *fbuf = reinterpret_cast<uint8_t *>(obuf + ((size_t)TL * a.stride * 3 + TL + 8));   // [TL * stride] flags of the range's points (LCM runs)
    const int nch = (cnt + SPC - 1) / SPC;
    const int l = tid % TL, sl = tid / TL;                      // thread = (line, slot within the SPC-slot chunk)
    const int nf = s_nf[l], nb = s_nb[l], bs = s_bs[l], o = s_o[l];
    struct P3 { float x, y, z; };
    const P3 *tbase = reinterpret_cast<const P3 *>(a.scratch + scratch_line_base(line0, a.nslots)) + l + sl * SCR_ROW;
    for (int c0 = 0; c0 < nch; c0 += 4) {
        P3 v[4];
        int pos[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = c0 + i;
            const int t = c * SPC + sl;                         // the slot = the loop trip that emitted the point
            pos[i] = -1;                                        // forward points reversed, backward points behind them (stream.jl:652)
            if (c < nch) {
                if (t < nf) pos[i] = nf - 1 - t;
                else if (t >= bs && t - bs < nb) pos[i] = nf + (t - bs);
            }
            if (pos[i] >= 0) {                                  // read once: non-temporal (the orientation field should keep the Infinity Cache)
                const float *src = reinterpret_cast<const float *>(tbase + (int64_t)c * (SPC * SCR_ROW));
#ifdef FIB_AB_VARIANTS
                if (a.scratch_plain) { f32x3_t q; asm volatile("global_load_dwordx3 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(src) : "memory"); v[i] = P3{q[0], q[1], q[2]}; }
                else
#endif
                v[i] = P3{__builtin_nontemporal_load(src), __builtin_nontemporal_load(src + 1), __builtin_nontemporal_load(src + 2)};
            } else v[i] = P3{0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (pos[i] < 0) continue;
            float *d = obuf + o + pos[i] * 3;
            if (a.trk) {                                        // T.((xyz .+ .5) .* voxel_size), Float64 arithmetic (trk.jl:475-476)
                d[0] = (float)(((double)v[i].x + 0.5) * (double)a.vs[0]); d[1] = (float)(((double)v[i].y + 0.5) * (double)a.vs[1]);
                d[2] = (float)(((double)v[i].z + 0.5) * (double)a.vs[2]);
            } else if (a.lcm) {
                d[0] = fabsf(v[i].x); d[1] = v[i].y; d[2] = v[i].z;
                fbuf[(o - mis) / 3 + pos[i]] = (uint8_t)(__float_as_uint(v[i].x) >> 31);
            } else { d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; }
        }
    }
    __syncthreads();
    if (a.lcm && a.out_flags) {                                 // the range's flags: one byte per point, in the order of the points
        uint8_t *fg = a.out_flags + s_g0 / 3;
        for (int k = tid; k < len / 3; k += SPC * TL) fg[k] = fbuf[k];
    }
    // obuf[mis .. mis + len) -> out_xyz[g0 .. g0 + len): obuf[4k..4k+3] lands on a 16-byte aligned address
    float *gbase = a.out_xyz + s_g0 - mis;
    const int nq = (mis + len + 3) >> 2;
    for (int k = tid; k < nq; k += SPC * TL) {
        const float4 q = reinterpret_cast<const float4 *>(obuf)[k];
        if (4 * k >= mis && 4 * k + 4 <= mis + len) {            // written once, read by nobody on the device: non-temporal
            typedef float nt4_t __attribute__((ext_vector_type(4)));
            const nt4_t qn = {q.x, q.y, q.z, q.w};
            __builtin_nontemporal_store(qn, reinterpret_cast<nt4_t *>(gbase + 4 * k));
        }
        else {
            const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; j++) if (4 * k + j >= mis && 4 * k + j < mis + len) gbase[4 * k + j] = e[j];
        }
    }
}

// ---- [r5] FUSED: the block that traced 256 lines packs them (fibd_stream_run) --------------------------------------
// After its trace loop a block knows its kept lines and points; a decoupled look-back over the blocks' totals (one 64-bit granule per
// block: status | kept lines | points -- the data is the flag, agent-scope atomics, no fence; blocks are dispatched in order, so every
// block a block waits for is running or done) gives it its place in the packed output, and it then packs its own 16 scratch tiles
// exactly as stream_pack_tile_kernel does: the tile's kept lines are ONE contiguous output range, assembled in LDS and streamed out with
// aligned 16-byte non-temporal stores.  No scan kernels, no pack launch; the scratch is still written and read back (256 lines x 144
// slots x 12 B = 442 KB per block: it does not stay on chip), but blocks that pack overlap with blocks that still trace.
constexpr unsigned long long FG_PTS_MASK = (1ull << 36) - 1ull, FG_VAL_MASK = (1ull << 62) - 1ull;
__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int off) {
    return ((unsigned long long)(unsigned)__shfl_up((int)(v >> 32), off) << 32) | (unsigned)__shfl_up((int)(unsigned)v, off);
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int off) {
    return ((unsigned long long)(unsigned)__shfl_xor((int)(v >> 32), off) << 32) | (unsigned)__shfl_xor((int)(unsigned)v, off);
}
typedef __attribute__((address_space(1))) unsigned long long fib_gu64s;
template <int NVEC>
__device__ void fused_pack_block(const TraceArgs &a, int64_t b, int64_t li, bool live, int npts, int nf, int gap) {
    extern __shared__ __attribute__((aligned(16))) float f_obuf[];              // [FT * stride * 3 + slack]: a tile's output range
    __shared__ uint16_t f_nf[FUSED_BLOCK], f_nb[FUSED_BLOCK], f_bs[FUSED_BLOCK];                        // (16 bits: a tile of such lines fits the LDS, so len_max < 2^15)
    __shared__ int64_t f_p0[FUSED_BLOCK];
    constexpr int FB = FUSED_BLOCK, FW = FB / 64;
    __shared__ unsigned long long f_wsum[FW], f_base;
    constexpr int FT = FUSED_TILE, FSL = FB / FT;                               // lines per tile; slots a pass of the 256 threads covers
    __shared__ int t_nf[FT], t_nb[FT], t_bs[FT], t_o[FT], t_cnt[4];
    __shared__ int64_t t_g0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __builtin_amdgcn_s_setprio(3);                                               // the pack's few memory instructions go ahead of the neighbours' trace loops
    const bool keep = live && npts >= a.len_min;                                 // stream.jl:769
    const unsigned long long v = keep ? ((1ull << 36) | (unsigned long long)npts) : 0ull;
    unsigned long long inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned long long o = shfl_up_u64(inc, off); if (lane >= off) inc += o; }
    if (lane == 63) f_wsum[wave] = inc;
    __syncthreads();
    unsigned long long wbase = 0ull;
    for (int w = 0; w < wave; w++) wbase += f_wsum[w];
    unsigned long long agg = 0ull;
    for (int w = 0; w < FW; w++) agg += f_wsum[w];
    if (wave == 0) {
        if (lane == 0)
            __hip_atomic_store((fib_gu64s *)(a.fstate + b), ((b == 0 ? 2ull : 1ull) << 62) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long excl = 0ull;
        int64_t j0 = b - 1;                                                     // lane l looks at block j0 - l
        while (j0 >= 0) {
            const int64_t j = j0 - lane;
            unsigned long long g = 2ull << 62;                                   // (before block 0: an inclusive prefix of nothing)
            if (j >= 0) {
                for (;;) {
                    g = __hip_atomic_load((fib_gu64s *)(a.fstate + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((g >> 62) != 0ull) break;
                    __builtin_amdgcn_s_sleep(4);                                 // (a predecessor still tracing: do not take its issue slots)
                }
            }
            const unsigned long long pre = __ballot((g >> 62) == 2ull);          // nearest predecessor that knows its inclusive prefix
            const int first = pre ? __builtin_ctzll(pre) : 63;
            unsigned long long c = lane <= first ? (g & FG_VAL_MASK) : 0ull;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) c += shfl_xor_u64(c, off);
            excl += c;
            if (pre) break;
            j0 -= 64;
        }
        if (lane == 0) {
            if (b > 0) __hip_atomic_store((fib_gu64s *)(a.fstate + b), (2ull << 62) | (excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f_base = excl;
            if (b == (int64_t)gridDim.x - 1) {
                const unsigned long long tot = excl + agg;
                a.ftotal->pts = (int64_t)(tot & FG_PTS_MASK); a.ftotal->lines = (int64_t)(tot >> 36);
                if (a.fcounts) { a.fcounts[0] = (int64_t)(tot >> 36); a.fcounts[1] = (int64_t)(tot & FG_PTS_MASK); }
            }
        }
    }
    __syncthreads();
    {
        const unsigned long long ex = f_base + wbase + inc - v;                  // kept lines / points before this line
        const int64_t l0 = (int64_t)(ex >> 36), pt = (int64_t)(ex & FG_PTS_MASK);
        int mnf = 0, mnb = 0, mbs = 0;
        int64_t p0 = 0;
        if (keep && !(l0 >= a.lines_cap || pt + npts > a.points_cap)) {          // (no room: dropped; the totals say what was needed)
            a.out_npts[l0] = npts; a.out_seed[l0] = a.line0 + li;
            mnf = nf; mnb = npts - nf; mbs = nf + gap; p0 = pt * 3;
        }
        f_nf[tid] = (uint16_t)mnf; f_nb[tid] = (uint16_t)mnb; f_bs[tid] = (uint16_t)mbs; f_p0[tid] = p0;
    }
    __syncthreads();
    struct P3 { float x, y, z; };
    // per tile: [one wave: the tile's summary] barrier [gather the tile's points into its output range in LDS] barrier [stream the range out]
    // barrier.  (Preparing the next tile's summary beside the stream-out -- two barriers per tile -- measured SLOWER: C4 1.18 against 1.05 ms.)
    for (int t = 0; t < FB / FT; t++) {
        if (tid < 64) {                                                         // the tile's summary (as stream_pack_tile_kernel)
            int tnf = 0, tnb = 0, tbs = 0, n = 0;
            int64_t p0 = 0, gs = INT64_MAX, ge = -1;
            if (tid < FT) {
                tnf = f_nf[t * FT + tid]; tnb = f_nb[t * FT + tid]; tbs = f_bs[t * FT + tid]; n = tnf + tnb; p0 = f_p0[t * FT + tid];
                if (n > 0) { gs = p0; ge = p0 + (int64_t)n * 3; }
            }
            int mf = tnb > 0 ? tbs + tnb : tnf;
            int64_t g0 = gs, g1 = ge;
#pragma unroll
            for (int o = 1; o < FT; o <<= 1) {
                mf = max(mf, __shfl_xor(mf, o));
                const int64_t og0 = (int64_t)shfl_xor_u64((unsigned long long)g0, o), og1 = (int64_t)shfl_xor_u64((unsigned long long)g1, o);
                g0 = og0 < g0 ? og0 : g0; g1 = og1 > g1 ? og1 : g1;
            }
            const int mis = g1 >= 0 ? (int)((reinterpret_cast<uintptr_t>(a.out_xyz + g0) >> 2) & 3) : 0;
            if (tid < FT) { t_nf[tid] = tnf; t_nb[tid] = tnb; t_bs[tid] = tbs; t_o[tid] = n > 0 ? (int)(p0 - g0) + mis : 0; }
            if (tid == 0) { t_cnt[0] = mf; t_cnt[1] = g1 >= 0 ? (int)(g1 - g0) : 0; t_cnt[2] = mis; t_g0 = g0; }
        }
        __syncthreads();
        const int cnt = t_cnt[0], len = t_cnt[1], mis = t_cnt[2];
        if (len > 0) {
            const int nch = (cnt + FSL - 1) / FSL;
            const int l = tid % FT, sl = tid / FT;                              // thread = (line, slot within the pass): a pass reads FB x 12 contiguous bytes
            const int lnf = t_nf[l], lnb = t_nb[l], lbs = t_bs[l], lo = t_o[l];
            const P3 *tbase = reinterpret_cast<const P3 *>(a.scratch + scratch_line_base((b * (FB / FT) + t) * FT, a.nslots)) + l + sl * SCR_ROW;
            for (int c0 = 0; c0 < nch; c0 += 4) {
                P3 q[4];
                int pos[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int c = c0 + i, tt = c * FSL + sl;
                    pos[i] = -1;                                                // forward points reversed, backward points behind them (stream.jl:652)
                    if (c < nch) {
                        if (tt < lnf) pos[i] = lnf - 1 - tt;
                        else if (tt >= lbs && tt - lbs < lnb) pos[i] = lnf + (tt - lbs);
                    }
                    if (pos[i] >= 0) {
                        const float *src = reinterpret_cast<const float *>(tbase + (int64_t)c * (FSL * SCR_ROW));
                        q[i] = P3{__builtin_nontemporal_load(src), __builtin_nontemporal_load(src + 1), __builtin_nontemporal_load(src + 2)};
                    } else q[i] = P3{0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (pos[i] < 0) continue;
                    float *d = f_obuf + lo + pos[i] * 3;
                    d[0] = q[i].x; d[1] = q[i].y; d[2] = q[i].z;
                }
            }
        }
        __syncthreads();
        if (len > 0) {
            float *gbase = a.out_xyz + t_g0 - mis;
            const int nq = (mis + len + 3) >> 2;
            for (int k = tid; k < nq; k += FB) {
                const float4 qq = reinterpret_cast<const float4 *>(f_obuf)[k];
                if (4 * k >= mis && 4 * k + 4 <= mis + len) {
                    typedef float nt4_t __attribute__((ext_vector_type(4)));
                    const nt4_t qn = {qq.x, qq.y, qq.z, qq.w};
                    __builtin_nontemporal_store(qn, reinterpret_cast<nt4_t *>(gbase + 4 * k));
                } else {
                    const float e[4] = {qq.x, qq.y, qq.z, qq.w};
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) if (4 * k + jj >= mis && 4 * k + jj < mis + len) gbase[4 * k + jj] = e[jj];
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void copy_i32_kernel(const int32_t *src, int32_t *dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

}  // namespace

// Caller-owned, grow-only scratch arena of the tracer (fib_stream_params::ws): avoids a multi-GB hipMalloc/hipFree per job.
// One job at a time uses it; the job that releases it records an event on its stream and the next user waits for that
// event on its own stream, so a pack kernel still reading the arena is never overtaken.
struct fib_stream_ws {
    int device = 0;
    std::mutex mu;
    void *p = nullptr;
    size_t bytes = 0;
    bool busy = false, pending = false;
    hipEvent_t done = nullptr;
};

extern "C" int fibd_stream_ws_create(int device, fib_stream_ws **ws) try {
    FIB_CHECK(ws != nullptr, FIB_ERR_INVALID, "NULL argument");
    *ws = nullptr;
    fib::DeviceGuard guard;
    int rc = fib::use_device(device);
    if (rc != FIB_OK) return rc;
    fib_stream_ws *w = new fib_stream_ws();
    w->device = device;
    if (hipEventCreateWithFlags(&w->done, hipEventDisableTiming) != hipSuccess) { delete w; return fib::fail(FIB_ERR_HIP, "hipEventCreate failed"); }
    *ws = w;
    return FIB_OK;
} FIB_API_CATCH

extern "C" void fibd_stream_ws_destroy(fib_stream_ws *ws) try {
    if (!ws) return;
    fib::DeviceGuard guard;
    (void)hipSetDevice(ws->device);
    if (ws->pending) (void)hipEventSynchronize(ws->done);
    if (ws->p) (void)hipFree(ws->p);
    if (ws->done) (void)hipEventDestroy(ws->done);
    delete ws;
} FIB_API_CATCH_VOID

struct fib_stream_job {
    int device = 0;
    fib_stream_ws *ws = nullptr;    // arena the scratch was taken from (NULL: own allocation)
    hipStream_t last_stream = nullptr;   // stream of the job's latest launch (trace or pack)
    fib_stream_params prm{};
    int64_t nseed = 0, nlines = 0;
    int nsub = 1, stride = 0, nslots = 0;
    float *scratch = nullptr;
    // carved out of the same (cached) arena as the scratch: no hipMalloc/hipFree per call
    struct View32 { int32_t *p = nullptr; } npts, nfwd;
    struct ViewP { Pair *p = nullptr; } excl, block_tot, total;
    int64_t kept_lines = 0, kept_pts = 0;
    bool lcm = false;               // LCM run: the method-difference flag rides in the sign bit of x until unpacked
};

extern "C" void fib_stream_job_destroy(fib_stream_job *job) try {
    if (!job) return;
    fib::DeviceGuard guard;
    (void)hipSetDevice(job->device);
    if (job->scratch) {
        if (job->ws) {
            std::lock_guard<std::mutex> lk(job->ws->mu);
            job->ws->pending = hipEventRecord(job->ws->done, job->last_stream) == hipSuccess;
            if (!job->ws->pending) (void)hipStreamSynchronize(job->last_stream);
            job->ws->busy = false;
        } else {
            (void)hipFree(job->scratch);    // (hipFree waits for the device)
        }
    }
    delete job;
} FIB_API_CATCH_VOID

extern "C" int fibd_stream_field(int32_t nvec, int64_t nvox, const float *const *ovec, const float *const *f,
                                 float f_thresh, const float *fa, float fa_thresh, const uint8_t *mask,
                                 float *field4, uint8_t *mask_out, void *stream) try {
    FIB_CHECK(ovec && field4 && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nvec >= 1 && nvec <= 8, FIB_ERR_UNSUPPORTED, "1..8 orientation vectors per voxel are supported (got %d)", nvec);
    FieldArgs a{};
    for (int k = 0; k < nvec; k++) {
        FIB_CHECK(ovec[k] != nullptr, FIB_ERR_INVALID, "NULL orientation volume %d", k);
        a.ovec[k] = ovec[k];
        if (f) { FIB_CHECK(f[k] != nullptr, FIB_ERR_INVALID, "NULL amplitude volume %d", k); a.f[k] = f[k]; }
    }
    a.fa = fa; a.mask = mask; a.field = reinterpret_cast<float4 *>(field4); a.mask_out = mask_out;
    a.nvox = nvox; a.nvec = nvec; a.has_f = f ? 1 : 0; a.f_thresh = f_thresh; a.fa_thresh = fa_thresh;
    hipLaunchKernelGGL(stream_field_kernel, dim3((unsigned)fib::cdiv(nvox, 256)), dim3(256), 0, (hipStream_t)stream, a);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH

namespace {
// 2^28 vectors of 16 bytes = 4 GiB: from there on the gather offsets need 64 bits (stream_trace_kernel<.., WIDE>)
// (diagnostic build: FIBERS_STREAM_WIDE=1 takes the wide form on any field -- tools/stream_wide_check.py compares it with the 32-bit form)
bool field_is_wide(const fib_stream_params *prm) {
    return (int64_t)prm->nx * prm->ny * prm->nz * prm->nvec >= ((int64_t)1 << 28) || fib::ab_env("FIBERS_STREAM_WIDE") != nullptr;
}
// the one-lane-per-line tracer for (LCM, TRI); the vector count is a compile-time constant for 1, 2 and 3 vectors per voxel, wide
// fields take the run-time count (one instantiation per mode)
template <bool LCM, bool TRI>
void launch_trace(const TraceArgs &ta, int nvec, bool wide, unsigned grid, hipStream_t st) {
    if (wide)           hipLaunchKernelGGL((stream_trace_kernel<0, LCM, TRI, true>), dim3(grid), dim3(256), 0, st, ta);
    else if (nvec == 1) hipLaunchKernelGGL((stream_trace_kernel<1, LCM, TRI>), dim3(grid), dim3(256), 0, st, ta);
    else if (nvec == 2) hipLaunchKernelGGL((stream_trace_kernel<2, LCM, TRI>), dim3(grid), dim3(256), 0, st, ta);
    else if (nvec == 3) hipLaunchKernelGGL((stream_trace_kernel<3, LCM, TRI>), dim3(grid), dim3(256), 0, st, ta);
    else                hipLaunchKernelGGL((stream_trace_kernel<0, LCM, TRI>), dim3(grid), dim3(256), 0, st, ta);
}
struct LcmIn { const float *lcms = nullptr; float thresh = 0.0f; int sd0 = 0, sd1 = 1; unsigned long long seed = 0; };
int stream_trace_impl(const fib_stream_params *prm, const float *field4, const LcmIn &lin, const int64_t *seeds, int64_t nseed,
                      const float *sublist, int32_t nsub, void *stream,
                      fib_stream_job **job_out, int64_t *nlines_out, int64_t *npoints_out);
}

extern "C" int fibd_stream_trace(const fib_stream_params *prm, const float *field4, const int64_t *seeds, int64_t nseed,
                                 const float *sublist, int32_t nsub, void *stream,
                                 fib_stream_job **job_out, int64_t *nlines_out, int64_t *npoints_out) try {
    return stream_trace_impl(prm, field4, LcmIn{}, seeds, nseed, sublist, nsub, stream, job_out, nlines_out, npoints_out);
} FIB_API_CATCH

extern "C" int fibd_stream_trace_lcm(const fib_stream_params *prm, const float *field4, const float *lcms, float lcm_thresh,
                                     int32_t strdim0, int32_t strdim1, uint64_t rng_seed,
                                     const int64_t *seeds, int64_t nseed, const float *sublist, int32_t nsub, void *stream,
                                     fib_stream_job **job_out, int64_t *nlines_out, int64_t *npoints_out) try {
    FIB_CHECK(lcms != nullptr, FIB_ERR_INVALID, "NULL lcms volume");
    FIB_CHECK(strdim0 >= 0 && strdim0 < 3 && strdim1 >= 0 && strdim1 < 3 && strdim0 != strdim1, FIB_ERR_INVALID, "invalid in-plane dimensions");
    FIB_CHECK(prm && prm->search_dist == 0, FIB_ERR_UNSUPPORTED, "LCM-guided tracking is a macro-scale mode (search_dist must be 0)");
    LcmIn lin;
    lin.lcms = lcms; lin.thresh = lcm_thresh; lin.sd0 = strdim0; lin.sd1 = strdim1; lin.seed = rng_seed;
    return stream_trace_impl(prm, field4, lin, seeds, nseed, sublist, nsub, stream, job_out, nlines_out, npoints_out);
} FIB_API_CATCH

namespace {
int stream_trace_impl(const fib_stream_params *prm, const float *field4, const LcmIn &lin, const int64_t *seeds, int64_t nseed,
                      const float *sublist, int32_t nsub, void *stream,
                      fib_stream_job **job_out, int64_t *nlines_out, int64_t *npoints_out) {
    FIB_CHECK(prm && field4 && job_out && nlines_out && npoints_out, FIB_ERR_INVALID, "NULL argument");
    *job_out = nullptr;
    FIB_CHECK(nseed >= 0 && (nseed == 0 || seeds), FIB_ERR_INVALID, "invalid seed list");
    FIB_CHECK(nsub >= 1 && sublist, FIB_ERR_INVALID, "sublist must hold at least one offset (use [0,0,0] for nsub=0, stream.jl:180)");
    FIB_CHECK(prm->nx > 0 && prm->ny > 0 && prm->nz > 0 && prm->nvec >= 1 && prm->nvec <= 8, FIB_ERR_INVALID, "invalid volume / nvec");
    FIB_CHECK(prm->len_max >= 0 && prm->len_max < (1 << 24), FIB_ERR_INVALID, "invalid len_max");
    FIB_CHECK(prm->search_dist <= 60, FIB_ERR_UNSUPPORTED, "search_dist up to 60 voxels is supported (got %d)", prm->search_dist);
    FIB_CHECK(prm->interp == 0 || prm->interp == 1, FIB_ERR_INVALID, "interp must be 0 (nearest voxel, the reference) or 1 (trilinear)");
    FIB_CHECK(prm->search_flat_axis >= 0 && prm->search_flat_axis <= 3, FIB_ERR_INVALID, "search_flat_axis must be 0 (none) or 1..3 (x, y, z)");
    FIB_CHECK(prm->interp == 0 || (prm->search_dist == 0 && !lin.lcms), FIB_ERR_UNSUPPORTED,
              "trilinear interpolation is an option of macro-scale angle-picked tracking only");
    FIB_CHECK((int64_t)prm->nx * prm->ny * prm->nz < ((int64_t)1 << 40), FIB_ERR_UNSUPPORTED, "volumes of 2^40 voxels or more are not supported");
    const bool wide = field_is_wide(prm);
    int device = 0;
    FIB_HIP(hipGetDevice(&device));
    hipStream_t st = (hipStream_t)stream;
    fib_stream_job *job = new (std::nothrow) fib_stream_job();
    FIB_CHECK(job != nullptr, FIB_ERR_NOMEM, "out of host memory");
    job->device = device; job->prm = *prm; job->nseed = nseed; job->nsub = nsub;
    job->nlines = nseed * nsub; job->stride = prm->len_max + 2;
    job->nslots = (prm->len_max + 4 + 3) & ~3;          // trips of a line's loop: <= len_max + 1 points + two failed steps; whole groups of four (the tracer stores four slots together)
    const int64_t nl = job->nlines;
    *nlines_out = 0; *npoints_out = 0;
    if (nl == 0) { *job_out = job; return FIB_OK; }
    int rc = FIB_OK;
    auto bail = [&](int code) { fib_stream_job_destroy(job); return code; };
    const int nblocks = (int)fib::cdiv(nl, SCAN_B);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_scratch = up((size_t)fib::cdiv(nl, SCR_ROW) * SCR_ROW * job->nslots * 3 * sizeof(float));
    const size_t b_i32 = up((size_t)nl * sizeof(int32_t)), b_excl = up((size_t)nl * sizeof(Pair));
    const size_t b_btot = up((size_t)nblocks * sizeof(Pair)), b_tot = 256;
    const size_t sbytes = b_scratch + 2 * b_i32 + b_excl + b_btot + b_tot;
    job->last_stream = st;
    if (fib_stream_ws *ws = reinterpret_cast<fib_stream_ws *>(prm->ws)) {   // the caller's arena, when it is free and on this device
        std::lock_guard<std::mutex> lk(ws->mu);
        if (!ws->busy && ws->device == device) {
            if (ws->bytes < sbytes) {
                if (ws->pending) { (void)hipEventSynchronize(ws->done); ws->pending = false; }
                if (ws->p) (void)hipFree(ws->p);
                ws->p = nullptr; ws->bytes = 0;
                if (hipMalloc(&ws->p, sbytes) == hipSuccess) ws->bytes = sbytes;
            }
            if (ws->p) {
                if (ws->pending) { (void)hipStreamWaitEvent(st, ws->done, 0); ws->pending = false; }
                job->scratch = (float *)ws->p; job->ws = ws; ws->busy = true;
            }
        }
    }
    if (!job->scratch) {
        hipError_t e = hipMalloc((void **)&job->scratch, sbytes);
        if (e != hipSuccess) { job->scratch = nullptr; return bail(fib::fail(FIB_ERR_NOMEM, "cannot allocate %zu bytes of streamline scratch: %s", sbytes, hipGetErrorString(e))); }
    }
    {
        char *base = reinterpret_cast<char *>(job->scratch) + b_scratch;
        job->npts.p = reinterpret_cast<int32_t *>(base); base += b_i32;
        job->nfwd.p = reinterpret_cast<int32_t *>(base); base += b_i32;
        job->excl.p = reinterpret_cast<Pair *>(base); base += b_excl;
        job->block_tot.p = reinterpret_cast<Pair *>(base); base += b_btot;
        job->total.p = reinterpret_cast<Pair *>(base);
    }
    (void)rc;

    TraceArgs ta{};
    ta.field = reinterpret_cast<const float4 *>(field4); ta.seeds = seeds; ta.sublist = sublist;
    ta.scratch = job->scratch; ta.npts = job->npts.p; ta.nfwd = job->nfwd.p;
    ta.line0 = 0; ta.nlines = nl;
    ta.nx = prm->nx; ta.ny = prm->ny; ta.nz = prm->nz; ta.nvec = prm->nvec; ta.nsub = nsub;
    ta.len_max = prm->len_max; ta.stride = job->stride; ta.nslots = job->nslots;
    ta.cosang = prm->cosang_thresh; ta.step = prm->step_size; ta.smooth = prm->smooth_coeff;
    { const char *e = fib::ab_env("FIBERS_STREAM_SCRATCH_PLAIN"); ta.scratch_plain = e ? std::atoi(e) : 0; }
    ta.norm_generic = fib::ab_env("FIBERS_STREAM_NORM_GENERIC") != nullptr;
    { const char *e = fib::ab_env("FIBERS_STREAM_DBG"); ta.dbg = e ? std::atoi(e) : 0; }
    const unsigned grid = (unsigned)fib::cdiv(nl, 256);
    fib::DevBuf<float4> d_search;
    fib::DevBuf<int32_t> d_cell;
    fib::DevBuf<float> d_lcm;
    if (prm->search_dist > 0) {
        // search_area (stream.jl:255-277), Float32 arithmetic like the reference's T; one entry per antipodal pair
        const int d = prm->search_dist;
        int d3[3] = {d, d, d};
        if (prm->search_flat_axis >= 1 && prm->search_flat_axis <= 3) d3[prm->search_flat_axis - 1] = 0;   // micro_search_dist[thrudim] = 0, stream.jl:153-155
        const int Sx = 2 * d3[0] + 1, Sy = 2 * d3[1] + 1, Sz = 2 * d3[2] + 1;
        std::vector<float4> tab;
        const int64_t ncell = (int64_t)Sx * Sy * Sz;
        for (int64_t l = 0; l < ncell / 2; l++) {                 // the first half in column-major order; the centre is cell ncell/2
            const int kx = (int)(l % Sx), ky = (int)((l / Sx) % Sy), kz = (int)(l / ((int64_t)Sx * Sy));
            const float rx = (float)(kx - d3[0]) / ((float)d3[0] + 0.5f), ry = (float)(ky - d3[1]) / ((float)d3[1] + 0.5f),
                        rz = (float)(kz - d3[2]) / ((float)d3[2] + 0.5f);
            float q = rx * rx; q = q + ry * ry; q = q + rz * rz;
            const float r = sqrtf(q);
            if (!(r < 1.0f)) continue;
            const uint32_t cell = (uint32_t)kx | ((uint32_t)ky << 8) | ((uint32_t)kz << 16);
            float w; memcpy(&w, &cell, 4);
            tab.push_back(make_float4(rx / r, ry / r, rz / r, w));
        }
        // direction grid: cell size a little above the chord of the search cone, so a query box spans <= 3 cells per axis
        const float chord = std::sqrt(std::max(0.0f, 2.0f - 2.0f * prm->search_cosang));
        int G = (int)std::floor(2.0f / std::max(chord * 1.05f + 4e-3f, 0.1f));
        G = std::max(1, std::min(G, 24));
        const float gh = 2.0f / (float)G;
        auto cell_of = [&](const float4 &t) {
            auto cl = [&](float x) { int c = (int)std::floor((x + 1.0f) / gh); return c < 0 ? 0 : (c >= G ? G - 1 : c); };
            return cl(t.x) + G * (cl(t.y) + G * cl(t.z));
        };
        std::stable_sort(tab.begin(), tab.end(), [&](const float4 &p, const float4 &q) { return cell_of(p) < cell_of(q); });
        std::vector<int32_t> cstart((size_t)G * G * G + 1, 0);
        for (const float4 &t : tab) cstart[(size_t)cell_of(t) + 1]++;
        for (size_t i = 1; i < cstart.size(); i++) cstart[i] += cstart[i - 1];
        const size_t smem = tab.size() * sizeof(float4) + cstart.size() * sizeof(int32_t);
        if (smem > 150 * 1024) return bail(fib::fail(FIB_ERR_UNSUPPORTED, "search_dist %d needs %zu bytes of LDS for the search table (max 150 KiB)", d, smem));
        if ((rc = d_search.alloc(tab.size())) != FIB_OK) return bail(rc);
        if ((rc = d_cell.alloc(cstart.size())) != FIB_OK) return bail(rc);
        hipError_t ec = hipMemcpyAsync(d_search.p, tab.data(), tab.size() * sizeof(float4), hipMemcpyHostToDevice, st);
        if (ec == hipSuccess) ec = hipMemcpyAsync(d_cell.p, cstart.data(), cstart.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
        if (ec == hipSuccess) ec = hipStreamSynchronize(st);     // (the tables are locals: they must outlive the copies)
        if (ec != hipSuccess) return bail(fib::fail(FIB_ERR_HIP, "search table upload failed: %s", hipGetErrorString(ec)));
        ta.search = d_search.p; ta.nsearch = (int)tab.size(); ta.search_dist = d; ta.search_cosang = prm->search_cosang;
        ta.sdx = d3[0]; ta.sdy = d3[1]; ta.sdz = d3[2];
        ta.cell_start = d_cell.p; ta.G = G;
        int ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
        const unsigned mg = (unsigned)std::min<int64_t>(ncu, fib::cdiv(nl, 16));
        ec = hipFuncSetAttribute(reinterpret_cast<const void *>(stream_trace_micro_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (ec != hipSuccess) return bail(fib::fail(FIB_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(ec)));
        fib::ProfScope prof("stream_trace_micro", st);
        hipLaunchKernelGGL(stream_trace_micro_kernel, dim3(mg), dim3(1024), smem, st, ta);
    } else if (lin.lcms) {
        const int64_t nvox = (int64_t)prm->nx * prm->ny * prm->nz;
        if ((rc = d_lcm.alloc((size_t)nvox * 10)) != FIB_OK) return bail(rc);
        hipLaunchKernelGGL(lcm_prepare_kernel, dim3((unsigned)fib::cdiv(nvox, 256)), dim3(256), 0, st, lin.lcms, lin.thresh, nvox, d_lcm.p);
        ta.lcm = d_lcm.p; ta.sd0 = lin.sd0; ta.sd1 = lin.sd1; ta.rng_seed = lin.seed;
        ta.lcm_plain = lin.thresh >= 0x1p-40f && !ta.norm_generic;
        job->lcm = true;
        fib::ProfScope prof("stream_trace_lcm", st);
        launch_trace<true, false>(ta, prm->nvec, wide, grid, st);
    } else {
        fib::ProfScope prof("stream_trace", st);
        if (prm->interp) launch_trace<false, true>(ta, prm->nvec, wide, grid, st);   // trilinear option
        else launch_trace<false, false>(ta, prm->nvec, wide, grid, st);
    }
    { fib::ProfScope prof("stream_scan", st);
    hipLaunchKernelGGL(scan_block_kernel, dim3(nblocks), dim3(SCAN_T), 0, st, job->npts.p, nl, prm->len_min, job->excl.p, job->block_tot.p);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(SCAN_T), 0, st, job->block_tot.p, nblocks, job->total.p);
    }
    hipError_t e = hipGetLastError();
    Pair tot{0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(&tot, job->total.p, sizeof(Pair), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return bail(fib::fail(FIB_ERR_HIP, "streamline trace failed: %s", hipGetErrorString(e)));
    job->kept_lines = tot.lines; job->kept_pts = tot.pts;
    *nlines_out = tot.lines; *npoints_out = tot.pts;
    *job_out = job;
    return FIB_OK;
}
}  // namespace

// flags[i] = sign bit of x of point i, then the sign is cleared (LCM runs; see stream_trace_kernel)
__global__ __launch_bounds__(256) void stream_unpack_flags_kernel(float *xyz, uint8_t *flags, int64_t npts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const float x = xyz[3 * i];
    if (flags) flags[i] = (__float_as_uint(x) >> 31) ? 1 : 0;
    xyz[3 * i] = fabsf(x);
}

static int pack_plain(fib_stream_job *job, int32_t *npts, int64_t *seed_index, float *xyz, uint8_t *flags, bool *flags_done, void *stream);

// whole-tile kernel while 16 (len_max + 2) points (+ the .trk headers and the alignment slack) fit in LDS
static int launch_pack_n(const PackArgs &pa_in, int64_t nlines, int stride, hipStream_t st, bool *flags_done = nullptr) {
    PackArgs pa = pa_in;
    const size_t smem = ((size_t)PK_LINES * stride * 3 + PK_LINES + 8) * sizeof(float) + (pa.lcm ? (((size_t)PK_LINES * stride + 15) & ~(size_t)15) : 0);
    if (flags_done) *flags_done = smem <= 120 * 1024;          // (the wave-per-tile kernel below leaves the flag bit in x: the caller strips it)
    if (smem > 120 * 1024) { pa.lcm = 0; pa.out_flags = nullptr; }
    if (smem <= 120 * 1024) {
        if (smem > 48 * 1024)
            FIB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_pack_tile_kernel<PK_LINES, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((stream_pack_tile_kernel<PK_LINES, 16>), dim3((unsigned)fib::cdiv(nlines, PK_LINES)), dim3(16 * PK_LINES), smem, st, pa);
    } else
        hipLaunchKernelGGL(stream_pack_kernel, dim3((unsigned)fib::cdiv(nlines, PK_LINES * PK_WAVES)), dim3(PK_WAVES * 64), 0, st, pa);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
}
static int launch_pack(fib_stream_job *job, const PackArgs &pa, hipStream_t st, bool *flags_done = nullptr) {
    job->last_stream = st;
    return launch_pack_n(pa, job->nlines, job->stride, st, flags_done);
}

extern "C" int fibd_stream_pack_flags(fib_stream_job *job, int32_t *npts, int64_t *seed_index, float *xyz, uint8_t *flags, void *stream) try {
    bool done = false;
    int rc = pack_plain(job, npts, seed_index, xyz, flags, &done, stream);
    if (rc != FIB_OK || job->kept_pts == 0) return rc;
    if (job->lcm && done) return FIB_OK;                        // (the pack kernel stripped the flag bits and wrote the flags)
    if (job->lcm)
        hipLaunchKernelGGL(stream_unpack_flags_kernel, dim3((unsigned)fib::cdiv(job->kept_pts, 256)), dim3(256), 0, (hipStream_t)stream, xyz, flags, job->kept_pts);
    else if (flags) FIB_HIP(hipMemsetAsync(flags, 0, (size_t)job->kept_pts, (hipStream_t)stream));
    else return FIB_OK;
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fibd_stream_pack(fib_stream_job *job, int32_t *npts, int64_t *seed_index, float *xyz, void *stream) try {
    return fibd_stream_pack_flags(job, npts, seed_index, xyz, nullptr, stream);   // (strips the flag bit of LCM runs)
} FIB_API_CATCH

static int pack_plain(fib_stream_job *job, int32_t *npts, int64_t *seed_index, float *xyz, uint8_t *flags, bool *flags_done, void *stream) {
    FIB_CHECK(job != nullptr, FIB_ERR_INVALID, "job is NULL");
    if (job->kept_lines == 0) return FIB_OK;
    FIB_CHECK(npts && seed_index && xyz, FIB_ERR_INVALID, "NULL output buffer");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(job->device));
    PackArgs pa{};
    pa.scratch = job->scratch; pa.npts = job->npts.p; pa.nfwd = job->nfwd.p;
    pa.excl = job->excl.p; pa.block_off = job->block_tot.p;
    pa.out_npts = npts; pa.out_seed = seed_index; pa.out_xyz = xyz;
    pa.nlines = job->nlines; pa.line0 = 0; pa.out_line0 = 0; pa.out_pt0 = 0;
    pa.stride = job->stride; pa.nslots = job->nslots; pa.len_min = job->prm.len_min;
    pa.scratch_plain = fib::ab_env("FIBERS_STREAM_PACK_PLAIN") != nullptr;
    pa.lcm = job->lcm ? 1 : 0; pa.out_flags = job->lcm ? flags : nullptr;
    fib::ProfScope prof("stream_pack", (hipStream_t)stream);
    { const int rcl = launch_pack(job, pa, (hipStream_t)stream, flags_done); if (rcl != FIB_OK) return rcl; }
    FIB_HIP(hipGetLastError());
    return FIB_OK;                                      // (LCM jobs whose lines do not fit the tile kernel: x still carries the flag bit; the caller strips it)
}

extern "C" int fibd_stream_pack_trk(fib_stream_job *job, const float voxel_size[3], void *body, void *stream) try {
    FIB_CHECK(job != nullptr && voxel_size != nullptr, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(!job->lcm, FIB_ERR_UNSUPPORTED, "the .trk body serialiser does not carry the per-point scalars of an LCM run");
    if (job->kept_lines == 0) return FIB_OK;
    FIB_CHECK(body != nullptr, FIB_ERR_INVALID, "NULL output buffer");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(job->device));
    PackArgs pa{};
    pa.scratch = job->scratch; pa.npts = job->npts.p; pa.nfwd = job->nfwd.p;
    pa.excl = job->excl.p; pa.block_off = job->block_tot.p;
    pa.out_npts = nullptr; pa.out_seed = nullptr; pa.out_xyz = reinterpret_cast<float *>(body);
    pa.nlines = job->nlines; pa.line0 = 0; pa.out_line0 = 0; pa.out_pt0 = 0;
    pa.stride = job->stride; pa.nslots = job->nslots; pa.len_min = job->prm.len_min;
    pa.scratch_plain = fib::ab_env("FIBERS_STREAM_PACK_PLAIN") != nullptr;
    pa.trk = 1; pa.vs[0] = voxel_size[0]; pa.vs[1] = voxel_size[1]; pa.vs[2] = voxel_size[2];
    fib::ProfScope prof("stream_pack_trk", (hipStream_t)stream);
    { const int rcl = launch_pack(job, pa, (hipStream_t)stream); if (rcl != FIB_OK) return rcl; }
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH

// stream (stream.jl:730-790) in ONE call, straight into the caller's buffers (no second call, no allocation by the library):
// trace, scan and pack one after the other on the caller's stream.  ([r5] the batched form -- a batch packed on a second stream
// while the next one is traced -- was measured slower for every batch count, 1.33 ... 1.55 ms against 1.30: both kernels are bound
// by the same HBM traffic, profiles/r04/negative_results.txt; it and its side stream / events have been removed.)
// Macro-scale angle picking only (nearest voxel or trilinear); the microscopy regime and LCM runs use fibd_stream_trace / _pack.
// counts_dev != NULL (fibd_stream_run_enqueue): the call returns once everything is enqueued; the stream writes {lines, points} there.
static int stream_run_impl(const fib_stream_params *prm, const float *field4, const int64_t *seeds, int64_t nseed,
                           const float *sublist, int32_t nsub, int32_t *npts, int64_t *seed_index, int64_t lines_cap,
                           float *xyz, int64_t points_cap, int64_t *nlines_out, int64_t *npoints_out, int64_t *counts_dev, void *stream) {
    FIB_CHECK(prm && field4 && (counts_dev || (nlines_out && npoints_out)), FIB_ERR_INVALID, "NULL argument");
    if (nlines_out) *nlines_out = 0;
    if (npoints_out) *npoints_out = 0;
    FIB_CHECK(nseed >= 0 && (nseed == 0 || seeds), FIB_ERR_INVALID, "invalid seed list");
    FIB_CHECK(nsub >= 1 && sublist, FIB_ERR_INVALID, "sublist must hold at least one offset (use [0,0,0] for nsub=0, stream.jl:180)");
    FIB_CHECK(prm->nx > 0 && prm->ny > 0 && prm->nz > 0 && prm->nvec >= 1 && prm->nvec <= 8, FIB_ERR_INVALID, "invalid volume / nvec");
    FIB_CHECK(prm->len_max >= 0 && prm->len_max < (1 << 24), FIB_ERR_INVALID, "invalid len_max");
    FIB_CHECK(prm->search_dist == 0, FIB_ERR_UNSUPPORTED, "fibd_stream_run covers macro-scale tracking (search_dist 0); use fibd_stream_trace for the microscopy regime");
    FIB_CHECK(prm->interp == 0 || prm->interp == 1, FIB_ERR_INVALID, "interp must be 0 (nearest voxel, the reference) or 1 (trilinear)");
    FIB_CHECK((int64_t)prm->nx * prm->ny * prm->nz < ((int64_t)1 << 40), FIB_ERR_UNSUPPORTED, "volumes of 2^40 voxels or more are not supported");
    FIB_CHECK(lines_cap >= 0 && points_cap >= 0 && (lines_cap == 0 || (npts && seed_index)) && (points_cap == 0 || xyz), FIB_ERR_INVALID, "invalid output buffers");
    const int64_t nl = nseed * nsub;
    hipStream_t st = (hipStream_t)stream;
    if (nl == 0) {
        if (counts_dev) FIB_HIP(hipMemsetAsync(counts_dev, 0, 2 * sizeof(int64_t), st));
        return FIB_OK;
    }
    const bool wide = field_is_wide(prm);
    int device = 0;
    FIB_HIP(hipGetDevice(&device));
    const int stride = prm->len_max + 2, nslots = (prm->len_max + 4 + 3) & ~3;   // (whole groups of four slots: the tracer stores them together)
    const int nblk = (int)fib::cdiv(nl, SCAN_B);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_scr = up((size_t)fib::cdiv(nl, SCR_ROW) * SCR_ROW * nslots * 3 * sizeof(float));
    const size_t b_i32 = up((size_t)nl * sizeof(int32_t)), b_excl = up((size_t)nl * sizeof(Pair)), b_bt = up((size_t)nblk * sizeof(Pair));
    const size_t sbytes = b_scr + 2 * b_i32 + b_excl + b_bt + 256;
    // the caller's workspace when it is free, else one of our own for the call
    fib_stream_ws *ws = reinterpret_cast<fib_stream_ws *>(prm->ws), *own = nullptr;
    bool have = false;
    if (ws) {
        std::lock_guard<std::mutex> lk(ws->mu);
        if (!ws->busy && ws->device == device) { ws->busy = true; have = true; }
    }
    if (!have) {
        int rcw = fibd_stream_ws_create(device, &own);
        if (rcw != FIB_OK) return rcw;
        ws = own; ws->busy = true;
    }
    auto release = [&](int code) {
        if (own) { (void)hipStreamSynchronize(st); own->busy = false; fibd_stream_ws_destroy(own); }
        else {
            std::lock_guard<std::mutex> lk(ws->mu);
            ws->pending = hipEventRecord(ws->done, st) == hipSuccess;
            if (!ws->pending) (void)hipStreamSynchronize(st);
            ws->busy = false;
        }
        return code;
    };
    if (ws->bytes < sbytes) {
        if (ws->pending) { (void)hipEventSynchronize(ws->done); ws->pending = false; }
        if (ws->p) (void)hipFree(ws->p);
        ws->p = nullptr; ws->bytes = 0;
        if (hipMalloc(&ws->p, sbytes) != hipSuccess) { ws->p = nullptr; return release(fib::fail(FIB_ERR_NOMEM, "cannot allocate %zu bytes of streamline scratch", sbytes)); }
        ws->bytes = sbytes;
    }
    if (ws->pending) {
        if (hipStreamWaitEvent(st, ws->done, 0) != hipSuccess) (void)hipEventSynchronize(ws->done);
        ws->pending = false;
    }
    char *pb = reinterpret_cast<char *>(ws->p);
    float *scr = reinterpret_cast<float *>(pb);
    int32_t *bn = reinterpret_cast<int32_t *>(pb + b_scr), *bf = reinterpret_cast<int32_t *>(pb + b_scr + b_i32);
    Pair *bex = reinterpret_cast<Pair *>(pb + b_scr + 2 * b_i32), *bbt = reinterpret_cast<Pair *>(pb + b_scr + 2 * b_i32 + b_excl);
    Pair *total = reinterpret_cast<Pair *>(pb + b_scr + 2 * b_i32 + b_excl + b_bt);
    TraceArgs ta{};
    ta.field = reinterpret_cast<const float4 *>(field4); ta.seeds = seeds; ta.sublist = sublist;
    ta.scratch = scr; ta.npts = bn; ta.nfwd = bf; ta.line0 = 0; ta.nlines = nl;
    ta.nx = prm->nx; ta.ny = prm->ny; ta.nz = prm->nz; ta.nvec = prm->nvec; ta.nsub = nsub;
    ta.len_max = prm->len_max; ta.stride = stride; ta.nslots = nslots;
    ta.cosang = prm->cosang_thresh; ta.step = prm->step_size; ta.smooth = prm->smooth_coeff;
    { const char *e = fib::ab_env("FIBERS_STREAM_SCRATCH_PLAIN"); ta.scratch_plain = e ? std::atoi(e) : 0; }
    ta.norm_generic = fib::ab_env("FIBERS_STREAM_NORM_GENERIC") != nullptr;
    { const char *e = fib::ab_env("FIBERS_STREAM_DBG"); ta.dbg = e ? std::atoi(e) : 0; }
    const unsigned grid = (unsigned)fib::cdiv(nl, 256);
    // [r5] FUSED (fused_pack_block): trace + look-back + pack in ONE launch, for nearest-voxel tracking with 1, 2 or 3 vectors per voxel on fields
    // below 2^28 vectors whose lines fit a 16-line LDS tile (len_max <= ~200) -- from 2^21 lines on: the fused kernel wins by overlapping
    // workgroups that pack with workgroups that still trace, which needs several rounds of workgroups per CU (measured, tools/stream_fused_ab.py:
    // 9.96 M lines x 3 vectors 9.7-10.6 ms against 11.5-11.8; 1 M lines 1.05-1.17 ms against 1.09-1.13: two rounds, no steady state).
    // Everything else takes the three launches below.  (Diagnostic build: FIBERS_STREAM_UNFUSED=1 / FIBERS_STREAM_FUSED=1 force either.)
    const size_t fsmem = std::max(((size_t)FUSED_TILE * stride * 3 + FUSED_TILE + 8) * sizeof(float), (size_t)(FUSED_BLOCK / 16) * 4 * SCR_SLOT_FLOATS * sizeof(float));   // the pack's tile buffer; the trace loop parks four trips of points there
    const bool fused_ok = !wide && !prm->interp && prm->nvec >= 1 && prm->nvec <= 3 && fsmem <= 40 * 1024 &&
                          nl < ((int64_t)1 << 26) && nl * (int64_t)(prm->len_max + 2) < ((int64_t)1 << 36) && b_excl >= ((size_t)fib::cdiv(nl, FUSED_BLOCK) + 1) * sizeof(unsigned long long);
    const bool fused = fused_ok && fib::ab_env("FIBERS_STREAM_UNFUSED") == nullptr && (nl >= ((int64_t)1 << 21) || fib::ab_env("FIBERS_STREAM_FUSED") != nullptr);
    if (fused) {
        const unsigned fgrid = (unsigned)fib::cdiv(nl, FUSED_BLOCK);
        ta.fstate = reinterpret_cast<unsigned long long *>(bex);          // (the scan's array is free: no scan)
        ta.out_npts = npts; ta.out_seed = seed_index; ta.out_xyz = xyz; ta.ftotal = total; ta.fcounts = counts_dev;
        ta.lines_cap = lines_cap; ta.points_cap = points_cap; ta.len_min = prm->len_min;
        if (hipMemsetAsync(ta.fstate, 0, ((size_t)fgrid + 1) * sizeof(unsigned long long), st) != hipSuccess)      // (+ the ticket counter behind the granules)
            return release(fib::fail(FIB_ERR_HIP, "hipMemsetAsync failed"));
        // [r6] one code path per vector count 1..3 (DTI e1 | e1 + e2 or two peaks | three peaks)
        const void *fk = prm->nvec == 1 ? reinterpret_cast<const void *>(stream_trace_kernel<1, false, false, false, true>)
                       : prm->nvec == 2 ? reinterpret_cast<const void *>(stream_trace_kernel<2, false, false, false, true>)
                                        : reinterpret_cast<const void *>(stream_trace_kernel<3, false, false, false, true>);
        if (fsmem > 48 * 1024 && hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fsmem) != hipSuccess)
            return release(fib::fail(FIB_ERR_HIP, "hipFuncSetAttribute failed"));
        fib::ProfScope prof("stream_trace", st);
        if (prm->nvec == 1)      hipLaunchKernelGGL((stream_trace_kernel<1, false, false, false, true>), dim3(fgrid), dim3(FUSED_BLOCK), fsmem, st, ta);
        else if (prm->nvec == 2) hipLaunchKernelGGL((stream_trace_kernel<2, false, false, false, true>), dim3(fgrid), dim3(FUSED_BLOCK), fsmem, st, ta);
        else                     hipLaunchKernelGGL((stream_trace_kernel<3, false, false, false, true>), dim3(fgrid), dim3(FUSED_BLOCK), fsmem, st, ta);
    } else {
    {
        fib::ProfScope prof("stream_trace", st);
        if (prm->interp) launch_trace<false, true>(ta, prm->nvec, wide, grid, st);
        else launch_trace<false, false>(ta, prm->nvec, wide, grid, st);
    }
    {
        fib::ProfScope prof("stream_scan", st);
        hipLaunchKernelGGL(scan_block_kernel, dim3(nblk), dim3(SCAN_T), 0, st, bn, nl, prm->len_min, bex, bbt);
        hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(SCAN_T), 0, st, bbt, nblk, total, (const Pair *)nullptr, counts_dev);
    }
    if (hipGetLastError() != hipSuccess) return release(fib::fail(FIB_ERR_HIP, "streamline trace launch failed"));
    PackArgs pa{};
    pa.scratch = scr; pa.npts = bn; pa.nfwd = bf; pa.excl = bex; pa.block_off = bbt;
    pa.out_npts = npts; pa.out_seed = seed_index; pa.out_xyz = xyz;
    pa.nlines = nl; pa.line0 = 0; pa.out_line0 = 0; pa.out_pt0 = 0;
    pa.stride = stride; pa.nslots = nslots; pa.len_min = prm->len_min;
    pa.scratch_plain = fib::ab_env("FIBERS_STREAM_PACK_PLAIN") != nullptr;
    pa.lines_cap = lines_cap; pa.points_cap = points_cap; pa.capped = 1;   // a line whose place lies beyond the buffers is dropped, the totals say what was needed
    {
        fib::ProfScope prof("stream_pack", st);
        const int rcl = launch_pack_n(pa, nl, stride, st);
        if (rcl != FIB_OK) return release(rcl);
    }
    }   // (!fused)
    if (hipGetLastError() != hipSuccess) return release(fib::fail(FIB_ERR_HIP, "streamline launch failed"));
    if (counts_dev) return release(FIB_OK);                       // ({lines, points} were written by the scan / the last workgroup: no host round trip)
    Pair tot{0, 0};
    hipError_t e = hipMemcpyAsync(&tot, total, sizeof(Pair), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return release(fib::fail(FIB_ERR_HIP, "streamline run failed: %s", hipGetErrorString(e)));
    *nlines_out = tot.lines; *npoints_out = tot.pts;
    if (tot.lines > lines_cap || tot.pts > points_cap)
        return release(fib::fail(FIB_ERR_CAPACITY, "output buffers too small: %lld lines / %lld points needed, %lld / %lld given",
                                 (long long)tot.lines, (long long)tot.pts, (long long)lines_cap, (long long)points_cap));
    return release(FIB_OK);
}

extern "C" int fibd_stream_run(const fib_stream_params *prm, const float *field4, const int64_t *seeds, int64_t nseed,
                               const float *sublist, int32_t nsub, int32_t *npts, int64_t *seed_index, int64_t lines_cap,
                               float *xyz, int64_t points_cap, int64_t *nlines_out, int64_t *npoints_out, void *stream) try {
    FIB_CHECK(nlines_out && npoints_out, FIB_ERR_INVALID, "NULL argument");
    return stream_run_impl(prm, field4, seeds, nseed, sublist, nsub, npts, seed_index, lines_cap, xyz, points_cap, nlines_out, npoints_out, nullptr, stream);
} FIB_API_CATCH

// [r5] fibd_stream_run without its host round trip: a step of a stream of volumes ends with a 16-byte download and a synchronisation
// during which the GPU idles (~45 us of C4's 1.03 ms between back-to-back calls); here the counts stay on the device.
extern "C" int fibd_stream_run_enqueue(const fib_stream_params *prm, const float *field4, const int64_t *seeds, int64_t nseed,
                                       const float *sublist, int32_t nsub, int32_t *npts, int64_t *seed_index, int64_t lines_cap,
                                       float *xyz, int64_t points_cap, int64_t *counts_dev, void *stream) try {
    FIB_CHECK(counts_dev != nullptr, FIB_ERR_INVALID, "NULL argument");
    return stream_run_impl(prm, field4, seeds, nseed, sublist, nsub, npts, seed_index, lines_cap, xyz, points_cap, nullptr, nullptr, counts_dev, stream);
} FIB_API_CATCH

extern "C" int fibd_stream_all_npts(fib_stream_job *job, int32_t *all_npts, void *stream) try {
    FIB_CHECK(job != nullptr, FIB_ERR_INVALID, "job is NULL");
    if (job->nlines == 0) return FIB_OK;
    FIB_CHECK(all_npts != nullptr, FIB_ERR_INVALID, "NULL output buffer");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(job->device));
    job->last_stream = (hipStream_t)stream;
    hipLaunchKernelGGL(copy_i32_kernel, dim3((unsigned)fib::cdiv(job->nlines, 256)), dim3(256), 0, (hipStream_t)stream,
                       job->npts.p, all_npts, job->nlines);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH
