// odf.hip — K2..K5: GQI / DSI ODF reconstruction, ODF peak finder, global QA normalisation (gfx950).
//
//   K2/K5  odf_gemm3_kernel  O[M x Nvox] = A[M x K] * clamp(S[K x Nvox]) as an exact f32 contraction on the bf16 matrix
//                            cores (three bf16 pieces per operand, v_mfma_f32_32x32x16_bf16; default), with the DSI
//                            antipodal fold fused into the sample load (FOLD)
//          odf_gemm_kernel   the same contraction on v_mfma_f32_32x32x2_f32 (format f32 -- FIBERS_ODF_FORMAT=f32 -- and every plan the
//                            split kernel does not take)
//                            (gqi.jl:139-145 `mul!(o, A, s)`; dsi.jl:204-246 recast as two dense maps)
//          mask_compact_kernel, odf_post_kernel, odf_inf_fix_kernel: voxel-list compaction + outputs outside the mask (one launch),
//                            columns of voxels with a +Inf sample
//   K3     odf_peaks642_kernel (sphere_642, specialised scan + candidate lists), odf_peaks64_kernel (any tessellation),
//          odf_peaks_kernel (32-voxel tiles): find_peaks! + peak/qa extraction (gqi.jl:147-159,180-201; dsi.jl:244-258)
//   K4     max-of-means reduction + qa ./= odfmax (gqi.jl:164-168; dsi.jl:263-267)
//
// GEMM design (both kernels).  M (ODF vertices, 321 for sphere_642) is small, K (frames, 270) is small, N (voxels,
// 2.7 M) is huge and contiguous in memory for both S (planar frames) and O (planar vertices).  So the
// voxel index sits on the MFMA lane (N = column): a wave owns 32 voxels and ALL rows of its M tile;
// accumulators stay in registers for the whole K loop (MB blocks of 32x32 = 16*MB VGPRs), S is read
// exactly once straight into VGPRs as the B operand (two 128-B segments per wave load), and the only
// shared operand, the matrix A (347 KB: larger than LDS), is streamed K-tile by K-tile through a
// double-buffered LDS ring with direct-to-LDS loads (global_load_lds), laid out so that the A-fragment
// reads are bank-conflict free.  Both kernels are exact f32 (a k-ordered fma chain / exact piece products):
// the ODF matches a CPU sgemv to rounding, which the strict-inequality peak finder needs.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KT = 16;        // frames per LDS stage
constexpr int WG_VOX = 128;   // voxels per workgroup (4 waves x 32)

#include "odf_gemm_f32.inc"
#include "odf_fused_epilogue.inc"
#include "odf_gemm3.inc"
#include "odf_mask.inc"
#include "odf_peaks.inc"
#include "odf_post.inc"

}  // namespace

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
struct fib_odf_plan {
    int device = 0;
    int nvol = 0, nvert = 0, nrows = 0, nrow0 = 0;   // nrow0 = rows that go to the pdf output (DSI), else 0
    int MB = 11, NX = 0, ntile_m = 1, Kpad = 0, maxdeg = 0;   // M tile = MB 32-row MFMA blocks + NX VALU rows
    int scale_frame = -1;
    float scale_coef = 0.0f;
    bool has_ineff = false;                          // some frame never reaches the model (DSI duplicates)
    // what the GEMM actually runs: G [gM x gK] (== A unless the DSI lattice is folded by symmetry)
    int gM = 0, gK = 0, gRow0 = 0;
    std::vector<float> G;
    bool folded = false;
    int fold_span_max = 0, scale_frame_raw = -1;     // fused fold: most frames a 16-sample stage spans on one side; raw frame of sum(p)
    fib::DevBuf<int32_t> foldA, foldB;               // [gK] frames summed into folded sample J; [gRow0] == same tables map pdf rows
    mutable fib::DevBuf<float> folded_dwi;           // [gK x nvox] scratch of the fold pre-pass (grow-only)
    std::vector<float> A;                            // host copy [nrows x nvol] column-major
    fib::DevBuf<float> At, verts;
    fib::DevBuf<uint16_t> At3;                       // split-bf16 image of G (odf_gemm3_kernel), empty in f32-MFMA mode
    fib::DevBuf<float> Aextra;                       // f32 coefficients of the NX extra rows [ntile_m][NX][Kpad]
    fib::DevBuf<float> Gdev;                         // G, column-major [gM x gK] (odf_inf_fix_kernel)
    mutable fib::DevBuf<int32_t> inf_list;           // [nvox] voxels with a +Inf sample (GQI, split-bf16 kernel; grow-only)
    bool split_bf16 = false;
    bool h2 = false;                                 // .. with two fp16 pieces per element (default) instead of three bf16 pieces (format bf16x3)
    float h2_sa = 1.0f;                              // power of two the matrix is scaled by before it is split into fp16 pieces
    bool fused_shape = false;                        // (GQI, 10 blocks + 1 extra row: the shape the fused scan is generated for)
    bool fused = false;                              // sphere_642 GQI plan: the contraction kernel finds the peaks on its accumulators
    bool dsi2_shape = false, dsi2 = false;           // folded DSI plan on sphere_642: odf_dsi2_kernel (fused ODF tile + pdf tile of MBB blocks)
    int MBB = 0;
    fib::DevBuf<uint16_t> At3b;                      // image of the pdf tile (the ODF tile's image / pole row: At3f / Aextraf)
    mutable fib::DevBuf<unsigned> pair_flags;        // item counters of the ODF-tile workgroups (odf_dsi2_kernel's pairing hint)
    fib::DevBuf<uint16_t> At3f;                      // split-bf16 image with the rows in the order of sphere642_fused.inc
    fib::DevBuf<float> Aextraf;                      // its extra row (the pole of the layout's rotation)
    mutable fib::DevBuf<float> mean_hi;              // [nvox] per-voxel upper bound of the mean (fused path)
    mutable fib::DevBuf<int32_t> redo_list;          // [nvox] voxels left to odf_redo_kernel
    fib::DevBuf<uint32_t> effbits;
    fib::DevBuf<int32_t> nbr, nbr64; // [nvert_even][deg_pad] LDS row of each neighbour (sentinel-padded)
    int deg_pad = 6, rows_pad = 0;
    bool is_s642 = false;                            // neighbour table == the compiled-in sphere_642 table (specialised scan)
    mutable fib::DevBuf<unsigned> maxenc;
    mutable fib::DevBuf<int32_t> live_vox, live_tiles, live_counts;   // mask compaction scratch (grow-only), counts = {voxels, tiles, +Inf voxels}
    mutable fib::DevBuf<unsigned long long> compact_state;   // the chunk granules of mask_compact_kernel [1024]
    mutable fib::DevBuf<unsigned long long> compact_sub;     // .. and its per-sub-chunk hints for the workgroups that clear outputs
    mutable fib::DevBuf<unsigned long long> compact_state2;  // .. the chunks' counts under both list units [1024]
    mutable fib::DevBuf<unsigned> compact_mode;              // .. and the unit of the next calls [2] (mask_compact_kernel)
    mutable fib::DevBuf<unsigned long long> compact_oct;     // .. which octets hold a voxel inside the mask (granules of 32 octets), for the clearing workgroups
    mutable unsigned compact_epoch = 0;              // .. and the call counter they are tagged with
    fib::DevBuf<unsigned> tickets;                   // [4]: chunk dispenser of mask_compact_kernel, arrival counter of odf_post_kernel (both 0 between calls), odf_post_kernel's divisor granule
    fib::DevBuf<unsigned> selctr;                    // [2] head / tail of odf_post_kernel's list of voxels whose exact mean is wanted (0 between calls)
    fib::DevBuf<unsigned long long> sellist;         // [POST_SELCAP] its entries, tagged with the call's epoch
    mutable fib::DevBuf<float> odfmax;
};

namespace {


// FIB_ODF_FORMAT_DEFAULT -> what the environment asks for (FIBERS_ODF_FORMAT = fp16x2 | bf16x3 | f32), else two fp16 pieces
// (the names rounds 2-4 used, FIBERS_ODF_GEMM=f32 and FIBERS_ODF_EXACT=1, are honoured: a stale setup must not change numerics silently)
int resolve_format(int format) {
    if (format != FIB_ODF_FORMAT_DEFAULT) return format;
    const char *e = fib::env("FIBERS_ODF_FORMAT");
    if (!e) {
        const char *g = fib::env("FIBERS_ODF_GEMM"), *x = fib::env("FIBERS_ODF_EXACT");
        if (g && (!strcmp(g, "f32") || !strcmp(g, "F32"))) return FIB_ODF_FORMAT_F32;
        if (x && x[0] != '\0' && x[0] != '0') return FIB_ODF_FORMAT_BF16X3;
    }
    if (e && (!strcmp(e, "f32") || !strcmp(e, "F32"))) return FIB_ODF_FORMAT_F32;
    if (e && (!strcmp(e, "bf16x3") || !strcmp(e, "BF16X3"))) return FIB_ODF_FORMAT_BF16X3;
    return FIB_ODF_FORMAT_FP16X2;
}

int finish_plan(fib_odf_plan *p, const float *verts, int nverts, const int32_t *faces, int nfaces,
                const std::vector<float> &frame_eff, int format = FIB_ODF_FORMAT_DEFAULT) {
    if (p->G.empty()) { p->G = p->A; p->gM = p->nrows; p->gK = p->nvol; p->gRow0 = p->nrow0; }
    const int M = p->gM, K = p->gK;
    // the operand format of the contraction (header: FIB_ODF_FORMAT_*): two fp16 pieces per f32 operand (default), three exact
    // bf16 pieces, or v_mfma_f32_32x32x2_f32 (a k-ordered f32 fma chain, bit-identical to the oracle's loop)
    format = resolve_format(format);
    p->split_bf16 = format != FIB_ODF_FORMAT_F32;
    // pick (MB, NX) minimising the per-k-step issue cost ntile*(64*MB + 4*NX) cycles (MFMA block = 64, v_fmac = 4)
    int best_cost = INT32_MAX;
    const int nxs[] = {0, 1};                            // (tiles with 2 or 4 VALU rows never won for a shape in use: 13 variants instead of 25)
    for (int mb = p->split_bf16 ? 10 : 11; mb >= 5; mb--)
        for (int nx : nxs) {
            if (nx > 0 && mb > 10) continue;            // register budget
            const int rows = mb * 32 + nx;
            const int nt = (M + rows - 1) / rows;
            const int cost = nt * (64 * mb + 4 * nx);
            if (cost < best_cost) { best_cost = cost; p->MB = mb; p->NX = nx; p->ntile_m = nt; }
        }
    p->Kpad = (K + KT - 1) / KT * KT;
    const int MW = gemm_row_stride(p->MB, p->NX), ROWS = p->MB * 32 + p->NX;
    std::vector<float> At((size_t)p->ntile_m * p->Kpad * MW, 0.0f);
    for (int k = 0; k < K; k++)
        for (int r = 0; r < M; r++) {
            const int tm = r / ROWS, rr = r % ROWS;
            At[((size_t)tm * p->Kpad + k) * MW + rr] = p->G[r + (size_t)M * k];
        }
    // the split-bf16 kernel needs: every frame effective (DSI frames that share a lattice point are not), >= 2 stages,
    // MB <= 10 (LDS: two workgroups per CU), and the extra rows' coefficient table within its 4-KiB LDS slot
    bool any_ineff = false;
    for (int k = 0; k < K; k++) if (frame_eff[k] == 0.0f) any_ineff = true;
    if (p->split_bf16 && (any_ineff || p->Kpad / KT < 2 || p->MB > 10 || (size_t)p->ntile_m * p->NX * p->Kpad > 2048)) p->split_bf16 = false;
    if (p->split_bf16) {
        // piece format: two fp16 pieces of sa * G (sa = the power of two that puts max |G| into [2^8, 2^9): the second piece of an
        // element stays a normal fp16 number down to 2^-20 of the largest, gemm3_body H2), or three exact bf16 pieces
        p->h2 = format == FIB_ODF_FORMAT_FP16X2;
        float gmax = 0.0f;
        for (float v : p->G) if (std::isfinite(v)) gmax = std::max(gmax, std::fabs(v));
        if (p->h2 && !(gmax > 0.0f)) p->h2 = false;
        for (float v : p->G) if (!std::isfinite(v)) p->h2 = false;       // (a matrix with NaN / Inf entries keeps the exact split's behaviour)
        if (p->h2) { int eg; std::frexp(gmax, &eg); p->h2_sa = std::ldexp(1.0f, 9 - eg); }   // gmax = m 2^eg, m in [0.5, 1)
        const int NP = p->h2 ? 2 : 3;
        const int npiece = NP * p->MB, nst = p->Kpad / KT;
        auto f16_rn = [](float f) -> uint16_t { const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; };
        auto f16_f = [](uint16_t u) -> float { _Float16 h; memcpy(&h, &u, 2); return (float)h; };
        auto bf16_rn = [](float f) -> uint16_t {
            uint32_t u; memcpy(&u, &f, 4);
            if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
            u += 0x7fffu + ((u >> 16) & 1u);
            return (uint16_t)(u >> 16);
        };
        auto bf16_f = [](uint16_t h) -> float { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
        auto pieces = [&](float v, uint16_t (&hs)[3]) {
            if (p->h2) {
                const float w = v * p->h2_sa;                          // exact
                hs[0] = f16_rn(w);
                hs[1] = f16_rn(w - f16_f(hs[0]));                       // (the difference is exact)
                hs[2] = 0;
            } else {
                hs[0] = bf16_rn(v);
                const float r1 = v - bf16_f(hs[0]);
                hs[1] = bf16_rn(r1);
                hs[2] = bf16_rn(r1 - bf16_f(hs[1]));
            }
        };
        // image row `row` of the kernel holds row rowmap[row] of G (identity, or the layout of the fused peak scan)
        auto build = [&](const short *rowmap, std::vector<uint16_t> &A3, std::vector<float> &AX) {
            A3.assign((size_t)p->ntile_m * nst * npiece * 512, 0);
            AX.assign((size_t)std::max(1, p->ntile_m * p->NX * p->Kpad), 0.0f);
            for (int tm = 0; tm < p->ntile_m; tm++) {
                for (int t = 0; t < nst; t++) {
                    uint16_t *st = A3.data() + ((size_t)tm * nst + t) * npiece * 512;
                    for (int m = 0; m < p->MB; m++)
                        for (int l = 0; l < 64; l++)
                            for (int j = 0; j < 8; j++) {
                                const int row = tm * ROWS + m * 32 + (l & 31), k = t * KT + 8 * (l >> 5) + j;
                                if (row >= M || k >= K) continue;
                                uint16_t hs[3];
                                pieces(p->G[(rowmap ? rowmap[row] : row) + (size_t)M * k], hs);
                                for (int pc = 0; pc < NP; pc++) st[((size_t)(pc * p->MB + m) * 64 + l) * 8 + j] = hs[pc];
                            }
                }
                for (int x = 0; x < p->NX; x++)
                    for (int k = 0; k < K; k++) {
                        const int row = tm * ROWS + p->MB * 32 + x;
                        if (row < M) AX[((size_t)tm * p->NX + x) * p->Kpad + k] = p->G[(rowmap ? rowmap[row] : row) + (size_t)M * k];
                    }
            }
        };
        std::vector<uint16_t> A3;
        std::vector<float> AX;
        build(nullptr, A3, AX);
        int rc3 = p->At3.alloc(A3.size());
        if (rc3 != FIB_OK) return rc3;
        if (p->gRow0 == 0 && p->scale_frame < 0 && faces) {     // GQI: +Inf samples are repaired after the GEMM
            if ((rc3 = p->Gdev.alloc(p->G.size())) != FIB_OK) return rc3;
            FIB_HIP(hipMemcpy(p->Gdev.p, p->G.data(), p->G.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if ((rc3 = p->Aextra.alloc(AX.size())) != FIB_OK) return rc3;
        FIB_HIP(hipMemcpy(p->At3.p, A3.data(), A3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        FIB_HIP(hipMemcpy(p->Aextra.p, AX.data(), AX.size() * sizeof(float), hipMemcpyHostToDevice));
        // single-tile image of mb blocks (+ nx extra rows) whose image row r holds row rowfn(r) of G (< 0: padding)
        auto build_one = [&](int mb, int nx, auto rowfn, std::vector<uint16_t> &A3o, std::vector<float> &AXo) {
            A3o.assign((size_t)nst * NP * mb * 512, 0);
            AXo.assign((size_t)std::max(1, nx * p->Kpad), 0.0f);
            for (int t = 0; t < nst; t++) {
                uint16_t *st = A3o.data() + (size_t)t * NP * mb * 512;
                for (int m = 0; m < mb; m++)
                    for (int l = 0; l < 64; l++)
                        for (int j = 0; j < 8; j++) {
                            const int gr = rowfn(m * 32 + (l & 31)), k = t * KT + 8 * (l >> 5) + j;
                            if (gr < 0 || k >= K) continue;
                            uint16_t hs[3];
                            pieces(p->G[gr + (size_t)M * k], hs);
                            for (int pc = 0; pc < NP; pc++) st[((size_t)(pc * mb + m) * 64 + l) * 8 + j] = hs[pc];
                        }
            }
            for (int x = 0; x < nx; x++)
                for (int k = 0; k < K; k++) { const int gr = rowfn(mb * 32 + x); if (gr >= 0) AXo[(size_t)x * p->Kpad + k] = p->G[gr + (size_t)M * k]; }
        };
        // folded DSI on sphere_642: ODF tile in the fused scan's row order + pdf tile (odf_dsi2_kernel)
        static const int mbbs[] = {5, 7, 9};
        for (int mbb : mbbs) if (p->MBB == 0 && p->gRow0 <= mbb * 32) p->MBB = mbb;
        p->dsi2_shape = faces && p->folded && p->gRow0 > 0 && p->MBB > 0 && M == p->gRow0 + FQ_NV && p->scale_frame >= 0 && p->Kpad <= 512 && nst >= 2;
        if (p->dsi2_shape) {
            const int r0 = p->gRow0;
            build_one(10, 1, [&](int r) { return r < FQ_NV ? r0 + fib_f642_pos_vertex[r] : -1; }, A3, AX);
            if ((rc3 = p->At3f.alloc(A3.size())) != FIB_OK || (rc3 = p->Aextraf.alloc(AX.size())) != FIB_OK) return rc3;
            FIB_HIP(hipMemcpy(p->At3f.p, A3.data(), A3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            FIB_HIP(hipMemcpy(p->Aextraf.p, AX.data(), AX.size() * sizeof(float), hipMemcpyHostToDevice));
            build_one(p->MBB, 0, [&](int r) { return r < r0 ? r : -1; }, A3, AX);
            if ((rc3 = p->At3b.alloc(A3.size())) != FIB_OK || (rc3 = p->pair_flags.alloc(8 * 32)) != FIB_OK) return rc3;
            FIB_HIP(hipMemcpy(p->At3b.p, A3.data(), A3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        p->fused_shape = faces && p->gRow0 == 0 && p->scale_frame < 0 && M == FQ_NV && p->MB == 10 && p->NX == 1 && p->ntile_m == 1 && p->Kpad <= 512 &&
                         nst >= 3;                              // (the sample tiles are requested two stages ahead)
        if (p->fused_shape) {                                   // second image in the row order of sphere642_fused.inc
            build(fib_f642_pos_vertex, A3, AX);
            if ((rc3 = p->At3f.alloc(A3.size())) != FIB_OK || (rc3 = p->Aextraf.alloc(AX.size())) != FIB_OK) return rc3;
            FIB_HIP(hipMemcpy(p->At3f.p, A3.data(), A3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            FIB_HIP(hipMemcpy(p->Aextraf.p, AX.data(), AX.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    std::vector<int32_t> nbr32;
    int rc = FIB_OK;
    if (faces) {
        rc = fib::host_neighbours(faces, nfaces, nverts, nbr32, &p->maxdeg);
        if (rc != FIB_OK) return rc;
    } else {                                            // matrix-only plan (fib::matrix_plan_create): no peak finder tables
        p->maxdeg = 0;
    }
    p->deg_pad = p->maxdeg <= 6 ? 6 : (p->maxdeg <= 8 ? 8 : 16);
    p->rows_pad = (p->nvert + 7) / 8 * 8;
    const int nv_even = (p->nvert + 1) / 2 * 2;
    std::vector<int32_t> nbr((size_t)nv_even * p->deg_pad, p->rows_pad);          // default: sentinel row
    for (int v = 0; v < p->nvert; v++)
        for (int d = 0; d < p->maxdeg; d++) {
            const int32_t u = nbr32[(size_t)v * p->maxdeg + d];
            if (u >= 0) nbr[(size_t)v * p->deg_pad + d] = u;
        }
    // the default tessellation has a scan specialised at compile time (sphere642_scan.inc): use it iff the tables agree
    p->is_s642 = p->nvert == FIB_S642_NVERT && p->maxdeg <= FIB_S642_DEG;
    for (int v = 0; v < p->nvert && p->is_s642; v++) {
        std::vector<int32_t> mine, ref;
        for (int d = 0; d < p->maxdeg; d++) { const int32_t u = nbr32[(size_t)v * p->maxdeg + d]; if (u >= 0) mine.push_back(u); }
        for (int d = 0; d < FIB_S642_DEG; d++) if (fib_s642_nbr[v][d] < FIB_S642_NVERT) ref.push_back(fib_s642_nbr[v][d]);
        std::sort(mine.begin(), mine.end());
        if (mine != ref) p->is_s642 = false;
    }
    p->fused = p->fused_shape && p->is_s642 && p->At3f.p != nullptr;
    p->dsi2 = p->dsi2_shape && p->is_s642 && p->At3f.p != nullptr && p->At3b.p != nullptr;
    std::vector<int32_t> nbr64(nbr);
    for (auto &u : nbr64) if (u == p->rows_pad) u = p->nvert;
    if ((rc = p->nbr64.alloc(nbr64.size())) != FIB_OK) return rc;
    FIB_HIP(hipMemcpy(p->nbr64.p, nbr64.data(), nbr64.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    std::vector<float> v3((size_t)p->nvert * 3, 0.0f);
    if (verts)
        for (int v = 0; v < p->nvert; v++)
            for (int c = 0; c < 3; c++) v3[3 * v + c] = verts[v + (size_t)nverts * c];
    if ((rc = p->At.alloc(At.size())) != FIB_OK) return rc;
    std::vector<uint32_t> effbits((size_t)p->Kpad / KT, 0u);
    for (int k = 0; k < K; k++) { if (frame_eff[k] != 0.0f) effbits[k / KT] |= 1u << (k % KT); else p->has_ineff = true; }
    if ((rc = p->effbits.alloc(effbits.size())) != FIB_OK) return rc;
    if ((rc = p->verts.alloc(v3.size())) != FIB_OK) return rc;
    if ((rc = p->nbr.alloc(nbr.size())) != FIB_OK) return rc;
    if ((rc = p->maxenc.alloc(4)) != FIB_OK) return rc;
    if ((rc = p->live_counts.alloc(4)) != FIB_OK) return rc;
    if ((rc = p->tickets.alloc(4)) != FIB_OK) return rc;
    FIB_HIP(hipMemset(p->tickets.p, 0, 4 * sizeof(unsigned)));
    if ((rc = p->selctr.alloc(2)) != FIB_OK) return rc;
    FIB_HIP(hipMemset(p->selctr.p, 0, 2 * sizeof(unsigned)));
    if ((rc = p->sellist.alloc(POST_SELCAP)) != FIB_OK) return rc;
    FIB_HIP(hipMemset(p->sellist.p, 0, POST_SELCAP * sizeof(unsigned long long)));
    if ((rc = p->odfmax.alloc(2)) != FIB_OK) return rc;
    FIB_HIP(hipMemcpy(p->At.p, At.data(), At.size() * sizeof(float), hipMemcpyHostToDevice));
    FIB_HIP(hipMemcpy(p->effbits.p, effbits.data(), effbits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    FIB_HIP(hipMemcpy(p->verts.p, v3.data(), v3.size() * sizeof(float), hipMemcpyHostToDevice));
    FIB_HIP(hipMemcpy(p->nbr.p, nbr.data(), nbr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return FIB_OK;
}

int check_plan_args(const float *bval, const float *bvec, int nvol, const float *verts, int nverts,
                    const int32_t *faces, int nfaces, fib_odf_plan **plan) {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan output pointer is NULL");
    *plan = nullptr;
    FIB_CHECK(bval != nullptr && nvol > 0, FIB_ERR_MISSING_BVAL, "Missing b-value table from input DWI structure");
    FIB_CHECK(bvec != nullptr, FIB_ERR_MISSING_BVEC, "Missing gradient table from input DWI structure");
    FIB_CHECK(verts && faces && nverts >= 2 && nverts % 2 == 0 && nfaces > 0, FIB_ERR_INVALID, "invalid ODF tessellation");
    FIB_CHECK(nverts / 2 < 32768, FIB_ERR_UNSUPPORTED, "too many ODF vertices");
    return FIB_OK;
}

}  // namespace

extern "C" int fib_gqi_plan_create(int device, const float *bval, const float *bvec, int nvol,
                                   const float *verts, int nverts, const int32_t *faces, int nfaces,
                                   float sigma, fib_odf_plan **plan) {
    return fib_gqi_plan_create_fmt(device, bval, bvec, nvol, verts, nverts, faces, nfaces, sigma, FIB_ODF_FORMAT_DEFAULT, plan);
}

extern "C" int fib_odf_default_format(void) try { return resolve_format(FIB_ODF_FORMAT_DEFAULT); } FIB_API_CATCH

extern "C" int fib_odf_plan_format(const fib_odf_plan *plan) try {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan is NULL");
    return !plan->split_bf16 ? FIB_ODF_FORMAT_F32 : (plan->h2 ? FIB_ODF_FORMAT_FP16X2 : FIB_ODF_FORMAT_BF16X3);
} FIB_API_CATCH

// the unit of the voxel list the plan's NEXT reconstruction call will use (mask_compact_kernel chooses it from the previous call's mask):
// 1 = aligned groups of 32 voxels ("octets" of quads: a wave's 128-byte row segments are whole cache lines), 0 = aligned groups of 4
extern "C" int fib_odf_plan_list_unit(const fib_odf_plan *plan, void *stream) try {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan is NULL");
    { const char *e = fib::ab_env("FIBERS_ODF_LIST"); if (e && (e[0] == 'q' || e[0] == 'o')) return e[0] == 'o' ? 1 : 0; }
    if (!plan->compact_mode.p) return 1;                       // (no call yet)
    fib::DeviceGuard guard;
    { const int rcd = fib::use_device(plan->device); if (rcd != FIB_OK) return rcd; }
    unsigned w[2] = {1u, 1u};
    FIB_HIP(hipMemcpyAsync(w, plan->compact_mode.p, sizeof w, hipMemcpyDeviceToHost, (hipStream_t)stream));
    FIB_HIP(hipStreamSynchronize((hipStream_t)stream));
    return (int)(w[(plan->compact_epoch + 1u) & 1u] != 0u);
} FIB_API_CATCH

extern "C" int fib_gqi_plan_create_fmt(int device, const float *bval, const float *bvec, int nvol,
                                       const float *verts, int nverts, const int32_t *faces, int nfaces,
                                       float sigma, int format, fib_odf_plan **plan) try {
    int rc = check_plan_args(bval, bvec, nvol, verts, nverts, faces, nfaces, plan);
    if (rc != FIB_OK) return rc;
    FIB_CHECK(format >= FIB_ODF_FORMAT_DEFAULT && format <= FIB_ODF_FORMAT_F32, FIB_ERR_INVALID, "unknown operand format %d", format);
    fib::DeviceGuard guard;
    if ((rc = fib::use_device(device)) != FIB_OK) return rc;
    fib_odf_plan *p = new (std::nothrow) fib_odf_plan();
    FIB_CHECK(p != nullptr, FIB_ERR_NOMEM, "out of host memory");
    p->device = device; p->nvol = nvol; p->nvert = nverts / 2; p->nrows = p->nvert; p->nrow0 = 0;
    p->A.resize((size_t)p->nrows * nvol);
    fib::host_gqi_matrix(bval, bvec, nvol, verts, nverts, sigma, p->A.data());
    std::vector<float> eff((size_t)nvol, 1.0f);
    rc = finish_plan(p, verts, nverts, faces, nfaces, eff, format);
    if (rc != FIB_OK) { delete p; return rc; }
    *plan = p;
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fib_dsi_plan_create(int device, const float *bval, const float *bvec, int nvol,
                                   const float *verts, int nverts, const int32_t *faces, int nfaces,
                                   int hann_width, fib_odf_plan **plan) {
    return fib_dsi_plan_create_fmt(device, bval, bvec, nvol, verts, nverts, faces, nfaces, hann_width, FIB_ODF_FORMAT_DEFAULT, plan);
}

extern "C" int fib_dsi_plan_create_fmt(int device, const float *bval, const float *bvec, int nvol,
                                       const float *verts, int nverts, const int32_t *faces, int nfaces,
                                       int hann_width, int format, fib_odf_plan **plan) try {
    int rc = check_plan_args(bval, bvec, nvol, verts, nverts, faces, nfaces, plan);
    if (rc != FIB_OK) return rc;
    FIB_CHECK(format >= FIB_ODF_FORMAT_DEFAULT && format <= FIB_ODF_FORMAT_F32, FIB_ERR_INVALID, "unknown operand format %d", format);
    FIB_CHECK(hann_width >= 0, FIB_ERR_INVALID, "hann_width must be >= 0");
    fib::DeviceGuard guard;
    if ((rc = fib::use_device(device)) != FIB_OK) return rc;
    fib_odf_plan *p = new (std::nothrow) fib_odf_plan();
    FIB_CHECK(p != nullptr, FIB_ERR_NOMEM, "out of host memory");
    p->device = device; p->nvol = nvol; p->nvert = nverts / 2; p->nrows = nvol + p->nvert; p->nrow0 = nvol;
    p->A.resize((size_t)p->nrows * nvol);
    std::vector<int> iq;
    rc = fib::host_dsi_matrix(bval, bvec, nvol, verts, nverts, hann_width, p->A.data(), &p->scale_frame, &p->scale_coef, &iq);
    if (rc != FIB_OK) { delete p; return rc; }
    // a frame overwritten by a later one on the same lattice point never reaches X (dsi.jl:205): its column is 0
    std::vector<float> eff((size_t)nvol, 0.0f);
    for (int j = 0; j < nvol; j++)
        for (int r = 0; r < p->nrows; r++) if (p->A[r + (size_t)p->nrows * j] != 0.0f) { eff[j] = 1.0f; break; }
    if (p->scale_frame < 0) { p->scale_frame = 0; p->scale_coef = 0.0f; }   // no q=0 sample: sum(p) = 0 -> Inf/NaN
    // ---- antipodal folding (see dsi_fold_kernel): every frame needs a partner at -q with an identical column ----
    {
        std::vector<int> partner(nvol, -1);
        bool ok = true;
        for (int j = 0; j < nvol && ok; j++) if (eff[j] == 0.0f) ok = false;
        for (int j = 0; j < nvol && ok; j++) {
            for (int k = 0; k < nvol; k++)
                if (iq[3 * k] == -iq[3 * j] && iq[3 * k + 1] == -iq[3 * j + 1] && iq[3 * k + 2] == -iq[3 * j + 2]) { partner[j] = k; break; }
            if (partner[j] < 0) ok = false;
        }
        for (int j = 0; j < nvol && ok; j++) {
            const float *cj = &p->A[(size_t)p->nrows * j], *ck = &p->A[(size_t)p->nrows * partner[j]];
            for (int r = 0; r < p->nrows && ok; r++) if (cj[r] != ck[r]) ok = false;   // cos is even: columns must be identical
        }
        if (ok) {
            std::vector<int32_t> fa, fb;
            for (int j = 0; j < nvol; j++) if (j <= partner[j]) { fa.push_back(j); fb.push_back(partner[j] == j ? -1 : partner[j]); }
            const int nrep = (int)fa.size();
            p->folded = true; p->gK = nrep; p->gRow0 = nrep; p->gM = nrep + p->nvert;
            p->G.assign((size_t)p->gM * nrep, 0.0f);
            for (int c = 0; c < nrep; c++) {
                const float *col = &p->A[(size_t)p->nrows * fa[c]];
                for (int r = 0; r < nrep; r++) p->G[r + (size_t)p->gM * c] = col[fa[r]];
                for (int v = 0; v < p->nvert; v++) p->G[nrep + v + (size_t)p->gM * c] = col[nvol + v];
                if (fa[c] == p->scale_frame) p->scale_frame = -1000 - c;          // re-index below
            }
            if (p->scale_frame <= -1000) { p->scale_frame = -(p->scale_frame + 1000); p->scale_frame_raw = fa[p->scale_frame]; }
            for (int t0 = 0; t0 < nrep; t0 += KT)
                for (const std::vector<int32_t> *side : {&fa, &fb}) {
                    int lo = 1 << 30, hi = -1;
                    for (int j = t0; j < std::min(nrep, t0 + KT); j++) if ((*side)[j] >= 0) { lo = std::min(lo, (*side)[j]); hi = std::max(hi, (*side)[j]); }
                    if (hi >= 0) p->fold_span_max = std::max(p->fold_span_max, hi - lo + 1);
                }
            if ((rc = p->foldA.alloc(nrep)) != FIB_OK || (rc = p->foldB.alloc(nrep)) != FIB_OK) { delete p; return rc; }
            (void)hipMemcpy(p->foldA.p, fa.data(), nrep * sizeof(int32_t), hipMemcpyHostToDevice);
            (void)hipMemcpy(p->foldB.p, fb.data(), nrep * sizeof(int32_t), hipMemcpyHostToDevice);
            eff.assign((size_t)nrep, 1.0f);
        }
    }
    rc = finish_plan(p, verts, nverts, faces, nfaces, eff, format);
    if (rc != FIB_OK) { delete p; return rc; }
    *plan = p;
    return FIB_OK;
} FIB_API_CATCH

extern "C" void fib_odf_plan_destroy(fib_odf_plan *plan) try {
    if (!plan) return;
    fib::DeviceGuard guard;
    (void)hipSetDevice(plan->device);
    delete plan;
} FIB_API_CATCH_VOID

extern "C" int fib_odf_plan_matrix(const fib_odf_plan *plan, float *A, int *nrows, int *nvol, int *nvert) try {
    FIB_CHECK(plan != nullptr, FIB_ERR_INVALID, "plan is NULL");
    if (nrows) *nrows = plan->nrows;
    if (nvol) *nvol = plan->nvol;
    if (nvert) *nvert = plan->nvert;
    if (A) memcpy(A, plan->A.data(), plan->A.size() * sizeof(float));
    return FIB_OK;
} FIB_API_CATCH

// ------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------
namespace {

constexpr int FOLD_MB_MAX = 10;   // M tiles that the fused-fold variant is built for (room for 16 raw samples per lane and stage)
template <int MB, int NX>
void launch_gemm(const GemmArgs &ga, unsigned grid, hipStream_t st) {
    if (ga.At3) {
        // persistent grid: one 8-wave workgroup per CU (two 4-wave workgroups per CU were measured 6 % slower), a multiple of 8
        // so that blockIdx & 7 is the XCD
        int ncu = 256, dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        GemmArgs g2 = ga;
        const int nw = 8;
        const int64_t items = fib::cdiv(ga.nvox, nw * 32) * ga.ntile_m;
        unsigned pg = (unsigned)std::min<int64_t>((int64_t)ncu * (8 / nw), items);
        pg = (pg + 7) / 8 * 8;
        auto go = [&](auto h2_) {
            constexpr bool H2 = decltype(h2_)::value;
            if (ga.fold) {
                if constexpr (MB <= FOLD_MB_MAX) hipLaunchKernelGGL((odf_gemm3_kernel<MB, NX, 8, true, false, H2>), dim3(pg), dim3(512), 0, st, g2);
                return;
            }
            if constexpr (MB == 10 && NX == 1) {
                if (ga.mean_hi) { hipLaunchKernelGGL((odf_gemm3_kernel<10, 1, 8, false, true, H2>), dim3(pg), dim3(512), 0, st, g2); return; }
            }
            hipLaunchKernelGGL((odf_gemm3_kernel<MB, NX, 8, false, false, H2>), dim3(pg), dim3(512), 0, st, g2);
        };
        if (ga.h2) go(std::true_type{}); else go(std::false_type{});
    }
    else hipLaunchKernelGGL((odf_gemm_kernel<MB, NX>), dim3(grid), dim3(256), 0, st, ga);
}

size_t peaks_smem(const fib_odf_plan *p) {
    return ((size_t)(p->rows_pad + 1) * PV + (size_t)PG * PV * PREC) * sizeof(float);
}

template <int DEG, bool EXACT>
int launch_peaks_t(const PeakArgs &pa, size_t smem, unsigned grid, hipStream_t st) {
    FIB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(odf_peaks_kernel<DEG, EXACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((odf_peaks_kernel<DEG, EXACT>), dim3(grid), dim3(PW * 64), smem, st, pa);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
}

template <int DEG, bool EXACT>
int launch_peaks64_t(const PeakArgs &pa, size_t smem, int64_t ntiles, unsigned grid, hipStream_t st) {
    FIB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(odf_peaks64_kernel<DEG, EXACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((odf_peaks64_kernel<DEG, EXACT>), dim3(grid), dim3(P64_T), smem, st, pa, ntiles);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
}

int launch_peaks(const fib_odf_plan *plan, const float *odf, int64_t nvox, int64_t stride, float *const peak[3], float *const qa[3],
                 int32_t *isort_top, int32_t *nvalid, bool reduce, hipStream_t st, bool small_tiles = false,
                 const int32_t *tiles = nullptr, const int32_t *ntl = nullptr) {
    PeakArgs pa{};
    pa.tiles = tiles; pa.ntl = ntl;
    pa.odf = odf; pa.nbr = plan->nbr.p; pa.nbr64 = plan->nbr64.p; pa.verts = plan->verts.p;
    for (int k = 0; k < 3; k++) { pa.peak[k] = peak ? peak[k] : nullptr; pa.qa[k] = qa ? qa[k] : nullptr; }
    pa.isort_top = isort_top; pa.nvalid = nvalid;
    pa.maxenc = reduce ? plan->maxenc.p : nullptr;
    pa.mean_hi = reduce ? plan->mean_hi.p : nullptr;
    pa.nvox = nvox; pa.stride = stride; pa.nvert = plan->nvert; pa.rows_pad = plan->rows_pad;
    pa.vec_ok = (stride % 4 == 0 && ((uintptr_t)odf & 15) == 0) ? 1 : 0;
    const size_t smem = peaks_smem(plan);
    FIB_CHECK(smem <= 160 * 1024, FIB_ERR_UNSUPPORTED, "ODF with %d vertices does not fit the peak finder's LDS tile", plan->nvert);
    const unsigned grid = (unsigned)fib::cdiv(nvox, PV);
    const bool exact = isort_top != nullptr;
    fib::ProfScope prof("odf_peaks", st);
    const size_t smem64 = ((size_t)(plan->nvert + 1) * 64 + (size_t)P64_W * 64 * PREC + (size_t)plan->nvert * (plan->deg_pad + 3)) * sizeof(float);
    if (!small_tiles && plan->nvert * 16 <= P64_NI * P64_T && smem64 <= 160 * 1024) {
        int ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, plan->device);
        const int64_t ntiles = fib::cdiv(nvox, 64);
        const unsigned g64 = (unsigned)std::min<int64_t>(ntiles, ncu);
        if (plan->is_s642 && !exact) {
            const size_t smem642 = ((size_t)(FIB_S642_NVERT + 1) * 64 + (size_t)PQ_CAP * 64 * 2 + 128 + 2 * P64_W * 64 + 2 * 8 * 64 +
                                    (size_t)FIB_S642_NVERT * 3) * sizeof(float);
            FIB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(odf_peaks642_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem642));
            hipLaunchKernelGGL(odf_peaks642_kernel, dim3(g64), dim3(P64_T), smem642, st, pa, ntiles);
            FIB_HIP(hipGetLastError());
            return FIB_OK;
        }
        switch (plan->deg_pad) {
            case 6:  return exact ? launch_peaks64_t<6, true>(pa, smem64, ntiles, g64, st) : launch_peaks64_t<6, false>(pa, smem64, ntiles, g64, st);
            case 8:  return exact ? launch_peaks64_t<8, true>(pa, smem64, ntiles, g64, st) : launch_peaks64_t<8, false>(pa, smem64, ntiles, g64, st);
            default: return exact ? launch_peaks64_t<16, true>(pa, smem64, ntiles, g64, st) : launch_peaks64_t<16, false>(pa, smem64, ntiles, g64, st);
        }
    }
    switch (plan->deg_pad) {
        case 6:  return exact ? launch_peaks_t<6, true>(pa, smem, grid, st) : launch_peaks_t<6, false>(pa, smem, grid, st);
        case 8:  return exact ? launch_peaks_t<8, true>(pa, smem, grid, st) : launch_peaks_t<8, false>(pa, smem, grid, st);
        default: return exact ? launch_peaks_t<16, true>(pa, smem, grid, st) : launch_peaks_t<16, false>(pa, smem, grid, st);
    }
}

}  // namespace

namespace {
// the one-launch mask compaction (mask_compact_kernel) on the plan's scratch; z != NULL: also clear those outputs outside the mask
int launch_mask_compact(const fib_odf_plan *plan, const uint8_t *mask, int64_t nvox, unsigned *maxenc, const ZeroArgs *z, hipStream_t st) {
    // at most 256 chunks (one ticket each: returning atomics on one address are served at ~90 per microsecond), all resident at once
    int iters = (int)std::max<int64_t>(1, fib::cdiv(fib::cdiv(nvox, CB), 256));
    if (iters > CB_ITERS_MAX) iters = CB_ITERS_MAX;
    const int nchunks = (int)fib::cdiv(nvox, (int64_t)CB * iters);                      // <= 1024 (2^27 voxels)
    FIB_CHECK(nchunks <= 1024, FIB_ERR_UNSUPPORTED, "volume too large for the mask compaction");
    int rc;
    if ((rc = plan->live_vox.ensure((size_t)nvox)) != FIB_OK) return rc;
    if ((rc = plan->live_tiles.ensure((size_t)fib::cdiv(nvox, 64))) != FIB_OK) return rc;
    if (!plan->compact_state.p) {                                                       // fresh granules carry epoch 0 = "never written"
        if ((rc = plan->compact_state.alloc(1024)) != FIB_OK) return rc;
        FIB_HIP(hipMemsetAsync(plan->compact_state.p, 0, 1024 * sizeof(unsigned long long), st));
    }
    if (!plan->compact_state2.p) {
        if ((rc = plan->compact_state2.alloc(1024)) != FIB_OK) return rc;
        FIB_HIP(hipMemsetAsync(plan->compact_state2.p, 0, 1024 * sizeof(unsigned long long), st));
        if ((rc = plan->compact_mode.alloc(2)) != FIB_OK) return rc;
        FIB_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(plan->compact_mode.p), 1, 2, st));   // octets until a call's counts say otherwise
    }
    const size_t nsub = (size_t)fib::cdiv(nvox, CB);
    if (z && plan->compact_sub.n < nsub) {
        if ((rc = plan->compact_sub.alloc(nsub)) != FIB_OK) return rc;
        FIB_HIP(hipMemsetAsync(plan->compact_sub.p, 0, nsub * sizeof(unsigned long long), st));
    }
    if (++plan->compact_epoch == 0u) plan->compact_epoch = 1u;
    CompactArgs c{};
    c.mask = mask; c.nvox = nvox; c.vidx = plan->live_vox.p; c.tiles = plan->live_tiles.p; c.state = plan->compact_state.p;
    c.ticket = plan->tickets.p; c.totals = plan->live_counts.p; c.maxenc = maxenc; c.epoch = plan->compact_epoch; c.nchunks = nchunks; c.iters = iters;
    c.zero = z != nullptr;
    if (z) c.z = *z;
    c.clear = plan->pair_flags.p; c.nclear = plan->pair_flags.p ? 8 * 32 : 0;
    c.sub_state = plan->compact_sub.p;
    c.state2 = plan->compact_state2.p; c.mode = plan->compact_mode.p;
    const size_t noct = (size_t)fib::cdiv(nvox, 1024);
    if (z && noct * sizeof(unsigned) <= 48 * 1024) {           // (the clearing workgroups keep the bits in LDS: volumes up to 12.5 M voxels)
        if (plan->compact_oct.n < noct) {
            if ((rc = plan->compact_oct.alloc(noct)) != FIB_OK) return rc;
            FIB_HIP(hipMemsetAsync(plan->compact_oct.p, 0, noct * sizeof(unsigned long long), st));
        }
        c.oct_bits = plan->compact_oct.p; c.oct_words = (int)noct;
    }
    { const char *e = fib::ab_env("FIBERS_COMPACT_UNKNOWN"); c.dbg_unknown = e ? atoi(e) : 0; }
    { const char *e = fib::ab_env("FIBERS_ODF_LIST"); c.force = !e ? -1 : (e[0] == 'q' ? 0 : (e[0] == 'o' ? 1 : -1)); }   // quads | octets | auto
    fib::ProfScope prof("mask_compact", st);
    // with outputs to clear: helpers behind the compacting workgroups, so that a volume that is mostly outside the mask is cleared by the whole chip
    const int grid = nchunks + (z ? 256 : 0);
    hipLaunchKernelGGL(mask_compact_kernel, dim3(grid), dim3(1024), c.oct_words * sizeof(unsigned), st, c);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
}
}  // namespace

extern "C" int fibd_odf_rec(const fib_odf_plan *plan, const float *dwi, const uint8_t *mask, int64_t nvox,
                            float *pdf, float *odf, float *const peak[3], float *const qa[3],
                            float *odfmax_dev, int flags, void *stream) try {
    FIB_CHECK(plan && dwi && mask && odf && peak && qa, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nvox > 0, FIB_ERR_INVALID, "nvox must be positive");
    FIB_CHECK(nvox <= ((int64_t)1 << 27), FIB_ERR_UNSUPPORTED, "volumes of more than 2^27 voxels are not supported (32-bit lane offsets)");
    FIB_CHECK(plan->nrow0 == 0 || pdf != nullptr, FIB_ERR_INVALID, "DSI plans need a pdf output volume");
    for (int k = 0; k < 3; k++) FIB_CHECK(peak[k] && qa[k], FIB_ERR_INVALID, "NULL peak/qa output volume");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    hipStream_t st = (hipStream_t)stream;
    GemmArgs ga{};
    // launch 1: compact the mask -- voxel list for the GEMM, 64-voxel tile list for the peak finder (counts stay on the device) -- and
    // clear the outputs outside it
    {
        ZeroArgs z{};
        z.out0 = pdf; z.out1 = odf; z.n0 = plan->nrow0; z.n1 = plan->nrows - plan->nrow0; z.nvox = nvox; z.stride = nvox;
        for (int k = 0; k < 3; k++) { z.peak[k] = peak[k]; z.qa[k] = qa[k]; }
        int rcc = launch_mask_compact(plan, mask, nvox, plan->maxenc.p, !(flags & FIB_ODF_PREZEROED) ? &z : nullptr, st);
        if (rcc != FIB_OK) return rcc;
    }
    ga.At = plan->At.p; ga.At3 = (plan->split_bf16 && nvox <= ((int64_t)1 << 26)) ? plan->At3.p : nullptr; ga.S = dwi;
    ga.Aextra = plan->Aextra.p; ga.h2 = plan->h2 ? 1 : 0; ga.h2_inv_sa = 1.0f / plan->h2_sa;
#ifdef FIB_CLOCK_STAMP
    { const char *pi = fib::ab_env("FIBERS_PHASE_ITEM"); ga.phase_item = pi ? atoi(pi) : 2; }
    { const char *pw = fib::ab_env("FIBERS_PHASE_WG"); ga.phase_wg = pw ? atoi(pw) : 8; }        // (odf_dsi2_kernel: 8 = an ODF-tile workgroup, 136 = a pdf-tile one)
#endif
    ga.vec_ok = (nvox % 4 == 0 && ((uintptr_t)odf & 15) == 0 && (pdf == nullptr || ((uintptr_t)pdf & 15) == 0)) ? 1 : 0; ga.vidx = plan->live_vox.p; ga.nlive = plan->live_counts.p; ga.mask = mask; ga.effbits = plan->effbits.p;
    ga.out0 = pdf; ga.out1 = odf; ga.nvox = nvox;
    if (ga.At3 && plan->Gdev.p) {                        // GQI: voxels with a +Inf sample are listed and recomputed (no cap: the list holds every voxel)
        int rci = plan->inf_list.ensure((size_t)nvox);
        if (rci != FIB_OK) return rci;
        ga.fix_count = plan->live_counts.p + 2; ga.fix_list = plan->inf_list.p; ga.fix_cap = (int)std::min<int64_t>(nvox, INT32_MAX);
    }
    // sphere_642 GQI plans: find_peaks! runs on the contraction kernel's accumulators (gemm3_epilogue_fused)
    const bool sep = (flags & FIB_ODF_SEPARATE_PEAKS) != 0;
    const bool fuse = plan->fused && ga.At3 != nullptr && ga.vec_ok && !sep;
    {
        int rcm = plan->mean_hi.ensure((size_t)nvox + 4);
        if (rcm != FIB_OK) return rcm;
    }
    auto setup_fused = [&]() -> int {                   // the arguments of gemm3_epilogue_fused
        int rcf = plan->redo_list.ensure((size_t)nvox);
        if (rcf != FIB_OK) return rcf;
        ga.At3 = plan->At3f.p; ga.Aextra = plan->Aextraf.p;
        for (int k = 0; k < 3; k++) { ga.peak[k] = peak[k]; ga.qa[k] = qa[k]; }
        ga.verts = plan->verts.p; ga.maxenc = plan->maxenc.p; ga.mean_hi = plan->mean_hi.p;
        ga.redo_count = plan->live_counts.p + 3; ga.redo_list = plan->redo_list.p; ga.redo_cap = (int)std::min<int64_t>(nvox, INT32_MAX);
        { const char *pa = fib::ab_env("FIBERS_ODF_ANTI"); ga.anti = pa ? atoi(pa) : 3; }   // bit 0: anti-phase wave halves, bit 1: s_setprio around the MFMA block (default both; 0 = neither)
        return FIB_OK;
    };
    if (fuse) { int rcf = setup_fused(); if (rcf != FIB_OK) return rcf; }
    ga.K = plan->gK; ga.Kpad = plan->Kpad; ga.M = plan->gM; ga.nrow0 = plan->gRow0; ga.ntile_m = plan->ntile_m;
    const bool fold_ok = plan->folded && ga.At3 != nullptr && plan->Kpad <= FKMAX && plan->scale_frame_raw >= 0 &&
                         (int64_t)plan->fold_span_max * nvox * 4 < (int64_t)0xE0000000ll;
    const bool fuse_fold = fold_ok && plan->MB <= FOLD_MB_MAX;
    // folded DSI on sphere_642: one launch of odf_dsi2_kernel does contraction, scale, ODF / pdf rows and find_peaks!
    const bool dsi2 = plan->dsi2 && fold_ok && ga.vec_ok && !sep;
    if (dsi2) {
        int rcf = setup_fused();
        if (rcf != FIB_OK) return rcf;
        ga.At3b = plan->At3b.p;
        ga.ntile_m = 2;
    }
    if (fuse_fold || dsi2) {
        ga.fold = 1;
        ga.rowA = plan->foldA.p; ga.rowB = plan->foldB.p;
    } else if (plan->folded) {
        int rcf = plan->folded_dwi.ensure((size_t)plan->gK * nvox);
        if (rcf != FIB_OK) return rcf;
        fib::ProfScope prof("dsi_fold", st);
        if (nvox % 4 == 0 && ((uintptr_t)dwi & 15) == 0 && ((uintptr_t)mask & 3) == 0)
            hipLaunchKernelGGL(dsi_fold4_kernel, dim3((unsigned)fib::cdiv(nvox / 4, 256)), dim3(256), 0, st, dwi, mask, plan->foldA.p, plan->foldB.p,
                               plan->gK, nvox, plan->folded_dwi.p);
        else
            hipLaunchKernelGGL(dsi_fold_kernel, dim3((unsigned)fib::cdiv(nvox, 256)), dim3(256), 0, st, dwi, mask, plan->foldA.p, plan->foldB.p,
                               plan->gK, nvox, plan->folded_dwi.p);
        ga.S = plan->folded_dwi.p;
        ga.rowA = plan->foldA.p; ga.rowB = plan->foldB.p;
    }
    ga.scale_frame = plan->nrow0 > 0 ? ((fuse_fold || dsi2) ? plan->scale_frame_raw : plan->scale_frame) : -1;
    ga.scale_coef = plan->scale_coef;
    ga.stride = nvox;
    ga.has_ineff = plan->has_ineff ? 1 : 0;
    FIB_CHECK(fib::cdiv(nvox, WG_VOX) * plan->ntile_m < ((int64_t)1 << 31), FIB_ERR_UNSUPPORTED, "volume too large for one launch");

    auto run_gemm = [&](GemmArgs g, hipStream_t s) -> int {
        const unsigned grid = (unsigned)(fib::cdiv(g.nvox, WG_VOX) * plan->ntile_m);
        fib::ProfScope prof("odf_gemm", s);
        if (dsi2) {                                      // persistent grid, an even number of workgroups per XCD (ODF tile | pdf tile)
            int ncu = 256;
            (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, plan->device);
            const int64_t items = fib::cdiv(g.nvox, 8 * 32) * 2;
            unsigned pg = (unsigned)std::min<int64_t>((int64_t)ncu, items);
            pg = (pg + 15) / 16 * 16;
            const int nslot = (int)pg / 8;
            // share of the ODF tile's cost per voxel group (tools/dsi_na_sweep.py, 140^3 x 515: 68 us per ODF item, 59.7 per pdf item of 9
            // blocks; ~4.5 us per block): 17 of 32 workgroups per XCD for the 515-point lattice
            const double ca = 68.0, cb = 19.0 + 4.5 * plan->MBB;
            int na = (int)(nslot * ca / (ca + cb) + 0.5);
            // The pdf-tile workgroups follow the ODF-tile ones through the XCD's voxel groups (see gemm3_body: a bounded wait on the
            // partner's item counter), so the samples are fetched from HBM once and the second reader finds them in the XCD's L2 --
            // [r4] for any split, not only 16 | 16 (FIBERS_DSI_NA=<n> sets the split, FIBERS_DSI_PAIR=0 switches the waiting off:
            // tools/dsi_na_sweep.py).
            const char *ena = fib::ab_env("FIBERS_DSI_NA");
            if (ena) na = atoi(ena);
            g.dsi_na = std::max(1, std::min(nslot - 1, na));
            const char *epair = fib::ab_env("FIBERS_DSI_PAIR");
            if (plan->pair_flags.p && nslot <= 32 && !(epair && epair[0] == '0')) g.pair_flags = plan->pair_flags.p;   // (the counters were cleared by mask_compact_kernel)
            switch (plan->MBB) {
                case 5: if (g.h2) hipLaunchKernelGGL((odf_dsi2_kernel<5, true>), dim3(pg), dim3(512), 0, s, g); else hipLaunchKernelGGL((odf_dsi2_kernel<5, false>), dim3(pg), dim3(512), 0, s, g); break;
                case 7: if (g.h2) hipLaunchKernelGGL((odf_dsi2_kernel<7, true>), dim3(pg), dim3(512), 0, s, g); else hipLaunchKernelGGL((odf_dsi2_kernel<7, false>), dim3(pg), dim3(512), 0, s, g); break;
                default: if (g.h2) hipLaunchKernelGGL((odf_dsi2_kernel<9, true>), dim3(pg), dim3(512), 0, s, g); else hipLaunchKernelGGL((odf_dsi2_kernel<9, false>), dim3(pg), dim3(512), 0, s, g); break;
            }
            FIB_HIP(hipGetLastError());
            return FIB_OK;
        }
#define FIB_GEMM_CASE(MBV, NXV) if (plan->MB == MBV && plan->NX == NXV) { launch_gemm<MBV, NXV>(g, grid, s); launched = true; }
        bool launched = false;
        FIB_GEMM_CASE(5, 0) FIB_GEMM_CASE(6, 0) FIB_GEMM_CASE(7, 0) FIB_GEMM_CASE(8, 0) FIB_GEMM_CASE(9, 0) FIB_GEMM_CASE(10, 0) FIB_GEMM_CASE(11, 0)
        FIB_GEMM_CASE(5, 1) FIB_GEMM_CASE(6, 1) FIB_GEMM_CASE(7, 1) FIB_GEMM_CASE(8, 1) FIB_GEMM_CASE(9, 1) FIB_GEMM_CASE(10, 1)
#undef FIB_GEMM_CASE
        if (!launched) return fib::fail(FIB_ERR_INVALID, "internal: no GEMM variant for MB=%d NX=%d", plan->MB, plan->NX);
        FIB_HIP(hipGetLastError());
        return FIB_OK;
    };

    int rc = run_gemm(ga, st);                           // launch 2
    if (rc != FIB_OK) return rc;
    const bool fused_scan = fuse || dsi2;
    if (!fused_scan) {                                  // separate peak finder on the stored ODF (other tessellations, unaligned pieces, FIB_ODF_SEPARATE_PEAKS)
        if (ga.fix_list) {
            InfFixArgs fx{plan->Gdev.p, dwi, odf, ga.fix_count, ga.fix_list, ga.fix_cap, plan->gM, plan->gK, nvox};
            hipLaunchKernelGGL(odf_inf_fix_kernel, dim3(64), dim3(256), 0, st, fx);
        }
        rc = launch_peaks(plan, odf, nvox, nvox, peak, qa, nullptr, nullptr, true, st, false, plan->live_tiles.p, plan->live_counts.p + 1);
        if (rc != FIB_OK) return rc;
    }
    float *om = odfmax_dev ? odfmax_dev : plan->odfmax.p;
    {                                                   // launch 3: redo list (+ column repair), exact odfmax, its two floats
        fib::ProfScope prof("odf_post", st);
        PostArgs pa{};
        pa.redo = RedoArgs{odf, nvox, fused_scan ? ga.redo_count : nullptr, ga.redo_list, ga.redo_cap, plan->verts.p, {peak[0], peak[1], peak[2]}, {qa[0], qa[1], qa[2]}, plan->maxenc.p};
        pa.G = (fused_scan && ga.fix_list) ? plan->Gdev.p : nullptr; pa.S = dwi; pa.out = odf; pa.M = plan->gM; pa.K = plan->gK;
        pa.refine = RefineArgs{odf, nvox, nvox, plan->nvert, plan->live_vox.p, plan->live_counts.p, plan->mean_hi.p, plan->maxenc.p};
        pa.arrive = plan->tickets.p + 1; pa.odfmax = om; pa.raw = (flags & FIB_ODF_RAW_ODFMAX) ? 1 : 0;
        pa.selctr = plan->selctr.p; pa.sellist = plan->sellist.p; pa.selcap = POST_SELCAP; pa.epoch = plan->compact_epoch;
        if (flags & FIB_ODF_NORMALIZE) {                // [r6] qa ./= odfmax in the same launch (gqi.jl:166-168)
            for (int k = 0; k < 3; k++) pa.nq[k] = qa[k];
            pa.nq_n = nvox; pa.done = reinterpret_cast<unsigned long long *>(plan->tickets.p + 2); pa.epoch = plan->compact_epoch;
        }
        { const char *e = fib::ab_env("FIBERS_POST_SKIP"); pa.dbg = e ? atoi(e) : 0; }
        hipLaunchKernelGGL(odf_post_kernel, dim3(384), dim3(256), 0, st, pa);
        FIB_HIP(hipGetLastError());
    }
    return FIB_OK;
} FIB_API_CATCH

// ---- plain use of the contraction kernels: O[M x n] = A[M x K] * max(S[K x n], 0) (row N4, RUMBA-SD) -----------------
int fib::matrix_plan_create(int device, const float *A, int nrows, int ncols, fib_odf_plan **plan) {
    FIB_CHECK(A && plan && nrows > 0 && ncols > 0, FIB_ERR_INVALID, "invalid matrix plan arguments");
    *plan = nullptr;
    fib::DeviceGuard guard;
    int rc = fib::use_device(device);
    if (rc != FIB_OK) return rc;
    fib_odf_plan *p = new (std::nothrow) fib_odf_plan();
    FIB_CHECK(p != nullptr, FIB_ERR_NOMEM, "out of host memory");
    p->device = device; p->nvol = ncols; p->nvert = 2; p->nrows = nrows; p->nrow0 = 0;
    p->A.assign(A, A + (size_t)nrows * ncols);
    std::vector<float> eff((size_t)ncols, 1.0f);
    rc = finish_plan(p, nullptr, 4, nullptr, 0, eff);
    if (rc != FIB_OK) { delete p; return rc; }
    *plan = p;
    return FIB_OK;
}

int fib::matrix_plan_run(const fib_odf_plan *plan, const float *S, const uint8_t *ones, int64_t n, float *out, bool recompact, void *stream) {
    FIB_CHECK(plan && S && ones && out && n > 0, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(n <= ((int64_t)1 << 26), FIB_ERR_UNSUPPORTED, "more than 2^26 columns");
    hipStream_t st = (hipStream_t)stream;
    if (recompact) {                                    // the column list of an all-ones mask: kept in the plan between calls
        int rcc = launch_mask_compact(plan, ones, n, nullptr, nullptr, st);
        if (rcc != FIB_OK) return rcc;
    }
    GemmArgs ga{};
    ga.At = plan->At.p; ga.At3 = plan->split_bf16 ? plan->At3.p : nullptr; ga.S = S;
    ga.Aextra = plan->Aextra.p; ga.h2 = plan->h2 ? 1 : 0; ga.h2_inv_sa = 1.0f / plan->h2_sa;
    ga.vec_ok = (n % 4 == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    ga.vidx = plan->live_vox.p; ga.nlive = plan->live_counts.p; ga.mask = ones; ga.effbits = plan->effbits.p;
    ga.out0 = nullptr; ga.out1 = out; ga.nvox = n;
    ga.K = plan->gK; ga.Kpad = plan->Kpad; ga.M = plan->gM; ga.nrow0 = 0; ga.ntile_m = plan->ntile_m;
    ga.scale_frame = -1; ga.scale_coef = 0.0f; ga.stride = n; ga.has_ineff = 0;
    const unsigned grid = (unsigned)(fib::cdiv(n, WG_VOX) * plan->ntile_m);
    fib::ProfScope prof("matrix_gemm", st);
#define FIB_GEMM_CASE(MBV, NXV) if (plan->MB == MBV && plan->NX == NXV) { launch_gemm<MBV, NXV>(ga, grid, st); launched = true; }
    bool launched = false;
    FIB_GEMM_CASE(5, 0) FIB_GEMM_CASE(6, 0) FIB_GEMM_CASE(7, 0) FIB_GEMM_CASE(8, 0) FIB_GEMM_CASE(9, 0) FIB_GEMM_CASE(10, 0) FIB_GEMM_CASE(11, 0)
    FIB_GEMM_CASE(5, 1) FIB_GEMM_CASE(6, 1) FIB_GEMM_CASE(7, 1) FIB_GEMM_CASE(8, 1) FIB_GEMM_CASE(9, 1) FIB_GEMM_CASE(10, 1)
#undef FIB_GEMM_CASE
    if (!launched) return fib::fail(FIB_ERR_INVALID, "internal: no GEMM variant for MB=%d NX=%d", plan->MB, plan->NX);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
}

extern "C" int fibd_qa_normalize(float *const qa[3], int64_t nvox, float odfmax, void *stream) try {
    FIB_CHECK(qa && qa[0] && qa[1] && qa[2] && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    hipLaunchKernelGGL(qa_normalize_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, qa[0], qa[1], qa[2], nvox,
                       (float *)nullptr, odfmax, 0);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fibd_qa_normalize_dev(float *const qa[3], int64_t nvox, const float *odfmax_dev, void *stream) try {
    FIB_CHECK(qa && qa[0] && qa[1] && qa[2] && odfmax_dev && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    hipLaunchKernelGGL(qa_normalize_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, qa[0], qa[1], qa[2], nvox, const_cast<float *>(odfmax_dev), 0.0f, 0);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fibd_qa_normalize_pair(float *const qa[3], int64_t nvox, float *odfmax_pair_dev, void *stream) try {
    FIB_CHECK(qa && qa[0] && qa[1] && qa[2] && odfmax_pair_dev && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    hipLaunchKernelGGL(qa_normalize_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, qa[0], qa[1], qa[2], nvox, odfmax_pair_dev, 0.0f, 1);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fibd_find_peaks(const fib_odf_plan *plan, const float *odf, int64_t nvox,
                               int32_t *isort_top, int32_t *nvalid, void *stream) try {
    FIB_CHECK(plan && odf && isort_top && nvalid && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    return launch_peaks(plan, odf, nvox, nvox, nullptr, nullptr, isort_top, nvalid, false, (hipStream_t)stream);
} FIB_API_CATCH

#ifdef FIB_CLOCK_STAMP
// diagnostic build: {cycles, ticks, kernel id, items} of the first `cap` workgroups of the last contraction launch
extern "C" int fib_debug_clock_stamps(unsigned long long *out, int cap) try {
    FIB_CHECK(out && cap > 0 && cap <= 2048, FIB_ERR_INVALID, "invalid stamp buffer");
    FIB_HIP(hipDeviceSynchronize());
    FIB_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(fib_clock_stamps), (size_t)cap * 4 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost));
    return FIB_OK;
} FIB_API_CATCH
extern "C" int fib_debug_phase_stamps(unsigned long long *out) try {      // [2][256]: (s_memtime << 8) | mark id
    FIB_CHECK(out, FIB_ERR_INVALID, "invalid stamp buffer");
    FIB_HIP(hipDeviceSynchronize());
    FIB_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(fib_phase_stamps), sizeof(unsigned long long) * 512, 0, hipMemcpyDeviceToHost));
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fib_debug_clock_clear(void) try {
    static unsigned long long zeros[2048][4];
    FIB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(fib_clock_stamps), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice));
    return FIB_OK;
} FIB_API_CATCH
#endif

extern "C" int fibd_find_peaks_work(const fib_odf_plan *plan, const float *odf, int64_t nvox,
                                    float *odf_peak, int32_t *isort, int32_t *nvalid, void *stream) try {
    FIB_CHECK(plan && odf && odf_peak && isort && nvalid && nvox > 0, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(plan->nbr64.p != nullptr && plan->maxdeg > 0, FIB_ERR_INVALID, "the plan has no tessellation");
    fib::DeviceGuard guard;
    FIB_HIP(hipSetDevice(plan->device));
    const size_t smem = (size_t)plan->nvert * sizeof(unsigned long long) + (size_t)(plan->nvert + 1) * sizeof(float);
    FIB_CHECK(smem <= 64 * 1024, FIB_ERR_UNSUPPORTED, "ODF with %d vertices does not fit the peak finder's LDS", plan->nvert);
    const unsigned grid = (unsigned)std::min<int64_t>(nvox, 8192);
    hipLaunchKernelGGL(odf_peaks_work_kernel, dim3(grid), dim3(256), smem, (hipStream_t)stream, odf, nvox, nvox, plan->nvert, plan->deg_pad,
                       plan->nbr64.p, odf_peak, isort, nvalid);
    FIB_HIP(hipGetLastError());
    return FIB_OK;
} FIB_API_CATCH
