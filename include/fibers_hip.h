/*
 * fibers_hip.h — C ABI of libfibers_hip.so, the MI355X (gfx950) back end for the
 * Fibers.jl per-voxel reconstruction + streamline hot path.
 *
 * The reference (lincbrain/Fibers.jl) has no FFI: its boundary is the exported Julia
 * function surface plus the MRI / Tract field layouts.  Each entry point below names the
 * reference function (file:line under the reference's src/) whose body it replaces; the
 * Julia `ccall` stubs a maintainer would add are in INTEGRATION.md.
 *
 * Conventions
 *   - All volumes are float32, Julia column-major [nx,ny,nz,nframes] exactly as
 *     `MRI.vol` (mri.jl:81): x fastest, frame slowest ("planar").  nvox = nx*ny*nz.
 *   - bvec is [nvol x 3] column-major like `MRI.bvec` (mri.jl:129).
 *   - Streamline coordinates are float32, 1-based voxel coordinates exactly as
 *     `pos_now` in stream.jl:660.
 *   - Every function returns FIB_OK (0) or a negative fib_status; the message is
 *     available from fib_last_error() (thread-local).  No C++ exception and no abort()
 *     crosses this boundary.  There is NO CPU fallback: without a usable HIP device
 *     every compute entry point fails with FIB_ERR_NO_DEVICE.
 *   - fib_*  : host-buffer ("drop-in") entry points; blocking; buffers are caller-owned
 *              host memory (Julia arrays under GC.@preserve); nothing is retained.
 *   - fibd_* : device-resident entry points (pointers are HIP device pointers, `stream`
 *              is a hipStream_t passed as void*, may be NULL); asynchronous on `stream`
 *              unless stated.  Used by the multi-GPU host layer and by bench.py.
 */
#ifndef FIBERS_HIP_H
#define FIBERS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    FIB_OK = 0,
    FIB_ERR_INVALID = -1,       /* bad argument (NULL pointer, non-positive size, ...) */
    FIB_ERR_NO_DEVICE = -2,     /* no HIP device / device index out of range */
    FIB_ERR_HIP = -3,           /* a HIP runtime call failed; see fib_last_error() */
    FIB_ERR_MISSING_BVAL = -4,  /* "Missing b-value table from input DWI structure"  (dti.jl:167,224 gqi.jl:112 dsi.jl:174) */
    FIB_ERR_MISSING_BVEC = -5,  /* "Missing gradient table from input DWI structure" (dti.jl:228 gqi.jl:116 dsi.jl:178) */
    FIB_ERR_DIM_MISMATCH = -6,  /* "Dimension mismatch between seed mask ... and brain mask ..." (stream.jl:746-749) */
    FIB_ERR_UNSUPPORTED = -7,   /* e.g. q-space grid other than 16^3, ODF vertex degree too large */
    FIB_ERR_NOMEM = -8,
    FIB_ERR_CAPACITY = -9       /* caller-provided output buffers too small (fibd_stream_run): the counts say what is needed */
} fib_status;

/* element type of a mask / seed volume handed over by the host (any numeric array in Julia) */
typedef enum {
    FIB_U8 = 0, FIB_I8 = 1, FIB_I16 = 2, FIB_U16 = 3, FIB_I32 = 4, FIB_U32 = 5,
    FIB_F32 = 6, FIB_F64 = 7, FIB_I64 = 8, FIB_BOOL = 9
} fib_dtype;

const char *fib_last_error(void);
const char *fib_version(void);
int fib_device_count(void);                 /* number of visible HIP devices (0 if none) */

/* ------------------------------------------------------------------------------------ */
/* Measurement hooks (bench.py): per-kernel HIP-event timing on the launch stream         */
/* ------------------------------------------------------------------------------------ */
/* When enabled, every hot-path kernel launch is bracketed by a hipEvent pair recorded on the
 * stream it is launched on.  fib_profile_get synchronises the pending events of `kernel`
 * ("dti_fit", "odf_gemm", "odf_peaks", "qa_normalize", "stream_trace", "stream_pack", ...) and
 * returns the accumulated device time and launch count since the last fib_profile_reset.
 * fib_profile_filter("odf_gemm,dti_fit"): only the named kernels are bracketed (NULL or "": all of them) -- an event pair is two packets in
 * the queue, and a measurement of one kernel inside a timed region should not pay for the others' brackets. */
int fib_profile_enable(int on);
int fib_profile_filter(const char *names);
int fib_profile_reset(void);
int fib_profile_get(const char *kernel, double *total_ms, int64_t *count);

/* ------------------------------------------------------------------------------------ */
/* Plans = the reference's pre-computed work structs, resident on one device             */
/* ------------------------------------------------------------------------------------ */
/* A plan also owns the scratch of the calls made on it (voxel lists, counters, work arrays: the reference's per-thread
 * work structs): use one plan per concurrent call / stream; plans themselves are independent of each other. */
typedef struct fib_dti_plan fib_dti_plan;   /* DTIwork / ADCwork  (dti.jl:39-84, 101-155) */
typedef struct fib_odf_plan fib_odf_plan;   /* GQIwork (gqi.jl:32-82) or DSIwork (dsi.jl:41-143) */

/* DTIwork(bval, bvec) (dti.jl:110): design matrix A[nvol x 7] and pA = pinv(A); with
 * bvec == NULL builds ADCwork(bval) (dti.jl:48): A = [-b, 1].  Host tables in, device plan out. */
int fib_dti_plan_create(int device, const float *bval, const float *bvec, int nvol, fib_dti_plan **plan);
void fib_dti_plan_destroy(fib_dti_plan *plan);
/* copies of the host-side tables for inspection/tests: A [nvol x np] and pA [np x nvol], column-major */
int fib_dti_plan_tables(const fib_dti_plan *plan, float *A, float *pA, int *np);

/* GQIwork(bval, bvec, odf_dirs, sigma) (gqi.jl:42): A = sinc.(V[nvert+1:end,:] * bq') and the
 * folded face table.  verts [nverts x 3] column-major, faces [nfaces x 3] column-major, 1-based. */
int fib_gqi_plan_create(int device, const float *bval, const float *bvec, int nvol,
                        const float *verts, int nverts, const int32_t *faces, int nfaces,
                        float sigma, fib_odf_plan **plan);
/* DSIwork(bval, bvec, odf_dirs, hann_width) (dsi.jl:59).  The per-voxel chain
 * scatter -> Hanning -> centred 16^3 FFT -> Re -> radial trilinear integration (dsi.jl:204-242)
 * is linear in the clamped signal up to the 1/sum(p) scalar, so the plan holds it as two dense
 * maps: pdf = C[nvol x nvol] s / sum(p), odf = M[nvert x nvol] s / sum(p). */
int fib_dsi_plan_create(int device, const float *bval, const float *bvec, int nvol,
                        const float *verts, int nverts, const int32_t *faces, int nfaces,
                        int hann_width, fib_odf_plan **plan);
/* The same with the operand format of the contraction A*s chosen per plan.  The reference runs an f32 sgemv (gqi.jl:144);
 * the kernels run it on the matrix cores in one of three forms, all f32 in / f32 accumulate / f32 out:
 *   FIB_ODF_FORMAT_FP16X2  two fp16 pieces per f32 operand (23 of 24 significant bits, three piece products; the default)
 *   FIB_ODF_FORMAT_BF16X3  three bf16 pieces per operand, six piece products: every f32 product exact
 *   FIB_ODF_FORMAT_F32     v_mfma_f32_32x32x2_f32: a k-ordered f32 fma chain
 *   FIB_ODF_FORMAT_DEFAULT what the environment selects (FIBERS_ODF_FORMAT = fp16x2 | bf16x3 | f32), else FP16X2.
 * A plan may fall back to a wider form when its matrix does not fit a narrower one (non-finite entries, < 2 stages):
 * fib_odf_plan_format reports what the plan's kernels actually run; fib_odf_default_format what DEFAULT resolves to now. */
#define FIB_ODF_FORMAT_DEFAULT 0
#define FIB_ODF_FORMAT_FP16X2 1
#define FIB_ODF_FORMAT_BF16X3 2
#define FIB_ODF_FORMAT_F32 3
int fib_gqi_plan_create_fmt(int device, const float *bval, const float *bvec, int nvol,
                            const float *verts, int nverts, const int32_t *faces, int nfaces,
                            float sigma, int format, fib_odf_plan **plan);
int fib_dsi_plan_create_fmt(int device, const float *bval, const float *bvec, int nvol,
                            const float *verts, int nverts, const int32_t *faces, int nfaces,
                            int hann_width, int format, fib_odf_plan **plan);
int fib_odf_plan_format(const fib_odf_plan *plan);   /* FIB_ODF_FORMAT_* (> 0) or a negative error code */
/* Diagnostic: the unit of the voxel list the plan's next fibd_odf_rec call will use -- 1: aligned groups of 32 voxels (a wave's
 * 128-byte row segments are whole cache lines whatever the mask's runs look like; the default), 0: aligned groups of 4 (chosen by
 * the previous call when groups of 32 would list more than 1.5 x the voxels: sparse masks).  Results do not depend on it.
 * Waits for `stream`. */
int fib_odf_plan_list_unit(const fib_odf_plan *plan, void *stream);
int fib_odf_default_format(void);
void fib_odf_plan_destroy(fib_odf_plan *plan);
/* host copy of the reconstruction matrix [nrows x nvol] column-major (GQI: nrows = nvert;
 * DSI: nrows = nvol + nvert, pdf rows first); pass NULL to query sizes only. */
int fib_odf_plan_matrix(const fib_odf_plan *plan, float *A, int *nrows, int *nvol, int *nvert);

/* ------------------------------------------------------------------------------------ */
/* Device-resident hot path                                                              */
/* ------------------------------------------------------------------------------------ */

/* 10 output volumes of dti_fit_ls (dti.jl:247-256): scalars [nvox], eigvecs [nvox*3] planar */
typedef struct {
    float *s0, *eigval1, *eigval2, *eigval3;
    float *eigvec1, *eigvec2, *eigvec3;
    float *rd, *md, *fa;
} fib_dti_out;

/* dti_fit_ls(dwi::MRI, mask::MRI) volume loop (dti.jl:258-275) + per-voxel fit (dti.jl:286-316)
 * + dti_maps (dti.jl:325-335).  mask: uint8 [nvox], non-zero = fit (dti.jl:261).  All outputs are
 * fully written (zeros where the reference leaves its zero-initialised volumes untouched). */
int fibd_dti_fit(const fib_dti_plan *plan, const float *dwi, const uint8_t *mask, int64_t nvox,
                 const fib_dti_out *out, void *stream);
/* adc_fit (dti.jl:164-213) */
int fibd_adc_fit(const fib_dti_plan *plan, const float *dwi, const uint8_t *mask, int64_t nvox,
                 float *adc, float *s0, void *stream);
/* st_eigen (structens.jl:13-37): eigen(Symmetric(S, :L)) per voxel with the diffusion tensor's 3x3 solver.
 * S = {Sxx, Sxy, Sxz, Syy, Syz, Szz}, each [nvox]; eigval [nvox*3] ascending (eigval[ix,iy,iz,k]);
 * eigvec [nvox*9], component i of eigenvector j at (i + 3*j)*nvox + vox (eigvec[ix,iy,iz,i,j]). */
int fibd_st_eigen(const float *const S[6], int64_t nvox, float *eigvec, float *eigval, void *stream);
/* number of voxels the last fibd_dti_fit/fibd_adc_fit call on this plan sent through the
 * per-voxel pinv branch (dti.jl:297-298, 206-207); synchronises `stream`. */
int fibd_dti_last_partial_count(const fib_dti_plan *plan, void *stream, int64_t *count);

/* gqi_rec / dsi_rec volume loop + find_peaks! + peak/qa extraction (gqi.jl:132-162,
 * dsi.jl:197-261).  odf [nvox*nvert] planar; pdf [nvox*nvol] (DSI plans only, else NULL);
 * peak[k] [nvox*3] planar, qa[k] [nvox].  `flags` is a bit set:
 *   FIB_ODF_NORMALIZE  the global step qa ./= maximum(mean(odf, dims=4)) (gqi.jl:164-168,
 *                      dsi.jl:263-267) is applied in-stream; without it qa is left un-normalised and
 *                      *odfmax_dev (device float[2]: {max, nan-flag}) holds this call's local maximum so
 *                      that ranks can all-reduce it and call fibd_qa_normalize;
 *   FIB_ODF_PREZEROED  the caller guarantees that every output is already 0 at the voxels outside
 *                      `mask` (e.g. buffers from a previous call with the same mask); otherwise they
 *                      are zero-filled here, like the reference's freshly allocated volumes;
 *   FIB_ODF_SEPARATE_PEAKS  run find_peaks! as its own kernel on the stored ODF instead of on the contraction
 *                      kernel's accumulators (the fused form needs 16-byte aligned rows, nvox % 4 == 0, and
 *                      computes two of the 321 rows of sphere_642 with a different rounding): a caller that
 *                      cuts one volume into pieces sets it for ALL pieces when any piece is unaligned, so
 *                      that the result does not depend on the cut (the host-buffer tier does).
 *   FIB_ODF_RAW_ODFMAX the form of *odfmax_dev a MAX all-reduce can take as it is: {maximum of the voxel means that
 *                      are not NaN (-Inf if there is none), nan-flag}.  Without it the first element is NaN when the
 *                      flag is set (what maximum() returns, gqi.jl:164).  fibd_qa_normalize_pair consumes the raw form.
 * Only voxels inside the mask are computed: the mask is compacted on the device into a voxel list
 * (reconstruction) and a list of 64-voxel tiles (peak finder), so cost scales with the mask. */
#define FIB_ODF_NORMALIZE 1
#define FIB_ODF_PREZEROED 2
#define FIB_ODF_SEPARATE_PEAKS 4
#define FIB_ODF_RAW_ODFMAX 8
int fibd_odf_rec(const fib_odf_plan *plan, const float *dwi, const uint8_t *mask, int64_t nvox,
                 float *pdf, float *odf, float *const peak[3], float *const qa[3],
                 float *odfmax_dev, int flags, void *stream);
/* qa[k] ./= odfmax for all voxels (gqi.jl:166-168) */
int fibd_qa_normalize(float *const qa[3], int64_t nvox, float odfmax, void *stream);
/* same with the divisor read from device memory (the all-reduced odfmax stays on the device: no host round trip) */
int fibd_qa_normalize_dev(float *const qa[3], int64_t nvox, const float *odfmax_dev, void *stream);
/* .. from the raw pair {maximum of the non-NaN means, nan-flag} (FIB_ODF_RAW_ODFMAX, MAX-all-reduced over the ranks): the
 * divisor is NaN when the flag is set, and odfmax_pair_dev[0] is rewritten to that divisor (what maximum() returns) */
int fibd_qa_normalize_pair(float *const qa[3], int64_t nvox, float *odfmax_pair_dev, void *stream);

/* find_peaks!(W) (gqi.jl:180-201) on a planar ODF volume [nvox*nvert]: for every voxel the
 * indices (0-based, first-half vertex rows) of the first 3 entries of `isort` and `nvalid`.
 * isort_top [3*nvox] planar int32 (-1 where fewer than k+1 vertices exist). */
int fibd_find_peaks(const fib_odf_plan *plan, const float *odf, int64_t nvox,
                    int32_t *isort_top, int32_t *nvalid, void *stream);
/* .. with all of the work struct's outputs: odf_peak [nvert*nvox] planar (gqi.jl:184-196), isort [nvert*nvox] planar int32
 * (the complete permutation, 0-based, gqi.jl:198), nvalid [nvox] (gqi.jl:200) */
int fibd_find_peaks_work(const fib_odf_plan *plan, const float *odf, int64_t nvox,
                         float *odf_peak, int32_t *isort, int32_t *nvalid, void *stream);

/* ------------------------------------------------------------------------------------ */
/* RUMBA-SD (rusd.jl), SURVEY.md row N4                                                   */
/* ------------------------------------------------------------------------------------ */
typedef struct fib_rumba_plan fib_rumba_plan;   /* kernel of the multi-tensor model + the two contraction plans */

/* rumba_rec's set-up (rusd.jl:449, 466-521, 529-531): ib0 = (bval .== minimum(bval)); the kernel K [ndir x (nvert+2)]
 * (ndir = 1 + number of non-low-b frames; one prolate tensor per half-sphere vertex, isotropic CSF and GM columns),
 * the angular peak neighbourhoods (12.5 deg for sphere_642/724, 16 deg for sphere_362) and the uniform initial fODF.
 * verts [nverts x 3] column-major as for the GQI plan.  Reference defaults: lam_para 1.7e-3, lam_perp 0.2e-3,
 * lam_csf 3.0e-3, lam_gm 0.8e-4. */
int fib_rumba_plan_create(int device, const float *bval, const float *bvec, int nvol, const float *verts, int nverts,
                          float lam_para, float lam_perp, float lam_csf, float lam_gm, fib_rumba_plan **plan);
void fib_rumba_plan_destroy(fib_rumba_plan *plan);
/* host copy of the kernel, column-major [ndir x ncomp]; K == NULL queries the sizes only */
int fib_rumba_plan_kernel(const fib_rumba_plan *plan, float *K, int *ndir, int *ncomp);

/* outputs of rumba_rec (the RUMBASD struct, rusd.jl:11-20): fodf planar [nvert][nvox]; fgm, fcsf, gfa, var [nvox];
 * peak[k] planar [3][nvox], k = 0..4 */
typedef struct {
    float *fodf, *fgm, *fcsf, *gfa, *var;
    float *peak[5];
} fib_rumba_out;

/* rumba_rec(dwi, mask, odf_dirs, niter, ...) (rusd.jl:419-636) on device-resident volumes: dwi planar [nvol][nvox],
 * mask uint8 [nvox] (already `> 0`-tested).  niter: Richardson-Lucy iterations (reference default 600); ncoils /
 * sos_grappa: coil_combine == "SoS-GRAPPA" uses n_order = ncoils, "SMF-SENSE" n_order = 1; ipat_factor >= 1; use_tv:
 * total-variation prior.  snr_mean / snr_std: host scalars (may be NULL).  Blocking.  The plan keeps the call's work arrays
 * (8 x [ndir or ncomp][masked voxels] floats) for the next call: one call at a time per plan; destroy the plan to release them. */
int fibd_rumba_rec(const fib_rumba_plan *plan, const float *dwi, const uint8_t *mask, int nx, int ny, int nz,
                   int niter, int ncoils, int sos_grappa, int ipat_factor, int use_tv,
                   const fib_rumba_out *out, float *snr_mean, float *snr_std, void *stream);

/* ------------------------------------------------------------------------------------ */
/* Streamlines                                                                            */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int32_t nx, ny, nz, nvec;
    int32_t len_min;        /* default 3 */
    int32_t len_max;        /* default max(nx,ny,nz)      (stream.jl:74) */
    float cosang_thresh;    /* cosd(ang_thresh), default cosd(45) (stream.jl:193) */
    float step_size;        /* default .5 */
    float smooth_coeff;     /* default .2 */
    /* microscopy regime (stream.jl:83: minimum(volres) <= 0.05 mm; 252-287, 547-619): search_dist > 0 selects it.
     * Each step then moves to the voxel, within search_dist voxels of the tentative position and a cone of
     * search_ang around the current direction, whose first orientation vector is best aligned with it.
     * Reference defaults in that regime: search_dist 15, search_ang 10, nsub 0, ang_thresh 20, step 1, smooth 0. */
    int32_t search_dist;    /* 0: macro-scale tracking */
    float search_cosang;    /* cosd(search_ang) */
    struct fib_stream_ws *ws;   /* optional scratch arena (fibd_stream_ws_create); NULL: the job allocates and frees its own */
    /* 0: the reference's nearest-voxel lookup (stream.jl:514).  1: NOT in the reference — the direction followed is the trilinear
     * blend of the 8 voxels around the tentative position: w = normalise(sum_c t_c s_c u_c), t_c the trilinear weight of corner c
     * (corners outside the volume dropped), u_c the corner's vector picked by the angle rule of stream.jl:340-374 against the
     * current direction (corners without a vector dropped), s_c the sign of its cosine; a zero blend ends the line.  Bounds,
     * mask and the nearest voxel's pick still decide termination exactly as in the reference; macro scale, angle picking only. */
    int32_t interp;
    /* microscopy regime with 2-D orientation-angle inputs (one frame per volume, stream.jl:147-172): StreamWork sets the search
     * distance of the through-plane axis -- the one with the largest voxel size -- to 0 (stream.jl:153-155).  0: none (a cubic
     * search area); 1, 2, 3: the x, y, z axis is searched over one voxel only.  (The angles themselves are expanded to 3-D
     * vectors by the host-language wrapper, cos / sin or cosd / sind like the constructor does: ovec stays [nvox*3].) */
    int32_t search_flat_axis;
} fib_stream_params;

typedef struct fib_stream_job fib_stream_job;
/* Grow-only scratch arena of the tracer, owned by the caller (the reference's per-thread StreamWork scratch, stream.jl:43-60):
 * worst-case point rows for every line (3.4 GB per million lines at len_max 140) are expensive to allocate per call.  One job
 * at a time takes the arena; a job that finds it busy (or on another device) allocates its own scratch.  Releasing is
 * stream-ordered: the next job waits on its own stream for the previous job's last launch. */
typedef struct fib_stream_ws fib_stream_ws;
int fibd_stream_ws_create(int device, fib_stream_ws **ws);
void fibd_stream_ws_destroy(fib_stream_ws *ws);

/* StreamWork mask + vector repack (stream.jl:95-145): mask_out = (mask > 0 | any nonzero vector)
 * & (fa >= fa_thresh); field[vox][k] = ovec[k][vox,:] * (mask_out & f[k] >= f_thresh), stored as
 * float4 (xyz0) [nvox*nvec].  ovec[k] planar [nvox*3]; f[k] [nvox] or f == NULL; fa, mask may be NULL. */
int fibd_stream_field(int32_t nvec, int64_t nvox, const float *const *ovec, const float *const *f,
                      float f_thresh, const float *fa, float fa_thresh, const uint8_t *mask,
                      float *field4, uint8_t *mask_out, void *stream);

/* stream_new_line for every (seed, sub-voxel offset) pair (stream.jl:761-781 + 625-690, angle
 * picking, non-LCM, macro scale).  seeds: int64 [nseed] 0-based column-major linear voxel indices
 * in the reference's findall order; sublist [nsub*3] (xyz per offset; caller-generated, stream.jl:176-181).
 * Traces into library-owned scratch, applies len_min (stream.jl:769) and computes output offsets.
 * Synchronises `stream`; returns the number of kept lines and their total point count.
 * field4, seeds and sublist must stay valid and unchanged until the job has been packed.  Orientation fields of 2^28 vectors
 * (4 GiB) or more take a form of the tracer with 64-bit gather offsets, chosen at launch. */
int fibd_stream_trace(const fib_stream_params *prm, const float *field4, const int64_t *seeds, int64_t nseed,
                      const float *sublist, int32_t nsub, void *stream,
                      fib_stream_job **job, int64_t *nlines, int64_t *npoints);
/* packs the kept lines, in (seed, sub) order == reference order, into caller device buffers:
 * npts [nlines] int32, seed_index [nlines] int64 (= seed*nsub+sub), xyz [3*npoints] (x,y,z per point,
 * line after line, each line ordered [fwd_N..fwd_1, bwd_1..bwd_M] as stream.jl:652 builds it). */
int fibd_stream_pack(fib_stream_job *job, int32_t *npts, int64_t *seed_index, float *xyz, void *stream);

/* trace + pack in ONE call into caller-provided device buffers (npts [lines_cap] int32, seed_index [lines_cap] int64, xyz
 * [3*points_cap] float): the same lines, order and layout as fibd_stream_trace + fibd_stream_pack without the second call and
 * without any allocation on the caller's side of the boundary between them.  From 2^21 lines on (nearest-voxel tracking, 1 or 3
 * vectors per voxel, lines that fit a 16-line LDS tile) it is ONE kernel: the workgroup that traced 512 lines packs them behind a
 * decoupled look-back over the workgroups' totals.  *nlines / *npoints receive the totals; when they exceed the capacities the call
 * returns FIB_ERR_CAPACITY (lines that did not fit are missing from the buffers; call again with larger ones).  Macro-scale angle
 * picking only (prm->search_dist == 0, no LCMs); synchronises `stream`. */
int fibd_stream_run(const fib_stream_params *prm, const float *field4, const int64_t *seeds, int64_t nseed,
                    const float *sublist, int32_t nsub, int32_t *npts, int64_t *seed_index, int64_t lines_cap,
                    float *xyz, int64_t points_cap, int64_t *nlines, int64_t *npoints, void *stream);

/* fibd_stream_run without the host round trip at its end: returns once the work is enqueued on `stream`, which also writes
 * counts_dev[0] = lines, counts_dev[1] = points (DEVICE memory, 2 x int64; the totals the run NEEDED: larger than the capacities when
 * lines were dropped for lack of room -- compare after synchronising, there is no FIB_ERR_CAPACITY here).  The buffers and the
 * workspace of prm->ws stay in use until `stream` has passed the call.  For callers that keep a stream of volumes in flight: the
 * synchronising form idles the GPU for the download of its two counts between consecutive calls. */
int fibd_stream_run_enqueue(const fib_stream_params *prm, const float *field4, const int64_t *seeds, int64_t nseed,
                            const float *sublist, int32_t nsub, int32_t *npts, int64_t *seed_index, int64_t lines_cap,
                            float *xyz, int64_t points_cap, int64_t *counts_dev, void *stream);

/* LCM-guided tracking (stream(...; lcms, lcm_thresh), stream.jl:200-236, 380-495, 526-538): when a line enters a new
 * voxel the exit edge is drawn from the voxel's local connection matrix restricted to the entry edge, and the
 * orientation vector best aligned with a jump towards that edge is followed; the angle threshold is not applied
 * (stream.jl:668).  lcms: planar [10][nvox] = MRI.vol[nx,ny,nz,10]; elements below lcm_thresh are dropped
 * (stream.jl:217).  strdim0/1: the two in-plane dimensions (0-based; the reference takes the dimension in which the
 * first orientation volume is zero everywhere as through-plane, stream.jl:221-223).
 * RANDOM-NUMBER CONTRACT.  The reference draws `rand(Categorical(lcm))` from Julia's global RNG, which no other
 * implementation can reproduce.  Here the k-th uniform consumed by streamline `line` (= seed*nsub + sub) is
 *     u = float(splitmix64(rng_seed ^ splitmix64(line * 0xD1342543DE82EF95 + k)) >> 40) * 2^-24   in [0,1),
 * and the category is the first index whose running sum of the normalised weights exceeds u (what
 * Distributions.jl's sampler does with its uniform).  Results depend on rng_seed only, not on scheduling.
 * fibd_stream_pack_flags additionally returns one byte per point: the LCM pick and the angle pick chose different
 * vectors (the `flags` the reference stores as a per-point scalar of the Tract, stream.jl:538, 666, 787). */
int fibd_stream_trace_lcm(const fib_stream_params *prm, const float *field4, const float *lcms, float lcm_thresh,
                          int32_t strdim0, int32_t strdim1, uint64_t rng_seed,
                          const int64_t *seeds, int64_t nseed, const float *sublist, int32_t nsub, void *stream,
                          fib_stream_job **job, int64_t *nlines, int64_t *npoints);
int fibd_stream_pack_flags(fib_stream_job *job, int32_t *npts, int64_t *seed_index, float *xyz, uint8_t *flags, void *stream);
/* same lines serialised as the body of a TrackVis .trk file (everything after the 1000-byte header, as
 * trk_write emits it, trk.jl:469-485): per line Int32 npts then npts x 3 Float32 = (xyz + .5) * voxel_size.
 * body: device buffer of 4*nlines + 12*npoints bytes. */
int fibd_stream_pack_trk(fib_stream_job *job, const float voxel_size[3], void *body, void *stream);
/* per-(seed,sub) point counts of every traced line, incl. those dropped by len_min: int32 [nseed*nsub] */
int fibd_stream_all_npts(fib_stream_job *job, int32_t *all_npts, void *stream);
void fib_stream_job_destroy(fib_stream_job *job);

/* ------------------------------------------------------------------------------------ */
/* Host-buffer drop-in entry points (what the Julia wrapper ccalls)                       */
/* ------------------------------------------------------------------------------------ */

/* The host tier mirrors the reference's own parallel decomposition — `Threads.@threads for iz` over z-slices in the fits
 * (dti.jl:258, gqi.jl:132, dsi.jl:197), contiguous seed chunks in `stream` (stream.jl:757-761) — with GPUs in place of
 * threads.  `device` is a HIP device index, or FIB_DEVICE_ALL for the device set declared with fib_init: the volume is
 * then cut into contiguous voxel slabs (one per entry, one host thread each), seeds are dealt round-robin, the global
 * odfmax (gqi.jl:164) is reduced over the slabs before qa is normalised, and streamlines are merged back into the
 * reference's (seed, sub) order.  Results do not depend on the device set.  Each entry of the set runs a three-stage
 * pipeline over voxel chunks (pinned staging ring: upload || kernels || download), and caches its plans and buffers
 * between calls; calls that share an entry are serialised, calls on different entries run concurrently.
 * fib_init(ndev, devs): devs[i] may repeat (two pipelines on one GPU); ndev == 0 selects every visible device (also
 * the default of FIB_DEVICE_ALL without fib_init).  fib_shutdown releases every cached plan, stream and buffer.
 * What a worker KEEPS between calls (grow-only, so that the next call of the same size allocates nothing): its pinned staging ring and the
 * ring's device mirror (3 x (rows in + rows out) x chunk x 4 bytes each: ~1.9 GB of host and of device memory after fib_gqi_rec on 270
 * frames), the device buffers of fib_stream (orientation field, seeds, and the packed result: 1.5 GB after 129 M points) and the tracer's
 * workspace (scratch for every line in flight).  A process that shares the GPU with other users of its memory calls fib_trim() when it
 * is done with a batch: everything listed above goes back to the driver (plans are kept: small, and costly to rebuild), the next call
 * re-allocates what it needs.  fib_trim waits for calls in flight; it returns FIB_OK. */
#define FIB_DEVICE_ALL (-1)
/* May be OR-ed into the mask_dtype of fib_dti_fit / fib_adc_fit / fib_gqi_rec / fib_dsi_rec: the caller's output arrays are zero already
 * (freshly allocated -- what the reference does itself: MRI(mask, n, Float32) -> zeros, mri.jl:251-255).  Voxels outside the mask
 * need then not be written, and where only the voxels inside the mask travel (a mask that keeps < 90 % of the volume in runs of 16
 * voxels or more) they are not: with a mask that keeps a third of the volume the scatter stage of the transfer pipeline writes a
 * third of the bytes.  Without the flag every output voxel is written (outside the mask: 0). */
#define FIB_MASK_OUTPUTS_ZEROED 0x100
int fib_init(int ndev, const int *devs);
int fib_trim(void);
void fib_shutdown(void);

/* dti_fit(dwi::MRI, mask::MRI)::DTI (dti.jl:221).  bval/bvec NULL or nvol<=0 reproduce the
 * reference's error() as FIB_ERR_MISSING_BVAL / FIB_ERR_MISSING_BVEC. */
int fib_dti_fit(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                const void *mask, int mask_dtype, const float *bval, const float *bvec,
                const fib_dti_out *out);
/* adc_fit(dwi::MRI, mask::MRI) (dti.jl:164) */
/* host-buffer form of fibd_st_eigen (structens.jl:13-37) */
int fib_st_eigen(int device, const float *const S[6], int64_t nvox, float *eigvec, float *eigval);
int fib_adc_fit(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                const void *mask, int mask_dtype, const float *bval, float *adc, float *s0);
/* gqi_rec(dwi, mask, odf_dirs, sigma)::GQI (gqi.jl:109) */
int fib_gqi_rec(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                const void *mask, int mask_dtype, const float *bval, const float *bvec,
                const float *verts, int nverts, const int32_t *faces, int nfaces, float sigma,
                float *odf, float *const peak[3], float *const qa[3]);
/* dsi_rec(dwi, mask, odf_dirs, hann_width)::DSI (dsi.jl:171) */
int fib_dsi_rec(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                const void *mask, int mask_dtype, const float *bval, const float *bvec,
                const float *verts, int nverts, const int32_t *faces, int nfaces, int hann_width,
                float *pdf, float *odf, float *const peak[3], float *const qa[3]);

/* rumba_rec(dwi, mask, odf_dirs, niter, lam_para, lam_perp, lam_csf, lam_gm, ncoils, coil_combine, ipat_factor, use_tv)
 * ::RUMBASD (rusd.jl:419); host buffers, outputs caller-allocated like the other fits. */
int fib_rumba_rec(int device, const float *dwi, int nx, int ny, int nz, int nvol, const void *mask, int mask_dtype,
                  const float *bval, const float *bvec, const float *verts, int nverts, int niter,
                  float lam_para, float lam_perp, float lam_csf, float lam_gm, int ncoils, int sos_grappa, int ipat_factor,
                  int use_tv, const fib_rumba_out *out, float *snr_mean, float *snr_std);

/* find_peaks!(W) (gqi.jl:180-201) for nvox ODFs in host memory.  odf [nvox x nvert] planar (row v = the nvox
 * amplitudes of half-sphere vertex v, like MRI.vol[:,:,:,v]); isort_top [3 x nvox] planar: the first three entries
 * of `isort` (0-based first-half vertex rows, -1 where the sphere has fewer vertices); nvalid [nvox] (gqi.jl:200). */
int fib_find_peaks(int device, const float *odf, int64_t nvox, const float *verts, int nverts,
                   const int32_t *faces, int nfaces, int32_t *isort_top, int32_t *nvalid);
/* find_peaks!(W) with every output the reference's work struct receives (gqi.jl:180-201): odf_peak [nvox x nvert] planar
 * (W.odf_peak: the amplitudes of the local peaks, 0 elsewhere, :184-196), isort [nvox x nvert] planar (W.isort: the complete
 * sortperm(odf_peak, rev=true), 0-based, :198) and nvalid [nvox] (the return value, :200). */
int fib_find_peaks_work(int device, const float *odf, int64_t nvox, const float *verts, int nverts,
                        const int32_t *faces, int nfaces, float *odf_peak, int32_t *isort, int32_t *nvalid);

/* stream(ovec; f, f_thresh, fa, fa_thresh, mask, seed, ...)::Tract (stream.jl:730), non-LCM macro path.
 * ovec[k] [nx,ny,nz,3]; f[k] [nx,ny,nz] or f == NULL; fa / mask / seed may be NULL (mask == NULL:
 * any-nonzero-vector mask, stream.jl:96-100; seed == NULL: brain mask seeds, stream.jl:744).
 * Output is library-allocated (release with fib_tract_free): */
typedef struct {
    int64_t nlines;       /* streamlines kept (npts >= len_min) */
    int64_t npoints;      /* sum of npts */
    int32_t *npts;        /* [nlines] */
    int64_t *seed_index;  /* [nlines] seed*nsub + sub, seeds counted in findall order */
    float *xyz;           /* [3*npoints] */
    uint8_t *flags;       /* [npoints] (fib_stream_lcm only, else NULL): LCM and angle pick disagreed at this point */
} fib_tract_out;

int fib_stream(int device, const fib_stream_params *prm, const float *const *ovec, const float *const *f,
               float f_thresh, const float *fa, float fa_thresh, const void *mask, int mask_dtype,
               const void *seed, int seed_dtype, const float *sublist, int32_t nsub, fib_tract_out *out);
/* stream(ovec; ..., lcms, lcm_thresh) (stream.jl:730): fib_stream with local connection matrices lcms [nx,ny,nz,10]
 * (see fibd_stream_trace_lcm for the semantics and the random-number contract); out->flags is filled. */
int fib_stream_lcm(int device, const fib_stream_params *prm, const float *const *ovec, const float *const *f,
                   float f_thresh, const float *fa, float fa_thresh, const void *mask, int mask_dtype,
                   const void *seed, int seed_dtype, const float *sublist, int32_t nsub,
                   const float *lcms, float lcm_thresh, uint64_t rng_seed, fib_tract_out *out);
void fib_tract_free(fib_tract_out *out);

#ifdef __cplusplus
}
#endif
#endif /* FIBERS_HIP_H */
