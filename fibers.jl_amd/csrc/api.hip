// api.hip — host-buffer ("drop-in") entry points: stage caller-owned host arrays through HBM,
// run the device-resident path (fibd_*), copy results back.  These are the functions the Julia
// wrapper ccalls in place of the bodies of dti_fit / adc_fit / gqi_rec / dsi_rec / stream.
#include <memory>

#include "common.h"

namespace {

// mask.vol[...] == 0 && continue (dti.jl:261, gqi.jl:135, dsi.jl:200)  -> nonzero test
// mask.vol .> 0 (stream.jl:102), seed.vol .> 0 (stream.jl:751)         -> positive test
template <typename T>
void mask_convert_t(const T *m, int64_t n, bool positive, uint8_t *out) {
    if (positive) for (int64_t i = 0; i < n; i++) out[i] = m[i] > (T)0 ? 1 : 0;
    else          for (int64_t i = 0; i < n; i++) out[i] = m[i] != (T)0 ? 1 : 0;
}

int mask_convert(const void *m, int dtype, int64_t n, bool positive, std::vector<uint8_t> &out) {
    out.resize((size_t)n);
    switch (dtype) {
        case FIB_U8: case FIB_BOOL: mask_convert_t((const uint8_t *)m, n, positive, out.data()); break;
        case FIB_I8:  mask_convert_t((const int8_t *)m, n, positive, out.data()); break;
        case FIB_I16: mask_convert_t((const int16_t *)m, n, positive, out.data()); break;
        case FIB_U16: mask_convert_t((const uint16_t *)m, n, positive, out.data()); break;
        case FIB_I32: mask_convert_t((const int32_t *)m, n, positive, out.data()); break;
        case FIB_U32: mask_convert_t((const uint32_t *)m, n, positive, out.data()); break;
        case FIB_I64: mask_convert_t((const int64_t *)m, n, positive, out.data()); break;
        case FIB_F32: mask_convert_t((const float *)m, n, positive, out.data()); break;
        case FIB_F64: mask_convert_t((const double *)m, n, positive, out.data()); break;
        default: return fib::fail(FIB_ERR_INVALID, "unknown mask dtype %d", dtype);
    }
    return FIB_OK;
}

struct PlanDeleter { void operator()(fib_dti_plan *p) const { fib_dti_plan_destroy(p); } };

#define RC(x) do { int _rc = (x); if (_rc != FIB_OK) return _rc; } while (0)

int h2d(void *dst, const void *src, size_t bytes) {
    FIB_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return FIB_OK;
}
int d2h(void *dst, const void *src, size_t bytes) {
    FIB_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return FIB_OK;
}

}  // namespace

extern "C" int fib_dti_fit(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                           const void *mask, int mask_dtype, const float *bval, const float *bvec,
                           const fib_dti_out *out) {
    FIB_CHECK(bval != nullptr && nvol > 0, FIB_ERR_MISSING_BVAL, "Missing b-value table from input DWI structure");
    FIB_CHECK(bvec != nullptr, FIB_ERR_MISSING_BVEC, "Missing gradient table from input DWI structure");
    FIB_CHECK(dwi && mask && out, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nx > 0 && ny > 0 && nz > 0, FIB_ERR_INVALID, "volume dimensions must be positive");
    fib::DeviceGuard guard;
    RC(fib::use_device(device));
    const int64_t nvox = (int64_t)nx * ny * nz;
    fib_dti_plan *praw = nullptr;
    RC(fib_dti_plan_create(device, bval, bvec, nvol, &praw));
    std::unique_ptr<fib_dti_plan, PlanDeleter> plan(praw);
    std::vector<uint8_t> m8;
    RC(mask_convert(mask, mask_dtype, nvox, false, m8));
    fib::DevBuf<float> d_dwi, d_out;
    fib::DevBuf<uint8_t> d_mask;
    RC(d_dwi.alloc((size_t)nvox * nvol));
    RC(d_mask.alloc((size_t)nvox));
    RC(d_out.alloc((size_t)nvox * 16));
    RC(h2d(d_dwi.p, dwi, sizeof(float) * nvox * nvol));
    RC(h2d(d_mask.p, m8.data(), (size_t)nvox));
    float *b = d_out.p;
    fib_dti_out dev{b, b + nvox, b + 2 * nvox, b + 3 * nvox, b + 4 * nvox, b + 7 * nvox, b + 10 * nvox,
                    b + 13 * nvox, b + 14 * nvox, b + 15 * nvox};
    RC(fibd_dti_fit(plan.get(), d_dwi.p, d_mask.p, nvox, &dev, nullptr));
    FIB_HIP(hipDeviceSynchronize());
    const size_t sb = sizeof(float) * nvox;
    RC(d2h(out->s0, dev.s0, sb));
    RC(d2h(out->eigval1, dev.eigval1, sb));
    RC(d2h(out->eigval2, dev.eigval2, sb));
    RC(d2h(out->eigval3, dev.eigval3, sb));
    RC(d2h(out->eigvec1, dev.eigvec1, 3 * sb));
    RC(d2h(out->eigvec2, dev.eigvec2, 3 * sb));
    RC(d2h(out->eigvec3, dev.eigvec3, 3 * sb));
    RC(d2h(out->rd, dev.rd, sb));
    RC(d2h(out->md, dev.md, sb));
    RC(d2h(out->fa, dev.fa, sb));
    return FIB_OK;
}

extern "C" int fib_st_eigen(int device, const float *const S[6], int64_t nvox, float *eigvec, float *eigval) {
    FIB_CHECK(S && eigvec && eigval, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nvox > 0, FIB_ERR_INVALID, "nvox must be positive");
    for (int k = 0; k < 6; k++) FIB_CHECK(S[k] != nullptr, FIB_ERR_INVALID, "NULL structure tensor volume %d", k);
    fib::DeviceGuard guard;
    RC(fib::use_device(device));
    fib::DevBuf<float> d_in, d_out;
    RC(d_in.alloc((size_t)nvox * 6));
    RC(d_out.alloc((size_t)nvox * 12));
    const float *dS[6];
    for (int k = 0; k < 6; k++) { dS[k] = d_in.p + (size_t)k * nvox; RC(h2d(d_in.p + (size_t)k * nvox, S[k], sizeof(float) * nvox)); }
    RC(fibd_st_eigen(dS, nvox, d_out.p, d_out.p + (size_t)9 * nvox, nullptr));
    FIB_HIP(hipDeviceSynchronize());
    RC(d2h(eigvec, d_out.p, sizeof(float) * nvox * 9));
    RC(d2h(eigval, d_out.p + (size_t)9 * nvox, sizeof(float) * nvox * 3));
    return FIB_OK;
}

extern "C" int fib_adc_fit(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                           const void *mask, int mask_dtype, const float *bval, float *adc, float *s0) {
    FIB_CHECK(bval != nullptr && nvol > 0, FIB_ERR_MISSING_BVAL, "Missing b-value table from input DWI structure");
    FIB_CHECK(dwi && mask && adc && s0, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nx > 0 && ny > 0 && nz > 0, FIB_ERR_INVALID, "volume dimensions must be positive");
    fib::DeviceGuard guard;
    RC(fib::use_device(device));
    const int64_t nvox = (int64_t)nx * ny * nz;
    fib_dti_plan *praw = nullptr;
    RC(fib_dti_plan_create(device, bval, nullptr, nvol, &praw));
    std::unique_ptr<fib_dti_plan, PlanDeleter> plan(praw);
    std::vector<uint8_t> m8;
    RC(mask_convert(mask, mask_dtype, nvox, false, m8));
    fib::DevBuf<float> d_dwi, d_out;
    fib::DevBuf<uint8_t> d_mask;
    RC(d_dwi.alloc((size_t)nvox * nvol));
    RC(d_mask.alloc((size_t)nvox));
    RC(d_out.alloc((size_t)nvox * 2));
    RC(h2d(d_dwi.p, dwi, sizeof(float) * nvox * nvol));
    RC(h2d(d_mask.p, m8.data(), (size_t)nvox));
    RC(fibd_adc_fit(plan.get(), d_dwi.p, d_mask.p, nvox, d_out.p, d_out.p + nvox, nullptr));
    FIB_HIP(hipDeviceSynchronize());
    RC(d2h(adc, d_out.p, sizeof(float) * nvox));
    RC(d2h(s0, d_out.p + nvox, sizeof(float) * nvox));
    return FIB_OK;
}

// ------------------------------------------------------------------------------------------
// gqi_rec / dsi_rec
// ------------------------------------------------------------------------------------------
namespace {

struct OdfPlanDeleter { void operator()(fib_odf_plan *p) const { fib_odf_plan_destroy(p); } };

int odf_rec_host(fib_odf_plan *praw, int nvol, const float *dwi, int nx, int ny, int nz,
                 const void *mask, int mask_dtype, int nvert, float *pdf, float *odf,
                 float *const peak[3], float *const qa[3]) {
    std::unique_ptr<fib_odf_plan, OdfPlanDeleter> plan(praw);
    FIB_CHECK(dwi && mask && odf && peak && qa, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nx > 0 && ny > 0 && nz > 0, FIB_ERR_INVALID, "volume dimensions must be positive");
    for (int k = 0; k < 3; k++) FIB_CHECK(peak[k] && qa[k], FIB_ERR_INVALID, "NULL peak/qa output volume");
    const int64_t nvox = (int64_t)nx * ny * nz;
    std::vector<uint8_t> m8;
    RC(mask_convert(mask, mask_dtype, nvox, false, m8));
    fib::DevBuf<float> d_dwi, d_odf, d_pdf, d_pq;
    fib::DevBuf<uint8_t> d_mask;
    RC(d_dwi.alloc((size_t)nvox * nvol));
    RC(d_mask.alloc((size_t)nvox));
    RC(d_odf.alloc((size_t)nvox * nvert));
    if (pdf) RC(d_pdf.alloc((size_t)nvox * nvol));
    RC(d_pq.alloc((size_t)nvox * 12));
    RC(h2d(d_dwi.p, dwi, sizeof(float) * nvox * nvol));
    RC(h2d(d_mask.p, m8.data(), (size_t)nvox));
    float *pk[3] = {d_pq.p, d_pq.p + 3 * nvox, d_pq.p + 6 * nvox};
    float *q[3] = {d_pq.p + 9 * nvox, d_pq.p + 10 * nvox, d_pq.p + 11 * nvox};
    RC(fibd_odf_rec(plan.get(), d_dwi.p, d_mask.p, nvox, pdf ? d_pdf.p : nullptr, d_odf.p, pk, q, nullptr, 1, nullptr));
    FIB_HIP(hipDeviceSynchronize());
    RC(d2h(odf, d_odf.p, sizeof(float) * nvox * nvert));
    if (pdf) RC(d2h(pdf, d_pdf.p, sizeof(float) * nvox * nvol));
    for (int k = 0; k < 3; k++) {
        RC(d2h(peak[k], pk[k], sizeof(float) * nvox * 3));
        RC(d2h(qa[k], q[k], sizeof(float) * nvox));
    }
    return FIB_OK;
}

}  // namespace

extern "C" int fib_gqi_rec(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                           const void *mask, int mask_dtype, const float *bval, const float *bvec,
                           const float *verts, int nverts, const int32_t *faces, int nfaces, float sigma,
                           float *odf, float *const peak[3], float *const qa[3]) {
    fib::DeviceGuard guard;
    fib_odf_plan *p = nullptr;
    RC(fib_gqi_plan_create(device, bval, bvec, nvol, verts, nverts, faces, nfaces, sigma, &p));
    FIB_HIP(hipSetDevice(device));
    return odf_rec_host(p, nvol, dwi, nx, ny, nz, mask, mask_dtype, nverts / 2, nullptr, odf, peak, qa);
}

extern "C" int fib_dsi_rec(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                           const void *mask, int mask_dtype, const float *bval, const float *bvec,
                           const float *verts, int nverts, const int32_t *faces, int nfaces, int hann_width,
                           float *pdf, float *odf, float *const peak[3], float *const qa[3]) {
    FIB_CHECK(pdf != nullptr, FIB_ERR_INVALID, "NULL pdf output volume");
    fib::DeviceGuard guard;
    fib_odf_plan *p = nullptr;
    RC(fib_dsi_plan_create(device, bval, bvec, nvol, verts, nverts, faces, nfaces, hann_width, &p));
    FIB_HIP(hipSetDevice(device));
    return odf_rec_host(p, nvol, dwi, nx, ny, nz, mask, mask_dtype, nverts / 2, pdf, odf, peak, qa);
}

// rumba_rec (rusd.jl:419-636), host buffers
extern "C" int fib_rumba_rec(int device, const float *dwi, int nx, int ny, int nz, int nvol, const void *mask, int mask_dtype,
                             const float *bval, const float *bvec, const float *verts, int nverts, int niter,
                             float lam_para, float lam_perp, float lam_csf, float lam_gm, int ncoils, int sos_grappa, int ipat_factor,
                             int use_tv, const fib_rumba_out *out, float *snr_mean, float *snr_std) {
    FIB_CHECK(dwi && mask && out, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nx > 0 && ny > 0 && nz > 0, FIB_ERR_INVALID, "volume dimensions must be positive");
    fib::DeviceGuard guard;
    fib_rumba_plan *p = nullptr;
    RC(fib_rumba_plan_create(device, bval, bvec, nvol, verts, nverts, lam_para, lam_perp, lam_csf, lam_gm, &p));
    struct PlanDel { fib_rumba_plan *p; ~PlanDel() { fib_rumba_plan_destroy(p); } } del{p};
    FIB_HIP(hipSetDevice(device));
    const int64_t nvox = (int64_t)nx * ny * nz;
    const int nvert = nverts / 2;
    std::vector<uint8_t> m8;
    RC(mask_convert(mask, mask_dtype, nvox, true, m8));            // mask.vol .> 0, rusd.jl:446
    fib::DevBuf<float> d_dwi, d_fodf, d_sc, d_pk;
    fib::DevBuf<uint8_t> d_mask;
    RC(d_dwi.alloc((size_t)nvox * nvol));
    RC(d_mask.alloc((size_t)nvox));
    RC(d_fodf.alloc((size_t)nvox * nvert));
    RC(d_sc.alloc((size_t)nvox * 4));
    RC(d_pk.alloc((size_t)nvox * 15));
    RC(h2d(d_dwi.p, dwi, sizeof(float) * nvox * nvol));
    RC(h2d(d_mask.p, m8.data(), (size_t)nvox));
    fib_rumba_out dev{};
    dev.fodf = d_fodf.p; dev.fgm = d_sc.p; dev.fcsf = d_sc.p + nvox; dev.gfa = d_sc.p + 2 * nvox; dev.var = d_sc.p + 3 * nvox;
    for (int i = 0; i < 5; i++) dev.peak[i] = d_pk.p + (size_t)i * 3 * nvox;
    RC(fibd_rumba_rec(p, d_dwi.p, d_mask.p, nx, ny, nz, niter, ncoils, sos_grappa, ipat_factor, use_tv, &dev, snr_mean, snr_std, nullptr));
    RC(d2h(out->fodf, dev.fodf, sizeof(float) * nvox * nvert));
    RC(d2h(out->fgm, dev.fgm, sizeof(float) * nvox));
    RC(d2h(out->fcsf, dev.fcsf, sizeof(float) * nvox));
    RC(d2h(out->gfa, dev.gfa, sizeof(float) * nvox));
    RC(d2h(out->var, dev.var, sizeof(float) * nvox));
    for (int i = 0; i < 5; i++) RC(d2h(out->peak[i], dev.peak[i], sizeof(float) * nvox * 3));
    return FIB_OK;
}

// find_peaks!(W) (gqi.jl:180-201) for nvox ODFs held in host memory: odf [nvox x nvert] planar (vertex-major rows of
// nvox values, like MRI.vol[:,:,:,v]); isort_top [3 x nvox] planar, 0-based first-half vertex rows, -1 where the
// tessellation has fewer vertices; nvalid [nvox] = count(odf_peak .> 0) (gqi.jl:200).
extern "C" int fib_find_peaks(int device, const float *odf, int64_t nvox, const float *verts, int nverts,
                              const int32_t *faces, int nfaces, int32_t *isort_top, int32_t *nvalid) {
    FIB_CHECK(odf && verts && faces && isort_top && nvalid, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nvox > 0 && nverts >= 2 && nverts % 2 == 0 && nfaces > 0, FIB_ERR_INVALID, "invalid sizes");
    fib::DeviceGuard guard;
    // the plan only contributes the folded neighbour table: one dummy frame is enough
    const float bval1[1] = {1000.0f}, bvec1[3] = {1.0f, 0.0f, 0.0f};
    fib_odf_plan *p = nullptr;
    RC(fib_gqi_plan_create(device, bval1, bvec1, 1, verts, nverts, faces, nfaces, 1.25f, &p));
    struct PlanDel { fib_odf_plan *p; ~PlanDel() { fib_odf_plan_destroy(p); } } del{p};
    FIB_HIP(hipSetDevice(device));
    const int nvert = nverts / 2;
    fib::DevBuf<float> d_odf;
    fib::DevBuf<int32_t> d_top, d_nv;
    RC(d_odf.alloc((size_t)nvox * nvert));
    RC(d_top.alloc((size_t)nvox * 3));
    RC(d_nv.alloc((size_t)nvox));
    FIB_HIP(hipMemcpy(d_odf.p, odf, (size_t)nvox * nvert * sizeof(float), hipMemcpyHostToDevice));
    RC(fibd_find_peaks(p, d_odf.p, nvox, d_top.p, d_nv.p, nullptr));
    FIB_HIP(hipDeviceSynchronize());
    FIB_HIP(hipMemcpy(isort_top, d_top.p, (size_t)nvox * 3 * sizeof(int32_t), hipMemcpyDeviceToHost));
    FIB_HIP(hipMemcpy(nvalid, d_nv.p, (size_t)nvox * sizeof(int32_t), hipMemcpyDeviceToHost));
    return FIB_OK;
}

// ------------------------------------------------------------------------------------------
// stream
// ------------------------------------------------------------------------------------------
extern "C" void fib_tract_free(fib_tract_out *out) {
    if (!out) return;
    free(out->npts); free(out->seed_index); free(out->xyz); free(out->flags);
    out->npts = nullptr; out->seed_index = nullptr; out->xyz = nullptr; out->flags = nullptr;
    out->nlines = 0; out->npoints = 0;
}

static int stream_host(int device, const fib_stream_params *prm, const float *const *ovec, const float *const *f,
                       float f_thresh, const float *fa, float fa_thresh, const void *mask, int mask_dtype,
                       const void *seed, int seed_dtype, const float *sublist, int32_t nsub,
                       const float *lcms, float lcm_thresh, uint64_t rng_seed, fib_tract_out *out);

extern "C" int fib_stream(int device, const fib_stream_params *prm, const float *const *ovec, const float *const *f,
                          float f_thresh, const float *fa, float fa_thresh, const void *mask, int mask_dtype,
                          const void *seed, int seed_dtype, const float *sublist, int32_t nsub, fib_tract_out *out) {
    return stream_host(device, prm, ovec, f, f_thresh, fa, fa_thresh, mask, mask_dtype, seed, seed_dtype, sublist, nsub,
                       nullptr, 0.0f, 0, out);
}

extern "C" int fib_stream_lcm(int device, const fib_stream_params *prm, const float *const *ovec, const float *const *f,
                              float f_thresh, const float *fa, float fa_thresh, const void *mask, int mask_dtype,
                              const void *seed, int seed_dtype, const float *sublist, int32_t nsub,
                              const float *lcms, float lcm_thresh, uint64_t rng_seed, fib_tract_out *out) {
    FIB_CHECK(lcms != nullptr, FIB_ERR_INVALID, "NULL lcms volume");
    return stream_host(device, prm, ovec, f, f_thresh, fa, fa_thresh, mask, mask_dtype, seed, seed_dtype, sublist, nsub,
                       lcms, lcm_thresh, rng_seed, out);
}

static int stream_host(int device, const fib_stream_params *prm, const float *const *ovec, const float *const *f,
                       float f_thresh, const float *fa, float fa_thresh, const void *mask, int mask_dtype,
                       const void *seed, int seed_dtype, const float *sublist, int32_t nsub,
                       const float *lcms, float lcm_thresh, uint64_t rng_seed, fib_tract_out *out) {
    FIB_CHECK(prm && ovec && sublist && out, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(prm->nx > 0 && prm->ny > 0 && prm->nz > 0, FIB_ERR_INVALID, "volume dimensions must be positive");
    FIB_CHECK(prm->nvec >= 1 && prm->nvec <= 8, FIB_ERR_UNSUPPORTED, "1..8 orientation vectors per voxel are supported");
    FIB_CHECK(nsub >= 1, FIB_ERR_INVALID, "sublist must hold at least one offset");
    memset(out, 0, sizeof *out);
    fib::DeviceGuard guard;
    RC(fib::use_device(device));
    const int nvec = prm->nvec;
    const int64_t nvox = (int64_t)prm->nx * prm->ny * prm->nz;
    fib::DevBuf<float> d_vec, d_f, d_fa, d_field, d_sub;
    fib::DevBuf<uint8_t> d_mask, d_mout;
    RC(d_vec.alloc((size_t)nvox * 3 * nvec));
    RC(d_field.alloc((size_t)nvox * 4 * nvec));
    RC(d_mout.alloc((size_t)nvox));
    const float *dv[8] = {}, *df[8] = {};
    for (int k = 0; k < nvec; k++) {
        FIB_CHECK(ovec[k] != nullptr, FIB_ERR_INVALID, "NULL orientation volume %d", k);
        RC(h2d(d_vec.p + (size_t)k * nvox * 3, ovec[k], sizeof(float) * nvox * 3));
        dv[k] = d_vec.p + (size_t)k * nvox * 3;
    }
    if (f) {
        RC(d_f.alloc((size_t)nvox * nvec));
        for (int k = 0; k < nvec; k++) {
            FIB_CHECK(f[k] != nullptr, FIB_ERR_INVALID, "NULL amplitude volume %d", k);
            RC(h2d(d_f.p + (size_t)k * nvox, f[k], sizeof(float) * nvox));
            df[k] = d_f.p + (size_t)k * nvox;
        }
    }
    if (fa) { RC(d_fa.alloc((size_t)nvox)); RC(h2d(d_fa.p, fa, sizeof(float) * nvox)); }
    std::vector<uint8_t> m8;
    if (mask) {
        RC(mask_convert(mask, mask_dtype, nvox, true, m8));        // mask.vol .> 0, stream.jl:102
        RC(d_mask.alloc((size_t)nvox));
        RC(h2d(d_mask.p, m8.data(), (size_t)nvox));
    }
    RC(fibd_stream_field(nvec, nvox, dv, f ? df : nullptr, f_thresh, fa ? d_fa.p : nullptr, fa_thresh,
                         mask ? d_mask.p : nullptr, d_field.p, d_mout.p, nullptr));
    // seed voxels: findall(W.mask) (stream.jl:744) or findall(seed.vol .> 0) (stream.jl:751), column-major order
    std::vector<uint8_t> s8;
    if (seed) {
        RC(mask_convert(seed, seed_dtype, nvox, true, s8));
    } else {
        s8.resize((size_t)nvox);
        FIB_HIP(hipDeviceSynchronize());
        RC(d2h(s8.data(), d_mout.p, (size_t)nvox));
    }
    std::vector<int64_t> seeds;
    for (int64_t i = 0; i < nvox; i++) if (s8[i]) seeds.push_back(i);
    fib::DevBuf<int64_t> d_seeds;
    RC(d_seeds.alloc(seeds.size()));
    if (!seeds.empty()) RC(h2d(d_seeds.p, seeds.data(), sizeof(int64_t) * seeds.size()));
    RC(d_sub.alloc((size_t)nsub * 3));
    RC(h2d(d_sub.p, sublist, sizeof(float) * 3 * nsub));
    fib_stream_job *job = nullptr;
    int64_t nl = 0, np = 0;
    fib::DevBuf<float> d_lcms;
    if (lcms) {
        // through-plane dimension = the one in which the first orientation volume is zero everywhere (stream.jl:221-223)
        bool allzero[3] = {true, true, true};
        for (int c = 0; c < 3; c++)
            for (int64_t i = 0; i < nvox && allzero[c]; i++) if (ovec[0][(size_t)c * nvox + i] != 0.0f) allzero[c] = false;
        int strd[3], ns = 0;
        for (int c = 0; c < 3; c++) if (!allzero[c]) strd[ns++] = c;
        FIB_CHECK(ns >= 2, FIB_ERR_INVALID, "LCM-guided tracking needs two in-plane dimensions with non-zero orientation components");
        RC(d_lcms.alloc((size_t)nvox * 10));
        RC(h2d(d_lcms.p, lcms, sizeof(float) * nvox * 10));
        RC(fibd_stream_trace_lcm(prm, d_field.p, d_lcms.p, lcm_thresh, strd[0], strd[1], rng_seed, d_seeds.p, (int64_t)seeds.size(),
                                 d_sub.p, nsub, nullptr, &job, &nl, &np));
    } else
    RC(fibd_stream_trace(prm, d_field.p, d_seeds.p, (int64_t)seeds.size(), d_sub.p, nsub, nullptr, &job, &nl, &np));
    struct JobGuard { fib_stream_job *j; ~JobGuard() { fib_stream_job_destroy(j); } } jg{job};
    out->nlines = nl; out->npoints = np;
    out->npts = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nl > 0 ? nl : 1));
    out->seed_index = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nl > 0 ? nl : 1));
    out->xyz = (float *)malloc(sizeof(float) * 3 * (size_t)(np > 0 ? np : 1));
    if (lcms) out->flags = (uint8_t *)malloc((size_t)(np > 0 ? np : 1));
    if (!out->npts || !out->seed_index || !out->xyz || (lcms && !out->flags)) { fib_tract_free(out); return fib::fail(FIB_ERR_NOMEM, "out of host memory"); }
    if (nl > 0) {
        fib::DevBuf<int32_t> d_npts;
        fib::DevBuf<int64_t> d_sidx;
        fib::DevBuf<float> d_xyz;
        fib::DevBuf<uint8_t> d_flags;
        int rc = d_npts.alloc((size_t)nl);
        if (rc == FIB_OK) rc = d_sidx.alloc((size_t)nl);
        if (rc == FIB_OK) rc = d_xyz.alloc((size_t)np * 3);
        if (rc == FIB_OK && lcms) rc = d_flags.alloc((size_t)np);
        if (rc == FIB_OK) rc = fibd_stream_pack_flags(job, d_npts.p, d_sidx.p, d_xyz.p, lcms ? d_flags.p : nullptr, nullptr);
        if (rc == FIB_OK && hipDeviceSynchronize() != hipSuccess) rc = fib::fail(FIB_ERR_HIP, "streamline pack failed");
        if (rc == FIB_OK) rc = d2h(out->npts, d_npts.p, sizeof(int32_t) * nl);
        if (rc == FIB_OK) rc = d2h(out->seed_index, d_sidx.p, sizeof(int64_t) * nl);
        if (rc == FIB_OK) rc = d2h(out->xyz, d_xyz.p, sizeof(float) * 3 * np);
        if (rc == FIB_OK && lcms) rc = d2h(out->flags, d_flags.p, (size_t)np);
        if (rc != FIB_OK) { fib_tract_free(out); return rc; }
    }
    return FIB_OK;
}
