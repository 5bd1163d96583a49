// setup.cpp — host-side construction of the reference's work structs (run once per plan).
//   DTIwork/ADCwork  dti.jl:39-84,101-155   -> host_dti_design + host_pinv
//   GQIwork          gqi.jl:32-82           -> host_gqi_matrix + host_neighbours
//   DSIwork          dsi.jl:41-143          -> host_dsi_matrix (+ the linear per-voxel chain dsi.jl:204-242)
// Nothing here runs per voxel; tables are built in float64 where the reference uses LAPACK/FFTW
// float32 routines and rounded once to float32.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>

#include "common.h"

namespace fib {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// the library's only reader of the process environment (common.h lists the names)
const char *env(const char *name) { return std::getenv(name); }

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

const char *last_error() { return g_err; }

int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(FIB_ERR_NO_DEVICE, "no HIP device available (%s); libfibers_hip has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n) return fail(FIB_ERR_NO_DEVICE, "device %d out of range [0,%d)", device, n);
    FIB_HIP(hipSetDevice(device));
    return FIB_OK;
}

// ------------------------------------------------------------------------------------------
// pinv: one-sided (Hestenes) Jacobi SVD in float64
// ------------------------------------------------------------------------------------------
void host_pinv(const float *A, int m, int n, float *pA) {
    std::vector<double> U((size_t)m * n), V((size_t)n * n, 0.0), sig(n);
    for (size_t i = 0; i < (size_t)m * n; i++) U[i] = A[i];
    for (int j = 0; j < n; j++) V[j + (size_t)n * j] = 1.0;
    for (int sweep = 0; sweep < 80; sweep++) {
        bool rotated = false;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double *up = &U[(size_t)m * p], *uq = &U[(size_t)m * q];
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < m; i++) { alpha += up[i] * up[i]; beta += uq[i] * uq[i]; gamma += up[i] * uq[i]; }
                if (gamma == 0.0 || std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
                rotated = true;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < m; i++) { double a = up[i], b = uq[i]; up[i] = c * a - s * b; uq[i] = s * a + c * b; }
                double *vp = &V[(size_t)n * p], *vq = &V[(size_t)n * q];
                for (int i = 0; i < n; i++) { double a = vp[i], b = vq[i]; vp[i] = c * a - s * b; vq[i] = s * a + c * b; }
            }
        if (!rotated) break;
    }
    double smax = 0;
    for (int j = 0; j < n; j++) {
        double s = 0;
        for (int i = 0; i < m; i++) s += U[i + (size_t)m * j] * U[i + (size_t)m * j];
        sig[j] = std::sqrt(s);
        smax = std::max(smax, sig[j]);
    }
    // LinearAlgebra.pinv: rtol = eps(Float32) * min(m, n); keep S .> rtol * maximum(S)
    const double tol = (double)1.1920929e-07f * std::min(m, n) * smax;
    for (int r = 0; r < n; r++)
        for (int i = 0; i < m; i++) {
            double acc = 0;
            for (int j = 0; j < n; j++)
                if (sig[j] > tol) acc += V[r + (size_t)n * j] * U[i + (size_t)m * j] / (sig[j] * sig[j]);
            pA[r + (size_t)n * i] = (float)acc;   // U columns are unnormalised: u = U/sig
        }
}

// ------------------------------------------------------------------------------------------
// DTI / ADC design matrix (float32 like the reference)
// ------------------------------------------------------------------------------------------
void host_dti_design(const float *bval, const float *bvec, int nvol, int np, float *A) {
    for (int i = 0; i < nvol; i++) {
        const float nb = -bval[i];
        if (np == 2) {                          // ADCwork, dti.jl:68-69
            A[i] = nb;
            A[i + nvol] = 1.0f;
            continue;
        }
        const float gx = bvec[i], gy = bvec[i + nvol], gz = bvec[i + 2 * nvol];
        A[i + 0 * nvol] = (gx * gx) * nb;       // dti.jl:131-138
        A[i + 1 * nvol] = ((2.0f * gx) * gy) * nb;
        A[i + 2 * nvol] = ((2.0f * gx) * gz) * nb;
        A[i + 3 * nvol] = (gy * gy) * nb;
        A[i + 4 * nvol] = ((2.0f * gy) * gz) * nb;
        A[i + 5 * nvol] = (gz * gz) * nb;
        A[i + 6 * nvol] = 1.0f;                 // dti.jl:140
    }
}

// ------------------------------------------------------------------------------------------
// GQI system matrix
// ------------------------------------------------------------------------------------------
void host_gqi_matrix(const float *bval, const float *bvec, int nvol, const float *verts, int nverts,
                     float sigma, float *A) {
    const int nvert = nverts / 2;
    const float pif = 3.14159274101257324f;
    const float sop = sigma / pif;              // T(sigma / pi), gqi.jl:68
    for (int j = 0; j < nvol; j++) {
        const float sc = std::sqrt(bval[j] * 0.01506f) * sop;
        const float bq[3] = {bvec[j] * sc, bvec[j + nvol] * sc, bvec[j + 2 * nvol] * sc};
        for (int v = 0; v < nvert; v++) {
            const int r = nvert + v;            // second half of the sphere, gqi.jl:69
            float x = verts[r] * bq[0];
            x += verts[r + nverts] * bq[1];
            x += verts[r + 2 * nverts] * bq[2];
            const double xd = x;
            const double y = (xd == 0.0) ? 1.0 : std::sin(M_PI * xd) / (M_PI * xd);   // Base.sinc
            A[v + (size_t)nvert * j] = (float)y;
        }
    }
}

// ------------------------------------------------------------------------------------------
// DSI as two dense maps
// ------------------------------------------------------------------------------------------
int host_dsi_matrix(const float *bval, const float *bvec, int nvol, const float *verts, int nverts,
                    int hann_width, float *A, int *scale_frame, float *scale_coef, std::vector<int> *iq_out) {
    const int nvert = nverts / 2;
    float bmin = bval[0];
    for (int j = 1; j < nvol; j++) bmin = std::min(bmin, bval[j]);
    float b1 = INFINITY;
    for (int j = 0; j < nvol; j++) if (bval[j] > bmin) b1 = std::min(b1, bval[j]);
    if (!(b1 < INFINITY)) return fail(FIB_ERR_UNSUPPORTED, "DSI needs at least two distinct b-values");
    const float dq = std::sqrt(b1);                                   // dsi.jl:66
    std::vector<int> iq((size_t)3 * nvol);
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int j = 0; j < nvol; j++) {
        const float sb = std::sqrt(bval[j]);
        for (int c = 0; c < 3; c++) {
            const float q = bvec[j + c * nvol] * sb;                  // dsi.jl:62
            const int v = (int)std::nearbyint(q / dq);               // dsi.jl:67 (ties to even)
            iq[3 * j + c] = v;
            lo = std::min(lo, v); hi = std::max(hi, v);
        }
    }
    if (iq_out) *iq_out = iq;
    int nfft = 1;
    while (nfft < hi - lo + 1) nfft *= 2;                             // dsi.jl:70-71
    const int shift = nfft / 2 + 1;                                   // dsi.jl:73 (1-based)
    if (lo + shift < 1 || hi + shift > nfft)
        return fail(FIB_ERR_UNSUPPORTED, "q-space lattice [%d,%d] does not fit the %d^3 FFT grid (BoundsError in the reference)", lo, hi, nfft);
    if (nfft < 4) return fail(FIB_ERR_UNSUPPORTED, "q-space grid too small (nfft=%d)", nfft);

    // effective frame per lattice point: later frames overwrite earlier ones (dsi.jl:205)
    std::vector<int64_t> lin(nvol);
    std::vector<char> eff(nvol, 1);
    for (int j = 0; j < nvol; j++)
        lin[j] = (iq[3 * j] + shift - 1) + (int64_t)nfft * ((iq[3 * j + 1] + shift - 1) + (int64_t)nfft * (iq[3 * j + 2] + shift - 1));
    for (int j = 0; j < nvol; j++)
        for (int k = j + 1; k < nvol; k++) if (lin[k] == lin[j]) { eff[j] = 0; break; }
    std::vector<float> H(nvol);
    for (int j = 0; j < nvol; j++) {
        if (hann_width == 0) { H[j] = 1.0f; continue; }
        const double r = std::sqrt((double)(iq[3 * j] * iq[3 * j] + iq[3 * j + 1] * iq[3 * j + 1] + iq[3 * j + 2] * iq[3 * j + 2]));
        H[j] = (float)((1.0 + std::cos(r * (2.0 * M_PI / hann_width))) * 0.5);   // dsi.jl:84
    }
    std::vector<double> ctab(nfft);
    for (int k = 0; k <= nfft / 2; k++) ctab[k] = std::cos(2.0 * M_PI * k / nfft);
    for (int k = nfft / 2 + 1; k < nfft; k++) ctab[k] = ctab[nfft - k];   // exactly even: columns of q and -q are bit-identical
    auto cosk = [&](int64_t k) { int64_t r = k % nfft; if (r < 0) r += nfft; return ctab[r]; };

    *scale_frame = -1; *scale_coef = 0.0f;
    const int nrows = nvol + nvert;
    // pdf rows: p[iq_i] for a unit sample at frame j = H_j cos(2 pi iq_i.iq_j / nfft)  (dsi.jl:212-227)
    for (int j = 0; j < nvol; j++) {
        float *col = A + (size_t)nrows * j;
        if (!eff[j]) { for (int i = 0; i < nrows; i++) col[i] = 0.0f; continue; }
        if (iq[3 * j] == 0 && iq[3 * j + 1] == 0 && iq[3 * j + 2] == 0) {
            *scale_frame = j;                                          // sum(p) = nfft^3 * H(0) * s_j
            *scale_coef = (float)((double)nfft * nfft * nfft * H[j]);
        }
        for (int i = 0; i < nvol; i++) {
            const int64_t d = (int64_t)iq[3 * i] * iq[3 * j] + (int64_t)iq[3 * i + 1] * iq[3 * j + 1] + (int64_t)iq[3 * i + 2] * iq[3 * j + 2];
            col[i] = (float)((double)H[j] * cosk(d));
        }
    }
    // odf rows: dqr * sum_r qr2[r] * trilinear(p; v*qr[r] + shift)   (dsi.jl:104-109, 230-242)
    const int nrad = 21;
    float qr[nrad], qr2[nrad];
    for (int r = 0; r < nrad; r++) {
        const double t = 0.3 + 0.03 * r;                               // .3:.03:.9
        qr[r] = (float)(nfft / 2 - 1) * (float)t;
        qr2[r] = qr[r] * qr[r];
    }
    const float dqr = qr[1] - qr[0];
    struct Tap { int g[3]; double w; };
    std::vector<Tap> taps((size_t)8 * nrad);
    for (int v = 0; v < nvert; v++) {
        const int rv = nvert + v;                                      // second-half vertex, dsi.jl:108
        for (int r = 0; r < nrad; r++) {
            int i0[3]; double f[3];
            for (int c = 0; c < 3; c++) {
                const float x = verts[rv + c * nverts] * qr[r] + (float)shift;    // 1-based coordinate, float32
                int ix = (int)std::floor(x);
                ix = std::min(std::max(ix, 1), nfft - 1);
                i0[c] = ix - 1 - nfft / 2;                              // centred 0-based grid offset (r - nfft/2)
                f[c] = (double)(x - (float)ix);
            }
            for (int t = 0; t < 8; t++) {
                Tap &tp = taps[(size_t)8 * r + t];
                double w = (double)qr2[r];
                for (int c = 0; c < 3; c++) {
                    const int b = (t >> c) & 1;
                    tp.g[c] = i0[c] + b;
                    w *= b ? f[c] : 1.0 - f[c];
                }
                tp.w = w;
            }
        }
        for (int j = 0; j < nvol; j++) {
            float *col = A + (size_t)nrows * j;
            if (!eff[j]) { col[nvol + v] = 0.0f; continue; }
            double acc = 0;
            for (size_t t = 0; t < taps.size(); t++) {
                const Tap &tp = taps[t];
                const int64_t d = (int64_t)tp.g[0] * iq[3 * j] + (int64_t)tp.g[1] * iq[3 * j + 1] + (int64_t)tp.g[2] * iq[3 * j + 2];
                acc += tp.w * cosk(d);
            }
            col[nvol + v] = (float)(acc * (double)H[j] * (double)dqr);
        }
    }
    return FIB_OK;
}

// ------------------------------------------------------------------------------------------
// neighbour table of the folded half-sphere
// ------------------------------------------------------------------------------------------
int host_neighbours(const int32_t *faces, int nfaces, int nverts, std::vector<int32_t> &nbr, int *maxdeg) {
    const int nvert = nverts / 2;
    std::vector<std::vector<int32_t>> adj(nvert);
    for (int f = 0; f < nfaces; f++) {
        int32_t v[3];
        for (int c = 0; c < 3; c++) {
            int32_t x = faces[f + (size_t)nfaces * c];
            if (x < 1 || x > nverts) return fail(FIB_ERR_INVALID, "face %d references vertex %d outside 1..%d", f, x, nverts);
            if (x > nvert) x -= nvert;                                 // gqi.jl:64
            v[c] = x - 1;
        }
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                if (a == b) continue;
                // a vertex that meets ITSELF in a folded face is always zeroed (o[a] >= o[a]); keep the self edge
                auto &l = adj[v[a]];
                if (std::find(l.begin(), l.end(), v[b]) == l.end()) l.push_back(v[b]);
            }
    }
    int md = 0;
    for (auto &l : adj) md = std::max(md, (int)l.size());
    if (md > 16) return fail(FIB_ERR_UNSUPPORTED, "ODF vertex degree %d exceeds the supported maximum of 16", md);
    *maxdeg = md;
    nbr.assign((size_t)nvert * (md > 0 ? md : 1), -1);
    for (int v = 0; v < nvert; v++)
        for (size_t k = 0; k < adj[v].size(); k++) nbr[(size_t)v * md + k] = adj[v][k];
    return FIB_OK;
}

// ------------------------------------------------------------------------------------------
// per-kernel event timing
// ------------------------------------------------------------------------------------------
struct ProfPair { hipEvent_t first, second; int dev; };
struct ProfEntry { std::string name; std::vector<ProfPair> pending; double ms = 0; int64_t count = 0; };
static std::mutex g_prof_mu;
static std::vector<ProfEntry> g_prof;
static bool g_prof_on = false;
static std::vector<std::string> g_prof_filter;          // empty: every kernel
static std::vector<std::pair<int, hipEvent_t>> g_prof_pool;   // (device, event) between uses: an event is recorded on streams of the device it was created on

bool profiling_on() { return g_prof_on; }

bool profile_wants(const char *name) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_filter.empty()) return true;
    for (auto &f : g_prof_filter) if (f == name) return true;
    return false;
}

bool profile_events(hipEvent_t *a, hipEvent_t *b) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    hipEvent_t ev[2] = {nullptr, nullptr};
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        int got = 0;
        for (size_t i = g_prof_pool.size(); i-- > 0 && got < 2;)
            if (g_prof_pool[i].first == dev) { ev[got++] = g_prof_pool[i].second; g_prof_pool.erase(g_prof_pool.begin() + (long)i); }
    }
    for (int i = 0; i < 2; i++)
        if (!ev[i] && hipEventCreate(&ev[i]) != hipSuccess) {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            for (int j = 0; j < 2; j++) if (ev[j]) g_prof_pool.emplace_back(dev, ev[j]);
            return false;
        }
    *a = ev[0]; *b = ev[1];
    return true;
}
static void profile_recycle(const ProfPair &pr) {         // (g_prof_mu held) back to the pool of the device the pair belongs to
    g_prof_pool.emplace_back(pr.dev, pr.first);
    g_prof_pool.emplace_back(pr.dev, pr.second);
}

void profile_push(const char *name, hipEvent_t a, hipEvent_t b) {
    int dev = 0;
    (void)hipGetDevice(&dev);                                // (the scope that recorded the pair runs under the device it launches on)
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &e : g_prof) if (e.name == name) { e.pending.push_back(ProfPair{a, b, dev}); return; }
    g_prof.emplace_back();
    g_prof.back().name = name;
    g_prof.back().pending.push_back(ProfPair{a, b, dev});
}

// a host-side stage's duration (the host tier's gather / scatter stages), into the same table
void profile_add_ms(const char *name, double ms) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &e : g_prof) if (e.name == name) { e.ms += ms; e.count++; return; }
    g_prof.emplace_back();
    g_prof.back().name = name;
    g_prof.back().ms = ms; g_prof.back().count = 1;
}

}  // namespace fib

extern "C" int fib_profile_enable(int on) { fib::g_prof_on = on != 0; return FIB_OK; }

// only the kernels named in the comma-separated list are bracketed (NULL or "": every kernel).  An event pair is two packets in the queue:
// bench.py brackets the one kernel its roofline is about during the timed steps
extern "C" int fib_profile_filter(const char *names) try {
    std::lock_guard<std::mutex> lk(fib::g_prof_mu);
    fib::g_prof_filter.clear();
    if (!names) return FIB_OK;
    std::string cur;
    for (const char *c = names;; c++) {
        if (*c == ',' || *c == '\0') { if (!cur.empty()) fib::g_prof_filter.push_back(cur); cur.clear(); if (*c == '\0') break; }
        else if (*c != ' ') cur.push_back(*c);
    }
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fib_profile_reset(void) try {
    std::lock_guard<std::mutex> lk(fib::g_prof_mu);
    for (auto &e : fib::g_prof) {
        for (auto &pr : e.pending) { (void)hipEventSynchronize(pr.second); fib::profile_recycle(pr); }
    }
    fib::g_prof.clear();
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fib_profile_get(const char *kernel, double *total_ms, int64_t *count) try {
    FIB_CHECK(kernel && total_ms && count, FIB_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(fib::g_prof_mu);
    *total_ms = 0; *count = 0;
    for (auto &e : fib::g_prof) {
        if (e.name != kernel) continue;
        for (auto &pr : e.pending) {
            float ms = 0;
            if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { e.ms += ms; e.count++; }
            fib::profile_recycle(pr);
        }
        e.pending.clear();
        *total_ms = e.ms; *count = e.count;
    }
    return FIB_OK;
} FIB_API_CATCH

extern "C" const char *fib_last_error(void) { return fib::last_error(); }
extern "C" const char *fib_version(void) { return "fibers-hip 0.1 (gfx950)"; }
extern "C" int fib_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} FIB_API_CATCH
