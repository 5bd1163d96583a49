// common.h — internal helpers shared by the translation units of libfibers_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <vector>

#include "../../include/fibers_hip.h"

namespace fib {

// thread-local last-error message (fib_last_error)
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define FIB_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t _e = (call);                                                              \
        if (_e != hipSuccess)                                                                \
            return fib::fail(FIB_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                            \
    } while (0)

#define FIB_CHECK(cond, code, ...)                          \
    do {                                                    \
        if (!(cond)) return fib::fail((code), __VA_ARGS__); \
    } while (0)

// No C++ exception crosses the C ABI (include/fibers_hip.h): every extern "C" body is a function-try-block that ends in one of these.
#define FIB_API_CATCH                                                                                                  \
    catch (const std::bad_alloc &) { return fib::fail(FIB_ERR_NOMEM, "out of host memory"); }                           \
    catch (const std::exception &e_) { return fib::fail(FIB_ERR_INVALID, "internal error: %s", e_.what()); }            \
    catch (...) { return fib::fail(FIB_ERR_INVALID, "internal error (unknown exception)"); }
#define FIB_API_CATCH_VOID catch (...) { }

// selects `device` after validating it; FIB_ERR_NO_DEVICE when there is none (no CPU fallback)
int use_device(int device);

struct DeviceGuard {  // restores the caller's current device on scope exit
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

template <typename T>
struct DevBuf {  // RAII device allocation
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    int alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return fail(FIB_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e)); }
        n = count;
        return FIB_OK;
    }
    int ensure(size_t count) { return (count <= n && p) ? FIB_OK : alloc(count); }
};

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Environment variables.  The product library reads these names, none of which selects a kernel variant: FIBERS_ODF_FORMAT (what
// FIB_ODF_FORMAT_DEFAULT means: fp16x2 | bf16x3 | f32 -- the format itself is a plan-creation parameter; its former spellings
// FIBERS_ODF_GEMM=f32 / FIBERS_ODF_EXACT=1 are still honoured) and the host tier's pipeline shape (FIBERS_HOST_CHUNK, FIBERS_HOST_PACK,
// FIBERS_COPY_THREADS, FIBERS_HOST_NT), through env() below.  The A/B partners of the shipped kernels
// and the knobs of the measurement tools (schedule variants, the split of an XCD's workgroups, the list unit, the phase stamps) exist
// in the DIAGNOSTIC build only (`make stamp`: -DFIB_CLOCK_STAMP -DFIB_AB_VARIANTS -> libfibers_hip_stamp.so, which tools/ load
// through FIBERS_HIP_LIB): ab_env() is a constant nullptr in the product.
const char *env(const char *name);
#ifdef FIB_AB_VARIANTS
static inline const char *ab_env(const char *name) { return env(name); }
#else
static inline const char *ab_env(const char *) { return nullptr; }
#endif

// hipEvent bracket around a kernel launch when fib_profile_enable(1) is active (no-op otherwise)
bool profiling_on();
bool profile_wants(const char *name);                 // fib_profile_filter: only the named kernels are bracketed (every event is a packet in the queue)
bool profile_events(hipEvent_t *a, hipEvent_t *b);   // a pair from the pool (events are reused: creating one costs more than recording it)
void profile_push(const char *name, hipEvent_t a, hipEvent_t b);
void profile_add_ms(const char *name, double ms);
struct ProfScope {
    const char *name; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(const char *n, hipStream_t s) : name(n), st(s) {
        if (!profiling_on() || !profile_wants(n)) return;
        if (!profile_events(&a, &b)) { a = b = nullptr; return; }
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        profile_push(name, a, b);
    }
};

// ---- host-side work-struct mathematics (setup.cpp) ----------------------------------------
// pinv of a column-major float32 [m x n] matrix (n <= 8): float64 one-sided Jacobi SVD,
// Julia's LinearAlgebra.pinv cut-off (singular values <= eps(Float32)*min(m,n)*smax dropped).
void host_pinv(const float *A, int m, int n, float *pA /* [n x m] column-major */);
// DTIwork / ADCwork design matrix (dti.jl:129-140, 66-69); A column-major [nvol x np]
void host_dti_design(const float *bval, const float *bvec, int nvol, int np, float *A);
// GQIwork system matrix (gqi.jl:67-69): A column-major [nvert x nvol]
void host_gqi_matrix(const float *bval, const float *bvec, int nvol, const float *verts, int nverts,
                     float sigma, float *A);
// DSIwork as dense maps (dsi.jl:59-143 + 204-242): A column-major [(nvol+nvert) x nvol];
// returns FIB_OK or FIB_ERR_UNSUPPORTED.  scale_frame/scale_coef: sum(p) = scale_coef*max(s[scale_frame],0).
int host_dsi_matrix(const float *bval, const float *bvec, int nvol, const float *verts, int nverts,
                    int hann_width, float *A, int *scale_frame, float *scale_coef, std::vector<int> *iq_out = nullptr);
// folded-face neighbour table (gqi.jl:63-64 + 185-196): nbr [nvert x maxdeg] row-major, -1 padded
int host_neighbours(const int32_t *faces, int nfaces, int nverts, std::vector<int32_t> &nbr, int *maxdeg);

// O[M x n] = A[M x K] * max(S[K x n], 0) on the contraction kernels of odf.hip (row N4, RUMBA-SD): A column-major
// [nrows x ncols]; S, out planar (row stride n); `ones` = n bytes of 1 on the device; recompact = (re)build the column
// list (needed once per n)
int matrix_plan_create(int device, const float *A, int nrows, int ncols, fib_odf_plan **plan);
int matrix_plan_run(const fib_odf_plan *plan, const float *S, const uint8_t *ones, int64_t n, float *out, bool recompact, void *stream);

}  // namespace fib
