// api.hip — host-buffer ("drop-in") entry points: the functions the Julia wrapper ccalls in place of the bodies of
// dti_fit / adc_fit / gqi_rec / dsi_rec / stream.  Caller-owned host arrays in, caller-owned host arrays out.
//
// The reference threads its volume loops over z-slices (dti.jl:258, gqi.jl:132, dsi.jl:197) and its seed list over
// contiguous chunks (stream.jl:757-761).  Here the same decomposition feeds GPUs:
//   * a device set (fib_init; default: device 0) — the voxel range is cut into contiguous slabs, one per device, each
//     driven by its own host thread; seeds shard round-robin;
//   * per device a three-stage pipeline over voxel chunks: gather rows of the planar host arrays into a pinned ring
//     (threaded memcpy) -> H2D on one stream || the device-resident path (fibd_*) on a second || D2H on a third ->
//     scatter rows back.  PCIe runs in both directions at once; the kernels hide completely behind the link;
//   * the only exchange step of the fits, odfmax = maximum(mean(odf, dims=4)) (gqi.jl:164, dsi.jl:263), is one float per
//     chunk: reduced on the host, then qa ./= odfmax runs on every device before the qa volumes are copied out;
//   * plans (the reference's work structs) are cached per device, keyed by the tables they were built from.
#include <sched.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "common.h"
#include "host_tier.h"

namespace {

using fibh::bind_this_thread; using fibh::chunk_count; using fibh::CopyPool; using fibh::dtype_size; using fibh::LiveMap; using fibh::NBUF; using fibh::Rows;
using fibh::slab;

#define RC(x) do { int _rc = (x); if (_rc != FIB_OK) return _rc; } while (0)

// (mask element types, the copy pool, live maps, the chunk schedule and the chunk pipeline itself: host_tier.h -- pure host code, also built
// under the sanitizers by tests/test_host_sanitizers.py)
int mask_convert_range(const void *m, int dtype, int64_t i0, int64_t n, bool positive, uint8_t *out) {
    if (!fibh::mask_convert_range(m, dtype, i0, n, positive, out)) return fib::fail(FIB_ERR_INVALID, "unknown mask dtype %d", dtype);
    return FIB_OK;
}
int mask_convert(const void *m, int dtype, int64_t n, bool positive, std::vector<uint8_t> &out) {
    out.resize((size_t)n);
    return mask_convert_range(m, dtype, 0, n, positive, out.data());
}

int h2d(void *dst, const void *src, size_t bytes) {
    FIB_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return FIB_OK;
}
int d2h(void *dst, const void *src, size_t bytes) {
    FIB_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return FIB_OK;
}

// ---- NUMA: the host CPUs next to a device ---------------------------------------------------------------------------------------
// [r5] The pinned ring a device's DMA engines read and write, and the threads that fill and drain it, belong on the NUMA node the device
// hangs off (a two-socket box: the driver's run of fib_gqi_rec took 122 ms where a run whose pages happened to sit on the device's
// node took 90).  The node's CPUs come from sysfs (local_cpulist of the device's PCI function), intersected with the CPUs this process
// may use; an empty list (no sysfs, one node, a container without the file) leaves everything where the scheduler puts it.
std::vector<int> device_local_cpus(int device) {
    std::vector<int> cpus;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return cpus;
    for (char *c = bdf; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return cpus;
    char line[4096] = {0};
    const bool ok = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    if (!ok) return cpus;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return cpus;
    for (const char *q = line; *q;) {                        // "0-63,128-191"
        char *e = nullptr;
        const long a = strtol(q, &e, 10);
        if (e == q) break;
        long b = a;
        q = e;
        if (*q == '-') { b = strtol(q + 1, &e, 10); q = e; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (CPU_ISSET((int)c, &allowed)) cpus.push_back((int)c);
        if (*q == ',') q++; else break;
    }
    return cpus;
}
// a host-side phase, into the profile table while fib_profile_enable(1) is active
struct HostTimer {
    const char *name; std::chrono::steady_clock::time_point t0;
    explicit HostTimer(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { if (fib::profiling_on()) fib::profile_add_ms(name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};

// Result arrays the library hands to the caller (fib_tract_free releases them with free()): large ones are 2-MB aligned and advised for
// transparent huge pages -- C4 returns 1.5 GB of points, i.e. 377 000 first-touch faults of 4-KB pages inside the call otherwise.
void *alloc_result(size_t bytes) {
    if (bytes < ((size_t)32 << 20)) return malloc(bytes ? bytes : 1);
    void *p = nullptr;
    if (posix_memalign(&p, (size_t)2 << 20, bytes) != 0) return nullptr;
    (void)madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}

// ---- per-device state of the host tier -----------------------------------------------------------------------------------
struct PinBuf {                                          // grow-only pinned host buffer
    char *p = nullptr; size_t n = 0;
    // cpus: the allocation (and the first touch of its pages) happens on a thread bound to these CPUs -- the device's NUMA node
    int ensure(size_t bytes, int device, const std::vector<int> &cpus) {
        if (bytes <= n && p) return FIB_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipSuccess;
        auto body = [&] {
            bind_this_thread(cpus);
            (void)hipSetDevice(device);
            e = hipHostMalloc((void **)&p, bytes ? bytes : 1, hipHostMallocDefault);
            if (e == hipSuccess) for (size_t o = 0; o < bytes; o += 4096) p[o] = 0;    // (first touch, should the driver have left any page untouched)
        };
        if (cpus.empty()) body(); else { std::thread t(body); t.join(); }
        if (e != hipSuccess) { p = nullptr; return fib::fail(FIB_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
        n = bytes;
        return FIB_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
    ~PinBuf() { release(); }
};

struct CachedPlan { std::string key; void *plan = nullptr; int kind = 0; uint64_t stamp = 0; };   // kind 0: dti/adc, 1: odf

struct DevState {
    int device = 0;
    std::mutex mu;                                       // one host-tier call at a time per entry of the device set
    bool ready = false;
    hipStream_t s_in = nullptr, s_cmp = nullptr, s_out = nullptr;
    hipEvent_t e_in[NBUF] = {}, e_cmp[NBUF] = {}, e_out[NBUF] = {};
    PinBuf pin_in[NBUF], pin_out[NBUF];
    fib::DevBuf<char> dev_in[NBUF], dev_out[NBUF];
    std::vector<int> cpus;                               // the host CPUs on the device's NUMA node (empty: unknown / not bound)
    std::unique_ptr<CopyPool> pool, pool_out;            // row copies into the ring (gather) | out of it (scatter): the two stages run side by side
    std::vector<CachedPlan> plans;
    uint64_t clock = 0;
    fib_stream_ws *ws = nullptr;
    // device buffers of fib_stream, kept between calls (grow-only, like the ring and the tracer's workspace): a hipMalloc / hipFree pair
    // per 1.5-GB result costs more than the tracking itself
    struct StreamBufs {
        fib::DevBuf<float> vec, f, fa, field, sub, lcms, xyz;
        fib::DevBuf<uint8_t> mask, mout, flags;
        fib::DevBuf<int64_t> seeds, sidx;
        fib::DevBuf<int32_t> npts;
    } sb;

    int init(int nthreads) {
        if (ready) return FIB_OK;
        FIB_HIP(hipSetDevice(device));
        FIB_HIP(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
        FIB_HIP(hipStreamCreateWithFlags(&s_cmp, hipStreamNonBlocking));
        FIB_HIP(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
        for (int b = 0; b < NBUF; b++) {
            FIB_HIP(hipEventCreateWithFlags(&e_in[b], hipEventDisableTiming));
            FIB_HIP(hipEventCreateWithFlags(&e_cmp[b], hipEventDisableTiming));
            FIB_HIP(hipEventCreateWithFlags(&e_out[b], hipEventDisableTiming));
        }
        { const char *e = fib::ab_env("FIBERS_HOST_NUMA"); if (!(e && e[0] == '0')) cpus = device_local_cpus(device); }
        pool.reset(new CopyPool(nthreads, cpus));
        pool_out.reset(new CopyPool(nthreads, cpus));
        ready = true;
        return FIB_OK;
    }
    // fib_trim: everything this worker keeps BETWEEN calls to make the next call cheap -- the pinned ring and its device mirror, fib_stream's
    // device buffers (C4: 1.5 GB of points), the tracer's workspace -- goes back to the driver; plans stay (they are small and costly to
    // rebuild).  The caller holds `mu`: no call is in flight on this worker.
    void trim() {
        if (!ready) return;
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(s_in); (void)hipStreamSynchronize(s_cmp); (void)hipStreamSynchronize(s_out);
        for (int b = 0; b < NBUF; b++) { pin_in[b].release(); pin_out[b].release(); dev_in[b].release(); dev_out[b].release(); }
        sb.vec.release(); sb.f.release(); sb.fa.release(); sb.field.release(); sb.sub.release(); sb.lcms.release(); sb.xyz.release();
        sb.mask.release(); sb.mout.release(); sb.flags.release(); sb.seeds.release(); sb.sidx.release(); sb.npts.release();
        if (ws) { fibd_stream_ws_destroy(ws); ws = nullptr; }
    }
    void drop_plans() {
        for (auto &c : plans) {
            if (c.kind == 0) fib_dti_plan_destroy((fib_dti_plan *)c.plan); else fib_odf_plan_destroy((fib_odf_plan *)c.plan);
        }
        plans.clear();
    }
    ~DevState() {
        if (!ready) return;
        (void)hipSetDevice(device);
        drop_plans();
        if (ws) fibd_stream_ws_destroy(ws);
        for (int b = 0; b < NBUF; b++) { (void)hipEventDestroy(e_in[b]); (void)hipEventDestroy(e_cmp[b]); (void)hipEventDestroy(e_out[b]); }
        (void)hipStreamDestroy(s_in); (void)hipStreamDestroy(s_cmp); (void)hipStreamDestroy(s_out);
    }
};

// The device set of the host tier (fib_init).  Entry i is an independent worker: the same device may appear twice (two
// concurrent pipelines on one GPU; used by the tests on a 1-GPU box).
// Workers are handed out as shared pointers: a call that fib_init / fib_shutdown overtakes keeps its workers alive until it returns.
using Worker = std::shared_ptr<DevState>;
struct HostCtx {
    std::mutex mu;
    std::vector<Worker> devs;                            // the set for device == FIB_DEVICE_ALL
    std::vector<Worker> single;                          // workers for calls that name one device
};
// Never destroyed: at process exit the HIP runtime's own exit handlers may already have run, and destroying streams, events and
// device buffers then crashes or hangs.  Only fib_shutdown releases resources.
HostCtx &ctx() { static HostCtx *c = new HostCtx(); return *c; }

int copy_threads(int nworkers) {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 8;
    if (const char *e = fib::env("FIBERS_COPY_THREADS")) { const int t = atoi(e); if (t >= 1) return t; }
    int t = (int)hw / (2 * (nworkers > 0 ? nworkers : 1));
    // 8-16 threads reach the host's copy bandwidth on whole rows (tools/probes/host_probe.hip); the runs of a masked volume are a few hundred
    // bytes each; [r6] 32 or 64 threads per stage were measured and are WORSE, dense and masked (profiles/r06/host_tier_threads.txt)
    return t < 2 ? 2 : (t > 16 ? 16 : t);
}

// workers of a call: the fib_init set for FIB_DEVICE_ALL, else the (lazily created) worker of that device
int workers_for(int device, std::vector<Worker> &out) {
    HostCtx &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (device == FIB_DEVICE_ALL) {
        if (c.devs.empty()) {                            // no fib_init: every visible device
            const int n = fib_device_count();
            if (n <= 0) return fib::fail(FIB_ERR_NO_DEVICE, "no HIP device is available and this back end has no CPU fallback");
            for (int d = 0; d < n; d++) { c.devs.emplace_back(new DevState()); c.devs.back()->device = d; }
        }
        for (auto &d : c.devs) out.push_back(d);
        return FIB_OK;
    }
    RC(fib::use_device(device));
    for (auto &d : c.single) if (d->device == device) { out.push_back(d); return FIB_OK; }
    c.single.emplace_back(new DevState());
    c.single.back()->device = device;
    out.push_back(c.single.back());
    return FIB_OK;
}

// ---- plan cache ----------------------------------------------------------------------------------------------------------------
void key_add(std::string &k, const void *p, size_t bytes) { k.append((const char *)&bytes, sizeof bytes); if (p) k.append((const char *)p, bytes); }

template <typename MakeFn>
int cached_plan(DevState &d, int kind, const std::string &key, MakeFn make, void **plan) {
    for (auto &c : d.plans) if (c.kind == kind && c.key == key) { c.stamp = ++d.clock; *plan = c.plan; return FIB_OK; }
    void *p = nullptr;
    RC(make(&p));
    if (d.plans.size() >= 4) {                           // evict the least recently used
        size_t lru = 0;
        for (size_t i = 1; i < d.plans.size(); i++) if (d.plans[i].stamp < d.plans[lru].stamp) lru = i;
        if (d.plans[lru].kind == 0) fib_dti_plan_destroy((fib_dti_plan *)d.plans[lru].plan); else fib_odf_plan_destroy((fib_odf_plan *)d.plans[lru].plan);
        d.plans.erase(d.plans.begin() + lru);
    }
    d.plans.push_back(CachedPlan{key, p, kind, ++d.clock});
    *plan = p;
    return FIB_OK;
}

// ---- the chunk pipeline (host_tier.h: fibh::run_chunks) on HIP streams ------------------------------------------------------------
// what a fit does with one chunk, all pointers on the device: din = the input rows back to back (row stride n), dmask = n
// bytes, dout = the output rows back to back (row stride n)
using ChunkFn = std::function<int(int chunk, int64_t rel, int64_t n, const float *din, const uint8_t *dmask, float *dout, hipStream_t st)>;

// the slab's LiveMap where packing pays (*use = &lm), NULL where the mask keeps (nearly) everything.  FIBERS_HOST_PACK=0: never.
int live_map_for(DevState &d, const void *mask, int mask_dtype, int64_t v0, int64_t v1, LiveMap &lm, const LiveMap **use) {
    *use = nullptr;
    const char *e = fib::env("FIBERS_HOST_PACK");
    if ((e && e[0] == '0') || v1 <= v0) return FIB_OK;
    if (!fibh::build_live_map(*d.pool, mask, mask_dtype, v0, v1, lm)) return fib::fail(FIB_ERR_INVALID, "unknown mask dtype %d", mask_dtype);
    if (fibh::live_pack_pays(lm, v0, v1)) *use = &lm;
    return FIB_OK;
}
int64_t pick_chunk(int64_t nrange, int rows_in, int rows_out) { return fibh::pick_chunk(nrange, rows_in, rows_out, fib::env("FIBERS_HOST_CHUNK")); }

// the device back end of fibh::run_chunks: the worker's pinned ring, its three streams (upload | kernels | download) and one event per
// ring slot and stage
struct HipDev {
    DevState &d;
    const ChunkFn &fn;
    hipStream_t st(fibh::Stream s) const { return s == fibh::S_IN ? d.s_in : (s == fibh::S_CMP ? d.s_cmp : d.s_out); }
    hipEvent_t ev(fibh::Event e, int b) const { return e == fibh::E_IN ? d.e_in[b] : (e == fibh::E_CMP ? d.e_cmp[b] : d.e_out[b]); }
    int ensure(size_t in_bytes, size_t out_bytes) {
        for (int b = 0; b < NBUF; b++) {
            RC(d.pin_in[b].ensure(in_bytes, d.device, d.cpus));
            RC(d.pin_out[b].ensure(out_bytes, d.device, d.cpus));
            RC(d.dev_in[b].ensure(in_bytes));
            RC(d.dev_out[b].ensure(out_bytes));
        }
        return FIB_OK;
    }
    char *pin_in(int b) { return d.pin_in[b].p; }
    char *pin_out(int b) { return d.pin_out[b].p; }
    int upload(int b, size_t bytes) {
        fib::ProfScope prof("host_h2d", d.s_in);
        FIB_HIP(hipMemcpyAsync(d.dev_in[b].p, d.pin_in[b].p, bytes, hipMemcpyHostToDevice, d.s_in));
        return FIB_OK;
    }
    int compute(int k, int b, int64_t rel, int64_t nd, int rin) {
        return fn(k, rel, nd, reinterpret_cast<const float *>(d.dev_in[b].p), reinterpret_cast<const uint8_t *>(d.dev_in[b].p) + (size_t)rin * nd * 4,
                  reinterpret_cast<float *>(d.dev_out[b].p), d.s_cmp);
    }
    int download(int b, size_t bytes) {
        fib::ProfScope prof("host_d2h", d.s_out);
        FIB_HIP(hipMemcpyAsync(d.pin_out[b].p, d.dev_out[b].p, bytes, hipMemcpyDeviceToHost, d.s_out));
        return FIB_OK;
    }
    int record(fibh::Event e, int b) { FIB_HIP(hipEventRecord(ev(e, b), st((fibh::Stream)e))); return FIB_OK; }      // (event kinds and streams are numbered alike)
    int stream_wait(fibh::Stream s, fibh::Event e, int b) { FIB_HIP(hipStreamWaitEvent(st(s), ev(e, b), 0)); return FIB_OK; }
    int host_wait(fibh::Event e, int b) { (void)hipSetDevice(d.device); return hipEventSynchronize(ev(e, b)) == hipSuccess ? FIB_OK : FIB_ERR_HIP; }
    void drain() { (void)hipStreamSynchronize(d.s_in); (void)hipStreamSynchronize(d.s_cmp); (void)hipStreamSynchronize(d.s_out); }
    void prof(const char *name, double ms) { if (fib::profiling_on()) fib::profile_add_ms(name, ms); }
    int fail(int code, const char *msg) { return fib::fail(code, "%s", msg); }
    std::string last_error() { return fib_last_error(); }
    void set_error(const std::string &m) { fib::set_error("%s", m.c_str()); }
};

// voxels [vbeg, vend) of a volume of nvox voxels through device d.  Blocking.  The caller holds d.mu.  (lm, outputs_zeroed: fibh::run_chunks)
int run_chunks(DevState &d, int64_t vbeg, int64_t vend, int64_t nvox, const std::vector<Rows> &ins, const void *mask, int mask_dtype,
               const std::vector<Rows> &outs, int64_t chunk, const ChunkFn &fn, const LiveMap *lm = nullptr, bool outputs_zeroed = false) {
    if (vend <= vbeg) return FIB_OK;
    FIB_HIP(hipSetDevice(d.device));
    HipDev dev{d, fn};
    const char *ent = fib::env("FIBERS_HOST_NT");        // streaming stores in the row copies: on unless FIBERS_HOST_NT=0
    const bool nt = !(ent && ent[0] == '0');
    return fibh::run_chunks(dev, *d.pool, *d.pool_out, vbeg, vend, nvox, ins, mask, mask_dtype, outs, chunk, lm, outputs_zeroed, nt);
}

// runs job(worker index, worker) on every worker of the set, one host thread each; the first error wins
int for_each_worker(const std::vector<Worker> &ws, const std::function<int(int, DevState &)> &job) {
    std::vector<int> rcs(ws.size(), FIB_OK);
    std::vector<std::string> msgs(ws.size());
    auto body = [&](int i) {
        try {
            std::lock_guard<std::mutex> lk(ws[i]->mu);
            fib::DeviceGuard guard;
            int rc = ws[i]->init(copy_threads((int)ws.size()));
            if (rc == FIB_OK) rc = job(i, *ws[i]);
            rcs[i] = rc;
            if (rc != FIB_OK) msgs[i] = fib_last_error();       // (thread-local message of the worker thread)
        } catch (const std::bad_alloc &) { rcs[i] = FIB_ERR_NOMEM; msgs[i] = "out of host memory"; }
        catch (...) { rcs[i] = FIB_ERR_INVALID; msgs[i] = "internal error in a device worker"; }
    };
    if (ws.size() == 1) body(0);
    else {
        std::vector<std::thread> th;
        for (size_t i = 0; i < ws.size(); i++) th.emplace_back(body, (int)i);
        for (auto &t : th) t.join();
    }
    for (size_t i = 0; i < ws.size(); i++) if (rcs[i] != FIB_OK) return fib::fail(rcs[i], "%s", msgs[i].c_str());
    return FIB_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------------
// device set
// ------------------------------------------------------------------------------------------------------------------------------
extern "C" int fib_init(int ndev, const int *devs) try {
    FIB_CHECK(ndev >= 0 && (ndev == 0 || devs), FIB_ERR_INVALID, "invalid device list");
    const int have = fib_device_count();
    FIB_CHECK(have > 0, FIB_ERR_NO_DEVICE, "no HIP device is available and this back end has no CPU fallback");
    for (int i = 0; i < ndev; i++) FIB_CHECK(devs[i] >= 0 && devs[i] < have, FIB_ERR_NO_DEVICE, "device %d is not available (%d devices)", devs[i], have);
    HostCtx &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    fib::DeviceGuard guard;
    for (auto &d : c.devs) { std::lock_guard<std::mutex> lk2(d->mu); }   // wait for calls in flight
    c.devs.clear();
    const int n = ndev > 0 ? ndev : have;
    for (int i = 0; i < n; i++) { c.devs.emplace_back(new DevState()); c.devs.back()->device = ndev > 0 ? devs[i] : i; }
    return FIB_OK;
} FIB_API_CATCH

extern "C" int fib_trim(void) try {
    HostCtx &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    fib::DeviceGuard guard;
    for (auto *set : {&c.devs, &c.single})
        for (auto &d : *set) { std::lock_guard<std::mutex> lk2(d->mu); d->trim(); }     // (waits for a call in flight on that worker)
    return FIB_OK;
} FIB_API_CATCH

extern "C" void fib_shutdown(void) try {
    HostCtx &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    fib::DeviceGuard guard;
    for (auto &d : c.devs) { std::lock_guard<std::mutex> lk2(d->mu); }     // wait for calls in flight (a call that has its workers but
    for (auto &d : c.single) { std::lock_guard<std::mutex> lk2(d->mu); }   // not yet their locks keeps them alive through its shared pointers)
    c.devs.clear();
    c.single.clear();
} FIB_API_CATCH_VOID

// ------------------------------------------------------------------------------------------------------------------------------
// dti_fit / adc_fit
// ------------------------------------------------------------------------------------------------------------------------------
namespace {
int dti_plan_for(DevState &d, const float *bval, const float *bvec, int nvol, fib_dti_plan **plan) {
    std::string key;
    key_add(key, bval, sizeof(float) * nvol);
    key_add(key, bvec, bvec ? sizeof(float) * 3 * nvol : 0);
    void *p = nullptr;
    RC(cached_plan(d, 0, key, [&](void **out) { fib_dti_plan *q = nullptr; int rc = fib_dti_plan_create(d.device, bval, bvec, nvol, &q); *out = q; return rc; }, &p));
    *plan = (fib_dti_plan *)p;
    return FIB_OK;
}
}  // namespace

extern "C" int fib_dti_fit(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                           const void *mask, int mask_dtype, const float *bval, const float *bvec,
                           const fib_dti_out *out) try {
    FIB_CHECK(bval != nullptr && nvol > 0, FIB_ERR_MISSING_BVAL, "Missing b-value table from input DWI structure");
    FIB_CHECK(bvec != nullptr, FIB_ERR_MISSING_BVEC, "Missing gradient table from input DWI structure");
    FIB_CHECK(dwi && mask && out, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nx > 0 && ny > 0 && nz > 0, FIB_ERR_INVALID, "volume dimensions must be positive");
    const bool zeroed = (mask_dtype & FIB_MASK_OUTPUTS_ZEROED) != 0;
    mask_dtype &= ~FIB_MASK_OUTPUTS_ZEROED;
    FIB_CHECK(dtype_size(mask_dtype) > 0, FIB_ERR_INVALID, "unknown mask dtype %d", mask_dtype);
    FIB_CHECK(out->s0 && out->eigval1 && out->eigval2 && out->eigval3 && out->eigvec1 && out->eigvec2 && out->eigvec3 && out->rd && out->md && out->fa,
              FIB_ERR_INVALID, "NULL output volume");
    const int64_t nvox = (int64_t)nx * ny * nz;
    std::vector<Worker> ws;
    RC(workers_for(device, ws));
    const std::vector<Rows> ins = {{dwi, nullptr, nvol}};
    const std::vector<Rows> outs = {{nullptr, out->s0, 1}, {nullptr, out->eigval1, 1}, {nullptr, out->eigval2, 1}, {nullptr, out->eigval3, 1},
                                    {nullptr, out->eigvec1, 3}, {nullptr, out->eigvec2, 3}, {nullptr, out->eigvec3, 3},
                                    {nullptr, out->rd, 1}, {nullptr, out->md, 1}, {nullptr, out->fa, 1}};
    return for_each_worker(ws, [&](int i, DevState &d) -> int {
        int64_t v0, v1;
        slab(nvox, (int)ws.size(), i, v0, v1);
        fib_dti_plan *plan = nullptr;
        RC(dti_plan_for(d, bval, bvec, nvol, &plan));
        LiveMap lm;
        const LiveMap *use = nullptr;
        RC(live_map_for(d, mask, mask_dtype, v0, v1, lm, &use));
        return run_chunks(d, v0, v1, nvox, ins, mask, mask_dtype, outs, pick_chunk(use ? use->nlive : v1 - v0, nvol, 16),
                          [&](int, int64_t, int64_t n, const float *din, const uint8_t *dm, float *b, hipStream_t st) -> int {
                              fib_dti_out dev{b, b + n, b + 2 * n, b + 3 * n, b + 4 * n, b + 7 * n, b + 10 * n, b + 13 * n, b + 14 * n, b + 15 * n};
                              return fibd_dti_fit(plan, din, dm, n, &dev, st);
                          }, use, zeroed);
    });
} FIB_API_CATCH

extern "C" int fib_adc_fit(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                           const void *mask, int mask_dtype, const float *bval, float *adc, float *s0) try {
    FIB_CHECK(bval != nullptr && nvol > 0, FIB_ERR_MISSING_BVAL, "Missing b-value table from input DWI structure");
    FIB_CHECK(dwi && mask && adc && s0, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nx > 0 && ny > 0 && nz > 0, FIB_ERR_INVALID, "volume dimensions must be positive");
    const bool zeroed = (mask_dtype & FIB_MASK_OUTPUTS_ZEROED) != 0;
    mask_dtype &= ~FIB_MASK_OUTPUTS_ZEROED;
    FIB_CHECK(dtype_size(mask_dtype) > 0, FIB_ERR_INVALID, "unknown mask dtype %d", mask_dtype);
    const int64_t nvox = (int64_t)nx * ny * nz;
    std::vector<Worker> ws;
    RC(workers_for(device, ws));
    const std::vector<Rows> ins = {{dwi, nullptr, nvol}};
    const std::vector<Rows> outs = {{nullptr, adc, 1}, {nullptr, s0, 1}};
    return for_each_worker(ws, [&](int i, DevState &d) -> int {
        int64_t v0, v1;
        slab(nvox, (int)ws.size(), i, v0, v1);
        fib_dti_plan *plan = nullptr;
        RC(dti_plan_for(d, bval, nullptr, nvol, &plan));
        LiveMap lm;
        const LiveMap *use = nullptr;
        RC(live_map_for(d, mask, mask_dtype, v0, v1, lm, &use));
        return run_chunks(d, v0, v1, nvox, ins, mask, mask_dtype, outs, pick_chunk(use ? use->nlive : v1 - v0, nvol, 2),
                          [&](int, int64_t, int64_t n, const float *din, const uint8_t *dm, float *b, hipStream_t st) -> int {
                              return fibd_adc_fit(plan, din, dm, n, b, b + n, st);
                          }, use, zeroed);
    });
} FIB_API_CATCH

extern "C" int fib_st_eigen(int device, const float *const S[6], int64_t nvox, float *eigvec, float *eigval) try {
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
} FIB_API_CATCH

// ------------------------------------------------------------------------------------------------------------------------------
// gqi_rec / dsi_rec
// ------------------------------------------------------------------------------------------------------------------------------
namespace {

struct OdfSpec {                                        // what identifies a GQIwork / DSIwork
    bool dsi; const float *bval, *bvec; int nvol; const float *verts; int nverts; const int32_t *faces; int nfaces; float sigma; int hann_width;
};
int odf_plan_for(DevState &d, const OdfSpec &s, fib_odf_plan **plan) {
    std::string key;
    key.push_back(s.dsi ? 'D' : 'G');
    key_add(key, s.bval, sizeof(float) * s.nvol);
    key_add(key, s.bvec, sizeof(float) * 3 * s.nvol);
    key_add(key, s.verts, sizeof(float) * 3 * s.nverts);
    key_add(key, s.faces, sizeof(int32_t) * 3 * s.nfaces);
    key_add(key, s.dsi ? (const void *)&s.hann_width : (const void *)&s.sigma, 4);
    const int fmt = fib_odf_default_format();            // the operand format is part of a plan's identity (it may change with the environment)
    key_add(key, &fmt, sizeof(fmt));
    void *p = nullptr;
    RC(cached_plan(d, 1, key, [&](void **out) {
        fib_odf_plan *q = nullptr;
        const int rc = s.dsi ? fib_dsi_plan_create_fmt(d.device, s.bval, s.bvec, s.nvol, s.verts, s.nverts, s.faces, s.nfaces, s.hann_width, fmt, &q)
                             : fib_gqi_plan_create_fmt(d.device, s.bval, s.bvec, s.nvol, s.verts, s.nverts, s.faces, s.nfaces, s.sigma, fmt, &q);
        *out = q;
        return rc;
    }, &p));
    *plan = (fib_odf_plan *)p;
    return FIB_OK;
}

int odf_rec_host(int device, const OdfSpec &spec, const float *dwi, int nx, int ny, int nz, const void *mask, int mask_dtype,
                 float *pdf, float *odf, float *const peak[3], float *const qa[3]) {
    FIB_CHECK(spec.bval != nullptr && spec.nvol > 0, FIB_ERR_MISSING_BVAL, "Missing b-value table from input DWI structure");
    FIB_CHECK(spec.bvec != nullptr, FIB_ERR_MISSING_BVEC, "Missing gradient table from input DWI structure");
    FIB_CHECK(spec.verts && spec.faces && spec.nverts >= 2 && spec.nverts % 2 == 0 && spec.nfaces > 0, FIB_ERR_INVALID, "invalid ODF tessellation");
    FIB_CHECK(dwi && mask && odf && peak && qa, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nx > 0 && ny > 0 && nz > 0, FIB_ERR_INVALID, "volume dimensions must be positive");
    const bool zeroed = (mask_dtype & FIB_MASK_OUTPUTS_ZEROED) != 0;   // (the qa planes below are written in full either way: their outside value can be NaN)
    mask_dtype &= ~FIB_MASK_OUTPUTS_ZEROED;
    FIB_CHECK(dtype_size(mask_dtype) > 0, FIB_ERR_INVALID, "unknown mask dtype %d", mask_dtype);
    for (int k = 0; k < 3; k++) FIB_CHECK(peak[k] && qa[k], FIB_ERR_INVALID, "NULL peak/qa output volume");
    const int64_t nvox = (int64_t)nx * ny * nz;
    const int nvol = spec.nvol, nvert = spec.nverts / 2;
    std::vector<Worker> ws;
    RC(workers_for(device, ws));
    const int nw = (int)ws.size();
    const std::vector<Rows> ins = {{dwi, nullptr, nvol}};
    std::vector<Rows> outs;
    if (pdf) outs.push_back({nullptr, pdf, nvol});
    outs.push_back({nullptr, odf, nvert});
    for (int k = 0; k < 3; k++) outs.push_back({nullptr, peak[k], 3});
    const int rows_out = (pdf ? nvol : 0) + nvert + 9;
    // per worker: qa of its slab stays on the device until the global odfmax is known; one {max, NaN flag} pair per chunk
    // (the buffer belongs to the CALL, not to the worker: the worker's lock is released between the two passes below, and another
    // thread's call on the same worker must not find -- or reallocate -- this call's qa)
    struct Slab { int64_t v0 = 0, v1 = 0, nq = 0; float *qa = nullptr; std::vector<float> maxes; fib::DevBuf<float> keep; LiveMap lm; const LiveMap *use = nullptr; };
    std::vector<Slab> slabs((size_t)nw);
    RC(for_each_worker(ws, [&](int i, DevState &d) -> int {
        Slab &sl = slabs[i];
        slab(nvox, nw, i, sl.v0, sl.v1);
        const int64_t nr = sl.v1 - sl.v0;
        if (nr <= 0) return FIB_OK;
        fib_odf_plan *plan = nullptr;
        RC(odf_plan_for(d, spec, &plan));
        RC(live_map_for(d, mask, mask_dtype, sl.v0, sl.v1, sl.lm, &sl.use));
        const int64_t ntr = sl.use ? sl.use->nlive : nr;        // voxels that travel
        const int64_t chunk = pick_chunk(ntr, nvol, rows_out);
        const int nchunks = chunk_count(ntr, chunk);
        sl.nq = (ntr + 31) / 32 * 32 + 32;                      // a plane of the qa kept on the device (a packed chunk is padded to 32 voxels)
        FIB_HIP(hipSetDevice(d.device));
        RC(sl.keep.alloc((size_t)3 * sl.nq + (size_t)2 * std::max(nchunks, 1)));
        sl.qa = sl.keep.p;
        float *dmax = sl.qa + 3 * sl.nq;
        const int64_t nq = sl.nq;
        RC(run_chunks(d, sl.v0, sl.v1, nvox, ins, mask, mask_dtype, outs, chunk,
                      [&](int k, int64_t rel, int64_t n, const float *din, const uint8_t *dm, float *b, hipStream_t st) -> int {
                          float *dpdf = pdf ? b : nullptr, *dodf = b + (size_t)(pdf ? nvol : 0) * n, *dpk = dodf + (size_t)nvert * n;
                          float *pk[3] = {dpk, dpk + 3 * n, dpk + 6 * n};
                          float *q[3] = {sl.qa + rel, sl.qa + nq + rel, sl.qa + 2 * nq + rel};
                          // (one volume in pieces: the same peak-finder form for every piece, whatever the cut -- see FIB_ODF_SEPARATE_PEAKS)
                          return fibd_odf_rec(plan, din, dm, n, dpdf, dodf, pk, q, dmax + 2 * k, nvox % 4 != 0 ? FIB_ODF_SEPARATE_PEAKS : 0, st);
                      }, sl.use, zeroed));
        sl.maxes.resize((size_t)2 * nchunks);
        if (nchunks > 0) FIB_HIP(hipMemcpy(sl.maxes.data(), dmax, sl.maxes.size() * sizeof(float), hipMemcpyDeviceToHost));
        return FIB_OK;
    }));
    // odfmax = maximum(mean(odf, dims=4)) over the whole volume (gqi.jl:164, dsi.jl:263); maximum() propagates NaN
    float odfmax = -INFINITY;
    bool anynan = false;
    for (auto &sl : slabs)
        for (size_t c = 0; c + 1 < sl.maxes.size(); c += 2) {
            if (sl.maxes[c + 1] != 0.0f || sl.maxes[c] != sl.maxes[c]) anynan = true;
            else if (sl.maxes[c] > odfmax) odfmax = sl.maxes[c];
        }
    // (a packed slab leaves voxels out: their ODF is 0 and so is their mean -- the unpacked pipeline's kernels count it the same way)
    for (auto &sl : slabs) if (sl.use && sl.use->nlive < sl.v1 - sl.v0 && 0.0f > odfmax) odfmax = 0.0f;
    if (anynan) odfmax = __builtin_nanf("");
    const float qa_outside = 0.0f / odfmax;              // qa of a voxel outside the mask after `qa ./= odfmax`: 0, or NaN when odfmax is 0 / NaN
    // qa[k] ./= odfmax (gqi.jl:166-168) on every device, then out
    return for_each_worker(ws, [&](int i, DevState &d) -> int {
        Slab &sl = slabs[i];
        const int64_t nr = sl.v1 - sl.v0;
        if (nr <= 0) return FIB_OK;
        FIB_HIP(hipSetDevice(d.device));
        float *q[3] = {sl.qa, sl.qa + sl.nq, sl.qa + 2 * sl.nq};
        const int64_t ntr = sl.use ? sl.use->nlive : nr;
        if (ntr > 0) RC(fibd_qa_normalize(q, ntr, odfmax, d.s_cmp));
        if (!sl.use) {
            for (int k = 0; k < 3; k++) FIB_HIP(hipMemcpyAsync(qa[k] + sl.v0, q[k], (size_t)nr * sizeof(float), hipMemcpyDeviceToHost, d.s_cmp));
            FIB_HIP(hipStreamSynchronize(d.s_cmp));
            return FIB_OK;
        }
        // packed: the three planes come back dense and go to their runs; the gaps read 0
        std::vector<float> hq((size_t)3 * std::max<int64_t>(ntr, 1));
        for (int k = 0; k < 3 && ntr > 0; k++) FIB_HIP(hipMemcpyAsync(hq.data() + (size_t)k * ntr, q[k], (size_t)ntr * sizeof(float), hipMemcpyDeviceToHost, d.s_cmp));
        FIB_HIP(hipStreamSynchronize(d.s_cmp));
        const LiveMap &lm = *sl.use;
        d.pool->run(3, [&](int k) {
            float *row = qa[k];
            const float *src = hq.data() + (size_t)k * ntr;
            int64_t g0 = lm.vbeg;
            for (size_t r = 0; r < lm.start.size(); r++) {
                std::fill(row + g0, row + lm.start[r], qa_outside);
                memcpy(row + lm.start[r], src + lm.off[r], (size_t)lm.len[r] * 4);
                g0 = lm.start[r] + lm.len[r];
            }
            std::fill(row + g0, row + lm.vend, qa_outside);
        });
        return FIB_OK;
    });
}

}  // namespace

extern "C" int fib_gqi_rec(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                           const void *mask, int mask_dtype, const float *bval, const float *bvec,
                           const float *verts, int nverts, const int32_t *faces, int nfaces, float sigma,
                           float *odf, float *const peak[3], float *const qa[3]) try {
    const OdfSpec spec{false, bval, bvec, nvol, verts, nverts, faces, nfaces, sigma, 0};
    return odf_rec_host(device, spec, dwi, nx, ny, nz, mask, mask_dtype, nullptr, odf, peak, qa);
} FIB_API_CATCH

extern "C" int fib_dsi_rec(int device, const float *dwi, int nx, int ny, int nz, int nvol,
                           const void *mask, int mask_dtype, const float *bval, const float *bvec,
                           const float *verts, int nverts, const int32_t *faces, int nfaces, int hann_width,
                           float *pdf, float *odf, float *const peak[3], float *const qa[3]) try {
    FIB_CHECK(pdf != nullptr, FIB_ERR_INVALID, "NULL pdf output volume");
    const OdfSpec spec{true, bval, bvec, nvol, verts, nverts, faces, nfaces, 0.0f, hann_width};
    return odf_rec_host(device, spec, dwi, nx, ny, nz, mask, mask_dtype, pdf, odf, peak, qa);
} FIB_API_CATCH

// rumba_rec (rusd.jl:419-636), host buffers.  A whole-volume fixed-point iteration: not chunked.
extern "C" int fib_rumba_rec(int device, const float *dwi, int nx, int ny, int nz, int nvol, const void *mask, int mask_dtype,
                             const float *bval, const float *bvec, const float *verts, int nverts, int niter,
                             float lam_para, float lam_perp, float lam_csf, float lam_gm, int ncoils, int sos_grappa, int ipat_factor,
                             int use_tv, const fib_rumba_out *out, float *snr_mean, float *snr_std) try {
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
} FIB_API_CATCH

// find_peaks!(W) (gqi.jl:180-201) for nvox ODFs held in host memory: odf [nvox x nvert] planar (vertex-major rows of
// nvox values, like MRI.vol[:,:,:,v]); isort_top [3 x nvox] planar, 0-based first-half vertex rows, -1 where the
// tessellation has fewer vertices; nvalid [nvox] = count(odf_peak .> 0) (gqi.jl:200).
extern "C" int fib_find_peaks(int device, const float *odf, int64_t nvox, const float *verts, int nverts,
                              const int32_t *faces, int nfaces, int32_t *isort_top, int32_t *nvalid) try {
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
} FIB_API_CATCH

extern "C" int fib_find_peaks_work(int device, const float *odf, int64_t nvox, const float *verts, int nverts,
                                   const int32_t *faces, int nfaces, float *odf_peak, int32_t *isort, int32_t *nvalid) try {
    FIB_CHECK(odf && verts && faces && odf_peak && isort && nvalid, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(nvox > 0 && nverts >= 2 && nverts % 2 == 0 && nfaces > 0, FIB_ERR_INVALID, "invalid sizes");
    fib::DeviceGuard guard;
    const float bval1[1] = {1000.0f}, bvec1[3] = {1.0f, 0.0f, 0.0f};      // (the plan only contributes the folded neighbour table)
    fib_odf_plan *p = nullptr;
    RC(fib_gqi_plan_create(device, bval1, bvec1, 1, verts, nverts, faces, nfaces, 1.25f, &p));
    struct PlanDel { fib_odf_plan *p; ~PlanDel() { fib_odf_plan_destroy(p); } } del{p};
    FIB_HIP(hipSetDevice(device));
    const size_t n = (size_t)nvox * (nverts / 2);
    fib::DevBuf<float> d_odf, d_pk;
    fib::DevBuf<int32_t> d_is, d_nv;
    RC(d_odf.alloc(n)); RC(d_pk.alloc(n)); RC(d_is.alloc(n)); RC(d_nv.alloc((size_t)nvox));
    FIB_HIP(hipMemcpy(d_odf.p, odf, n * sizeof(float), hipMemcpyHostToDevice));
    RC(fibd_find_peaks_work(p, d_odf.p, nvox, d_pk.p, d_is.p, d_nv.p, nullptr));
    FIB_HIP(hipDeviceSynchronize());
    FIB_HIP(hipMemcpy(odf_peak, d_pk.p, n * sizeof(float), hipMemcpyDeviceToHost));
    FIB_HIP(hipMemcpy(isort, d_is.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    FIB_HIP(hipMemcpy(nvalid, d_nv.p, (size_t)nvox * sizeof(int32_t), hipMemcpyDeviceToHost));
    return FIB_OK;
} FIB_API_CATCH

// ------------------------------------------------------------------------------------------------------------------------------
// stream
// ------------------------------------------------------------------------------------------------------------------------------
extern "C" void fib_tract_free(fib_tract_out *out) try {
    if (!out) return;
    free(out->npts); free(out->seed_index); free(out->xyz); free(out->flags);
    out->npts = nullptr; out->seed_index = nullptr; out->xyz = nullptr; out->flags = nullptr;
    out->nlines = 0; out->npoints = 0;
} FIB_API_CATCH_VOID

namespace {

struct StreamIn {
    const fib_stream_params *prm; const float *const *ovec; const float *const *f; float f_thresh; const float *fa; float fa_thresh;
    const float *sublist; int32_t nsub; const float *lcms; float lcm_thresh; uint64_t rng_seed; int strd0, strd1;
};
// what one worker traced: lines in (seed, sub) order of ITS seed shard; seed_index counts seeds of the whole list.  The arrays are
// malloc'ed and never zero-filled ([r5]: C4 returns 1.5 GB of points -- a std::vector's value-initialisation and a second copy into the
// result cost more than the PCIe transfer); with one worker they ARE the result (fib_tract_free releases them with free()).
struct Shard {
    int64_t nl = 0, np = 0;
    int32_t *npts = nullptr; int64_t *sidx = nullptr; float *xyz = nullptr; uint8_t *flags = nullptr;
    Shard() = default;
    Shard(const Shard &) = delete;
    Shard &operator=(const Shard &) = delete;
    ~Shard() { free(npts); free(sidx); free(xyz); free(flags); }
};

// device -> pageable host memory, pipelined: pieces of up to 64 MB come down into two slots of the worker's pinned output ring while the
// scatter pool copies the previous piece to its destination (a plain hipMemcpy into pageable memory stages through a small driver buffer
// at a third of the link's rate, and the first touch of a freshly malloc'ed destination falls on one thread)
int d2h_pipelined(DevState &d, void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return FIB_OK;
    const size_t piece = std::min<size_t>(bytes, (size_t)64 << 20);
    for (int b = 0; b < 2; b++) RC(d.pin_out[b].ensure(piece, d.device, d.cpus));
    const size_t np = (bytes + piece - 1) / piece;
    const int nt = 16;
    auto drain = [&](size_t i) -> int {                  // piece i: ring -> destination
        const int b = (int)(i & 1);
        FIB_HIP(hipEventSynchronize(d.e_out[b]));
        const size_t o = i * piece, n = std::min(piece, bytes - o);
        const size_t part = ((n + nt - 1) / nt + 63) & ~(size_t)63;
        d.pool_out->run(nt, [&](int t) {
            const size_t a = (size_t)t * part;
            if (a < n) memcpy((char *)dst + o + a, d.pin_out[b].p + a, std::min(part, n - a));
        });
        return FIB_OK;
    };
    for (size_t i = 0; i < np; i++) {
        const int b = (int)(i & 1);
        const size_t o = i * piece, n = std::min(piece, bytes - o);
        {
            fib::ProfScope prof("host_d2h", st);
            FIB_HIP(hipMemcpyAsync(d.pin_out[b].p, (const char *)src + o, n, hipMemcpyDeviceToHost, st));
        }
        FIB_HIP(hipEventRecord(d.e_out[b], st));
        if (i > 0) RC(drain(i - 1));
    }
    return drain(np - 1);
}

// seeds[i], i = w, w + nw, ... (round-robin: balances line length; the reference's contiguous chunks, stream.jl:757-759, do not)
int stream_worker(DevState &d, int w, int nw, const StreamIn &in, const std::vector<uint8_t> *m8, const std::vector<int64_t> &seeds,
                  std::vector<uint8_t> *seed_mask_out, Shard &sh) {
    const fib_stream_params &prm0 = *in.prm;
    const int nvec = prm0.nvec;
    const int64_t nvox = (int64_t)prm0.nx * prm0.ny * prm0.nz;
    FIB_HIP(hipSetDevice(d.device));
    if (!d.ws) RC(fibd_stream_ws_create(d.device, &d.ws));
    hipStream_t st = d.s_cmp;
    std::unique_ptr<HostTimer> tm(new HostTimer("stream_host_upload_field"));
    auto &d_vec = d.sb.vec; auto &d_f = d.sb.f; auto &d_fa = d.sb.fa; auto &d_field = d.sb.field; auto &d_sub = d.sb.sub; auto &d_lcms = d.sb.lcms;
    auto &d_mask = d.sb.mask; auto &d_mout = d.sb.mout;
    RC(d_vec.ensure((size_t)nvox * 3 * nvec));
    RC(d_field.ensure((size_t)nvox * 4 * nvec));
    RC(d_mout.ensure((size_t)nvox));
    const float *dv[8] = {}, *df[8] = {};
    for (int k = 0; k < nvec; k++) {
        RC(h2d(d_vec.p + (size_t)k * nvox * 3, in.ovec[k], sizeof(float) * nvox * 3));
        dv[k] = d_vec.p + (size_t)k * nvox * 3;
    }
    if (in.f) {
        RC(d_f.ensure((size_t)nvox * nvec));
        for (int k = 0; k < nvec; k++) { RC(h2d(d_f.p + (size_t)k * nvox, in.f[k], sizeof(float) * nvox)); df[k] = d_f.p + (size_t)k * nvox; }
    }
    if (in.fa) { RC(d_fa.ensure((size_t)nvox)); RC(h2d(d_fa.p, in.fa, sizeof(float) * nvox)); }
    if (m8) { RC(d_mask.ensure((size_t)nvox)); RC(h2d(d_mask.p, m8->data(), (size_t)nvox)); }
    RC(fibd_stream_field(nvec, nvox, dv, in.f ? df : nullptr, in.f_thresh, in.fa ? d_fa.p : nullptr, in.fa_thresh,
                         m8 ? d_mask.p : nullptr, d_field.p, d_mout.p, st));
    if (seed_mask_out) {                                 // first pass (no seed volume): the tracking mask back to the host
        seed_mask_out->resize((size_t)nvox);
        FIB_HIP(hipStreamSynchronize(st));
        RC(d2h(seed_mask_out->data(), d_mout.p, (size_t)nvox));
        return FIB_OK;
    }
    tm.reset(new HostTimer("stream_host_seeds_trace"));
    std::vector<int64_t> mine;
    for (size_t i = (size_t)w; i < seeds.size(); i += (size_t)nw) mine.push_back(seeds[i]);
    auto &d_seeds = d.sb.seeds;
    RC(d_seeds.ensure(mine.size()));
    if (!mine.empty()) RC(h2d(d_seeds.p, mine.data(), sizeof(int64_t) * mine.size()));
    RC(d_sub.ensure((size_t)in.nsub * 3));
    RC(h2d(d_sub.p, in.sublist, sizeof(float) * 3 * in.nsub));
    fib_stream_params prm = prm0;
    prm.ws = d.ws;
    fib_stream_job *job = nullptr;
    int64_t nl = 0, np = 0;
    if (in.lcms) {
        // the uniforms of a line are a function of its index in the WHOLE list (the header's random-number contract), which a
        // shard of a round-robin split cannot express -> LCM runs use one worker (stream_host)
        RC(d_lcms.ensure((size_t)nvox * 10));
        RC(h2d(d_lcms.p, in.lcms, sizeof(float) * nvox * 10));
        RC(fibd_stream_trace_lcm(&prm, d_field.p, d_lcms.p, in.lcm_thresh, in.strd0, in.strd1, in.rng_seed, d_seeds.p, (int64_t)mine.size(),
                                 d_sub.p, in.nsub, st, &job, &nl, &np));
    } else {
        RC(fibd_stream_trace(&prm, d_field.p, d_seeds.p, (int64_t)mine.size(), d_sub.p, in.nsub, st, &job, &nl, &np));
    }
    struct JobGuard { fib_stream_job *j; ~JobGuard() { fib_stream_job_destroy(j); } } jg{job};
    tm.reset(new HostTimer("stream_host_results"));
    sh.nl = nl; sh.np = np;
    sh.npts = (int32_t *)alloc_result(sizeof(int32_t) * (size_t)std::max<int64_t>(nl, 1));
    sh.sidx = (int64_t *)alloc_result(sizeof(int64_t) * (size_t)std::max<int64_t>(nl, 1));
    sh.xyz = (float *)alloc_result(sizeof(float) * 3 * (size_t)std::max<int64_t>(np, 1));
    if (in.lcms) sh.flags = (uint8_t *)alloc_result((size_t)std::max<int64_t>(np, 1));
    if (!sh.npts || !sh.sidx || !sh.xyz || (in.lcms && !sh.flags)) return fib::fail(FIB_ERR_NOMEM, "out of host memory");
    if (nl > 0) {
        auto &d_npts = d.sb.npts; auto &d_sidx = d.sb.sidx; auto &d_xyz = d.sb.xyz; auto &d_flags = d.sb.flags;
        RC(d_npts.ensure((size_t)nl));
        RC(d_sidx.ensure((size_t)nl));
        RC(d_xyz.ensure((size_t)np * 3));
        if (in.lcms) RC(d_flags.ensure((size_t)np));
        RC(fibd_stream_pack_flags(job, d_npts.p, d_sidx.p, d_xyz.p, in.lcms ? d_flags.p : nullptr, st));
        RC(d2h_pipelined(d, sh.xyz, d_xyz.p, sizeof(float) * 3 * (size_t)np, st));     // (stream order: behind the pack kernels)
        RC(d2h_pipelined(d, sh.npts, d_npts.p, sizeof(int32_t) * (size_t)nl, st));
        RC(d2h_pipelined(d, sh.sidx, d_sidx.p, sizeof(int64_t) * (size_t)nl, st));
        if (in.lcms) RC(d2h_pipelined(d, sh.flags, d_flags.p, (size_t)np, st));
        // seed_index = local seed * nsub + sub  ->  global: seed (w + nw * local) of the whole list
        if (nw > 1) for (int64_t i = 0; i < nl; i++) { const int64_t s = sh.sidx[i], ls = s / in.nsub, sub = s % in.nsub; sh.sidx[i] = ((int64_t)w + (int64_t)nw * ls) * in.nsub + sub; }
    }
    return FIB_OK;
}

int stream_host(int device, const fib_stream_params *prm, const float *const *ovec, const float *const *f,
                float f_thresh, const float *fa, float fa_thresh, const void *mask, int mask_dtype,
                const void *seed, int seed_dtype, const float *sublist, int32_t nsub,
                const float *lcms, float lcm_thresh, uint64_t rng_seed, fib_tract_out *out) {
    FIB_CHECK(prm && ovec && sublist && out, FIB_ERR_INVALID, "NULL argument");
    FIB_CHECK(prm->nx > 0 && prm->ny > 0 && prm->nz > 0, FIB_ERR_INVALID, "volume dimensions must be positive");
    FIB_CHECK(prm->nvec >= 1 && prm->nvec <= 8, FIB_ERR_UNSUPPORTED, "1..8 orientation vectors per voxel are supported");
    FIB_CHECK(nsub >= 1, FIB_ERR_INVALID, "sublist must hold at least one offset");
    memset(out, 0, sizeof *out);
    const int nvec = prm->nvec;
    const int64_t nvox = (int64_t)prm->nx * prm->ny * prm->nz;
    for (int k = 0; k < nvec; k++) {
        FIB_CHECK(ovec[k] != nullptr, FIB_ERR_INVALID, "NULL orientation volume %d", k);
        if (f) FIB_CHECK(f[k] != nullptr, FIB_ERR_INVALID, "NULL amplitude volume %d", k);
    }
    std::vector<Worker> ws;
    RC(workers_for(device, ws));
    if (lcms && ws.size() > 1) ws.resize(1);             // (see stream_worker)
    StreamIn in{prm, ovec, f, f_thresh, fa, fa_thresh, sublist, nsub, lcms, lcm_thresh, rng_seed, 0, 1};
    if (lcms) {
        // through-plane dimension = the one in which the first orientation volume is zero everywhere (stream.jl:221-223)
        bool allzero[3] = {true, true, true};
        for (int c = 0; c < 3; c++)
            for (int64_t i = 0; i < nvox && allzero[c]; i++) if (ovec[0][(size_t)c * nvox + i] != 0.0f) allzero[c] = false;
        int strd[3], ns = 0;
        for (int c = 0; c < 3; c++) if (!allzero[c]) strd[ns++] = c;
        FIB_CHECK(ns >= 2, FIB_ERR_INVALID, "LCM-guided tracking needs two in-plane dimensions with non-zero orientation components");
        in.strd0 = strd[0]; in.strd1 = strd[1];
    }
    std::unique_ptr<HostTimer> tm(new HostTimer("stream_host_masks_first_pass"));
    std::vector<uint8_t> m8, s8;
    if (mask) RC(mask_convert(mask, mask_dtype, nvox, true, m8));   // mask.vol .> 0, stream.jl:102
    // seed voxels: findall(W.mask) (stream.jl:744) or findall(seed.vol .> 0) (stream.jl:751), column-major order
    if (seed) RC(mask_convert(seed, seed_dtype, nvox, true, s8));
    else {
        Shard none;
        std::vector<int64_t> noseeds;
        RC(for_each_worker({ws[0]}, [&](int, DevState &d) { return stream_worker(d, 0, 1, in, mask ? &m8 : nullptr, noseeds, &s8, none); }));
    }
    tm.reset(new HostTimer("stream_host_seed_list"));
    std::vector<int64_t> seeds;
    for (int64_t i = 0; i < nvox; i++) if (s8[i]) seeds.push_back(i);
    tm.reset();
    const int nw = (int)ws.size();
    std::vector<Shard> shards((size_t)nw);
    RC(for_each_worker(ws, [&](int w, DevState &d) { return stream_worker(d, w, nw, in, mask ? &m8 : nullptr, seeds, nullptr, shards[w]); }));
    // merge in (seed, sub) order == the reference's order under static scheduling (stream.jl:757-787)
    int64_t nl = 0, np = 0;
    for (auto &s : shards) { nl += s.nl; np += s.np; }
    out->nlines = nl; out->npoints = np;
    if (nw == 1) {                                       // one worker: its arrays are the result
        Shard &s = shards[0];
        out->npts = s.npts; out->seed_index = s.sidx; out->xyz = s.xyz; out->flags = s.flags;
        s.npts = nullptr; s.sidx = nullptr; s.xyz = nullptr; s.flags = nullptr;
        return FIB_OK;
    }
    out->npts = (int32_t *)alloc_result(sizeof(int32_t) * (size_t)(nl > 0 ? nl : 1));
    out->seed_index = (int64_t *)alloc_result(sizeof(int64_t) * (size_t)(nl > 0 ? nl : 1));
    out->xyz = (float *)alloc_result(sizeof(float) * 3 * (size_t)(np > 0 ? np : 1));
    if (lcms) out->flags = (uint8_t *)alloc_result((size_t)(np > 0 ? np : 1));
    if (!out->npts || !out->seed_index || !out->xyz || (lcms && !out->flags)) { fib_tract_free(out); return fib::fail(FIB_ERR_NOMEM, "out of host memory"); }
    std::vector<int64_t> li((size_t)nw, 0), pi((size_t)nw, 0);
    int64_t ol = 0, op = 0;
    while (ol < nl) {
        int best = -1;
        for (int w = 0; w < nw; w++)
            if (li[w] < shards[w].nl && (best < 0 || shards[w].sidx[li[w]] < shards[best].sidx[li[best]])) best = w;
        Shard &s = shards[best];
        const int32_t n = s.npts[li[best]];
        out->npts[ol] = n; out->seed_index[ol] = s.sidx[li[best]];
        memcpy(out->xyz + 3 * op, s.xyz + 3 * pi[best], sizeof(float) * 3 * n);
        ol++; op += n; li[best]++; pi[best] += n;
    }
    return FIB_OK;
}

}  // namespace

extern "C" int fib_stream(int device, const fib_stream_params *prm, const float *const *ovec, const float *const *f,
                          float f_thresh, const float *fa, float fa_thresh, const void *mask, int mask_dtype,
                          const void *seed, int seed_dtype, const float *sublist, int32_t nsub, fib_tract_out *out) try {
    return stream_host(device, prm, ovec, f, f_thresh, fa, fa_thresh, mask, mask_dtype, seed, seed_dtype, sublist, nsub,
                       nullptr, 0.0f, 0, out);
} FIB_API_CATCH

extern "C" int fib_stream_lcm(int device, const fib_stream_params *prm, const float *const *ovec, const float *const *f,
                              float f_thresh, const float *fa, float fa_thresh, const void *mask, int mask_dtype,
                              const void *seed, int seed_dtype, const float *sublist, int32_t nsub,
                              const float *lcms, float lcm_thresh, uint64_t rng_seed, fib_tract_out *out) try {
    FIB_CHECK(lcms != nullptr, FIB_ERR_INVALID, "NULL lcms volume");
    return stream_host(device, prm, ovec, f, f_thresh, fa, fa_thresh, mask, mask_dtype, seed, seed_dtype, sublist, nsub,
                       lcms, lcm_thresh, rng_seed, out);
} FIB_API_CATCH
