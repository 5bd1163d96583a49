// host_tier.h -- the host tier's pure host logic: no HIP call in this file.
//
// api.hip (the drop-in entry points fib_dti_fit / fib_adc_fit / fib_gqi_rec / fib_dsi_rec, which replace the bodies of dti_fit
// dti.jl:221, adc_fit dti.jl:164, gqi_rec gqi.jl:109, dsi_rec dsi.jl:171) includes it and supplies the device back end (pinned ring,
// three HIP streams, events).  tests/host_tier_check.cpp includes it with a back end made of threads and memcpy and is built with
// -fsanitize=address,undefined and with -fsanitize=thread (tests/test_host_sanitizers.py): the ring's ordering, the run arithmetic and
// the chunk schedule are read by tools, not only by reviewers.
//
// Contents: mask element-type conversion; CPU binding; the copy pool; LiveMap (the runs of a mask) + the piece list of a chunk;
// row gather / scatter; chunk size, chunk schedule, slabs; run_chunks<Dev> -- the three-stage pipeline over voxel chunks.
#pragma once
#include <emmintrin.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fibers_hip.h"

namespace fibh {

// mask.vol[...] == 0 && continue (dti.jl:261, gqi.jl:135, dsi.jl:200)  -> nonzero test
// mask.vol .> 0 (stream.jl:102), seed.vol .> 0 (stream.jl:751)         -> positive test
template <typename T>
inline void mask_convert_t(const T *m, int64_t n, bool positive, uint8_t *out) {
    if (positive) for (int64_t i = 0; i < n; i++) out[i] = m[i] > (T)0 ? 1 : 0;
    else          for (int64_t i = 0; i < n; i++) out[i] = m[i] != (T)0 ? 1 : 0;
}
inline int dtype_size(int dtype) {
    switch (dtype) {
        case FIB_U8: case FIB_BOOL: case FIB_I8: return 1;
        case FIB_I16: case FIB_U16: return 2;
        case FIB_I32: case FIB_U32: case FIB_F32: return 4;
        case FIB_I64: case FIB_F64: return 8;
        default: return 0;
    }
}
// elements [i0, i0 + n) of a mask / seed volume of any numeric type -> bytes; false: unknown dtype
inline bool mask_convert_range(const void *m, int dtype, int64_t i0, int64_t n, bool positive, uint8_t *out) {
    switch (dtype) {
        case FIB_U8: case FIB_BOOL: mask_convert_t((const uint8_t *)m + i0, n, positive, out); break;
        case FIB_I8:  mask_convert_t((const int8_t *)m + i0, n, positive, out); break;
        case FIB_I16: mask_convert_t((const int16_t *)m + i0, n, positive, out); break;
        case FIB_U16: mask_convert_t((const uint16_t *)m + i0, n, positive, out); break;
        case FIB_I32: mask_convert_t((const int32_t *)m + i0, n, positive, out); break;
        case FIB_U32: mask_convert_t((const uint32_t *)m + i0, n, positive, out); break;
        case FIB_I64: mask_convert_t((const int64_t *)m + i0, n, positive, out); break;
        case FIB_F32: mask_convert_t((const float *)m + i0, n, positive, out); break;
        case FIB_F64: mask_convert_t((const double *)m + i0, n, positive, out); break;
        default: return false;
    }
    return true;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline void bind_this_thread(const std::vector<int> &cpus) {
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    (void)sched_setaffinity(0, sizeof set, &set);            // (best effort)
}

// ---- a small pool for the row copies between the caller's arrays and the pinned ring --------------------------------------
class CopyPool {
  public:
    CopyPool(int nthreads, const std::vector<int> &cpus) {
        for (int i = 0; i < nthreads; i++) th_.emplace_back([this, cpus] { bind_this_thread(cpus); work(); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int threads() const { return (int)th_.size(); }
    // runs fn(0..n-1), the caller takes part; returns when all are done.  One run at a time per pool.
    void run(int n, const std::function<void(int)> &fn) {
        if (n <= 0) return;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; n_ = n; next_ = 0; left_ = n;
        }
        cv_.notify_all();
        for (;;) {
            int i;
            { std::lock_guard<std::mutex> lk(mu_); if (next_ >= n_) break; i = next_++; }
            fn(i);
            std::lock_guard<std::mutex> lk(mu_);
            if (--left_ == 0) done_.notify_all();
        }
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return left_ == 0; });
        fn_ = nullptr;
    }

  private:
    void work() {
        for (;;) {
            int i;
            const std::function<void(int)> *fn;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || (fn_ && next_ < n_); });
                if (stop_) return;
                i = next_++; fn = fn_;
            }
            (*fn)(i);
            std::lock_guard<std::mutex> lk(mu_);
            if (--left_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *fn_ = nullptr;
    int n_ = 0, next_ = 0, left_ = 0;
    bool stop_ = false;
};

struct Rows { const float *in; float *out; int nrows; };   // a planar host array: nrows rows of nvox floats (row stride = nvox)

// [r4] The voxels of [vbeg, vend) inside the mask as runs.  With a mask that leaves a good part of the volume out, the host tier moves the
// voxels inside only: the gather stage packs their runs densely into the pinned ring, the device works on a dense all-inside chunk (whose
// rows are aligned whatever the mask looks like), the scatter stage puts the runs back and zero-fills the gaps.  A ball mask (36 % inside)
// moves 36 % of the bytes over PCIe.  Voxels are independent (dti.jl:258, gqi.jl:132, dsi.jl:197): results do not depend on it.
struct LiveMap {
    std::vector<int64_t> start, len, off;                // run i: voxels [start, start + len), `off` voxels inside the mask before it
    int64_t nlive = 0, vbeg = 0, vend = 0;
    size_t run_at(int64_t l) const { return (size_t)(std::upper_bound(off.begin(), off.end(), l) - off.begin()) - 1; }   // the run that holds inside-voxel l
};
// false: unknown mask dtype
inline bool build_live_map(CopyPool &pool, const void *mask, int mask_dtype, int64_t vbeg, int64_t vend, LiveMap &m) {
    m = LiveMap{};
    m.vbeg = vbeg; m.vend = vend;
    const int64_t n = vend - vbeg;
    if (n <= 0) return true;
    std::vector<uint8_t> m8((size_t)n);
    const int64_t piece = 1 << 18;
    std::atomic<bool> ok{true};
    pool.run((int)cdiv(n, piece), [&](int i) {
        const int64_t a = (int64_t)i * piece, c = std::min<int64_t>(piece, n - a);
        if (!mask_convert_range(mask, mask_dtype, vbeg + a, c, false, m8.data() + a)) ok = false;
    });
    if (!ok) return false;
    for (int64_t i = 0; i < n;) {
        const uint8_t *p = (const uint8_t *)memchr(m8.data() + i, 1, (size_t)(n - i));      // (mask_convert_range writes 0 / 1)
        if (!p) break;
        const int64_t a = p - m8.data();
        const uint8_t *q = (const uint8_t *)memchr(p, 0, (size_t)(n - a));
        const int64_t b = q ? q - m8.data() : n;
        m.start.push_back(vbeg + a); m.len.push_back(b - a); m.off.push_back(m.nlive);
        m.nlive += b - a;
        i = b;
    }
    return true;
}
// a mask that keeps less than this share of a slab is worth the packing ..
constexpr double LIVE_PACK_BELOW = 0.9;
// .. and whose runs are long enough: every run is a copy per row in both directions; below ~16 voxels (64 bytes) per run the per-run
// overhead outweighs the bytes saved (a noisy threshold mask), and the unpacked pipeline moves whole rows
inline bool live_pack_pays(const LiveMap &lm, int64_t v0, int64_t v1) {
    const bool long_runs = lm.start.empty() || lm.nlive >= (int64_t)16 * (int64_t)lm.start.size();
    return (double)lm.nlive < LIVE_PACK_BELOW * (double)(v1 - v0) && long_runs;
}

// [r6] The pieces of the caller's rows that make up inside-voxels [l0, l0 + n) of a LiveMap, listed ONCE per chunk (every one of the
// chunk's 270 + 333 rows walks the same list; round 5 walked the runs through a std::function per row and piece).
struct Piece {
    int64_t vox;                                         // first voxel of the piece in the caller's row
    int64_t gap0;                                        // scatter: the voxels [gap0, vox) in front of it lie outside the mask and read 0 (gap0 == vox: no gap)
    int32_t pos, len;                                    // position in the packed chunk, voxels
};
inline void build_pieces(const LiveMap &lm, int64_t l0, int64_t n, std::vector<Piece> &out) {
    out.clear();
    if (n <= 0) return;
    size_t ri = lm.run_at(l0);
    int64_t done = 0;
    while (done < n) {
        const int64_t inrun = l0 + done - lm.off[ri];
        const int64_t c = std::min<int64_t>(lm.len[ri] - inrun, n - done);
        const int64_t vox = lm.start[ri] + inrun;
        const int64_t g0 = inrun != 0 ? vox : (ri == 0 ? lm.vbeg : lm.start[ri - 1] + lm.len[ri - 1]);
        out.push_back(Piece{vox, g0, (int32_t)done, (int32_t)c});
        done += c; ri++;
    }
}
// A piece is a few hundred bytes at a new address: its first cache lines miss, and a thread that copies piece after piece waits
// for DRAM once per piece (round 5: 22 GB/s for the masked gather against 64 GB/s for whole rows).  The source lines of the piece
// PF pieces ahead are requested while the current one is copied.
constexpr size_t PIECE_PREFETCH = 6;
inline void prefetch_span(const void *p, size_t bytes) {
    const char *c = (const char *)p;
    for (size_t o = 0; o < bytes && o < 512; o += 64) __builtin_prefetch(c + o, 0, 0);
}
// Streaming (non-temporal) forms of the two row operations, for destinations that are written once and not read by this CPU soon: whole
// 64-byte lines leave through the write-combining buffers without the read-for-ownership an ordinary store to a cold line pays (half
// the memory traffic of a scatter); the partial lines at either end are written normally.  SSE2: part of every x86-64.
inline void copy_stream(float *d, const float *s, int64_t n) {
    while (n > 0 && ((uintptr_t)d & 63)) { *d++ = *s++; n--; }
    for (; n >= 16; n -= 16, d += 16, s += 16) {
        const __m128 a = _mm_loadu_ps(s), b = _mm_loadu_ps(s + 4), c = _mm_loadu_ps(s + 8), e = _mm_loadu_ps(s + 12);
        _mm_stream_ps(d, a); _mm_stream_ps(d + 4, b); _mm_stream_ps(d + 8, c); _mm_stream_ps(d + 12, e);
    }
    for (; n > 0; n--) *d++ = *s++;
}
inline void zero_stream(float *d, int64_t n) {
    while (n > 0 && ((uintptr_t)d & 63)) { *d++ = 0.0f; n--; }
    const __m128 z = _mm_setzero_ps();
    for (; n >= 16; n -= 16, d += 16) { _mm_stream_ps(d, z); _mm_stream_ps(d + 4, z); _mm_stream_ps(d + 8, z); _mm_stream_ps(d + 12, z); }
    for (; n > 0; n--) *d++ = 0.0f;
}
// gather: the caller's row -> the packed row of the ring (nd - n zeros behind the n voxels: the device sees whole groups of 32)
inline void gather_row(float *dst, const float *row, const Piece *pc, size_t np, int64_t n, int64_t nd, bool nt = false) {
    for (size_t i = 0; i < np; i++) {
        if (i + PIECE_PREFETCH < np) prefetch_span(row + pc[i + PIECE_PREFETCH].vox, (size_t)pc[i + PIECE_PREFETCH].len * 4);
        if (nt) copy_stream(dst + pc[i].pos, row + pc[i].vox, pc[i].len);
        else memcpy(dst + pc[i].pos, row + pc[i].vox, (size_t)pc[i].len * 4);
    }
    if (nd > n) memset(dst + n, 0, (size_t)(nd - n) * 4);
    if (nt) _mm_sfence();                                // the streamed lines are globally visible before the row is reported done
}
// scatter: the packed row of the ring -> the caller's row; zero_gaps: the voxels outside the mask in front of each piece read 0
// (and, for the volume's last chunk, those behind the last run up to tail_end)
inline void scatter_row(float *row, const float *src, const Piece *pc, size_t np, bool zero_gaps, int64_t tail_end, bool nt = false) {
    for (size_t i = 0; i < np; i++) {
        if (i + PIECE_PREFETCH < np) prefetch_span(src + pc[i + PIECE_PREFETCH].pos, (size_t)pc[i + PIECE_PREFETCH].len * 4);
        if (zero_gaps && pc[i].vox > pc[i].gap0) {
            if (nt) zero_stream(row + pc[i].gap0, pc[i].vox - pc[i].gap0);
            else memset(row + pc[i].gap0, 0, (size_t)(pc[i].vox - pc[i].gap0) * 4);
        }
        if (nt) copy_stream(row + pc[i].vox, src + pc[i].pos, pc[i].len);
        else memcpy(row + pc[i].vox, src + pc[i].pos, (size_t)pc[i].len * 4);
    }
    if (zero_gaps && np > 0 && tail_end >= 0) {
        const int64_t g0 = pc[np - 1].vox + pc[np - 1].len;
        if (tail_end > g0) { if (nt) zero_stream(row + g0, tail_end - g0); else memset(row + g0, 0, (size_t)(tail_end - g0) * 4); }
    }
    if (nt) _mm_sfence();
}

// chunk size: [r4] 262 144 voxels (four 256-voxel work items per CU for the contraction kernel) as long as a ring slot stays below 384 MB:
// fib_gqi_rec 140^3 x 270 takes 103 ms with chunks of 131 072 voxels, 90-93 ms with 262 144 (95 with 524 288: fewer chunks to overlap;
// tools/host_tier_sweep.py); the pinned ring is 3 x (rows_in + rows_out) x chunk x 4 bytes per device.  override: FIBERS_HOST_CHUNK.
inline int64_t pick_chunk(int64_t nrange, int rows_in, int rows_out, const char *override_env = nullptr) {
    const int rows = rows_in > rows_out ? rows_in : rows_out;
    int64_t c = 262144;
    while (c > 8192 && c * rows * 4 > (int64_t)384 << 20) c >>= 1;
    if (override_env) { const long long v = atoll(override_env); if (v >= 1024) c = v / 32 * 32; }
    return c < nrange ? c : std::max<int64_t>((nrange + 3) / 4 * 4, 4);
}
// [r5] The chunk schedule of a range of `total` voxels: 1/8, 1/4 and 1/2 of a chunk first, whole chunks after that.  The download stream
// is the pipeline's long pole (fib_gqi_rec moves 3.66 GB out against 2.96 GB in), and it cannot start before the first chunk has been
// gathered, uploaded and computed: with a small first chunk it starts after ~2 ms instead of ~8.  Offsets are multiples of 32 voxels.
inline std::vector<int64_t> chunk_schedule(int64_t total, int64_t chunk) {
    std::vector<int64_t> off;
    int64_t o = 0;
    for (int64_t c : {chunk / 8, chunk / 4, chunk / 2}) {
        c = c / 32 * 32;
        if (c >= 8192 && total - o > chunk + c) { off.push_back(o); o += c; }
    }
    while (o < total) { off.push_back(o); o += std::min(chunk, total - o); }
    off.push_back(total);
    return off;
}
inline int chunk_count(int64_t total, int64_t chunk) { return total > 0 ? (int)chunk_schedule(total, chunk).size() - 1 : 0; }

// contiguous slab of voxels for worker i of n (the reference's z-slice blocks, here at 4-voxel granularity so that rows
// stay 16-byte aligned on the device)
inline void slab(int64_t nvox, int n, int i, int64_t &v0, int64_t &v1) {
    const int64_t q = (nvox + 3) / 4, per = q / n, rem = q % n;
    const int64_t a = per * i + std::min<int64_t>(i, rem), b = a + per + (i < rem ? 1 : 0);
    v0 = std::min(a * 4, nvox); v1 = std::min(b * 4, nvox);
}

// ---- the chunk pipeline ----------------------------------------------------------------------------------------------------
constexpr int NBUF = 3;                                  // ring depth: chunk k uploads while k-1 computes and k-2 downloads
enum Stream { S_IN = 0, S_CMP = 1, S_OUT = 2 };          // upload | kernels | download
enum Event { E_IN = 0, E_CMP = 1, E_OUT = 2 };           // per ring slot: upload done | kernels done | download done

// What run_chunks needs of a device (api.hip: HIP streams + events + pinned / device buffers; tests: threads + memcpy):
//   int   ensure(size_t in_bytes, size_t out_bytes)     ring buffers of every slot, FIB_OK or an error code
//   char *pin_in(int b), *pin_out(int b)                the slot's pinned staging buffers
//   int   upload(int b, size_t bytes)                   pin_in[b] -> device, asynchronous on S_IN
//   int   compute(int k, int b, int64_t rel, int64_t nd, int rin)   the fit of chunk k on the slot's device buffers, asynchronous on S_CMP
//   int   download(int b, size_t bytes)                 device -> pin_out[b], asynchronous on S_OUT
//   int   record(Event e, int b)                        event of slot b on its stream (E_IN on S_IN, E_CMP on S_CMP, E_OUT on S_OUT)
//   int   stream_wait(Stream s, Event e, int b)         work queued on s after this call starts after the event
//   int   host_wait(Event e, int b)                     blocks the calling host thread until the event has happened
//   void  drain()                                       all three streams idle
//   void  prof(const char *name, double ms)             host-stage timing (no-op outside profiling)
//   int   fail(int code, const char *msg)               records the message for fib_last_error(), returns code
//   std::string last_error() / void set_error(const std::string &)   hand a worker thread's message to the calling thread
//
// voxels [vbeg, vend) of a volume of nvox voxels.  Blocking.  lm != NULL: only the voxels inside the mask travel (LiveMap); the fit then
// sees dense chunks whose mask is all ones (padded with voxels outside to a multiple of 32) and `rel` counts voxels inside the mask.
// outputs_zeroed (FIB_MASK_OUTPUTS_ZEROED): the caller's output arrays are zero already -- the gaps between the runs are left alone.
// nt: the row copies of the packed form and the scatter of whole rows use streaming stores (copy_stream / zero_stream).
//
// Ordering (what the sanitizer build checks): pin_in[b] is rewritten by the gather of chunk k only after the upload of chunk k - NBUF
// has completed (host_wait E_IN); the slot's device input is overwritten by upload k only after the kernels of chunk k - NBUF
// (S_IN waits E_CMP), its device output by kernels k only after download k - NBUF (S_CMP waits E_OUT); pin_out[b] is overwritten by
// download k only after chunk k - NBUF has been scattered (the enqueueing thread waits for `scat`), and read by the scatter of chunk k
// only after download k (host_wait E_OUT).
template <class Dev>
int run_chunks(Dev &dev, CopyPool &pool_in, CopyPool &pool_out, int64_t vbeg, int64_t vend, int64_t nvox, const std::vector<Rows> &ins,
               const void *mask, int mask_dtype, const std::vector<Rows> &outs, int64_t chunk, const LiveMap *lm = nullptr, bool outputs_zeroed = false, bool nt = false) {
    if (vend <= vbeg) return FIB_OK;
    int rin = 0, rout = 0;
    for (auto &r : ins) rin += r.nrows;
    for (auto &r : outs) rout += r.nrows;
    const int64_t total = lm ? lm->nlive : vend - vbeg;
    if (lm && total == 0) {                              // nothing inside the mask: every output row of the range reads 0
        if (!outputs_zeroed) {
            std::vector<float *> rows;
            for (auto &r : outs) for (int i = 0; i < r.nrows; i++) rows.push_back(r.out + (int64_t)i * nvox);
            pool_in.run((int)rows.size(), [&](int i) { memset(rows[i] + vbeg, 0, (size_t)(vend - vbeg) * 4); });
        }
        return FIB_OK;
    }
    if (lm) chunk = (chunk + 31) / 32 * 32;
    const size_t in_bytes = (size_t)rin * chunk * 4 + (size_t)chunk, out_bytes = (size_t)rout * chunk * 4;
    { const int rc = dev.ensure(in_bytes, out_bytes); if (rc != FIB_OK) return rc; }
    const std::vector<int64_t> coff = chunk_schedule(total, chunk);
    const int nchunks = (int)coff.size() - 1;
    std::atomic<int> err{FIB_OK};
    // chunk k: `n` voxels that travel, `nd` voxels the device sees (lm: padded to a multiple of 32 so that its rows stay on cache lines)
    auto span = [&](int k, int64_t &o0, int64_t &n, int64_t &nd) {
        o0 = coff[k]; n = coff[k + 1] - coff[k];
        nd = lm ? (n + 31) / 32 * 32 : n;
    };
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto scatter = [&](int k) {                          // chunk k: pinned ring -> the caller's arrays (the scatter thread)
        const int b = k % NBUF;
        int64_t o0, n, nd;
        span(k, o0, n, nd);
        const auto tw0 = clk::now();
        if (dev.host_wait(E_OUT, b) != FIB_OK) { err = dev.fail(FIB_ERR_HIP, "device-to-host copy of a chunk failed"); return; }
        const auto tw1 = clk::now();
        std::vector<std::pair<float *, const float *>> rows;
        const float *src = reinterpret_cast<const float *>(dev.pin_out(b));
        if (!lm) {
            const int64_t v0 = vbeg + o0;
            for (auto &r : outs) for (int i = 0; i < r.nrows; i++) { rows.emplace_back(r.out + (int64_t)i * nvox + v0, src); src += n; }
            pool_out.run((int)rows.size(), [&](int i) { if (nt) { copy_stream(rows[i].first, rows[i].second, n); _mm_sfence(); } else memcpy(rows[i].first, rows[i].second, (size_t)n * 4); });
        } else {
            for (auto &r : outs) for (int i = 0; i < r.nrows; i++) { rows.emplace_back(r.out + (int64_t)i * nvox, src); src += nd; }
            std::vector<Piece> pc;
            build_pieces(*lm, o0, n, pc);
            const int64_t tail = k == nchunks - 1 ? lm->vend : -1;
            pool_out.run((int)rows.size(), [&](int i) { scatter_row(rows[i].first, rows[i].second, pc.data(), pc.size(), !outputs_zeroed, tail, nt); });
        }
        dev.prof("host_scatter", ms(tw1, clk::now()));
        dev.prof("host_scatter_wait", ms(tw0, tw1));
    };
    auto enqueue = [&](int k) -> int {                   // chunk k: gather, upload, compute, download (asynchronous from the upload on)
        const int b = k % NBUF;
        int64_t o0, n, nd;
        span(k, o0, n, nd);
        // the pinned input buffer is free once the upload of chunk k - NBUF has completed
        const auto tg0 = clk::now();
        if (k >= NBUF && dev.host_wait(E_IN, b) != FIB_OK) return dev.fail(FIB_ERR_HIP, "host-to-device copy of a chunk failed");
        const auto tg1 = clk::now();
        {
            std::vector<std::pair<float *, const float *>> rows;
            float *dst = reinterpret_cast<float *>(dev.pin_in(b));
            uint8_t *m8 = reinterpret_cast<uint8_t *>(dev.pin_in(b)) + (size_t)rin * nd * 4;
            std::atomic<bool> mok{true};
            if (!lm) {
                const int64_t v0 = vbeg + o0;
                for (auto &r : ins) for (int i = 0; i < r.nrows; i++) { rows.emplace_back(dst, r.in + (int64_t)i * nvox + v0); dst += n; }
                pool_in.run((int)rows.size() + 1, [&](int i) {
                    if (i < (int)rows.size()) memcpy(rows[i].first, rows[i].second, (size_t)n * 4);
                    else if (!mask_convert_range(mask, mask_dtype, v0, n, false, m8)) mok = false;
                });
            } else {
                for (auto &r : ins) for (int i = 0; i < r.nrows; i++) { rows.emplace_back(dst, r.in + (int64_t)i * nvox); dst += nd; }
                std::vector<Piece> pc;
                build_pieces(*lm, o0, n, pc);
                pool_in.run((int)rows.size() + 1, [&](int i) {
                    if (i == (int)rows.size()) { memset(m8, 1, (size_t)n); memset(m8 + n, 0, (size_t)(nd - n)); return; }
                    gather_row(rows[i].first, rows[i].second, pc.data(), pc.size(), n, nd, nt);
                });
            }
            if (!mok) return dev.fail(FIB_ERR_INVALID, "unknown mask dtype");
        }
        dev.prof("host_gather", ms(tg1, clk::now()));
        dev.prof("host_gather_wait", ms(tg0, tg1));
        const size_t ib = (size_t)rin * nd * 4 + (size_t)nd;
        int rc;
        // device buffers of this ring slot: the kernels of chunk k - NBUF have read the input, its download has read the output
        if (k >= NBUF) {
            if ((rc = dev.stream_wait(S_IN, E_CMP, b)) != FIB_OK) return rc;
            if ((rc = dev.stream_wait(S_CMP, E_OUT, b)) != FIB_OK) return rc;
        }
        if ((rc = dev.upload(b, ib)) != FIB_OK) return rc;
        if ((rc = dev.record(E_IN, b)) != FIB_OK) return rc;
        if ((rc = dev.stream_wait(S_CMP, E_IN, b)) != FIB_OK) return rc;
        if ((rc = dev.compute(k, b, o0, nd, rin)) != FIB_OK) return rc;
        if ((rc = dev.record(E_CMP, b)) != FIB_OK) return rc;
        if ((rc = dev.stream_wait(S_OUT, E_CMP, b)) != FIB_OK) return rc;
        if ((rc = dev.download(b, (size_t)rout * nd * 4)) != FIB_OK) return rc;
        return dev.record(E_OUT, b);
    };
    // [r5] Two host stages side by side: this thread gathers chunk k into the ring and enqueues its upload, kernels and download; a second
    // thread waits for downloads and scatters them into the caller's arrays (each stage has its own copy pool).
    // Ring slot b = k % NBUF of the OUTPUT side is free for chunk k once chunk k - NBUF has been scattered.
    {
        std::mutex mu;
        std::condition_variable cv;
        int enq = 0, scat = 0;                               // chunks enqueued by this thread | scattered by the other
        bool stop = false;
        std::string smsg;
        std::thread ts([&] {
            try {
                for (int k = 0; k < nchunks; k++) {
                    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return enq > k || stop; }); if (enq <= k) break; }
                    scatter(k);
                    if (err != FIB_OK) { std::lock_guard<std::mutex> lk(mu); smsg = dev.last_error(); }   // (the message is thread-local: hand it over)
                    { std::lock_guard<std::mutex> lk(mu); scat = k + 1; }
                    cv.notify_all();
                    if (err != FIB_OK) break;
                }
            } catch (...) {
                err = FIB_ERR_INVALID;
                std::lock_guard<std::mutex> lk(mu);
                smsg = "internal error in the scatter stage"; scat = nchunks;
                cv.notify_all();
            }
        });
        // [r6] whatever happens on this thread (an error code, std::bad_alloc from a row list), the scatter thread is told to stop and
        // joined before the function is left: an exception that unwound past a joinable std::thread would terminate the process
        // instead of returning an error code
        struct Joiner { std::thread &t; std::mutex &mu; std::condition_variable &cv; bool &stop;
                        ~Joiner() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (t.joinable()) t.join(); } };
        int main_rc = FIB_OK;
        std::string main_msg;
        {
            Joiner joiner{ts, mu, cv, stop};
            try {
                for (int k = 0; k < nchunks && err == FIB_OK; k++) {
                    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return scat >= k - NBUF + 1 || err != FIB_OK; }); }
                    if (err != FIB_OK) break;
                    const int rc = enqueue(k);
                    if (rc != FIB_OK) { main_rc = rc; main_msg = dev.last_error(); break; }
                    { std::lock_guard<std::mutex> lk(mu); enq = k + 1; }
                    cv.notify_all();
                }
            } catch (const std::bad_alloc &) { main_rc = FIB_ERR_NOMEM; main_msg = "out of host memory"; }
            catch (...) { main_rc = FIB_ERR_INVALID; main_msg = "internal error in the gather stage"; }
        }
        if (main_rc != FIB_OK) { err = main_rc; dev.set_error(main_msg); }
        else if (err != FIB_OK && !smsg.empty()) dev.set_error(smsg);
    }
    // leave the streams idle whatever happened (the ring buffers are reused by the next call)
    dev.drain();
    return err;
}

}  // namespace fibh
