// host_tier_check.cpp -- the host tier's pure host code (fibers.jl_amd/csrc/host_tier.h) under the sanitizers.
// Built three times by tests/test_host_sanitizers.py: -fsanitize=address,undefined, -fsanitize=thread, and -fsanitize=thread with
// -DFIBH_MUTATE=1 (a back end whose host_wait(E_OUT) does not wait: the harness must FAIL then -- a test of the test).
// No HIP here: the device back end of fibh::run_chunks is made of three threads ("streams": upload | kernels | download) that execute
// queued operations in order, events are generation counters under a mutex, copies are memcpy, and the "fit" of a chunk is arithmetic
// whose result is known in closed form.  What is checked: the results (every voxel of every output row, zero-filling outside the mask
// included) and -- by ThreadSanitizer -- that no ring buffer is touched by two stages at once; plus the run arithmetic (LiveMap, piece
// lists), the chunk schedule, slabs, chunk sizes and the mask element types against brute force.
#include <cmath>
#include <cstdio>
#include <deque>
#include <random>

#include "../fibers.jl_amd/csrc/host_tier.h"

using namespace fibh;

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); exit(1); } } while (0)

// ---- a device made of threads ------------------------------------------------------------------------------------------------------
struct FakeStream {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    FakeStream() { th = std::thread([this] { work(); }); }
    ~FakeStream() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); th.join(); }
    void push(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); } cv.notify_all(); }
    void work() {
        for (;;) {
            std::function<void()> f;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return stop || !q.empty(); }); if (q.empty()) return; f = std::move(q.front()); q.pop_front(); busy = true; }
            f();
            { std::lock_guard<std::mutex> lk(mu); busy = false; }
            cv.notify_all();
        }
    }
    void sync() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return q.empty() && !busy; }); }
};
struct FakeEvent {
    std::mutex mu;
    std::condition_variable cv;
    long issued = 0, done = 0;                           // records made | records executed
    long mark() { std::lock_guard<std::mutex> lk(mu); return ++issued; }
    long latest() { std::lock_guard<std::mutex> lk(mu); return issued; }
    void signal(long g) { { std::lock_guard<std::mutex> lk(mu); if (g > done) done = g; } cv.notify_all(); }
    void wait(long g) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done >= g; }); }
};
// the "fit": output row j of a voxel = (input row j % rin) * (j + 2) + chunk-relative bookkeeping is NOT used (results must not depend on the cut)
static inline float fit_value(float in, int j) { return in * (float)(j + 2) + 1.0f; }

struct FakeDev {
    int rin = 0, rout = 0;
    FakeStream st[3];
    FakeEvent ev[3][NBUF];
    std::vector<char> pin_in_[NBUF], pin_out_[NBUF], dev_in_[NBUF], dev_out_[NBUF];
    std::mutex emu;
    std::string err;
    int chunks_computed = 0;
    int ensure(size_t ib, size_t ob) { for (int b = 0; b < NBUF; b++) { pin_in_[b].resize(ib); dev_in_[b].resize(ib); pin_out_[b].resize(ob); dev_out_[b].resize(ob); } return FIB_OK; }
    char *pin_in(int b) { return pin_in_[b].data(); }
    char *pin_out(int b) { return pin_out_[b].data(); }
    int upload(int b, size_t bytes) { CHECK(bytes <= dev_in_[b].size()); st[S_IN].push([this, b, bytes] { memcpy(dev_in_[b].data(), pin_in_[b].data(), bytes); }); return FIB_OK; }
    int download(int b, size_t bytes) { CHECK(bytes <= dev_out_[b].size()); st[S_OUT].push([this, b, bytes] { memcpy(pin_out_[b].data(), dev_out_[b].data(), bytes); }); return FIB_OK; }
    int compute(int, int b, int64_t, int64_t nd, int rin_) {
        CHECK(rin_ == rin);
        st[S_CMP].push([this, b, nd] {
            const float *in = reinterpret_cast<const float *>(dev_in_[b].data());
            const uint8_t *m = reinterpret_cast<const uint8_t *>(dev_in_[b].data()) + (size_t)rin * nd * 4;
            float *out = reinterpret_cast<float *>(dev_out_[b].data());
            for (int j = 0; j < rout; j++)
                for (int64_t v = 0; v < nd; v++) out[(size_t)j * nd + v] = m[v] ? fit_value(in[(size_t)(j % rin) * nd + v], j) : 0.0f;   // (outside the mask: 0, as the kernels write)
            chunks_computed++;
        });
        return FIB_OK;
    }
    int record(Event e, int b) { const long g = ev[e][b].mark(); st[(int)e].push([this, e, b, g] { ev[e][b].signal(g); }); return FIB_OK; }
    int stream_wait(Stream s, Event e, int b) { const long g = ev[e][b].latest(); st[s].push([this, e, b, g] { ev[e][b].wait(g); }); return FIB_OK; }
    int host_wait(Event e, int b) {
#if FIBH_MUTATE
        if (e == E_OUT) return FIB_OK;                   // the mutant: the scatter of chunk k reads pin_out[b] without waiting for its download
#endif
        ev[e][b].wait(ev[e][b].latest());
        return FIB_OK;
    }
    void drain() { for (auto &s : st) s.sync(); }
    void prof(const char *, double) {}
    int fail(int code, const char *msg) { std::lock_guard<std::mutex> lk(emu); err = msg; return code; }
    std::string last_error() { std::lock_guard<std::mutex> lk(emu); return err; }
    void set_error(const std::string &m) { std::lock_guard<std::mutex> lk(emu); err = m; }
};

// ---- the pipeline against the closed form ------------------------------------------------------------------------------------------
static void pipeline_case(int64_t nvox, int64_t vbeg, int64_t vend, int rin, const std::vector<int> &out_rows, int64_t chunk, int mask_kind, bool packed,
                          bool outputs_zeroed, unsigned seed, bool nt = false) {
    std::mt19937 rng(seed);
    std::vector<float> in((size_t)rin * nvox);
    for (auto &x : in) x = (float)(rng() % 1000) * 0.25f;
    std::vector<uint8_t> mask((size_t)nvox);
    for (int64_t v = 0; v < nvox; v++) {
        switch (mask_kind) {
            case 0: mask[v] = 1; break;                                               // all inside
            case 1: mask[v] = (v / 37) % 3 != 1; break;                                // runs of 37 / 74
            case 2: mask[v] = (rng() % 100) < 35; break;                               // noisy
            default: mask[v] = 0;
        }
    }
    int rout = 0;
    for (int r : out_rows) rout += r;
    const float stale = -777.0f;                                                        // what the caller's arrays hold before the call
    std::vector<std::vector<float>> outs_mem;
    std::vector<Rows> ins = {{in.data(), nullptr, rin}}, outs;
    for (int r : out_rows) { outs_mem.emplace_back((size_t)r * nvox, outputs_zeroed ? 0.0f : stale); }
    for (size_t i = 0; i < out_rows.size(); i++) outs.push_back({nullptr, outs_mem[i].data(), out_rows[i]});
    CopyPool pin(3, {}), pout(3, {});
    FakeDev dev;
    dev.rin = rin; dev.rout = rout;
    LiveMap lm;
    const LiveMap *use = nullptr;
    if (packed) { CHECK(build_live_map(pin, mask.data(), FIB_U8, vbeg, vend, lm)); use = &lm; }
    const int rc = run_chunks(dev, pin, pout, vbeg, vend, nvox, ins, mask.data(), FIB_U8, outs, chunk, use, outputs_zeroed, nt);
    CHECK(rc == FIB_OK);
    int j = 0;
    for (size_t a = 0; a < out_rows.size(); a++)
        for (int i = 0; i < out_rows[a]; i++, j++)
            for (int64_t v = 0; v < nvox; v++) {
                const float got = outs_mem[a][(size_t)i * nvox + v];
                float want;
                if (v < vbeg || v >= vend) want = outputs_zeroed ? 0.0f : stale;                 // outside the range: untouched
                else if (mask[v]) want = fit_value(in[(size_t)(j % rin) * nvox + v], j);
                else want = 0.0f;                                                               // outside the mask: 0 (written, or left at the caller's 0)
                if (got != want) { fprintf(stderr, "pipeline mismatch: row %d voxel %ld got %g want %g (nvox %ld chunk %ld mask %d packed %d zeroed %d)\n", j, (long)v, got, want,
                                           (long)nvox, (long)chunk, mask_kind, (int)packed, (int)outputs_zeroed); exit(1); }
            }
}

static void units() {
    // mask element types
    {
        const double vals[6] = {0, 1, -2, 0.5, 0, 3};
        uint8_t nz[6], pos[6];
        std::vector<float> f(vals, vals + 6); std::vector<double> d(vals, vals + 6); std::vector<int16_t> i16 = {0, 1, -2, 0, 0, 3}; std::vector<int64_t> i64 = {0, 1, -2, 0, 0, 3};
        CHECK(mask_convert_range(f.data(), FIB_F32, 0, 6, false, nz) && mask_convert_range(f.data(), FIB_F32, 0, 6, true, pos));
        const uint8_t wnz[6] = {0, 1, 1, 1, 0, 1}, wpos[6] = {0, 1, 0, 1, 0, 1};
        CHECK(!memcmp(nz, wnz, 6) && !memcmp(pos, wpos, 6));
        CHECK(mask_convert_range(d.data(), FIB_F64, 1, 5, false, nz) && !memcmp(nz, wnz + 1, 5));
        const uint8_t inz[6] = {0, 1, 1, 0, 0, 1}, ipos[6] = {0, 1, 0, 0, 0, 1};
        CHECK(mask_convert_range(i16.data(), FIB_I16, 0, 6, false, nz) && !memcmp(nz, inz, 6));
        CHECK(mask_convert_range(i64.data(), FIB_I64, 0, 6, true, pos) && !memcmp(pos, ipos, 6));
        CHECK(!mask_convert_range(f.data(), 99, 0, 6, false, nz));
        CHECK(dtype_size(FIB_U8) == 1 && dtype_size(FIB_I16) == 2 && dtype_size(FIB_F32) == 4 && dtype_size(FIB_F64) == 8 && dtype_size(99) == 0);
    }
    // live map + piece lists against brute force
    CopyPool pool(2, {});
    std::mt19937 rng(5);
    for (int trial = 0; trial < 40; trial++) {
        const int64_t n = 1 + rng() % 5000, vbeg = rng() % n, vend = vbeg + rng() % (n - vbeg + 1);
        std::vector<int32_t> mask((size_t)n);
        const int p = 1 + rng() % 99;
        for (auto &m : mask) m = (int)(rng() % 100) < p ? (int)(rng() % 7) - 3 : 0;
        LiveMap lm;
        CHECK(build_live_map(pool, mask.data(), FIB_I32, vbeg, vend, lm));
        std::vector<int64_t> inside;
        for (int64_t v = vbeg; v < vend; v++) if (mask[v] != 0) inside.push_back(v);
        CHECK(lm.nlive == (int64_t)inside.size() && lm.vbeg == vbeg && lm.vend == vend);
        int64_t cnt = 0;
        for (size_t r = 0; r < lm.start.size(); r++) {
            CHECK(lm.len[r] > 0 && lm.off[r] == cnt);
            if (r) CHECK(lm.start[r] > lm.start[r - 1] + lm.len[r - 1]);               // maximal runs: a gap between them
            for (int64_t i = 0; i < lm.len[r]; i++) CHECK(inside[(size_t)(cnt + i)] == lm.start[r] + i);
            cnt += lm.len[r];
        }
        CHECK(cnt == lm.nlive);
        if (lm.nlive == 0) continue;
        const int64_t l0 = rng() % lm.nlive, len = 1 + rng() % (lm.nlive - l0);
        std::vector<Piece> pc;
        build_pieces(lm, l0, len, pc);
        int64_t pos = 0;
        for (size_t i = 0; i < pc.size(); i++) {
            CHECK(pc[i].pos == pos && pc[i].len > 0);
            for (int32_t k = 0; k < pc[i].len; k++) CHECK(inside[(size_t)(l0 + pos + k)] == pc[i].vox + k);
            CHECK(pc[i].gap0 <= pc[i].vox && pc[i].gap0 >= vbeg);
            for (int64_t v = pc[i].gap0; v < pc[i].vox; v++) CHECK(mask[v] == 0);         // the gap really lies outside the mask
            if (pc[i].gap0 < pc[i].vox && pc[i].gap0 > vbeg) CHECK(mask[pc[i].gap0 - 1] != 0);   // .. and starts right behind the previous run
            pos += pc[i].len;
        }
        CHECK(pos == len);
    }
    // chunk schedule, chunk size, slabs
    for (int64_t total : {1ll, 31ll, 4096ll, 262144ll, 262145ll, 998592ll, 2744000ll, 5000000ll})
        for (int64_t chunk : {32ll, 8192ll, 65536ll, 262144ll}) {
            const auto off = chunk_schedule(total, chunk);
            CHECK(off.front() == 0 && off.back() == total && chunk_count(total, chunk) == (int)off.size() - 1);
            for (size_t i = 0; i + 1 < off.size(); i++) { CHECK(off[i + 1] > off[i] && off[i + 1] - off[i] <= chunk); if (i + 2 < off.size()) CHECK(off[i + 1] % 32 == 0); }
        }
    CHECK(pick_chunk(2744000, 270, 330) == 262144 && pick_chunk(1000, 270, 330) == 1000 && pick_chunk(2744000, 515, 845) <= 262144);
    CHECK(pick_chunk(2744000, 270, 330, "65536") == 65536 && pick_chunk(2744000, 270, 330, "12") == 262144);
    CHECK((int64_t)pick_chunk(2744000, 515, 845) * 845 * 4 <= ((int64_t)384 << 20));
    for (int64_t nvox : {1ll, 7ll, 1000ll, 2744000ll, 1803200ll})
        for (int n : {1, 2, 3, 8}) {
            int64_t prev = 0;
            for (int i = 0; i < n; i++) { int64_t a, b; slab(nvox, n, i, a, b); CHECK(a == prev && b >= a && (b % 4 == 0 || b == nvox)); prev = b; }
            CHECK(prev == nvox);
        }
}

int main() {
    units();
    unsigned seed = 1;
    // whole rows (no packing): one chunk, exactly NBUF chunks, many more chunks than ring slots, a sub-range, every mask
    for (int mk : {0, 1, 2}) {
        pipeline_case(500, 0, 500, 3, {2, 3}, 512, mk, false, false, seed++);
        pipeline_case(3 * 64, 0, 3 * 64, 2, {1, 3}, 64, mk, false, false, seed++);
        pipeline_case(1700, 0, 1700, 4, {1, 1, 3}, 64, mk, false, false, seed++);
        pipeline_case(1700, 128, 1500, 4, {2}, 96, mk, false, false, seed++);
    }
    // packed (only the voxels inside the mask travel): gaps are zero-filled unless the caller says its arrays are zero already
    for (int mk : {0, 1, 2, 3})
        for (bool zeroed : {false, true}) {
            pipeline_case(1700, 0, 1700, 3, {1, 3}, 64, mk, true, zeroed, seed++);
            pipeline_case(2500, 100, 2404, 2, {2, 2}, 160, mk, true, zeroed, seed++);
            pipeline_case(90, 0, 90, 2, {1}, 32, mk, true, zeroed, seed++);
        }
    // the streaming-store forms of the row copies (every alignment of source and destination: the sub-range starts are odd)
    for (int mk : {0, 1, 2})
        for (bool zeroed : {false, true}) {
            pipeline_case(1700, 3, 1699, 3, {1, 3}, 64, mk, true, zeroed, seed++, true);
            pipeline_case(1701, 7, 1690, 2, {2}, 96, mk, false, zeroed, seed++, true);
        }
    for (int64_t n = 0; n < 70; n++)
        for (int off = 0; off < 17; off++) {
            std::vector<float> a(128, 5.0f), b(128);
            for (int i = 0; i < 128; i++) b[i] = (float)i;
            copy_stream(a.data() + off, b.data() + 3, n);
            for (int i = 0; i < 128; i++) CHECK(a[i] == ((i >= off && i < off + n) ? (float)(i - off + 3) : 5.0f));
            zero_stream(a.data() + off, n);
            for (int i = 0; i < 128; i++) CHECK(a[i] == ((i >= off && i < off + n) ? 0.0f : 5.0f));
        }
    // randomised sweep (fixed seeds): volume, range, rows, chunk, mask density, packing, flags
    {
        std::mt19937 rng(2024);
        for (int t = 0; t < 40; t++) {
            const int64_t nvox = 64 + rng() % 4000;
            int64_t vbeg = rng() % (nvox / 2), vend = vbeg + 1 + rng() % (nvox - vbeg);
            if (rng() % 3 == 0) { vbeg = 0; vend = nvox; }
            const int rin = 1 + rng() % 5;
            std::vector<int> out_rows(1 + rng() % 3);
            for (auto &r : out_rows) r = 1 + rng() % 3;
            const int64_t chunk = 32 * (1 + rng() % 12);
            pipeline_case(nvox, vbeg, vend, rin, out_rows, chunk, (int)(rng() % 4), rng() % 2 != 0, rng() % 2 != 0, seed++, rng() % 2 != 0);
        }
    }
    printf("host_tier_check ok\n");
    return 0;
}
