// host_probe: what the host tier can get from this box: threaded memcpy pageable -> pinned, hipHostRegister, pinned H2D / D2H alone and together
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = (size_t)1 << 30;
    char *page = (char *)malloc(N); memset(page, 1, N);
    char *pin = nullptr, *pin2 = nullptr; hipHostMalloc((void **)&pin, N, hipHostMallocDefault); hipHostMalloc((void **)&pin2, N, hipHostMallocDefault); memset(pin, 2, N); memset(pin2, 3, N);
    char *dev = nullptr, *dev2 = nullptr; hipMalloc((void **)&dev, N); hipMalloc((void **)&dev2, N);
    printf("hardware threads %u\n", std::thread::hardware_concurrency());
    for (int T : {1, 2, 4, 8, 16, 32, 64}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&, t] { size_t a = N / T * t, b = t == T - 1 ? N : N / T * (t + 1); memcpy(pin + a, page + a, b - a); });
            for (auto &x : th) x.join();
            best = std::min(best, now() - t0);
        }
        printf("memcpy pageable->pinned %2d threads: %.1f GB/s\n", T, N / best / 1e9);
    }
    { char *p2 = (char *)malloc(N); memset(p2, 1, N); double t0 = now(); hipError_t e = hipHostRegister(p2, N, hipHostRegisterDefault); double t1 = now(); printf("hipHostRegister 1 GiB: %.1f ms (%s)\n", (t1 - t0) * 1e3, hipGetErrorString(e)); if (e == hipSuccess) { t0 = now(); hipMemcpy(dev, p2, N, hipMemcpyHostToDevice); printf("  H2D from registered: %.1f GB/s\n", N / (now() - t0) / 1e9); hipHostUnregister(p2); } free(p2); }
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); double t1 = now();
        hipMemcpyAsync(pin2, dev2, N, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2); double t2 = now();
        hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s1); hipMemcpyAsync(pin2, dev2, N, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s1); hipStreamSynchronize(s2); double t3 = now();
        printf("pinned H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f GB/s total\n", N / (t1 - t0) / 1e9, N / (t2 - t1) / 1e9, 2.0 * N / (t3 - t2) / 1e9);
    }
    { double t0 = now(); hipMemcpy(dev, page, N, hipMemcpyHostToDevice); printf("pageable H2D %.1f GB/s\n", N / (now() - t0) / 1e9); t0 = now(); hipMemcpy(page, dev, N, hipMemcpyDeviceToHost); printf("pageable D2H %.1f GB/s\n", N / (now() - t0) / 1e9); }
    return 0;
}
