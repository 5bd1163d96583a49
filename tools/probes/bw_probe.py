"""HBM calibration on the box: device copy / fill / read-reduce rates for a 1.5 GB buffer (torch kernels)."""
import torch, time
dev = torch.device("cuda", 0)
n = 1544120016 // 4
a = torch.rand(n, device=dev); b = torch.empty_like(a)
def t(f, rep=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep
gb = n * 4 / 1e9
ms = t(lambda: b.copy_(a)); print("copy  %.3f ms  %.0f GB/s (read+write)" % (ms, 2 * gb / ms * 1e3))
ms = t(lambda: b.zero_()); print("fill  %.3f ms  %.0f GB/s" % (ms, gb / ms * 1e3))
ms = t(lambda: a.sum()); print("sum   %.3f ms  %.0f GB/s" % (ms, gb / ms * 1e3))
ms = t(lambda: torch.add(a, 1.0, out=b)); print("add   %.3f ms  %.0f GB/s (read+write)" % (ms, 2 * gb / ms * 1e3))
