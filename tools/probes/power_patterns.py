import subprocess, sys, threading, time, torch
def smi(tag):
    o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    sclk = [l.split("(")[-1].split(")")[0] for l in o.split("\n") if "sclk" in l]
    pw = [l.split(":")[-1].strip() for l in o.split("\n") if "Power (W)" in l]
    print("[%s] sclk %s  power %s W" % (tag, sclk[:1], pw[:1]), flush=True)
dev = torch.device("cuda", 0)
N = 140**3
x = torch.rand(270, N, device=dev)
y = torch.empty(321, N, device=dev)
x60 = x[:60]
cases = {
  "planar column sum [60][N] -> [N]": (lambda: torch.sum(x60, 0), 60*4.0*N),
  "planar column sum [270][N] -> [N]": (lambda: torch.sum(x, 0), 270*4.0*N),
  "linear sum of the same 270 N floats": (lambda: x.view(-1).sum(), 270*4.0*N),
  "broadcast write [N] -> [321][N]": (lambda: y.copy_(x[0].expand(321, N)), 321*4.0*N),
}
for name,(fn,nb) in cases.items():
    fn(); torch.cuda.synchronize(); stop=False
    def sampler():
        k=0
        while not stop:
            time.sleep(1.0)
            if not stop: smi("%s, %d s"%(name,k+1))
            k+=1
    th=threading.Thread(target=sampler); th.start()
    t0=time.perf_counter(); it=0
    while time.perf_counter()-t0<3.2:
        for _ in range(20): fn()
        torch.cuda.synchronize(); it+=20
    el=time.perf_counter()-t0; stop=True; th.join()
    print("%s: %.3f ms, %.2f TB/s"%(name, el/it*1e3, nb*it/el/1e12), flush=True)
