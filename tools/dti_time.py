import sys, time; sys.path.insert(0,'.')
import torch, fibers_jl_amd as fj
from fibers_jl_amd import phantom
import ctypes as C
dev=torch.device('cuda',0); SHAPE=(140,140,140); nvox=140**3
b2,g2=phantom.scheme_dti(60,4,1000.0,seed=2)
d2,_=phantom.make_dwi_torch(SHAPE,b2,g2,seed=2,device=dev,nfib=1)
mask=torch.ones(nvox,dtype=torch.uint8,device=dev)
p2=fj.DtiPlan(b2,g2,device=0); o2=fj.dti_fit_device(p2,d2,mask)
L=fj.lib(); torch.cuda.synchronize()
L.fib_profile_enable(1); L.fib_profile_reset()
for _ in range(20): fj.dti_fit_device(p2,d2,mask,out=o2)
torch.cuda.synchronize()
ms,n=C.c_double(0),C.c_int64(0); L.fib_profile_get(b"dti_fit",C.byref(ms),C.byref(n))
k=ms.value/n.value; by=(4*64+1+64)*nvox
print("dti_fit kernel %.1f us  %.2f TB/s  %.1f%% of 8 TB/s"%(k*1e3, by/k/1e9, by/k/1e9/80))
print("checksum", float(o2["fa"].double().sum()), float(o2["eigval1"].double().sum()))
