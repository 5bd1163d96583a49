import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import fibers_jl_amd as fj
from fibers_jl_amd import phantom
SHAPE=(140,140,140); dev=torch.device("cuda",0); L=fj.lib()
bval,bvec=phantom.scheme_dti(60,4,1000.0,2)
dwi,_=phantom.make_dwi_torch(SHAPE,bval,bvec,2,dev,nfib=1)
plan=fj.DtiPlan(bval,bvec)
o=fj.dti_fit_device(plan,dwi,torch.ones(140**3,dtype=torch.uint8,device=dev))
bm=phantom.ball_mask_torch(SHAPE,dev)
field,mout=fj.stream_field_device([o["eigvec1"]],fa=o["fa"],fa_thresh=0.1,mask=bm)
seeds=torch.nonzero(mout).flatten()
sub=torch.tensor([[0.1,-0.2,0.3]],dtype=torch.float32,device=dev)
del dwi
xyz={}
def xyz_out(n):
    if xyz.get("t") is None or xyz["t"].numel()<3*n: xyz["t"]=torch.empty(int(3*n*1.05)+16,dtype=torch.float32,device=dev)
    return xyz["t"]
for _ in range(5): fj.stream_device(field,SHAPE,seeds,sub,xyz_out=xyz_out)
torch.cuda.synchronize()
L.fib_profile_enable(1); L.fib_profile_reset()
for _ in range(30): fj.stream_device(field,SHAPE,seeds,sub,xyz_out=xyz_out)
torch.cuda.synchronize(); L.fib_profile_enable(0)
out=[]
for nm in (b"stream_trace",b"stream_pack",b"stream_scan"):
    ms,cnt=C.c_double(),C.c_int64(); L.fib_profile_get(nm,C.byref(ms),C.byref(cnt))
    out.append("%s %.3f ms"%(nm.decode(), ms.value/max(cnt.value,1)))
print(os.path.basename(os.environ.get("FIBERS_HIP_LIB","libfibers_hip.so")), " | ".join(out))
