import sys, os, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sigma_amd as sg
from sigma_amd import problems as P
sg.init(0); dev=torch.device("cuda",0)
st=torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); sg.use_torch_stream(); sg.set_async(True)
def timed(fn,reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)*1e-3/reps
for name,(ptr,node,val),n in (("tridiag",P.tridiag_csr(10_000_000,2.0,-1.0,-1.0),10_000_000),("poisson2d",P.poisson2d_csr(3162,3162),3162*3162)):
    deg=np.diff(ptr); md=int(deg.max())
    nd=np.zeros((n,md),np.int32); vl=np.zeros((n,md))
    rows=np.repeat(np.arange(n),deg); slot=np.arange(len(node))-np.repeat(ptr[:-1]-1,deg)
    nd[rows,slot]=node; vl[rows,slot]=val
    last=nd[np.arange(n),deg-1]
    for k in range(md):
        m=k>=deg; nd[m,k]=last[m]
    A=sg.ellpack_matrix(n,n,nd,vl)
    x=torch.sin(0.001*torch.arange(1,n+1,dtype=torch.float64,device=dev)); y=torch.zeros(n,dtype=torch.float64,device=dev)
    t=timed(lambda:A.matvec(x,y)); b=12*n*md+16*n
    print(json.dumps({"ell":name,"n":n,"max_d":md,"us":t*1e6,"GBs":b/t/1e9,"frac":b/t/1e9/8000}))
