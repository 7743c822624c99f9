import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from deeppointmap_amd import ops
dev='cuda'
def t(B,M,N):
    q=torch.randn(B*M,256,device=dev); k=torch.randn(B*N,256,device=dev); v=torch.randn(B*N,256,device=dev)
    o=torch.empty(B*M,256,device=dev)
    for _ in range(3): ops.attention(q,k,v,B,M,N,8,out=o)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.attention(q,k,v,B,M,N,8,out=o)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/20*1e3
for N in (64,128,256,512,1024):
    print(N, round(t(64,256,N),1), 'us')
