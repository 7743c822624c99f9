import torch, sys
sys.path.insert(0,'/root/repo')
from deeppointmap_amd import ops
torch.manual_seed(0)
dev='cuda:0'
for R,Cin,Cout in [(1000,128,256),(32768,256,256),(4097,32,128),(262144,128,32),(777,512,64),(5000,100,128),(64,768,256)]:
    x=torch.randn(R,Cin,device=dev); W=torch.randn(Cout,Cin,device=dev)/Cin**0.5; b=torch.randn(Cout,device=dev)
    g=torch.rand(Cout,device=dev)+0.5; be=torch.randn(Cout,device=dev); pre=torch.randn(R,Cout,device=dev); post=torch.randn(R,Cout,device=dev)
    for act in (0,1):
        for pr,po in ((None,None),(pre,None),(pre,post),(None,post)):
            y=ops.linear_layernorm(x,W,b,g,be,act=act,pre=pr,post=po)
            ref=ops.layernorm(ops.linear(x,W,b,residual=pr),g,be,act=act,post=po)
            t=torch.nn.functional.layer_norm((x.double()@W.double().T+b.double()+(pr.double() if pr is not None else 0)),(Cout,),g.double(),be.double(),1e-5)
            if po is not None: t=t+po.double()
            if act==1: t=t.relu()
            e1=float((y-ref).abs().max()); e2=float((y.double()-t).abs().max()); e3=float((ref.double()-t).abs().max())
            assert e1<2e-5 and e2<2e-5, (R,Cin,Cout,act,e1,e2,e3)
    print(R,Cin,Cout,'ok', e1,e2,e3)
import time
x=torch.randn(32768,256,device=dev); W=torch.randn(256,256,device=dev)/16; b=torch.randn(256,device=dev); g=torch.ones(256,device=dev); be=torch.zeros(256,device=dev); pre=torch.randn(32768,256,device=dev)
def tm(f,n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
print('fused us', tm(lambda: ops.linear_layernorm(x,W,b,g,be,pre=pre)), 'split us', tm(lambda: ops.layernorm(ops.linear(x,W,b,residual=pre),g,be)))
