import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/midvision-probe_amd")
import numpy as np, torch, torch.nn.functional as F
from mvp import functional as MF, ops, lib
def rel(a,b): a=a.double().cpu(); b=b.double().cpu(); return float((a-b).norm()/b.norm())
dev=torch.device("cuda")
g=torch.Generator().manual_seed(0)
B,C,h,w,K=3,128,14,14,256
feats=[torch.randn(B,C,h,w,generator=g) for _ in range(4)]
W=torch.randn(K,4*C,1,1,generator=g)*0.02; b=torch.randn(K,generator=g)*0.02
gy=torch.randn(B,1,4*h,4*w,generator=g)
# reference in double, commuted form with intermediates
Wd=W.double().requires_grad_(True); bd=b.double().requires_grad_(True)
Fd=torch.cat(feats,1).double()
l0=torch.einsum("bchw,kc->bkhw",Fd,Wd.reshape(K,-1))+bd[None,:,None,None]; l0.retain_grad()
lq=F.interpolate(l0,scale_factor=4,mode="bilinear"); lq.retain_grad()
bins=torch.linspace(0.001,10,K).double()
p=lq.relu()+0.1; p=p/p.sum(1,keepdim=True); d=torch.einsum("ikhw,k->ihw",p,bins).unsqueeze(1)
(d*gy.double()).sum().backward()
# ours
Wg=W.to(dev).requires_grad_(True); bg=b.to(dev).requires_grad_(True)
fg=[f.to(dev) for f in feats]
lqg=MF.linear_head_k1(fg,Wg,bg,lib.PREC_BF16X3); lqg.retain_grad()
dg=MF.depth_bins(lqg,K,0.001,10.0)
(dg*gy.to(dev)).sum().backward()
torch.cuda.synchronize()
print("fwd lq", rel(lqg.permute(0,3,1,2),lq), "depth", rel(dg,d))
print("grad lq", rel(lqg.grad.permute(0,3,1,2), lq.grad))
print("grad W", rel(Wg.grad, Wd.grad), "grad b", rel(bg.grad, bd.grad))
# isolate: dW from reference gl0 through our GEMM
gl0=l0.grad.permute(0,2,3,1).reshape(B*h*w,K).float().to(dev).contiguous()
pack=MF.pack_features(fg, lib.PREC_BF16X3)
gT=ops.zeros_pair((K,pack.Mpad),lib.PREC_BF16X3,dev)
ops.pack_nchw_tokens(gl0,1,pack.M,K,tok=gT,ld_tok=pack.Mpad,col_off=0)
dW=torch.empty(K,4*C,device=dev)
ops.gemm(gT,pack.tokT,K,4*C,pack.Mpad,out_f32=dW,precision=lib.PREC_BF16X3)
torch.cuda.synchronize()
print("dW from ref gl0", rel(dW, Wd.grad.reshape(K,-1)))
gTf=(gT[0].float()+gT[1].float()); print("gT", rel(gTf[:, :pack.M], l0.grad.permute(1,0,2,3).reshape(K,-1)))
tT=(pack.tokT[0].float()+pack.tokT[1].float()); print("tokT", rel(tT[:, :pack.M], Fd.permute(1,0,2,3).reshape(4*C,-1)))
# our gl0 vs ref
glq=lq.grad.permute(0,2,3,1).float().contiguous().to(dev)
gl0o=torch.empty(B*h*w,K,device=dev)
ops.resize(glq,gl0o,B,h,w,4*h,4*w,lib.RESIZE_BILINEAR,channels_last=True,Cdim=K,scale_h=4.0,scale_w=4.0,backward=True)
torch.cuda.synchronize()
print("gl0 via our resize_bwd of ref glq", rel(gl0o, gl0))
