import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/midvision-probe_amd")
import torch, torch.nn.functional as F
from evals.models.probes import SurfaceNormalHead
from mvp import functional as MF, dpt as mdpt
from oracle import probes as oprobes
dev = torch.device("cuda")
C, Hd, B, h, w = 128, 128, 2, 5, 6
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 77
g = torch.Generator().manual_seed(seed)
feats = [torch.randn(B, C, h, w, generator=g) for _ in range(4)]
probe = SurfaceNormalHead(feat_dim=[C] * 4, head_type="dpt", uncertainty_aware=True, hidden_dim=Hd, kernel_size=3)
sd = oprobes.make_dpt_weights([C] * 4, 4, hidden=Hd, k=3, seed=5)
probe.load_state_dict(sd, strict=True); probe = probe.to(dev)
pack = MF.pack_features([f.to(dev) for f in feats], probe.head.precision)
lq = mdpt.dpt_vit_logits(pack, probe.head, probe.head.precision)
ctx = lq.grad_fn
gy = torch.randn(lq.shape, generator=g)
(lq * gy.to(dev)).sum().backward(); torch.cuda.synchronize()

class MRelu(torch.autograd.Function):
    @staticmethod
    def forward(c, x, m): c.save_for_backward(m); return x.relu()
    @staticmethod
    def backward(c, g): return g * c.saved_tensors[0], None
def cl2nchw(m, H, W): return m.cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2).double()
H1, W1, H2, W2 = 2*h, 2*w, 8*h, 8*w
P = {n: t.clone().double().requires_grad_(True) for n, t in sd.items()}
f = [F.interpolate(F.conv2d(feats[i].double(), P[f"head.conv_{i}.weight"], P[f"head.conv_{i}.bias"]), scale_factor=2) for i in range(4)]
def rcu(x, pre, n):
    xP, aP, ma, mb = ctx.saved_rcu[n]
    a = MRelu.apply(F.conv2d(x, P[pre+"conv.0.weight"], P[pre+"conv.0.bias"], padding=1), cl2nchw(ma, H1, W1))
    return MRelu.apply(F.conv2d(a, P[pre+"conv.2.weight"], P[pre+"conv.2.bias"], padding=1), cl2nchw(mb, H1, W1)) + x
out = None
for n, (blk, unit) in enumerate(mdpt.RCU_ORDER):
    pre = f"head.ref_{blk}.resConfUnit{unit}."
    if unit == 1: out = rcu(f[blk], pre, n) + out
    elif blk == 3: out = rcu(f[3], pre, n)
    else: out = rcu(out, pre, n)
out = F.interpolate(out, scale_factor=4)
h0 = MRelu.apply(F.conv2d(out, P["head.out_conv.0.weight"], P["head.out_conv.0.bias"], padding=1), cl2nchw(ctx.m0, H2, W2))
y = F.conv2d(h0, P["head.out_conv.2.weight"], P["head.out_conv.2.bias"], padding=1)
(y * gy.permute(0, 3, 1, 2).double()).sum().backward()
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
print("logits", rel(lq.detach().permute(0, 3, 1, 2), y.detach()))
for n, p in probe.named_parameters():
    e = rel(p.grad, P[n].grad)
    if e > 5e-5 or "ref_1.resConfUnit1" in n: print(f"{n:45s} {e:.3e}")
print("done")
