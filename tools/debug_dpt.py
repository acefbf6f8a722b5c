import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/midvision-probe_amd")
import torch
from evals.models.probes import SurfaceNormalHead
from oracle import probes as oprobes
dev = torch.device("cuda")
C, Hd, B, h, w = 128, 128, 2, 5, 6
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 77)
feats = [torch.randn(B, C, h, w, generator=g) for _ in range(4)]
probe = SurfaceNormalHead(feat_dim=[C] * 4, head_type="dpt", uncertainty_aware=True, hidden_dim=Hd, kernel_size=3)
sd = oprobes.make_dpt_weights([C] * 4, 4, hidden=Hd, k=3, seed=int(sys.argv[2]) if len(sys.argv) > 2 else 5)
probe.load_state_dict(sd, strict=True); probe = probe.to(dev)
y = probe([f.to(dev) for f in feats])
sd_r = {n: t.clone().double().requires_grad_(True) for n, t in sd.items()}
y_ref = oprobes.snorm_head(sd_r, [f.double() for f in feats], "dpt", 3)
gy = torch.randn(y_ref.shape, generator=g)
(y_ref * gy.double()).sum().backward(); (y * gy.to(dev)).sum().backward(); torch.cuda.synchronize()
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
print("y", rel(y.detach(), y_ref.detach()))
for n, p in probe.named_parameters():
    print(f"{n:45s} {rel(p.grad, sd_r[n].grad):.3e}")
