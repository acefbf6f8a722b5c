import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/midvision-probe_amd")
import torch
import test_gpu_configs as T
worst = {}
orig = T._grad_check
def spy(probe, ref_sd, rel=5e-2, cos=2e-3):
    mr, mc = 0, 0
    for n, p in probe.named_parameters():
        a, b = p.grad.double().cpu().flatten(), ref_sd[n].grad.double().flatten()
        mr = max(mr, float((a - b).norm() / b.norm().clamp_min(1e-30)))
        mc = max(mc, 1 - float(a @ b / (a.norm() * b.norm()).clamp_min(1e-30)))
    print(f"   grad worst rel-L2 {mr:.2e} (limit {rel:.0e})  worst 1-cos {mc:.2e} (limit {cos:.0e})")
    orig(probe, ref_sd, rel, cos)
T._grad_check = spy
dev = torch.device("cuda:0")
for name in ("test_config2_dino_vitb16_linear_bindepth_480x640_whole_step", "test_config3_mocov3_resnet50_dpt_snorm_step", "test_config4_mae_vitb16_dpt_bindepth_step"):
    print(name); getattr(T, name)(dev)
