for v in default tn_1_32 tn_2_32 tn_2_64; do
  if [ $v = default ]; then unset MVP_LIB; else export MVP_LIB=$PWD/tools/micro/libmvp_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python - <<'PY'
import os, sys
sys.path.insert(0, "midvision-probe_amd")
import torch
from mvp import lib, ops, conv
dev = torch.device("cuda")
for (B, H, C, Co) in ((16, 28, 512, 512), (16, 56, 256, 256), (16, 112, 256, 128), (8, 64, 512, 512)):
    M = B * H * H
    x = ops.split_bf16(torch.randn(M, C, device=dev), 3); g = ops.split_bf16(torch.randn(M, max(Co, 128), device=dev) * 1e-2, 3)
    geo = conv.geom(B, H, H, C, 3, 3, 1, 1)
    dw = torch.empty(Co, C, 3, 3, device=dev)
    best = 1e9
    for rnd in range(3):
        f = lambda: conv.conv_dw(g, max(Co, 128), x, C, geo, Co, dw, precision=3)
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
    fl = 2.0 * M * C * Co * 9
    print(f"  B={B} {H}x{H} Cin={C} Cout={Co}: {best:7.1f} us  {fl / best / 1e6:5.0f} TF/s")
PY
done
