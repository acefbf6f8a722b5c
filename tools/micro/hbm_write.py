#!/usr/bin/env python3
"""HBM write / read-modify-write ceilings on this box (diagnostic for the GEMM epilogue bursts): torch fill, copy, in-place add at 58-232 MB."""
import torch, time
dev = torch.device("cuda:0")
for mb in (58, 174, 232, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev); z = torch.randn(n, device=dev)
    def t(fn, it=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e-3
    tf = t(lambda: x.fill_(1.0)); tc = t(lambda: y.copy_(z)); ta = t(lambda: x.add_(1.0)); tz = t(lambda: torch.add(z, 1.0, out=y))
    b = n * 4
    print(f"{mb:5d} MB: fill {b / tf / 1e12:.2f} TB/s written | copy {2 * b / tc / 1e12:.2f} TB/s (r+w) | in-place add {2 * b / ta / 1e12:.2f} TB/s (r+w) | out-of-place add {2 * b / tz / 1e12:.2f} TB/s")
