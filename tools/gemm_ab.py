#!/usr/bin/env python3
"""A/B of two gemm.hip sources in ONE process (interleaved rounds): current tree vs a saved copy."""
import ctypes as C, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from mvp import lib, ops
OUT = os.path.join(REPO, "gpurun_out")
def build(src_dir, inc, tag):
    so = os.path.join(OUT, f"libgemm_{tag}.so")
    subprocess.run(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I{inc} -I{src_dir} {src_dir}/gemm.hip -o {so}", shell=True, check=True)
    return C.CDLL(so)
class OldArgs(C.Structure):
    _fields_ = lib.GemmArgs._fields_[:23]
new = build(os.path.join(REPO, "midvision-probe_amd", "csrc"), os.path.join(REPO, "include"), "new")
old = build(os.path.join(REPO, "tools", "_oldsrc"), os.path.join(REPO, "tools", "_oldsrc"), "old")
dev = torch.device("cuda"); M = 16 * 197
for name, m, n, k in [("qkv", M, 2304, 768), ("proj", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072)]:
    a = ops.split_bf16(torch.randn(m, k, device=dev), 3); w = ops.split_bf16(torch.randn(n, k, device=dev) * 0.05, 3)
    out = ops.empty_pair((m, n), 3, dev); bias = torch.randn(n, device=dev); o32 = torch.empty(m, n, device=dev)
    vals = (a[0].data_ptr(), a[1].data_ptr(), w[0].data_ptr(), w[1].data_ptr(), bias.data_ptr(), None, None, out[0].data_ptr(), out[1].data_ptr(), m, n, k, k, k, n, n, n, 0, 3, 0, 0, 0, 0)
    an, ao = lib.GemmArgs(*vals), OldArgs(*vals)
    st = torch.cuda.current_stream().cuda_stream
    res = {"new": [], "old": []}
    for rnd in range(5):
        for tag, l, ar in (("new", new, an), ("old", old, ao)):
            for _ in range(3): l.mvp_gemm_bias_act_res(C.byref(ar), st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): l.mvp_gemm_bias_act_res(C.byref(ar), st)
            e1.record(); torch.cuda.synchronize()
            res[tag].append(e0.elapsed_time(e1) / 30 * 1e3)
    print(name, {t: f"min {min(v):.1f} med {sorted(v)[len(v)//2]:.1f}" for t, v in res.items()}, flush=True)
