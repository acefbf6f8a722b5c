#!/usr/bin/env python3
"""GEMM microbenchmark + ablation (diagnostic).  Builds ablated copies of gemm.hip into
gpurun_out/ and times the hot-path shapes with HIP events, interleaved rounds in one process."""
import ctypes as C, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from mvp import lib, ops

CS = os.path.join(REPO, "midvision-probe_amd", "csrc")
OUT = os.path.join(REPO, "gpurun_out")
os.makedirs(OUT, exist_ok=True)

def build(ablate, extra="", src=None, tag=""):
    so = os.path.join(OUT, f"libgemm_ab{ablate}{extra.replace(' ','').replace('-D','_').replace('=','')}{tag}.so")
    src = src or f"{CS}/gemm.hip"
    cmd = f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I{REPO}/include -I{CS} -DMVP_ABLATE={ablate} {extra} {src} {CS}/gemm_sk.hip -o {so}"
    subprocess.run(cmd, shell=True, check=True)
    l = C.CDLL(so)
    l.mvp_gemm_bias_act_res.argtypes = [C.POINTER(lib.GemmArgs), C.c_void_p]
    l.mvp_gemm_bias_act_res.restype = C.c_int
    return l

def main():
    variants = {"auto": build(0)}
    def cfg(bm, bn, bk, st, nw=4, wnw=2):
        return build(0, f"-DMVP_F_BM={bm} -DMVP_F_BN={bn} -DMVP_F_BK={bk} -DMVP_F_ST={st} -DMVP_F_NW={nw} -DMVP_F_WNW={wnw}")
    if "--tiles" in sys.argv:
        for c in ((64,128,64,1,4,2),(64,128,64,1,8,4),(64,64,64,1,4,2),(64,64,64,1,8,4),(128,64,64,1,8,2),(128,128,64,1,8,4),(128,128,64,1,8,2),(64,256,64,1,8,4)):
            try:
                variants["%dx%dk%ds%dw%dn%d" % c] = cfg(*c)
            except Exception as e:
                print("build failed", c)
    prev = os.path.join(REPO, "tools", "micro", "gemm_prev.hip")
    if "--ab" in sys.argv and os.path.exists(prev):  # A/B against a saved earlier version of gemm.hip
        variants = {"auto": variants["auto"], "prev": build(0, src=prev, tag="_prev")}
    if "--sk" in sys.argv:
        variants = {"auto": variants["auto"], "streamk_spec": build(0, "-DMVP_SK_SPEC=1"), "streamk_il": build(0, "-DMVP_SK_SPEC=0 -DMVP_SK_DMA_INTERLEAVE=1"),
                    "streamk_burst": build(0, "-DMVP_SK_SPEC=0 -DMVP_SK_DMA_INTERLEAVE=0")}
    if "--ablate" in sys.argv:
        variants.update({"no_load": build(2), "load_only": build(3), "epi_only": build(4)})
    dev = torch.device("cuda")
    B = int(os.environ.get("B", 16)); M = B * 197
    shapes = [("qkv", M, 2304, 768), ("proj", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072), ("head", B * 196, 256, 3072)]
    for prec in (3, 1):
        for name, m, n, k in shapes:
            a = ops.split_bf16(torch.randn(m, k, device=dev), 3); w = ops.split_bf16(torch.randn(n, k, device=dev) * 0.05, 3)
            out = ops.empty_pair((m, n), 3, dev); o32 = torch.empty(m, n, device=dev); bias = torch.randn(n, device=dev)
            args = lib.GemmArgs(a[0].data_ptr(), a[1].data_ptr(), w[0].data_ptr(), w[1].data_ptr(), bias.data_ptr(), None,
                                None, out[0].data_ptr(), out[1].data_ptr(), m, n, k, k, k, n, n, n, 0, prec, 0, 0, 0, 0)
            st = torch.cuda.current_stream().cuda_stream
            skws = ops._streamk_workspace(dev)
            sk_args = lib.GemmArgs(a[0].data_ptr(), a[1].data_ptr(), w[0].data_ptr(), w[1].data_ptr(), bias.data_ptr(), None,
                                   None, out[0].data_ptr(), out[1].data_ptr(), m, n, k, k, k, n, n, n, 0, prec, 0, 0, 0, 0)
            sk_args.splitk, sk_args.splitk_ws, sk_args.splitk_ws_bytes = -1, skws.data_ptr(), skws.numel()
            base_args = args
            res = {}
            for rnd in range(3):
                for vn, l in variants.items():
                    args = sk_args if vn.startswith("streamk") else base_args
                    for _ in range(3): l.mvp_gemm_bias_act_res(C.byref(args), st)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20): l.mvp_gemm_bias_act_res(C.byref(args), st)
                    e1.record(); torch.cuda.synchronize()
                    res.setdefault(vn, []).append(e0.elapsed_time(e1) / 20 * 1e3)
            fl = 2.0 * m * n * k
            line = f"prec={prec} {name:5s} M={m} N={n} K={k}: " + "  ".join(f"{vn}={min(v):6.1f}us({fl/min(v)/1e6:4.0f})" for vn, v in res.items())
            print(line, flush=True)

if __name__ == "__main__":
    main()
