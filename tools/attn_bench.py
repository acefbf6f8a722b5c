#!/usr/bin/env python3
"""Attention microbenchmark + ablation (diagnostic).  Builds ablated copies of attention.hip under
gpurun_out/ and times the hot-path shape with HIP events, interleaved rounds in one process.
ablate: 0 full, 1 no softmax VALU, 2 no PV MFMAs, 3 no S MFMAs, 4 no K/V staging, 5 no tiles (prologue+epilogue)."""
import ctypes as C, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from mvp import lib, ops

CS = os.path.join(REPO, "midvision-probe_amd", "csrc")
OUT = os.path.join(REPO, "tools", "micro")  # built in the build container, travels with the tree
os.makedirs(OUT, exist_ok=True)


def build(ablate, extra="", src=None, tag=""):
    so = os.path.join(OUT, f"libattn_ab{ablate}{tag}.so")
    src = src or f"{CS}/attention.hip"
    so = so.replace(".so", extra.replace(" ", "").replace("-D", "_").replace("=", "") + ".so")
    # the product's flags (csrc/Makefile print-cxxflags: no packed fp32, no SLP vectoriser), so that what is timed is what ships
    fl = subprocess.run(["make", "-s", "-C", CS, "print-cxxflags"], capture_output=True, text=True, check=True).stdout.strip().replace("-I../../include", f"-I{REPO}/include")
    cmd = f"set -o pipefail; /opt/rocm/bin/hipcc {fl} -w -shared -I{CS} -DMVP_ATT_ABLATE={ablate} {extra} {src} -o {so} 2>&1 | {{ grep -v 'recognized feature' || true; }}"
    if not os.path.exists(so):  # (reused when already built in the build container)
        subprocess.run(["bash", "-c", cmd], check=True)
    l = C.CDLL(so)
    l.mvp_attention_fwd.argtypes = [C.POINTER(lib.AttentionArgs), C.c_void_p]
    l.mvp_attention_fwd.restype = C.c_int
    return l


def main():
    names = {0: "full", 1: "no_softmax", 2: "no_pv", 3: "no_s", 4: "no_stage", 5: "no_tiles"}
    variants = {names[a]: build(a) for a in names}
    prev = os.path.join(REPO, "tools", "micro", "attention_prev.hip")
    if "--ab" in sys.argv and os.path.exists(prev):  # A/B against a saved earlier version of the kernel
        variants = {"full": build(0), "prev": build(0, src=prev, tag="_prev")}
        for n in os.environ.get("SGB", "").split():  # the interleave ratio of the streaming kernel's S(t+1) / softmax(t) block
            variants[f"sgb{n}"] = build(0, extra=f"-DMVP_ATT_SGB={n}")
    if "--prio" in sys.argv:  # s_setprio around the MFMA clusters x stagger of waves 4-7 (MVP_ATT_PRIO / MVP_ATT_STAGGER)
        variants = {"base": build(0)}
        for pr, stg in ((1, 0), (0, 12), (0, 25), (1, 12), (1, 25), (1, 40)):
            variants[f"prio{pr}_stag{stg}"] = build(0, extra=f"-DMVP_ATT_PRIO={pr} -DMVP_ATT_STAGGER={stg}")
    # (tried in round 3 and dropped: delaying waves 4-7 of the resident kernel by 8 .. 48 x 64 cycles so that SIMD partners run out of
    # phase — 77.8 us -> 77.6 .. 80.4 at B = 96, 16.7 -> 16.8 .. 17.1 at B = 16: nothing)
    dev = torch.device("cuda")
    B = int(os.environ.get("B", 16)); H = 12
    for N in (197, 785, 1201):
        for prec in (3, 1):
            qkv = ops.split_bf16(torch.randn(B * N, 3 * H * 64, device=dev), 3)
            out = ops.empty_pair((B * N, H * 64), 3, dev)
            args = lib.AttentionArgs()
            vals = dict(qkv_hi=qkv[0].data_ptr(), qkv_lo=qkv[1].data_ptr(), out_hi=out[0].data_ptr(), out_lo=out[1].data_ptr(),
                        B=B, N=N, H=H, ld_qkv=3 * H * 64, ld_out=H * 64, scale=0.125, precision=prec)
            for k, v in vals.items():
                setattr(args, k, v)
            st = torch.cuda.current_stream().cuda_stream
            res = {}
            for rnd in range(3):
                for vn, l in variants.items():
                    for _ in range(3):
                        rc = l.mvp_attention_fwd(C.byref(args), st)
                        assert rc == 0, rc
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        l.mvp_attention_fwd(C.byref(args), st)
                    e1.record(); torch.cuda.synchronize()
                    res.setdefault(vn, []).append(e0.elapsed_time(e1) / 20 * 1e3)
            print(f"N={N} prec={prec}: " + "  ".join(f"{vn}={min(v):6.1f}us" for vn, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
