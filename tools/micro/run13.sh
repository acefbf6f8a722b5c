set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -x -q -k "attention or vit or backbone" > gpurun_out/r3_t13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t13.log
tail -4 gpurun_out/r3_t13.log
python3 - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, "midvision-probe_amd")
import torch
from mvp import lib, ops
dev = torch.device("cuda"); H = 12
libs = {"full": lib.load(), "no_tiles": C.CDLL("tools/micro/libattn_ab5.so")}
libs["no_tiles"].mvp_attention_fwd.argtypes = [C.POINTER(lib.AttentionArgs), C.c_void_p]; libs["no_tiles"].mvp_attention_fwd.restype = C.c_int
for B in (16, 96):
    for N in (197, 150):
        qkv = ops.split_bf16(torch.randn(B * N, 3 * H * 64, device=dev), 3)
        out = ops.empty_pair((B * N, H * 64), 3, dev)
        a = lib.AttentionArgs(qkv[0].data_ptr(), qkv[1].data_ptr(), out[0].data_ptr(), out[1].data_ptr(), B, N, H, 3 * H * 64, H * 64, 0.125, 3, 0)
        st = torch.cuda.current_stream().cuda_stream
        res = {}
        for rnd in range(3):
            for vn, l in libs.items():
                for _ in range(3): assert l.mvp_attention_fwd(C.byref(a), st) == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): l.mvp_attention_fwd(C.byref(a), st)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(vn, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        # repeatability (race screen): 50 launches must agree bit for bit
        l = libs["full"]; l.mvp_attention_fwd(C.byref(a), st); torch.cuda.synchronize(); base = (out[0].clone(), out[1].clone()); bad = 0
        for _ in range(50):
            out[0].zero_(); out[1].zero_(); l.mvp_attention_fwd(C.byref(a), st); torch.cuda.synchronize()
            bad += int(not (torch.equal(out[0], base[0]) and torch.equal(out[1], base[1])))
        print(f"B={B} N={N}: " + "  ".join(f"{k}={min(v):6.1f}us" for k, v in res.items()), f" | {bad} of 50 launches differ", flush=True)
PY
