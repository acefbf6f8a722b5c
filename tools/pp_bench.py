#!/usr/bin/env python3
"""Ping-pong 256x256 bf16x3 GEMM (csrc/gemm_pp.hip): correctness against fp64 and against the tile kernels (bit-identity),
then timing, interleaved rounds in one process, on random operands.

  python tools/pp_bench.py [--check-only] [--B 64] [--ablate]

Variants timed: `tile` = mvp_gemm_bias_act_res (ALONE rule), `shared` = the same with MVP_TILES_SHARED, `pp_sep` = the new kernel on
separate hi / lo arrays, `pp_ilv` = the new kernel on hi|lo-interleaved operands (re-laid out here with torch)."""
import ctypes as C, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from mvp import lib, ops

CS = os.path.join(REPO, "midvision-probe_amd", "csrc")
OUT = os.path.join(REPO, "gpurun_out")


def interleave(pair):
    """(hi, lo) [R, K] -> one [R, 2K] array, hi | lo interleaved per 32-deep k block."""
    hi, lo = pair
    R, K = hi.shape
    return torch.stack((hi.view(R, K // 32, 32), lo.view(R, K // 32, 32)), dim=2).reshape(R, 2 * K).contiguous()


def build_ablate(n, define="MVP_PP_ABLATE", tag="ab"):
    """Ablated copies of gemm_pp.hip, built with the product's flags (csrc/Makefile print-cxxflags) into tools/micro/ — build them in
    the build container (they travel to the GPU box with the tree); an existing one is reused."""
    so = os.path.join(REPO, "tools", "micro", f"libpp_{tag}{n}.so")
    if not os.path.exists(so):
        fl = subprocess.run(["make", "-s", "-C", CS, "print-cxxflags"], capture_output=True, text=True, check=True).stdout.strip()
        fl += " " + subprocess.run(["make", "-s", "-C", CS, "print-ppflags"], capture_output=True, text=True, check=True).stdout.strip()  # gemm_pp.hip's own
        fl = fl.replace("-I../../include", f"-I{REPO}/include")
        cmd = f"set -o pipefail; /opt/rocm/bin/hipcc {fl} -shared -I{CS} -D{define}={n} -w {CS}/gemm_pp.hip -o {so} 2>&1 | {{ grep -v 'recognized feature' || true; }}"
        subprocess.run(["bash", "-c", cmd], check=True)
    l = C.CDLL(so)
    l.mvp_gemm_pp.argtypes = [C.POINTER(lib.GemmArgs), C.c_void_p]
    l.mvp_gemm_pp.restype = C.c_int
    return l


PREC = 2 if "--f16x2" in sys.argv else 3  # --f16x2: MVP_PREC_F16X2 (two fp16 products; timing / stamps only: the operands here are bf16 pairs)


def main():
    dev = torch.device("cuda")
    L = lib.load()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    ok = True

    def mk(m, n, k, *, bias=True, act=0, residual=False, pair_out=True, f32_out=False):
        a = ops.split_bf16(torch.randn(m, k, device=dev), 3)
        w = ops.split_bf16(torch.randn(n, k, device=dev) * 0.05, 3)
        b = torch.randn(n, device=dev) if bias else None
        r = torch.randn(m, n, device=dev) if residual else None
        return dict(a=a, w=w, ai=interleave(a), wi=interleave(w), bias=b, res=r, act=act, m=m, n=n, k=k, pair_out=pair_out, f32_out=f32_out)

    def args_for(d, layout, out, o32):
        """layout: mvp_gemm_args.pair_layout — bit 0: A interleaved, bit 1: W interleaved."""
        layout = int(layout)
        ia, iw = bool(layout & 1), bool(layout & 2)
        k = d["k"]
        g = lib.GemmArgs(d["ai"].data_ptr() if ia else d["a"][0].data_ptr(), None if ia else d["a"][1].data_ptr(),
                         d["wi"].data_ptr() if iw else d["w"][0].data_ptr(), None if iw else d["w"][1].data_ptr(),
                         lib.ptr(d["bias"]), lib.ptr(d["res"]), lib.ptr(o32), lib.ptr(out[0]) if out else None, lib.ptr(out[1]) if out else None,
                         d["m"], d["n"], k, 2 * k if ia else k, 2 * k if iw else k, d["n"], d["n"], d["n"], d["act"], PREC, 0, 0, 0, 0)
        g.pair_layout = layout
        return g

    def run(d, which, out, o32, libpp=None):
        if which in ("tile", "shared"):
            g = args_for(d, False, out, o32)
            g.tile_policy = (1 if which == "shared" else 0) | 2  # MVP_TILES_NO_PP: the tile kernels themselves
            rc = L.mvp_gemm_bias_act_res(C.byref(g), st)
        else:
            layout = {"sep": 0, "ailv": 1, "wilv": 2, "ilv": 3}[which.rsplit("_", 1)[1]]
            g = args_for(d, layout, out, o32)
            rc = (libpp or L).mvp_gemm_pp(C.byref(g), st)
        if rc != 0:
            raise RuntimeError(f"{which}: rc {rc}")

    # ------------------------------------------------------------------ correctness
    cases = [(256, 256, 64, {}), (300, 520, 96, dict(act=1)), (3152, 768, 768, dict(residual=True, pair_out=False, f32_out=True)),
             (3152, 2304, 768, {}), (1000, 3072, 768, dict(act=1)), (777, 768, 3072, dict(residual=True, f32_out=True)),
             (12608, 768, 768, dict(residual=True, pair_out=False, f32_out=True)),
             # more tiles than CUs: the persistent tile loop (765 / 1020 tiles of pair output, 516 with the residual epilogue, ragged last
             # row tile and a ragged last column tile)
             (21670, 2304, 768, {}), (21670, 3072, 768, dict(act=1)), (43900, 768, 768, dict(residual=True, pair_out=False, f32_out=True)),
             (17000, 1800, 96, dict(residual=True, f32_out=True))]
    for m, n, k, kw in ([] if PREC == 2 else cases):  # (--f16x2: timing only; tests/test_gpu_kernels.py checks that mode)
        d = mk(m, n, k, **kw)
        A = (d["a"][0].double() + d["a"][1].double())
        W = (d["w"][0].double() + d["w"][1].double())
        ref = A @ W.t() + d["bias"].double()
        if d["act"] == 1:
            ref = torch.nn.functional.gelu(ref)
        if d["res"] is not None:
            ref = ref + d["res"].double()
        outs = {}
        for which in ("tile", "pp_sep", "pp_ilv", "pp_wilv", "pp_ailv"):
            out = ops.empty_pair((m, n), 3, dev) if d["pair_out"] else None
            o32 = torch.empty(m, n, device=dev) if d["f32_out"] else None
            if out:
                out[0].fill_(float("nan")); out[1].fill_(float("nan"))
            if o32 is not None:
                o32.fill_(float("nan"))
            run(d, which, out, o32)
            torch.cuda.synchronize()
            val = o32.double() if o32 is not None else (out[0].double() + out[1].double())
            err = ((val - ref).abs().max() / ref.abs().max()).item()
            outs[which] = (out, o32, err)
        same = {}
        for which in ("pp_sep", "pp_ilv", "pp_wilv", "pp_ailv"):
            o, o32, _ = outs[which]
            t, t32, _ = outs["tile"]
            eq = True
            if o32 is not None:
                eq = eq and torch.equal(o32, t32)
            if o is not None:
                eq = eq and torch.equal(o[0], t[0]) and torch.equal(o[1], t[1])
            same[which] = eq
        good = all(v[2] < 2e-5 for v in outs.values()) and all(same.values())
        ok = ok and good
        print(f"check M={m} N={n} K={k} {kw}: rel-max-err tile {outs['tile'][2]:.2e} pp_sep {outs['pp_sep'][2]:.2e} pp_ilv {outs['pp_ilv'][2]:.2e}; "
              f"bit-identical to tile: {same}  {'OK' if good else 'FAIL'}", flush=True)
    # reproducibility / race screen: the same launch many times must give the same bits
    for (rm, rn, rk) in ((3152, 768, 3072), (21670, 768 * 2, 768)):  # one tile per workgroup / two to three tiles per workgroup
        d = mk(rm, rn, rk, residual=True, pair_out=False, f32_out=True)
        for which in ("pp_sep", "pp_ilv"):
            base = torch.empty(rm, rn, device=dev)
            run(d, which, None, base)
            bad = 0
            for _ in range(200):
                o = torch.empty(rm, rn, device=dev)
                run(d, which, None, o)
                bad += int(not torch.equal(o, base))
            ok = ok and bad == 0
            print(f"repro {which} M={rm} N={rn} K={rk}: {bad} of 200 launches differ", flush=True)
    print("CHECK", "PASS" if ok else "FAIL", flush=True)
    if "--build-only" in sys.argv:
        return 0
    if "--check-only" in sys.argv or (not ok and "--zeros" not in sys.argv and PREC == 3):
        return 0 if ok else 1

    if "--stamp" in sys.argv:  # in-kernel s_memtime breakdown of the phases (diagnostic build, MVP_PP_STAMP)
        lst = build_ablate(1, "MVP_PP_STAMP", "stamp")
        names = ["issue (reads + DMA)", "counted waits", "barrier 1", "MFMA cluster", "barrier 2"]
        for B in (96, 110):
            M = B * 197
            for name, n, k, kw in (("qkv", 2304, 768, {}), ("fc1", 3072, 768, dict(act=1)), ("fc2", 768, 3072, dict(residual=True, pair_out=False, f32_out=True))):
                for ilv in (1,):
                    d = mk(M, n, k, **kw)
                    out = ops.empty_pair((M, n), 3, dev) if d["pair_out"] else None
                    o32 = torch.empty(M, n, device=dev) if d["f32_out"] else None
                    tiles = ((M + 255) // 256) * ((n + 255) // 256)
                    grid = min(tiles, 256)  # the persistent launch: one workgroup per CU
                    per_wg = torch.tensor([len(range(w, tiles, grid)) for w in range(grid)], dtype=torch.float64, device=dev)
                    dbg = torch.zeros(grid * 2 * 16, dtype=torch.int64, device=dev)
                    g = args_for(d, 3 if ilv else 0, out, o32)
                    g.splitk_ws, g.splitk_ws_bytes = dbg.data_ptr(), dbg.numel() * 8
                    for _ in range(3):
                        rc = lst.mvp_gemm_pp(C.byref(g), st)
                    torch.cuda.synchronize()
                    assert rc == 0
                    raw = dbg.view(grid, 2, 16).double()
                    t = raw[:, :, :10].reshape(grid, 2, 2, 5) / (k // 32) / per_wg.view(grid, 1, 1, 1)  # cycles per phase, [workgroup, group, phase type, segment]
                    mean = t.mean(dim=0)
                    print(f"stamp B={B} {name} ilv={ilv} tiles={tiles} workgroups={grid}: cycles per phase (mean over workgroups; group 0 | group 1)")
                    for ph in range(2):
                        print(f"   P{ph + 1}: " + "  ".join(f"{names[c]}={mean[0, ph, c]:.0f}|{mean[1, ph, c]:.0f}" for c in range(5))
                              + f"   total={mean[0, ph].sum():.0f}|{mean[1, ph].sum():.0f}", flush=True)
                    # per TILE: what precedes the main loop (cold prologue of the first tile, tile switch of the others), the main loop, the epilogue
                    pro, loop, epi = ((raw[:, :, c] / per_wg.view(grid, 1)).mean().item() for c in (10, 11, 12))
                    lifes = (raw[:, 0, 14] - raw[:, 0, 13])
                    life = lifes.mean().item()
                    print(f"   per tile: prologue / tile switch {pro:.0f} cycles, main loop {loop:.0f}, epilogue {epi:.0f}; workgroup lifetime {life:.0f} cycles "
                          f"for {per_wg.mean().item():.2f} tiles", flush=True)
                    # what bounds the launch is the SLOWEST workgroup: lifetime spread, and the spread of the end times on the 100 MHz
                    # real-time counter (comparable across XCDs, unlike s_memtime)
                    full = per_wg == per_wg.max()
                    lf = lifes[full]
                    ends = raw[:, 0, 15]
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        lst.mvp_gemm_pp(C.byref(g), st)
                    e1.record(); torch.cuda.synchronize()
                    wall = e0.elapsed_time(e1) / 10 * 1e3
                    raw2 = dbg.view(grid, 2, 16).double()
                    ends = raw2[:, 0, 15]
                    print(f"   lifetime of the workgroups with {int(per_wg.max().item())} tiles: min {lf.min().item():.0f}  median {lf.median().item():.0f}  max {lf.max().item():.0f} cycles "
                          f"(max / mean {lf.max().item() / lf.mean().item():.3f}); end times spread over {(ends.max() - ends.min()).item() * 0.01:.1f} us; "
                          f"launch wall time (stamp build) {wall:.1f} us = {life / wall / 1e3:.2f} GHz if the mean lifetime filled it", flush=True)
        return 0

    # ------------------------------------------------------------------ timing
    Bs = [int(x) for x in (sys.argv[sys.argv.index("--B") + 1].split(",") if "--B" in sys.argv else ["16", "64", "96", "110"])]
    variants = ["tile", "pp_sep", "pp_wilv", "pp_ailv", "pp_ilv"] if "--ilv-only" not in sys.argv else ["pp_ilv"]
    extra = {}
    if "--ablate" in sys.argv:
        for n, nm in ((1, "no_mfma"), (2, "no_dma"), (3, "no_read")):
            extra[nm] = build_ablate(n)
    if "--prio" in sys.argv:  # priority forms of the main loop (MVP_PP_PRIO): static priority for waves 4-7 / no priority instructions
        extra["static_prio"] = build_ablate(1, "MVP_PP_PRIO", "prio")
        extra["no_prio"] = build_ablate(2, "MVP_PP_PRIO", "prio")
    if "--store-policy" in sys.argv:  # cache policy of the wide epilogue's output stores (MVP_EPI_AUX): nt, sc0 sc1, nt sc1
        extra["stores_default"] = build_ablate(0, "MVP_EPI_AUX", "aux")  # (the shipped build: pair-only forms nt sc1, the rest default)
        extra["stores_nt"] = build_ablate(2, "MVP_EPI_AUX", "aux")
        extra["stores_sc0_sc1"] = build_ablate(17, "MVP_EPI_AUX", "aux")
        extra["stores_nt_sc1"] = build_ablate(18, "MVP_EPI_AUX", "aux")
    if "--persist-ab" in sys.argv:  # the tile loop of round 4: one tile per workgroup (no loop) / the loop without the prefetch ahead of the epilogue
        extra["one_tile_per_wg"] = build_ablate(1, "MVP_PP_NOLOOP", "noloop")
        extra["loop_cold_prologue"] = build_ablate(0, "MVP_PP_PREFETCH", "prefetch")
        extra["loop_strict_first_wait"] = build_ablate(0, "MVP_PP_RELAXED", "relaxed")
        r03 = os.path.join(REPO, "tools", "micro", "libpp_r03.so")  # round 3's kernel (git show 6fdc23d:.../gemm_pp.hip + gemm_epilogue.h), built in the build container
        if os.path.exists(r03) and PREC == 3:  # (round 3's kernel has no two-product mode)
            l3 = C.CDLL(r03)
            l3.mvp_gemm_pp.argtypes = [C.POINTER(lib.GemmArgs), C.c_void_p]
            l3.mvp_gemm_pp.restype = C.c_int
            extra["round3_kernel"] = l3
    if "--two-products" in sys.argv:  # what a 2-MFMA-per-product precision mode would cost (the third product dropped: wrong results, timing only)
        extra["two_products"] = build_ablate(4)
    if "--epilogue-ab" in sys.argv:  # the generic epilogue (gemm_epilogue) instead of the wide one, same main loop
        extra["generic_epilogue"] = build_ablate(0, "MVP_PP_WIDE_EPILOGUE", "wide")
    for B in Bs:
        M = B * 197
        for name, n, k, kw in (("qkv", 2304, 768, {}), ("proj", 768, 768, dict(residual=True, pair_out=False, f32_out=True)),
                               ("fc1", 3072, 768, dict(act=1)), ("fc2", 768, 3072, dict(residual=True, pair_out=False, f32_out=True))):
            d = mk(M, n, k, **kw)
            if "--zeros" in sys.argv:  # all-zero operands: the same instruction stream at a fraction of the switching power (DVFS check)
                for t in (d["a"][0], d["a"][1], d["w"][0], d["w"][1], d["ai"], d["wi"]):
                    t.zero_()
            out = ops.empty_pair((M, n), 3, dev) if d["pair_out"] else None
            o32 = torch.empty(M, n, device=dev) if d["f32_out"] else None
            res = {}
            todo = [(v, None) for v in variants] + [(nm + "_ilv", l) for nm, l in extra.items()]
            for rnd in range(3):
                for vn, lpp in todo:
                    for _ in range(2):
                        run(d, vn, out, o32, lpp)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        run(d, vn, out, o32, lpp)
                    e1.record(); torch.cuda.synchronize()
                    res.setdefault(vn, []).append(e0.elapsed_time(e1) / 10 * 1e3)
            fl = 2.0 * M * n * k
            print(f"B={B:3d} {name:5s} M={M} N={n} K={k}: " + "  ".join(f"{vn}={min(v):7.1f}us({fl / min(v) / 1e6:4.0f}TF)" for vn, v in res.items()), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
