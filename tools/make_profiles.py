#!/usr/bin/env python3
"""Copy the judged rocprofv3 evidence from gpurun_out/ (scratch) into profiles/ (tracked):
  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of the default bench.py run
  profiles/<tag>_step_trace.txt       per-step per-kernel totals (tools/step_trace.py)
  profiles/<tag>_pmc_summary.txt      mean PMC counters per dispatch per kernel (separate --pmc passes)
  profiles/<tag>_pmc_traffic.json     per kernel: HBM-side bytes per launch, corrected as MI355X_MICROARCH.md prescribes
                                      (FETCH_SIZE is KiB and reports 1/2 of wide coalesced reads on gfx950 -> x2; WRITE_SIZE KiB exact)
usage: make_profiles.py <tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> [<pmc_other_dir> ...]"""
import csv, glob, json, os, re, shutil, subprocess, sys, collections

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats_dir, fetch_dir, write_dir, *others = sys.argv[1:]
out = os.path.join(REPO, "profiles")
os.makedirs(out, exist_ok=True)
ks = glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(ks, os.path.join(out, f"{tag}_kernel_stats.csv"))
kt = glob.glob(stats_dir + "/**/*kernel_trace.csv", recursive=True)[0]
txt = subprocess.run([sys.executable, os.path.join(REPO, "tools", "step_trace.py"), kt], capture_output=True, text=True).stdout
open(os.path.join(out, f"{tag}_step_trace.txt"), "w").write(txt)
txt = subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"), fetch_dir, write_dir, *others], capture_output=True, text=True).stdout
open(os.path.join(out, f"{tag}_pmc_summary.txt"), "w").write(
    "# mean per dispatch; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (uncorrected); one --pmc pass per counter group\n" + txt)


def per_kernel(d, counter):
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
            name = re.sub(r"^void ", "", name)
            name = re.sub(r"\(mvp_\w+( const\*)?(, [^)]*)?\)$", "", name)
            acc[name] += float(row["Counter_Value"]); n[name] += 1
    return {k: acc[k] / n[k] for k in acc}, n


f, nf = per_kernel(fetch_dir, "FETCH_SIZE")
w, _ = per_kernel(write_dir, "WRITE_SIZE")
traffic = {}
for k in f:
    if k.startswith("at::") or "rocclr" in k:
        continue
    fb, wb = 2.0 * f[k] * 1024.0, w.get(k, 0.0) * 1024.0
    traffic[k] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "hbm_bytes": round(fb + wb), "dispatches": nf[k]}
json.dump({"workload": "bench.py default (B=16, 224x224, linear probe; precision = the bench default of that round: bf16x3 through round 3, f16x2 from round 4)", "correction": "FETCH_SIZE[KiB] x 1024 x 2 + WRITE_SIZE[KiB] x 1024",
           "per_launch": traffic}, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(out)))
