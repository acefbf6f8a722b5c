set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_b.json 2> gpurun_out/r3_bench_b.err; echo "rc=$?"
tail -3 gpurun_out/r3_bench_b.err
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_bench_b.json") if l.startswith("{")][-1])
r = d.pop("roofline")
print({k: d[k] for k in ("value", "ms_per_step", "sustained", "cpu_baseline")})
print(d["pipeline"]["serial"], d["pipeline"]["check"])
print({k: r[k] for k in r if k not in ("all_kernels", "hbm_kernels", "note", "regime", "traffic_unit")})
print(r["all_kernels"]); print(r["hbm_kernels"])
PY
