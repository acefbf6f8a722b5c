set -o pipefail
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r3_fwdg_stats -- python3 $ROOT/tools/micro/fwd_group.py > $ROOT/gpurun_out/r3_fwdg_stats.log 2>&1
cd $ROOT
f=$(find gpurun_out/r3_fwdg_stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.2f} total_ms={float(r["TotalDurationNs"])/1e6:8.2f} {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -x -q > gpurun_out/r3_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t3.log
tail -4 gpurun_out/r3_t3.log
