set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -x -q > gpurun_out/r3_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t1.log
tail -5 gpurun_out/r3_t1.log
F="--no-cpu-baseline --no-roofline --no-live-pmc --sustained-steps 0 --no-serial-leg"
for cfg in "" "--group 1" "--group 6 --inflight 3" "--group 4" "--group 3 --inflight 3"; do
  for st in 2 1; do
    echo "== cfg: $cfg streams=$st" >> gpurun_out/r3_b1.log
    MVP_PIPELINE_STREAMS=$st timeout -k 10 300 python bench.py --steps 20 --warmup 5 $F $cfg 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['pipeline'])
    elif 'Error' in l or 'error' in l: print(l.strip())
" >> gpurun_out/r3_b1.log
  done
done
echo "== 60 steps default" >> gpurun_out/r3_b1.log
timeout -k 10 300 python bench.py --steps 60 --warmup 6 $F 2>&1 | grep "^{" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['pipeline'])" >> gpurun_out/r3_b1.log
cat gpurun_out/r3_b1.log
