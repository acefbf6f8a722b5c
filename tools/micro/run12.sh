set -o pipefail
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-roofline --no-live-pmc --sustained-steps 0"
run() { echo "== $*" >> gpurun_out/r3_b12.log; timeout -k 10 400 python bench.py $F "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['pipeline']; print(d['value'], d['ms_per_step'], {k: p[k] for k in ('inflight','group','streams','gemm_tiles')}, p.get('serial'))
    elif 'Error' in l or 'rror:' in l: print(l.strip())
" >> gpurun_out/r3_b12.log; }
rm -f gpurun_out/r3_b12.log
run --image-size 480x640 --steps 12 --warmup 4
MVP_PIPELINE_STREAMS=1 run --image-size 480x640 --steps 12 --warmup 4 --inflight 2
run --image-size 480x640 --steps 12 --warmup 4 --inflight 1
run --batch 64 --steps 12 --warmup 4
run --probe dpt --steps 10 --warmup 3
run --probe dpt --steps 10 --warmup 3 --inflight 2 --group 6
run --precision bf16 --steps 20 --warmup 5
cat gpurun_out/r3_b12.log
