#!/bin/bash
# Spans against whole-batch groups on the bench configurations.  Run through gpurun from the repo root.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
Q="--no-cpu-baseline --no-live-pmc --no-roofline"
run() { echo "== $*"; python3 bench.py "$@" $Q 2>&1 | grep '^{' | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); p=d.get('pipeline',{}); print(d['value'], d['ms_per_step'], {k:p.get(k) for k in ('inflight','group','span_images','streams','hipgraph_forward')}, 'serial', (p.get('serial') or {}).get('value'), 'sustained', (d.get('sustained') or {}).get('value'))
"; }
cd $ROOT
run --steps 20 --warmup 5
run --steps 20 --warmup 5 --span 104
MVP_PIPELINE_GRAPHS=0 run --steps 20 --warmup 5 --no-serial-leg
run --image-size 480x640 --steps 12 --warmup 4 --no-serial-leg --sustained-steps 60
run --image-size 480x640 --steps 12 --warmup 4 --span 0 --no-serial-leg --sustained-steps 60
run --batch 64 --steps 12 --warmup 4 --no-serial-leg --sustained-steps 60
run --batch 64 --steps 12 --warmup 4 --span 0 --no-serial-leg --sustained-steps 60
