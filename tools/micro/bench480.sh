#!/bin/bash
# 480x640 (BASELINE config #2 shape): pipeline variants.  Run through gpurun from the repo root.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
Q="--no-cpu-baseline --no-live-pmc --no-roofline --no-serial-leg --sustained-steps 48 --image-size 480x640 --steps 16 --warmup 4"
run() { echo "== $*"; env $ENVV python3 bench.py "$@" $Q 2>&1 | grep '^{' | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); p=d.get('pipeline',{}); print(d['value'], d['ms_per_step'], {k:p.get(k) for k in ('inflight','group','span_images','streams')}, 'sustained', (d.get('sustained') or {}).get('value'))
"; }
cd $ROOT
run
run --inflight 3
run --inflight 5
run --inflight 6
ENVV="MVP_PIPELINE_STREAMS=2" run --inflight 4
