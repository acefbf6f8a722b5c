#!/bin/bash
# Two forward chains side by side, each GEMM launch capped to a share of the CUs (MVP_PP_GRID), against the one-chain span pipeline.
# usage: tools/micro/two_chains.sh > gpurun_out/two_chains.txt
B="python bench.py --steps 40 --warmup 10 --no-alt-precision --no-serial-leg --no-cpu-baseline --no-roofline --sustained-steps 0"
run() { echo "## $*"; a=""; while [ "${1#--}" != "$1" ]; do a="$a $1 $2"; shift 2; done; env "$@" $B $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); p=d['pipeline']; print(d['value'], d['ms_per_step'], {k:p[k] for k in ('inflight','group','span_images','streams')})"; }
run X=1
run --inflight 4 --group 4 MVP_PIPELINE_STREAMS=2 MVP_PIPELINE_SPAN=0 MVP_PP_GRID=128
run --inflight 4 --group 4 MVP_PIPELINE_STREAMS=2 MVP_PIPELINE_SPAN=0
run --inflight 3 --group 4 MVP_PIPELINE_STREAMS=2 MVP_PIPELINE_SPAN=0 MVP_PP_GRID=128
run --inflight 4 --group 7 MVP_PIPELINE_STREAMS=2 MVP_PIPELINE_SPAN=0 MVP_PP_GRID=128
run --inflight 4 --group 7 MVP_PIPELINE_STREAMS=2 MVP_PIPELINE_SPAN=0 MVP_PP_GRID=160
