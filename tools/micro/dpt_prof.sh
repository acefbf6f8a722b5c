#!/bin/bash
# DPT-probe step: kernel stats of the serial step (one batch at a time: every kernel alone) + the default bench line.  Run through gpurun.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
Q="--no-cpu-baseline --no-live-pmc --sustained-steps 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dpt_stats -- python3 $ROOT/bench.py --probe dpt --steps 6 --warmup 2 --inflight 1 --no-roofline --no-serial-leg $Q > $OUT/dpt_stats.log 2>&1
cp $(find $OUT/dpt_stats -name "*kernel_stats.csv" | head -1) $OUT/dpt_kernel_stats.csv
cd $ROOT
python3 bench.py --probe dpt --steps 14 --warmup 3 $Q > $OUT/dpt_default.log 2>&1
python3 bench.py --probe dpt --prediction sigdepth --steps 14 --warmup 3 $Q > $OUT/dpt_sigdepth.log 2>&1
echo done
