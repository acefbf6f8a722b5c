#!/bin/bash
# Secondary workloads of a round (run through gpurun from the repo root): bench lines + rocprofv3 kernel stats.
set -eo pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
python3 bench.py --image-size 480x640 --no-cpu-baseline --sustained-steps 0 --steps 10 > $OUT/${TAG}_bench_480x640.json 2> $OUT/${TAG}_bench_480x640.err
python3 bench.py --probe dpt --no-cpu-baseline --sustained-steps 0 --steps 10 > $OUT/${TAG}_bench_dpt.json 2> $OUT/${TAG}_bench_dpt.err
python3 bench.py --batch 64 --no-cpu-baseline --sustained-steps 0 --steps 10 > $OUT/${TAG}_bench_b64.json 2> $OUT/${TAG}_bench_b64.err
python3 tools/resnet_bench.py > $OUT/${TAG}_resnet_bench.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_dpt_stats -- python3 $ROOT/bench.py --probe dpt --no-cpu-baseline --no-roofline --sustained-steps 0 --steps 5 --warmup 2 > $OUT/${TAG}_dpt_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_resnet_stats -- python3 $ROOT/tools/resnet_bench.py > $OUT/${TAG}_resnet_stats.log 2>&1
cd $ROOT
mkdir -p $OUT/${TAG}_profiles
cp $(find $OUT/${TAG}_dpt_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_profiles/${TAG}_dpt_probe_kernel_stats.csv
cp $(find $OUT/${TAG}_resnet_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_profiles/${TAG}_resnet50_forward_kernel_stats.csv
python3 tools/step_trace.py $(find $OUT/${TAG}_dpt_stats -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_profiles/${TAG}_dpt_probe_step_trace.txt
for f in 480x640 dpt b64; do cp $OUT/${TAG}_bench_$f.json $OUT/${TAG}_profiles/; done
cp $OUT/${TAG}_resnet_bench.txt $OUT/${TAG}_profiles/
echo secondary done
