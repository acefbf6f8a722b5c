#!/bin/bash
# Round-4 evidence for the BASELINE configs other than the headline (run through gpurun from the repo root):
#   tools/profile_secondary4.sh r04 [part]     part = a (bench lines) | b (rocprofv3 kernel stats) | all
# Bench lines carry `roofline`; every profiled command is the plain program behind `--` (no env / shell hop: the profiler's preload
# initialises the GPU before the program starts).  Summaries land in gpurun_out/<tag>_profiles/ (then commit them under profiles/).
set -eo pipefail
TAG=${1:-r04}
PART=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
P=$OUT/${TAG}_profiles
mkdir -p $P
cd $ROOT
if [ "$PART" = "a" ] || [ "$PART" = "all" ]; then
  # config #2 (480x640, linear depth probe): default pipeline, and spans of 18 images on one stream
  python3 bench.py --image-size 480x640 --no-cpu-baseline --sustained-steps 60 --steps 12 > $P/${TAG}_bench_480x640.json 2> $OUT/${TAG}_bench_480x640.err
  python3 bench.py --image-size 480x640 --span 18 --no-cpu-baseline --no-live-pmc --sustained-steps 60 --steps 12 > $P/${TAG}_bench_480x640_span18.json 2> $OUT/${TAG}_bench_480x640_span18.err || true
  echo "480x640 done"
  python3 bench.py --probe dpt --no-cpu-baseline --no-live-pmc --sustained-steps 30 --steps 10 > $P/${TAG}_bench_dpt.json 2> $OUT/${TAG}_bench_dpt.err
  python3 bench.py --batch 64 --no-cpu-baseline --no-live-pmc --sustained-steps 60 --steps 10 > $P/${TAG}_bench_b64.json 2> $OUT/${TAG}_bench_b64.err
  python3 bench.py --h2d --no-cpu-baseline --no-live-pmc --sustained-steps 100 > $P/${TAG}_bench_h2d.json 2> $OUT/${TAG}_bench_h2d.err
  echo "dpt / b64 / h2d done"
  python3 tools/resnet_bench.py > $P/${TAG}_resnet_bench.txt 2> $OUT/${TAG}_resnet_bench.err
  python3 tools/fullsize_smoke.py 3 4 > $P/${TAG}_fullsize_configs.txt 2> $OUT/${TAG}_fullsize.err
  python3 tools/micro/spair_probe.py >> $P/${TAG}_fullsize_configs.txt 2>> $OUT/${TAG}_fullsize.err
  echo "resnet / configs 3 4 5 done"
fi
if [ "$PART" = "b" ] || [ "$PART" = "all" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_480_stats -- python3 $ROOT/bench.py --image-size 480x640 --no-cpu-baseline --no-live-pmc --no-roofline --no-serial-leg --sustained-steps 0 --steps 6 --warmup 2 > $OUT/${TAG}_480_stats.log 2>&1
  echo "480x640 stats done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_resnet_stats -- python3 $ROOT/tools/resnet_bench.py > $OUT/${TAG}_resnet_stats.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cfg3_stats -- python3 $ROOT/tools/fullsize_smoke.py 3 > $OUT/${TAG}_cfg3_stats.log 2>&1
  echo "resnet / config 3 stats done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cfg4_stats -- python3 $ROOT/tools/fullsize_smoke.py 4 > $OUT/${TAG}_cfg4_stats.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cfg5_stats -- python3 $ROOT/tools/micro/spair_probe.py > $OUT/${TAG}_cfg5_stats.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_dpt_stats -- python3 $ROOT/bench.py --probe dpt --no-cpu-baseline --no-live-pmc --no-roofline --no-serial-leg --sustained-steps 0 --steps 5 --warmup 2 > $OUT/${TAG}_dpt_stats.log 2>&1
  echo "configs 4 5 / dpt stats done"
  cd $ROOT
  for n in 480 resnet cfg3 cfg4 cfg5 dpt; do
    f=$(find $OUT/${TAG}_${n}_stats -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $P/${TAG}_${n}_kernel_stats.csv
  done
fi
echo "secondary done: $(ls $P | wc -l) files"
