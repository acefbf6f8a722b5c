#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r03     -> gpurun_out/<tag>_* , summaries copied into gpurun_out/<tag>_profiles/ (then commit them under profiles/)
# Counters are collected in their own passes with --kernel-trace only (never with sys/hip/hsa tracing), one group per pass
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ 8 slots, TCC 4; FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -eo pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
# the default bench command (what the driver runs), without its CPU legs and without the nested PMC children
DEF="--steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --sustained-steps 0"
# the timed run's pipeline as the PMC child runs it: grouped forwards, eager launches, 1 + 2 groups of steps, nothing else
CHILD="--pmc-child"
# the serial leg: one batch at a time on one stream (its own tile set: the 64x64 family)
SERIAL="--steps 8 --warmup 3 --inflight 1 --no-cpu-baseline --no-live-pmc --no-roofline --sustained-steps 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $ROOT/bench.py $DEF > $OUT/${TAG}_stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_child_stats -- python3 $ROOT/bench.py $CHILD > $OUT/${TAG}_child_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $ROOT/bench.py $CHILD > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $ROOT/bench.py $CHILD > $OUT/${TAG}_pmc_write.log 2>&1
echo "traffic passes done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 $ROOT/bench.py $CHILD > $OUT/${TAG}_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/${TAG}_pmc_tcc -- python3 $ROOT/bench.py $CHILD > $OUT/${TAG}_pmc_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_grbm -- python3 $ROOT/bench.py $CHILD > $OUT/${TAG}_pmc_grbm.log 2>&1 || true
echo "counter passes done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_serial_stats -- python3 $ROOT/bench.py $SERIAL > $OUT/${TAG}_serial_stats.log 2>&1
cd $ROOT
python3 tools/make_profiles.py $TAG $OUT/${TAG}_child_stats $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_tcc $OUT/${TAG}_pmc_grbm > $OUT/${TAG}_make_profiles.log 2>&1 || true
mkdir -p $OUT/${TAG}_profiles && cp profiles/${TAG}_* $OUT/${TAG}_profiles/ 2>/dev/null || true
cp $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_profiles/${TAG}_default_bench_kernel_stats.csv || true
cp $(find $OUT/${TAG}_serial_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_profiles/${TAG}_serial_kernel_stats.csv || true
python3 tools/step_trace.py $(find $OUT/${TAG}_serial_stats -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_profiles/${TAG}_serial_step_trace.txt || true
python3 tools/micro/chain_timeline.py 7 2 1 110 > $OUT/${TAG}_profiles/${TAG}_chain_timeline.txt 2>/dev/null || true
echo "profile passes done: $(ls $OUT/${TAG}_profiles | wc -l) files"
