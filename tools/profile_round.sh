#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02            -> gpurun_out/<tag>_{stats,pmc_fetch,pmc_write,pmc_sq,pmc_tcc}/ , then
#   python tools/make_profiles.py <tag> ... copies the summaries into profiles/ (tracked).
# Counters are collected in their own passes with --kernel-trace only (never with sys/hip/hsa tracing), one group per pass
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ 8 slots, TCC 4; FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -eo pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
# --inflight 1 --tiles shared: the default run's kernel instantiations (shared-chip 128x128 GEMM tiles) as one serial chain, so a kernel's
# duration and counters are the kernel alone on the chip (what roofline.achieved in bench.py is quoted on).  The default run keeps two forwards in flight (mvp/pipeline.py): its trace is collected separately below.
ARGS="--no-cpu-baseline --no-roofline --sustained-steps 0 --steps 7 --warmup 3 --inflight 1 --tiles shared"
PARGS="--no-cpu-baseline --no-roofline --sustained-steps 0 --steps 8 --warmup 4 --no-serial-leg"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/${TAG}_pmc_tcc -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_pmc_tcc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_pipelined_stats -- python3 $ROOT/bench.py $PARGS > $OUT/${TAG}_pipelined_stats.log 2>&1
cd $ROOT
python3 tools/make_profiles.py $TAG $OUT/${TAG}_stats $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_tcc > $OUT/${TAG}_make_profiles.log 2>&1 || true
mkdir -p $OUT/${TAG}_profiles && cp profiles/${TAG}_* $OUT/${TAG}_profiles/ 2>/dev/null || true
cp $(find $OUT/${TAG}_pipelined_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_profiles/${TAG}_pipelined_kernel_stats.csv || true
python3 tools/step_trace.py $(find $OUT/${TAG}_pipelined_stats -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_profiles/${TAG}_pipelined_step_trace.txt || true
echo "profile passes done: $(ls $OUT | grep ${TAG}_ | wc -l) entries"
