set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/pp_bench.py --B 96 > gpurun_out/pp_bench4.log 2>&1; tail -14 gpurun_out/pp_bench4.log
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "warmup or timed or grouped or prefetcher" > gpurun_out/r3_t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t4.log
tail -4 gpurun_out/r3_t4.log
