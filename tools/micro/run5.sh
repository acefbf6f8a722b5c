set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -x -q > gpurun_out/r3_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t5.log
tail -6 gpurun_out/r3_t5.log
timeout -k 10 300 python tools/micro/group_probe.py > gpurun_out/r3_group_probe2.log 2>&1; grep -v Warn gpurun_out/r3_group_probe2.log | head -16
