set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/micro/group_probe.py > gpurun_out/r3_group_probe.log 2>&1; echo "rc=$?" >> gpurun_out/r3_group_probe.log
cat gpurun_out/r3_group_probe.log | grep -v Warning
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -x -q > gpurun_out/r3_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t2.log
tail -5 gpurun_out/r3_t2.log
