set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_dist.py tests/test_gpu_probe.py tests/test_gpu_pipeline.py tests/test_gpu_wrappers.py -x -q -k "rccl or two_backwards or streamk or hf_vitmae or two_rank" > gpurun_out/r3_t9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t9.log
tail -12 gpurun_out/r3_t9.log
