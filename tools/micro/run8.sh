set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t8.log
tail -5 gpurun_out/r3_t8.log
