set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/pp_bench.py --B 96 --epilogue-ab > gpurun_out/pp_bench6.log 2>&1; grep "CHECK\|^B=" gpurun_out/pp_bench6.log
timeout -k 10 300 python tools/micro/group_probe.py > gpurun_out/r3_group_probe3.log 2>&1; grep -v Warn gpurun_out/r3_group_probe3.log | grep "G=6\|probe step"
