set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/pp_bench.py --B 96 --epilogue-ab > gpurun_out/pp_bench5.log 2>&1; tail -14 gpurun_out/pp_bench5.log
timeout -k 10 300 python tools/pp_bench.py --stamp > gpurun_out/pp_stamp3.log 2>&1; grep -A3 "^stamp B=96" gpurun_out/pp_stamp3.log | grep "stamp\|per workgroup"
