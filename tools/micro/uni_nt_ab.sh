#!/bin/bash
# gemm_epilogue_uni's pair-only outputs nt sc1 (-DMVP_EPI_UNI_NT=1 whole-library build) against the shipped library, alternating on one box:
# ResNet-50 forward, DPT-probe step, the serial loop of the headline (tile kernels).
for rep in 1 2; do
for v in shipped uninn; do
  if [ $v = shipped ]; then unset MVP_LIB; else export MVP_LIB=$PWD/tools/micro/libmvp_hip_$v.so; fi
  echo "## $v"
  INFLIGHT=1 python tools/resnet_bench.py 2>/dev/null | grep "img/s" | head -1
  python bench.py --probe dpt --steps 10 --warmup 3 --no-alt-precision --no-serial-leg --no-cpu-baseline --no-roofline --sustained-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('dpt', d['value'])"
  python bench.py --inflight 1 --steps 20 --warmup 5 --no-alt-precision --no-serial-leg --no-cpu-baseline --no-roofline --sustained-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('serial', d['value'])"
done
done
