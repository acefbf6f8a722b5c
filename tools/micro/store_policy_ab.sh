#!/bin/bash
# Cache policy of loads / stores whose data the kernel touches once (whole-library builds, alternating on one box; forward-only rate and the bench line).
# usage: tools/micro/store_policy_ab.sh shipped attld lnld resld   (variant = tools/micro/libmvp_hip_<variant>.so; shipped = the product library)
B="python bench.py --steps 60 --warmup 10 --no-alt-precision --no-serial-leg --no-cpu-baseline --no-roofline --sustained-steps 200"
for rep in 1 2; do
for v in "$@"; do
  if [ $v = shipped ]; then unset MVP_LIB; else export MVP_LIB=$PWD/tools/micro/libmvp_hip_$v.so; fi
  echo "## $v"
  python tools/micro/fwd_rows.py 110 2>/dev/null | grep "T="
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['sustained']['value'])"
done
done
