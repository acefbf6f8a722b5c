#!/bin/bash
# Cache policy of output stores that the writing kernel never reads back (whole-library builds, alternating on one box; forward-only rate and the bench line):
#   pair0   = -DMVP_EPI_AUX_PAIR=0: the wide epilogue's pair-only forms (qkv, fc1) with the default policy (the state before this experiment)
#   shipped = those stores nt sc1 (MVP_EPI_AUX_PAIR 18)
#   lnnt    = shipped + LayerNorm's output pair non-temporal (-DMVP_LN_NT=1);  lnattnt = lnnt + attention's output pair (-DMVP_ATT_NT=1)
B="python bench.py --steps 60 --warmup 10 --no-alt-precision --no-serial-leg --no-cpu-baseline --no-roofline --sustained-steps 200"
for rep in 1 2; do
for v in pair0 shipped lnnt lnattnt; do
  if [ $v = shipped ]; then unset MVP_LIB; else export MVP_LIB=$PWD/tools/micro/libmvp_hip_$v.so; fi
  echo "## $v"
  python tools/micro/fwd_rows.py 110 2>/dev/null | grep "T="
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['sustained']['value'])"
done
done
