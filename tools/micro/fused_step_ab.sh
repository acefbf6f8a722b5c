#!/bin/bash
# The tape-free probe step (mvp/fused_step.py) against the autograd step, alternating on one box: the driver's window (20 steps after 5) and 200 sustained steps.
B="python bench.py --steps 20 --warmup 5 --no-alt-precision --no-serial-leg --no-cpu-baseline --no-roofline --sustained-steps 200"
for rep in 1 2 3; do
for v in 0 1; do
  echo "## MVP_FUSED_STEP=$v"
  MVP_FUSED_STEP=$v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); s=d['sustained']; print('value', d['value'], 'ms/step', d['ms_per_step'], 'sustained', s['value'], 'host_work_ms_per_step', s.get('host_work_ms_per_step'), 'host_enqueue', s.get('host_enqueue_ms_per_step'))"
done
done
