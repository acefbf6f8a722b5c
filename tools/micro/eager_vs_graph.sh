#!/bin/bash
# The timed region with the frozen forwards launched eagerly (what jobs with more than one rank do by default) against hipGraph replay.
B="python bench.py --steps 60 --warmup 10 --no-alt-precision --no-serial-leg --no-cpu-baseline --no-roofline --sustained-steps 200"
for g in 1 0 1 0; do
  echo "## MVP_PIPELINE_GRAPHS=$g"
  MVP_PIPELINE_GRAPHS=$g $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['sustained'])"
done
