#!/bin/bash
# Does leaving a few CUs out of every large-M GEMM launch (MVP_PP_GRID, spans sized to that many CUs) pay beside the probe steps?  Alternating on one box.
B="python bench.py --steps 20 --warmup 5 --no-alt-precision --no-serial-leg --no-cpu-baseline --no-roofline --sustained-steps 300"
run() {  # grid span
  echo "## MVP_PP_GRID=${1:-unset} span=${2:-default}"
  if [ -n "$1" ]; then export MVP_PP_GRID=$1; else unset MVP_PP_GRID; fi
  $B ${2:+--span $2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); s=d['sustained']; print('value', d['value'], 'sustained', s['value'], 'span', d.get('pipeline',{}).get('span'))"
}
for rep in 1 2; do
  run "" ""
  run 248 106
  run 240 102
  run 248 110
done
