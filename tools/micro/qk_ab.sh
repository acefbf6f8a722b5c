#!/bin/bash
# Q.K^T as two f16 products over compensated fp16 pairs (MVP_ATT_QK=f16) against three bf16 products over bf16 pairs (pair), alternating on one box.
B="python bench.py --steps 20 --warmup 5 --no-alt-precision --no-serial-leg --no-cpu-baseline --no-roofline --sustained-steps 300"
for rep in 1 2 3; do
for v in pair f16; do
  echo "## MVP_ATT_QK=$v $*"
  MVP_ATT_QK=$v $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); s=d['sustained']; print('value', d['value'], 'sustained', s['value'])"
done
done
