#!/bin/bash
# What the driver runs at round end, in one gpurun call: the whole -m gpu suite, smoke(), the default bench line.
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/validate_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/validate_tests.log
tail -3 gpurun_out/validate_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/validate_bench.json 2> gpurun_out/validate_bench.err; echo "bench rc=$?"
python3 -c "
import json
d = json.loads([l for l in open('gpurun_out/validate_bench.json') if l.startswith('{')][-1])
r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'sustained', d['sustained']['value'], 'serial', d['pipeline']['serial']['value'], 'check', d['pipeline']['check'])
print('roofline', r['kernel'], r['achieved'], r['frac'], 'alone', r['kernel_alone']['frac'], 'chip', r['chip_level']['frac'], 'traffic', r['traffic'], 'busy', r['mfma_busy'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['variants'])
"
