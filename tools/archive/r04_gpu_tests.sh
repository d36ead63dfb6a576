#!/bin/bash
# GPU test tier + the driver's bench command.   gpurun --timeout 1800 -- 'bash tools/r04_gpu_tests.sh <tag> [pytest -k expr]'
set -u
TAG=${1:-a}
KEXPR=${2:-}
O=gpurun_out/r04_tests_$TAG; mkdir -p $O
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/pytest_gpu.txt 2>&1
else
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
fi
tail -25 $O/pytest_gpu.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_20.json 2> $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench_default_20.json')); c=d['config']
print('%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'blocks', c['blocks'], 'first %.2f med %.2f' % (c['block_ms_first'], c['block_ms_median']), {k: round(v['value']/1e6,1) for k,v in (d.get('secondary') or {}).items()}, d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:120])"
tail -5 $O/bench.err
