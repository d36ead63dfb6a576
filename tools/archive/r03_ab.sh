#!/bin/bash
# A/B of an environment switch on the headline workload (no CPU baseline, no secondary): bash tools/r03_ab.sh <tag> VAR=a VAR=b ...
set -u
TAG=$1; shift
O=gpurun_out/r03_ab_$TAG; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0.3 > /dev/null 2>&1   # warm the box / spec cache
for kv in "$@"; do
  for rep in 1 2; do
    env $kv timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_${kv//[^A-Za-z0-9=_]/_}_$rep.json 2> $O/err.txt
    python3 -c "
import json; d=json.load(open('$O/bench_${kv//[^A-Za-z0-9=_]/_}_$rep.json')); c=d['config']; r=d['roofline']
print('$kv', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'kernel %.4f seq %.4f' % (r['avg_launch_ms'], r['avg_step_sequence_ms'] or 0), 'err', c['solver_errors'])"
  done
done
