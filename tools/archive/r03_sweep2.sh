#!/bin/bash
set -u
O=gpurun_out/r03_sweep2; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0.3 > /dev/null 2>&1
run() { tag=$1; shift; timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary "$@" > $O/b_$tag.json 2>> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/b_$tag.json')); c=d['config']; r=d['roofline']
print('$tag', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'groups', c['step_groups'], 'kernel %.4f seq %.4f' % (r['avg_launch_ms'], r['avg_step_sequence_ms'] or 0))"; }
for g in 3 4 5 6 8; do run groups$g --step-groups $g; done
for c in 16 8 4; do REM2D_TILE_CREATURES=$c run cap$c; done
REM2D_HEAVY_PER_WAVE=1 run heavy1
REM2D_HEAVY_PER_WAVE=4 run heavy4
