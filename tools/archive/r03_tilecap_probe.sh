#!/bin/bash
# how long is the slowest tile of the velocity kernel when a tile holds 1 / 2 / 4 / all (default) creatures?  small population
# (one round of wavefronts), one step group: the launch time IS the slowest tile
set -u
O=gpurun_out/r03_tilecap; mkdir -p $O
for n in 4096 16384; do
for cap in 0 4 2 1; do
  REM2D_TILE_CREATURES=$cap timeout 600 python3 bench.py --envs $n --step-groups 1 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --min-time 0.3 > $O/b_${n}_$cap.json 2>> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/b_${n}_$cap.json')); c=d['config']; r=d['roofline']
print('envs $n cap $cap', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'vel4 launch %.4f ms, step sequence %.4f' % (r['avg_launch_ms'], r['avg_step_sequence_ms'] or 0))"
done
done
