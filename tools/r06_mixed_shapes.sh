#!/bin/bash
# Round 6 experiment: a tile shape per lane bucket in ONE train launch -- the heavy 16-lane bucket on 64-lane tiles (short chains), the
# light buckets on 128-lane tiles (fewer wave-instructions) -- against the uniform policies.  REM2D_TILE_SHAPE_BY_LANES is the Python host
# layer's experiment knob (gym_rem2d_amd/env.py).
set -u
O=gpurun_out/r06_mixed; mkdir -p $O
A="--steps 100 --warmup 40 --settle 60 --no-cpu-baseline --no-secondary --min-time 2"
run() { name=$1; shift; env "$@" timeout 400 python3 bench.py $A $EXTRA > $O/$name.json 2>/dev/null; python3 -c "
import json
try:
    d=json.load(open('$O/$name.json')); c=d['config']; print('$name', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], c.get('launch'), 'err', c.get('solver_errors'))
except Exception as e: print('$name FAILED', e)"; }
EXTRA=""
run n65536_base X=1
run n65536_light128 REM2D_TILE_SHAPE_BY_LANES=2:1,4:1,8:1
run n65536_light128_24 REM2D_TILE_SHAPE_BY_LANES=2:1,4:1
EXTRA="--envs 131072"
run n131072_policy X=1
run n131072_heavy64 REM2D_TILE_SHAPE_BY_LANES=16:3
run n131072_all64 REM2D_TILE_SHAPE=3
EXTRA="--envs 196608"
run n196608_policy X=1
run n196608_heavy64_train REM2D_TILE_SHAPE_BY_LANES=16:3 REM2D_FUSE_VELPOST=2
EXTRA="--envs 262144"
run n262144_policy X=1
run n262144_heavy64_train REM2D_TILE_SHAPE_BY_LANES=16:3 REM2D_FUSE_VELPOST=2
