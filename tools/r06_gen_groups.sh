#!/bin/bash
# Round 6: the generation workload (whole episodes, 131 072 individuals on one GPU) by train shape x step groups -- two ranks sharing the
# GPU with 65 536 each ran at 96.9 M where one rank with 131 072 runs at 71 M: is it the second dispatch queue?
set -u
O=gpurun_out/r06_gen_groups; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 400 python3 bench.py --workload generation --no-cpu-baseline $EXTRA > $O/$name.json 2>/dev/null; python3 -c "
import json
try:
    d=json.load(open('$O/$name.json')); c=d['config']; print('$name', '%.2fM'%(d['value']/1e6), 'steps', d['steps'], '%.1f ms total'%(c['blocks_ms'][0]), c.get('launch'), 'groups', c.get('step_groups'), 'err', c.get('solver_errors'))
except Exception as e: print('$name FAILED', e)"; }
EXTRA=""
run policy X=1
run s3_g1 REM2D_TILE_SHAPE=3 REM2D_STEP_GROUPS=1
run s3_g2 REM2D_TILE_SHAPE=3 REM2D_STEP_GROUPS=2
run s3_g4 REM2D_TILE_SHAPE=3 REM2D_STEP_GROUPS=4
run s1_g2 REM2D_STEP_GROUPS=2
run s1_g4 REM2D_STEP_GROUPS=4
run steps_g4 REM2D_FUSE_VELPOST=1
EXTRA="--envs 65536"
run n65536_g1 X=1
run n65536_g2 REM2D_STEP_GROUPS=2
