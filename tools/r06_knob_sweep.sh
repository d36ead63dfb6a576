#!/bin/bash
# Round 6: the scheduling knobs that were tuned under per-step launches (rounds 3-5), re-swept under the step train on config 3.
# All host-layer overrides (_lib.env_options); no result depends on them.
set -u
O=gpurun_out/r06_knobs; mkdir -p $O
A="--steps 100 --warmup 40 --settle 60 --no-cpu-baseline --no-secondary --min-time 2"
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py $A > $O/$name.json 2>/dev/null; python3 -c "
import json
try:
    d=json.load(open('$O/$name.json')); c=d['config']; print('$name', '%.2fM'%(d['value']/1e6), '%.4f ms/step'%d['ms_per_step'], 'err', c.get('solver_errors'))
except Exception as e: print('$name FAILED', e)"; }
run base X=1
run base2 X=1
for p in 0 1 4 7; do run prio$p REM2D_PRIO=$p; done
run prio5_t40_60 REM2D_PRIO_T1=40 REM2D_PRIO_T2=60
run prio5_t50_65 REM2D_PRIO_T1=50 REM2D_PRIO_T2=65
run prio5_t70_90 REM2D_PRIO_T1=70 REM2D_PRIO_T2=90
run prio5_t30_45 REM2D_PRIO_T1=30 REM2D_PRIO_T2=45
for r in 20 30 75 100 200; do run rebalance$r REM2D_REBALANCE=$r; done
run base3 X=1
