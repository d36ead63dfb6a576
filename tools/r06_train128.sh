#!/bin/bash
# Round 6: the 128-lane step train against per-step launches (gpurun -- 'bash tools/r06_train128.sh').  REM2D_FUSE_VELPOST=1 is the
# Python host layer's override (_lib.env_options) that keeps per-step launches for the 128-lane shapes.
set -u
O=gpurun_out/r06_train128; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
for n in 131072 196608 262144; do
  timeout 400 python3 bench.py --workload generation --envs $n --no-cpu-baseline > $O/gen_${n}_train.json 2>/dev/null
  REM2D_FUSE_VELPOST=1 timeout 400 python3 bench.py --workload generation --envs $n --no-cpu-baseline > $O/gen_${n}_steps.json 2>/dev/null
done
timeout 300 python3 bench.py --workload chain8 --no-cpu-baseline > $O/chain8_train.json 2>/dev/null
REM2D_FUSE_VELPOST=1 timeout 300 python3 bench.py --workload chain8 --no-cpu-baseline > $O/chain8_steps.json 2>/dev/null
REM2D_TILE_SHAPE=3 timeout 300 python3 bench.py --workload chain8 --no-cpu-baseline > $O/chain8_train64.json 2>/dev/null
for n in 131072 196608; do
  timeout 400 python3 bench.py --envs $n --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/lsys_${n}_train.json 2>/dev/null
  REM2D_FUSE_VELPOST=1 timeout 400 python3 bench.py --envs $n --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/lsys_${n}_steps.json 2>/dev/null
done
for f in $O/*.json; do python3 -c "
import json,sys
try:
    d=json.load(open('$f')); c=d['config']; print('$f'.split('/')[-1], '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], c.get('launch'), 'err', c.get('solver_errors'))
except Exception as e: print('$f', 'FAILED', e)"; done
