#!/bin/bash
# env-steps/s of whole L-system generations by creatures per GPU x tile shape of the velocity kernel
# (3 = 64 bodies / 4 waves per SIMD, 1 = 128 / 4 flexible, 2 = 192 / 3 flexible, 0 = 256 / 2):  bash tools/r04_sweep_pop_shape.sh "<envs ...>" "<shapes ...>"
ENVS=${1:-"196608 393216"}; SHAPES=${2:-"3 1 2 0"}
for n in $ENVS; do
for sh in $SHAPES; do
REM2D_TILE_SHAPE=$sh timeout 900 python3 bench.py --workload generation --envs $n --no-cpu-baseline > /tmp/g_${n}_$sh.json 2>/dev/null
python3 -c "
import json; d=json.load(open('/tmp/g_${n}_$sh.json')); print('envs $n shape $sh  %.2fM  %.3f ms/step err %d' % (d['value']/1e6, d['ms_per_step'], d['config']['solver_errors']))"
done; done
