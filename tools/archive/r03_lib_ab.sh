#!/bin/bash
# A/B of two builds on the driver's own command (20-step blocks):  bash tools/r03_lib_ab.sh <tag> <variant.so> [pairs]
set -u
TAG=$1; V=$2; N=${3:-3}
O=gpurun_out/r03_libab_$TAG; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0.3 > /dev/null 2>&1
for i in $(seq 1 $N); do
for v in base variant; do
if [ $v = variant ]; then export REM2D_LIB_PATH=$V; else unset REM2D_LIB_PATH; fi
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/b_${v}_$i.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/b_${v}_$i.json')); c=d['config']; r=d['roofline']; print('$v', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'kernel %.4f seq %.4f' % (r['avg_launch_ms'], r['avg_step_sequence_ms'] or 0))"
done; done
