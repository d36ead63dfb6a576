#!/bin/bash
# the 128-lane tile kernel at 3 / 4 wavefronts per SIMD (variant build via REM2D_LIB_PATH) on the workloads that use it
for i in 1 2; do for v in base variant; do
if [ $v = variant ]; then export REM2D_LIB_PATH=$1; else unset REM2D_LIB_PATH; fi
timeout 300 python3 bench.py --workload chain8 --no-cpu-baseline > /tmp/x.json 2>/dev/null; python3 -c "
import json; d=json.load(open('/tmp/x.json')); print('$v chain8 %.1fM' % (d['value']/1e6))"
timeout 600 python3 bench.py --workload generation --envs 262144 --no-cpu-baseline > /tmp/x.json 2>/dev/null; python3 -c "
import json; d=json.load(open('/tmp/x.json')); print('$v generation 262144 %.2fM  %.1f s' % (d['value']/1e6, d['config']['timed_region_s']))"
done; done
