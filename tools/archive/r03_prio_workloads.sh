#!/bin/bash
set -u
O=gpurun_out/r03_prio2; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0.3 > /dev/null 2>&1
run() { tag=$1; shift; env "$@" timeout 600 python3 bench.py --no-cpu-baseline --no-secondary $EXTRA > $O/b_$tag.json 2>> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/b_$tag.json')); c=d['config']; r=d['roofline']
print('$tag', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'kernel %.4f seq %.4f' % (r['avg_launch_ms'] or 0, r['avg_step_sequence_ms'] or 0), 'err', c['solver_errors'])"; }
EXTRA="--steps 100 --warmup 10"
run off REM2D_PRIO=0
run veltoi REM2D_PRIO=5
run veltoi_fuse REM2D_PRIO=5 REM2D_FUSE_VELPOST=1
run veltoi_fuse_b REM2D_PRIO=5 REM2D_FUSE_VELPOST=1
EXTRA="--workload chain8"
run c8_off REM2D_PRIO=0
run c8_veltoi REM2D_PRIO=5
EXTRA="--workload cppn_hardcore"
run cppn_off REM2D_PRIO=0
run cppn_veltoi REM2D_PRIO=5
EXTRA="--workload generation"
run gen_off REM2D_PRIO=0
run gen_veltoi REM2D_PRIO=5
EXTRA="--workload chain4"
run c4_off REM2D_PRIO=0
run c4_veltoi REM2D_PRIO=5
