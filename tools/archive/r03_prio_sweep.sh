#!/bin/bash
# s_setprio experiment: which kernels (REM2D_PRIO bits: 1 velocity tiles by cost, 2 position blocks, 4 TOI wavefronts; bits 8.. = the
# period from which a position block counts as long), which thresholds (cost = 7 ticks + 10 sub-slots per iteration)
set -u
O=gpurun_out/r03_prio; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0.3 > /dev/null 2>&1
run() { tag=$1; shift; env "$@" timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/b_$tag.json 2>> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/b_$tag.json')); c=d['config']; r=d['roofline']
print('$tag', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'kernel %.4f seq %.4f' % (r['avg_launch_ms'], r['avg_step_sequence_ms'] or 0), 'err', c['solver_errors'])"; }
run off REM2D_PRIO=0
run veltoi REM2D_PRIO=5
run veltoi_b REM2D_PRIO=5
run veltoi_50_65 REM2D_PRIO=5 REM2D_PRIO_T1=50 REM2D_PRIO_T2=65
run veltoi_70_85 REM2D_PRIO=5 REM2D_PRIO_T1=70 REM2D_PRIO_T2=85
run veltoi_40_75 REM2D_PRIO=5 REM2D_PRIO_T1=40 REM2D_PRIO_T2=75
run veltoi_60_999 REM2D_PRIO=5 REM2D_PRIO_T1=60 REM2D_PRIO_T2=999
run all_post5 REM2D_PRIO=$((7 + 5*256))
run all_post6 REM2D_PRIO=$((7 + 6*256))
run all_post4 REM2D_PRIO=$((7 + 4*256))
run off2 REM2D_PRIO=0
