#!/bin/bash
# sweep of step groups / steps per ABI call on the headline workload with the one-call round-robin enqueue
set -u
O=gpurun_out/r03_sweep; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0.3 > /dev/null 2>&1
run() { tag=$1; shift; timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary "$@" > $O/b_$tag.json 2>> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/b_$tag.json')); c=d['config']; r=d['roofline']
print('$tag', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'groups', c['step_groups'], 'kernel %.4f seq %.4f' % (r['avg_launch_ms'], r['avg_step_sequence_ms'] or 0))"; }
for g in 2 3 4 5 6 8; do run groups$g --step-groups $g; done
for s in 5 10 50 100; do run spl$s --steps-per-launch $s; done
GPU_MAX_HW_QUEUES=4 run hwq4
GPU_MAX_HW_QUEUES=16 run hwq16_g6 --step-groups 6
run graph_g4 --graph 1
run graph_g6 --graph 1 --step-groups 6
