#!/bin/bash
# Round 3, VERDICT item 1: the driver's exact command as the FIRST GPU command of a fresh lease, then again (warm),
# the hipGraph replay, the 200-step form, and a kernel-trace timeline of the same command.
#   gpurun --timeout 900 -- 'bash tools/r03_cold_warm.sh <tag>'
set -u
TAG=${1:-a}
O=gpurun_out/r03_cold_warm_$TAG; mkdir -p $O
S=$(date +%s.%N)
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_run1_cold.json 2> $O/bench_run1.err
echo "run1 (cold, the driver's command) took $(echo "$(date +%s.%N) - $S" | bc) s" > $O/wall.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_run2_warm.json 2> $O/bench_run2.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --graph 1 > $O/bench_run3_graph.json 2> $O/bench_run3.err
timeout 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench_run4_200.json 2> $O/bench_run4.err
timeout 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --graph 1 > $O/bench_run5_200_graph.json 2> $O/bench_run5.err
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 420 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0 > $O/bench_traced.json 2> $O/trace.err
python3 tools/trace_overlap.py $O/trace > $O/trace_overlap.txt 2>&1
rm -rf $O/trace
for f in $O/bench_*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); c=d['config']; print('$f'.split('/')[-1], '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'blocks', c.get('blocks'), 'first %.2f min %.2f med %.2f ms' % (c['block_ms_first'], c['block_ms_min'], c['block_ms_median']), 'kernel', round(d['roofline']['avg_launch_ms'],4), {k: round(v['value']/1e6,1) for k,v in (d.get('secondary') or {}).items()})"; done
cat $O/wall.txt; tail -12 $O/trace_overlap.txt; tail -3 $O/*.err | head -40
