#!/bin/bash
# Round 6, GPU call C: the whole GPU suite on the train128 build, the driver's command, the issue micro-benchmark, and the A/B of the
# velocity-cost rebalance key (build/ab/librem2d_rbv{3,5}.so = -DREBALANCE_VEL=3 / 5).
set -u
O=gpurun_out/r06_c; mkdir -p $O
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_20_cold.json 2> $O/bench_cold.err
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 600 python3 bench.py > $O/bench_default.json 2>/dev/null
timeout 300 tools/ubench_latency.bin > $O/ubench_issue.txt 2>&1; tail -14 $O/ubench_issue.txt
REM2D_TILE_SHAPE=1 timeout 300 python3 bench.py --steps 100 --warmup 40 --settle 60 --no-cpu-baseline --no-secondary --min-time 2 > $O/lsys_65536_shape1_train128.json 2>/dev/null
BENCH_ARGS="--steps 100 --warmup 40 --settle 60 --no-cpu-baseline --no-secondary --min-time 2" bash tools/lib_ab.sh r06_rbv 2 base rbv5 rbv3
BENCH_ARGS="--workload cppn_hardcore --steps 100 --warmup 40 --settle 60 --no-cpu-baseline --no-secondary --min-time 2" bash tools/lib_ab.sh r06_rbv_cppn 1 base rbv5 rbv3
for f in $O/bench_default_20_cold.json $O/bench_default.json $O/lsys_65536_shape1_train128.json; do python3 -c "
import json,sys
try:
    d=json.load(open('$f')); c=d['config']; print('$f'.split('/')[-1], '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], c.get('launch'), 'err', c.get('solver_errors'), {k: round(v['value']/1e6,1) for k,v in (d.get('secondary') or {}).items()})
except Exception as e: print('$f', 'FAILED', e)"; done
