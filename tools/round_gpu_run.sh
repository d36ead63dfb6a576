#!/bin/bash
# Everything measured for a round, on the GPU box: gpurun --timeout 1500 -- 'bash tools/round_gpu_run.sh'
# (each command under its own timeout so that a hang cannot eat the GPU budget)
set -u
O=gpurun_out/r02_final; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default_20.json 2>/dev/null
timeout 300 python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2>/dev/null
timeout 300 python bench.py --workload chain8 --no-cpu-baseline > $O/bench_chain8.json 2>/dev/null
timeout 300 python bench.py --workload chain4 --no-cpu-baseline > $O/bench_chain4.json 2>/dev/null
timeout 300 python bench.py --workload cppn_hardcore --no-cpu-baseline > $O/bench_cppn.json 2>/dev/null
timeout 300 python bench.py --workload generation --no-cpu-baseline > $O/bench_generation.json 2>/dev/null
timeout 300 python bench.py --workload single --steps 1000 --warmup 0 > $O/bench_single.json 2>/dev/null
timeout 300 python bench.py --discrete --no-cpu-baseline > $O/bench_discrete.json 2>/dev/null
timeout 300 python bench.py --pipeline 0 --no-cpu-baseline > $O/bench_fused.json 2>/dev/null
timeout 300 python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_2ranks_1gpu.json 2>/dev/null
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'err', d['config']['solver_errors'])"; done
bash tools/profile_round.sh r02_b > $O/profile.log 2>&1; tail -3 $O/profile.log
REM2D_TILE_SHAPE=0 timeout 420 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/sq0 -- python3 bench.py --steps 10 --warmup 2 --settle 80 --no-cpu-baseline > /dev/null 2> $O/sq0.err
python3 tools/collect_profiles.py sq $O/sq0 $O/r02_b_sq_counters_tile_shape0.json "REM2D_TILE_SHAPE=0 python3 bench.py --steps 10 --warmup 2 --settle 80 --no-cpu-baseline"; rm -rf $O/sq0
timeout 600 python tools/soak_parity.py --n 6000 --steps 300 > $O/soak_parity.txt 2>&1; tail -5 $O/soak_parity.txt
