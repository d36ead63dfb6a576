#!/bin/bash
# Everything measured for a round, on the GPU box: gpurun --timeout 3000 -- 'bash tools/round_gpu_run.sh r05_final'
# (each command under its own timeout so that a hang cannot eat the GPU budget).  The driver's command runs FIRST, as the
# first GPU process of the lease (cold), then again (warm).
set -u
TAG=${1:-r05_final}
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_20_cold.json 2> $O/bench_cold.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_20.json 2>/dev/null
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 600 python3 bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench_default.json 2>/dev/null
timeout 300 python3 bench.py --workload chain8 --no-cpu-baseline > $O/bench_chain8.json 2>/dev/null
timeout 300 python3 bench.py --workload chain4 --no-cpu-baseline > $O/bench_chain4.json 2>/dev/null
timeout 300 python3 bench.py --workload cppn_hardcore --no-cpu-baseline > $O/bench_cppn.json 2>/dev/null
timeout 300 python3 bench.py --workload generation --no-cpu-baseline > $O/bench_generation.json 2>/dev/null
timeout 300 python3 bench.py --workload single --steps 1000 --warmup 0 --min-time 0 > $O/bench_single.json 2>/dev/null
timeout 300 python3 bench.py --discrete --no-cpu-baseline --no-secondary > $O/bench_discrete.json 2>/dev/null
timeout 300 python3 bench.py --pipeline 0 --no-cpu-baseline --no-secondary > $O/bench_fused.json 2>/dev/null
timeout 600 python3 bench.py --workload generation --envs 1048576 --no-cpu-baseline > $O/bench_generation_1M_1gpu.json 2>/dev/null
timeout 300 python3 bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline --min-time 2 > $O/bench_2ranks_1gpu_weak.json 2>/dev/null
timeout 300 python3 bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline --min-time 2 --scaling strong > $O/bench_2ranks_1gpu_strong.json 2>/dev/null
timeout 600 python3 bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --min-time 2 > $O/bench_8ranks_1gpu_weak.json 2>/dev/null
timeout 600 python3 bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --min-time 2 --scaling strong > $O/bench_8ranks_1gpu_strong.json 2>/dev/null
timeout 120 python3 tools/bench_facade.py 2000 > $O/facade.txt 2>/dev/null
for f in $O/bench_*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); c=d['config']; print('$f'.split('/')[-1], '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'blocks', c['blocks'], 'first %.1f med %.1f ms' % (c['block_ms_first'], c['block_ms_median']), 'err', c['solver_errors'], {k: round(v['value']/1e6,1) for k,v in (d.get('secondary') or {}).items()})"; done
cat $O/facade.txt
bash tools/profile_round.sh ${TAG%_final} > $O/profile.log 2>&1; tail -3 $O/profile.log
timeout 600 python tools/soak_parity.py --n 6000 --steps 300 > $O/soak_parity.txt 2>&1; tail -4 $O/soak_parity.txt
timeout 600 python tools/fuzz_launch_shapes.py --rounds 40 --seed 1 > $O/fuzz_launch_shapes.txt 2>&1; tail -1 $O/fuzz_launch_shapes.txt
timeout 600 python tools/fuzz_episode.py --rounds 30 --seed 1 > $O/fuzz_episode.txt 2>&1; tail -1 $O/fuzz_episode.txt
timeout 900 python3 tools/fma_tolerance.py --n 1024 --steps 150 > $O/fma_tolerance.txt 2>&1; tail -6 $O/fma_tolerance.txt
bash tools/archive/r05_bench_ea.sh 1048576 > $O/bench_ea.txt 2>&1; cp gpurun_out/r05_bench_ea/bench_ea_*.json $O/ 2>/dev/null; cat $O/bench_ea.txt
