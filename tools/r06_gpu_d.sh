#!/bin/bash
# Round 6, GPU call D (the round's record run): the driver's command cold, the GPU suite, the default line, soaks and fuzzers with the
# final launch forms, the discrete-mode line (what the TOI part of the train costs), the shared-GPU rank smoke lines, the profile passes.
set -u
O=gpurun_out/r06_d; mkdir -p $O
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_20_cold.json 2> $O/bench_cold.err
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 600 python3 bench.py > $O/bench_default.json 2>/dev/null
timeout 300 python3 bench.py --discrete --no-cpu-baseline --no-secondary > $O/bench_discrete.json 2>/dev/null
timeout 400 python3 bench.py --workload generation --no-cpu-baseline > $O/bench_generation.json 2>/dev/null
timeout 300 python3 bench.py --workload chain8 --no-cpu-baseline > $O/bench_chain8.json 2>/dev/null
timeout 300 python3 bench.py --workload chain4 --no-cpu-baseline > $O/bench_chain4.json 2>/dev/null
timeout 300 python3 bench.py --workload cppn_hardcore --no-cpu-baseline > $O/bench_cppn.json 2>/dev/null
timeout 300 python3 bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline --min-time 2 > $O/bench_2ranks_1gpu_weak.json 2>/dev/null
timeout 600 python3 bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --min-time 2 > $O/bench_8ranks_1gpu_weak.json 2>/dev/null
timeout 600 python3 bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --min-time 2 --scaling strong > $O/bench_8ranks_1gpu_strong.json 2>/dev/null
timeout 900 python3 tools/soak_train_vs_steps.py --tile-shape 1 --launches 800 --out gpurun_out/r06_d/soak_train128_vs_steps.json 2>&1 | tail -2
timeout 900 python3 tools/soak_train_vs_steps.py --launches 2400 --out gpurun_out/r06_d/soak_train_vs_steps.json 2>&1 | tail -2
timeout 600 python tools/soak_parity.py --n 6000 --steps 300 > $O/soak_parity.txt 2>&1; tail -3 $O/soak_parity.txt
timeout 900 python tools/fuzz_launch_shapes.py --rounds 120 --seed 6 > $O/fuzz_launch_shapes.txt 2>&1; tail -1 $O/fuzz_launch_shapes.txt
timeout 900 python tools/fuzz_episode.py --rounds 40 --seed 6 > $O/fuzz_episode.txt 2>&1; tail -1 $O/fuzz_episode.txt
for f in $O/bench_*.json; do python3 -c "
import json,sys
try:
    d=json.load(open('$f')); c=d['config']; print('$f'.split('/')[-1], '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], c.get('launch'), 'groups', c.get('step_groups'), 'err', c.get('solver_errors'), {k: round(v['value']/1e6,1) for k,v in (d.get('secondary') or {}).items()})
except Exception as e: print('$f', 'FAILED', e)"; done
bash tools/train_profile.sh r06 > $O/profile.log 2>&1; tail -3 $O/profile.log
