#!/bin/bash
# Round 5, after the step train became the default: the other workloads of DESIGN.md 6 / 7 and a bulk parity pass with the final code
set -u
O=gpurun_out/r05_train_final; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
one() { python -c "
import json,sys; d=json.loads(open('$1').read()); print('$2: %.4g %s  %.4f ms/step  groups %s  err %s' % (d['value'], d['unit'], d['ms_per_step'], d['config'].get('step_groups'), d['config'].get('solver_errors')))"; }
timeout 300 python bench.py --workload chain4 --no-cpu-baseline > $O/bench_chain4.json 2>/dev/null; one $O/bench_chain4.json chain4
timeout 300 python bench.py --workload single --no-cpu-baseline > $O/bench_single.json 2>/dev/null; one $O/bench_single.json single
timeout 200 python tools/bench_facade.py > $O/facade.txt 2>&1; tail -2 $O/facade.txt
for n in 8192 16384 32768; do timeout 300 python bench.py --envs $n --no-cpu-baseline --no-secondary --min-time 2 > $O/bench_envs_$n.json 2>/dev/null; one $O/bench_envs_$n.json "lsystem $n creatures"; done
timeout 400 python bench.py --workload generation --no-cpu-baseline > $O/bench_generation.json 2>/dev/null; one $O/bench_generation.json generation
timeout 600 python tools/soak_parity.py --n 20000 --steps 400 --rebalance 37 > $O/soak_parity.txt 2>&1; tail -2 $O/soak_parity.txt
timeout 600 python tools/soak_parity.py --n 8000 --steps 300 --encodings > $O/soak_parity_encodings.txt 2>&1; tail -2 $O/soak_parity_encodings.txt
timeout 900 python tools/soak_generation.py --encoding lsystem --n 65536 --cap 1000 --seed 41 >> $O/soak_generation.txt 2>&1; tail -2 $O/soak_generation.txt
timeout 600 python tools/soak_generation.py --encoding direct --n 32768 --cap 800 --seed 42 >> $O/soak_generation.txt 2>&1; tail -2 $O/soak_generation.txt
timeout 600 python tools/fuzz_launch_shapes.py --rounds 120 --seed 7 --creatures 2000 > $O/fuzz_launch_shapes.txt 2>&1; tail -1 $O/fuzz_launch_shapes.txt
timeout 600 python tools/fuzz_episode.py --rounds 60 --seed 7 --max-creatures 4000 > $O/fuzz_episode.txt 2>&1; tail -1 $O/fuzz_episode.txt
