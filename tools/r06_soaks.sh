#!/bin/bash
# Round 6: the bulk parity evidence re-made with the final code -- the explicit hand-over, the 128-lane train (forced with
# REM2D_TILE_SHAPE=1 / 4 and through the generation policy), other seeds than round 5's (GPU box; the oracle on the host cores takes
# most of the time).  Then config 5 at its full size on the one GPU once more.
set -u
O=gpurun_out/r06_soaks; mkdir -p $O
timeout 1500 python tools/soak_parity.py --n 40000 --steps 800 --rebalance 37 > $O/soak_parity_big_rebalance.txt 2>&1; tail -2 $O/soak_parity_big_rebalance.txt
REM2D_TILE_SHAPE=1 timeout 1200 python tools/soak_parity.py --n 20000 --steps 500 --rebalance 41 > $O/soak_parity_train128_flex.txt 2>&1; tail -2 $O/soak_parity_train128_flex.txt
REM2D_TILE_SHAPE=4 timeout 900 python tools/soak_parity.py --n 12000 --steps 400 --rebalance 0 > $O/soak_parity_train128_static.txt 2>&1; tail -2 $O/soak_parity_train128_static.txt
REM2D_TILE_SHAPE=4 timeout 900 python tools/soak_parity.py --n 12000 --steps 400 --rebalance 37 > $O/soak_parity_static_ordered_per_step.txt 2>&1; tail -2 $O/soak_parity_static_ordered_per_step.txt
timeout 900 python tools/soak_parity.py --n 12000 --steps 400 --encodings --wide > $O/soak_parity_wide_encodings.txt 2>&1; tail -2 $O/soak_parity_wide_encodings.txt
for spec in "lsystem 131072 1000 61" "direct 65536 800 62" "network_arrays 32768 600 63"; do set -- $spec
  timeout 1500 python tools/soak_generation.py --encoding $1 --n $2 --cap $3 --seed $4 >> $O/soak_generation.txt 2>&1; echo "rc=$?" >> $O/soak_generation.txt
done; grep "individuals\|SOAK\|rc=" $O/soak_generation.txt
timeout 900 python tools/fuzz_launch_shapes.py --rounds 200 --seed 7 --creatures 2000 > $O/fuzz_launch_shapes.txt 2>&1; tail -1 $O/fuzz_launch_shapes.txt
timeout 900 python tools/fuzz_episode.py --rounds 100 --seed 7 --max-creatures 4000 > $O/fuzz_episode.txt 2>&1; tail -1 $O/fuzz_episode.txt
timeout 600 python3 bench.py --workload generation --envs 1048576 --no-cpu-baseline > $O/bench_generation_1M_1gpu.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/bench_generation_1M_1gpu.json')); print('generation 1M', '%.2fM' % (d['value']/1e6), d['config'].get('launch'), 'err', d['config'].get('solver_errors'))"
