#!/bin/bash
# Round 5: the bulk parity evidence re-made with the final code (GPU box; ~10 GPU-minutes + the oracle on the host cores)
set -u
O=gpurun_out/r05_soaks; mkdir -p $O
timeout 1500 python tools/soak_parity.py --n 40000 --steps 800 --rebalance 37 > $O/soak_parity_big_rebalance.txt 2>&1; tail -2 $O/soak_parity_big_rebalance.txt
timeout 900 python tools/soak_parity.py --n 12000 --steps 400 --encodings > $O/soak_parity_encodings.txt 2>&1; tail -2 $O/soak_parity_encodings.txt
for spec in "lsystem 131072 1000" "direct 65536 800" "network_arrays 32768 600"; do set -- $spec
  timeout 1500 python tools/soak_generation.py --encoding $1 --n $2 --cap $3 >> $O/soak_generation.txt 2>&1; echo "rc=$?" >> $O/soak_generation.txt
done; grep "individuals\|SOAK\|rc=" $O/soak_generation.txt
timeout 900 python tools/fuzz_launch_shapes.py --rounds 200 --seed 5 --creatures 2000 > $O/fuzz_launch_shapes.txt 2>&1; tail -1 $O/fuzz_launch_shapes.txt
timeout 900 python tools/fuzz_episode.py --rounds 100 --seed 5 --max-creatures 4000 > $O/fuzz_episode.txt 2>&1; tail -1 $O/fuzz_episode.txt
