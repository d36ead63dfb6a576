#!/bin/bash
# Round 5, second bulk parity pass with the final code: other seeds, the other world flags, the wide build (GPU box; the oracle runs
# on the host cores and takes most of the time)
set -u
O=gpurun_out/r05_soaks2; mkdir -p $O
timeout 900 python tools/soak_parity.py --n 20000 --steps 500 --rebalance 23 --flags 3 > $O/soak_parity_flags3.txt 2>&1; tail -2 $O/soak_parity_flags3.txt
timeout 900 python tools/soak_parity.py --n 20000 --steps 500 --rebalance 50 --flags 5 > $O/soak_parity_flags5.txt 2>&1; tail -2 $O/soak_parity_flags5.txt
timeout 900 python tools/soak_parity.py --n 12000 --steps 400 --encodings --wide > $O/soak_parity_wide_encodings.txt 2>&1; tail -2 $O/soak_parity_wide_encodings.txt
for spec in "lsystem 65536 1200 21" "direct 65536 1000 22" "network_arrays 32768 800 23" "network 8192 800 24"; do set -- $spec
  timeout 1500 python tools/soak_generation.py --encoding $1 --n $2 --cap $3 --seed $4 >> $O/soak_generation.txt 2>&1; echo "rc=$?" >> $O/soak_generation.txt
done; grep "individuals\|SOAK\|rc=" $O/soak_generation.txt
timeout 900 python tools/fuzz_launch_shapes.py --rounds 200 --seed 6 --creatures 2000 > $O/fuzz_launch_shapes.txt 2>&1; tail -1 $O/fuzz_launch_shapes.txt
timeout 900 python tools/fuzz_episode.py --rounds 100 --seed 6 --max-creatures 4000 > $O/fuzz_episode.txt 2>&1; tail -1 $O/fuzz_episode.txt
