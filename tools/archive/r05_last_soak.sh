set -u
O=gpurun_out/r05_last; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
timeout 600 python tools/soak_parity.py --n 20000 --steps 300 --rebalance 50 > $O/soak_parity.txt 2>&1; tail -2 $O/soak_parity.txt
timeout 600 python tools/soak_generation.py --encoding lsystem --n 32768 --cap 1000 --seed 51 > $O/soak_generation.txt 2>&1; tail -2 $O/soak_generation.txt
timeout 600 python tools/soak_generation.py --encoding network_arrays --n 16384 --cap 600 --seed 52 >> $O/soak_generation.txt 2>&1; tail -2 $O/soak_generation.txt
timeout 600 python tools/fuzz_launch_shapes.py --rounds 80 --seed 8 --creatures 2000 > $O/fuzz_launch_shapes.txt 2>&1; tail -1 $O/fuzz_launch_shapes.txt
timeout 600 python tools/fuzz_episode.py --rounds 40 --seed 8 --max-creatures 4000 > $O/fuzz_episode.txt 2>&1; tail -1 $O/fuzz_episode.txt
