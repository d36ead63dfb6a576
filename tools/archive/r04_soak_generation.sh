#!/bin/bash
# whole-episode parity of a generation, every individual against the oracle (GPU box)
mkdir -p gpurun_out
python tools/soak_generation.py --n 131072 --cap 1000 > gpurun_out/soak_generation.txt 2>&1
echo "rc=$?" >> gpurun_out/soak_generation.txt
python tools/soak_generation.py --n 40000 --cap 2500 --seed 12 --max-modules 20 >> gpurun_out/soak_generation.txt 2>&1
echo "rc=$?" >> gpurun_out/soak_generation.txt
tail -n 8 gpurun_out/soak_generation.txt
