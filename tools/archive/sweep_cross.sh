#!/bin/bash
# Scratch sweep (GPU box): where the 256-lane tiles overtake the 64-lane tiles, by creatures per GPU
O=gpurun_out/ub; mkdir -p $O
for n in 81920 98304 114688 196608 262144; do
  python bench.py --envs $n --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  for sh in 3 0; do
    REM2D_TILE_SHAPE=$sh timeout 300 python bench.py --envs $n --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n creatures, shape $sh: %.2f M  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"
  done
done
