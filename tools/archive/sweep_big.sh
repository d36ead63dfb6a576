#!/bin/bash
# Scratch sweep (GPU box): env-steps/s at 131 072 creatures per GPU (config 5's share) by tile shape x step groups
O=gpurun_out/ub; mkdir -p $O
python bench.py --envs 131072 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1   # population cache
for sh in 3 1 0; do for g in 3 4 6; do
  REM2D_TILE_SHAPE=$sh timeout 200 python bench.py --envs 131072 --steps 60 --warmup 10 --no-cpu-baseline --step-groups $g 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('131072 creatures, shape $sh groups $g: %.2f M  %.3f ms/step  vel4 %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
