#!/bin/bash
# Scratch sweep (GPU box): env-steps/s of config 3 by tile shape x number of step groups
O=gpurun_out/ub; mkdir -p $O
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1   # population cache
for sh in 3 1 0; do for g in 2 3 4 5 6 8; do
  REM2D_TILE_SHAPE=$sh timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --step-groups $g 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shape $sh groups $g: %.2f M  %.3f ms/step  vel4 %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
