#!/bin/bash
# full-state parity soak with the other tile shapes forced (REM2D_TILE_SHAPE is the Python host's experiment override)
# usage: r04_soak_shapes.sh [shapes...]   (default: 1 at full size, then 2 0 4)
mkdir -p gpurun_out
O=gpurun_out/soak_shapes.txt; : > $O
SHAPES=${@:-"1 2 0 4"}
for s in $SHAPES; do
  if [ "$s" = 1 ]; then N=40000; T=800; else N=12000; T=400; fi
  echo "# REM2D_TILE_SHAPE=$s tools/soak_parity.py --n $N --steps $T --rebalance 37" >> $O
  REM2D_TILE_SHAPE=$s python tools/soak_parity.py --n $N --steps $T --rebalance 37 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
