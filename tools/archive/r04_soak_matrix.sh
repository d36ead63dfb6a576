#!/bin/bash
# the parity soak over flags, builds and formulations the unit tests cover with a few dozen creatures each
mkdir -p gpurun_out
O=gpurun_out/soak_matrix.txt; : > $O
run() { echo "# ${ENVV[*]} tools/soak_parity.py --n 6000 --steps 300 $*" >> $O; env "${ENVV[@]}" python tools/soak_parity.py --n 6000 --steps 300 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O; }
ENVV=(X=1); run --flags 0
ENVV=(X=1); run --flags 3
ENVV=(X=1); run --flags 5
ENVV=(X=1); run --flags 1 --wide
ENVV=(REM2D_PIPELINE=0); run --flags 1
ENVV=(REM2D_FUSE_VELPOST=0); run --flags 1
ENVV=(REM2D_PRIO=0 REM2D_HEAVY_PER_WAVE=2); run --flags 1 --rebalance 11
cat $O
