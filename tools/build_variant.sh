#!/bin/bash
# Build a variant of librem2d.so for an A/B on the GPU box:  tools/build_variant.sh <name> [-DFLAG ...]  -> build/ab/librem2d_<name>.so
# (build/ is git-ignored but travels with gpurun; select it with REM2D_LIB_PATH=build/ab/librem2d_<name>.so)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/ab
hipcc -Os --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -shared -mllvm -instcombine-max-copied-from-constant-users=4000 -DREM2D_BUILD_ID="\"variant:$NAME\"" "$@" -I$ROOT/include \
  $ROOT/gym_rem2d_amd/csrc/rem2d.hip -o $ROOT/build/ab/librem2d_$NAME.so
echo built build/ab/librem2d_$NAME.so
