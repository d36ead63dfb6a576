#!/bin/bash
# A/B of library builds on the driver's own command:  bash tools/lib_ab.sh <tag> <pairs> <name1> [name2 ...]
# names: "base" = gym_rem2d_amd/librem2d.so, anything else = build/ab/librem2d_<name>.so (tools/build_variant.sh).
# Extra bench arguments through BENCH_ARGS, extra environment per variant as name:VAR=VALUE.
set -u
TAG=$1; N=$2; shift 2
O=gpurun_out/ab_$TAG; mkdir -p $O
ARGS=${BENCH_ARGS:---steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 1}
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0.3 > /dev/null 2>&1
for i in $(seq 1 $N); do
for spec in "$@"; do
  v=${spec%%:*}; envs=""
  if [ "$spec" != "$v" ]; then envs=$(echo ${spec#*:} | tr ',' ' '); fi
  lib=${v%%+*}
  if [ $lib = base ]; then unset REM2D_LIB_PATH; else export REM2D_LIB_PATH=$PWD/build/ab/librem2d_$lib.so; fi
  env $envs timeout 600 python3 bench.py $ARGS > $O/b_${v}_$i.json 2>$O/err_${v}_$i.txt
  python3 -c "
import json; d=json.load(open('$O/b_${v}_$i.json')); c=d['config']; r=d['roofline']; print('$spec', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'kernel %.4f seq %.4f' % (r['avg_launch_ms'] or 0, r['avg_step_sequence_ms'] or 0), 'err', c['solver_errors'])" || tail -3 $O/err_${v}_$i.txt
done; done
