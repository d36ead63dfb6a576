#!/bin/bash
# SQ counters (VALU wave-instructions, active lanes) of the step kernels for library builds x environment overrides:
#   bash tools/sq_ab.sh <tag> <spec> [spec ...]     spec = base | <variant>[+label][:VAR=VALUE[,VAR=VALUE]]
# One rocprofv3 --pmc pass per spec (counters only: --kernel-trace, no other trace domain), reduced by collect_profiles.py sq.
set -u
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/sq_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS=${BENCH_ARGS:---steps 10 --warmup 2 --settle 80 --no-cpu-baseline --no-secondary --min-time 0}
python3 bench.py $ARGS > /dev/null 2>&1     # (builds and caches the genomes: no fork under the profiler)
for spec in "$@"; do
  v=${spec%%:*}; lib=${v%%+*}
  unset REM2D_LIB_PATH REM2D_TILE_SHAPE REM2D_RETILE REM2D_PRIO REM2D_FUSE_VELPOST
  if [ $lib != base ]; then export REM2D_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/librem2d_$lib.so; fi
  if [ "$spec" != "$v" ]; then for kv in $(echo ${spec#*:} | tr ',' ' '); do export $kv; done; fi
  timeout 420 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/raw_$v -- python3 bench.py $ARGS > /dev/null 2> $O/err_$v.txt
  python3 tools/collect_profiles.py sq $O/raw_$v $O/sq_$v.json "python3 bench.py $ARGS ($spec)" ${GROUPS_N:-4} > /dev/null 2>&1
  rm -rf $O/raw_$v
  python3 -c "
import json; d=json.load(open('$O/sq_$v.json'))
for k,x in sorted(d['kernels'].items()): print('$spec', k[:40], 'VALU %.1f M/launch' % (x['SQ_INSTS_VALU']/1e6), 'active lanes %.1f' % x.get('active_lanes_per_valu_inst', 0), 'launches', x.get('launches'))" 2>/dev/null || tail -3 $O/err_$v.txt
done
