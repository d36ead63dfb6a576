#!/bin/bash
# bash tools/r05_ablate.sh <lib-variant|base> <debug bits> [debug bits ...]  -> gpurun_out/r05_ablate/sq_<lib>_<bits>.json
set -u
LIBV=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/r05_ablate; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
unset REM2D_LIB_PATH
if [ $LIBV != base ]; then export REM2D_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/librem2d_$LIBV.so; fi
python3 tools/r05_ablate.py --settle 5 --steps 1 > /dev/null 2>&1   # genome cache
for dbg in "$@"; do
  timeout 420 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/raw -- python3 tools/r05_ablate.py --debug $dbg > /dev/null 2> $O/err_${LIBV}_$dbg.txt
  python3 tools/collect_profiles.py sq $O/raw $O/sq_${LIBV}_$dbg.json "tools/r05_ablate.py --debug $dbg ($LIBV)" 4 --tail=32 > /dev/null 2>&1
  rm -rf $O/raw
  python3 -c "
import json; d=json.load(open('$O/sq_${LIBV}_$dbg.json'))
for k,x in sorted(d['kernels'].items()): print('dbg $dbg', k[:40], 'VALU %.1f M/launch' % (x['SQ_INSTS_VALU']/1e6), 'active lanes %.1f' % x.get('active_lanes_per_valu_inst', 0), 'launches', x.get('launches'))" 2>/dev/null || tail -3 $O/err_${LIBV}_$dbg.txt
done
