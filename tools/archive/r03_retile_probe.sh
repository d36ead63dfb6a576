#!/bin/bash
# post-kernel re-tiling (REM2D_RETILE): SQ counters of the post kernel and the generation workload, with and without
set -u
O=gpurun_out/r03_retile; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0.3 > /dev/null 2>&1
for r in 0 1; do
  REM2D_RETILE=$r timeout 600 python3 bench.py --workload generation --no-cpu-baseline > $O/gen_retile$r.json 2> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/gen_retile$r.json')); print('generation retile=$r', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], d['steps'], 'steps')"
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for r in 0 1; do
  REM2D_RETILE=$r timeout 420 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/sq$r -- python3 bench.py --steps 10 --warmup 2 --settle 80 --no-cpu-baseline --no-secondary --min-time 0 > /dev/null 2> $O/sq$r.err
  python3 tools/collect_profiles.py sq $O/sq$r $O/sq_counters_retile$r.json "REM2D_RETILE=$r python3 bench.py --steps 10 --warmup 2 --settle 80 --no-cpu-baseline --no-secondary --min-time 0"
  rm -rf $O/sq$r
  python3 -c "
import json; d=json.load(open('$O/sq_counters_retile$r.json'))
for k,v in d['kernels'].items():
    if k.startswith('rem2d_'): print('retile=$r', k[:40], 'VALU insts %.1fM' % (v['SQ_INSTS_VALU']/1e6), 'lanes %.1f' % v.get('active_lanes_per_valu_inst',0), 'busy', v.get('SQ_BUSY_CYCLES'))"
done
