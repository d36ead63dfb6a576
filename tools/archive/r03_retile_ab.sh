#!/bin/bash
set -u
O=gpurun_out/r03_retile_win; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "retiling or twin or config5" 2>&1 | tail -3
bash tools/r03_ab.sh retilewin REM2D_RETILE=0 REM2D_RETILE=1
for r in 0 1; do
  REM2D_RETILE=$r timeout 600 python3 bench.py --workload generation --no-cpu-baseline > $O/gen_retile$r.json 2> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/gen_retile$r.json')); print('generation retile=$r', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], d['steps'], 'steps')"
done
