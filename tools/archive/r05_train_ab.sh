python -c "import torch" >/dev/null 2>&1
export REM2D_FUSE_VELPOST=2 REM2D_STEP_GROUPS=1
for m in 2 1; do export REM2D_PROBE_QUEUE=$m; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config3 mode $m 1 group: %.2f M' % (d['value']/1e6), d['config']['solver_errors'])"; done
export REM2D_PROBE_QUEUE=2
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-time 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']; print('with secondary: %.2f M' % (d['value']/1e6), {k:(round(v['value']/1e6,2), v.get('tile_shape'), v.get('step_groups')) for k,v in s.items() if isinstance(v,dict) and 'value' in v})"
for wl in chain4 single; do for q in 0 2; do if [ $q = 0 ]; then unset REM2D_PROBE_QUEUE REM2D_FUSE_VELPOST REM2D_STEP_GROUPS; else export REM2D_PROBE_QUEUE=2 REM2D_FUSE_VELPOST=2 REM2D_STEP_GROUPS=1; fi; timeout 200 python bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline --min-time 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl queue=$q: %.4g' % d['value'], d['unit'])"; done; done
export REM2D_PROBE_QUEUE=2 REM2D_FUSE_VELPOST=2; unset REM2D_STEP_GROUPS
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config3 mode 2 default groups (%s): %.2f M' % (d['config'].get('step_groups'), d['value']/1e6))"
