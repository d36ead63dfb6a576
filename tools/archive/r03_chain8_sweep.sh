run() { tag=$1; shift; env "$@" timeout 300 python3 bench.py --workload chain8 --no-cpu-baseline $EXTRA > /tmp/c8_$tag.json 2>/dev/null; python3 -c "
import json; d=json.load(open('/tmp/c8_$tag.json')); c=d['config']; r=d['roofline']
print('$tag', '%.1fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'groups', c['step_groups'], 'kernel %.4f seq %.4f' % (r['avg_launch_ms'], r['avg_step_sequence_ms'] or 0))"; }
EXTRA=""; run default REM2D_NOOP=1
run shape3 REM2D_TILE_SHAPE=3
run shape0 REM2D_TILE_SHAPE=0
EXTRA="--step-groups 2"; run g2 REM2D_NOOP=1
EXTRA="--step-groups 4"; run g4 REM2D_NOOP=1
EXTRA="--step-groups 4"; run shape3g4 REM2D_TILE_SHAPE=3
EXTRA="--step-groups 6"; run g6 REM2D_NOOP=1
