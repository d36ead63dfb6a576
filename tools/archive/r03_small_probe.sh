#!/bin/bash
# small population, one step group: the vel4 launch time IS its slowest tile
set -u
for n in 4096; do
  timeout 600 python3 bench.py --envs $n --step-groups 1 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --min-time 0.3 "$@" > /tmp/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('/tmp/b.json')); c=d['config']; r=d['roofline']
print('envs $n', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'vel4 launch %.4f ms, step sequence %.4f' % (r['avg_launch_ms'], r['avg_step_sequence_ms'] or 0))"
done
