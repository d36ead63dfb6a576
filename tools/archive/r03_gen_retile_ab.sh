for i in 1 2; do for kv in REM2D_RETILE=1 REM2D_RETILE=0; do
env $kv timeout 600 python3 bench.py --workload generation --no-cpu-baseline > /tmp/x.json 2>/dev/null; python3 -c "
import json; d=json.load(open('/tmp/x.json')); print('$kv generation 131072 %.2fM  %.2f s' % (d['value']/1e6, d['config']['timed_region_s']))"
done; done
