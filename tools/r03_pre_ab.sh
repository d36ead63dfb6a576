for i in 1 2 3; do for v in base pre3; do
if [ $v = pre3 ]; then export REM2D_LIB_PATH=build/variants/librem2d_pre3.so; else unset REM2D_LIB_PATH; fi
for wl in chain8 cppn_hardcore; do
timeout 300 python3 bench.py --workload $wl --no-cpu-baseline > /tmp/x.json 2>/dev/null; python3 -c "
import json; d=json.load(open('/tmp/x.json')); print('$v $wl %.1fM' % (d['value']/1e6))"; done; done; done
