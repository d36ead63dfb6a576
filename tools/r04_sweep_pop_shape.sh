for n in 196608 393216; do
for sh in 3 1 0; do
REM2D_TILE_SHAPE=$sh timeout 900 python3 bench.py --workload generation --envs $n --no-cpu-baseline > /tmp/g_$n_$sh.json 2>/dev/null
python3 -c "
import json; d=json.load(open('/tmp/g_$n_$sh.json')); print('envs $n shape $sh  %.2fM  %.3f ms/step err %d' % (d['value']/1e6, d['ms_per_step'], d['config']['solver_errors']))"
done; done
