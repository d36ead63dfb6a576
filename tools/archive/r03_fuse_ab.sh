O=gpurun_out/r03_fuse_ab; mkdir -p $O
for i in 1 2 3; do
for f in 0 1; do
REM2D_FUSE_VELPOST=$f timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/b_${f}_$i.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/b_${f}_$i.json')); c=d['config']; print('fuse=$f', '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'first %.1f med %.1f' % (c['block_ms_first'], c['block_ms_median']))"
done; done
