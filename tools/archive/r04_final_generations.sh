O=gpurun_out/r04_final2; mkdir -p $O
timeout 300 python3 bench.py --workload generation --no-cpu-baseline > $O/bench_generation.json 2>/dev/null
timeout 600 python3 bench.py --workload generation --envs 1048576 --no-cpu-baseline > $O/bench_generation_1M_1gpu.json 2>/dev/null
timeout 600 python3 bench.py --workload generation --envs 393216 --no-cpu-baseline > $O/bench_generation_393216.json 2>/dev/null
for f in $O/bench_*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); c=d['config']; print('$f'.split('/')[-1], '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'err', c['solver_errors'], 'wall %.1f s' % (c['blocks_ms'][0]/1e3))"; done
