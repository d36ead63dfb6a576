#!/bin/bash
# `bench.py --gpus 8` on the one GPU of a gpurun box: eight ranks share GPU 0, the fitness all-gather goes over gloo -- a
# smoke test of the launcher / shard / collective path and of the ranks' start-up cost, NOT an 8-GPU number.
# gpurun --timeout 1500 -- 'bash tools/r04_ranks8.sh'
set -u
O=gpurun_out/r04_ranks8; mkdir -p $O
rm -f /tmp/rem2d_bench_genomes_*   # cold: the ranks build their genomes themselves
for mode in weak strong; do
  SECONDS=0
  timeout 1200 python3 bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --scaling $mode --min-time 2 > $O/bench_8ranks_1gpu_$mode.json 2> $O/err_$mode.txt
  python3 -c "
import json; d=json.load(open('$O/bench_8ranks_1gpu_$mode.json')); c=d['config']
print('$mode', 'n_gpus', d['n_gpus'], 'creatures_total', d['creatures_total'], '%.2fM'%(d['value']/1e6), '%.3f ms/step'%d['ms_per_step'], 'startup', c['startup'], 'shared', c['ranks_share_one_gpu'], 'err', c['solver_errors'])"
  echo "$mode: wall $SECONDS s"; tail -3 $O/err_$mode.txt
done
