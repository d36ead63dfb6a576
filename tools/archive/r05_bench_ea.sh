#!/bin/bash
# Round 5: one EA generation end to end at config 5's size on ONE GPU for the array populations of the three encodings
# (select + mutate + native expression on the host's cores, upload, whole episodes): bash tools/r05_bench_ea.sh [population]
set -u
N=${1:-1048576}
O=gpurun_out/r05_bench_ea; mkdir -p $O
for enc in direct network lsystem; do
  timeout 900 python3 tools/bench_ea.py --arrays --encoding $enc --population $N --generations 2 > $O/bench_ea_${enc}_$N.json 2> $O/err_${enc}_$N.txt
  python3 -c "
import json; d=json.load(open('$O/bench_ea_${enc}_$N.json'))
print(d['encoding'], d['population'], 'init %.1f s' % d['init_s'], 'cores', d['host_cores'], [(round(g['of_which_select_s'],2), round(g['select_clone_mutate_s'],2), round(g['encode_s'],2), round(g['host_s'],2), round(g['upload_and_episode_s'],1), g['steps']) for g in d['generations']])" || tail -3 $O/err_${enc}_$N.txt
done
