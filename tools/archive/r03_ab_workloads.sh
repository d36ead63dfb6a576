#!/bin/bash
# A/B of an environment switch on the secondary workloads: bash tools/r03_ab_workloads.sh <tag> "VAR=a" "VAR=b"
set -u
TAG=$1; shift
O=gpurun_out/r03_abw_$TAG; mkdir -p $O
python3 bench.py --workload chain4 --steps 20 --warmup 5 --no-cpu-baseline --min-time 0.2 > /dev/null 2>&1
for wl in chain4 single chain8 cppn_hardcore; do
for kv in "$@"; do
    extra=""; [ $wl = single ] && extra="--steps 1000 --warmup 0 --min-time 0"
    env $kv timeout 600 python3 bench.py --workload $wl --no-cpu-baseline $extra > $O/bench_${wl}_${kv//[^A-Za-z0-9=_]/_}.json 2> $O/err.txt
    python3 -c "
import json; d=json.load(open('$O/bench_${wl}_${kv//[^A-Za-z0-9=_]/_}.json')); c=d['config']
print('$wl $kv', '%.3fM'%(d['value']/1e6), '%.4f ms/step'%d['ms_per_step'], 'groups', c['step_groups'], 'err', c['solver_errors'])"
done
done
