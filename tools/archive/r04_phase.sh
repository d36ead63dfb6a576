#!/bin/bash
# Kernel trace of the driver's command: how the step groups' long kernels overlap (tools/trace_overlap.py).
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r04_phase; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0 > /dev/null 2>&1
for g in ${GROUPS_LIST:-4}; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$g -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --min-time 0 --step-groups $g > $O/bench_$g.json 2> $O/err_$g.txt
python3 tools/trace_overlap.py $O/tr_$g $((180 * g)) > $O/overlap_groups$g.txt 2>&1; cat $O/overlap_groups$g.txt | head -8
rm -rf $O/tr_$g
done
