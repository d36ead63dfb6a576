#!/bin/bash
# The step train: the profiles behind bench.py's roofline block for rem2d_step_train_kernel (GPU box):  bash tools/train_profile.sh <tag>
# (tag = the round, e.g. r06: the files land as gpurun_out/<tag>_train_prof/<tag>_step_train_{kernel_stats.csv,counters.json}).  Separate passes:
# --kernel-trace --stats, then FETCH_SIZE, WRITE_SIZE, the SQ counters (gpurun refuses --pmc with other trace domains).
# The profiled command steps the headline population in ABI calls of 50 steps, aligned with the re-ordering cadence (settle +
# warmup = 100), so that every launch of the timed region is one 50-step train of the whole population.
set -u
TAG=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_train_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="--steps 50 --warmup 40 --settle 60 --steps-per-launch 50 --no-cpu-baseline --no-secondary --min-time 0"
python3 bench.py $ARGS > $O/bench_unprofiled.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 200 --warmup 50 --settle 50 --steps-per-launch 50 --no-cpu-baseline --no-secondary --min-time 0 > $O/bench_under_stats.json 2> $O/stats.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_step_train_kernel_stats.csv
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 420 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/raw$i -- python3 bench.py $ARGS > /dev/null 2> $O/err$i.txt
  f=$(find $O/raw$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $O/counters$i.csv
  rm -rf $O/raw$i
done
python3 tools/train_reduce.py $O > $O/${TAG}_step_train_counters.json; cat $O/${TAG}_step_train_counters.json | head -c 1500
rm -rf $O/stats $O/counters*.csv
