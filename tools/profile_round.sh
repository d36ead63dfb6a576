#!/bin/bash
# Profiles behind bench.py's roofline block, collected on the GPU box (gpurun).  Separate passes: --stats, then the two
# HBM counters, then the SQ counters (gpurun refuses --pmc combined with the trace domains other than --kernel-trace).
# usage: tools/profile_round.sh r04   -> gpurun_out/<tag>_prof/{stats,fetch,write,sq}  + summaries in gpurun_out/<tag>_prof/
# The profiled command is the headline workload alone (--no-secondary), one timed block (--min-time 0): its LAST dispatches
# are the timed block and the kernel-timing pass that follows it (the same steps of the same population).
set -u
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0"
# unprofiled first: builds the population with the fork pool and caches it; the profiled passes then load the cache
python3 $CMD > $OUT/bench_unprofiled.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --min-time 0 > $OUT/bench_under_stats.json 2> $OUT/stats.err
timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $CMD > /dev/null 2> $OUT/fetch.err
timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $CMD > /dev/null 2> $OUT/write.err
timeout 420 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq -- python3 bench.py --steps 10 --warmup 2 --settle 80 --no-cpu-baseline --no-secondary --min-time 0 > /dev/null 2> $OUT/sq.err
python3 tools/collect_profiles.py stats $OUT/stats $OUT/${TAG}_lsystem65536_kernel_stats.csv
python3 tools/collect_profiles.py trace $OUT/stats $OUT/${TAG}_kernel_trace_timed_region.json 240 "python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --min-time 0"   # 60 steps x 4 step groups (the kernel-timing pass)
python3 tools/trace_overlap.py $OUT/stats 960 > $OUT/${TAG}_kernel_overlap.txt 2>&1
python3 tools/collect_profiles.py pmc $OUT/fetch $OUT/write $OUT/${TAG}_pmc_traffic.json "python3 $CMD"
python3 tools/collect_profiles.py sq $OUT/sq $OUT/${TAG}_sq_counters.json "python3 bench.py --steps 10 --warmup 2 --settle 80 --no-cpu-baseline --no-secondary --min-time 0" 4
# the raw traces are large: keep the summaries only
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq
ls -la $OUT
