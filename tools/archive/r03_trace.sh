#!/bin/bash
# per-kernel durations of the timed region (kernel trace)   bash tools/r03_trace.sh <tag> [bench args...]
set -u
TAG=$1; shift
O=gpurun_out/r03_trace_$TAG; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --min-time 0.3 "$@" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 420 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --min-time 0 "$@" > $O/bench_traced.json 2> $O/trace.err
python3 tools/trace_overlap.py $O/trace 640 > $O/trace_overlap.txt 2>&1
rm -rf $O/trace
cat $O/trace_overlap.txt
