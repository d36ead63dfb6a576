set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r02_extra; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for wl in chain8 cppn_hardcore; do
  python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$wl -- python3 bench.py --workload $wl --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_${wl}_under_stats.json 2> $O/$wl.err
  python3 tools/collect_profiles.py stats $O/$wl $O/r02_b_${wl}_kernel_stats.csv
  python3 tools/collect_profiles.py trace $O/$wl $O/r02_b_${wl}_kernel_trace_timed_region.json 180 "python3 bench.py --workload $wl --steps 60 --warmup 10 --no-cpu-baseline"
  rm -rf $O/$wl
done
ls $O
