set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05_train_prof2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --steps 50 --warmup 50 --settle 50 --no-cpu-baseline --no-secondary --min-time 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 200 --warmup 50 --settle 50 --steps-per-launch 50 --no-cpu-baseline --no-secondary --min-time 0 > $O/bench_under_stats.json 2> $O/stats.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r05_step_train_kernel_stats.csv
rm -rf $O/stats
head -3 $O/r05_step_train_kernel_stats.csv
python3 -c "
import json; b=json.load(open('$O/bench_under_stats.json')); r=b['roofline']; print(b['value'], r['avg_launch_ms'], r['launches'], r['frac'], r['traffic'], r['valu_issue'])"
timeout 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 -c "
import json; b=json.load(open('$O/bench_default.json')); r=b['roofline']; print(b['value'], b['ms_per_step'], r['avg_launch_ms'], r['launches'], r['frac'], r['traffic'], r['valu_issue']['frac'] if r['valu_issue'] else None, {k:round(v['value']/1e6,2) for k,v in b['secondary'].items() if isinstance(v,dict) and 'value' in v})"
