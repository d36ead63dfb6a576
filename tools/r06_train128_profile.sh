#!/bin/bash
# Round 6: rocprofv3 passes of the 128-lane step train on config 5's per-GPU share (131 072 L-system creatures, 100-step blocks in ABI
# calls of 50 steps): --kernel-trace --stats, then one --pmc pass for the VALU counters (gpurun refuses --pmc with other trace domains).
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r06_train128_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="--envs 131072 --steps 100 --warmup 50 --settle 50 --steps-per-launch 50 --no-cpu-baseline --no-secondary --min-time 0"
python3 bench.py $ARGS > $O/bench_unprofiled.json 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py $ARGS > $O/bench_under_stats.json 2> $O/stats.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r06_train128_kernel_stats.csv
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/raw1 -- python3 bench.py $ARGS > /dev/null 2> $O/err1.txt
f=$(find $O/raw1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/counters1.csv
python3 - <<'PY' > $O/r06_train128_counters.json
import collections, csv, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06_train128_prof")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(os.path.join(O, "counters1.csv"))):
    if "rem2d_step_train128" in r.get("Kernel_Name", ""):
        acc[(int(r["Dispatch_Id"]), int(r["Grid_Size"]))][r["Counter_Name"]] += float(r["Counter_Value"])
grids = collections.Counter(g for (_, g) in acc)
modal = max(grids, key=lambda g: (g, grids[g])) if grids else 0
per = collections.defaultdict(list)
for (d, g), cs in acc.items():
    if g == modal:
        for c, v in cs.items():
            per[c].append(v)
launch = {c: sum(v) / len(v) for c, v in per.items()}
out = {"kernel": "rem2d_step_train128_kernel<true, 4>", "creatures": 131072, "steps_per_launch": 50, "grid_size": modal,
       "launches_averaged": {c: len(v) for c, v in per.items()}, "per_launch": launch, "per_env_step": {c: v / 50 for c, v in launch.items()}}
if launch.get("SQ_INSTS_VALU"):
    out["active_lanes_per_valu_inst"] = launch["SQ_THREAD_CYCLES_VALU"] / launch["SQ_INSTS_VALU"]
print(json.dumps(out, indent=1))
PY
head -c 1200 $O/r06_train128_counters.json; grep -h "train128" $O/r06_train128_kernel_stats.csv | head -3
rm -rf $O/stats $O/raw1 $O/counters1.csv
