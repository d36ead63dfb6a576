#!/usr/bin/env python3
"""Reduce the counter passes of tools/train_profile.sh: per 50-step launch of rem2d_step_train_kernel (the modal grid size) and
per env-step of the population -> JSON on stdout (profiles/<round>_step_train_counters.json, read by bench.py)."""
import collections
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]
STEPS = 50
acc = collections.defaultdict(lambda: collections.defaultdict(float))   # (dispatch, grid) -> counter -> value
for f in sorted(glob.glob(os.path.join(out_dir, "counters*.csv"))):
    for r in csv.DictReader(open(f)):
        if "rem2d_step_train" not in r.get("Kernel_Name", ""):
            continue
        acc[(os.path.basename(f), int(r["Dispatch_Id"]), int(r["Grid_Size"]))][r["Counter_Name"]] += float(r["Counter_Value"])
grids = collections.Counter(g for (_, _, g) in acc)
modal = max(grids, key=lambda g: (g, grids[g])) if grids else 0     # the longest launches: the 50-step trains
per = collections.defaultdict(list)
for (f, d, g), cs in acc.items():
    if g == modal:
        for c, v in cs.items():
            per[c].append(v)
launch = {c: sum(v) / len(v) for c, v in per.items()}
res = {"command": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 50 --warmup 40 --settle 60 --steps-per-launch 50 "
                  "--no-cpu-baseline --no-secondary --min-time 0 (one pass per counter group, tools/train_profile.sh)",
       "kernel": "rem2d_step_train_kernel", "steps_per_launch": STEPS, "grid_size": modal,
       "launches_averaged": {c: len(v) for c, v in per.items()}, "per_launch": launch, "step_groups": 1}
pe = {c: v / STEPS for c, v in launch.items()}
if "FETCH_SIZE" in launch and "WRITE_SIZE" in launch:   # rocprofv3 KB units; FETCH doubled: the gfx950 correction of MI355X_MICROARCH.md
    pe["hbm_bytes"] = (2.0 * launch["FETCH_SIZE"] + launch["WRITE_SIZE"]) * 1024.0 / STEPS
    pe["hbm_bytes_raw"] = (launch["FETCH_SIZE"] + launch["WRITE_SIZE"]) * 1024.0 / STEPS
if "SQ_THREAD_CYCLES_VALU" in launch and launch.get("SQ_INSTS_VALU"):
    res["active_lanes_per_valu_inst"] = launch["SQ_THREAD_CYCLES_VALU"] / launch["SQ_INSTS_VALU"]
res["per_env_step"] = pe
print(json.dumps(res, indent=1))
