#!/usr/bin/env python3
"""Turn rocprofv3 CSV output directories into the small JSON / CSV summaries kept under profiles/.

  collect_profiles.py stats  <dir> <out.csv>                kernel_stats.csv of a `--kernel-trace --stats` run
  collect_profiles.py pmc    <fetch_dir> <write_dir> <out.json> <label>
                                                            HBM bytes per launch per kernel from the separate
                                                            `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes
  collect_profiles.py sq     <dir> <out.json> <label>       SQ counters per launch per kernel (lanes per VALU instruction)
  collect_profiles.py trace  <dir> <out.json> <n> <label>   per-kernel mean duration over the LAST n dispatches of each
                                                            kernel in kernel_trace.csv (= bench.py's timed region; the
                                                            --stats table averages the settle phase in as well)

Per-launch means are taken over the last `--tail` launches of each kernel (default 30: the timed region, settled
state).  Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB.  On gfx950 FETCH_SIZE counts a 128-byte fabric read
request as 64 bytes for wide coalesced loads (MI355X_MICROARCH.md, HBM): the corrected figure (x2) is reported next to
the raw one; this path loads 4 bytes per lane (64 consecutive words per wave), a width the guide lists as uncalibrated,
so the corrected number is an upper bound and the raw one a lower bound of the real fetch traffic.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def find(d, pattern):
    hits = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit("no %s under %s" % (pattern, d))
    return hits[0]


def short(name):
    name = re.sub(r"\s*\(.*$", "", name.replace("void ", ""))
    return name.strip()


def counters(d, tail):
    """kernel -> counter -> mean over the last `tail` dispatches."""
    per = defaultdict(lambda: defaultdict(dict))   # kernel -> dispatch id -> counter -> value
    with open(find(d, "*counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            per[k][int(row["Dispatch_Id"])][row["Counter_Name"]] = per[k][int(row["Dispatch_Id"])].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    out = {}
    for k, disp in per.items():
        ids = sorted(disp)[-tail:]
        names = sorted({c for i in ids for c in disp[i]})
        out[k] = {c: sum(disp[i].get(c, 0.0) for i in ids) / len(ids) for c in names}
        out[k]["launches_averaged"] = len(ids)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tail = 30
    for a in sys.argv[1:]:
        if a.startswith("--tail="):
            tail = int(a.split("=")[1])
    mode = args[0]
    if mode == "stats":
        src = find(args[1], "*kernel_stats.csv")
        with open(src) as f, open(args[2], "w") as g:
            g.write(f.read())
    elif mode == "pmc":
        fetch, write = counters(args[1], tail), counters(args[2], tail)
        kernels = {}
        for k in sorted(set(fetch) | set(write)):
            fk, wk = fetch.get(k, {}).get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
            kernels[k] = {"FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                          "hbm_bytes_per_launch_raw": (fk + wk) * 1024.0,
                          "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
        json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- " + args[4]
                   + " (two separate passes)",
                   "note": "per-launch means over the last %d launches of each kernel; rocprofv3 KB units; hbm_bytes_per_launch = "
                           "2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md for coalesced reads; upper bound "
                           "for this path's 4-byte-per-lane loads), hbm_bytes_per_launch_raw = FETCH_SIZE + WRITE_SIZE (lower bound)" % tail,
                   "kernels": kernels}, open(args[3], "w"), indent=1)
    elif mode == "sq":
        c = counters(args[1], tail)
        for k, v in c.items():
            if v.get("SQ_INSTS_VALU"):
                v["active_lanes_per_valu_inst"] = v.get("SQ_THREAD_CYCLES_VALU", 0.0) / v["SQ_INSTS_VALU"]
        json.dump({"command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES "
                   "--kernel-trace --output-format csv -- " + args[3],
                   "note": "means over the last %d launches of each kernel; active_lanes_per_valu_inst = SQ_THREAD_CYCLES_VALU / "
                           "SQ_INSTS_VALU (64 = every lane active)" % tail,
                   "step_groups": int(args[4]) if len(args) > 4 else None,
                   "kernels": c}, open(args[2], "w"), indent=1)
    elif mode == "trace":
        n = int(args[3])
        per = defaultdict(list)
        with open(find(args[1], "*kernel_trace.csv")) as f:
            for row in csv.DictReader(f):
                per[short(row["Kernel_Name"])].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
        out = {}
        for k, v in per.items():
            v.sort()
            d = [e - b for b, e in v]
            out[k] = {"dispatches": len(d), "avg_us_all": sum(d) / len(d) / 1e3,
                      "avg_us_last_%d" % n: sum(d[-n:]) / len(d[-n:]) / 1e3}
        json.dump({"command": "rocprofv3 --kernel-trace --stats --output-format csv -- " + args[4],
                   "note": "mean kernel duration from kernel_trace.csv; the last %d dispatches of the step kernels are bench.py's "
                           "timed region (steps x step groups) -- the number its HIP-event figure roofline.avg_launch_ms must "
                           "agree with; avg_us_all (what --stats prints) also averages the cheaper settle phase" % n,
                   "kernels": out}, open(args[2], "w"), indent=1)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
