#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV directory and reports, for the last launches of the run (the timed region), how
many kernels ran concurrently on average (sum of kernel durations / wall span), per queue launch counts and the
per-kernel average duration.   python3 tools/trace_overlap.py <dir> [n_last]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "rem2d_" in r["Kernel_Name"] and "reset" not in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                                 r.get("Queue_Id", "?")))
    rows.sort()
    rows = rows[-n_last:]
    # (the window must not span a pause between two passes of the program: keep what follows the last gap of > 5 ms)
    for i in range(len(rows) - 1, 0, -1):
        if rows[i][0] - max(r[1] for r in rows[max(0, i - 16):i]) > 5_000_000:
            rows = rows[i:]
            break
    if not rows:
        print("no rem2d kernels in trace")
        return
    span = max(r[1] for r in rows) - min(r[0] for r in rows)
    busy = sum(r[1] - r[0] for r in rows)
    print("last %d rem2d dispatches: wall span %.3f ms, sum of kernel durations %.3f ms -> %.2f kernels in flight on average"
          % (len(rows), span / 1e6, busy / 1e6, busy / span))
    # time with k kernels in flight
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    hist = defaultdict(int)
    cur, last = 0, ev[0][0]
    for t, dlt in ev:
        hist[cur] += t - last
        cur += dlt
        last = t
    print("time share by number of kernels in flight: " + ", ".join("%d: %.1f%%" % (k, 100.0 * v / span) for k, v in sorted(hist.items())))
    # phase relation of the step groups: how many of the LONG kernels (velocity / position) run side by side, and how much
    # of the time no long kernel of any group is running (all groups in their TOI / pre phase at once: the chip idles)
    evl = []
    for s, e, n, _ in rows:
        if "velpost" in n or "vel4" in n or "post_multi" in n or "step_multi" in n:
            evl.append((s, 1))
            evl.append((e, -1))
    if evl:
        evl.sort()
        histl = defaultdict(int)
        cur, last = 0, min(r[0] for r in rows)
        for t, dlt in evl:
            histl[cur] += t - last
            cur += dlt
            last = t
        histl[cur] += max(r[1] for r in rows) - last
        print("time share by number of velocity / position kernels in flight: " +
              ", ".join("%d: %.1f%%" % (k, 100.0 * v / span) for k, v in sorted(histl.items())))
    per = defaultdict(list)
    q = defaultdict(int)
    for s, e, n, qu in rows:
        per[n].append(e - s)
        q[qu] += 1
    for n, v in sorted(per.items()):
        print("  %-50s n=%4d avg %.4f ms max %.4f ms" % (n[:50], len(v), sum(v) / len(v) / 1e6, max(v) / 1e6))
    print("dispatches per queue: " + ", ".join("%s: %d" % kv for kv in sorted(q.items())))
    # per queue (= step group): its own kernels' average durations and the time from one pre to the next (its step)
    byq = defaultdict(list)
    for r in rows:
        byq[r[3]].append(r)
    for qu, rs in sorted(byq.items()):
        kd = defaultdict(list)
        for s, e, n, _ in rs:
            kd[n.split("<")[0].replace("rem2d_", "").replace("_multi_kernel", "").replace("_kernel", "")].append((e - s) / 1e3)
        starts = [s for s, e, n, _ in rs if "pre_multi" in n]
        per = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
        per.sort()
        print("  queue %s: %s | pre-to-pre median %.0f us, p90 %.0f, max %.0f" % (
            qu, ", ".join("%s avg %.0f p90 %.0f max %.0f us" % (k, sum(v) / len(v), sorted(v)[int(0.9 * (len(v) - 1))], max(v)) for k, v in sorted(kd.items())),
            per[len(per) // 2] if per else 0, per[int(0.9 * (len(per) - 1))] if per else 0, per[-1] if per else 0))


if __name__ == "__main__":
    main()
