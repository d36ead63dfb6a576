#!/usr/bin/env python3
"""Render a gym_rem2d_amd.statedump JSON-lines file to PNG frames (matplotlib, no display needed).
usage: render_dump.py dump.jsonl out_dir [--every N] [--creature K]"""
import json
import os
import sys

import matplotlib
matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402
from matplotlib.patches import Circle, Polygon  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gym_rem2d_amd.statedump import frame_to_draw_list  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    every, which = int(opt.get("every", 1)), int(opt.get("creature", 0))
    src, out = args[0], args[1]
    os.makedirs(out, exist_ok=True)
    with open(src) as f:
        head = json.loads(f.readline())
        frames = [json.loads(l) for l in f if l.strip()]
    tx, ty = head["terrain"]["x"], head["terrain"]["y"]
    for i, fr in enumerate(frames[::every]):
        prims = frame_to_draw_list(head, fr)[which]
        cx = fr["creatures"][which]["pose"][0][0]
        fig, ax = plt.subplots(figsize=(8, 4))
        ax.plot(tx, ty, color="#356635", lw=1.5)
        for box in head["terrain"]["boxes"]:
            ax.add_patch(Polygon(box, closed=True, color="#444444"))
        for p in prims:
            if p[0] == "polygon":
                ax.add_patch(Polygon(p[1], closed=True, facecolor="#7fa6d9", edgecolor="#1f3f66"))
            elif p[0] == "circle":
                ax.add_patch(Circle(p[1], p[2], facecolor="#d9a67f", edgecolor="#66401f"))
            else:
                ax.plot([p[1][0]], [p[1][1]], "k.", ms=3)
        wod = fr["creatures"][which]["wall_of_death"]
        ax.axvline(wod, color="red", lw=1)
        ax.set_xlim(cx - 6, cx + 6)
        ax.set_ylim(2, 10)
        ax.set_aspect("equal")
        ax.set_title("step %d  reward %.2f" % (fr["step"], fr["creatures"][which]["reward"]))
        fig.savefig(os.path.join(out, "frame_%05d.png" % i), dpi=80)
        plt.close(fig)
    print("wrote %d frames to %s" % (len(frames[::every]), out))


if __name__ == "__main__":
    main()
