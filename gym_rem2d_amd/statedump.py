"""State dump for external viewers.

The reference can only look at a run through its pyglet window (``Modular2D.render``, Modular2DEnv.py:655-738), which
needs a display and one env per process.  Here a frame is plain data: the terrain polyline and hardcore boxes once,
then per recorded step the pose of every body of the selected creatures (what ``render`` reads from
``body.fixtures / .transform``) plus the joint anchors (``joint.anchorA`` :725-738), the wall of death and the
reward.  One JSON object per line; ``tools/render_dump.py`` turns such a file into PNG frames with matplotlib, and
anything else (a browser, Blender, the reference's own viewer fed by ``frame_to_draw_list``) can read it.
"""
import json
import math

import numpy as np


def header(env, creatures):
    """Static part: terrain and the morphology of the selected creatures (population indices)."""
    terrain = env._terrain()
    xs, ys, polys = terrain.f32()
    where = _locate(env, creatures)
    out = {"kind": "rem2d_state_dump", "version": 1, "fps": 50,
           "terrain": {"x": [float(v) for v in xs], "y": [float(v) for v in ys],
                       "boxes": [[[float(p[0]), float(p[1])] for p in poly] for poly in np.asarray(polys).reshape(-1, 4, 2)]},
           "creatures": []}
    for e in creatures:
        w, local = where[e]
        shape = w.view("shape")[local].cpu().numpy()
        hx, hy = w.view("hx")[local].cpu().numpy(), w.view("hy")[local].cpu().numpy()
        parent = w.view("parent")[local].cpu().numpy()
        ax, ay = w.view("jax")[local].cpu().numpy(), w.view("jay")[local].cpu().numpy()
        bodies = []
        for k in range(w.lanes):
            if shape[k] == 0:
                continue
            b = {"slot": k, "shape": "box" if shape[k] == 1 else "circle", "parent": int(parent[k])}
            if shape[k] == 1:
                b["half_extents"] = [float(hx[k]), float(hy[k])]
            else:
                b["radius"] = float(hx[k])
            if parent[k] >= 0:
                b["anchor_in_parent"] = [float(ax[k]), float(ay[k])]   # revoluteJointDef.localAnchorA
            bodies.append(b)
        out["creatures"].append({"index": int(e), "bodies": bodies})
    return out


def frame(env, creatures, step):
    where = _locate(env, creatures)
    fr = {"step": int(step), "creatures": []}
    for e in creatures:
        w, local = where[e]
        px, py, ang = (w.view(k)[local].cpu().numpy() for k in ("px", "py", "ang"))
        shape = w.view("shape")[local].cpu().numpy()
        fr["creatures"].append({
            "index": int(e),
            "pose": [[float(px[k]), float(py[k]), float(ang[k])] for k in range(w.lanes) if shape[k] != 0],
            "wall_of_death": float(w.view("wod")[local]),
            "reward": float(w.view("reward")[local]), "done": bool(int(w.view("done")[local]))})
    return fr


def record_episode(env, path, steps, creatures=(0,), every=1):
    """Step a BatchedModular2D `steps` times and write header + one frame every `every` steps as JSON lines."""
    creatures = list(creatures)
    with open(path, "w") as f:
        f.write(json.dumps(header(env, creatures)) + "\n")
        f.write(json.dumps(frame(env, creatures, 0)) + "\n")
        done = 0
        while done < steps:
            n = min(every, steps - done)
            env.step(n)
            done += n
            f.write(json.dumps(frame(env, creatures, done)) + "\n")
    return path


def frame_to_draw_list(head, fr):
    """Polygons / circles in world coordinates, the primitives ``Modular2D.render`` draws (:689-724):
    [("polygon", [[x, y] * 4]) | ("circle", [x, y], r) | ("anchor", [x, y])] per creature."""
    out = []
    for c_head, c_fr in zip(head["creatures"], fr["creatures"]):
        prims = []
        poses = {b["slot"]: p for b, p in zip(c_head["bodies"], c_fr["pose"])}
        for b in c_head["bodies"]:
            x, y, a = poses[b["slot"]]
            ca, sa = math.cos(a), math.sin(a)
            if b["shape"] == "box":
                hx, hy = b["half_extents"]
                prims.append(("polygon", [[x + ca * u - sa * v, y + sa * u + ca * v]
                                          for u, v in ((-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy))]))
            else:
                prims.append(("circle", [x, y], b["radius"]))
            if b["parent"] >= 0 and b["parent"] in poses:
                px, py, pa = poses[b["parent"]]
                u, v = b["anchor_in_parent"]
                prims.append(("anchor", [px + math.cos(pa) * u - math.sin(pa) * v, py + math.sin(pa) * u + math.cos(pa) * v]))
        out.append(prims)
    return out


def _locate(env, creatures):
    where = {}
    want = set(int(e) for e in creatures)
    for w, idx in env.worlds:
        for local, e in enumerate(idx.tolist()):
            if e in want:
                where[e] = (w, local)
    missing = want - set(where)
    if missing:
        raise IndexError("creatures %s are not in this environment" % sorted(missing))
    return where
