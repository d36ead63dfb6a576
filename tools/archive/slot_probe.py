#!/usr/bin/env python3
"""Scratch probe (GPU box): how many contact-solve bodies a wave executes per velocity iteration
(per phase of the pipeline period), from a settled state of the largest lane bucket."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from gym_rem2d_amd import make_terrain
from gym_rem2d_amd.world import BatchedWorld

workload = sys.argv[1] if len(sys.argv) > 1 else "lsystem"
morphs, desc = bench.build_population(workload, 65536, 0)
terrain = make_terrain(4, flat=True)
m = max(morphs, key=lambda mm: mm.lanes)
w = BatchedWorld(m.n_envs, m.lanes, flags=1)
w.set_terrain(terrain)
w.reset(m)
w.step(int(sys.argv[2]) if len(sys.argv) > 2 else 200)
torch.cuda.synchronize()
K = m.lanes
ct = (w.view("cinfo").cpu().numpy() & 0xff)            # [slot][lane]
ntouch = (ct > 0).sum(0)[: m.n_envs * K].reshape(-1)
jr = m["jround"].reshape(-1)
offc = (jr >> 8) & 0xff
per = (jr >> 16) & 0xff
per_env = per.reshape(m.n_envs, K).max(1)
P = np.repeat(per_env, K)
phase = np.where(P > 0, offc % np.maximum(P, 1), 0)
nw = (m.n_envs * K) // 64
nt = ntouch[: nw * 64].reshape(nw, 64)
ph = phase[: nw * 64].reshape(nw, 64)
Pw = P[: nw * 64].reshape(nw, 64).max(1)
solves = np.zeros(nw); slots = np.zeros(nw)
for c in range(4):
    mx = np.where(ph == c, nt, 0).max(1)
    solves += mx; slots += mx > 0
print("waves", nw, "mean period", Pw.mean())
print("touching bodies per wave %.1f, constraints per wave %.1f" % ((nt > 0).sum(1).mean(), nt.sum(1).mean()))
print("contact slots executed per iteration per wave %.2f, contact_solve bodies executed %.2f" % (slots.mean(), solves.mean()))
canon = (ph == (Pw[:, None] - 1))
print("touching bodies at canonical phase: %.3f" % (((nt > 0) & canon).sum() / max(1, (nt > 0).sum())))
print("useful lane-solves per executed solve: %.3f of 64" % (nt.sum() / solves.sum()))
# workgroup of 4 waves: canonical bodies compacted, stragglers per phase
g = nw // 4
ntg = nt[: g * 4].reshape(g, 256); phg = ph[: g * 4].reshape(g, 256); Pg = Pw[: g * 4].reshape(g, 4).max(1)
tot = np.zeros(g)
for c in range(4):
    sel = np.where(phg == c, ntg, 0)
    nb = (sel > 0).sum(1)
    tot += np.ceil(nb / 64.0) * sel.max(1)
print("4-wave workgroup, bodies compacted per phase: solves per iteration per workgroup %.2f (now %.2f)" % (tot.mean(), solves[: g * 4].reshape(g, 4).sum(1).mean()))
# ---- what a per-step cyclic shift of every creature's schedule would buy: greedy alignment of the
# creatures' touching-phase masks inside each wavefront
cpw = 64 // K
ntc = ntouch[: nw * 64].reshape(nw, cpw, K)
phc = phase[: nw * 64].reshape(nw, cpw, K)
Pc = P[: nw * 64].reshape(nw, cpw, K).max(2)
aligned = np.zeros(nw)
weighted_now = np.zeros(nw); weighted_al = np.zeros(nw)
for wv in range(0, nw, max(1, nw // 1500)):
    U = {}
    per = int(Pc[wv].max())
    if per <= 0:
        continue
    slots_max = np.zeros(per, dtype=np.int64)
    for c in range(cpw):
        m = np.zeros(per, dtype=np.int64)
        for k in range(K):
            if ntc[wv, c, k] > 0:
                m[phc[wv, c, k] % per] = max(m[phc[wv, c, k] % per], ntc[wv, c, k])
        best, bs = None, 0
        for s in range(per):
            r = np.roll(m, s)
            cost = np.maximum(slots_max, r).sum()
            if best is None or cost < best:
                best, bs = cost, s
        slots_max = np.maximum(slots_max, np.roll(m, bs))
    aligned[wv] = (slots_max > 0).sum()
    weighted_al[wv] = slots_max.sum()
    weighted_now[wv] = solves[wv]
sel = aligned > 0
print("sampled waves %d: contact slots per iteration now %.2f -> aligned %.2f; solves executed now %.2f -> aligned %.2f" %
      (sel.sum(), slots[sel].mean(), aligned[sel].mean(), weighted_now[sel].mean(), weighted_al[sel].mean()))
