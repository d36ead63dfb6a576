#!/usr/bin/env python3
"""Throughput of the population-diversity kernel (rem2d_tree_diversity) next to the CPU restatement.

    python tools/bench_diversity.py [--trees 16384] [--reps 5]

One JSON line: tree pairs/s and node-pair comparisons/s on the GPU (HIP events), the oracle's rate on a
bounded sample, and the kernel's bound (VALU: 4 compare/logic operations per node pair; the data --
N * 1 KB -- stays in L2)."""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from gym_rem2d_amd import get_module_list
    from gym_rem2d_amd.diversity import MAX_NODES, pack_positions, tree_positions
    from gym_rem2d_amd.encodings import LSystem
    base = []
    for seed in range(512):
        random.seed(seed)
        g = LSystem(get_module_list())
        base.append(tree_positions(g.create(8)))
    rng = np.random.RandomState(0)
    pops = [base[i] for i in rng.randint(0, len(base), size=args.trees)]
    pos, cnt = pack_positions(pops)

    import ctypes as C
    import torch
    from gym_rem2d_amd import _lib
    dev = torch.device("cuda", 0)
    pos_d, cnt_d = torch.from_numpy(pos).to(dev), torch.from_numpy(cnt).to(dev)
    out_d = torch.zeros(args.trees, dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch():
        _lib.check(_lib.lib().rem2d_tree_diversity(C.c_void_p(pos_d.data_ptr()), C.c_void_p(cnt_d.data_ptr()), args.trees,
                                                   MAX_NODES, C.c_void_p(out_d.data_ptr()), 0, st))
    launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    pairs = args.trees * (args.trees - 1)
    node_pairs = float(cnt.astype(np.int64).sum()) ** 2 - float((cnt.astype(np.int64) ** 2).sum())
    # CPU restatement on a bounded sample
    from oracle import diversity_oracle as D
    k = 160
    t0 = time.time()
    ref = D.tree_edit_distance(pops[:k])
    cpu_s = time.time() - t0
    sub = torch.zeros(k, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().rem2d_tree_diversity(C.c_void_p(pos_d.data_ptr()), C.c_void_p(cnt_d.data_ptr()), k, MAX_NODES,
                                               C.c_void_p(sub.data_ptr()), 0, st))
    assert sub.cpu().numpy().astype(np.float64).tolist() == ref
    print(json.dumps({"metric": "tree pairs/s (population diversity, AdvancedDataAnalysis.tree_edit_distance)",
                      "value": pairs / (ms * 1e-3), "unit": "pairs/s", "trees": args.trees, "ms_per_launch": ms,
                      "node_pair_compares_per_s": node_pairs / (ms * 1e-3),
                      "roofline": {"bound": "valu", "note": "4 VALU operations per node pair (2 f64 compares, and, or); "
                                   "positions N x 1 KB stay in L2", "valu_ops_per_s": 4 * node_pairs / (ms * 1e-3)},
                      "cpu_baseline": {"value": k * (k - 1) / cpu_s, "unit": "pairs/s", "cores": 1, "kind": "port",
                                       "sample": "%d trees, oracle/diversity_oracle.py (pure Python)" % k}}))


if __name__ == "__main__":
    main()
