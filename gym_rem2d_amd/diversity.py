"""Population diversity: the reference's "tree edit distance" (SURVEY.md 8f rank 4).

Restates ``DataAnalysis/AdvancedDataAnalysis.py:291-381``: every tree is laid out on a plane
(``get_tree_pos``: root at the origin, a child one unit away from its parent at an angle set by its
connection site and depth), and the distance between two trees is the number of nodes of either tree
whose position does not occur in the other (``compare_distance``; positions are compared with ``==``).
``tree_edit_distance(population)[i]`` is the sum of the distances from individual i to all others -- an
O(N^2 n^2) loop in the reference, here one HIP kernel over all pairs (``rem2d_tree_diversity``).

The layout runs on the host in binary64 with ``math.sin`` / ``math.cos`` exactly like the reference, so
two nodes are at "the same position" for this build iff they are for the reference on the same libm.
"""
import ctypes as C
import math

import numpy as np

MAX_NODES = 64  # nodes per tree the kernel accepts (the reference's max_size is 20, 40 with 0.cfg)


class _Vis:
    __slots__ = ("pos", "theta", "index", "handled")

    def __init__(self, pos, theta, index):
        self.pos, self.theta, self.index, self.handled = pos, theta, index, False


def tree_positions(tree):
    """``get_tree_pos`` (AdvancedDataAnalysis.py:315-365): list of (x, y) in expansion order."""
    nodes = tree.getNodes()
    vis = []
    for i, n in enumerate(nodes):
        if n.parent == -1 or n.parent is None:
            vis.append(_Vis((0, 0), 0, i))
            break
    for i in range(10):
        for j in range(len(vis)):
            vn = vis[j]
            if vn.handled:
                continue
            vn.handled = True
            for n in nodes:
                if n.parent == vn.index:
                    angle = n.parent_connection_coordinates.value[0]
                    theta = (angle / math.pow(2, (i + 1.5)) * (2 * math.pi)) + vn.theta
                    pos = ((math.sin(theta) * 1) + vn.pos[0], (math.cos(theta) * 1) + vn.pos[1])
                    vis.append(_Vis(pos, theta, n.index))
    return [(float(v.pos[0]), float(v.pos[1])) for v in vis]


def pack_positions(position_lists):
    """[N][MAX_NODES][2] float64 + [N] int32 node counts."""
    n = len(position_lists)
    pos = np.zeros((n, MAX_NODES, 2), dtype=np.float64)
    cnt = np.zeros(n, dtype=np.int32)
    for i, pl in enumerate(position_lists):
        if len(pl) > MAX_NODES:
            raise ValueError("tree %d has %d nodes; the diversity kernel takes at most %d" % (i, len(pl), MAX_NODES))
        cnt[i] = len(pl)
        if pl:
            pos[i, : len(pl)] = np.asarray(pl, dtype=np.float64)
    return pos, cnt


PAIRS_LIMIT = 32768  # above this the all-pairs kernel (N^2 n^2) gives way to the grouped form (N n log)


def diversity_grouped(pos, cnt, device):
    """The same sums without visiting pairs: group equal positions over the whole population.

    With m[c,p] = nodes of tree c at position p, T[p] = trees that contain p and M[p] = nodes at p,
        out[c] = (N-1) n_c + sum_{t != c} n_t - sum_{p in c} ( m[c,p] (T[p] - 1) + M[p] - m[c,p] ),
    which is compare_distance summed over t term by term.  Integer arithmetic on device tensors (torch sort /
    unique are plumbing here); used for populations too large for the all-pairs kernel, and checked against it.
    """
    import torch
    pos_d = torch.as_tensor(pos, device=device)
    cnt_d = torch.as_tensor(cnt, device=device).to(torch.int64)
    n, nmax = pos_d.shape[0], pos_d.shape[1]
    valid = torch.arange(nmax, device=device)[None, :] < cnt_d[:, None]
    tree = torch.arange(n, device=device)[:, None].expand(n, nmax)[valid]
    xy = (pos_d[valid] + 0.0)                                  # -0.0 -> +0.0, as == treats them
    keys = xy.view(torch.int64).clone()                        # [nodes, 2] bit patterns
    nan = torch.isnan(xy).any(dim=1)
    if bool(nan.any()):                                        # NaN equals nothing: give each such node its own key
        idx = torch.nonzero(nan).flatten()
        keys[idx, 0] = torch.iinfo(torch.int64).min + idx
        keys[idx, 1] = torch.iinfo(torch.int64).min
    _, p = torch.unique(keys, dim=0, return_inverse=True)
    n_pos = int(p.max().item()) + 1 if p.numel() else 0
    cp, m = torch.unique(tree * n_pos + p, return_counts=True)  # one entry per (tree, position)
    c_of, p_of = cp // n_pos, cp % n_pos
    T = torch.zeros(n_pos, dtype=torch.int64, device=device).index_add_(0, p_of, torch.ones_like(m))
    M = torch.zeros(n_pos, dtype=torch.int64, device=device).index_add_(0, p_of, m)
    shared = torch.zeros(n, dtype=torch.int64, device=device).index_add_(0, c_of, m * (T[p_of] - 1) + M[p_of] - m)
    return (n - 1) * cnt_d + (cnt_d.sum() - cnt_d) - shared


def diversity_from_positions(position_lists, device=None, method="auto"):
    """Sum over all other trees of compare_distance, for every tree; float64 numpy array.
    method: "pairs" = rem2d_tree_diversity (the reference's loop nest as one HIP kernel), "grouped" =
    diversity_grouped, "auto" = pairs up to PAIRS_LIMIT trees."""
    import torch
    from . import _lib
    if not torch.cuda.is_available():
        raise RuntimeError("gym_rem2d_amd needs a ROCm GPU (no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    pos, cnt = pack_positions(position_lists)
    n = len(position_lists)
    if n == 0:
        return np.zeros(0, dtype=np.float64)
    if method == "grouped" or (method == "auto" and n > PAIRS_LIMIT):
        return diversity_grouped(pos, cnt, dev).cpu().numpy().astype(np.float64)
    pos_d = torch.from_numpy(pos).to(dev)
    cnt_d = torch.from_numpy(cnt).to(dev)
    out_d = torch.zeros(n, dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(_lib.lib().rem2d_tree_diversity(C.c_void_p(pos_d.data_ptr()), C.c_void_p(cnt_d.data_ptr()), n, MAX_NODES,
                                               C.c_void_p(out_d.data_ptr()), dev.index or 0, st))
    return out_d.cpu().numpy().astype(np.float64)


def tree_edit_distance(population, device=None):
    """``tree_edit_distance(population)`` (AdvancedDataAnalysis.py:367-381): list of floats."""
    lists = [tree_positions(ind.genome.create(ind.tree_depth)) for ind in population]
    return [float(v) for v in diversity_from_positions(lists, device)]
