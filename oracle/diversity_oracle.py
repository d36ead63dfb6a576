"""CPU restatement of the reference's pairwise tree distance -- TEST INFRASTRUCTURE ONLY.

Follows ``compare_distance`` / ``tree_edit_distance`` of the reference
(``DataAnalysis/AdvancedDataAnalysis.py:291-313`` and ``:367-381``) line for line in plain Python loops
(small cases only).  Pinned by ``tests/golden/diversity.json``, captured from the reference itself by
``tools/capture_golden.py``.  Only tests/ may import this; the product path is the HIP kernel
``rem2d_tree_diversity``.
"""


def compare_distance(tree, target):
    """tree, target: lists of (x, y).  Nodes of `tree` whose position is not in `target`, plus nodes of
    `target` that no node of `tree` sits on."""
    edit_value = 0
    found = [False] * len(target)
    for node in tree:
        same_node = False
        for k, t_node in enumerate(target):
            if node[0] == t_node[0] and node[1] == t_node[1]:
                same_node = True
                found[k] = True
        if not same_node:
            edit_value += 1
    for f in found:
        if not f:
            edit_value += 1
    return edit_value


def tree_edit_distance(position_lists):
    out = []
    for ci, c in enumerate(position_lists):
        div = 0.0
        for ti, t in enumerate(position_lists):
            if ti == ci:
                continue
            div += compare_distance(c, t)
        out.append(div)
    return out
