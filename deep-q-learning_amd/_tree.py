"""haiku-style parameter trees <-> the flat f32 layout of the C ABI (include/dqn_hip.h):
w1 b1 w2 b2 wv bv wa ba, w is [in,out] row-major (LunarLander/dddqn.py:19-22)."""
from __future__ import annotations

import torch

NAMES = ("model/~/linear", "model/~/linear_1", "model/~/linear_2", "model/~/linear_3")


class Params(dict):
    """dict[str, dict['w'|'b', tensor]] that remembers the flat device tensor it was cut from, so that
    passing it back to the library costs no copy"""
    flat = None


def shapes(dims):
    D, H1, H2, A = dims
    return [(NAMES[0], "w", (D, H1)), (NAMES[0], "b", (H1,)), (NAMES[1], "w", (H1, H2)), (NAMES[1], "b", (H2,)),
            (NAMES[2], "w", (H2, 1)), (NAMES[2], "b", (1,)), (NAMES[3], "w", (H2, A)), (NAMES[3], "b", (A,))]


def unflatten(flat, dims):
    out, o = Params(), 0
    for mod, leaf, shp in shapes(dims):
        n = 1
        for s in shp:
            n *= s
        out.setdefault(mod, {})[leaf] = flat[o:o + n].view(shp)
        o += n
    out.flat = flat
    return out


def flatten(tree):
    if isinstance(tree, Params) and tree.flat is not None:
        return tree.flat
    leaves = []
    for mod in NAMES:
        leaves += [torch.as_tensor(tree[mod]["w"]).reshape(-1), torch.as_tensor(tree[mod]["b"]).reshape(-1)]
    return torch.cat([t.to(torch.float32) for t in leaves])


def dims_of(tree):
    w1, w2, wa = tree[NAMES[0]]["w"], tree[NAMES[1]]["w"], tree[NAMES[3]]["w"]
    return (int(w1.shape[0]), int(w1.shape[1]), int(w2.shape[1]), int(wa.shape[1]))
