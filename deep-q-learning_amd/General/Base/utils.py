"""Persistence helpers with the reference's names (General/Base/utils.py:21-40). The reference pickles jax
DeviceArrays; this build writes the same dict-of-dicts (haiku names) and optimizer state as plain arrays in
`<dir>/params.npz` / `<dir>/opt_state.npz` -- data only, loadable with numpy.load(allow_pickle=False). A directory that holds
the REFERENCE's `params.pickle` / `opt_state.pickle` instead (e.g. its Test/lunar_lander/) is read token by token, without
unpickling (pickle_tokens.py): `generate_loading` resumes from a reference checkpoint."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from ..._tree import NAMES, Params, dims_of, flatten, unflatten
from ...optim import EmptyState, ScaleByAdamState


def stop_time(name, fun, *args):
    start = time.time()                                   # utils.py:13-18
    out = fun(*args)
    print("{}: {}s".format(name, time.time() - start))
    return out


def _tree_arrays(prefix, tree):
    return {f"{prefix}{mod}|{leaf}": t.detach().cpu().numpy() for mod, leaves in tree.items() for leaf, t in leaves.items()}


def generate_saving(directory):
    def save_state(params, opt_state):
        os.makedirs(directory, exist_ok=True)             # :23-24
        np.savez(os.path.join(directory, "params.npz"), **_tree_arrays("", params))
        adam = opt_state[0]
        np.savez(os.path.join(directory, "opt_state.npz"), count=np.int32(adam.count), n_empty=np.int32(len(opt_state) - 1),
                 **_tree_arrays("mu:", adam.mu), **_tree_arrays("nu:", adam.nu))
    return save_state


def generate_loading(directory, device=None):
    def load_state():
        from ...engine import default_device
        dev = device or default_device()

        def tree(z, prefix):
            out = {}
            for key in z.files:
                if prefix and not key.startswith(prefix):
                    continue
                if not prefix and ":" in key.split("|")[0]:
                    continue
                mod, leaf = key[len(prefix):].split("|")
                out.setdefault(mod, {})[leaf] = torch.as_tensor(z[key])
            return out
        if not os.path.exists(os.path.join(directory, "params.npz")) and os.path.exists(os.path.join(directory, "params.pickle")):
            from .pickle_tokens import read_adam_state, read_haiku_params
            as_t = lambda t: {m: {k: torch.as_tensor(v) for k, v in lv.items()} for m, lv in t.items()}   # noqa: E731
            p = as_t(read_haiku_params(os.path.join(directory, "params.pickle")))
            dims = dims_of(p)
            params = unflatten(flatten(p).to(dev), dims)
            count, mu, nu, n_empty = read_adam_state(os.path.join(directory, "opt_state.pickle"))
            adam = ScaleByAdamState(count, unflatten(flatten(as_t(mu)).to(dev), dims), unflatten(flatten(as_t(nu)).to(dev), dims))
            return params, (adam,) + tuple(EmptyState() for _ in range(n_empty))
        with np.load(os.path.join(directory, "params.npz"), allow_pickle=False) as z:
            p = tree(z, "")
        dims = dims_of(p)
        params = unflatten(flatten(p).to(dev), dims)
        with np.load(os.path.join(directory, "opt_state.npz"), allow_pickle=False) as z:
            mu, nu = tree(z, "mu:"), tree(z, "nu:")
            count, n_empty = int(z["count"]), int(z["n_empty"])
        adam = ScaleByAdamState(count, unflatten(flatten(mu).to(dev), dims), unflatten(flatten(nu).to(dev), dims))
        return params, (adam,) + tuple(EmptyState() for _ in range(n_empty))
    return load_state
