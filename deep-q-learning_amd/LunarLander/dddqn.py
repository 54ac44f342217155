"""Dueling Q-network with the reference's parameter tree and call surface
(LunarLander/dddqn.py:11-34): Linear(H1) -> ReLU -> Linear(H2) -> ReLU -> {val: Linear(1),
adv: Linear(A)}, Q = val + adv - mean(adv). Parameters are a dict-of-dicts with haiku's names
(`model/~/linear`, `linear_1`, `linear_2` (val), `linear_3` (adv); leaves `w` [in,out] and `b`).
The forward pass is the HIP kernel k_qnet_fwd behind dqn_qnet_forward.

    model = transform(lambda *a: Model(num_actions)(*a))     # or Model(num_actions).transformed()
    params = model.init(seed, test_input);  q = model.apply(params, x)
"""
from __future__ import annotations

import numpy as np
import torch

from .._tree import NAMES, Params, flatten, unflatten


class _Trace:
    """argument used by transform() to find the Model a user lambda builds"""
    model = None


class Model:
    def __init__(self, num_actions: int, hidden=(32, 64)):
        self.num_actions = int(num_actions)            # dddqn.py:17
        self.hidden = (int(hidden[0]), int(hidden[1]))  # dddqn.py:19-20 (32, 64 in the reference)

    def transformed(self):
        return Transformed(self.num_actions, self.hidden)

    def __call__(self, x, return_features=False):
        if isinstance(x, _Trace):
            x.model = self
            return None
        raise TypeError("wrap the model with transform(...) / Model(...).transformed() and call .apply(params, x)")


class Transformed:
    """what hk.without_apply_rng(hk.transform(...)) gives the reference: .init(rng, x), .apply(params, x)"""

    def __init__(self, num_actions, hidden=(32, 64)):
        self.num_actions, self.hidden = num_actions, tuple(hidden)
        self._engines = {}

    def dims(self, obs_dim):
        return (int(obs_dim), self.hidden[0], self.hidden[1], self.num_actions)

    def init(self, rng, x):
        """hk.Linear defaults (dddqn.py:19-22): w ~ TruncatedNormal(stddev 1/sqrt(fan_in)) cut at 2 sigma, b = 0.
        rng: int seed or torch.Generator (the reference passes a jax PRNGKey)."""
        D = int(np.asarray(x.shape if hasattr(x, "shape") else np.shape(x))[-1])
        g = rng if isinstance(rng, torch.Generator) else torch.Generator().manual_seed(int(rng) & 0x7FFFFFFFFFFFFFFF)
        leaves = []
        for (k, n) in ((D, self.hidden[0]), (self.hidden[0], self.hidden[1]), (self.hidden[1], 1),
                       (self.hidden[1], self.num_actions)):
            w = torch.empty(k, n)
            sd = 1.0 / k ** 0.5
            torch.nn.init.trunc_normal_(w, std=sd, a=-2 * sd, b=2 * sd, generator=g)
            leaves += [w, torch.zeros(n)]
        flat = torch.cat([t.reshape(-1) for t in leaves])
        from ..engine import default_device
        return unflatten(flat.to(default_device()), self.dims(D))

    def engine_for(self, obs_dim, batch, **cfg):
        from ..engine import Engine, EngineConfig
        key = (obs_dim,) + tuple(sorted(cfg.items()))
        e = self._engines.get(key)
        if e is None or e.cfg.max_batch < batch:
            if e is not None:
                e.close()
            D, H1, H2, A = self.dims(obs_dim)
            e = Engine(EngineConfig(obs_dim=D, hidden1=H1, hidden2=H2, num_actions=A, capacity=1,
                                    max_batch=max(int(batch), 64), **cfg))
            self._engines[key] = e
        return e

    def apply(self, params, x, return_features=False):
        """Model.__call__ (dddqn.py:24-34)"""
        x = torch.as_tensor(np.asarray(x)) if not isinstance(x, torch.Tensor) else x
        x2 = x.reshape(-1, x.shape[-1])
        e = self.engine_for(x2.shape[-1], x2.shape[0])
        e.load(params)
        return e.forward(x2, return_features=return_features)


def transform(fn):
    """hk.transform counterpart for the one pattern the reference uses
    (Test/lunar_lander.py:47): `lambda *args: Model(NUM_ACTIONS)(*args)`"""
    t = _Trace()
    fn(t)
    if t.model is None:
        raise TypeError("transform expects `lambda *args: Model(num_actions)(*args)`")
    return Transformed(t.model.num_actions, t.model.hidden)


def without_apply_rng(t):
    return t
