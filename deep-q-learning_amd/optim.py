"""Optimizer descriptions with the call surface the reference takes from optax
(Test/lunar_lander.py:48 `optax.adamw(LEARNING_RATE)`, Test/lunar_lander_hyper_params.py:41
`optax.adam(...)`). The arithmetic itself runs in the HIP library (k_adam / the dW epilogue);
these objects only carry hyper-parameters and build / hold the optimizer state pytree:
    adamw -> (ScaleByAdamState(count, mu, nu), EmptyState(), EmptyState())
    adam  -> (ScaleByAdamState(count, mu, nu), EmptyState())
"""
from __future__ import annotations

from collections import namedtuple
from dataclasses import dataclass

import torch

ScaleByAdamState = namedtuple("ScaleByAdamState", "count mu nu")
EmptyState = namedtuple("EmptyState", "")


@dataclass(frozen=True)
class GradientTransformation:
    kind: str                    # "adam" | "adamw"
    learning_rate: float
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 0.0

    def init(self, params):
        """zeroed moments with the tree structure of `params`, count = 0 (optax semantics)"""
        zeros = {mod: {leaf: torch.zeros_like(t) for leaf, t in leaves.items()} for mod, leaves in params.items()}
        zeros2 = {mod: {leaf: torch.zeros_like(t) for leaf, t in leaves.items()} for mod, leaves in params.items()}
        adam = ScaleByAdamState(count=0, mu=zeros, nu=zeros2)
        return (adam, EmptyState(), EmptyState()) if self.kind == "adamw" else (adam, EmptyState())

    def update(self, grads, state, params=None):
        """optax's `updates, opt_state = optimizer.update(grads, opt_state, params)` (q_learning_functions.py:24):
        one dqn_optimizer_step (k_adam) on a handle of the parameters' shape -- bit-exact scale_by_adam ->
        (adamw: add_decayed_weights) -> scale(-lr). Returns (updates, new_state); `updates` remembers the parameters
        the device pass produced, so that apply_updates(params, updates) returns exactly those bits."""
        from . import _lib as L
        from ._tree import dims_of, flatten, unflatten
        from .engine import Engine, EngineConfig
        if params is None:
            if self.kind == "adamw":
                raise ValueError("adamw needs `params` (weight decay), as optax.adamw does")
            params = {mod: {leaf: torch.zeros_like(torch.as_tensor(t)) for leaf, t in leaves.items()} for mod, leaves in grads.items()}
        dims = dims_of(params)
        cache = _ENGINES
        e = cache.get((dims, self))
        if e is None:
            e = cache[(dims, self)] = Engine(EngineConfig(obs_dim=dims[0], hidden1=dims[1], hidden2=dims[2], num_actions=dims[3], capacity=1,
                                                          max_batch=64, optimizer=self.kind, lr=float(self.learning_rate), b1=float(self.b1),
                                                          b2=float(self.b2), eps=float(self.eps), weight_decay=float(self.weight_decay)))
        adam = state[0]
        p_flat = flatten(params)
        e.set_params(p_flat, L.BUF_PARAMS); e.set_params(flatten(adam.mu), L.BUF_MU); e.set_params(flatten(adam.nu), L.BUF_NU)
        e.set_params(flatten(grads), L.BUF_GRAD)
        e.set_opt_count(int(adam.count))
        e.optimizer_step()
        new_p = e.get_params(L.BUF_PARAMS)
        upd = Updates(unflatten(new_p - p_flat.to(new_p.device), dims))
        upd.flat = None
        upd.new_params = unflatten(new_p, dims)
        new_state = (ScaleByAdamState(int(adam.count) + 1, unflatten(e.get_params(L.BUF_MU), dims), unflatten(e.get_params(L.BUF_NU), dims)),) + tuple(state[1:])
        return upd, new_state


_ENGINES = {}


class Updates(dict):
    """the `updates` pytree of GradientTransformation.update; new_params = the parameters k_adam wrote"""
    new_params = None
    flat = None


def apply_updates(params, updates):
    """optax.apply_updates (q_learning_functions.py:25): params + updates -- for an Updates object of
    GradientTransformation.update the device pass's own result (same value, no second rounding)"""
    if isinstance(updates, Updates) and updates.new_params is not None:
        return updates.new_params
    return {mod: {leaf: torch.as_tensor(params[mod][leaf]) + torch.as_tensor(u) for leaf, u in leaves.items()} for mod, leaves in updates.items()}


def adamw(learning_rate, b1=0.9, b2=0.999, eps=1e-8, weight_decay=1e-4):
    return GradientTransformation("adamw", learning_rate, b1, b2, eps, weight_decay)


def adam(learning_rate, b1=0.9, b2=0.999, eps=1e-8):
    return GradientTransformation("adam", learning_rate, b1, b2, eps, 0.0)
