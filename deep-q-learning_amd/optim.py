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
        raise NotImplementedError("the update is fused into train_step (generate_train_step); "
                                  "optimizer.update is not a separate device pass in this build")


def adamw(learning_rate, b1=0.9, b2=0.999, eps=1e-8, weight_decay=1e-4):
    return GradientTransformation("adamw", learning_rate, b1, b2, eps, weight_decay)


def adam(learning_rate, b1=0.9, b2=0.999, eps=1e-8):
    return GradientTransformation("adam", learning_rate, b1, b2, eps, 0.0)
