"""The reference's RL-math factories (General/QLearning/q_learning_functions.py:14-85), same names and
positional signatures, backed by libdqn_hip.so instead of jax.jit closures. Arrays are torch tensors on
the GPU (numpy inputs are uploaded); parameter / optimizer pytrees keep haiku / optax structure.

Each factory owns one library handle sized on first use; results are returned as NEW pytrees (the
reference is functional), cut from fresh device tensors.
"""
from __future__ import annotations

import numpy as np
import torch

from ... import _lib as L
from ..._tree import Params, dims_of, flatten, unflatten
from ...optim import EmptyState, ScaleByAdamState


def _t(x, dtype=None):
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x))
    return x if dtype is None else x.to(dtype)


def _opt_cfg(optimizer):
    return dict(optimizer=optimizer.kind, lr=float(optimizer.learning_rate), b1=float(optimizer.b1),
                b2=float(optimizer.b2), eps=float(optimizer.eps), weight_decay=float(optimizer.weight_decay))


def generate_train_step(optimizer, model):
    """q_learning_functions.py:14-28: grads of compute_loss -> optimizer.update -> apply_updates"""

    def train_step(params, opt_state, states, q_targets):
        states = _t(states, torch.float32)
        states = states.reshape(-1, states.shape[-1])
        e = model.engine_for(states.shape[-1], states.shape[0], **_opt_cfg(optimizer))
        adam_state = opt_state[0]
        e.load(params)
        e.load(adam_state.mu, L.BUF_MU)
        e.load(adam_state.nu, L.BUF_NU)
        if e.__dict__.get("_count") != int(adam_state.count):
            e.set_opt_count(int(adam_state.count))
        e.train_step(states, _t(q_targets, torch.float32))                                  # :23-25
        dims = dims_of(params)
        new_p = unflatten(e.get_params(L.BUF_PARAMS), dims)
        new_mu = unflatten(e.get_params(L.BUF_MU), dims)
        new_nu = unflatten(e.get_params(L.BUF_NU), dims)
        count = int(adam_state.count) + 1
        e._count = count
        # the handle now holds exactly these tensors' contents: a following call with them uploads nothing
        e._loaded = {L.BUF_PARAMS: (new_p.flat.data_ptr(), new_p.flat._version, L.BUF_PARAMS),
                     L.BUF_MU: (new_mu.flat.data_ptr(), new_mu.flat._version, L.BUF_MU),
                     L.BUF_NU: (new_nu.flat.data_ptr(), new_nu.flat._version, L.BUF_NU)}
        return new_p, (ScaleByAdamState(count, new_mu, new_nu),) + tuple(opt_state[1:])

    return train_step


def generate_loss_computation(model):
    """:31-39: mean over the batch of the Huber loss summed over actions"""

    def compute_loss(params, states, q_targets):
        states = _t(states, torch.float32)
        states = states.reshape(-1, states.shape[-1])
        e = model.engine_for(states.shape[-1], states.shape[0])
        e.load(params)
        return e.loss(states, _t(q_targets, torch.float32))[0]

    return compute_loss


def generate_q_target_comp(model, gamma, env):
    """:42-64: double-Q targets (three forwards, argmax of the online net, target net's value), including
    the reference's terminal handling (:58) and q + delta * one_hot form (:59). `env` is only asked for
    env.action_space.n (:59)."""
    num_actions = int(env.action_space.n) if env is not None else model.num_actions
    if num_actions != model.num_actions:
        raise ValueError("env.action_space.n does not match the model's number of actions")

    def compute_q_targets(params, target_params, states, actions, rewards, observations, dones):
        states = _t(states, torch.float32)
        states = states.reshape(-1, states.shape[-1])
        e = model.engine_for(states.shape[-1], states.shape[0], gamma=float(gamma))
        e.load(params)
        e.load(target_params, L.BUF_TARGET)
        return e.q_targets(states, _t(actions), _t(rewards), _t(observations, torch.float32), _t(dones))

    return compute_q_targets


def action_computation(network):
    """:67-73: argmax over the network output (first maximum)"""

    def compute_action(params, state):
        state = _t(state, torch.float32)
        x = state.reshape(-1, state.shape[-1])
        e = network.engine_for(x.shape[-1], x.shape[0])
        e.load(params)
        a = e.act(x, epsilon=-1.0)             # epsilon < U(0,1) always: greedy
        return a[0] if a.numel() == 1 else a

    return compute_action


def preprocessing(states, actions, rewards, observations, dones):
    """:76-85: states -> device array, dones -> float32; the rest passes through (int64 actions become
    int32 as under JAX's default x64-off)"""
    from ...engine import default_device
    dev = default_device()
    states = _t(states, torch.float32).to(dev)
    dones = _t(dones).to(dev).to(torch.float32)
    return states, _t(actions).to(dev).to(torch.int32), _t(rewards).to(dev).to(torch.float32), \
        _t(observations, torch.float32).to(dev), dones
