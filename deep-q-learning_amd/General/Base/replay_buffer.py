"""ReplayBuffer / sample_batch with the reference's call surface (General/Base/replay_buffer.py:10-85);
the ring lives in HBM inside a library handle, `add` is k_replay_add, `sample_batch` is the Philox-indexed
(or caller-indexed) gather k_sample_uniform. The numba RNG of the reference (:77) is opaque, so indices are
an explicit input/output here: sample_batch(..., indices=...) reproduces any given draw exactly.
"""
from __future__ import annotations

import numpy as np
import torch

from ... import _lib as L
from ...engine import Engine, EngineConfig

_RINGS = {}       # data_ptr of the states view -> ReplayBuffer (sample_batch receives arrays, not the buffer)


class ReplayBuffer:
    def __init__(self, buffer_size, obs_shape, ac_shape, prioritized=False, max_batch=1024):
        # obs_shape = (buffer_size, D), ac_shape = (buffer_size,)  (Test/lunar_lander.py:40-41)
        self._buffer_size = int(buffer_size)
        D = int(obs_shape[-1])
        self._engine = Engine(EngineConfig(obs_dim=D, hidden1=16, hidden2=16, num_actions=1,
                                           capacity=self._buffer_size, use_per=bool(prioritized), max_batch=max_batch))
        e, N = self._engine, self._buffer_size
        self._states = e.buffer(L.BUF_STATES).view(N, D)                   # :28
        self._actions = e.buffer(L.BUF_ACTIONS, torch.int32)               # :29 (int64 in the reference)
        self._rewards = e.buffer(L.BUF_REWARDS)                            # :30
        self._observations = e.buffer(L.BUF_OBSERVATIONS).view(N, D)       # :31
        self._dones = e.buffer(L.BUF_DONES, torch.uint8)                   # :32 (bool in the reference)
        self._counter = 0                                                  # :33 host mirror
        self._num_samples = 0                                              # :34
        self._draws = 0
        _RINGS[self._states.data_ptr()] = self

    size = property(lambda self: self._num_samples)
    states = property(lambda self: self._states)
    actions = property(lambda self: self._actions)
    rewards = property(lambda self: self._rewards)
    observations = property(lambda self: self._observations)
    dones = property(lambda self: self._dones)

    def add(self, state, action, reward, observation, done):
        """:58-65 -- one transition, or a batch of n rows at consecutive slots"""
        s = np.atleast_2d(np.asarray(state.cpu() if isinstance(state, torch.Tensor) else state, np.float32))
        n = s.shape[0]
        self._engine.replay_add(state if isinstance(state, torch.Tensor) else s,
                                np.reshape(action.cpu() if isinstance(action, torch.Tensor) else action, n),
                                np.reshape(reward.cpu() if isinstance(reward, torch.Tensor) else reward, n),
                                observation if isinstance(observation, torch.Tensor) else np.atleast_2d(np.asarray(observation, np.float32)),
                                np.reshape(done.cpu() if isinstance(done, torch.Tensor) else done, n))
        self._counter += n
        self._num_samples = min(self._counter, self._buffer_size)


def sample_batch(num_samples, states, actions, rewards, observations, dones, batch_size, indices=None, seed=0):
    """:68-85. `states` must be the `.states` view of a ReplayBuffer (that is what Agent._step passes,
    q_agent.py:147-153)."""
    ring = _RINGS.get(states.data_ptr()) if isinstance(states, torch.Tensor) else None
    if ring is None:
        raise TypeError("sample_batch needs the arrays of a deep_q_learning_amd ReplayBuffer (device-resident ring)")
    if int(num_samples) != ring.size:
        raise ValueError("num_samples must be the buffer's size")
    batch, idx = ring._engine.sample_uniform(int(batch_size), seed=seed, ctr=ring._draws, idx=indices)
    ring._draws += 1
    ring.last_indices = idx
    return batch
