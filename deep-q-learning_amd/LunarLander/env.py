"""ObsWrapper with the reference's behaviour (LunarLander/env.py:9-31): the wrapped env's observation gets
the fraction of the episode elapsed appended and a leading axis, as float32. gym is not a dependency here:
any object with reset() and step(action) -> (obs, reward, done, info) works (gym < 0.26 API, as the
reference uses, env.py:25,30)."""
from __future__ import annotations

import numpy as np


class _Space:
    def __init__(self, n=None, shape=None):
        self.n, self.shape = n, shape


class ObsWrapper:
    def __init__(self, environment, max_steps: int):
        self.env = environment
        self._step = 0                                     # :15
        self._max_steps = max_steps                        # :16
        base = getattr(getattr(environment, "observation_space", None), "shape", None)
        d = (int(base[-1]) + 1) if base else 9
        self.observation_space = _Space(shape=(1, d))      # :17 (1, 9) for LunarLander
        self.action_space = getattr(environment, "action_space", None)

    def observation(self, observation):
        frac = self._step / self._max_steps                # :20 python float (f64) division
        return np.append(np.asarray(observation), frac)[np.newaxis, ...].astype(np.float32)   # :21

    def step(self, action):
        self._step += 1                                    # :24
        observation, reward, done, info = self.env.step(action)   # :25
        return self.observation(observation), reward, done, info

    def reset(self, **kwargs):
        self._step = 0                                     # :29
        return self.observation(self.env.reset())          # :30-31
