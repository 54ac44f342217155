"""Control loop for DEVICE-RESIDENT vectorised environments: the counterpart of the reference's `Agent`
(General/QLearning/q_agent.py:171-222) when the env steps on the GPU too. Same schedule rules, applied to a vector
of envs whose episodes end asynchronously:

  * epsilon decays by `epsilon_decay_rate` per finished episode *per env* (q_agent.py:120-121, :202), floored at
    `min_epsilon`;
  * one `_step` per `train_frequency` vector env steps once `training_start` transitions exist (:186-187);
  * target hard copy every `replace_frequency` finished episodes per env (:192-193), and at start;
  * stop when the mean return of the episodes finished in the last window exceeds `reward_to_reach` (:219-222).

`term_reward` (CartPole): reward of the terminating step. The reference's target rule keeps `q + r` at terminals
(q_learning_functions.py:58), so the default here is -1 (a failure penalty); gym's +1 would reward falling.

All numerical work happens inside `Engine.train_iters` (one hipGraph launch per `chunk` iterations); the host only
reads two episode counters per chunk.
"""
from __future__ import annotations

import math

import torch


class VectorAgent:
    def __init__(self, engine, n_envs, batch_size, *, env="cartpole", max_steps=500, term_reward=-1.0, epsilon=1.0,
                 epsilon_decay_rate=0.99, min_epsilon=0.15, training_start=None, train_frequency=4,
                 replace_frequency=20, reward_to_reach=195.0, window_episodes=None, chunk=20, per_beta=(0.4, 1.0),
                 verbose=0):
        self.e, self.n_envs, self.B = engine, int(n_envs), int(batch_size)
        self.epsilon, self.decay, self.min_eps = float(epsilon), float(epsilon_decay_rate), float(min_epsilon)
        self.training_start = int(training_start if training_start is not None else 4 * batch_size)
        self.train_frequency, self.replace_frequency = int(train_frequency), int(replace_frequency)
        self.reward_to_reach = float(reward_to_reach)
        self.window = int(window_episodes if window_episodes is not None else max(50, n_envs))
        self.chunk, self.per_beta, self.verbose = int(chunk), per_beta, verbose
        self.history = []                      # (updates, finished episodes, mean return of the last window)
        engine.env_config(env, max_steps, term_reward)
        dev = engine.device
        g = torch.Generator(device=dev); g.manual_seed(engine.cfg.seed + 17)
        obs0 = (torch.rand(self.n_envs, engine.cfg.obs_dim, device=dev, generator=g) * 0.1 - 0.05) if env == "cartpole" \
            else torch.randn(self.n_envs, engine.cfg.obs_dim, device=dev, generator=g)
        engine.env_reset(obs0)
        engine.set_epsilon(self.epsilon)
        engine.sync_target()
        self.updates = 0

    def inject(self, gamma, epsilon, epsilon_decay_rate, min_epsilon, replace_frequency, batch_size, train_frequency,
               rebuild_closures=False):
        """ParamAgent.inject (General/QLearning/hyperparameter_optimization.py:76-91) for the device-resident loop: the seven
        searched hyper-parameters are replaced between two training runs; epsilon lives on the device, batch size / train
        frequency select the loop graph. gamma: the reference only rebinds `_gamma` (:84) -- its jitted q-target closure
        (q_agent.py:111, built once in the constructor) keeps the constructor's discount, so an injected gamma has NO effect
        on training there. Same here by default, exactly as the `ParamAgent` mirror; `rebuild_closures=True` (not in the
        reference) applies it: gamma is baked into the captured launches and dqn_set_gamma drops and re-captures them."""
        self.gamma = float(gamma)                                            # :84 (an attribute, as in the reference)
        if rebuild_closures:
            self.e.set_gamma(self.gamma)
        self.epsilon, self.decay, self.min_eps = float(epsilon), float(epsilon_decay_rate), float(min_epsilon)
        self.e.set_epsilon(self.epsilon)
        self.replace_frequency, self.train_frequency = int(replace_frequency), int(train_frequency)
        if int(batch_size) > self.e.cfg.max_batch:
            raise ValueError(f"batch_size {batch_size} exceeds the engine's max_batch {self.e.cfg.max_batch}")
        self.B = int(batch_size)

    def _mean_return(self, ep0, st0, ep1, st1):
        return (st1 - st0) / (ep1 - ep0) if ep1 > ep0 else float("nan")

    def training(self, max_updates):
        e = self.e
        st = e.stream
        with torch.cuda.stream(st):
            while e.replay_size()[0] < min(self.training_start, e.cfg.capacity):        # warm-up: act only (:186)
                e.actor_step(st)
            win = [e.env_stats()]
            eps_marker = repl_marker = win[0][0]
            while self.updates < max_updates:
                e.train_iters(self.chunk, self.train_frequency, self.B, st)
                self.updates += self.chunk
                ep, steps = e.env_stats()
                win.append((ep, steps))
                while len(win) > 2 and ep - win[1][0] >= self.window:
                    win.pop(0)
                mean_ret = self._mean_return(win[0][0], win[0][1], ep, steps)
                # schedules in units of "finished episodes per env"
                done_per_env = (ep - eps_marker) / self.n_envs
                if done_per_env >= 1.0:
                    k = math.floor(done_per_env)
                    self.epsilon = max(self.epsilon * self.decay ** k, self.min_eps)          # :121
                    e.set_epsilon(self.epsilon)
                    eps_marker += k * self.n_envs
                if (ep - repl_marker) / self.n_envs >= self.replace_frequency:                # :192-193
                    e.sync_target()
                    repl_marker = ep
                if self.per_beta is not None and e.cfg.use_per:
                    b0, b1 = self.per_beta
                    e.set_schedule(per_beta=b0 + (b1 - b0) * min(1.0, self.updates / max_updates))
                self.history.append((self.updates, ep, mean_ret))
                if self.verbose and len(self.history) % 50 == 0:
                    print(f"updates {self.updates}  episodes {ep}  eps {self.epsilon:.3f}  mean return {mean_ret:.1f}")
                if ep - win[0][0] >= self.window and mean_ret > self.reward_to_reach:         # :219
                    break
        return self.history
