"""The reference's control loop (`Agent.training`, General/QLearning/q_agent.py:171-222) for BASELINE configs[4]'s shape: vector
envs that emit stacks of four 84x84 u8 frames, the Nature-CNN dueling Q-net, prioritized replay. Host logic only -- every
numerical step is a C-ABI call:

  act + env + add  dqn_cnn_env_step_synth (r03: ONE call per vector step -- CNN forward + epsilon-greedy (q_agent.py:137-141), the
                                          synthetic transition drawn on the device, ReplayBuffer.add into the frame ring
                                          (replay_buffer.py:58-65); the envs' current frames are the ring's own s' rows)
                   dqn_per_index_advance  on the index engine: the new positions enter its sum tree at the running maximum
                                          priority; the index holds POSITIONS only (r02 fed it rows of zeros)
  sample           dqn_per_sample         on the index engine (a dqn_handle of the same capacity)
  n-step           rows are 1-step; dqn_cnn_update_replay(n_step, n_envs) assembles the n-step transition that starts at a sampled
                   row from its n - 1 successors; the index is told of a step n - 1 steps late
  update           dqn_cnn_update_replay  (gather, three forwards, TD rule, backward, AdamW; q_agent.py:146-169)
  write-back       dqn_per_update_sorted  with |delta|

The env is synthetic (there is no ALE here and no network): frames are uniform u8 (Philox, drawn by the step kernel), rewards
Irwin-Hall normals, dones Bernoulli(p_done) -- the shape of PongNoFrameskip-v4 after the usual wrappers, not its dynamics.
"""
from __future__ import annotations

import torch

from ...cnn import CnnEngine
from ...engine import Engine, EngineConfig


class CnnVectorAgent:
    def __init__(self, n_envs=512, num_actions=6, capacity=1 << 14, batch_size=512, precision="bf16", gamma=0.99, epsilon=1.0,
                 epsilon_decay_rate=0.999, min_epsilon=0.1, train_frequency=4, replace_frequency=250, per_beta=0.4, p_done=0.01,
                 lr=1e-4, seed=0, n_step=1, device=None):
        if capacity % n_envs:
            raise ValueError("capacity must be a multiple of n_envs (whole vector steps per ring lap)")
        if not 1 <= n_step <= 8 or n_step * n_envs > capacity:
            raise ValueError("n_step must be 1..8 and n_step * n_envs must fit the ring")
        self.n_envs, self.B, self.gamma, self.seed, self.n_step = int(n_envs), int(batch_size), float(gamma), int(seed), int(n_step)
        self.epsilon, self.decay, self.min_eps = float(epsilon), float(epsilon_decay_rate), float(min_epsilon)
        self.train_frequency, self.replace_frequency, self.per_beta, self.p_done = int(train_frequency), int(replace_frequency), float(per_beta), float(p_done)
        self.cnn = CnnEngine(num_actions=num_actions, max_batch=max(n_envs, batch_size), precision=precision, device=device)
        self.cnn.replay_init(capacity)
        self.cnn.set_optimizer(lr=lr)
        # the PER tree: a dqn_handle used as a positions-only index (dqn_per_index_advance; its own ring is one float per position)
        self.index = Engine(EngineConfig(obs_dim=1, hidden1=16, hidden2=16, num_actions=2, capacity=capacity, use_per=True,
                                         max_batch=max(n_envs, batch_size), seed=seed), device=device)
        dev = self.cnn.device
        self.cnn.env_reset_synth(n_envs, seed)
        self.td_abs = torch.empty((self.B,), dtype=torch.float32, device=dev)
        # r03: the index works on a stream of its own. Its launches (one per vector step, the draw, the write-back) depend on the CNN
        # stream only through |delta| (write-back after the update) and feed it only idx / isw (update after the draw): they run
        # beside the envs' forward passes instead of between them. Caller-held sample buffers: nothing is allocated across streams.
        self.ist = self.index.stream
        self.ist.wait_stream(torch.cuda.current_stream(dev))
        self.ev_draw, self.ev_upd = torch.cuda.Event(), torch.cuda.Event()
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        self._bufs = (mk((self.B, 1), torch.float32), mk((self.B,), torch.int32), mk((self.B,), torch.float32), mk((self.B, 1), torch.float32),
                      mk((self.B,), torch.uint8), mk((self.B,), torch.int32), mk((self.B,), torch.float32))
        self.env_steps = self.updates = 0
        self.losses = []

    def close(self):
        self.cnn.close(); self.index.close()

    def init_params(self, flat, target_flat=None):
        self.cnn.set_params(flat); self.cnn.set_params(flat if target_flat is None else target_flat, target=True)

    def env_step(self):
        """one vector env step: act + synthetic transition + frame ring (one C-ABI call), then the index"""
        n = self.n_envs
        first = self.cnn.env_step_synth(self.epsilon, self.p_done)
        assert first == (self.env_steps * n) % self.cnn.capacity, (first, self.env_steps)      # the two rings move in lockstep
        if self.n_step == 1:
            with torch.cuda.stream(self.ist):
                self.index.per_index_advance(n)
        else:
            # the frame ring holds one row per env step; the n-step transition that starts at a row exists once its n - 1
            # successors do. So the PER index learns of step t - n + 1 when step t arrives (its own ring counter is n - 1 steps
            # behind: the same positions), and rows whose frames have just been overwritten are taken out of the draw until then.
            # (one launch: dqn_per_index_step; r02 / early r03: per_set_sorted(zeros) + per_index_advance = three launches and an index add)
            zero_n = n if self.env_steps * n >= self.cnn.capacity else 0
            adv = n if self.env_steps >= self.n_step - 1 else 0
            if zero_n or adv:
                with torch.cuda.stream(self.ist):
                    self.index.per_index_step(adv, first, zero_n)
        self.env_steps += 1
        self.epsilon = max(self.epsilon * self.decay, self.min_eps)

    def update(self, want_loss=False):
        cur = torch.cuda.current_stream(self.cnn.device)
        idx, isw = self._bufs[5], self._bufs[6]
        with torch.cuda.stream(self.ist):
            self.index.per_sample_into(self.B, self.per_beta, self.seed, self.updates, self._bufs)
            self.ev_draw.record(self.ist)
        cur.wait_event(self.ev_draw)
        loss = self.cnn.update_from_replay(idx, isw, self.gamma, td_abs_out=self.td_abs, want_loss=want_loss, n_step=self.n_step, n_envs=self.n_envs)
        self.ev_upd.record(cur)
        with torch.cuda.stream(self.ist):
            self.ist.wait_event(self.ev_upd)
            self.index.per_update_sorted(idx, self.td_abs)
        self.updates += 1
        if self.updates % self.replace_frequency == 0:
            self.cnn.sync_target()                                          # q_agent.py:192-193
        if want_loss:
            self.losses.append(loss)
        return loss

    def training(self, n_updates, warmup_steps=None, want_loss=False):
        """q_agent.py:174-187: act every step, one update per train_frequency steps once the ring holds a batch"""
        warm = warmup_steps if warmup_steps is not None else (self.B + self.n_envs - 1) // self.n_envs + self.n_step - 1
        while self.env_steps < warm:
            self.env_step()
        for _ in range(n_updates):
            for _ in range(self.train_frequency):
                self.env_step()
            self.update(want_loss)
        torch.cuda.current_stream(self.cnn.device).wait_stream(self.ist)     # (the last write-back: callers read the index after this)
        return self.losses
