"""Nature-CNN dueling Q-network: forward, loss gradient, Adam step (BASELINE.json configs[4], PongNoFrameskip-v4 shape).
The reference has no CNN: this is the model a `Model`-like object would wrap for Atari, ending in the reference's dueling
head (LunarLander/dddqn.py:29-31) and feeding the reference's TD rule (General/QLearning/q_learning_functions.py:55-60).
Every computation goes through the C ABI (include/dqn_hip.h: dqn_cnn_*); there is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .engine import _DevView, _ptr


class CnnEngine:
    def __init__(self, num_actions: int = 6, max_batch: int = 512, precision: str = "bf16", device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("deep_q_learning_amd needs an MI355X (gfx950) GPU: torch.cuda is not available "
                               "and there is no CPU fallback")
        self.lib = L.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        self.num_actions, self.max_batch, self.precision = num_actions, max_batch, precision
        h = C.c_void_p()
        L.check(self.lib.dqn_cnn_create(num_actions, max_batch, {"f32": L.PREC_F32, "bf16": L.PREC_BF16}[precision], C.byref(h)))
        self.h = h
        n = C.c_int64()
        L.check(self.lib.dqn_cnn_param_count(self.h, C.byref(n)))
        self.param_count = n.value

    def set_flags(self, flags: int):
        """diagnostics (tests): _lib.CNN_FLAG_FC_WIDE_TILE | CNN_FLAG_NO_SIDE_STREAM | CNN_FLAG_LAYERWISE_CONV; results stay bit-identical"""
        L.check(self.lib.dqn_cnn_set_flags(self.h, int(flags)))

    def comm_init_native(self):
        """the handle's own RCCL communicator (dqn_cnn_comm_init), as Engine.comm_init_native: rank 0 makes the unique id,
        torch.distributed carries it to the other ranks. Afterwards update() / update_from_replay() all-reduce the gradient."""
        import torch.distributed as dist
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        uid = (C.c_char * 128)()
        if rank == 0:
            L.check(self.lib.dqn_comm_unique_id(uid))
        if world > 1:
            dev = self.device if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor(list(uid.raw), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=0)
            uid = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().tolist()))
        L.check(self.lib.dqn_cnn_comm_init(self.h, uid, rank, world))

    def comm_ranks(self) -> int:
        n = C.c_int32()
        L.check(self.lib.dqn_cnn_comm_count_host(self.h, C.byref(n)))
        return n.value

    def allreduce_grads(self, fc_leaf_done=False):
        """in-place SUM all-reduce of the "grad" buffer over the handle's communicator (between grads() and
        optimizer_step(grad_scale=1 / world))"""
        L.check(self.lib.dqn_cnn_allreduce_grads(self.h, int(bool(fc_leaf_done)), self._s()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.dqn_cnn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _s(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _frames(self, x):
        x = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
        x = x.to(self.device, torch.uint8).contiguous()
        assert x.shape[1:] == (84, 84, 4), x.shape
        return x

    def set_params(self, flat, target=False):
        a = np.ascontiguousarray(np.asarray(flat.cpu() if isinstance(flat, torch.Tensor) else flat, np.float32))
        assert a.size == self.param_count, (a.size, self.param_count)
        L.check(self.lib.dqn_cnn_set_params(self.h, L.NET_TARGET if target else L.NET_ONLINE, a.ctypes.data_as(C.c_void_p), 1, self._s()))

    def forward(self, frames, target=False, out=None):
        """frames: u8 [B, 84, 84, 4] (four stacked frames, NHWC) -> Q [B, A]"""
        x = self._frames(frames)
        q = out if out is not None else torch.empty((x.shape[0], self.num_actions), dtype=torch.float32, device=self.device)
        L.check(self.lib.dqn_cnn_forward(self.h, L.NET_TARGET if target else L.NET_ONLINE, _ptr(x), x.shape[0], _ptr(q), self._s()))
        return q

    def q_targets(self, s, a, r, s2, d, gamma=0.99):
        """compute_q_targets (q_learning_functions.py:42-64) with the CNN as the model"""
        s, s2 = self._frames(s), self._frames(s2)
        B = s.shape[0]
        t = lambda v, dt: (v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))).to(self.device, dt).contiguous()
        a, r, d = t(a, torch.int32), t(r, torch.float32), t(d, torch.float32)
        out = torch.empty((B, self.num_actions), dtype=torch.float32, device=self.device)
        L.check(self.lib.dqn_cnn_q_targets(self.h, _ptr(s), _ptr(a), _ptr(r), _ptr(s2), _ptr(d), float(gamma), B, _ptr(out), self._s()))
        return out

    # ---- training (train_step / Agent._step on a given minibatch)
    def _t(self, v, dt):
        return (v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))).to(self.device, dt).contiguous()

    def set_optimizer(self, lr=3e-4, b1=0.9, b2=0.999, eps=1e-8, weight_decay=1e-4, adamw=True):
        """optax.adam / adamw (Test/lunar_lander.py:48); resets the moments and the step count"""
        L.check(self.lib.dqn_cnn_set_optimizer(self.h, int(bool(adamw)), float(lr), float(b1), float(b2), float(eps), float(weight_decay), self._s()))

    def get_buffer(self, which):
        """flat f32 copy of "params" / "target" / "grad" / "mu" / "nu" """
        out = torch.empty(self.param_count, dtype=torch.float32, device=self.device)
        L.check(self.lib.dqn_cnn_get_buffer(self.h, {"params": L.BUF_PARAMS, "target": L.BUF_TARGET, "grad": L.BUF_GRAD, "mu": L.BUF_MU, "nu": L.BUF_NU}[which],
                                           _ptr(out), 0, self._s()))
        return out

    def buffer(self, which):
        """torch view (no copy) of "params" / "target" / "grad" / "mu" / "nu": per-GPU learners all-reduce the "grad" view in
        place between grads() and optimizer_step(grad_scale=1 / world)"""
        p, n = C.c_void_p(), C.c_int64()
        L.check(self.lib.dqn_cnn_buffer(self.h, {"params": L.BUF_PARAMS, "target": L.BUF_TARGET, "grad": L.BUF_GRAD, "mu": L.BUF_MU, "nu": L.BUF_NU}[which],
                                        C.byref(p), C.byref(n)))
        return torch.as_tensor(_DevView(p.value, (n.value // 4,), "<f4", self), device=self.device)

    def grads(self, frames, targets, isw=None):
        """jax.grad(compute_loss) (q_learning_functions.py:23): fills the gradient buffer, returns the loss"""
        x = self._frames(frames)
        t = self._t(targets, torch.float32); assert t.shape == (x.shape[0], self.num_actions), t.shape
        w = None if isw is None else self._t(isw, torch.float32)
        loss = C.c_float(0)
        L.check(self.lib.dqn_cnn_grads(self.h, _ptr(x), _ptr(t), _ptr(w) if w is not None else None, x.shape[0], C.byref(loss), self._s()))
        return loss.value

    def optimizer_step(self, grad_scale=1.0):
        L.check(self.lib.dqn_cnn_optimizer_step(self.h, float(grad_scale), self._s()))

    def train_step(self, frames, targets, isw=None):
        """train_step (q_learning_functions.py:14-28)"""
        x = self._frames(frames)
        t = self._t(targets, torch.float32)
        w = None if isw is None else self._t(isw, torch.float32)
        L.check(self.lib.dqn_cnn_train_step(self.h, _ptr(x), _ptr(t), _ptr(w) if w is not None else None, x.shape[0], self._s()))

    def update(self, s, a, r, s2, d, isw=None, gamma=0.99, want_loss=False):
        """Agent._step (q_agent.py:146-169) on a given minibatch: compute_q_targets + train_step"""
        s, s2 = self._frames(s), self._frames(s2)
        a, r, d = self._t(a, torch.int32), self._t(r, torch.float32), self._t(d, torch.float32)
        w = None if isw is None else self._t(isw, torch.float32)
        loss = C.c_float(0)
        L.check(self.lib.dqn_cnn_update(self.h, _ptr(s), _ptr(a), _ptr(r), _ptr(s2), _ptr(d), _ptr(w) if w is not None else None, float(gamma), s.shape[0],
                                        C.byref(loss) if want_loss else None, self._s()))
        return loss.value if want_loss else None

    def sync_target(self):
        L.check(self.lib.dqn_cnn_sync_target(self.h, self._s()))

    # ---- acting and the frame replay ring (the loop of BASELINE configs[4])
    def act(self, frames, epsilon, seed=0, ctr=0, out=None):
        """Agent._policy (q_agent.py:137-141) with the CNN as the model: epsilon-greedy actions [n] (int32)"""
        x = self._frames(frames)
        a = out if out is not None else torch.empty((x.shape[0],), dtype=torch.int32, device=self.device)
        L.check(self.lib.dqn_cnn_act(self.h, _ptr(x), x.shape[0], float(epsilon), int(seed), int(ctr), _ptr(a), self._s()))
        return a

    def env_reset_synth(self, n_envs, seed=0):
        """reset() of n synthetic frame-stack envs that live on the device (dqn_cnn_env_reset_synth; needs replay_init first)"""
        L.check(self.lib.dqn_cnn_env_reset_synth(self.h, int(n_envs), int(seed), self._s()))

    def env_step_synth(self, epsilon, p_done=0.01):
        """one vector env step on the device: CNN act + synthetic transition + ring add; returns the ring row of env 0"""
        first = C.c_int64(0)
        L.check(self.lib.dqn_cnn_env_step_synth(self.h, float(epsilon), float(p_done), C.byref(first), self._s()))
        return first.value

    def replay_init(self, capacity):
        """ReplayBuffer.__init__ (replay_buffer.py:20-34) for frame stacks"""
        L.check(self.lib.dqn_cnn_replay_init(self.h, int(capacity)))
        self.capacity = int(capacity)

    def replay_add(self, s, a, r, s2, d):
        """ReplayBuffer.add (replay_buffer.py:58-65) for n transitions; returns the ring position of the first"""
        s, s2 = self._frames(s), self._frames(s2)
        a, r, d = self._t(a, torch.int32), self._t(r, torch.float32), self._t(d, torch.float32)
        first = C.c_int64(0)
        L.check(self.lib.dqn_cnn_replay_add(self.h, _ptr(s), _ptr(a), _ptr(r), _ptr(s2), _ptr(d), s.shape[0], C.byref(first), self._s()))
        return first.value

    def replay_size(self):
        size, ctr = C.c_int64(), C.c_int64()
        L.check(self.lib.dqn_cnn_replay_size_host(self.h, C.byref(size), C.byref(ctr)))
        return size.value, ctr.value

    def replay_gather(self, idx, n_step=1, n_envs=0, gamma=0.99):
        """the rows idx of the ring: (s, a, r, s2, d); n_step > 1: the n-step transitions that start there (rows stored
        step-major, n_envs per env step)"""
        idx = self._t(idx, torch.int32); B = idx.numel()
        s = torch.empty((B, 84, 84, 4), dtype=torch.uint8, device=self.device); s2 = torch.empty_like(s)
        a = torch.empty((B,), dtype=torch.int32, device=self.device)
        r = torch.empty((B,), dtype=torch.float32, device=self.device); d = torch.empty_like(r)
        L.check(self.lib.dqn_cnn_replay_gather(self.h, _ptr(idx), B, int(n_step), int(n_envs), float(gamma), _ptr(s), _ptr(a), _ptr(r), _ptr(s2), _ptr(d), self._s()))
        return s, a, r, s2, d

    def update_from_replay(self, idx, isw=None, gamma=0.99, td_abs_out=None, want_loss=False, n_step=1, n_envs=0):
        """Agent._step (q_agent.py:146-169) on the ring rows idx (n-step transitions when n_step > 1); td_abs_out (float32 [B],
        optional) receives |delta|"""
        idx = self._t(idx, torch.int32)
        w = None if isw is None else self._t(isw, torch.float32)
        loss = C.c_float(0)
        L.check(self.lib.dqn_cnn_update_replay(self.h, _ptr(idx), _ptr(w) if w is not None else None, float(gamma), int(n_step), int(n_envs), idx.numel(),
                                               _ptr(td_abs_out) if td_abs_out is not None else None, C.byref(loss) if want_loss else None, self._s()))
        return loss.value if want_loss else None
