"""Host-side owner of one libdqn_hip handle: torch supplies device memory for the caller's
tensors, the current HIP stream and torch.distributed; every computation goes through the
C ABI (include/dqn_hip.h). No CPU fallback anywhere in this module.

The reference wires its jitted closures in General/QLearning/q_agent.py:110-113 and calls
them from Agent._step (:146-169) / Agent._policy (:137-141); `Engine` is what those closures
are backed by in this build.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib as L


@dataclass
class EngineConfig:
    """Mirrors the constants of Test/lunar_lander.py:23-48 + the net of LunarLander/dddqn.py:19-22."""
    obs_dim: int = 9
    hidden1: int = 32
    hidden2: int = 64
    num_actions: int = 4
    capacity: int = 100_000
    use_per: bool = False
    max_batch: int = 64
    optimizer: str = "adamw"          # optax.adamw (Test/lunar_lander.py:48) / "adam" (hyper-param script :41)
    lr: float = 2e-4
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 1e-4
    gamma: float = 0.99
    per_alpha: float = 0.6
    per_eps: float = 1e-6
    per_beta: float = 0.4
    precision: str = "f32"
    seed: int = 0
    world_size: int = 1
    n_step: int = 1                   # n-step returns of the device-resident vector actor (not in the reference)
    flags: int = 0                    # dqn_flags (diagnostics): _lib.FLAG_NO_HANDOVER | FLAG_NO_ACTOR16 | FLAG_BF16_F32_ACTOR

    @property
    def dims(self):
        return (self.obs_dim, self.hidden1, self.hidden2, self.num_actions)


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("deep_q_learning_amd needs an MI355X (gfx950) GPU; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class _DevView:
    """__cuda_array_interface__ holder so torch can alias handle-owned device memory."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}
        self._owner = owner


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    def __init__(self, cfg: EngineConfig, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("deep_q_learning_amd needs an MI355X (gfx950) GPU: torch.cuda is not available "
                               "and there is no CPU fallback")
        self.cfg = cfg
        self.lib = L.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        c = L.DqnConfig()
        self.lib.dqn_default_config(C.byref(c))
        c.obs_dim, c.hidden1, c.hidden2, c.num_actions = cfg.obs_dim, cfg.hidden1, cfg.hidden2, cfg.num_actions
        c.capacity, c.use_per, c.max_batch = cfg.capacity, int(cfg.use_per), cfg.max_batch
        c.optimizer = {"adam": L.OPT_ADAM, "adamw": L.OPT_ADAMW}[cfg.optimizer]
        c.lr, c.b1, c.b2, c.eps, c.weight_decay = cfg.lr, cfg.b1, cfg.b2, cfg.eps, cfg.weight_decay
        c.gamma, c.per_alpha, c.per_eps, c.per_beta = cfg.gamma, cfg.per_alpha, cfg.per_eps, cfg.per_beta
        c.precision = {"f32": L.PREC_F32, "bf16": L.PREC_BF16}[cfg.precision]
        c.seed, c.world_size, c.n_step = cfg.seed, cfg.world_size, cfg.n_step
        c.flags = int(cfg.flags)
        h = C.c_void_p()
        L.check(self.lib.dqn_create(C.byref(c), C.byref(h)))
        self.h = h
        n = C.c_int64()
        L.check(self.lib.dqn_param_count(self.h, C.byref(n)))
        self.param_count = n.value
        self.stream = torch.cuda.Stream(device=self.device)    # non-default: graph capture needs one

    def close(self):
        if getattr(self, "h", None):
            self.lib.dqn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _s(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        return C.c_void_p(s.cuda_stream)

    def dev(self, x, dtype):
        """tensor on this device, contiguous, of `dtype` (numpy / python inputs are uploaded)."""
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x))
        if x.dtype != dtype:
            x = x.to(dtype)
        if x.device != self.device:
            x = x.to(self.device)
        return x.contiguous()

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def buffer(self, which, dtype=torch.float32, shape=None):
        """torch view (no copy) of a handle-owned device buffer."""
        p, n = C.c_void_p(), C.c_int64()
        L.check(self.lib.dqn_buffer(self.h, which, C.byref(p), C.byref(n)))
        item = torch.empty((), dtype=dtype).element_size()
        numel = n.value // item
        typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
        t = torch.as_tensor(_DevView(p.value, (numel,), typestr, self), device=self.device)
        return t.view(shape) if shape is not None else t

    # ------------------------------------------------------------- params / state
    def load(self, params, which=L.BUF_PARAMS):
        """upload a haiku-style tree (or flat tensor) unless it is the very tensor uploaded last"""
        from ._tree import flatten
        flat = params if isinstance(params, torch.Tensor) else flatten(params)
        tag = (flat.data_ptr(), flat._version, which) if isinstance(flat, torch.Tensor) else None
        cache = self.__dict__.setdefault("_loaded", {})
        if cache.get(which) == tag and tag is not None:
            return
        self.set_params(flat, which)
        cache[which] = tag

    def set_params(self, flat, which=L.BUF_PARAMS):
        if isinstance(flat, torch.Tensor) and flat.is_cuda:
            flat = self.dev(flat, torch.float32)
            L.check(self.lib.dqn_set_params(self.h, which, _ptr(flat), 0, self._s()))
        else:
            a = np.ascontiguousarray(np.asarray(flat.cpu() if isinstance(flat, torch.Tensor) else flat, np.float32))
            assert a.size == self.param_count, (a.size, self.param_count)
            L.check(self.lib.dqn_set_params(self.h, which, a.ctypes.data_as(C.c_void_p), 1, self._s()))

    def get_params(self, which=L.BUF_PARAMS, host=False):
        if host:
            a = np.empty(self.param_count, np.float32)
            L.check(self.lib.dqn_get_params(self.h, which, a.ctypes.data_as(C.c_void_p), 1, self._s()))
            return a
        t = self.empty((self.param_count,), torch.float32)
        L.check(self.lib.dqn_get_params(self.h, which, _ptr(t), 0, self._s()))
        return t

    def set_opt_count(self, count: int):
        L.check(self.lib.dqn_set_opt_count(self.h, int(count), self._s()))

    def opt_count(self) -> int:
        c = C.c_int32()
        L.check(self.lib.dqn_get_opt_count_host(self.h, C.byref(c)))
        return c.value

    def set_schedule(self, per_beta=None, lr=None):
        if per_beta is not None:
            self.cfg.per_beta = per_beta
        if lr is not None:
            self.cfg.lr = lr
        L.check(self.lib.dqn_set_schedule(self.h, self.cfg.per_beta, self.cfg.lr, self._s()))

    def set_gamma(self, gamma):
        """discount of the TD rule (q_learning_functions.py:58) for the handle's own update; captured graphs are rebuilt"""
        self.cfg.gamma = float(gamma)
        L.check(self.lib.dqn_set_gamma(self.h, float(gamma)))

    def sync_target(self):
        L.check(self.lib.dqn_sync_target(self.h, self._s()))

    # -------------------------------------------------------------------- replay
    def replay_add(self, s, a, r, s2, d):
        s = self.dev(s, torch.float32).view(-1, self.cfg.obs_dim)
        n = s.shape[0]
        s2 = self.dev(s2, torch.float32).view(n, self.cfg.obs_dim)
        a = self.dev(a, torch.int32).view(n)
        r = self.dev(r, torch.float32).view(n)
        d = self.dev(d, torch.uint8 if not (isinstance(d, torch.Tensor) and d.dtype == torch.bool) else torch.bool)
        d = d.to(torch.uint8).view(n)
        L.check(self.lib.dqn_replay_add(self.h, _ptr(s), _ptr(a), _ptr(r), _ptr(s2), _ptr(d), n, self._s()))

    def per_index_advance(self, n):
        """the handle as a positions-only prioritized index: n new positions at the running max priority (no row data)"""
        L.check(self.lib.dqn_per_index_advance(self.h, int(n), self._s()))

    def per_index_step(self, n, zero_first=0, zero_n=0):
        """per_index_advance(n) preceded, in the same launch, by priority 0 for the zero_n positions from zero_first on (rows being overwritten)"""
        L.check(self.lib.dqn_per_index_step(self.h, int(n), int(zero_first), int(zero_n), self._s()))

    def replay_size(self):
        size, ctr = C.c_int64(), C.c_int64()
        L.check(self.lib.dqn_replay_size_host(self.h, C.byref(size), C.byref(ctr)))
        return size.value, ctr.value

    def _batch_out(self, B):
        D = self.cfg.obs_dim
        return (self.empty((B, D), torch.float32), self.empty((B,), torch.int32), self.empty((B,), torch.float32),
                self.empty((B, D), torch.float32), self.empty((B,), torch.uint8))

    def sample_uniform(self, B, seed=0, ctr=0, idx=None):
        s, a, r, s2, d = self._batch_out(B)
        idx_in = None if idx is None else self.dev(idx, torch.int32)
        idx_out = self.empty((B,), torch.int32)
        L.check(self.lib.dqn_replay_sample_uniform(self.h, B, seed, ctr, _ptr(idx_in), _ptr(s), _ptr(a), _ptr(r),
                                                   _ptr(s2), _ptr(d), _ptr(idx_out), self._s()))
        return (s, a, r, s2, d), idx_out

    def per_sample(self, B, beta, seed=0, ctr=0):
        s, a, r, s2, d = self._batch_out(B)
        idx, isw = self.empty((B,), torch.int32), self.empty((B,), torch.float32)
        L.check(self.lib.dqn_per_sample(self.h, B, float(beta), seed, ctr, _ptr(s), _ptr(a), _ptr(r), _ptr(s2),
                                        _ptr(d), _ptr(idx), _ptr(isw), self._s()))
        return (s, a, r, s2, d), idx, isw

    def per_sample_into(self, B, beta, seed, ctr, bufs):
        """per_sample into caller-held buffers (s, a, r, s2, d, idx, isw): no allocation between launches"""
        s, a, r, s2, d, idx, isw = bufs
        L.check(self.lib.dqn_per_sample(self.h, B, float(beta), seed, ctr, _ptr(s), _ptr(a), _ptr(r), _ptr(s2),
                                        _ptr(d), _ptr(idx), _ptr(isw), self._s()))

    def per_update(self, idx, td_abs):
        idx, td_abs = self.dev(idx, torch.int32), self.dev(td_abs, torch.float32)
        L.check(self.lib.dqn_per_update(self.h, _ptr(idx), _ptr(td_abs), idx.numel(), self._s()))

    def per_update_sorted(self, idx, td_abs):
        """as per_update, for non-decreasing idx (the output of per_sample); runs on many CUs"""
        idx, td_abs = self.dev(idx, torch.int32), self.dev(td_abs, torch.float32)
        L.check(self.lib.dqn_per_update_sorted(self.h, _ptr(idx), _ptr(td_abs), idx.numel(), self._s()))

    def per_set_sorted(self, idx, prio):
        """as per_set, for non-decreasing idx"""
        idx, prio = self.dev(idx, torch.int32), self.dev(prio, torch.float32)
        L.check(self.lib.dqn_per_set_sorted(self.h, _ptr(idx), _ptr(prio), idx.numel(), self._s()))

    def per_set(self, idx, prio):
        idx, prio = self.dev(idx, torch.int32), self.dev(prio, torch.float32)
        L.check(self.lib.dqn_per_set(self.h, _ptr(idx), _ptr(prio), idx.numel(), self._s()))

    # ------------------------------------------------------------------- network
    def forward(self, x, target=False, return_features=False):
        x = self.dev(x, torch.float32).view(-1, self.cfg.obs_dim)
        B = x.shape[0]
        q = self.empty((B, self.cfg.num_actions), torch.float32)
        feat = self.empty((B, self.cfg.hidden2), torch.float32) if return_features else None
        L.check(self.lib.dqn_qnet_forward(self.h, L.NET_TARGET if target else L.NET_ONLINE, _ptr(x), B, _ptr(q),
                                          _ptr(feat), self._s()))
        return (q, feat) if return_features else q

    def td_targets(self, q, nq, nt, a, r, d, isw=None, gamma=None, want=("targets", "td", "dq", "loss")):
        q, nq, nt = (self.dev(t, torch.float32) for t in (q, nq, nt))
        B, A = q.shape
        a, r, d = self.dev(a, torch.int32), self.dev(r, torch.float32), self.dev(d, torch.float32)
        isw = None if isw is None else self.dev(isw, torch.float32)
        out = {"targets": self.empty((B, A), torch.float32) if "targets" in want else None,
               "td": self.empty((B,), torch.float32) if "td" in want else None,
               "dq": self.empty((B, A), torch.float32) if "dq" in want else None,
               "loss": self.empty((1,), torch.float32) if "loss" in want else None}
        L.check(self.lib.dqn_td_targets(self.h, _ptr(q), _ptr(nq), _ptr(nt), _ptr(a), _ptr(r), _ptr(d), _ptr(isw),
                                        float(self.cfg.gamma if gamma is None else gamma), B, _ptr(out["targets"]),
                                        _ptr(out["td"]), _ptr(out["dq"]), _ptr(out["loss"]), self._s()))
        return out

    def q_targets(self, s, a, r, s2, d):
        s = self.dev(s, torch.float32).view(-1, self.cfg.obs_dim)
        B = s.shape[0]
        s2 = self.dev(s2, torch.float32).view(B, self.cfg.obs_dim)
        a, r, d = self.dev(a, torch.int32), self.dev(r, torch.float32), self.dev(d, torch.float32)
        t = self.empty((B, self.cfg.num_actions), torch.float32)
        L.check(self.lib.dqn_q_targets(self.h, _ptr(s), _ptr(a), _ptr(r), _ptr(s2), _ptr(d), B, _ptr(t), self._s()))
        return t

    def loss(self, s, targets, isw=None):
        s = self.dev(s, torch.float32).view(-1, self.cfg.obs_dim)
        targets = self.dev(targets, torch.float32)
        isw = None if isw is None else self.dev(isw, torch.float32)
        out = self.empty((1,), torch.float32)
        L.check(self.lib.dqn_loss(self.h, _ptr(s), _ptr(targets), _ptr(isw), s.shape[0], _ptr(out), self._s()))
        return out

    def grads(self, s, targets, isw=None):
        """fills the handle's gradient buffer; returns (grad view, loss tensor)"""
        s = self.dev(s, torch.float32).view(-1, self.cfg.obs_dim)
        targets = self.dev(targets, torch.float32)
        isw = None if isw is None else self.dev(isw, torch.float32)
        out = self.empty((1,), torch.float32)
        L.check(self.lib.dqn_grads(self.h, _ptr(s), _ptr(targets), _ptr(isw), s.shape[0], _ptr(out), self._s()))
        return self.get_params(L.BUF_GRAD), out

    def optimizer_step(self):
        L.check(self.lib.dqn_optimizer_step(self.h, self._s()))

    def train_step(self, s, targets):
        s = self.dev(s, torch.float32).view(-1, self.cfg.obs_dim)
        targets = self.dev(targets, torch.float32)
        L.check(self.lib.dqn_train_step(self.h, _ptr(s), _ptr(targets), s.shape[0], self._s()))

    def act(self, s, epsilon, seed=0, ctr=0):
        s = self.dev(s, torch.float32).view(-1, self.cfg.obs_dim)
        n = s.shape[0]
        a = self.empty((n,), torch.int32)
        L.check(self.lib.dqn_act(self.h, _ptr(s), n, float(epsilon), seed, ctr, _ptr(a), self._s()))
        return a

    # -------------------------------------------------------------- fused update
    def update(self, B, stream=None):
        """Agent._step (q_agent.py:146-169) on the handle's own replay, graph-replayed."""
        L.check(self.lib.dqn_update_fused(self.h, B, self._s(stream)))

    def update_backward(self, B, stream=None):
        L.check(self.lib.dqn_update_backward(self.h, B, self._s(stream)))

    def update_apply(self, B, stream=None):
        L.check(self.lib.dqn_update_apply(self.h, B, self._s(stream)))

    # ------------------------------------------------------------ synthetic actor
    def set_epsilon(self, epsilon):
        L.check(self.lib.dqn_set_epsilon(self.h, float(epsilon), self._s()))

    def env_reset(self, obs, p_done=0.01):
        obs = self.dev(obs, torch.float32).view(-1, self.cfg.obs_dim)
        L.check(self.lib.dqn_env_reset(self.h, _ptr(obs), obs.shape[0], float(p_done), self._s()))
        self.n_envs = obs.shape[0]

    def env_config(self, kind="synthetic", max_steps=500, term_reward=1.0):
        L.check(self.lib.dqn_env_config(self.h, {"synthetic": L.ENV_SYNTHETIC, "cartpole": L.ENV_CARTPOLE}[kind],
                                        int(max_steps), float(term_reward)))

    def env_time_feature(self, enable=True):
        """ObsWrapper's step / max_steps feature (LunarLander/env.py:19-24) as the last observation column of the vector envs"""
        L.check(self.lib.dqn_env_time_feature(self.h, int(bool(enable))))

    def env_stats(self):
        """(finished episodes, summed episode length) of the device-resident envs; synchronises"""
        ep, st = C.c_int64(), C.c_int64()
        L.check(self.lib.dqn_env_stats_host(self.h, C.byref(ep), C.byref(st)))
        return ep.value, st.value

    def actor_step(self, stream=None):
        """one vector env step (q_agent.py:176-183) on the device-resident synthetic envs"""
        L.check(self.lib.dqn_actor_step(self.h, self.n_envs, self._s(stream)))

    def actor_steps(self, env_steps, stream=None):
        """env_steps consecutive vector env steps in one launch (the train_frequency actor steps between two updates)"""
        L.check(self.lib.dqn_actor_steps(self.h, int(env_steps), self.n_envs, self._s(stream)))

    def train_iters(self, n_iters, env_steps, B, stream=None):
        """n_iters x (env_steps vector env steps + one update) in one graph launch (q_agent.py:174-187)"""
        L.check(self.lib.dqn_train_iters(self.h, n_iters, env_steps, getattr(self, "n_envs", 0), B, self._s(stream)))

    def actor_backward(self, env_steps, B, stream=None):
        """data-parallel iteration, first half: env_steps vector env steps + sample..grads (+ PER write-back) in one
        graph launch; then all-reduce DQN_BUF_GRAD and call update_apply"""
        L.check(self.lib.dqn_actor_backward(self.h, env_steps, getattr(self, "n_envs", 0), B, self._s(stream)))

    # ------------------------------------------------------------- native RCCL
    def comm_init_native(self):
        """create the handle's own RCCL communicator (dqn_comm_init): rank 0 makes the unique id, torch.distributed
        carries it to the other ranks. Afterwards train_iters captures the gradient all-reduce inside its graph."""
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
        L.check(self.lib.dqn_comm_init(self.h, uid, rank, world))

    def comm_ranks(self) -> int:
        """ranks of the handle's own RCCL communicator (ncclCommCount); 0 without one"""
        n = C.c_int32()
        L.check(self.lib.dqn_comm_count_host(self.h, C.byref(n)))
        return n.value

    def device_errors(self) -> int:
        """in-kernel hand-over waits that gave up since the handle was created (synchronises); must be 0"""
        n = C.c_int64()
        L.check(self.lib.dqn_device_errors_host(self.h, C.byref(n)))
        return n.value

    def clear_device_errors(self):
        """after device_errors() != 0: hand-over words and the error count back to their initial state (synchronises)"""
        L.check(self.lib.dqn_clear_device_errors(self.h))

    def debug_withhold_handover(self, on: bool):
        """diagnostic (tests): make the fused update's hand-over waits run into their 0.2 s bound"""
        L.check(self.lib.dqn_debug_withhold_handover(self.h, int(bool(on))))

    def allreduce_grads_native(self, stream=None):
        L.check(self.lib.dqn_allreduce_grads(self.h, self._s(stream)))

    # ------------------------------------------------------------------ profiling
    def profile_begin(self, stream=None):
        L.check(self.lib.dqn_profile_begin(self.h, self._s(stream)))

    def profile_end(self, stream=None, max_entries=256):
        """-> [(kernel name, elapsed ms)] for every launch since profile_begin (HIP events)"""
        names = C.create_string_buffer(32 * max_entries)
        ms = (C.c_float * max_entries)()
        n = C.c_int32()
        L.check(self.lib.dqn_profile_end(self.h, self._s(stream), names, 32, ms, max_entries, C.byref(n)))
        return [(names.raw[32 * i:32 * i + 32].split(b"\0", 1)[0].decode(), ms[i]) for i in range(n.value)]

    def last_loss(self):
        return self.buffer(L.BUF_LOSS)[:1]
