"""Test-side access to the CPU oracle: the plain-C restatement (oracle/_build/liboracle.so, via
ctypes) and the numpy one (oracle/oracle_np.py). Checker only; never imported by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

import oracle_np as onp  # noqa: F401  (re-exported)

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "liboracle.so")


def build():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


class Dims(C.Structure):
    _fields_ = [("D", C.c_int32), ("H1", C.c_int32), ("H2", C.c_int32), ("A", C.c_int32)]


class Opt(C.Structure):
    _fields_ = [("lr", C.c_float), ("b1", C.c_float), ("b2", C.c_float), ("eps", C.c_float),
                ("wd", C.c_float), ("adamw", C.c_int32)]


class Replay(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("obs_dim", C.c_int32), ("states", C.c_void_p),
                ("actions", C.c_void_p), ("rewards", C.c_void_p), ("observations", C.c_void_p),
                ("dones", C.c_void_p), ("counter", C.c_int64), ("size", C.c_int64)]


class Per(C.Structure):
    _fields_ = [("L", C.c_int32), ("N", C.c_int64), ("tree", C.c_void_p), ("pmax", C.c_float),
                ("alpha", C.c_float), ("eps", C.c_float)]


class Learner(C.Structure):
    _fields_ = [("m", Dims), ("opt", Opt), ("gamma", C.c_float), ("beta", C.c_float),
                ("rb", C.POINTER(Replay)), ("per", C.POINTER(Per)),
                ("P", C.POINTER(C.c_float)), ("Pt", C.POINTER(C.c_float)), ("mu", C.POINTER(C.c_float)),
                ("nu", C.POINTER(C.c_float)), ("count", C.c_int32), ("b1pow", C.c_double),
                ("b2pow", C.c_double), ("seed", C.c_uint64), ("ctr", C.c_uint64), ("maxB", C.c_int32),
                ("idx", C.POINTER(C.c_int32)), ("a", C.POINTER(C.c_int32)), ("isw", C.POINTER(C.c_float)),
                ("s", C.POINTER(C.c_float)), ("s2", C.POINTER(C.c_float)), ("r", C.POINTER(C.c_float)),
                ("df", C.POINTER(C.c_float)), ("targets", C.POINTER(C.c_float)),
                ("delta", C.POINTER(C.c_float)), ("grad", C.POINTER(C.c_float)), ("d", C.POINTER(C.c_uint8)),
                ("n_step", C.c_int32), ("hist_n", C.c_int32), ("hist_steps", C.c_uint64), ("gamma_n", C.c_float),
                ("hist_s", C.POINTER(C.c_float)), ("hist_r", C.POINTER(C.c_float)), ("hist_a", C.POINTER(C.c_int32)),
                ("hist_d", C.POINTER(C.c_uint8))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_u01.restype = C.c_float
        _lib.orc_pow_det.restype = C.c_float
        _lib.orc_pow_det.argtypes = [C.c_float, C.c_float]
        _lib.orc_param_count.restype = C.c_int64
        _lib.orc_param_count.argtypes = [Dims]
        _lib.orc_loss.restype = C.c_float
        _lib.orc_learner_update.restype = C.c_float
        _lib.orc_learner_update_omp.restype = C.c_float
        _lib.orc_omp_threads.restype = C.c_int32
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def forward(dims, P, x):
    x = f32(x).reshape(-1, dims[0]); B = x.shape[0]
    q = np.empty((B, dims[3]), np.float32); h1 = np.empty((B, dims[1]), np.float32); h2 = np.empty((B, dims[2]), np.float32)
    lib().orc_forward(Dims(*dims), _p(f32(P)), _p(x), C.c_int32(B), _p(q), _p(h1), _p(h2))
    return q, h1, h2


def q_targets(dims, P, Pt, s, a, r, s2, d, gamma):
    s = f32(s); B = s.shape[0]; A = dims[3]
    out = dict(targets=np.empty((B, A), np.float32), q=np.empty((B, A), np.float32), nq=np.empty((B, A), np.float32),
               nt=np.empty((B, A), np.float32), astar=np.empty(B, np.int32), delta=np.empty(B, np.float32))
    lib().orc_q_targets(Dims(*dims), _p(f32(P)), _p(f32(Pt)), _p(s), _p(i32(a)), _p(f32(r)), _p(f32(s2)), _p(f32(d)),
                        C.c_float(gamma), C.c_int32(B), _p(out["targets"]), _p(out["q"]), _p(out["nq"]), _p(out["nt"]),
                        _p(out["astar"]), _p(out["delta"]))
    return out


def grads(dims, P, s, targets, isw=None):
    s = f32(s); B = s.shape[0]
    n = lib().orc_param_count(Dims(*dims))
    g = np.empty(n, np.float32); loss = C.c_float(); dq = np.empty((B, dims[3]), np.float32)
    lib().orc_grads(Dims(*dims), _p(f32(P)), _p(s), _p(f32(targets)), _p(None if isw is None else f32(isw)),
                    C.c_int32(B), _p(g), C.byref(loss), _p(dq))
    return g, loss.value, dq


def adam_step(opt, P, g, mu, nu, count, b1pow, b2pow, grad_scale=1.0):
    P, mu, nu = f32(P).copy(), f32(mu).copy(), f32(nu).copy()
    c = C.c_int32(count); p1 = C.c_double(b1pow); p2 = C.c_double(b2pow)
    lib().orc_adam_step(opt, _p(P), _p(f32(g)), _p(mu), _p(nu), C.byref(c), C.byref(p1), C.byref(p2),
                        C.c_int64(P.size), C.c_float(grad_scale))
    return P, mu, nu, c.value, p1.value, p2.value


def act(dims, P, s, epsilon, seed, ctr):
    s = f32(s).reshape(-1, dims[0]); n = s.shape[0]
    a = np.empty(n, np.int32)
    lib().orc_act(Dims(*dims), _p(f32(P)), _p(s), C.c_int32(n), C.c_float(epsilon), C.c_uint64(seed), C.c_uint64(ctr), _p(a))
    return a


class CReplay:
    def __init__(self, capacity, D):
        self.rb = Replay(); self.D = D
        assert lib().orc_replay_init(C.byref(self.rb), C.c_int64(capacity), C.c_int32(D)) == 0

    def add(self, s, a, r, s2, d):
        s = f32(s).reshape(-1, self.D); n = s.shape[0]
        slots = np.empty(n, np.int32)
        lib().orc_replay_add(C.byref(self.rb), _p(s), _p(i32(a)), _p(f32(r)), _p(f32(s2)), _p(u8(d)), C.c_int64(n), _p(slots))
        return slots

    def arrays(self):
        N, D = self.rb.capacity, self.D
        def view(ptr, n, t):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(t)), shape=(n,))
        return (view(self.rb.states, N * D, C.c_float).reshape(N, D), view(self.rb.actions, N, C.c_int32),
                view(self.rb.rewards, N, C.c_float), view(self.rb.observations, N * D, C.c_float).reshape(N, D),
                view(self.rb.dones, N, C.c_uint8))

    def gather(self, idx):
        idx = i32(idx); B = idx.size
        s = np.empty((B, self.D), np.float32); s2 = np.empty((B, self.D), np.float32)
        a = np.empty(B, np.int32); r = np.empty(B, np.float32); d = np.empty(B, np.uint8)
        lib().orc_replay_gather(C.byref(self.rb), _p(idx), C.c_int32(B), _p(s), _p(a), _p(r), _p(s2), _p(d))
        return s, a, r, s2, d

    @property
    def size(self):
        return self.rb.size


def uniform_indices(size, B, seed, ctr):
    idx = np.empty(B, np.int32)
    lib().orc_uniform_indices(C.c_int64(size), C.c_int32(B), C.c_uint64(seed), C.c_uint64(ctr), _p(idx))
    return idx


class CPer:
    def __init__(self, L, alpha=0.6, eps=1e-6):
        self.t = Per()
        assert lib().orc_per_init(C.byref(self.t), C.c_int32(L), C.c_float(alpha), C.c_float(eps)) == 0

    def add(self, slots):
        slots = i32(slots)
        lib().orc_per_add(C.byref(self.t), _p(slots), C.c_int64(slots.size))

    def sample(self, size, B, beta, seed, ctr):
        idx = np.empty(B, np.int32); isw = np.empty(B, np.float32)
        lib().orc_per_sample(C.byref(self.t), C.c_int64(size), C.c_int32(B), C.c_float(beta), C.c_uint64(seed),
                             C.c_uint64(ctr), _p(idx), _p(isw))
        return idx, isw

    def update(self, idx, td_abs):
        idx = i32(idx)
        lib().orc_per_update(C.byref(self.t), _p(idx), _p(f32(td_abs)), C.c_int32(idx.size))

    def set(self, idx, prio):
        idx = i32(idx)
        lib().orc_per_set(C.byref(self.t), _p(idx), _p(f32(prio)), C.c_int32(idx.size))

    @property
    def tree(self):
        return np.ctypeslib.as_array(C.cast(self.t.tree, C.POINTER(C.c_float)), shape=(2 * self.t.N,))

    @property
    def pmax(self):
        return self.t.pmax


class CLearner:
    """oracle whole-update driver (orc_learner_update)"""

    def __init__(self, dims, opt, gamma, maxB, replay: CReplay, per: CPer, P0, seed, beta=0.4):
        self.l = Learner(); self.replay, self.per = replay, per
        self.n = lib().orc_param_count(Dims(*dims))
        lib().orc_learner_init(C.byref(self.l), Dims(*dims), opt, C.c_float(gamma), C.c_int32(maxB),
                               C.byref(replay.rb), C.byref(per.t) if per is not None else None, _p(f32(P0)),
                               C.c_uint64(seed))
        self.l.beta = beta

    def update(self, B):
        return lib().orc_learner_update(C.byref(self.l), C.c_int32(B))

    def update_omp(self, B):
        """all-core form (oracle/dqn_oracle_omp.c): same bits as update()"""
        return lib().orc_learner_update_omp(C.byref(self.l), C.c_int32(B))

    def actor_step_omp(self, obs, epsilon, p_done, env_ctr):
        c = C.c_uint64(env_ctr)
        lib().orc_learner_actor_step_omp(C.byref(self.l), _p(obs), C.c_int32(obs.shape[0]), C.c_float(epsilon),
                                         C.c_float(p_done), C.byref(c))
        return c.value

    def _arr(self, p, n, t=C.c_float):
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    @property
    def params(self):
        return self._arr(self.l.P, self.n)

    @property
    def mu(self):
        return self._arr(self.l.mu, self.n)

    @property
    def nu(self):
        return self._arr(self.l.nu, self.n)

    def sync_target(self):
        C.memmove(self.l.Pt, self.l.P, self.n * 4)

    def set_nstep(self, n_step, n_envs):
        """n-step returns for the vector actor (orc_learner_set_nstep); resets the history"""
        lib().orc_learner_set_nstep(C.byref(self.l), C.c_int32(n_step), C.c_int32(n_envs))

    def actor_step_tf(self, obs, t, epsilon, p_done, max_steps, env_ctr):
        """vector step with ObsWrapper's time-fraction feature as the last column; obs and t advance in place"""
        c = C.c_uint64(env_ctr)
        lib().orc_learner_actor_step_tf(C.byref(self.l), _p(obs), _p(t), C.c_int32(obs.shape[0]), C.c_float(epsilon),
                                        C.c_float(p_done), C.c_int32(max_steps), C.byref(c))
        return c.value

    def actor_step(self, obs, epsilon, p_done, env_ctr):
        """advances obs in place; returns the new env counter"""
        c = C.c_uint64(env_ctr)
        lib().orc_learner_actor_step(C.byref(self.l), _p(obs), C.c_int32(obs.shape[0]), C.c_float(epsilon),
                                     C.c_float(p_done), C.byref(c))
        return c.value


def cnn_forward(P, frames, A):
    """C restatement of the Nature-CNN dueling forward (oracle/dqn_oracle_cnn.c): (q [B,A], fc features [B,512])"""
    frames = u8(frames); B = frames.shape[0]
    q = np.empty((B, A), np.float32); feat = np.empty((B, 512), np.float32)
    lib().orc_cnn_param_count.restype = C.c_int64
    assert lib().orc_cnn_param_count(C.c_int32(A)) == np.asarray(P).size
    lib().orc_cnn_forward(_p(f32(P)), _p(frames), C.c_int32(B), C.c_int32(A), _p(q), _p(feat))
    return q, feat


def cnn_grads(P, frames, targets, isw, A, f64=False):
    """gradient of the reference's loss (q_learning_functions.py:31-39) through the Nature-CNN, every leaf, flat:
    (grad, loss). f64=True: the double-precision form (tolerance reference)."""
    frames = u8(frames); B = frames.shape[0]
    targets = f32(targets); assert targets.shape == (B, A)
    n = np.asarray(P).size
    w = None if isw is None else f32(isw)
    if f64:
        g = np.empty(n, np.float64); loss = C.c_double(0)
        lib().orc_cnn_grads_f64(_p(f32(P)), _p(frames), _p(targets), _p(w) if w is not None else None, C.c_int32(B), C.c_int32(A), _p(g), C.byref(loss))
    else:
        g = np.empty(n, np.float32); loss = C.c_float(0)
        lib().orc_cnn_grads(_p(f32(P)), _p(frames), _p(targets), _p(w) if w is not None else None, C.c_int32(B), C.c_int32(A), _p(g), C.byref(loss))
    return g, loss.value


def synth_env(n, D, seed, env_ctr, p_done):
    obs = np.empty((n, D), np.float32); r = np.empty(n, np.float32); d = np.empty(n, np.uint8)
    lib().orc_synth_env(C.c_int32(n), C.c_int32(D), C.c_uint64(seed), C.c_uint64(env_ctr), C.c_float(p_done),
                        _p(obs), _p(r), _p(d))
    return obs, r, d
