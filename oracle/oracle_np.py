"""oracle/oracle_np.py -- numpy restatement of the reference's DDDQN hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, tests/golden/make_golden.py,
__graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker -- never by the
product path (deep-q-learning_amd/).

PARITY UNPINNED: the reference holds no tests / golden vectors / known-answer outputs
for this path, jax/haiku/optax/numba are absent here, and its only fixture
(Test/lunar_lander/*.pickle) is refused by torch.load(weights_only=True) and is not
read. This module is a second, independent restatement (the first is the plain-C one in
dqn_oracle_*.c); tests require the two to agree (bit-exactly on integer paths) and
cross-check the FP maths against PyTorch-CPU autograd + torch.optim.

FP functions take a `dtype` (np.float64 = "truth" for tolerance tests).
Citations are file:line into /root/reference.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------- RNG
STREAM_PER, STREAM_UNIFORM, STREAM_POLICY, STREAM_ENV = 0, 1, 2, 3
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al. SC'11). ctr: (...,4) uint32, key: (...,2) uint32."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32)
    k1 = np.asarray(key[..., 1], dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c[0].astype(np.uint64)
            p1 = _M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + _W0).astype(np.uint32)
            k1 = (k1 + _W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def philox_draw(seed: int, ctr: int, n: int, stream: int):
    """(n,4) uint32: counter = (ctr_lo, ctr_hi, k, stream), key = (seed_lo, seed_hi)."""
    c = np.zeros((n, 4), dtype=np.uint32)
    c[:, 0] = ctr & 0xFFFFFFFF
    c[:, 1] = (ctr >> 32) & 0xFFFFFFFF
    c[:, 2] = np.arange(n, dtype=np.uint32)
    c[:, 3] = stream
    k = np.zeros((n, 2), dtype=np.uint32)
    k[:, 0] = seed & 0xFFFFFFFF
    k[:, 1] = (seed >> 32) & 0xFFFFFFFF
    return philox4x32_10(c, k)


def u01(x):
    return (np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


# ------------------------------------------------------- deterministic f32 pow
_f32 = np.float32


def log2_det(x):
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    e = ((u >> np.uint32(23)) & np.uint32(0xFF)).astype(np.int32) - 127
    m = ((u & np.uint32(0x007FFFFF)) | np.uint32(0x3F800000)).view(np.float32)
    big = m > _f32(1.41421354)
    m = np.where(big, m * _f32(0.5), m).astype(np.float32)
    e = e + big.astype(np.int32)
    s = ((m - _f32(1.0)) / (m + _f32(1.0))).astype(np.float32)
    z = (s * s).astype(np.float32)
    p = np.full_like(z, _f32(0.111111112))
    for c in (0.142857149, 0.2, 0.333333343, 1.0):
        p = (p * z).astype(np.float32)
        p = (p + _f32(c)).astype(np.float32)
    ln_m = ((_f32(2.0) * s).astype(np.float32) * p).astype(np.float32)
    r = (ln_m * _f32(1.44269502)).astype(np.float32)
    return (e.astype(np.float32) + r).astype(np.float32)


def exp2_det(y):
    y = np.asarray(y, dtype=np.float32)
    fi = np.floor((y + _f32(0.5)).astype(np.float32)).astype(np.float32)
    i = fi.astype(np.int32)
    f = (y - fi).astype(np.float32)
    t = (f * _f32(0.693147182)).astype(np.float32)
    p = np.full_like(t, _f32(1.98412701e-4))
    for c in (1.38888892e-3, 8.33333377e-3, 4.16666679e-2, 0.166666672, 0.5, 1.0, 1.0):
        p = (p * t).astype(np.float32)
        p = (p + _f32(c)).astype(np.float32)
    i = np.clip(i, -126, 127)
    scale = ((i + 127).astype(np.uint32) << np.uint32(23)).view(np.float32)
    return (p * scale).astype(np.float32)


def pow_det(x, a):
    return exp2_det((_f32(a) * log2_det(x)).astype(np.float32))


# ------------------------------------------------------------------ replay ring
class ReplayRing:
    """General/Base/replay_buffer.py:20-65 (actions i32 / dones u8 on this build)."""

    def __init__(self, capacity: int, obs_dim: int):
        self.capacity, self.obs_dim = capacity, obs_dim
        self.states = np.zeros((capacity, obs_dim), np.float32)        # :28
        self.actions = np.zeros((capacity,), np.int32)                  # :29
        self.rewards = np.zeros((capacity,), np.float32)                # :30
        self.observations = np.zeros((capacity, obs_dim), np.float32)   # :31
        self.dones = np.zeros((capacity,), np.uint8)                    # :32
        self.counter = 0                                                # :33
        self.size = 0                                                   # :34

    def add(self, s, a, r, s2, d):
        """n rows in order, each at counter % N (:58-65). Returns the slots."""
        s = np.atleast_2d(np.asarray(s, np.float32))
        s2 = np.atleast_2d(np.asarray(s2, np.float32))
        n = s.shape[0]
        a = np.asarray(a, np.int32).reshape(n)
        r = np.asarray(r, np.float32).reshape(n)
        d = np.asarray(d).reshape(n).astype(bool).astype(np.uint8)
        slots = (self.counter + np.arange(n)) % self.capacity
        for j in range(n):            # sequential: later rows overwrite earlier ones on wrap
            k = slots[j]
            self.states[k], self.actions[k], self.rewards[k] = s[j], a[j], r[j]
            self.observations[k], self.dones[k] = s2[j], d[j]
        self.counter += n
        self.size = min(self.counter, self.capacity)
        return slots.astype(np.int32)

    def gather(self, idx):
        """replay_buffer.py:78-84"""
        return (self.states[idx], self.actions[idx], self.rewards[idx],
                self.observations[idx], self.dones[idx])


def uniform_indices(size: int, B: int, seed: int, ctr: int):
    """replay_buffer.py:77 randint(0, size, B) restated with Philox (numba RNG is opaque)."""
    x = philox_draw(seed, ctr, B, STREAM_UNIFORM)[:, 0].astype(np.uint64)
    return ((x * np.uint64(size)) >> np.uint64(32)).astype(np.int32)


# --------------------------------------------------------------------- sum-tree
class SumTree:
    """Proportional PER; not in the reference. float32[2N], root 1."""

    def __init__(self, L: int, alpha: float = 0.6, eps: float = 1e-6):
        self.L, self.N = L, 1 << L
        self.tree = np.zeros(2 * self.N, np.float32)
        self.pmax = np.float32(1.0)
        self.alpha, self.eps = np.float32(alpha), np.float32(eps)

    def _refresh(self, leaves):
        """Level-synchronous recompute of every touched parent as left + right."""
        nodes = np.unique(np.asarray(leaves, np.int64) + self.N)
        for _ in range(self.L):
            nodes = np.unique(nodes >> 1)
            self.tree[nodes] = (self.tree[2 * nodes] + self.tree[2 * nodes + 1]).astype(np.float32)

    def add(self, slots):
        slots = np.asarray(slots, np.int64)
        self.tree[self.N + slots] = self.pmax
        self._refresh(slots)

    def set(self, idx, prio):
        idx = np.asarray(idx, np.int64)
        prio = np.asarray(prio, np.float32)
        # highest batch position wins: numpy fancy assignment keeps the LAST occurrence
        # only by implementation accident, so make it explicit
        order = np.arange(len(idx))
        last = {}
        for i, k in zip(order, idx):
            last[int(k)] = i
        keys = np.fromiter(last.keys(), np.int64)
        pos = np.fromiter(last.values(), np.int64)
        self.tree[self.N + keys] = prio[pos]
        if len(prio):
            self.pmax = np.float32(max(self.pmax, prio.max()))
        self._refresh(keys)

    def priorities(self, td_abs):
        return pow_det((np.asarray(td_abs, np.float32) + self.eps).astype(np.float32), self.alpha)

    def update(self, idx, td_abs):
        self.set(idx, self.priorities(td_abs))

    def sample(self, size: int, B: int, beta: float, seed: int, ctr: int):
        total = self.tree[1]
        seg = np.float32(total / np.float32(B))
        U = u01(philox_draw(seed, ctr, B, STREAM_PER)[:, 0])
        u = ((np.arange(B, dtype=np.float32) + U).astype(np.float32) * seg).astype(np.float32)
        node = np.ones(B, np.int64)
        for _ in range(self.L):
            l = self.tree[2 * node]
            right = ~(u < l)
            u = np.where(right, (u - l).astype(np.float32), u).astype(np.float32)
            node = 2 * node + right.astype(np.int64)
        leaf = np.minimum(node - self.N, size - 1)
        p = self.tree[self.N + leaf]
        x = ((np.float32(size) * p).astype(np.float32) / total).astype(np.float32)
        w = pow_det(x, -np.float32(beta))
        return leaf.astype(np.int32), (w / w.max()).astype(np.float32)


# ---------------------------------------------------------------------- Q-net
def param_shapes(D, H1, H2, A):
    """haiku leaf order; w is [in,out] (LunarLander/dddqn.py:19-22)."""
    return [("model/~/linear", "w", (D, H1)), ("model/~/linear", "b", (H1,)),
            ("model/~/linear_1", "w", (H1, H2)), ("model/~/linear_1", "b", (H2,)),
            ("model/~/linear_2", "w", (H2, 1)), ("model/~/linear_2", "b", (1,)),
            ("model/~/linear_3", "w", (H2, A)), ("model/~/linear_3", "b", (A,))]


def param_count(D, H1, H2, A):
    return sum(int(np.prod(s)) for _, _, s in param_shapes(D, H1, H2, A))


def unflatten(P, dims):
    out, o = [], 0
    for _, _, s in param_shapes(*dims):
        n = int(np.prod(s))
        out.append(np.asarray(P[o:o + n]).reshape(s))
        o += n
    return out


def init_params(dims, seed: int):
    """hk.Linear default init: w ~ TruncatedNormal(stddev=1/sqrt(fan_in)) cut at +-2 sigma,
    b = 0 (dddqn.py:19-22 uses the defaults). Synthetic; own RNG (numpy PCG64)."""
    rng = np.random.default_rng(seed)
    flat = []
    for _, leaf, s in param_shapes(*dims):
        if leaf == "b":
            flat.append(np.zeros(s, np.float32).ravel())
        else:
            sd = 1.0 / np.sqrt(s[0])
            w = rng.standard_normal(s)
            bad = np.abs(w) > 2.0
            while bad.any():
                w[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(w) > 2.0
            flat.append((w * sd).astype(np.float32).ravel())
    return np.concatenate(flat)


def forward(P, x, dims, dtype=np.float64, return_hidden=False):
    """LunarLander/dddqn.py:24-31"""
    w1, b1, w2, b2, wv, bv, wa, ba = [t.astype(dtype) for t in unflatten(P, dims)]
    x = np.asarray(x, dtype)
    h1 = np.maximum(x @ w1 + b1, 0)                      # :25-26
    h2 = np.maximum(h1 @ w2 + b2, 0)                     # :27-28
    v = h2 @ wv + bv                                      # :29
    adv = h2 @ wa + ba                                    # :30
    q = v + adv - adv.mean(axis=1, keepdims=True)         # :31
    return (q, h1, h2) if return_hidden else q


def q_targets(P, Pt, s, a, r, s2, d, gamma, dims, dtype=np.float64, full=False):
    """General/QLearning/q_learning_functions.py:52-61, quirks Q3/Q4 kept."""
    A = dims[3]
    q = forward(P, s, dims, dtype)                        # :52
    nq = forward(P, s2, dims, dtype)                      # :53
    nt = forward(Pt, s2, dims, dtype)                     # :54
    astar = np.argmax(nq, axis=1)                         # :55 first max
    i = np.arange(len(a))
    r = np.asarray(r, dtype)
    d = np.asarray(d, dtype)
    delta = r + (1.0 - d) * (dtype(gamma) * nt[i, astar] - q[i, a])   # :58
    onehot = np.eye(A, dtype=dtype)[a]
    targets = q + delta[:, None] * onehot                 # :59
    if full:
        return dict(q=q, next_q=nq, next_q_tm=nt, astar=astar.astype(np.int32), delta=delta, targets=targets)
    return targets


def huber(e):
    """optax.huber_loss(delta=1)"""
    ae = np.abs(e)
    qd = np.minimum(ae, 1.0)
    return 0.5 * qd * qd + (ae - qd)


def loss(P, s, targets, dims, isw=None, dtype=np.float64):
    """q_learning_functions.py:35-36"""
    pred = forward(P, s, dims, dtype)
    rows = huber(pred - np.asarray(targets, dtype)).sum(axis=1)
    if isw is not None:
        rows = rows * np.asarray(isw, dtype)
    return rows.mean()


def grads(P, s, targets, dims, isw=None, dtype=np.float64):
    """Hand-derived jax.grad(compute_loss) (q_learning_functions.py:23). Returns
    (flat grad, loss, dL/dQ)."""
    D, H1, H2, A = dims
    w1, b1, w2, b2, wv, bv, wa, ba = [t.astype(dtype) for t in unflatten(P, dims)]
    x = np.asarray(s, dtype)
    B = x.shape[0]
    pred, h1, h2 = forward(P, x, dims, dtype, return_hidden=True)
    e = pred - np.asarray(targets, dtype)
    w = np.ones(B, dtype) if isw is None else np.asarray(isw, dtype)
    L = (huber(e).sum(axis=1) * w).mean()
    g = w[:, None] * np.clip(e, -1.0, 1.0) / B            # dL/dpred
    dv = g.sum(axis=1, keepdims=True)                      # dueling backward
    dadv = g - g.sum(axis=1, keepdims=True) / A
    gwv, gbv = h2.T @ dv, dv.sum(axis=0)
    gwa, gba = h2.T @ dadv, dadv.sum(axis=0)
    dz2 = (dv @ wv.T + dadv @ wa.T) * (h2 > 0)
    gw2, gb2 = h1.T @ dz2, dz2.sum(axis=0)
    dz1 = (dz2 @ w2.T) * (h1 > 0)
    gw1, gb1 = x.T @ dz1, dz1.sum(axis=0)
    flat = np.concatenate([t.ravel() for t in (gw1, gb1, gw2, gb2, gwv, gbv, gwa, gba)])
    return flat, L, g


def adam_step(P, g, mu, nu, count, lr, b1=0.9, b2=0.999, eps=1e-8, wd=1e-4, adamw=True,
              dtype=np.float64, grad_scale=1.0):
    """optax scale_by_adam -> add_decayed_weights (adamw only, all leaves) -> scale(-lr)
    -> apply_updates. Call sites Test/lunar_lander.py:48, q_learning_functions.py:24-25."""
    P, g, mu, nu = (np.asarray(t, dtype) for t in (P, g, mu, nu))
    b1f, b2f = dtype(np.float32(b1)), dtype(np.float32(b2))
    g = g * dtype(grad_scale)
    count = count + 1
    mu = b1f * mu + (1 - b1f) * g
    nu = b2f * nu + (1 - b2f) * g * g
    mhat = mu / (1 - b1f ** count)
    vhat = nu / (1 - b2f ** count)
    u = mhat / (np.sqrt(vhat) + dtype(np.float32(eps)))
    if adamw:
        u = u + dtype(np.float32(wd)) * P
    P = P - dtype(np.float32(lr)) * u
    return P, mu, nu, count


def act(P, s, dims, epsilon, seed, ctr, dtype=np.float64):
    """q_agent.py:137-141 + q_learning_functions.py:67-73, vectorised over rows."""
    n = len(s)
    q = forward(P, s, dims, dtype)
    o = philox_draw(seed, ctr, n, STREAM_POLICY)
    greedy = np.float32(epsilon) < u01(o[:, 0])
    rnd = ((o[:, 1].astype(np.uint64) * np.uint64(dims[3])) >> np.uint64(32)).astype(np.int32)
    return np.where(greedy, np.argmax(q, axis=1).astype(np.int32), rnd), q


def obs_augment(obs, step, max_steps):
    """LunarLander/env.py:19-21"""
    obs = np.asarray(obs, np.float32)
    frac = (np.asarray(step, np.float64) / max_steps)
    return np.concatenate([obs, frac[:, None]], axis=1).astype(np.float32)


def synth_env(n, D, seed, env_ctr, p_done):
    """Synthetic vector env (build spec); Philox stream 3, counter = env step.
    Returns (obs_next [n,D], r [n], d [n] u8)."""
    k = (np.arange(n, dtype=np.uint32)[:, None] * np.uint32(D + 1) + np.arange(D + 1, dtype=np.uint32)[None, :]).ravel()
    c = np.zeros((k.size, 4), np.uint32)
    c[:, 0] = env_ctr & 0xFFFFFFFF
    c[:, 1] = (env_ctr >> 32) & 0xFFFFFFFF
    c[:, 2] = k
    c[:, 3] = STREAM_ENV
    key = np.zeros((k.size, 2), np.uint32)
    key[:, 0] = seed & 0xFFFFFFFF
    key[:, 1] = (seed >> 32) & 0xFFFFFFFF
    o = philox4x32_10(c, key).reshape(n, D + 1, 4)
    u = u01(o)
    f = np.float32
    nrm = ((((u[..., 0] + u[..., 1]).astype(f) + (u[..., 2] + u[..., 3]).astype(f)).astype(f) - f(2.0)).astype(f)
           * f(1.73205078)).astype(f)
    obs_next = nrm[:, :D]
    last = u[:, D, :]
    done = last[:, 0] < f(p_done)
    rew = ((((last[:, 1] + last[:, 2]).astype(f) + (last[:, 3] + last[:, 0]).astype(f)).astype(f) - f(2.0)).astype(f)
           * f(1.73205078)).astype(f)
    rew = np.where(done, np.where(o[:, D, 1] & np.uint32(1), f(100.0), f(-100.0)), rew).astype(f)
    return obs_next.astype(f), rew, done.astype(np.uint8)


# ------------------------------------------------------------------ CartPole-v1
def _cp_poly(t2, coefs):
    p = np.full_like(t2, np.float32(coefs[0]))
    for c in coefs[1:]:
        p = (p * t2).astype(np.float32)
        p = (p + np.float32(c)).astype(np.float32)
    return p


def cartpole_step(s, a):
    """One Euler step of CartPole-v1 for n envs (classic-control cart-pole equations; BASELINE.json configs[2]),
    f32, one rounding per operation, polynomial sin / cos. s: (n,4) f32, a: (n,) {0,1}.
    Returns (next state, terminated)."""
    f = np.float32
    s = np.asarray(s, f).copy()
    x, xd, th, thd = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    force = np.where(np.asarray(a) == 1, f(10.0), f(-10.0)).astype(f)
    t2 = (th * th).astype(f)
    sn = (_cp_poly(t2, (-1.98412701e-4, 8.33333377e-3, -0.166666672, 1.0)) * th).astype(f)
    ct = _cp_poly(t2, (-1.38888892e-3, 4.16666679e-2, -0.5, 1.0))
    temp = ((force + ((f(0.05) * (thd * thd).astype(f)).astype(f) * sn).astype(f)).astype(f) / f(1.1)).astype(f)
    den = (f(0.5) * (f(1.33333337) - ((f(0.1) * (ct * ct).astype(f)).astype(f) / f(1.1)).astype(f)).astype(f)).astype(f)
    thacc = (((f(9.8) * sn).astype(f) - (ct * temp).astype(f)).astype(f) / den).astype(f)
    xacc = (temp - (((f(0.05) * thacc).astype(f) * ct).astype(f) / f(1.1)).astype(f)).astype(f)
    nx = (x + (f(0.02) * xd).astype(f)).astype(f)
    nxd = (xd + (f(0.02) * xacc).astype(f)).astype(f)
    nth = (th + (f(0.02) * thd).astype(f)).astype(f)
    nthd = (thd + (f(0.02) * thacc).astype(f)).astype(f)
    out = np.stack([nx, nxd, nth, nthd], axis=1).astype(f)
    term = (nx < f(-2.4)) | (nx > f(2.4)) | (nth < f(-0.20943951)) | (nth > f(0.20943951))
    return out, term


def cartpole_reset_states(n, seed, env_ctr):
    """fresh U(-0.05, 0.05)^4 states for all n envs at vector step env_ctr (only the done envs use theirs)"""
    o = philox_draw(seed, env_ctr, n, STREAM_ENV)
    return ((u01(o) * np.float32(0.1)).astype(np.float32) - np.float32(0.05)).astype(np.float32)


# ------------------------------------------------------------------ Nature-CNN dueling Q-network (BASELINE configs[4])
# Not in the reference: conv 32x8x8/4, 64x4x4/2, 64x3x3/1, fc 512, ReLU; the reference's dueling
# head (LunarLander/dddqn.py:29-31). NHWC, HWIO weights, frames u8 / 255. numpy f64 = truth for tolerances.
CNN_LAYERS = ((84, 84, 4, 20, 20, 32, 8, 8, 4), (20, 20, 32, 9, 9, 64, 4, 4, 2), (9, 9, 64, 7, 7, 64, 3, 3, 1), (1, 1, 3136, 1, 1, 512, 1, 1, 1))


def cnn_param_count(A):
    return sum(kh * kw * ic * oc + oc for (_, _, ic, _, _, oc, kh, kw, _) in CNN_LAYERS) + 512 + 1 + 512 * A + A


def cnn_init_params(A, seed):
    """haiku-style init: w ~ TruncatedNormal(1/sqrt(fan_in)) cut at 2 sigma, b = 0"""
    rng = np.random.default_rng(seed)
    parts = []
    fans = [(kh * kw * ic, oc) for (_, _, ic, _, _, oc, kh, kw, _) in CNN_LAYERS] + [(512, 1), (512, A)]
    for k, n in fans:
        sd = 1.0 / np.sqrt(k)
        w = np.clip(rng.standard_normal(k * n) * sd, -2 * sd, 2 * sd)
        parts += [w, np.zeros(n)]
    return np.concatenate(parts).astype(np.float32)


def cnn_forward(P, frames, A, dtype=np.float64, return_feat=False):
    """frames u8 [B,84,84,4] -> Q [B,A]"""
    P = np.asarray(P, dtype)
    x = np.asarray(frames).astype(dtype) / dtype(255.0)
    o = 0
    for (ih, iw, ic, oh, ow, oc, kh, kw, s) in CNN_LAYERS:
        K = kh * kw * ic
        W = P[o:o + K * oc].reshape(K, oc); o += K * oc
        b = P[o:o + oc]; o += oc
        B = x.shape[0]
        x = x.reshape(B, ih, iw, ic)
        cols = np.empty((B, oh, ow, kh, kw, ic), dtype)
        for a in range(kh):
            for c in range(kw):
                cols[:, :, :, a, c, :] = x[:, a:a + s * oh:s, c:c + s * ow:s, :]
        x = np.maximum(cols.reshape(B * oh * ow, K) @ W + b, 0).reshape(B, oh * ow * oc)
    wv = P[o:o + 512]; o += 512
    bv = P[o]; o += 1
    wa = P[o:o + 512 * A].reshape(512, A); o += 512 * A
    ba = P[o:o + A]
    v = x @ wv + bv
    adv = x @ wa + ba
    q = v[:, None] + adv - adv.mean(1, keepdims=True)
    return (q, x) if return_feat else q
