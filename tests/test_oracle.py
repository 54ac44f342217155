"""CPU tests of the oracle itself (no GPU): the plain-C and numpy restatements must agree
(bit-exactly on integer paths), and the FP maths must match PyTorch-CPU autograd + torch.optim.

PARITY UNPINNED: the reference holds no known-answer vectors for this path, so
the oracle is pinned only by these mutual cross-checks and by the Philox known-answer test.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import _oracle as oc
from _oracle import onp

CFGS = {"cfg1": (9, 32, 64, 4), "cfg2": (8, 256, 256, 4), "cfg3": (4, 64, 64, 2)}


def make_batch(dims, B, seed, terminal_frac=0.15):
    """inputs with >=10 % terminals (quirk Q3), a forced argmax tie, |delta| on both sides of the Huber knee"""
    rng = np.random.default_rng(seed)
    D, _, _, A = dims
    s = rng.standard_normal((B, D)).astype(np.float32)
    s2 = rng.standard_normal((B, D)).astype(np.float32)
    a = rng.integers(0, A, B).astype(np.int32)
    r = rng.standard_normal(B).astype(np.float32)
    d = (rng.random(B) < terminal_frac).astype(np.float32)
    r[d > 0] = rng.choice([-100.0, 100.0], int((d > 0).sum())).astype(np.float32)   # LunarLander terminal rewards
    r[::7] *= 0.05                                                                    # small |delta| rows
    s2[1] = 0.0                                                                       # with b=0: all Q equal -> tie
    return s, a, r, s2, d


# ------------------------------------------------------------------------------- RNG
def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10"""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got_np = onp.philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert tuple(int(x) for x in got_np) == want
        c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
        oc.lib().orc_philox4x32_10(c, k, o)
        assert tuple(o) == want


def test_pow_det_c_equals_numpy_and_is_accurate():
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(-14, 6, 4000)).astype(np.float32)
    for a in (0.6, -0.4, -1.0, 0.5, -0.7):
        c = np.array([oc.lib().orc_pow_det(float(v), a) for v in x], np.float32)
        n = onp.pow_det(x, a)
        assert np.array_equal(c.view(np.uint32), n.view(np.uint32))
        assert np.max(np.abs(n / np.power(x.astype(np.float64), a) - 1)) < 3e-6


# ----------------------------------------------------------------------- replay ring
def test_ring_add_wraps_like_reference():
    """General/Base/replay_buffer.py:58-65 semantics incl. wrap-around; C == numpy"""
    N, D = 50, 3
    cr, nr = oc.CReplay(N, D), onp.ReplayRing(N, D)
    rng = np.random.default_rng(1)
    for n in (7, 1, 30, 20, 13):
        s = rng.standard_normal((n, D)).astype(np.float32); s2 = rng.standard_normal((n, D)).astype(np.float32)
        a = rng.integers(0, 4, n); r = rng.standard_normal(n); d = rng.random(n) < 0.3
        assert np.array_equal(cr.add(s, a, r, s2, d), nr.add(s, a, r, s2, d))
    assert cr.size == nr.size == N and cr.rb.counter == nr.counter == 71
    for x, y in zip(cr.arrays(), (nr.states, nr.actions, nr.rewards, nr.observations, nr.dones)):
        assert np.array_equal(x, y)
    idx = oc.uniform_indices(cr.size, 64, 5, 9)
    assert np.array_equal(idx, onp.uniform_indices(nr.size, 64, 5, 9))
    assert idx.min() >= 0 and idx.max() < N
    for x, y in zip(cr.gather(idx), nr.gather(idx)):
        assert np.array_equal(x, y)


# -------------------------------------------------------------------------- sum-tree
@pytest.mark.parametrize("L,n_add", [(6, 40), (10, 1024), (12, 3000)])
def test_sumtree_c_equals_numpy_bitwise(L, n_add):
    ct, nt = oc.CPer(L), onp.SumTree(L)
    rng = np.random.default_rng(L)
    slots = np.arange(n_add, dtype=np.int32)
    ct.add(slots); nt.add(slots)
    assert np.array_equal(ct.tree, nt.tree)
    for it in range(4):
        B = 64
        ci, cw = ct.sample(n_add, B, 0.4 + 0.1 * it, 3, it)
        ni, nw = nt.sample(n_add, B, 0.4 + 0.1 * it, 3, it)
        assert np.array_equal(ci, ni)
        assert np.array_equal(cw.view(np.uint32), nw.view(np.uint32))
        assert np.all(np.diff(ci) >= 0)                       # stratified => sorted
        td = np.abs(rng.standard_normal(B)).astype(np.float32) * 3
        idx = ci.copy(); idx[5] = idx[40]                     # force a duplicate: position 40 must win
        ct.update(idx, td); nt.update(idx, td)
        assert np.array_equal(ct.tree.view(np.uint32), nt.tree.view(np.uint32))
        assert ct.pmax == nt.pmax
        prio = onp.pow_det((td + np.float32(1e-6)).astype(np.float32), np.float32(0.6))
        for leaf in np.unique(idx):                           # highest batch position wins
            assert ct.tree[(1 << L) + leaf] == prio[np.nonzero(idx == leaf)[0].max()]
    # invariants: every parent is the f32 sum of its children; root == f32 pairwise total
    t = ct.tree; N = 1 << L
    k = np.arange(1, N)
    assert np.array_equal(t[k], t[2 * k] + t[2 * k + 1])


def test_sumtree_alpha0_is_uniform():
    """alpha=0 => all priorities 1 => stratified-uniform over [0,size)"""
    L, n = 8, 200
    t = onp.SumTree(L, alpha=0.0)
    t.add(np.arange(n))
    t.update(np.arange(n), np.abs(np.random.default_rng(0).standard_normal(n)).astype(np.float32))
    assert np.all(t.tree[(1 << L):(1 << L) + n] == 1.0)
    idx, w = t.sample(n, 100, 0.5, 1, 0)
    assert np.all(w == 1.0) and idx.max() < n
    assert np.all(np.diff(idx) >= 1)                          # 100 strata over 200 equal leaves


# ------------------------------------------------------------------- network maths
@pytest.mark.parametrize("name", list(CFGS))
def test_forward_targets_c_vs_numpy_f64(name):
    dims = CFGS[name]
    P = onp.init_params(dims, 0); Pt = onp.init_params(dims, 1)
    P = P + 0.05 * np.random.default_rng(2).standard_normal(P.size).astype(np.float32)     # non-zero biases
    s, a, r, s2, d = make_batch(dims, 64, 3)
    qc, h1, h2 = oc.forward(dims, P, s)
    qn, n1, n2 = onp.forward(P, s, dims, np.float64, return_hidden=True)
    assert np.allclose(qc, qn, rtol=1e-5, atol=1e-5)
    assert np.allclose(h2, n2, rtol=1e-5, atol=1e-5)
    tc = oc.q_targets(dims, P, Pt, s, a, r, s2, d, 0.99)
    tn = onp.q_targets(P, Pt, s, a, r, s2, d, 0.99, dims, np.float64, full=True)
    assert np.array_equal(tc["astar"], tn["astar"])
    assert np.allclose(tc["targets"], tn["targets"], rtol=1e-5, atol=2e-5)
    # quirk Q3: terminal rows have target[a] == q[a] + r exactly (not r)
    term = d > 0
    i = np.arange(64)
    assert np.array_equal(tc["targets"][i, a][term], (tc["q"][i, a] + r)[term])
    # untaken actions keep q exactly (quirk Q4: q + delta*0)
    mask = np.ones_like(tc["q"], bool); mask[i, a] = False
    assert np.array_equal(tc["targets"][mask], tc["q"][mask])


def torch_model(P, dims):
    ws = [torch.tensor(np.array(t, np.float64), requires_grad=True) for t in onp.unflatten(P, dims)]
    def fwd(x):
        w1, b1, w2, b2, wv, bv, wa, ba = ws
        h1 = torch.relu(x @ w1 + b1); h2 = torch.relu(h1 @ w2 + b2)
        v = h2 @ wv + bv; adv = h2 @ wa + ba
        return v + adv - adv.mean(dim=1, keepdim=True)
    return ws, fwd


@pytest.mark.parametrize("name", list(CFGS))
@pytest.mark.parametrize("weighted", [False, True])
def test_grads_match_torch_autograd(name, weighted):
    """hand-derived backward (numpy f64 and C f32) == autograd of the same Huber loss"""
    dims = CFGS[name]
    P = onp.init_params(dims, 4) + 0.02
    Pt = onp.init_params(dims, 5)
    s, a, r, s2, d = make_batch(dims, 64, 6)
    targets = onp.q_targets(P, Pt, s, a, r, s2, d, 0.99, dims, np.float64)
    isw = np.random.default_rng(7).uniform(0.2, 1.0, 64) if weighted else None
    g_np, L_np, _ = onp.grads(P, s, targets, dims, isw, np.float64)
    ws, fwd = torch_model(P, dims)
    pred = fwd(torch.tensor(s, dtype=torch.float64))
    hub = torch.nn.functional.huber_loss(pred, torch.tensor(targets), reduction="none", delta=1.0).sum(dim=1)
    if weighted:
        hub = hub * torch.tensor(isw)
    loss = hub.mean()
    loss.backward()
    g_t = np.concatenate([w.grad.numpy().ravel() for w in ws])
    assert abs(L_np - loss.item()) < 1e-12
    assert np.max(np.abs(g_np - g_t)) < 1e-13
    g_c, L_c, _ = oc.grads(dims, P, s, targets.astype(np.float32), None if isw is None else isw.astype(np.float32))
    assert abs(L_c - L_np) < 1e-4 * max(1.0, abs(L_np))
    assert np.allclose(g_c, g_np, rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("adamw", [True, False])
def test_adam_matches_torch_optim(adamw):
    """SURVEY.md 8(a) O1 == torch.optim.AdamW/Adam (same maths as optax); C f32 tracks f64"""
    rng = np.random.default_rng(8)
    n = 500
    P0 = rng.standard_normal(n).astype(np.float32)
    lr = 2e-4 if adamw else 1e-4
    p_t = torch.tensor(P0.astype(np.float64), requires_grad=True)
    opt = (torch.optim.AdamW([p_t], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4) if adamw
           else torch.optim.Adam([p_t], lr=lr, betas=(0.9, 0.999), eps=1e-8))
    P, mu, nu, cnt = P0.astype(np.float64), np.zeros(n), np.zeros(n), 0
    Pc, muc, nuc, cc, p1, p2 = P0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32), 0, 1.0, 1.0
    copt = oc.Opt(lr, 0.9, 0.999, 1e-8, 1e-4, int(adamw))
    for _ in range(5):
        g = rng.standard_normal(n).astype(np.float32) * 0.1
        p_t.grad = torch.tensor(g.astype(np.float64)); opt.step()
        P, mu, nu, cnt = onp.adam_step(P, g, mu, nu, cnt, lr, adamw=adamw, dtype=np.float64)
        Pc, muc, nuc, cc, p1, p2 = oc.adam_step(copt, Pc, g, muc, nuc, cc, p1, p2)
    # f32-rounded hyper-parameters (b1=0.9f etc.) vs torch's f64 ones: ~1e-8 relative drift
    assert np.max(np.abs(P - p_t.detach().numpy())) < 5e-9
    assert np.allclose(Pc, P, rtol=1e-6, atol=1e-7) and cc == 5
    assert abs(p1 - 0.9 ** 5) < 1e-7


def test_act_c_equals_numpy():
    dims = CFGS["cfg1"]
    P = onp.init_params(dims, 9)
    s = np.random.default_rng(10).standard_normal((256, dims[0])).astype(np.float32)
    for eps in (0.0, 0.15, 1.0):
        a_c = oc.act(dims, P, s, eps, 11, 3)
        a_n, q = onp.act(P, s, dims, eps, 11, 3)
        assert np.array_equal(a_c, a_n)
    assert np.array_equal(oc.act(dims, P, s, 0.0, 1, 1), np.argmax(oc.forward(dims, P, s)[0], axis=1))


def test_obs_augment_matches_env_wrapper():
    """LunarLander/env.py:19-21"""
    obs = np.random.default_rng(12).standard_normal((5, 8)).astype(np.float32)
    step = np.array([0, 1, 749, 1499, 1500], np.int32)
    out = np.empty((5, 9), np.float32)
    oc.lib().orc_obs_augment(oc._p(obs), oc._p(step), C.c_int32(1500), C.c_int32(5), C.c_int32(8), oc._p(out))
    assert np.array_equal(out, onp.obs_augment(obs, step, 1500))
    assert np.array_equal(out[:, :8], obs) and out[4, 8] == 1.0


def test_learner_update_runs_and_learns():
    """the whole-update driver (cpu_baseline 'port') decreases loss on a fixed replay"""
    dims = CFGS["cfg1"]
    N = 512
    rb = oc.CReplay(N, dims[0]); per = oc.CPer(9)
    s, a, r, s2, d = make_batch(dims, N, 13, terminal_frac=0.05)
    r = np.clip(r, -1, 1)
    per.add(rb.add(s, a, r, s2, d))
    lrn = oc.CLearner(dims, oc.Opt(2e-3, 0.9, 0.999, 1e-8, 1e-4, 1), 0.99, 64, rb, per, onp.init_params(dims, 14), 15)
    losses = [lrn.update(64) for _ in range(300)]
    assert np.isfinite(losses).all() and np.mean(losses[-20:]) < np.mean(losses[:20])


@pytest.mark.parametrize("name,B,per_on", [("cfg1", 64, True), ("cfg2", 200, True), ("cfg3", 96, False)])
def test_omp_driver_is_bit_identical_to_scalar_driver(name, B, per_on):
    """oracle/dqn_oracle_omp.c (bench.py's all-core cpu_baseline leg) == the scalar whole-update driver and actor
    step, bit for bit: losses, parameters, moments, sampled indices, ring and tree after several iterations"""
    dims = CFGS[name]
    L_ = 10; N = 1 << L_
    s, a, r, s2, d = make_batch(dims, N, 21, terminal_frac=0.1)
    runs = []
    for omp in (False, True):
        rb = oc.CReplay(N, dims[0]); per = oc.CPer(L_) if per_on else None
        slots = rb.add(s, a, r, s2, d)
        if per is not None:
            per.add(slots)
        lrn = oc.CLearner(dims, oc.Opt(1e-3, 0.9, 0.999, 1e-8, 1e-4, 1), 0.99, B, rb, per, onp.init_params(dims, 3), 7)
        obs = np.random.default_rng(5).standard_normal((37, dims[0])).astype(np.float32)
        env_ctr, losses = 0, []
        for _ in range(5):
            for _ in range(2):
                env_ctr = (lrn.actor_step_omp if omp else lrn.actor_step)(obs, 0.3, 0.05, env_ctr)
            losses.append((lrn.update_omp if omp else lrn.update)(B))
        runs.append((np.array(losses, np.float32), lrn.params, lrn.mu, lrn.nu, obs.copy(),
                     np.ctypeslib.as_array(lrn.l.idx, shape=(B,)).copy(), [x.copy() for x in rb.arrays()],
                     per.tree.copy() if per is not None else np.zeros(1)))
    assert oc.lib().orc_omp_threads() >= 1
    for x, y in zip(runs[0][:6], runs[1][:6]):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
    for x, y in zip(runs[0][6], runs[1][6]):
        assert np.array_equal(x, y)
    assert np.array_equal(runs[0][7].view(np.uint32), runs[1][7].view(np.uint32))


def test_cnn_oracle_c_vs_numpy_f64():
    """Nature-CNN dueling forward (BASELINE configs[4]): the C restatement (f32 fmaf chains) against the numpy f64 one"""
    A = 6
    P = onp.cnn_init_params(A, 3)
    assert P.size == onp.cnn_param_count(A)
    rng = np.random.default_rng(4)
    P[-A:] = rng.standard_normal(A).astype(np.float32) * 0.1            # non-zero head biases
    frames = rng.integers(0, 256, (3, 84, 84, 4), dtype=np.uint8)
    q, feat = oc.cnn_forward(P, frames, A)
    q64, f64 = onp.cnn_forward(P, frames, A, np.float64, return_feat=True)
    assert np.allclose(feat, f64, rtol=1e-5, atol=1e-5) and np.allclose(q, q64, rtol=1e-5, atol=1e-5)
    assert np.abs(q64).max() > 1e-3 and feat.max() > 0
    # dueling identity: mean_a Q = val  (LunarLander/dddqn.py:31)
    o = onp.cnn_param_count(A) - (512 + 1 + 512 * A + A)
    val = f64 @ P[o:o + 512].astype(np.float64) + P[o + 512]
    assert np.allclose(q64.mean(1), val, rtol=1e-9, atol=1e-9)


def test_cnn_grad_oracle_finite_differences():
    """the hand-derived Nature-CNN backward (oracle/dqn_oracle_cnn_grad.inc): its f64 form against central differences of its
    own f64 loss along sign(gradient) restricted to one leaf at a time, and the f32 fmaf-chain form against the f64 one"""
    A = 6
    rng = np.random.default_rng(11)
    P = onp.cnn_init_params(A, 5)
    P = (P + 0.01 * rng.standard_normal(P.size)).astype(np.float32)
    B = 2
    frames = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
    q, _ = oc.cnn_forward(P, frames, A)
    targets = (q + rng.standard_normal((B, A)) * np.array([[0.3], [2.5]])).astype(np.float32)     # linear and clipped Huber branches
    isw = np.array([0.7, 1.0], np.float32)
    g64, l64 = oc.cnn_grads(P, frames, targets, isw, A, f64=True)
    g32, l32 = oc.cnn_grads(P, frames, targets, isw, A)
    assert abs(l32 - l64) < 1e-5 * max(1.0, abs(l64))
    # leaves: w, b of conv1..3, fc; wv, bv, wa, ba
    sizes = [8 * 8 * 4 * 32, 32, 4 * 4 * 32 * 64, 64, 3 * 3 * 64 * 64, 64, 3136 * 512, 512, 512, 1, 512 * A, A]
    assert sum(sizes) == P.size
    o = 0
    for n in sizes:
        leaf64 = g64[o:o + n]
        scale = np.abs(leaf64).max()
        assert scale > 0
        assert np.abs(g32[o:o + n] - leaf64).max() <= 1e-5 * max(scale, 1e-3), (o, n)
        d = np.zeros(P.size, np.float32)
        d[o:o + n] = np.where(leaf64 >= 0, 1.0, -1.0).astype(np.float32) * 2.0 ** -18   # along sign(grad): the derivative is sum |g|
        lp = oc.cnn_grads((P + d).astype(np.float32), frames, targets, isw, A, f64=True)[1]
        lm = oc.cnn_grads((P - d).astype(np.float32), frames, targets, isw, A, f64=True)[1]
        step = ((P + d).astype(np.float32).astype(np.float64) - (P - d).astype(np.float32).astype(np.float64))
        want = float(g64 @ step)
        assert abs((lp - lm) - want) <= 5e-3 * abs(want) + 1e-9, (o, n, lp - lm, want)
        o += n
