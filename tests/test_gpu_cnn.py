"""GPU parity tests (-m gpu) of the Nature-CNN dueling Q-network forward (BASELINE configs[4], PongNoFrameskip-v4 shape;
SURVEY.md 8(f) rank 4 -- not in the reference, PARITY UNPINNED): the implicit-GEMM kernels of dqn_cnn.hip, called through
the C ABI (dqn_cnn_*), against the CPU restatement (oracle/dqn_oracle_cnn.c, oracle_np.cnn_forward).

Bars: exact-f32 mode bit-identical to the C restatement's fmaf chains and within 1e-5 of f64; bf16 mode within 2e-2 of the
output scale; compute_q_targets (q_learning_functions.py:42-64) on top of it with the reference's terminal / one-hot rule."""
import numpy as np
import pytest

import _oracle as oc
from _oracle import onp

pytestmark = pytest.mark.gpu
A = 6            # Pong's action count


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def dq(torch_cuda):
    import deep_q_learning_amd as pkg
    return pkg


def make_params(seed):
    rng = np.random.default_rng(seed + 50)
    P = onp.cnn_init_params(A, seed)
    P = (P + 0.01 * rng.standard_normal(P.size)).astype(np.float32)       # non-zero biases everywhere
    return P


@pytest.mark.parametrize("B", [1, 5, 37])
def test_cnn_forward_f32_exact(dq, B):
    """exact-f32 MFMA mode: every conv / fc output is the k-ascending fmaf chain of the restatement -> identical bits; ragged
    last row tile of every layer (B * 400, B * 81, B * 49, B not multiples of the 64 / 128-row tiles)"""
    e = dq.CnnEngine(num_actions=A, max_batch=64, precision="f32")
    P, Pt = make_params(1), make_params(2)
    e.set_params(P); e.set_params(Pt, target=True)
    frames = np.random.default_rng(3 + B).integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
    q = host(e.forward(frames)); qt = host(e.forward(frames, target=True))
    qc, _ = oc.cnn_forward(P, frames, A)
    assert np.array_equal(q, qc), np.max(np.abs(q - qc))
    assert np.array_equal(qt, oc.cnn_forward(Pt, frames, A)[0])
    q64 = onp.cnn_forward(P, frames, A, np.float64)
    assert np.allclose(q, q64, rtol=1e-5, atol=1e-5)
    e.close()


def test_cnn_forward_bf16(dq):
    """bf16 MFMA mode (v_mfma_f32_32x32x16_bf16, f32 accumulate): 2e-2 of the output scale against f64; deterministic"""
    e = dq.CnnEngine(num_actions=A, max_batch=64, precision="bf16")
    P = make_params(4)
    e.set_params(P)
    frames = np.random.default_rng(5).integers(0, 256, (33, 84, 84, 4), dtype=np.uint8)
    q = host(e.forward(frames))
    q64 = onp.cnn_forward(P, frames, A, np.float64)
    scale = np.abs(q64).max()
    assert np.max(np.abs(q - q64)) <= 2e-2 * scale, (np.max(np.abs(q - q64)), scale)
    assert np.array_equal(q, host(e.forward(frames)))
    # not silently the f32 path
    assert np.max(np.abs(q - q64)) > 1e-6 * scale
    e.close()


def test_cnn_q_targets(dq):
    """compute_q_targets (q_learning_functions.py:42-64) with the CNN as the model: three forwards + the TD rule, incl.
    the terminal quirk Q3 (target of the taken action = q + r) and Q4 (q + delta * one_hot)"""
    B = 24
    e = dq.CnnEngine(num_actions=A, max_batch=32, precision="f32")
    P, Pt = make_params(6), make_params(7)
    e.set_params(P); e.set_params(Pt, target=True)
    rng = np.random.default_rng(8)
    s = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8); s2 = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
    a = rng.integers(0, A, B).astype(np.int32); r = rng.standard_normal(B).astype(np.float32); d = (rng.random(B) < 0.3).astype(np.float32)
    t = host(e.q_targets(s, a, r, s2, d, 0.99))
    q, _ = oc.cnn_forward(P, s, A); nq, _ = oc.cnn_forward(P, s2, A); nt, _ = oc.cnn_forward(Pt, s2, A)
    astar = nq.argmax(1)
    i = np.arange(B)
    t1 = np.float32(0.99) * nt[i, astar]; t2 = t1 - q[i, a]; t3 = (np.float32(1.0) - d) * t2; delta = r + t3      # :58
    want = q.copy(); want[i, a] = q[i, a] + delta                                                                 # :59
    assert np.array_equal(t, want)
    term = d > 0
    assert term.sum() >= 3 and np.array_equal(t[i, a][term], (q[i, a] + r)[term])                                 # quirk Q3
    e.close()


LEAVES = [("conv1.w", 8 * 8 * 4 * 32), ("conv1.b", 32), ("conv2.w", 4 * 4 * 32 * 64), ("conv2.b", 64), ("conv3.w", 3 * 3 * 64 * 64), ("conv3.b", 64),
          ("fc.w", 3136 * 512), ("fc.b", 512), ("val.w", 512), ("val.b", 1), ("adv.w", 512 * A), ("adv.b", A)]


def leaf_errors(g, ref):
    """max |g - ref| per leaf, relative to the leaf's own scale"""
    out, o = {}, 0
    for name, n in LEAVES:
        scale = np.abs(ref[o:o + n]).max()
        out[name] = (np.abs(g[o:o + n] - ref[o:o + n]).max() / max(scale, 1e-12), scale)
        o += n
    assert o == g.size
    return out


def grad_case(B, seed):
    rng = np.random.default_rng(seed)
    P = make_params(seed)
    frames = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
    q, _ = oc.cnn_forward(P, frames, A)
    targets = (q + rng.standard_normal((B, A)) * rng.choice([0.2, 2.5], (B, 1))).astype(np.float32)    # both Huber branches
    isw = rng.uniform(0.3, 1.0, B).astype(np.float32)
    return P, frames, targets, isw


@pytest.mark.parametrize("B,use_isw", [(1, False), (3, True), (16, False), (37, True)])
def test_cnn_grads_f32(dq, B, use_isw):
    """jax.grad(compute_loss) through the CNN (exact-f32 mode) against the f64 form of the restatement: 1e-5 of every leaf's
    scale (the north_star's bar), loss 1e-6; ragged tiles, slices and parity classes (B = 3, 37)"""
    P, frames, targets, isw = grad_case(B, 20 + B)
    e = dq.CnnEngine(num_actions=A, max_batch=64, precision="f32")
    e.set_params(P)
    loss = e.grads(frames, targets, isw if use_isw else None)
    g = host(e.get_buffer("grad"))
    g64, l64 = oc.cnn_grads(P, frames, targets, isw if use_isw else None, A, f64=True)
    assert abs(loss - l64) <= 1e-6 * max(1.0, abs(l64)), (loss, l64)
    errs = leaf_errors(g, g64)
    for name, (err, scale) in errs.items():
        assert scale > 0 and err <= 1e-5, (name, err, scale)
    # and as close to f64 as the f32 restatement itself is (same arithmetic, other summation order)
    g32, _ = oc.cnn_grads(P, frames, targets, isw if use_isw else None, A)
    e32 = leaf_errors(g32, g64)
    for name in errs:
        assert errs[name][0] <= max(4 * e32[name][0], 2e-6), (name, errs[name][0], e32[name][0])
    e.close()


def test_cnn_grads_bf16(dq):
    """bf16 mode. (a) every ReLU open (small weights, biases 1: the net is affine in its activations, the gradient a smooth
    function of the forward values): 2e-2 of every leaf's scale against f64 -- the arithmetic of the backward kernels.
    (b) the general case: a bf16 forward flips the gates of units whose pre-activation is within its rounding error of 0, and a
    flipped gate moves that unit's whole contribution (the gradient is discontinuous there; the f32 mode, bit-identical in the
    forward, has no such flips and meets 1e-5 above). Measured: relative rms error 4 % (fc) ... 12 % (conv1), correlation
    0.99+; asserted: rms <= 0.25, correlation >= 0.97 per weight leaf, head leaves 2e-2."""
    B = 24
    P, frames, targets, isw = grad_case(B, 77)
    e = dq.CnnEngine(num_actions=A, max_batch=64, precision="bf16")
    # (a)
    Po, o = P.copy(), 0
    for name, n in LEAVES[:8]:
        Po[o:o + n] = 1.0 if name.endswith(".b") else Po[o:o + n] * 0.1
        o += n
    e.set_params(Po)
    q, _ = oc.cnn_forward(Po, frames, A)
    tg = (q + (targets - oc.cnn_forward(P, frames, A)[0])).astype(np.float32)
    loss = e.grads(frames, tg, isw)
    g = host(e.get_buffer("grad"))
    g64, l64 = oc.cnn_grads(Po, frames, tg, isw, A, f64=True)
    assert abs(loss - l64) <= 2e-2 * max(1.0, abs(l64))
    for name, (err, scale) in leaf_errors(g, g64).items():
        assert scale > 0 and err <= 2e-2, (name, err, scale)
    # (b)
    e.set_params(P)
    loss = e.grads(frames, targets, isw)
    g = host(e.get_buffer("grad"))
    g64, l64 = oc.cnn_grads(P, frames, targets, isw, A, f64=True)
    assert abs(loss - l64) <= 2e-2 * max(1.0, abs(l64))
    o = 0
    for name, n in LEAVES:
        got, ref = g[o:o + n].astype(np.float64), g64[o:o + n]
        if name.startswith(("val", "adv")):
            assert np.abs(got - ref).max() <= 2e-2 * np.abs(ref).max(), name
        elif name.endswith(".w"):
            assert np.sqrt(((got - ref) ** 2).mean()) <= 0.25 * np.sqrt((ref ** 2).mean()), name
            assert np.corrcoef(got, ref)[0, 1] >= 0.97, name
        o += n
    e.close()


@pytest.mark.parametrize("precision,tol", [("f32", 2e-5), ("bf16", 3e-2)])
def test_cnn_update_tracks_oracle(dq, precision, tol):
    """Agent._step on given minibatches: compute_q_targets (three forwards) + train_step (gradient + AdamW) + a target copy,
    two updates, against the restatement (orc_cnn_forward / TD rule / orc_cnn_grads / orc_adam_step)"""
    B, gamma = 8, 0.99
    rng = np.random.default_rng(5)
    P, Pt = make_params(31), make_params(32)
    e = dq.CnnEngine(num_actions=A, max_batch=16, precision=precision)
    e.set_params(P); e.set_params(Pt, target=True)
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-8, 1e-4
    e.set_optimizer(lr=lr, b1=b1, b2=b2, eps=eps, weight_decay=wd, adamw=True)
    opt = oc.Opt(lr, b1, b2, eps, wd, 1)
    Pc, mu, nu, st = P.copy(), np.zeros_like(P), np.zeros_like(P), (0, 1.0, 1.0)
    for it in range(2):
        s = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8); s2 = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
        a = rng.integers(0, A, B).astype(np.int32); r = rng.standard_normal(B).astype(np.float32); d = (rng.random(B) < 0.3).astype(np.float32)
        isw = rng.uniform(0.3, 1.0, B).astype(np.float32)
        loss = e.update(s, a, r, s2, d, isw, gamma, want_loss=True)
        q, _ = oc.cnn_forward(Pc, s, A); nq, _ = oc.cnn_forward(Pc, s2, A); nt, _ = oc.cnn_forward(Pt, s2, A)
        i = np.arange(B)
        t3 = (np.float32(1.0) - d) * (np.float32(gamma) * nt[i, nq.argmax(1)] - q[i, a])                          # q_learning_functions.py:58
        tg = q.copy(); tg[i, a] = q[i, a] + (r + t3)                                                               # :59
        g, lc = oc.cnn_grads(Pc, s, tg, isw, A)
        assert abs(loss - lc) <= tol * max(1.0, abs(lc)), (it, loss, lc)
        Pc, mu, nu, *st = oc.adam_step(opt, Pc, g, mu, nu, *st)
        got = host(e.get_buffer("params"))
        if precision == "f32":
            assert np.abs(got - Pc).max() <= tol, (it, np.abs(got - Pc).max())
        else:        # an Adam step moves an element by at most ~lr whatever the gradient: sign flips of tiny gradients cost 2 lr per step
            assert np.abs(got - Pc).max() <= 2.1 * lr * (it + 1) and np.abs(got - Pc).mean() <= 0.25 * lr * (it + 1), (it, np.abs(got - Pc).max(), np.abs(got - Pc).mean())
        if it == 0:
            e.sync_target(); Pt = Pc.copy()
            assert np.array_equal(host(e.get_buffer("target")), host(e.get_buffer("params")))
    e.close()


def test_cnn_data_parallel_halves(dq):
    """per-GPU learners with the CNN: two handles take half a minibatch each, the gradient buffers are summed in
    place (what the all-reduce does) and applied with grad_scale = 1/2 -- the same parameters as one learner on the whole
    minibatch (1e-5 of the step; summation order differs)"""
    B = 16
    P, frames, targets, isw = grad_case(B, 91)
    full, a, b = (dq.CnnEngine(num_actions=A, max_batch=16, precision="f32") for _ in range(3))
    for e in (full, a, b):
        e.set_params(P); e.set_optimizer(lr=1e-3, weight_decay=0.0, adamw=False)
    full.grads(frames, targets, isw); full.optimizer_step()
    a.grads(frames[:8], targets[:8], isw[:8]); b.grads(frames[8:], targets[8:], isw[8:])
    ga, gb = a.buffer("grad"), b.buffer("grad")
    tot = ga + gb
    ga.copy_(tot); gb.copy_(tot)
    a.optimizer_step(grad_scale=0.5); b.optimizer_step(grad_scale=0.5)
    pf, pa, pb = host(full.get_buffer("params")), host(a.get_buffer("params")), host(b.get_buffer("params"))
    assert np.array_equal(pa, pb)
    gf = host(full.get_buffer("grad"))
    assert np.abs(host(tot) * 0.5 - gf).max() <= 1e-5 * np.abs(gf).max()
    moved = np.abs(pf - P) > 0
    # Adam's first step is lr * g / (|g| + eps): only elements whose gradient is within the summation noise of 0 may differ
    assert moved.mean() > 0.3 and np.mean(np.abs(pa - pf) > 2e-5) < 1e-3        # (inputs that are 0 for all 16 samples leave their fc rows untouched)
    for e in (full, a, b):
        e.close()


def test_cnn_frame_ring_and_gather(dq):
    """ReplayBuffer.add / the gather of sample_batch (replay_buffer.py:58-65, :79-85) for frame stacks: rows land at
    counter % capacity incl. the wrap, gathered rows are the bytes that were added (duplicates and the last row included)"""
    import torch
    cap, n = 24, 10
    e = dq.CnnEngine(num_actions=A, max_batch=16, precision="bf16")
    e.replay_init(cap)
    rng = np.random.default_rng(3)
    ring = {k: None for k in "s a r s2 d".split()}
    model_s = np.zeros((cap, 84, 84, 4), np.uint8); model_s2 = np.zeros_like(model_s)
    model_a = np.zeros(cap, np.int32); model_r = np.zeros(cap, np.float32); model_d = np.zeros(cap, np.float32)
    ctr = 0
    for step in range(4):                                               # 40 rows into 24: wraps in the third add
        s = rng.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8); s2 = rng.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8)
        a = rng.integers(0, A, n).astype(np.int32); r = rng.standard_normal(n).astype(np.float32); d = (rng.random(n) < 0.3).astype(np.float32)
        first = e.replay_add(s, a, r, s2, d)
        assert first == ctr % cap
        pos = (ctr + np.arange(n)) % cap
        model_s[pos], model_s2[pos], model_a[pos], model_r[pos], model_d[pos] = s, s2, a, r, d
        ctr += n
        assert e.replay_size() == (min(ctr, cap), ctr)
    idx = np.array([0, 0, 5, 9, 10, 17, 23, 23, 3], np.int32)
    gs, ga, gr, gs2, gd = (host(t) for t in e.replay_gather(idx))
    assert np.array_equal(gs, model_s[idx]) and np.array_equal(gs2, model_s2[idx])
    assert np.array_equal(ga, model_a[idx]) and np.array_equal(gr, model_r[idx]) and np.array_equal(gd, model_d[idx])
    # update_from_replay == update on the gathered rows (same handle state: two handles, same parameters)
    e2 = dq.CnnEngine(num_actions=A, max_batch=16, precision="bf16")
    P = make_params(4)
    for x in (e, e2):
        x.set_params(P); x.set_params(P, target=True); x.set_optimizer(lr=1e-3)
    isw = rng.uniform(0.3, 1.0, idx.size).astype(np.float32)
    td = torch.empty(idx.size, dtype=torch.float32, device="cuda")
    l1 = e.update_from_replay(idx, isw, 0.99, td_abs_out=td, want_loss=True)
    l2 = e2.update(gs, ga, gr, gs2, gd, isw, 0.99, want_loss=True)
    assert l1 == l2 and np.array_equal(host(e.get_buffer("params")), host(e2.get_buffer("params")))
    # |delta| of q_learning_functions.py:58 on the bf16 forward's own Q values
    e3 = dq.CnnEngine(num_actions=A, max_batch=16, precision="bf16"); e3.set_params(P); e3.set_params(P, target=True)
    q, nq, nt = host(e3.forward(gs)), host(e3.forward(gs2)), host(e3.forward(gs2, target=True))
    i = np.arange(idx.size)
    delta = gr + (np.float32(1.0) - gd) * (np.float32(0.99) * nt[i, nq.argmax(1)] - q[i, ga])
    assert np.array_equal(host(td), np.abs(delta))
    for x in (e, e2, e3):
        x.close()


def test_cnn_act(dq):
    """Agent._policy (q_agent.py:137-141) with the CNN: epsilon = 0 -> argmax of the CNN's Q (first maximum); epsilon = 1 -> the
    same Philox draws as dqn_act of the MLP engine (policy stream, (seed, ctr, i)); repeatable"""
    n = 40
    e = dq.CnnEngine(num_actions=4, max_batch=64, precision="f32")
    rng = np.random.default_rng(9)
    P = onp.cnn_init_params(4, 9); P = (P + 0.01 * rng.standard_normal(P.size)).astype(np.float32)
    e.set_params(P)
    frames = rng.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8)
    q, _ = oc.cnn_forward(P, frames, 4)
    assert np.array_equal(host(e.act(frames, 0.0, 5, 7)), q.argmax(1).astype(np.int32))
    a1 = host(e.act(frames, 1.0, 5, 7))
    assert np.array_equal(a1, host(e.act(frames, 1.0, 5, 7))) and not np.array_equal(a1, host(e.act(frames, 1.0, 5, 8)))
    m = dq.Engine(dq.EngineConfig(obs_dim=8, hidden1=16, hidden2=16, num_actions=4, capacity=64, use_per=False, max_batch=64, seed=1))
    am = host(m.act(np.zeros((n, 8), np.float32), 1.0, 5, 7))
    assert np.array_equal(a1, am) and set(a1) == {0, 1, 2, 3}
    half = host(e.act(frames, 0.5, 5, 7))
    greedy = q.argmax(1)
    assert 0.2 < np.mean(half == greedy) < 1.0 and np.mean(half != greedy) > 0.1
    e.close(); m.close()


def test_cnn_vector_agent_loop(dq):
    """the loop of BASELINE configs[4]'s shape (synthetic frames): act -> add to both rings -> PER sample -> update from the frame
    ring -> priority write-back. Checks the plumbing: ring counters in lockstep, priorities of sampled rows move off the
    initial maximum, parameters move, losses finite, no device error."""
    from deep_q_learning_amd.General.QLearning.cnn_agent import CnnVectorAgent
    ag = CnnVectorAgent(n_envs=16, num_actions=A, capacity=64, batch_size=32, precision="bf16", train_frequency=2, replace_frequency=2, lr=1e-3, seed=3)
    P = make_params(12)
    ag.init_params(P)
    losses = ag.training(3, want_loss=True)
    assert len(losses) == 3 and all(np.isfinite(l) and l > 0 for l in losses)
    assert ag.env_steps == 2 + 3 * 2 and ag.updates == 3
    assert ag.cnn.replay_size() == (64, ag.env_steps * 16) and ag.index.replay_size() == (64, ag.env_steps * 16)
    assert np.abs(host(ag.cnn.get_buffer("params")) - P).max() > 0
    tree = host(ag.index.buffer(dq._lib.BUF_TREE))
    leaves = tree[64:128]
    assert leaves.min() > 0 and len(np.unique(leaves)) > 8            # written-back |delta|^alpha next to max-priority new rows
    assert ag.index.device_errors() == 0
    ag.close()


def test_cnn_against_golden(dq):
    """the HIP path against the committed vector tests/golden/cnn_B4_seed7.npz (no restatement in the loop): exact-f32 forward
    bit-identical to the stored f32 Q and within 1e-5 of the stored f64 Q, loss and per-leaf gradient sums / samples within 1e-5
    of the leaf scale"""
    import os, sys
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    import make_cnn_golden as mk
    g = np.load(os.path.join(gold, "cnn_B4_seed7.npz"))
    P, frames, noise, scale, isw = mk.inputs(int(g["seed"]), int(g["B"]))
    e = dq.CnnEngine(num_actions=int(g["A"]), max_batch=8, precision="f32")
    e.set_params(P)
    q = host(e.forward(frames))
    assert np.array_equal(q, g["q32"]) and np.allclose(q, g["q64"], rtol=1e-5, atol=1e-5)
    loss = e.grads(frames, g["targets"], isw)
    assert abs(loss - float(g["loss64"])) <= 1e-6 * max(1.0, float(g["loss64"]))
    gr = host(e.get_buffer("grad")).astype(np.float64)
    o = 0
    for n, s_, a_ in zip(mk.LEAVES, g["leaf_sums"], g["leaf_abs"]):
        leaf = gr[o:o + n]
        assert abs(np.abs(leaf).sum() - a_) <= 1e-5 * a_ and abs(leaf.sum() - s_) <= 1e-5 * a_
        o += n
    ref = g["grad64_strided"]
    assert np.abs(gr[::997] - ref).max() <= 1e-5 * np.abs(ref).max()
    e.close()


def test_cnn_nstep_gather(dq):
    """n-step returns from the frame ring (arithmetic of nstep_row / the MLP engine): rows are stored one env
    step each, step-major; the transition that starts at row p is (s_p, a_p, R = r_0 + g (r_1 + g r_2) cut after the first done,
    s' of the last of the n rows, done_n) -- against a numpy model on a wrapped ring, f32 Horner form, bit-exact"""
    cap, n_envs, n, g = 32, 4, 3, np.float32(0.9)
    e = dq.CnnEngine(num_actions=A, max_batch=16, precision="bf16")
    e.replay_init(cap)
    rng = np.random.default_rng(13)
    S = np.zeros((cap, 84, 84, 4), np.uint8); S2 = np.zeros_like(S)
    Aa = np.zeros(cap, np.int32); R = np.zeros(cap, np.float32); D = np.zeros(cap, np.float32)
    steps = 11                                                     # 44 rows into 32: wrapped
    for t in range(steps):
        s = rng.integers(0, 256, (n_envs, 84, 84, 4), dtype=np.uint8); s2 = rng.integers(0, 256, (n_envs, 84, 84, 4), dtype=np.uint8)
        a = rng.integers(0, A, n_envs).astype(np.int32); r = rng.standard_normal(n_envs).astype(np.float32); d = (rng.random(n_envs) < 0.35).astype(np.float32)
        pos = (t * n_envs + np.arange(n_envs)) % cap
        e.replay_add(s, a, r, s2, d)
        S[pos], S2[pos], Aa[pos], R[pos], D[pos] = s, s2, a, r, d
    # rows of the steps steps-n .. whose windows are complete and whose frames are still there: steps 3 .. 8
    valid_steps = np.arange(steps - cap // n_envs, steps - n + 1)
    idx = np.concatenate([(t * n_envs + np.arange(n_envs)) % cap for t in valid_steps]).astype(np.int32)[:16]
    gs, ga, gr, gs2, gd = (host(x) for x in e.replay_gather(idx, n_step=n, n_envs=n_envs, gamma=float(g)))
    wr, wd, wlast = np.zeros(idx.size, np.float32), np.zeros(idx.size, np.float32), np.zeros(idx.size, np.int64)
    for i, p in enumerate(idx):
        rows = [(p + k * n_envs) % cap for k in range(n)]
        last = n - 1
        for k in range(n - 2, -1, -1):
            if D[rows[k]] != 0:
                last = k
        acc = R[rows[last]]
        for k in range(last - 1, -1, -1):
            acc = np.float32(R[rows[k]] + np.float32(g * acc))
        wr[i], wd[i], wlast[i] = acc, (D[rows[n - 1]] if last == n - 1 else 1.0), rows[n - 1]
    assert np.array_equal(ga, Aa[idx]) and np.array_equal(gr, wr) and np.array_equal(gd, wd)
    assert np.array_equal(gs, S[idx]) and np.array_equal(gs2, S2[wlast])
    assert 0 < wd.sum() < idx.size                                   # both truncated and full windows occurred
    # n_step = 1 is the plain gather
    g1 = [host(x) for x in e.replay_gather(idx)]
    assert np.array_equal(g1[2], R[idx]) and np.array_equal(g1[4], D[idx]) and np.array_equal(g1[3], S2[idx])
    e.close()


def test_cnn_vector_agent_nstep(dq):
    """the loop with n_step = 3: the PER index runs n - 1 vector steps behind the frame ring, overwritten rows leave the draw
    until their successors exist, sampled rows always have complete windows"""
    from deep_q_learning_amd.General.QLearning.cnn_agent import CnnVectorAgent
    ag = CnnVectorAgent(n_envs=8, num_actions=A, capacity=64, batch_size=16, precision="bf16", train_frequency=3, replace_frequency=2, lr=1e-3, seed=4, n_step=3, p_done=0.2)
    ag.init_params(make_params(21))
    seen = []
    orig = ag.cnn.update_from_replay
    def spy(idx, *a, **k):
        seen.append((ag.env_steps, host(idx).copy()))
        return orig(idx, *a, **k)
    ag.cnn.update_from_replay = spy
    losses = ag.training(6, want_loss=True)          # 4 + 18 = 22 vector steps: 176 rows into 64 -> the ring wraps twice
    assert len(losses) == 6 and all(np.isfinite(l) for l in losses)
    assert ag.index.replay_size()[1] == (ag.env_steps - 2) * 8 and ag.cnn.replay_size()[1] == ag.env_steps * 8
    for steps_done, idx in seen:
        # a sampled row's step must be one of the last capacity / n_envs steps, and at least n - 1 steps old
        newest = steps_done - 1
        step_of = lambda p: max(t for t in range(max(0, newest - 7), newest + 1) if (t * 8) % 64 <= p < (t * 8) % 64 + 8)
        for p in idx:
            assert newest - step_of(int(p)) >= 2, (steps_done, int(p))
    assert ag.index.device_errors() == 0
    ag.close()


def test_cnn_update_in_a_captured_graph(dq):
    """dqn_cnn_update forks to the handle's side stream and joins by events: captured into a hipGraph (torch.cuda.graph) and
    replayed it must do what the eager launches do -- same parameters after two steps, bit for bit"""
    import torch
    B = 8
    rng = np.random.default_rng(17)
    P = make_params(41)
    s = torch.as_tensor(rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)).cuda(); s2 = torch.as_tensor(rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)).cuda()
    a = torch.as_tensor(rng.integers(0, A, B).astype(np.int32)).cuda(); r = torch.as_tensor(rng.standard_normal(B).astype(np.float32)).cuda()
    d = torch.as_tensor((rng.random(B) < 0.3).astype(np.float32)).cuda(); w = torch.as_tensor(rng.uniform(0.3, 1.0, B).astype(np.float32)).cuda()
    eager, graphed = (dq.CnnEngine(num_actions=A, max_batch=8, precision="f32") for _ in range(2))
    for e in (eager, graphed):
        e.set_params(P); e.set_params(P, target=True); e.set_optimizer(lr=1e-3)
    for _ in range(2):
        eager.update(s, a, r, s2, d, w)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            graphed.update(s, a, r, s2, d, w)
    torch.cuda.synchronize()
    assert np.array_equal(host(graphed.get_buffer("params")), P)        # capture launches nothing
    g.replay(); g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(host(graphed.get_buffer("params")), host(eager.get_buffer("params")))
    assert np.array_equal(host(graphed.get_buffer("mu")), host(eager.get_buffer("mu")))
    del g
    eager.close(); graphed.close()


@pytest.mark.parametrize("B", [37, 130])
def test_cnn_fc_wide_tile_bf16(dq, B):
    """k_cnn_layer<..., 3, 1> -- the 128 x 128 fc tile (2 x 2 MFMA tiles per wave) the bf16 mode takes
    from 8 192 rows -- forced at test sizes (dqn_cnn_set_flags(DQN_CNN_FLAG_FC_WIDE_TILE)), ragged in m (37, 130 rows of a
    128-row tile). Every output element is the same k-ascending chain of v_mfma_f32_32x32x16_bf16 accumulations as in the
    32 x 32 tile, so Q and every gradient leaf must be BIT-IDENTICAL to the narrow-tile result; against f64: 2e-2 of scale for
    the forward and the loss."""
    e = dq.CnnEngine(num_actions=A, max_batch=B, precision="bf16")
    P, frames, targets, isw = grad_case(B, 300 + B)
    Po, o = P.copy(), 0
    for name, n in LEAVES[:8]:
        Po[o:o + n] = 1.0 if name.endswith(".b") else Po[o:o + n] * 0.1
        o += n
    e.set_params(P)
    q_narrow = host(e.forward(frames)).copy()
    e.set_params(Po)
    qo = oc.cnn_forward(Po, frames, A)[0]
    tg = (qo + (targets - oc.cnn_forward(P, frames, A)[0])).astype(np.float32)
    l_narrow = e.grads(frames, tg, isw); g_narrow = host(e.get_buffer("grad")).copy()
    e.set_flags(dq._lib.CNN_FLAG_FC_WIDE_TILE)
    e.set_params(P)
    q_wide = host(e.forward(frames))
    assert np.array_equal(q_wide, q_narrow)
    q64 = onp.cnn_forward(P, frames, A, np.float64)
    assert np.max(np.abs(q_wide - q64)) <= 2e-2 * np.abs(q64).max()
    e.set_params(Po)
    l_wide = e.grads(frames, tg, isw); g_wide = host(e.get_buffer("grad"))
    assert l_wide == l_narrow and np.array_equal(g_wide, g_narrow)
    g64, l64 = oc.cnn_grads(Po, frames, tg, isw, A, f64=True)
    assert abs(l_wide - l64) <= 2e-2 * max(1.0, abs(l64))
    # Against f64 the gradient leaves are exactly as far as the narrow tile's (bit-equal). How far that is depends on the batch, not
    # on the tile: with targets within the Huber knee the loss gradient is e = q - target itself, so the bf16 forward's error in
    # q (2e-2 of scale) enters every leaf, and a leaf whose f64 value is a sum with cancellation (val.b, val.w) shows it as a
    # large RELATIVE error -- measured (tools/diag/cnn_grad_b.py, every ReLU open): fc.w 2.3e-2 / 5.3e-2 / 8.9e-2 and val.w
    # 0.11 / 0.10 / 0.32 at B = 24 / 64 / 130, while the exact-f32 mode stays at 2e-6 on every leaf at all of them. The bf16
    # gradient bars are those of test_cnn_grads_bf16 and of test_cnn_configs4_size_512 (c); here: bit-identity + the loss.
    e.close()


@pytest.mark.parametrize("B", [1, 2, 5, 37, 130, 515])
def test_cnn_trunk_bitexact(dq, B):
    """r03: k_cnn_trunk16 -- conv1, conv2, conv3 of the bf16 mode as one persistent kernel (frames by LDS-DMA, maps in LDS, all
    weight fragments in registers) -- against the per-layer kernels (DQN_CNN_FLAG_LAYERWISE_CONV). Same MFMA, same k order per
    output element, same epilogue: Q (conv3's map through fc + heads), the loss and EVERY gradient leaf (conv1's and conv2's
    maps are read by the backward) must be BIT-IDENTICAL; odd B = a pair with one image, 515 = more pairs than CUs (the
    persistent loop and its DMA prefetch). Then one whole Agent._step (paired online pass from two tensors + target pass)."""
    import torch
    e = dq.CnnEngine(num_actions=A, max_batch=B, precision="bf16")
    rng = np.random.default_rng(900 + B)
    P = make_params(70 + (B % 7))
    frames = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
    targets = rng.normal(0, 1, (B, A)).astype(np.float32); isw = rng.uniform(0.2, 1.0, B).astype(np.float32)
    res = {}
    for name, flag in (("fused", 0), ("layerwise", dq._lib.CNN_FLAG_LAYERWISE_CONV)):
        e.set_flags(flag)
        e.set_params(P)
        q = host(e.forward(frames)).copy()
        loss = e.grads(frames, targets, isw); g = host(e.get_buffer("grad")).copy()
        res[name] = (q, loss, g)
    assert np.array_equal(res["fused"][0], res["layerwise"][0])
    assert res["fused"][1] == res["layerwise"][1] and np.array_equal(res["fused"][2], res["layerwise"][2])
    if B <= 37:
        q64 = onp.cnn_forward(P, frames, A, np.float64)
        assert np.max(np.abs(res["fused"][0] - q64)) <= 2e-2 * np.abs(q64).max()
    if B in (5, 130):
        s2 = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
        a = rng.integers(0, A, B).astype(np.int32); r = rng.normal(0, 1, B).astype(np.float32); d = (rng.random(B) < 0.1).astype(np.float32)
        outs = []
        for flag in (0, dq._lib.CNN_FLAG_LAYERWISE_CONV):
            e.set_flags(flag)
            e.set_params(P); e.set_params(make_params(71), target=True); e.set_optimizer(lr=1e-3)
            e.update(frames, a, r, s2, d, isw=isw, gamma=0.99)
            outs.append(host(e.get_buffer("params")).copy())
        assert np.array_equal(outs[0], outs[1])
    e.close()


def test_cnn_configs4_size_512(dq):
    """BASELINE configs[4]'s per-GPU size -- 512 frame stacks -- checked, not only timed.
    (a) exact-f32 forward of 512 stacks bit-identical to the C restatement (orc_cnn_forward) and 1e-5 of f64 on a subset;
    (b) Agent._step from a 512-env frame ring with 3-step returns (dqn_cnn_update_replay) == dqn_cnn_update on the rows
        dqn_cnn_replay_gather assembles: bit-identical parameters, in both precisions;
    (c) the bf16 gradient of the 512-row batch against the exact-f32 one: correlation >= 0.97 and rms <= 0.25 per weight
        leaf (the bars of test_cnn_grads_bf16 (b))."""
    import torch
    n, B, n_step, gamma = 512, 512, 3, 0.99
    rng = np.random.default_rng(512)
    P, Pt = make_params(61), make_params(62)
    frames = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
    e = dq.CnnEngine(num_actions=A, max_batch=B, precision="f32")
    e.set_params(P); e.set_params(Pt, target=True)
    q = host(e.forward(frames))
    qc, _ = oc.cnn_forward(P, frames, A)
    assert np.array_equal(q, qc), np.max(np.abs(q - qc))
    sub = rng.choice(B, 24, replace=False)
    assert np.allclose(q[sub], onp.cnn_forward(P, frames[sub], A, np.float64), rtol=1e-5, atol=1e-5)
    # (b) ring of 5 env steps x 512 envs (step-major), sampled rows whose 3-step window exists
    steps = 5
    grads = {}
    for precision in ("f32", "bf16"):
        out = {}
        for mode in ("ring", "rows"):
            c = dq.CnnEngine(num_actions=A, max_batch=B, precision=precision)
            c.set_params(P); c.set_params(Pt, target=True)
            c.set_optimizer(lr=1e-3)
            c.replay_init(steps * n)
            r2 = np.random.default_rng(77)
            for t in range(steps):
                c.replay_add(r2.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8), r2.integers(0, A, n), r2.standard_normal(n).astype(np.float32),
                             r2.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8), (r2.random(n) < 0.1).astype(np.float32))
            idx = np.sort(np.random.default_rng(78).integers(0, (steps - n_step + 1) * n, B)).astype(np.int32)
            isw = np.random.default_rng(79).uniform(0.3, 1.0, B).astype(np.float32)
            if mode == "ring":
                td = torch.empty(B, dtype=torch.float32, device="cuda")
                loss = c.update_from_replay(idx, isw, gamma, td_abs_out=td, want_loss=True, n_step=n_step, n_envs=n)
                out["td"] = host(td)
            else:
                s, a, r, s2, d = c.replay_gather(idx, n_step=n_step, n_envs=n, gamma=gamma)
                gn = np.float32(gamma)
                for _ in range(1, n_step):
                    gn = np.float32(gn * np.float32(gamma))                # gamma^n as n - 1 f32 products (dqn_cnn_update_replay)
                loss = c.update(s, a, r, s2, d, isw, float(gn), want_loss=True)
            torch.cuda.synchronize()
            out[mode] = (host(c.get_buffer("params")).copy(), loss, host(c.get_buffer("grad")).copy())
            c.close()
        df = np.flatnonzero(out["ring"][0] != out["rows"][0])
        assert df.size == 0, (precision, df.size, df[:8], float(np.abs(out["ring"][0] - out["rows"][0]).max()),
                              int((out["ring"][2] != out["rows"][2]).sum()), out["ring"][1], out["rows"][1])
        assert out["ring"][1] == out["rows"][1], precision
        assert np.isfinite(out["td"]).all() and out["td"].min() >= 0
        grads[precision] = out["ring"][2]
    # (c)
    o = 0
    for name, cnt in LEAVES:
        got, ref = grads["bf16"][o:o + cnt].astype(np.float64), grads["f32"][o:o + cnt].astype(np.float64)
        if name.endswith(".w"):
            assert np.sqrt(((got - ref) ** 2).mean()) <= 0.25 * np.sqrt((ref ** 2).mean()), name
            assert np.corrcoef(got, ref)[0, 1] >= 0.97, name
        o += cnt
    e.close()


def test_cnn_side_stream_ordering_across_calls(dq):
    """the handle's side stream (target pass, dW kernels, the fc leaf's optimizer step) is forked / joined by
    events inside each call; pinned here ACROSS calls on a non-default stream: two back-to-back dqn_cnn_update_replay +
    dqn_cnn_act, no host synchronisation in between, bit for bit against the same sequence with the side stream switched
    off (DQN_CNN_FLAG_NO_SIDE_STREAM: everything in stream order)."""
    import torch
    n, B = 16, 32
    res = {}
    for side in (True, False):
        c = dq.CnnEngine(num_actions=A, max_batch=B, precision="bf16")
        if not side:
            c.set_flags(dq._lib.CNN_FLAG_NO_SIDE_STREAM)
        c.set_params(make_params(71)); c.set_params(make_params(72), target=True)
        c.set_optimizer(lr=1e-3)
        c.replay_init(8 * n)
        r2 = np.random.default_rng(73)
        for t in range(6):
            c.replay_add(r2.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8), r2.integers(0, A, n), r2.standard_normal(n).astype(np.float32),
                         r2.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8), (r2.random(n) < 0.1).astype(np.float32))
        obs = torch.as_tensor(r2.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8)).cuda()
        idx1 = torch.as_tensor(np.sort(r2.integers(0, 6 * n, B)).astype(np.int32)).cuda()
        idx2 = torch.as_tensor(np.sort(r2.integers(0, 6 * n, B)).astype(np.int32)).cuda()
        isw = torch.as_tensor(r2.uniform(0.3, 1, B).astype(np.float32)).cuda()
        td1 = torch.empty(B, device="cuda"); td2 = torch.empty(B, device="cuda")
        acts = []
        st = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            c.update_from_replay(idx1, isw, 0.99, td_abs_out=td1)
            acts.append(c.act(obs, 0.0, seed=1, ctr=0))
            c.update_from_replay(idx2, isw, 0.99, td_abs_out=td2)
            acts.append(c.act(obs, 0.0, seed=1, ctr=1))
            st.synchronize()
        res[side] = (host(c.get_buffer("params")).copy(), host(td1).copy(), host(td2).copy(), host(acts[0]).copy(), host(acts[1]).copy())
        c.close()
    for x, y in zip(res[True], res[False]):
        assert np.array_equal(x, y)


def test_cnn_comm_world1(dq):
    """dqn_cnn_comm_init / the data-parallel flow of dqn_cnn_update at world 1 -- what is true there: the communicator exists
    (ncclCommCount = 1), the update takes the data-parallel route (backward without the early fc-leaf step, all-reduce calls,
    optimizer with grad_scale = 1 / 1) and leaves exactly the parameters and loss of the single-learner route; the stand-alone
    dqn_cnn_allreduce_grads is the identity on the gradient buffer. Nothing is claimed about world > 1 (no multi-GPU box)."""
    B = 8
    rng = np.random.default_rng(81)
    P, Pt = make_params(81), make_params(82)
    s = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8); s2 = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
    a = rng.integers(0, A, B).astype(np.int32); r = rng.standard_normal(B).astype(np.float32); d = (rng.random(B) < 0.3).astype(np.float32)
    isw = rng.uniform(0.3, 1.0, B).astype(np.float32)
    out = {}
    for dp in (False, True):
        e = dq.CnnEngine(num_actions=A, max_batch=B, precision="f32")
        e.set_params(P); e.set_params(Pt, target=True); e.set_optimizer(lr=1e-3)
        if dp:
            e.comm_init_native()
            assert e.comm_ranks() == 1
        else:
            assert e.comm_ranks() == 0
        losses = [e.update(s, a, r, s2, d, isw, 0.99, want_loss=True) for _ in range(2)]
        out[dp] = (host(e.get_buffer("params")).copy(), losses)
        if dp:
            q = host(e.forward(s))
            tg = (q + 0.5).astype(np.float32)
            e.grads(s, tg, isw)
            g0 = host(e.get_buffer("grad")).copy()
            e.allreduce_grads()
            assert np.array_equal(host(e.get_buffer("grad")), g0)
        e.close()
    assert out[True][1] == out[False][1] and np.array_equal(out[True][0], out[False][0])


def test_cnn_env_step_synth(dq):
    """r03 dqn_cnn_env_step_synth: the synthetic frame-stack vector env on the device (act + transition + ring add in one call).
    Against a numpy model: every next-frame byte is the Philox4x32-10 draw (seed, step, 16-byte piece index, env stream) of the
    restatement's generator; s of a step's rows == s' of the previous step's rows (the envs live in the ring's own s' rows);
    at epsilon = 0 the stored actions are the arg-max of the CNN on those frames; rewards / dones in range; counters."""
    import torch
    n, cap, seed = 8, 64, 11
    e = dq.CnnEngine(num_actions=A, max_batch=16, precision="f32")
    e.set_params(make_params(91)); e.replay_init(cap)
    e.env_reset_synth(n, seed)
    firsts = [e.env_step_synth(0.0, 0.3) for _ in range(10)]              # 80 rows into 64: wraps
    assert firsts == [(t * n) % cap for t in range(10)] and e.replay_size() == (cap, 10 * n)
    pieces = n * (84 * 84 * 4 // 16)
    for t in (9, 8, 3):                                                   # rows still in the ring (steps 2 .. 9)
        idx = ((t * n) % cap + np.arange(n)).astype(np.int32)
        s, a, r, s2, d = (host(x) for x in e.replay_gather(idx))
        want = onp.philox_draw(seed, t, pieces, onp.STREAM_ENV).astype("<u4").view(np.uint8).reshape(n, 84, 84, 4)
        assert np.array_equal(s2, want), t
        prev = ((t - 1) * n) % cap + np.arange(n)
        if t - 1 >= 2:
            assert np.array_equal(s, host(e.replay_gather(prev.astype(np.int32))[3])), t     # s(t) = s'(t - 1)
        q = host(e.forward(torch.as_tensor(s)))
        top2 = np.sort(q, axis=1)
        clear = top2[:, -1] - top2[:, -2] > 1e-6
        assert np.array_equal(a[clear], q.argmax(1)[clear])
        assert np.isfinite(r).all() and np.abs(r).max() < 3.5 and set(np.unique(d)) <= {0.0, 1.0}
    allr = host(e.replay_gather(np.arange(cap, dtype=np.int32))[2]); alld = host(e.replay_gather(np.arange(cap, dtype=np.int32))[4])
    assert 0.05 < alld.mean() < 0.6 and allr.std() > 0.5
    e.close()
