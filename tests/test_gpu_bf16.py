"""GPU tests (-m gpu) of the bf16 MFMA precision mode (dqn_config.precision = DQN_PREC_BF16): bf16 operands,
f32 accumulation, f32 master weights. This is the throughput path; its tolerance is bf16's (2^-8 per operand),
stated per assert -- the 1e-5 parity bar belongs to the f32 path (test_gpu_parity.py)."""
import numpy as np
import pytest

import _oracle as oc
from _oracle import onp
from test_oracle import CFGS, make_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dq(torch_cuda):
    import deep_q_learning_amd as pkg
    return pkg


def mk(dq, dims, **kw):
    return dq.Engine(dq.EngineConfig(obs_dim=dims[0], hidden1=dims[1], hidden2=dims[2], num_actions=dims[3],
                                     precision="bf16", **kw))


def host(t):
    return t.detach().cpu().numpy()


def rand_params(dims, seed):
    P = onp.init_params(dims, seed)
    return (P + 0.05 * np.random.default_rng(seed + 100).standard_normal(P.size)).astype(np.float32)


@pytest.mark.parametrize("name,B", [("cfg1", 64), ("cfg1", 37), ("cfg2", 1024), ("cfg3", 2048), ("cfg2", 100)])
def test_bf16_forward_and_targets(dq, name, B):
    dims = CFGS[name]
    e = mk(dq, dims, max_batch=B)
    P, Pt = rand_params(dims, 0), rand_params(dims, 1)
    e.set_params(P); e.set_params(Pt, dq._lib.BUF_TARGET)
    s, a, r, s2, d = make_batch(dims, B, 2)
    r = np.clip(r, -3, 3)
    q = host(e.forward(s))
    q64 = onp.forward(P, s, dims, np.float64)
    scale = np.abs(q64).max()
    assert np.max(np.abs(q - q64)) <= 2e-2 * scale, np.max(np.abs(q - q64)) / scale      # bf16: ~2^-8 per operand
    assert np.array_equal(q, host(e.forward(s)))                                          # deterministic
    t = host(e.q_targets(s, a, r, s2, d))
    full = onp.q_targets(P, Pt, s, a, r, s2, d, 0.99, dims, np.float64, full=True)
    t64 = full["targets"]
    top2 = np.sort(full["next_q"], axis=1)
    clear = (top2[:, -1] - top2[:, -2]) > 4e-2 * scale             # rows whose double-Q argmax bf16 cannot flip
    assert clear.mean() > 0.5
    assert np.max(np.abs(t - t64)[clear]) <= 3e-2 * max(np.abs(t64).max(), 1.0)
    assert np.array_equal(e.get_params(host=True), P)                                     # master weights stay f32
    e.close()


@pytest.mark.parametrize("name,B", [("cfg1", 64), ("cfg2", 1024), ("cfg3", 2048), ("cfg1", 50)])
def test_bf16_grads(dq, name, B):
    dims = CFGS[name]
    e = mk(dq, dims, max_batch=B)
    P, Pt = rand_params(dims, 3), rand_params(dims, 4)
    e.set_params(P)
    s, a, r, s2, d = make_batch(dims, B, 5)
    r = np.clip(r, -3, 3)
    targets = onp.q_targets(P, Pt, s, a, r, s2, d, 0.99, dims, np.float64).astype(np.float32)
    isw = np.random.default_rng(6).uniform(0.2, 1, B).astype(np.float32)
    g64, L64, _ = onp.grads(P, s, targets, dims, isw, np.float64)
    g, L = e.grads(s, targets, isw)
    g = host(g)
    assert abs(host(L)[0] - L64) <= 3e-2 * max(1.0, abs(L64))
    # cosine similarity and scale of the whole gradient; elementwise within 5 % of the largest entry
    cos = float(g @ g64 / (np.linalg.norm(g) * np.linalg.norm(g64)))
    assert cos > 0.999, cos
    assert np.max(np.abs(g - g64)) <= 5e-2 * np.abs(g64).max()
    g2, _ = e.grads(s, targets, isw)
    assert np.array_equal(g, host(g2))                                                    # fixed reduction order
    e.close()


def test_bf16_training_reduces_loss_and_tracks_f32(dq):
    """300 fused updates on a fixed PER replay: the bf16 learner learns, and stays close to the f32 learner"""
    import torch
    dims = CFGS["cfg1"]
    B, L_ = 64, 10
    N = 1 << L_
    s, a, r, s2, d = make_batch(dims, N, 7, terminal_frac=0.05)
    r = np.clip(r, -1, 1)
    P0 = onp.init_params(dims, 8)
    out = {}
    for prec in ("bf16", "f32"):
        e = dq.Engine(dq.EngineConfig(obs_dim=dims[0], hidden1=dims[1], hidden2=dims[2], num_actions=dims[3], capacity=N,
                                      use_per=True, max_batch=B, seed=9, lr=2e-3, precision=prec))
        e.replay_add(s, a, r, s2, d > 0)
        e.set_params(P0); e.sync_target()
        losses = []
        with torch.cuda.stream(e.stream):
            for it in range(300):
                e.update(B)
                if it % 10 == 0:
                    e.stream.synchronize(); losses.append(float(e.last_loss().item()))
            e.stream.synchronize()
        out[prec] = (np.array(losses), e.get_params(host=True), host(e.buffer(dq._lib.BUF_TREE)))
        assert e.opt_count() == 300
        e.close()
    lb, pb, tb = out["bf16"]; lf, pf, tf = out["f32"]
    assert np.isfinite(lb).all() and lb[-5:].mean() < 0.7 * lb[:3].mean()
    assert abs(lb[-10:].mean() - lf[-10:].mean()) <= 0.5 * lf[-10:].mean() + 1e-3     # minibatch losses are noisy
    k = np.arange(1, N)
    assert np.array_equal(tb[k], tb[2 * k] + tb[2 * k + 1])                               # sum-tree invariant holds
    db, df = pb - P0, pf - P0                                                             # same direction of travel
    cos = float(db @ df / (np.linalg.norm(db) * np.linalg.norm(df)))
    print("bf16 vs f32 after 300 updates: cos", cos, "norm ratio", np.linalg.norm(db) / np.linalg.norm(df))
    assert cos > 0.8 and 0.7 < np.linalg.norm(db) / np.linalg.norm(df) < 1.4


def test_bf16_actor_loop_runs(dq):
    import torch
    dims = CFGS["cfg2"]
    e = mk(dq, dims, capacity=1 << 12, use_per=True, max_batch=1024, seed=3)
    e.set_params(rand_params(dims, 10)); e.sync_target()
    rng = np.random.default_rng(11)
    e.replay_add(rng.standard_normal((2048, 8)), rng.integers(0, 4, 2048), rng.standard_normal(2048),
                 rng.standard_normal((2048, 8)), rng.random(2048) < 0.05)
    e.env_reset(rng.standard_normal((256, 8)).astype(np.float32), 0.01); e.set_epsilon(0.15)
    with torch.cuda.stream(e.stream):
        for _ in range(5):
            e.train_iters(4, 4, 1024)
        e.stream.synchronize()
    assert e.opt_count() == 20 and e.replay_size()[1] == 2048 + 20 * 4 * 256
    assert np.isfinite(e.get_params(host=True)).all() and np.isfinite(float(e.last_loss().item()))
    acts = host(e.buffer(dq._lib.BUF_ENV_ACTIONS, torch.int32))[:256]
    assert acts.min() >= 0 and acts.max() < 4
    e.close()


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_fused_row_backward_equals_separate_launch(dq, precision):
    """the paths whose workgroups wait for each other inside a launch (PER batch drawn by the actor launch's sampler
    workgroups; row backward riding in the forward launch after the partners' Q-row hand-over) against the path without
    any in-launch wait (dqn_config.flags = DQN_FLAG_NO_HANDOVER: the library's fallback for devices that cannot hold the
    co-resident grids -- the update draws its own batch, k_bwd_rows / k_bwd_rows16 is its own launch): identical
    parameters, tree, sampled indices and loss after a captured loop, bit for bit, in both precision modes; no bounded
    wait gave up"""
    import torch
    dims = CFGS["cfg2"]
    out = {}
    for fused in (True, False):
        e = dq.Engine(dq.EngineConfig(obs_dim=dims[0], hidden1=dims[1], hidden2=dims[2], num_actions=dims[3], capacity=1 << 12,
                                      use_per=True, max_batch=1024, seed=3, precision=precision,
                                      flags=0 if fused else dq._lib.FLAG_NO_HANDOVER))
        e.set_params(rand_params(dims, 10)); e.sync_target()
        rng = np.random.default_rng(11)
        e.replay_add(rng.standard_normal((2048, 8)), rng.integers(0, 4, 2048), rng.standard_normal(2048),
                     rng.standard_normal((2048, 8)), rng.random(2048) < 0.05)
        e.env_reset(rng.standard_normal((256, 8)).astype(np.float32), 0.01); e.set_epsilon(0.15)
        with torch.cuda.stream(e.stream):
            for _ in range(3):
                e.train_iters(4, 4, 1024)
            e.stream.synchronize()
        out[fused] = (e.get_params(host=True), host(e.buffer(dq._lib.BUF_TREE)).copy(), float(e.last_loss().item()),
                      host(e.buffer(dq._lib.BUF_BATCH_IDX, torch.int32))[:1024].copy())
        assert e.device_errors() == 0
        e.close()
    assert np.array_equal(out[True][0], out[False][0])
    assert np.array_equal(out[True][1], out[False][1])
    assert np.array_equal(out[True][3], out[False][3])
    assert out[True][2] == out[False][2] and np.isfinite(out[True][2])


def test_bf16_actor_chain_agrees_with_the_bf16_forward(dq):
    """bf16 mode's k_actor (v_mfma_f32_4x4x4_16b_bf16 chains, weights / activations rounded to bf16 in registers) must act
    like the bf16 forward kernel: with epsilon = 0 the chosen action is the argmax of dqn_qnet_forward's Q, except where
    the two best actions are closer than the bf16 tolerance; and it must differ from the f32 master weights' Q by no more
    than that tolerance allows (i.e. it is not silently the f32 chain)"""
    import torch
    dims = CFGS["cfg2"]
    n = 256
    e = mk(dq, dims, capacity=1 << 12, use_per=True, max_batch=1024, seed=3)
    P0 = rand_params(dims, 10)
    e.set_params(P0); e.sync_target()
    obs = np.random.default_rng(5).standard_normal((n, 8)).astype(np.float32)
    q_api = host(e.forward(obs))                                   # bf16 forward kernel
    e.env_reset(obs, 0.01); e.set_epsilon(0.0)
    with torch.cuda.stream(e.stream):
        e.actor_step()
        e.stream.synchronize()
    acts = host(e.buffer(dq._lib.BUF_ENV_ACTIONS, torch.int32))[:n]
    top2 = np.sort(q_api, axis=1)[:, -2:]
    gap = top2[:, 1] - top2[:, 0]
    scale = np.abs(q_api).max()
    clear = gap > 2e-2 * scale
    assert clear.sum() > n // 2
    assert np.array_equal(acts[clear], q_api.argmax(1)[clear])
    assert np.array_equal(acts, host(e.buffer(dq._lib.BUF_ACTIONS, torch.int32))[:n])      # the ring got the same actions
    e.close()



@pytest.mark.parametrize("n_step,per", [(1, False), (3, False), (1, True)])
def test_bf16_actor_counters_advance_without_per(dq, n_step, per):
    """with uniform replay (no tree / sampler workgroups in the actor launch) the counter commit at the
    end of k_actor hangs on an ACTOR workgroup drawing the last arrival ticket. The bf16 / H1 > 128 instantiations once read
    a stale ticket there (an inline-asm atomic whose result register the compiler copied before it had arrived): ring and
    env counters never advanced. Ring counter, size and the policy stream's env counter must move by T*n / T per launch."""
    import torch
    dims = (8, 256, 256, 4)
    n, T, N = 256, 4, 1 << 14
    e = mk(dq, dims, capacity=N, use_per=per, max_batch=n, seed=3, n_step=n_step)
    e.set_params(rand_params(dims, 7))
    obs = np.random.default_rng(8).standard_normal((n, dims[0])).astype(np.float32)
    e.env_reset(obs, p_done=0.05); e.set_epsilon(0.2)
    warm = n_step - 1                                            # the first n_step - 1 vector steps only fill the history
    with torch.cuda.stream(e.stream):
        for k in range(1, 20):
            e.actor_steps(T)
            e.stream.synchronize()
            filed = (k * T - warm) * n
            assert e.replay_size() == (min(filed, N), filed), (k, e.replay_size())
    first = host(e.buffer(dq._lib.BUF_OBSERVATIONS).view(N, dims[0]))
    assert np.isfinite(first).all() and np.abs(first).max() > 0
    assert e.device_errors() == 0
    e.close()


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_bounded_wait_gives_up_and_the_handle_recovers(dq, precision):
    """the give-up path of the bounded in-kernel waits (csrc/dqn_device.h: wait_word_eq, 0.2 s of the
    100 MHz clock) driven on purpose. dqn_debug_withhold_handover makes the fused forward's partner passes not count
    themselves in: the update must RETURN (no hang), every tile's pass-0 workgroup gives up once (error count = tiles), the
    loss is NaN. After dqn_clear_device_errors + a parameter reload the same handle runs the normal path again: no errors,
    finite losses."""
    import time
    import torch
    dims = CFGS["cfg2"]
    B, N = 1024, 1 << 12
    rng = np.random.default_rng(21)
    rows = (rng.standard_normal((2048, 8)), rng.integers(0, 4, 2048), rng.standard_normal(2048), rng.standard_normal((2048, 8)), rng.random(2048) < 0.05)
    P0 = rand_params(dims, 22)

    def fresh():
        # uniform replay: the row backward rides in the forward launch (with PER only behind the actor launch's presampling)
        e = dq.Engine(dq.EngineConfig(obs_dim=dims[0], hidden1=dims[1], hidden2=dims[2], num_actions=dims[3], capacity=N, use_per=False,
                                      max_batch=B, seed=9, precision=precision))
        e.set_params(P0); e.sync_target(); e.replay_add(*rows)
        return e
    e = fresh()
    with torch.cuda.stream(e.stream):
        e.update(B); e.stream.synchronize()                       # the normal path first (also: graphs exist and must be dropped)
        assert e.device_errors() == 0 and np.isfinite(float(e.last_loss().item()))
        e.debug_withhold_handover(True)
        t0 = time.time()
        e.update(B); e.stream.synchronize()
        dt = time.time() - t0
    assert 0.15 < dt < 5.0, dt                                    # sat out the 0.2 s bound once (tiles wait side by side), did not hang
    assert e.device_errors() == (B + 15) // 16
    assert np.isnan(float(e.last_loss().item()))
    e.debug_withhold_handover(False)
    e.clear_device_errors()
    assert e.device_errors() == 0
    # the next normal launches on the same handle are clean (its parameters carry the timed-out update: only health is
    # compared) -- hand-over counters in step again, graphs re-captured
    e.set_params(P0); e.sync_target()
    with torch.cuda.stream(e.stream):
        for _ in range(3):
            e.update(B)
        e.stream.synchronize()
    assert e.device_errors() == 0 and np.isfinite(float(e.last_loss().item()))
    e.close()


# ---------------------------------------------------------------- r03: the 64-row bf16 kernels (dqn_net_big16.hip)
def mk_big(dq, dims, **kw):
    return dq.Engine(dq.EngineConfig(obs_dim=dims[0], hidden1=dims[1], hidden2=dims[2], num_actions=dims[3],
                                     precision="bf16", flags=dq._lib.FLAG_BIG_ROWS, **kw))


@pytest.mark.parametrize("B,D", [(64, 8), (200, 8), (1000, 9), (130, 20)])
def test_bf16_big_rows_forward_and_targets(dq, B, D):
    """k_big_rows16 (v_mfma_f32_32x32x16_bf16, 64 rows per workgroup) forced at small sizes (DQN_FLAG_BIG_ROWS), ragged last
    tile, both layer-1 forms (D <= 16: one k-step; D = 20: two): Q, features and compute_q_targets within bf16's 2e-2 / 3e-2 of
    scale of f64, deterministic, and within 1e-2 of the 16-row bf16 kernels (same roundings, another accumulation order)."""
    dims = (D, 256, 256, 4)
    e = mk_big(dq, dims, max_batch=B)
    e16 = mk(dq, dims, max_batch=B)
    P, Pt = rand_params(dims, 0), rand_params(dims, 1)
    for x in (e, e16):
        x.set_params(P); x.set_params(Pt, dq._lib.BUF_TARGET)
    s, a, r, s2, d = make_batch(dims, B, 2)
    r = np.clip(r, -3, 3)
    q, feat = e.forward(s, return_features=True)
    q, feat = host(q), host(feat)
    q64, _, h64 = onp.forward(P, s, dims, np.float64, return_hidden=True)
    scale = np.abs(q64).max()
    assert np.max(np.abs(q - q64)) <= 2e-2 * scale, np.max(np.abs(q - q64)) / scale
    assert np.max(np.abs(feat - h64)) <= 2e-2 * np.abs(h64).max()
    assert np.array_equal(q, host(e.forward(s)))
    q16 = host(e16.forward(s))
    assert np.max(np.abs(q - q16)) <= 1e-2 * scale, np.max(np.abs(q - q16)) / scale
    qt = host(e.forward(s, target=True))
    assert np.max(np.abs(qt - onp.forward(Pt, s, dims, np.float64))) <= 2e-2 * scale
    t = host(e.q_targets(s, a, r, s2, d))
    full = onp.q_targets(P, Pt, s, a, r, s2, d, 0.99, dims, np.float64, full=True)
    top2 = np.sort(full["next_q"], axis=1)
    clear = (top2[:, -1] - top2[:, -2]) > 4e-2 * scale
    assert clear.mean() > 0.5
    assert np.max(np.abs(t - full["targets"])[clear]) <= 3e-2 * max(np.abs(full["targets"]).max(), 1.0)
    assert np.array_equal(e.get_params(host=True), P)
    e.close(); e16.close()


@pytest.mark.parametrize("B,weighted", [(64, False), (100, True), (1000, True), (1000, False)])
def test_bf16_big_rows_grads(dq, B, weighted):
    """jax.grad(compute_loss) through k_big_rows16 + k_big_dw16 (k-packed bf16 stashes, split-K slab) + k_big_reduce<true>:
    against f64 (cosine > 0.999, 5e-2 of the largest entry -- the bars of test_bf16_grads), every leaf against the 16-row bf16
    kernels' gradient (2e-2 of the leaf's scale), fixed reduction order."""
    dims = CFGS["cfg2"]
    e = mk_big(dq, dims, max_batch=B)
    e16 = mk(dq, dims, max_batch=B)
    P, Pt = rand_params(dims, 3), rand_params(dims, 4)
    e.set_params(P); e16.set_params(P)
    s, a, r, s2, d = make_batch(dims, B, 5)
    r = np.clip(r, -3, 3)
    targets = onp.q_targets(P, Pt, s, a, r, s2, d, 0.99, dims, np.float64).astype(np.float32)
    isw = np.random.default_rng(6).uniform(0.2, 1, B).astype(np.float32) if weighted else None
    g64, L64, _ = onp.grads(P, s, targets, dims, isw, np.float64)
    g, L = e.grads(s, targets, isw)
    g = host(g)
    assert abs(host(L)[0] - L64) <= 3e-2 * max(1.0, abs(L64))
    cos = float(g @ g64 / (np.linalg.norm(g) * np.linalg.norm(g64)))
    assert cos > 0.999, cos
    assert np.max(np.abs(g - g64)) <= 5e-2 * np.abs(g64).max()
    g2, _ = e.grads(s, targets, isw)
    assert np.array_equal(g, host(g2))
    g16 = host(e16.grads(s, targets, isw)[0])
    o = 0
    from deep_q_learning_amd._tree import shapes
    for mod, leaf, shp in shapes(dims):
        n = int(np.prod(shp))
        sc = np.abs(g64[o:o + n]).max()
        assert np.max(np.abs(g[o:o + n] - g16[o:o + n])) <= 2e-2 * sc, (mod, leaf, np.max(np.abs(g[o:o + n] - g16[o:o + n])) / sc)
        # per leaf against f64: no further from it than the 16-row bf16 kernels are (layer 1's leaf carries the roundings of two
        # backward layers: ~0.1 of its scale on both paths)
        e_big, e_16 = np.max(np.abs(g[o:o + n] - g64[o:o + n])), np.max(np.abs(g16[o:o + n] - g64[o:o + n]))
        assert e_big <= 1.5 * e_16 + 1e-2 * sc, (mod, leaf, e_big / sc, e_16 / sc)
        o += n
    e.close(); e16.close()


@pytest.mark.parametrize("per", [True, False])
def test_bf16_big_rows_update_tracks_the_16_row_path(dq, per):
    """Agent._step (q_agent.py:146-169) through the 64-row bf16 kernels (sample -> k_big_rows16 with the three forwards, TD rule
    and row backward -> k_big_dw16 -> reduce + AdamW + bf16 shadow refresh -> priority write-back), three updates at 1 000
    rows, against the 16-row bf16 path on the same replay: losses within 2e-2, parameter steps with cosine > 0.99, the shadows
    really refreshed (a forward after the updates uses the new weights), tree invariant, no wait gave up."""
    import torch
    dims = CFGS["cfg2"]
    B, N = 1000, 1 << 12
    s, a, r, s2, d = make_batch(dims, 3000, 70, terminal_frac=0.1)
    r = np.clip(r, -2, 2)
    P0 = rand_params(dims, 71)
    out = {}
    for name, fl in (("big", dq._lib.FLAG_BIG_ROWS), ("rows16", 0)):
        e = dq.Engine(dq.EngineConfig(obs_dim=dims[0], hidden1=dims[1], hidden2=dims[2], num_actions=dims[3], capacity=N, use_per=per,
                                      max_batch=B, seed=77, lr=1e-3, precision="bf16", flags=fl))
        e.replay_add(s, a, r, s2, d > 0)
        e.set_params(P0); e.set_params(P0, dq._lib.BUF_TARGET)
        losses = []
        with torch.cuda.stream(e.stream):
            for it in range(3):
                e.update(B); e.stream.synchronize()
                losses.append(float(e.last_loss().item()))
        Pn = e.get_params(host=True)
        x = s[:256]
        q_after = host(e.forward(x))
        q64 = onp.forward(Pn, x, dims, np.float64)
        assert np.max(np.abs(q_after - q64)) <= 2e-2 * np.abs(q64).max()               # shadows = the updated master weights
        assert e.opt_count() == 3 and e.device_errors() == 0
        if per:
            t = host(e.buffer(dq._lib.BUF_TREE)); k = np.arange(1, N)
            assert np.array_equal(t[k], t[2 * k] + t[2 * k + 1])
        out[name] = (np.array(losses), Pn)
        e.close()
    lb, pb = out["big"]; l16, p16 = out["rows16"]
    assert np.isfinite(lb).all() and np.all(np.abs(lb - l16) <= 2e-2 * np.maximum(1.0, np.abs(l16))), (lb, l16)
    db, d16 = pb - P0, p16 - P0
    cos = float(db @ d16 / (np.linalg.norm(db) * np.linalg.norm(d16)))
    assert cos > 0.99 and 0.9 < np.linalg.norm(db) / np.linalg.norm(d16) < 1.1, (cos, np.linalg.norm(db) / np.linalg.norm(d16))
