"""GPU tests (-m gpu) of the reference-named surface: the same calls a user of
hal9000universe/deep-q-learning makes (Test/lunar_lander.py:39-78, q_agent.py:146-169), checked against the
oracle and the committed golden vectors."""
import os

import numpy as np
import pytest

import _oracle as oc
from _oracle import onp
from test_oracle import CFGS, make_batch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ref(torch_cuda):
    import deep_q_learning_amd as dq
    from deep_q_learning_amd import optim
    from deep_q_learning_amd.General.Base.replay_buffer import ReplayBuffer, sample_batch
    from deep_q_learning_amd.General.Base.utils import generate_loading, generate_saving
    from deep_q_learning_amd.General.QLearning import q_learning_functions as qf
    from deep_q_learning_amd.General.QLearning.q_agent import Agent
    from deep_q_learning_amd.LunarLander.dddqn import Model, transform, without_apply_rng
    from deep_q_learning_amd.LunarLander.env import ObsWrapper
    from deep_q_learning_amd._tree import unflatten
    return dict(dq=dq, optim=optim, ReplayBuffer=ReplayBuffer, sample_batch=sample_batch, qf=qf, Agent=Agent,
                Model=Model, transform=transform, without_apply_rng=without_apply_rng, ObsWrapper=ObsWrapper,
                unflatten=unflatten, generate_saving=generate_saving, generate_loading=generate_loading,
                torch=torch_cuda)


@pytest.mark.parametrize("fn", sorted(f for f in os.listdir(GOLD) if f.startswith("cfg")))
def test_factories_against_golden(ref, fn):
    """compute_q_targets / compute_loss / train_step (q_learning_functions.py:14-64) on the golden inputs"""
    torch = ref["torch"]
    z = np.load(os.path.join(GOLD, fn), allow_pickle=False)
    dims = tuple(int(x) for x in z["dims"])
    k = int(z["sub"])
    model = ref["Model"](dims[3], hidden=(dims[1], dims[2])).transformed()
    dev = ref["dq"].engine.default_device()
    params = ref["unflatten"](torch.tensor(z["P"], device=dev), dims)
    target = ref["unflatten"](torch.tensor(z["Pt"], device=dev), dims)

    class Env:
        class action_space:
            n = dims[3]
    qf = ref["qf"]
    s, a, r, s2, d = qf.preprocessing(z["s"], z["a"], z["r"], z["s2"], z["d"] > 0)             # :76-85
    assert d.dtype == torch.float32 and a.dtype == torch.int32
    q = model.apply(params, s)
    assert np.allclose(host(q), z["q"], rtol=1e-5, atol=1e-5)
    targets = qf.generate_q_target_comp(model, 0.99, Env)(params, target, s, a, r, s2, d)
    assert np.allclose(host(targets), z["targets"], rtol=1e-5, atol=1e-4)
    loss = qf.generate_loss_computation(model)(params, s, targets)
    assert abs(float(loss) - float(z["loss"])) <= 1e-5 * max(1.0, abs(float(z["loss"])))
    greedy = qf.action_computation(model)(params, s[:1])
    assert int(greedy) == int(np.argmax(z["q"][0]))
    for kind, lr in (("adamw", 2e-4), ("adam", 1e-4)):
        opt = getattr(ref["optim"], kind)(lr)
        train_step = qf.generate_train_step(opt, model)
        p, st = params, opt.init(params)
        comp = qf.generate_q_target_comp(model, 0.99, Env)
        for it in range(3):
            tg = comp(p, target, s, a, r, s2, d)
            p, st = train_step(p, st, s, tg)
            if it == 0:
                assert np.allclose(host(p.flat)[::k], z[f"{kind}_params_1"], rtol=1e-5, atol=1e-6)
        assert st[0].count == 3 and len(st) == (3 if kind == "adamw" else 2)
        assert np.allclose(host(p.flat)[::k], z[f"{kind}_params_3"], rtol=1e-5, atol=1e-6)
        for got, want in ((st[0].mu, z[f"{kind}_mu_3"]), (st[0].nu, z[f"{kind}_nu_3"])):    # 1e-5 of the leaf scale
            assert np.max(np.abs(host(got.flat)[::k] - want)) <= 1e-5 * np.abs(want).max()
        assert abs(float(host(p.flat).astype(np.float64).sum()) - float(z[f"{kind}_params_3_sum"])) < 1e-3
    # functional style: the inputs were not modified
    assert np.array_equal(host(params.flat), z["P"])


@pytest.mark.parametrize("kind,lr", [("adamw", 2e-4), ("adam", 1e-4)])
def test_optimizer_update_and_apply_updates(ref, kind, lr):
    """`updates, opt_state = optimizer.update(grads, opt_state, params)` + `optax.apply_updates` (q_learning_functions.py:
    24-25) as separate calls: two consecutive steps bit-identical to the C oracle's Adam / AdamW on the same gradients"""
    torch = ref["torch"]
    dims = CFGS["cfg1"]
    optim, unflatten = ref["optim"], ref["unflatten"]
    opt = getattr(optim, kind)(lr)
    rng = np.random.default_rng(4)
    P = (onp.init_params(dims, 2) + 0.05 * rng.standard_normal(onp.param_count(*dims))).astype(np.float32)
    params = unflatten(torch.tensor(P, device="cuda"), dims)
    state = opt.init(params)
    copt = oc.Opt(lr, 0.9, 0.999, 1e-8, 1e-4, int(kind == "adamw"))
    Pc, muc, nuc, cnt, p1, p2 = P.copy(), np.zeros_like(P), np.zeros_like(P), 0, 1.0, 1.0
    for it in range(2):
        g = (rng.standard_normal(P.size) * 0.01).astype(np.float32)
        grads = unflatten(torch.tensor(g, device="cuda"), dims)
        updates, state = opt.update(grads, state, params)
        new_params = optim.apply_updates(params, updates)
        Pc2, muc, nuc, cnt, p1, p2 = oc.adam_step(copt, Pc, g, muc, nuc, cnt, p1, p2)
        assert np.array_equal(host(new_params.flat), Pc2)
        assert np.array_equal(host(state[0].mu.flat), muc) and np.array_equal(host(state[0].nu.flat), nuc) and state[0].count == it + 1
        got_u = np.concatenate([host(updates[m][l]).reshape(-1) for m, l, _ in __import__("deep_q_learning_amd")._tree.shapes(dims)])
        assert np.allclose(got_u, Pc2 - Pc, rtol=0, atol=1e-9)               # the updates pytree itself: new - old
        params, Pc = new_params, Pc2
    assert len(state) == (3 if kind == "adamw" else 2)


@pytest.mark.parametrize("fn", ["per_L12.npz", "per_L16.npz"])
def test_per_against_golden(ref, fn):
    z = np.load(os.path.join(GOLD, fn), allow_pickle=False)
    L_, n, B, seed = int(z["L"]), int(z["n_add"]), int(z["B"]), int(z["seed"])
    dq = ref["dq"]
    e = dq.Engine(dq.EngineConfig(obs_dim=4, hidden1=16, hidden2=16, num_actions=2, capacity=1 << L_, use_per=True,
                                  max_batch=max(B, 4096)))
    rng = np.random.default_rng(0)
    for k0 in range(0, n, 4096):
        m = min(4096, n - k0)
        e.replay_add(rng.standard_normal((m, 4)), rng.integers(0, 2, m), rng.standard_normal(m), rng.standard_normal((m, 4)), np.zeros(m))
        e.per_set(np.arange(k0, k0 + m, dtype=np.int32), z["prio"][k0:k0 + m])
    for it in range(3):
        _, idx, isw = e.per_sample(B, 0.4 + 0.2 * it, seed=seed, ctr=it)
        assert np.array_equal(host(idx), z[f"idx_{it}"])
        assert np.array_equal(host(isw).view(np.uint32), z[f"isw_{it}"].view(np.uint32))
        (e.per_update_sorted if it % 2 else e.per_update)(z[f"idx_{it}"], z[f"td_{it}"])
        tree = host(e.buffer(dq._lib.BUF_TREE))
        assert tree[1] == z[f"total_{it}"]
        assert tree[1 << L_:].astype(np.float64).sum() == z[f"leafsum_{it}"]
    assert np.array_equal(tree[:64], z["tree_top"])
    e.close()


def test_replay_buffer_surface(ref):
    """ReplayBuffer(buffer_size, obs_shape, ac_shape), .add, properties, sample_batch (replay_buffer.py:20-85)"""
    torch = ref["torch"]
    N, D = 500, 9
    rb = ref["ReplayBuffer"](N, (N, D), (N,))
    cr = oc.CReplay(N, D)
    rng = np.random.default_rng(1)
    for i in range(40):                                               # single transitions, like q_agent.py:182
        s, s2 = rng.standard_normal(D).astype(np.float32), rng.standard_normal(D).astype(np.float32)
        a, r, d = int(rng.integers(0, 4)), float(rng.standard_normal()), bool(rng.random() < 0.2)
        rb.add(s, a, r, s2, d); cr.add(s[None], [a], [r], s2[None], [d])
    for _ in range(2):                                                # vectorised adds; the second one wraps
        s = rng.standard_normal((300, D)).astype(np.float32)
        rb.add(s, rng.integers(0, 4, 300), rng.standard_normal(300), s[::-1].copy(), rng.random(300) < 0.1)
        cr.add(s, np.zeros(300, np.int32), np.zeros(300), s[::-1].copy(), np.zeros(300))
    with pytest.raises(ref["dq"]._lib.DqnError):
        rb.add(np.zeros((N + 1, D), np.float32), np.zeros(N + 1), np.zeros(N + 1), np.zeros((N + 1, D), np.float32), np.zeros(N + 1))
    assert rb.size == cr.size == N and rb.states.shape == (N, D)
    assert np.array_equal(host(rb.states), cr.arrays()[0]) and np.array_equal(host(rb.observations), cr.arrays()[3])
    idx = rng.integers(0, N, 64).astype(np.int32)
    batch = ref["sample_batch"](rb.size, rb.states, rb.actions, rb.rewards, rb.observations, rb.dones, 64, indices=idx)
    assert np.array_equal(host(batch[0]), cr.arrays()[0][idx]) and batch[4].dtype == torch.uint8
    batch = ref["sample_batch"](rb.size, rb.states, rb.actions, rb.rewards, rb.observations, rb.dones, 64)
    assert np.array_equal(host(rb.last_indices), oc.uniform_indices(N, 64, 0, 1))
    with pytest.raises(TypeError):
        ref["sample_batch"](10, torch.zeros(10, D), None, None, None, None, 4)


class ToyEnv:
    """8-d observation, 4 actions, reward favours action == argmax of the first 4 observation entries"""

    class action_space:
        n = 4

    class observation_space:
        shape = (8,)

    def __init__(self, seed=0):
        self.rng = np.random.default_rng(seed); self.t = 0

    def reset(self):
        self.t = 0; self.o = self.rng.standard_normal(8)
        return self.o

    def step(self, a):
        self.t += 1
        r = 1.0 if a == int(np.argmax(self.o[:4])) else -0.2
        self.o = self.rng.standard_normal(8)
        return self.o, r, self.t >= 20, {}


def test_agent_training_loop_and_checkpoint(ref, tmp_path):
    """Agent(...23 kwargs...).training() as in Test/lunar_lander.py:53-78, on a toy gym-style env"""
    torch = ref["torch"]
    env = ref["ObsWrapper"](ToyEnv(), 20)
    model = ref["without_apply_rng"](ref["transform"](lambda *args: ref["Model"](4)(*args)))
    optimizer = ref["optim"].adamw(2e-3)
    params = model.init(0, env.reset())
    opt_state = optimizer.init(params)
    p0 = host(params.flat).copy()
    agent = ref["Agent"](network=model, params=params, optimizer=optimizer, opt_state=opt_state, env=env,
                         buffer_size=2000, obs_shape=(2000, 9), ac_shape=(2000,), gamma=0.9, epsilon=1.0,
                         epsilon_decay_rate=0.9, min_epsilon=0.1, max_episodes=30, max_steps=20, training_start=64,
                         batch_size=64, train_frequency=4, back_up_frequency=10, replace_frequency=5,
                         reward_to_reach=1e9, num_actions=4, saving_directory=str(tmp_path / "ckpt"), verbose=0)
    agent.training()
    assert agent.updates == (30 * 20) // 4 - 64 // 4 + 1
    assert agent._opt_state[0].count == agent.updates
    p1 = host(agent._params.flat)
    assert np.isfinite(p1).all() and np.abs(p1 - p0).max() > 1e-4
    assert abs(agent._epsilon - max(0.9 ** 30, 0.1)) < 1e-12 and len(agent._reward_history) == 30
    # learnt something: greedy policy beats chance (0.25) on the toy task
    o = np.random.default_rng(5).standard_normal((512, 8)).astype(np.float32)
    x = np.concatenate([o, np.full((512, 1), 0.5, np.float32)], axis=1)
    acts = host(model.apply(agent._params, x).argmax(dim=1))
    assert (acts == o[:, :4].argmax(axis=1)).mean() > 0.4
    # checkpoint written at episode 0/10/20 (q_agent.py:195-196): data-only npz, round trip
    params2, st2 = ref["generate_loading"](str(tmp_path / "ckpt"))()
    assert set(params2) == set(params) and st2[0].count <= agent.updates and len(st2) == 3
    ref["generate_saving"](str(tmp_path / "ckpt2"))(agent._params, agent._opt_state)
    params3, st3 = ref["generate_loading"](str(tmp_path / "ckpt2"))()
    assert torch.equal(params3.flat, agent._params.flat) and st3[0].count == agent.updates
    assert torch.equal(st3[0].nu.flat, agent._opt_state[0].nu.flat)


def test_bench_prints_one_json_line_with_the_contract_fields(torch_cuda):
    """bench.py (short run): stdout is exactly ONE JSON line carrying the driver's contract fields, `roofline` and
    `cpu_baseline`; native libraries' chatter must not reach stdout"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DQN_BENCH_CPU_SECONDS="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "60", "--warmup", "20", "--profile-steps", "2",
                          "--no-secondary"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 60 and d["warmup"] == 20 and d["n_gpus"] == 1 and d["value"] > 1000 and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] in ("GB/s", "TFLOP/s")
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c

