"""GPU tests (-m gpu) of the device-resident CartPole-v1 env (BASELINE.json configs[2]) and the vector control
loop: the physics + bookkeeping are bit-exact against the numpy restatement, and the whole
loop learns (mean episode length rises far above a random policy's ~22 steps)."""
import numpy as np
import pytest

import _oracle as oc
from _oracle import onp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dq(torch_cuda):
    import deep_q_learning_amd as pkg
    return pkg


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("steps_per_launch", [1, 4])
def test_cartpole_env_bitexact(dq, steps_per_launch):
    import torch
    n, T, N, seed, max_steps = 64, 300, 1 << 15, 5, 40
    e = dq.Engine(dq.EngineConfig(obs_dim=4, hidden1=64, hidden2=64, num_actions=2, capacity=N, use_per=True,
                                  max_batch=64, seed=seed))
    e.set_params(onp.init_params((4, 64, 64, 2), 0)); e.sync_target()
    e.env_config("cartpole", max_steps, -1.0)
    obs = (np.random.default_rng(1).random((n, 4)).astype(np.float32) * np.float32(0.1) - np.float32(0.05)).astype(np.float32)
    e.env_reset(obs); e.set_epsilon(1.0)                       # epsilon = 1: every action is the Philox randint
    with torch.cuda.stream(e.stream):
        for _ in range(T // steps_per_launch):
            if steps_per_launch == 1:
                e.actor_step()
            else:
                e.actor_steps(steps_per_launch)                # k_actor: env state stays in LDS between the steps
        e.stream.synchronize()
    # CPU restatement of the same T vector steps
    rb = onp.ReplayRing(N, 4); tree = onp.SumTree(15)
    s = obs.copy(); t = np.zeros(n, np.int32); episodes = 0; ep_steps = 0
    for c in range(T):
        o = onp.philox_draw(seed, c, n, onp.STREAM_POLICY)
        a = ((o[:, 1].astype(np.uint64) * np.uint64(2)) >> np.uint64(32)).astype(np.int32)    # q_agent.py:141
        s2, term = onp.cartpole_step(s, a)
        t = t + 1
        done = term | (t >= max_steps)
        tree.add(rb.add(s, a, np.where(term, np.float32(-1.0), np.float32(1.0)).astype(np.float32), s2, done))
        episodes += int(done.sum()); ep_steps += int(t[done].sum())
        fresh = onp.cartpole_reset_states(n, seed, c)
        s = np.where(done[:, None], fresh, s2).astype(np.float32)
        t = np.where(done, 0, t)
    assert e.env_stats() == (episodes, ep_steps) and episodes > 100
    assert np.array_equal(host(e.buffer(dq._lib.BUF_ENV_OBS))[: n * 4].reshape(n, 4), s)
    L = dq._lib
    for got, want in ((e.buffer(L.BUF_STATES).view(N, 4), rb.states), (e.buffer(L.BUF_OBSERVATIONS).view(N, 4), rb.observations),
                      (e.buffer(L.BUF_ACTIONS, torch.int32), rb.actions), (e.buffer(L.BUF_REWARDS), rb.rewards),
                      (e.buffer(L.BUF_DONES, torch.uint8), rb.dones)):
        assert np.array_equal(host(got), want)
    assert np.array_equal(host(e.buffer(L.BUF_TREE)), tree.tree)
    assert 15 < ep_steps / episodes < 40                       # random policy: ~22 steps per episode
    e.close()


def test_cartpole_nstep_rows_follow_from_the_one_step_stream(dq):
    """3-step returns on the CartPole env: an engine with n_step = 3 must store exactly the rows that the numpy window rule
    (R = r_u + g*(r_{u+1} + g*r_{u+2}) cut after the first done) derives from the one-step stream
    of an identical engine with n_step = 1 (same seed, fixed parameters => same actions and transitions)."""
    import torch
    n, T, N, seed, ns, g = 32, 120, 1 << 13, 9, 3, np.float32(0.99)
    rings = {}
    for n_step in (1, ns):
        e = dq.Engine(dq.EngineConfig(obs_dim=4, hidden1=64, hidden2=64, num_actions=2, capacity=N, use_per=True,
                                      max_batch=64, seed=seed, n_step=n_step))
        e.set_params(onp.init_params((4, 64, 64, 2), 0)); e.sync_target()
        e.env_config("cartpole", 25, -1.0)
        obs = (np.random.default_rng(1).random((n, 4)).astype(np.float32) * np.float32(0.1) - np.float32(0.05)).astype(np.float32)
        e.env_reset(obs); e.set_epsilon(0.5)
        with torch.cuda.stream(e.stream):
            for _ in range(T // 4):
                e.actor_steps(4)
            e.stream.synchronize()
        L = dq._lib
        rings[n_step] = [host(x).copy() for x in (e.buffer(L.BUF_STATES).view(N, 4), e.buffer(L.BUF_ACTIONS, torch.int32), e.buffer(L.BUF_REWARDS),
                                                   e.buffer(L.BUF_OBSERVATIONS).view(N, 4), e.buffer(L.BUF_DONES, torch.uint8))]
        rings[n_step].append(e.replay_size())
        rings[n_step].append(host(e.buffer(L.BUF_TREE)).copy())
        e.close()
    s1, a1, r1, o1, d1, (size1, _), _ = rings[1]
    s3, a3, r3, o3, d3, (size3, _), tree3 = rings[ns]
    assert size1 == T * n and size3 == (T - ns + 1) * n
    s1, a1, r1, o1, d1 = (x[: T * n].reshape((T, n) + x.shape[1:]) for x in (s1, a1, r1, o1, d1))
    for u in range(T - ns + 1):                                  # window u .. u+ns-1 -> row block u of the n-step ring
        R = np.empty(n, np.float32); dn = np.empty(n, np.uint8)
        for i in range(n):
            last = ns - 1
            for k in range(ns):
                if d1[u + k, i]:
                    last = k; break
            acc = r1[u + last, i]
            for k in range(last - 1, -1, -1):
                acc = np.float32(r1[u + k, i] + np.float32(g * acc))
            R[i] = acc; dn[i] = d1[u + last, i]
        blk = slice(u * n, (u + 1) * n)
        assert np.array_equal(s3[blk], s1[u]) and np.array_equal(a3[blk], a1[u])
        assert np.array_equal(o3[blk], o1[u + ns - 1])
        assert np.array_equal(r3[blk], R) and np.array_equal(d3[blk], dn)
    k = np.arange(1, N)
    assert np.array_equal(tree3[k], tree3[2 * k] + tree3[2 * k + 1])
    assert d1.sum() > 50                                          # episodes did end inside windows


@pytest.mark.parametrize("precision,n_step", [("f32", 1), ("bf16", 1), ("f32", 3)])
def test_cartpole_learns(dq, precision, n_step):
    """1024-env CartPole, 2x64 dueling MLP, PER (configs[2] shape, smaller batch): the vector loop must lift the mean
    episode length far above the random policy's ~22 steps. The terminating step is rewarded -1: with gym's +1 the
    reference's own target rule (terminal target = q + r, q_learning_functions.py:58) rewards falling (see the last
    row of tools/cartpole_demo.py --sweep)."""
    from deep_q_learning_amd.General.QLearning.vector_agent import VectorAgent
    from deep_q_learning_amd.LunarLander.dddqn import Model
    n_envs, B = 1024, 512
    e = dq.Engine(dq.EngineConfig(obs_dim=4, hidden1=64, hidden2=64, num_actions=2, capacity=1 << 18, use_per=True,
                                  max_batch=max(n_envs, B), seed=3, lr=5e-4, gamma=0.95, precision=precision, n_step=n_step))
    e.load(Model(2, hidden=(64, 64)).transformed().init(3, np.zeros((1, 4), np.float32)))
    agent = VectorAgent(e, n_envs, B, env="cartpole", max_steps=500, term_reward=-1.0, epsilon=1.0, epsilon_decay_rate=0.99,
                        min_epsilon=0.05, train_frequency=1, replace_frequency=5, reward_to_reach=150.0, chunk=20)
    hist = agent.training(max_updates=60000)
    rets = np.array([h[2] for h in hist if np.isfinite(h[2])])
    assert rets[:5].mean() < 40                                 # starts near the random policy
    assert rets.max() > 100, (rets.max(), rets[-5:])            # learns: > 4x the random policy's episode length
    assert np.isfinite(e.get_params(host=True)).all()
    e.close()
