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


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_cartpole_learns(dq, precision):
    """1024-env CartPole, 2x64 dueling MLP, PER (configs[2] shape, smaller batch): the vector loop must lift the mean
    episode length far above the random policy's ~22 steps. The terminating step is rewarded -1: with gym's +1 the
    reference's own target rule (terminal target = q + r, q_learning_functions.py:58) rewards falling (see the last
    row of tools/cartpole_demo.py --sweep)."""
    from deep_q_learning_amd.General.QLearning.vector_agent import VectorAgent
    from deep_q_learning_amd.LunarLander.dddqn import Model
    n_envs, B = 1024, 512
    e = dq.Engine(dq.EngineConfig(obs_dim=4, hidden1=64, hidden2=64, num_actions=2, capacity=1 << 18, use_per=True,
                                  max_batch=max(n_envs, B), seed=3, lr=5e-4, gamma=0.95, precision=precision))
    e.load(Model(2, hidden=(64, 64)).transformed().init(3, np.zeros((1, 4), np.float32)))
    agent = VectorAgent(e, n_envs, B, env="cartpole", max_steps=500, term_reward=-1.0, epsilon=1.0, epsilon_decay_rate=0.99,
                        min_epsilon=0.05, train_frequency=1, replace_frequency=5, reward_to_reach=150.0, chunk=20)
    hist = agent.training(max_updates=60000)
    rets = np.array([h[2] for h in hist if np.isfinite(h[2])])
    assert rets[:5].mean() < 40                                 # starts near the random policy
    assert rets.max() > 100, (rets.max(), rets[-5:])            # learns: > 4x the random policy's episode length
    assert np.isfinite(e.get_params(host=True)).all()
    e.close()
