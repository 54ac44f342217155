#!/usr/bin/env python3
"""CartPole-v1 on the device-resident loop: prints the learning curve (mean episode length vs updates).
    python tools/cartpole_demo.py [--precision bf16] [--envs 1024] [--batch 512] [--updates 20000]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deep_q_learning_amd as dq  # noqa: E402
from deep_q_learning_amd.General.QLearning.vector_agent import VectorAgent  # noqa: E402
from deep_q_learning_amd.LunarLander.dddqn import Model  # noqa: E402


def run(args, **kw):
    e = dq.Engine(dq.EngineConfig(obs_dim=4, hidden1=64, hidden2=64, num_actions=2, capacity=1 << 18, use_per=not args.uniform,
                                  max_batch=max(args.envs, args.batch), seed=args.seed, lr=kw.get("lr", args.lr), gamma=kw.get("gamma", 0.99),
                                  precision=args.precision))
    e.load(Model(2, hidden=(64, 64)).transformed().init(args.seed, np.zeros((1, 4), np.float32)))
    agent = VectorAgent(e, args.envs, args.batch, env="cartpole", max_steps=500, term_reward=kw.get("term_reward", -1.0),
                        epsilon=1.0, epsilon_decay_rate=kw.get("decay", 0.9), min_epsilon=0.05,
                        train_frequency=kw.get("tf", 1), replace_frequency=kw.get("repl", 1), reward_to_reach=kw.get("goal", 400.0),
                        chunk=20)
    t0 = time.time()
    hist = agent.training(max_updates=args.updates)
    dt = time.time() - t0
    rets = [h[2] for h in hist]
    pts = [f"{hist[i][0]}:{rets[i]:.0f}" for i in range(0, len(hist), max(1, len(hist) // 12))]
    print(kw, f"best {np.nanmax(rets):.1f} final {rets[-1]:.1f} updates {hist[-1][0]} episodes {hist[-1][1]} in {dt:.1f}s  curve", " ".join(pts), flush=True)
    e.close()
    return hist


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f32"); ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=512); ap.add_argument("--updates", type=int, default=20000)
    ap.add_argument("--lr", type=float, default=5e-4); ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--uniform", action="store_true"); ap.add_argument("--sweep", action="store_true")
    args = ap.parse_args()
    if args.sweep:
        for kw in (dict(decay=0.99, repl=5, tf=1), dict(decay=0.995, repl=5, tf=1), dict(decay=0.99, repl=5, tf=1, gamma=0.95),
                   dict(decay=0.99, repl=5, tf=1, lr=1e-3, gamma=0.95), dict(decay=0.99, repl=20, tf=1, gamma=0.95),
                   dict(decay=0.99, repl=5, tf=1, term_reward=1.0, gamma=0.95)):
            run(args, **kw)
    else:
        run(args)
