#!/usr/bin/env python3
"""Steady-state rates of the captured inner loop on the other single-GPU shapes of BASELINE.json (parity-test cases, not the
bench line): configs[2] = CartPole-v1 physics, 4096 vectorised envs, 2x64 net, PER batch 8192; plus the reference's own
net size (9-32-64-4, batch 64, 64 envs). Prints updates/s and env-steps/s per precision.   python tools/cfg_rates.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deep_q_learning_amd as dq  # noqa: E402

CASES = [  # name, dims, n_envs, B, log2 capacity, env kind, train_frequency
    ("configs[2] CartPole 4096 envs, 2x64, PER B=8192", (4, 64, 64, 2), 4096, 8192, 20, "cartpole", 4),
    ("reference net 9-32-64-4, 64 envs, PER B=64", (9, 32, 64, 4), 64, 64, 17, "synthetic", 4),
]


def main():
    for name, dims, n, B, L, kind, tf in CASES:
        for prec in ("f32", "bf16"):
            e = dq.Engine(dq.EngineConfig(obs_dim=dims[0], hidden1=dims[1], hidden2=dims[2], num_actions=dims[3], capacity=1 << L,
                                          use_per=True, max_batch=max(n, B), seed=1, precision=prec))
            g = torch.Generator(device=e.device); g.manual_seed(0)
            e.set_params(torch.randn(e.param_count) * 0.05); e.sync_target()
            N = 1 << L
            for k in range(0, N, 1 << 16):
                c = min(1 << 16, N - k)
                e.replay_add(torch.randn(c, dims[0], device=e.device, generator=g) * 0.05, torch.randint(0, dims[3], (c,), device=e.device, generator=g, dtype=torch.int32),
                             torch.randn(c, device=e.device, generator=g), torch.randn(c, dims[0], device=e.device, generator=g) * 0.05,
                             torch.rand(c, device=e.device, generator=g) < 0.02)
            if kind == "cartpole":
                e.env_config("cartpole", 500, -1.0)
                obs = torch.rand(n, 4, device=e.device, generator=g) * 0.1 - 0.05
            else:
                obs = torch.randn(n, dims[0], device=e.device, generator=g)
            e.env_reset(obs, 0.01); e.set_epsilon(0.15)
            st = e.stream
            with torch.cuda.stream(st):
                for _ in range(5):
                    e.train_iters(20, tf, B, st)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                reps = 25
                for _ in range(reps):
                    e.train_iters(20, tf, B, st)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            ups = reps * 20 / dt
            print(f"{name} [{prec}]: {ups:,.0f} updates/s ({1e6 / ups:.1f} us/step), {ups * tf * n:,.0f} env-steps/s, "
                  f"{ups * B:,.0f} sampled transitions/s, loss {float(e.last_loss().item()):.4f}", flush=True)
            e.close()


if __name__ == "__main__":
    main()
