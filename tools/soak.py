#!/usr/bin/env python3
"""Soak test of the captured inner loop (both precision modes, bench shape): tens of thousands of graph replays with the
in-launch hand-overs (tree workgroup -> samplers in k_actor, Q rows -> pass-0 workgroup in the forward launch). A hand-over
that ever timed out poisons the loss with NaN; the sum-tree invariant is checked along the way.
    python tools/soak.py [--seconds 40]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import deep_q_learning_amd as dq  # noqa: E402


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=40.0)
    ap.add_argument("--cfg3", action="store_true", help="BASELINE configs[2] (CartPole, 4096 envs, 2x64, B = 8192) instead of the bench shape")
    ap.add_argument("--cnn", action="store_true", help="the configs[4] loop (CnnVectorAgent: 512 synthetic frame-stack envs, n-step 3, PER index on its own stream) instead")
    args = ap.parse_args()
    if args.cnn:
        return soak_cnn(args.seconds)
    if args.cfg3:
        bench.D, bench.H1, bench.H2, bench.A, bench.B, bench.N_ENVS = 4, 64, 64, 2, 8192, 4096
    for prec in ("f32", "bf16"):
        e = dq.Engine(dq.EngineConfig(obs_dim=bench.D, hidden1=bench.H1, hidden2=bench.H2, num_actions=bench.A, capacity=1 << bench.LOG2N,
                                      use_per=True, max_batch=bench.B, seed=7, precision=prec))
        gen = torch.Generator(device=e.device); gen.manual_seed(0)
        e.set_params(bench.init_params(e.param_count)); e.sync_target()
        bench.prefill(e, gen)
        if args.cfg3:
            e.env_config("cartpole", 500, -1.0)
        e.env_reset(torch.randn(bench.N_ENVS, bench.D, device=e.device, generator=gen) * (0.05 if args.cfg3 else 1.0), 0.01); e.set_epsilon(0.15)
        N = 1 << bench.LOG2N
        k = torch.arange(1, N, device=e.device)
        t0 = time.time(); iters = 0; checks = 0
        with torch.cuda.stream(e.stream):
            while time.time() - t0 < args.seconds:
                for _ in range(50):
                    e.train_iters(20, 4, bench.B, e.stream)
                iters += 1000
                e.stream.synchronize()
                loss = float(e.last_loss().item())
                assert loss == loss, f"{prec}: NaN loss after {iters} iterations (a hand-over timed out?)"
                if iters % 20000 == 0:
                    t = e.buffer(dq._lib.BUF_TREE)
                    bad = (t[k] != t[2 * k] + t[2 * k + 1]).nonzero()
                    assert bad.numel() == 0, (prec, iters, bad[:4].flatten().tolist())
                    checks += 1
                    e.sync_target()
        assert e.opt_count() == iters
        print(f"{prec}: {iters} iterations in {time.time() - t0:.1f} s ({iters / (time.time() - t0):.0f}/s incl. checks), loss {loss:.4f}, "
              f"{checks} tree checks ok", flush=True)
        e.close()


def soak_cnn(seconds):
    """the loop of BASELINE configs[4]'s shape: tens of thousands of host-driven iterations across the three streams (CNN, its side
    streams, the index stream); along the way: finite loss and parameters, no device error, the index's sum-tree invariant, both rings
    in lockstep"""
    from deep_q_learning_amd.General.QLearning.cnn_agent import CnnVectorAgent
    ag = CnnVectorAgent(n_envs=512, num_actions=6, capacity=1 << 14, batch_size=512, precision="bf16", train_frequency=4, seed=5, n_step=3)
    ag.init_params(torch.randn(ag.cnn.param_count) * 0.02)
    N = 1 << 14
    k = torch.arange(1, N, device=ag.cnn.device)
    t0 = time.time(); iters = 0; checks = 0
    while time.time() - t0 < seconds:
        ag.training(500)
        iters += 500
        loss = ag.update(want_loss=True); iters += 0
        assert loss == loss and abs(loss) < 1e6, f"loss {loss} after {iters} iterations"
        torch.cuda.synchronize()
        t = ag.index.buffer(dq._lib.BUF_TREE)
        bad = (t[k] != t[2 * k] + t[2 * k + 1]).nonzero()
        assert bad.numel() == 0, (iters, bad[:4].flatten().tolist())
        assert ag.index.device_errors() == 0
        assert bool(torch.isfinite(ag.cnn.get_buffer("params")).all())
        assert ag.cnn.replay_size()[1] == ag.env_steps * 512 and ag.index.replay_size()[1] == (ag.env_steps - 2) * 512
        checks += 1
    dt = time.time() - t0
    print(f"cnn loop (bf16, 512 envs, n-step 3): {iters} iterations in {dt:.1f} s ({iters / dt:.0f}/s incl. checks), loss {loss:.4f}, {checks} checks ok "
          f"(tree invariant, finite parameters, rings in lockstep, no device error)", flush=True)
    ag.close()


if __name__ == "__main__":
    main()
