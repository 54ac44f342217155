#!/usr/bin/env python3
"""large-batch Q-net kernels (dqn_net_big.hip) in isolation, for rocprofv3 runs:
    python tools/big_probe.py --mode fwd|grads|update --log2 15 [--reps 20] [--precision f32]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import deep_q_learning_amd as dq  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="fwd")
ap.add_argument("--log2", type=int, default=15)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--precision", default="f32")
ap.add_argument("--lib", default=None, help="a variant build of the library")
args = ap.parse_args()
if args.lib:
    dq._lib.LIB_PATH = os.path.abspath(args.lib)
D, H1, H2, A = bench.D, bench.H1, bench.H2, bench.A
B = 1 << args.log2
eng = dq.Engine(dq.EngineConfig(obs_dim=D, hidden1=H1, hidden2=H2, num_actions=A, capacity=1 << 20, use_per=True, max_batch=B, seed=3,
                                precision=args.precision))
gen = torch.Generator(device=eng.device); gen.manual_seed(0)
eng.set_params(torch.randn(eng.param_count) * 0.05); eng.sync_target()
bench.prefill(eng, gen)
x = torch.randn(B, D, device=eng.device, generator=gen)
tg = torch.randn(B, A, device=eng.device, generator=gen)
st = eng.stream
with torch.cuda.stream(st):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn = {"fwd": lambda: eng.forward(x), "grads": lambda: eng.lib.dqn_grads(eng.h, x.data_ptr(), tg.data_ptr(), None, B, None, eng._s()),
          "update": lambda: eng.update(B, st)}[args.mode]
    fn(); fn()
    st.synchronize(); e0.record(st)
    for _ in range(args.reps):
        fn()
    e1.record(st); e1.synchronize()
    print(args.mode, "B", B, "us per call", e0.elapsed_time(e1) * 1e3 / args.reps, flush=True)
assert eng.device_errors() == 0
eng.close()
