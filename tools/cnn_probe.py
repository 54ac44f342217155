#!/usr/bin/env python3
"""Nature-CNN forward / update in isolation (for rocprofv3 runs):
   python tools/cnn_probe.py [--precision bf16] [--batch 512] [--reps 20] [--mode forward|update]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deep_q_learning_amd as dq
ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16"); ap.add_argument("--batch", type=int, default=512); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--mode", default="forward"); ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph")
a = ap.parse_args()
B = a.batch
e = dq.CnnEngine(num_actions=6, max_batch=B, precision=a.precision)
P = torch.randn(e.param_count) * 0.02
e.set_params(P); e.set_params(P, target=True)
g = torch.Generator(device="cuda"); g.manual_seed(1)
frames = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device="cuda", generator=g)
frames2 = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device="cuda", generator=g)
act = torch.randint(0, 6, (B,), dtype=torch.int32, device="cuda", generator=g)
r = torch.randn(B, device="cuda", generator=g); d = (torch.rand(B, device="cuda", generator=g) < 0.05).float()
q = torch.empty((B, 6), dtype=torch.float32, device="cuda")
step = (lambda: e.forward(frames, out=q)) if a.mode == "forward" else (lambda: e.update(frames, act, r, frames2, d))
for _ in range(3):
    step()
if a.graph:
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        step(); torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_, stream=st):
            step()
    eager, step = step, g_.replay
    step(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(a.reps):
    step()
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.reps
fwd = 2 * (400 * 32 * 256 + 81 * 64 * 512 + 49 * 64 * 576 + 3136 * 512 + 512 * 7)
bwd = 2 * (2 * (81 * 64 * 512 + 49 * 64 * 576 + 3136 * 512 + 512 * 7) + 400 * 32 * 256)      # dX and dW per layer; conv1 has no dX
gf = B * (fwd if a.mode == "forward" else 3 * fwd + bwd)
print(a.precision, a.mode, "B", B, "us", round(us, 1), "TF", round(gf / us * 1e-6, 1))
e.close()
