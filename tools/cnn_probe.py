#!/usr/bin/env python3
"""Nature-CNN forward in isolation (for rocprofv3 runs):  python tools/cnn_probe.py [--precision bf16] [--batch 512] [--reps 20]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deep_q_learning_amd as dq
ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16"); ap.add_argument("--batch", type=int, default=512); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
e = dq.CnnEngine(num_actions=6, max_batch=a.batch, precision=a.precision)
e.set_params(torch.randn(e.param_count) * 0.02)
frames = torch.randint(0, 256, (a.batch, 84, 84, 4), dtype=torch.uint8, device="cuda")
q = torch.empty((a.batch, 6), dtype=torch.float32, device="cuda")
for _ in range(3):
    e.forward(frames, out=q)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(a.reps):
    e.forward(frames, out=q)
e1.record(); e1.synchronize()
print(a.precision, "B", a.batch, "us per forward", e0.elapsed_time(e1) * 1e3 / a.reps)
e.close()
