#!/usr/bin/env python3
"""Generates tests/golden/cnn_B4_seed7.npz from the CPU oracle of the Nature-CNN path (oracle/dqn_oracle_cnn.c: f32 fmaf chains
and the f64 form). Run from the repo root:  python tests/golden/make_cnn_golden.py

PARITY UNPINNED (the reference has no CNN): a regression pin of the build's own restatement. Data only: the inputs are
re-generated from the recorded seeds (numpy Generator streams are stable), the file holds seeds + expected outputs:
Q of the f32 and f64 forwards, loss, and per-leaf sums / absolute sums / a strided sample of the f64 gradient."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import _oracle as oc  # noqa: E402
from _oracle import onp  # noqa: E402

A, B, SEED = 6, 4, 7
LEAVES = [8 * 8 * 4 * 32, 32, 4 * 4 * 32 * 64, 64, 3 * 3 * 64 * 64, 64, 3136 * 512, 512, 512, 1, 512 * A, A]


def inputs(seed=SEED, B=B):
    rng = np.random.default_rng(seed)
    P = onp.cnn_init_params(A, seed)
    P = (P + 0.01 * rng.standard_normal(P.size)).astype(np.float32)
    frames = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
    noise = rng.standard_normal((B, A)); scale = rng.choice([0.2, 2.5], (B, 1)); isw = rng.uniform(0.3, 1.0, B).astype(np.float32)
    return P, frames, noise, scale, isw


if __name__ == "__main__":
    P, frames, noise, scale, isw = inputs()
    q32, feat32 = oc.cnn_forward(P, frames, A)
    q64 = onp.cnn_forward(P, frames, A, np.float64)
    targets = (q32 + noise * scale).astype(np.float32)
    g64, l64 = oc.cnn_grads(P, frames, targets, isw, A, f64=True)
    g32, l32 = oc.cnn_grads(P, frames, targets, isw, A)
    sums, abss, o = [], [], 0
    for n in LEAVES:
        sums.append(g64[o:o + n].sum()); abss.append(np.abs(g64[o:o + n]).sum()); o += n
    np.savez_compressed(os.path.join(HERE, f"cnn_B{B}_seed{SEED}.npz"), A=np.int64(A), B=np.int64(B), seed=np.int64(SEED),
                        param_sum=np.float64(P.astype(np.float64).sum()), frame_sum=np.int64(frames.astype(np.int64).sum()),
                        q32=q32, q64=q64, feat32_sum=np.float64(feat32.astype(np.float64).sum()), targets=targets,
                        loss64=np.float64(l64), loss32=np.float32(l32), leaf_sums=np.array(sums), leaf_abs=np.array(abss),
                        grad64_strided=g64[::997].copy(), grad32_strided=g32[::997].copy())
    print("written", f"cnn_B{B}_seed{SEED}.npz")
