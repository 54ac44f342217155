#!/usr/bin/env python3
"""Stand-alone PER-sample kernel (k_per_sample2 behind dqn_per_sample) over batch sizes: per-launch time from the
dispatch's own start/stop events (hipExtLaunchKernelGGL), algorithmic GB/s (4L + 16D + 26 bytes per
sampled transition) against the 8 TB/s HBM peak. Ring 2^20, D = 8, bench priorities.
    python tools/per_sample_probe.py [--log2 10 16 18 20 21 22] [--reps 30] [--json out.json] [--plain N]
--plain N: N un-profiled launches per size and nothing else (for rocprofv3 --pmc / --kernel-trace runs)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import deep_q_learning_amd as dq  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, nargs="+", default=[10, 12, 14, 16, 18, 20, 21, 22])
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--json", default=None)
    ap.add_argument("--plain", type=int, default=0)
    ap.add_argument("--lib", default=None, help="a variant build of the library (other PS_* macros)")
    ap.add_argument("--log2n", type=int, default=None, help="ring capacity 2^n instead of the bench's 2^20 (2^24: ring 1 GB + tree 128 MB, beyond the 256 MB Infinity Cache)")
    args = ap.parse_args()
    if args.lib:
        dq._lib.LIB_PATH = os.path.abspath(args.lib)
    if args.log2n:
        bench.LOG2N = args.log2n                     # (prefill and the byte formula read it)
    D, L = bench.D, bench.LOG2N
    maxB = 1 << max(args.log2)
    eng = dq.Engine(dq.EngineConfig(obs_dim=D, hidden1=16, hidden2=16, num_actions=bench.A, capacity=1 << L, use_per=True,
                                    max_batch=maxB, seed=77))
    gen = torch.Generator(device=eng.device); gen.manual_seed(99)
    bench.prefill(eng, gen)
    st = eng.stream
    rows = []
    with torch.cuda.stream(st):
        for lb in args.log2:
            B = 1 << lb
            bufs = eng._batch_out(B) + (eng.empty((B,), torch.int32), eng.empty((B,), torch.float32))
            if args.plain:
                for it in range(args.plain):
                    eng.per_sample_into(B, 0.4, 1, it, bufs)
                st.synchronize()
                continue
            ms = []
            for it in range(args.reps):
                eng.profile_begin(st)
                eng.per_sample_into(B, 0.4, 1, it, bufs)
                ms += [m for n, m in eng.profile_end(st) if n == "per_sample"]
            us = float(np.median(ms[3:])) * 1e3
            alg = (4 * L + 16 * D + 26) * B
            r = {"B": B, "us": us, "min_us": float(np.min(ms[3:])) * 1e3, "algorithmic_GBs": alg / us / 1e3, "frac_of_8TBs": alg / us / 1e3 / 8000.0}
            rows.append(r)
            print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
    assert eng.device_errors() == 0
    eng.close()
    if args.json:
        json.dump({"config": {"D": D, "log2N": L, "bytes_per_sample": 4 * L + 16 * D + 26}, "rows": rows}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
