#!/usr/bin/env python3
"""Stand-alone sorted PER priority write-back (dqn_per_update_sorted) over batch sizes: the r03 segment kernels
(k_per_write_seg + k_per_top_seg) beside the r02 wave-per-64-positions kernels (k_per_write_sorted + k_per_top). Time per call =
HIP events around `reps` back-to-back calls on one stream (both kernels of a call and the gap between them); algorithmic bytes
per updated leaf = 8 L + 12 against the 8 TB/s HBM peak. Ring 2^20, bench priorities, indices = one
stratified PER draw (sorted, with the duplicates such a draw has).
    python tools/per_write_probe.py [--log2 10 13 16 17 18 19 20] [--reps 20] [--json out.json] [--plain N]
--plain N: N un-timed calls per size on the auto path and nothing else (for rocprofv3 --kernel-trace / --pmc runs)."""
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
    ap.add_argument("--log2", type=int, nargs="+", default=[10, 13, 16, 17, 18, 19, 20])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--json", default=None)
    ap.add_argument("--plain", type=int, default=0)
    args = ap.parse_args()
    D, L = bench.D, bench.LOG2N
    maxB = 1 << max(args.log2)
    rows = []
    for path, flags in (("segments", dq._lib.FLAG_PW_SEGMENTS), ("chunks", dq._lib.FLAG_PW_CHUNKS), ("auto", 0)):
        if args.plain and path != "auto":
            continue
        eng = dq.Engine(dq.EngineConfig(obs_dim=D, hidden1=16, hidden2=16, num_actions=bench.A, capacity=1 << L, use_per=True,
                                        max_batch=maxB, seed=77, flags=flags))
        gen = torch.Generator(device=eng.device); gen.manual_seed(99)
        bench.prefill(eng, gen)
        st = eng.stream
        with torch.cuda.stream(st):
            for lb in args.log2:
                B = 1 << lb
                _, idx, _ = eng.per_sample(B, 0.4, seed=1, ctr=lb)
                td = torch.rand(B, device=eng.device, generator=gen) * 2
                for _ in range(max(3, args.plain)):
                    eng.per_update_sorted(idx, td)
                st.synchronize()
                if args.plain:
                    continue
                t = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    for _ in range(args.reps):
                        eng.per_update_sorted(idx, td)
                    e1.record(st); e1.synchronize()
                    t.append(e0.elapsed_time(e1) * 1e3 / args.reps)
                us = float(np.median(t))
                alg = (8 * L + 12) * B
                r = {"path": path, "B": B, "us_per_call": us, "algorithmic_GBs": alg / us / 1e3, "frac_of_8TBs": alg / us / 1e3 / 8000.0}
                rows.append(r)
                print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
        assert eng.device_errors() == 0
        eng.close()
    if args.json:
        json.dump({"config": {"D": D, "log2N": L, "bytes_per_leaf": 8 * L + 12}, "rows": rows}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
