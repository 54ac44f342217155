#!/usr/bin/env python3
"""Batch sweep: per-kernel achieved GB/s / TFLOP/s against the roofline for B = 2^10 .. 2^max,
to separate the latency floor (what bench.py's B=1024 sits on) from the bandwidth / MFMA slope.
    python tools/sweep.py [--max-log2 17] [--json out.json]
Timing: HIP events around `reps` back-to-back eager launches on one stream (launch gaps included)."""
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

D, H1, H2, A, L = bench.D, bench.H1, bench.H2, bench.A, bench.LOG2N
F = 2 * (D * H1 + H1 * H2 + H2 * (1 + A))


def timed(fn, reps, stream):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream.synchronize()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-log2", type=int, default=17)
    ap.add_argument("--json", default=None)
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--no-actor", action="store_true")
    ap.add_argument("--big-any", action="store_true", help="force the 64-row large-batch kernels at every batch size (crossover search)")
    args = ap.parse_args()
    peak = 157.3 if args.precision == "f32" else 2500.0
    maxB = 1 << args.max_log2
    eng = dq.Engine(dq.EngineConfig(obs_dim=D, hidden1=H1, hidden2=H2, num_actions=A, capacity=1 << L, use_per=True,
                                    max_batch=maxB, seed=3, precision=args.precision, flags=dq._lib.FLAG_BIG_ROWS if args.big_any else 0))
    gen = torch.Generator(device=eng.device); gen.manual_seed(0)
    eng.set_params(torch.randn(eng.param_count) * 0.05); eng.sync_target()
    bench.prefill(eng, gen)
    rows = []
    with torch.cuda.stream(eng.stream):
        st = eng.stream
        for lb in range(10, args.max_log2 + 1):
            B = 1 << lb
            reps = max(3, min(200, (1 << 22) // B))
            x = torch.randn(B, D, device=eng.device, generator=gen)
            tg = torch.randn(B, A, device=eng.device, generator=gen)
            batch, idx, isw = eng.per_sample(B, 0.4, seed=1, ctr=0)
            td = torch.rand(B, device=eng.device, generator=gen)
            r = {"B": B}
            t = timed(lambda: eng.per_sample(B, 0.4, seed=1, ctr=1), reps, st)          # + k_isw_normalize
            r["per_sample_us"] = t * 1e6; r["per_sample_GBs"] = (4 * L + 16 * D + 26) * B / t / 1e9
            t = timed(lambda: eng.per_update_sorted(idx, td), reps, st)
            r["per_update_sorted_us"] = t * 1e6; r["per_update_sorted_GBs"] = (8 * L + 12) * B / t / 1e9
            t = timed(lambda: eng.forward(x), reps, st)
            r["fwd_us"] = t * 1e6; r["fwd_TFs"] = F * B / t / 1e12
            t = timed(lambda: eng.lib.dqn_grads(eng.h, x.data_ptr(), tg.data_ptr(), None, B, None, eng._s()), reps, st)
            bk = F + 2 * (H1 * H2 + H2 * (1 + A))
            r["grads_us"] = t * 1e6; r["grads_TFs"] = (F + bk) * B / t / 1e12          # 1 fwd + bwd
            # the whole Agent._step on the handle's replay: PER sample, 3 forwards, TD, row backward, dW, AdamW, write-back
            t = timed(lambda: eng.update(B, st), max(3, reps // 2), st)
            r["update_us"] = t * 1e6; r["update_TFs"] = (3 * F + bk) * B / t / 1e12
            for k in ("fwd", "grads", "update"):
                r[k + "_frac_of_mfma_peak"] = r[k + "_TFs"] / peak
            rows.append(r)
            print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
    # k_actor (4 vector env steps per launch) over the number of envs: 4-env tiles, <= 255 actor workgroups
    actor_rows = []
    eng.close()
    for ln in ([] if args.no_actor else range(8, 17, 2)):
        n = 1 << ln
        eng = dq.Engine(dq.EngineConfig(obs_dim=D, hidden1=H1, hidden2=H2, num_actions=A, capacity=1 << L, use_per=True,
                                        max_batch=max(n, 1024), seed=3, precision=args.precision))
        eng.set_params(torch.randn(eng.param_count) * 0.05)
        eng.env_reset(torch.randn(n, D, device=eng.device), 0.01); eng.set_epsilon(0.15)
        with torch.cuda.stream(eng.stream):
            t = timed(lambda: eng.actor_steps(4, eng.stream), max(5, min(200, (1 << 20) // n)), eng.stream)
        r = {"n_envs": n, "actor_steps4_us": t * 1e6, "env_steps_per_s": 4 * n / t, "TFs": 4 * n * F / t / 1e12}
        actor_rows.append(r)
        print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
        eng.close()
    out = {"config": {"D": D, "H1": H1, "H2": H2, "A": A, "log2N": L, "dtype": args.precision},
           "peaks": {"hbm_GBs": 8000.0, "mfma_TFs": peak}, "rows": rows, "actor_rows": actor_rows}
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
