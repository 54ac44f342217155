#!/usr/bin/env python3
"""Batch sweep of the Nature-CNN forward and whole update (dqn_cnn_update), both precision modes:
    python tools/cnn_sweep.py [--json out.json] [--log2 9 11 13 15]
Rates against the dense MFMA peaks (bf16 2.5 PF, f32 157.3 TF); FLOP = exact layer arithmetic (3 forwards + dX of conv2 / conv3 / fc /
heads + dW of every layer for the update)."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deep_q_learning_amd as dq

FWD = 2 * (400 * 32 * 256 + 81 * 64 * 512 + 49 * 64 * 576 + 3136 * 512 + 512 * 7)
BWD = 2 * (2 * (81 * 64 * 512 + 49 * 64 * 576 + 3136 * 512 + 512 * 7) + 400 * 32 * 256)
PEAK = {"bf16": 2500.0, "f32": 157.3}


def timed(fn, iters):
    for _ in range(3):
        fn()
    reps = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters):
            fn()
        e1.record(); e1.synchronize()
        reps.append(e0.elapsed_time(e1) * 1e3 / iters)
    return float(np.median(reps))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--json", default=None); ap.add_argument("--log2", type=int, nargs="+", default=[9, 11, 13, 15])
    ap.add_argument("--precision", nargs="+", default=["bf16", "f32"]); ap.add_argument("--flags", type=int, default=0, help="dqn_cnn_set_flags (4 = per-layer convolutions)")
    a = ap.parse_args()
    rows = []
    for prec in a.precision:
        for lb in a.log2:
            B = 1 << lb
            e = dq.CnnEngine(num_actions=6, max_batch=B, precision=prec)
            if a.flags: e.set_flags(a.flags)
            P = torch.randn(e.param_count) * 0.02
            e.set_params(P); e.set_params(P, target=True)
            g = torch.Generator(device="cuda"); g.manual_seed(lb)
            f1 = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device="cuda", generator=g)
            f2 = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device="cuda", generator=g)
            act = torch.randint(0, 6, (B,), dtype=torch.int32, device="cuda", generator=g)
            r = torch.randn(B, device="cuda", generator=g); d = (torch.rand(B, device="cuda", generator=g) < 0.05).float()
            q = torch.empty((B, 6), dtype=torch.float32, device="cuda")
            iters = max(8, 4096 // B * 4)
            fu = timed(lambda: e.forward(f1, out=q), iters)
            uu = timed(lambda: e.update(f1, act, r, f2, d), max(6, iters // 3))
            row = {"precision": prec, "B": B, "forward_us": round(fu, 1), "forward_TFs": round(FWD * B / fu / 1e6, 1), "forward_frac_of_mfma_peak": round(FWD * B / fu / 1e6 / PEAK[prec], 4),
                   "update_us": round(uu, 1), "update_TFs": round((3 * FWD + BWD) * B / uu / 1e6, 1), "update_frac_of_mfma_peak": round((3 * FWD + BWD) * B / uu / 1e6 / PEAK[prec], 4)}
            print(row, flush=True)
            rows.append(row)
            e.close(); del f1, f2; torch.cuda.empty_cache()
    if a.json:
        json.dump({"rows": rows, "fwd_flop_per_sample": FWD, "update_flop_per_sample": 3 * FWD + BWD, "peaks_TFs": PEAK}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
