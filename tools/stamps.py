#!/usr/bin/env python3
"""Diagnostic (not part of the product): run the bench-shaped inner loop on the -DDQN_STAMPS build and
print where block (0,0) of each instrumented kernel spends its time, plus the shader clock it ran at
(s_memtime ticks / s_memrealtime 100 MHz ticks).   DQN_HIP_LIB=.../libdqn_hip_stamps.so python tools/stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DQN_HIP_LIB", os.path.join(ROOT, "deep-q-learning_amd", "libdqn_hip_stamps.so"))
import bench  # noqa: E402
import deep_q_learning_amd as dq  # noqa: E402

NAMES = {0: ("k_qnet_fwd (+ fused row backward)", ["start", "x staged", "L1 done", "L2 mfma done", "L2 epilogue", "heads done", "forward end",
                                                    "partners' Q rows in", "TD rows done", "dz2 done", "dz1 done"]),
         1: ("k_env_step", ["start", "synth+ring stores", "leaves set", "levels done", "post barrier"]),
         4: ("k_per_write_sorted", ["start", "ownership found", "leaf + sib loads", "levels done", "end"]),
         5: ("k_bwd_rows", ["start", "prefetch issued + wmax", "td rows done", "dz2 done", "end"]),
         6: ("k_dw", ["start", "mfma loop done", "epilogue (adam) done", "end"]),
         7: ("k_actor (T=4)", ["start", "weights requested, x staged"] +
             [f"t{t} {w}" for t in range(4) for w in ("L1 done", "L2 done", "heads+policy done", "env done")]),
         2: ("per_top_wg (tree workgroup of k_actor)", ["start", "depth-14 loads landed, depth 13 summed", "register tree written to the LDS image", "coalesced copy-out issued", "LDS levels done", "final drain"]),
         3: ("k_actor side chain (tree workgroup, then sampler workgroup 0)",
             ["tree wg start", "end nodes walked", "inner nodes in + top rebuilt", "flag released", "sampler start", "flag seen", "acquired", "batch drawn",
              "tile 0: u drawn", "levels 1-4", "levels 5-8", "levels 9-12", "levels 13-16", "levels 17-20", "leaf, weight, stores", "barrier"])}


def main():
    if os.environ.get("DQN_STAMPS_CFG3") == "1":           # BASELINE configs[2]: CartPole, 4096 envs, 2x64, B = 8192
        bench.D, bench.H1, bench.H2, bench.A, bench.B, bench.N_ENVS = 4, 64, 64, 2, 8192, 4096
    eng = dq.Engine(dq.EngineConfig(obs_dim=bench.D, hidden1=bench.H1, hidden2=bench.H2, num_actions=bench.A,
                                    capacity=1 << bench.LOG2N, use_per=True, max_batch=bench.B, seed=1,
                                    precision=os.environ.get("DQN_STAMPS_PRECISION", "f32")))
    gen = torch.Generator(device=eng.device); gen.manual_seed(0)
    eng.set_params(torch.randn(eng.param_count) * 0.05); eng.sync_target()
    bench.prefill(eng, gen)
    if os.environ.get("DQN_STAMPS_CFG3") == "1":
        eng.env_config("cartpole", 500, -1.0)
    eng.env_reset(torch.randn(bench.N_ENVS, bench.D, device=eng.device, generator=gen) * 0.05, 0.01)
    eng.set_epsilon(0.15)
    with torch.cuda.stream(eng.stream):
        for _ in range(100):
            eng.train_iters(10, 4, bench.B)
        eng.stream.synchronize()
    buf = (C.c_ulonglong * (8 * 64 * 2))()
    rc = eng.lib.dqn_debug_stamps(buf)
    assert rc == 0, rc
    st = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64, 2).astype(np.int64)
    pro = st[7, 20:25]
    if pro[0, 0]:
        print("k_actor prologue (cycles since start): " + ", ".join(
            f"{lab} +{int(c - st[7, 0, 0])}" for lab, c in zip(("small loads issued", "zero-fill barrier", "lwh + x staged", "W2 issued", "W2 landed"), pro[:, 0])))
    if st[7, 25, 0]:
        print(f"k_actor step 1 heads phase (cycles since its L2 done): chain done +{int(st[7, 25, 0] - st[7, 7, 0])}, "
              f"draw flags seen +{int(st[7, 26, 0] - st[7, 7, 0])}, policy + stores done +{int(st[7, 8, 0] - st[7, 7, 0])}")
    if st[7, 30, 0]:
        print("k_actor step 1, layer-1 phase (cycles since the previous step's end): " + ", ".join(
            f"{lab} +{int(st[7, i, 0] - st[7, 5, 0])}" for lab, i in (("loop top done", 30), ("MFMAs done", 31), ("epilogue stored", 32), ("barrier passed", 6))))
    if st[7, 27, 0]:
        print(f"k_actor step 1, wave 1 (real-time ns since the step's L2 done): draws start {int(st[7, 27, 1] - st[7, 7, 1]) * 10}, "
              f"flag written {int(st[7, 28, 1] - st[7, 7, 1]) * 10}; wave 0: chain done {int(st[7, 25, 1] - st[7, 7, 1]) * 10}, "
              f"flags seen {int(st[7, 26, 1] - st[7, 7, 1]) * 10}, phase end {int(st[7, 8, 1] - st[7, 7, 1]) * 10}")
    if st[0, 11, 0]:
        print("k_qnet_fwd layer-1 phase (cycles since x staged): " + ", ".join(
            f"{lab} +{int(st[0, i, 0] - st[0, 1, 0])}" for lab, i in (("MFMAs done", 11), ("epilogue stores issued", 12), ("head weights requested", 13), ("barrier passed", 2))))
    if st[2, 6, 0]:
        print("per_add_range_wg (cycles since its start): " + ", ".join(
            f"{lab} +{int(st[2, i, 0] - st[2, 6, 0])}" for lab, i in (("inner nodes stored", 7), ("barrier", 8), ("end-node chain done", 9))))
    if st[4, 0, 1] and st[6, 0, 1]:
        print("priority write-back inside k_dw, wave of chunk 0 (real-time us since block 0 of the same launch started): " + ", ".join(
            f"{lab} {(int(st[4, i, 1]) - int(st[6, 0, 1])) / 100.0:.2f}" for lab, i in (("start", 0), ("ownership found", 1), ("leaf + sib loads", 2), ("levels done", 3), ("end", 4)))
              + f"; block 0 (dW tile) ends at {(int(st[6, 3, 1]) - int(st[6, 0, 1])) / 100.0:.2f}")
    for k, (name, labels) in NAMES.items():
        t = st[k, :len(labels)]
        cyc = t[:, 0] - t[0, 0]; real = (t[:, 1] - t[0, 1]) * 10.0     # ns
        tot_c, tot_ns = cyc[-1], real[-1]
        mhz = tot_c / tot_ns * 1e3 if tot_ns > 0 else float("nan")
        print(f"{name}: {tot_c} cycles in {tot_ns / 1e3:.2f} us  => shader clock ~{mhz:.0f} MHz")
        for i, lab in enumerate(labels):
            print(f"    {lab:20s} +{cyc[i]:8d} cyc  {real[i] / 1e3:7.2f} us")


if __name__ == "__main__":
    main()
