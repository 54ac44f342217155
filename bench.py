#!/usr/bin/env python3
"""bench.py -- DDDQN inner training loop on MI355X: grad-updates/sec + env-steps/sec.

Workload (BASELINE.json configs[1]): LunarLander shape (obs 8, act 4), 256 vectorised synthetic envs,
proportional PER over a 2^20-transition ring resident in HBM, batch 1024, dueling MLP 8-256-256-{1,4},
AdamW. One "step" = the reference's inner loop at train_frequency=4 (Test/lunar_lander.py:30,
q_agent.py:176-187): 4 vector env steps (act -> synthetic transition -> replay.add, 1024 env-steps)
followed by one Agent._step (PER sample -> double-Q targets -> Huber grad -> AdamW -> priority write-back).
Inputs are synthetic and already resident in HBM when the timed region starts.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: independent learners (own envs, ring, tree, minibatch), one gradient all-reduce (RCCL) per
update; value = minibatch updates of all ranks per second (weak scaling, global batch N*1024).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, H1, H2, A = 8, 256, 256, 4
LOG2N = 20
B = 1024
N_ENVS = 256
TRAIN_FREQ = 4            # Test/lunar_lander.py:30
ITERS_PER_GRAPH = int(os.environ.get("DQN_BENCH_ITERS_PER_GRAPH", "20"))      # inner-loop iterations captured per hipGraph launch (single GPU)
P_DONE = 0.01
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0


def algorithmic_cost(name, L, n_params):
    """(bound, units per launch) -- SURVEY.md 8(d) per-unit figures x units per launch.
    bytes for HBM-bound kernels, FLOP for MFMA-bound ones."""
    F = 2 * (D * H1 + H1 * H2 + H2 * (1 + A))
    if name == "per_sample":
        return "hbm", (4 * L + 16 * D + 26) * B
    if name == "per_update":
        return "hbm", (8 * L + 12) * B
    if name == "per_add":
        return "hbm", (8 * L + 4) * N_ENVS
    if name == "adam":
        return "hbm", 28 * n_params
    if name == "env_step_add":        # synthetic transition + ring insert + PER leaf-range insert
        return "hbm", ((4 * D + 5) + 2 * (8 * D + 9) + (8 * L + 4)) * N_ENVS
    if name in ("qnet_fwd_x3", "sample_fwd_x3"):     # three forwards (+ the fused PER sampling of their rows)
        return "mfma", 3 * F * B
    if name == "sample_fwd_x3_bwd":                  # + TD / Huber gradient / row backward of the same tiles
        return "mfma", (3 * F + 2 * (1 + A) * H2 + 2 * H1 * H2) * B
    if name in ("act_fwd_policy", "actor_step"):      # forward + policy + env step + ring/tree insert, one launch
        return "mfma", F * N_ENVS
    if name == "actor_steps":         # k_actor: TRAIN_FREQ vector env steps in one launch (+ leaves, + next batch's PER draw)
        return "mfma", F * N_ENVS * TRAIN_FREQ
    if name == "td_bwd_rows":
        return "mfma", (2 * (1 + A) * H2 + 2 * H1 * H2) * B
    if name == "per_top":
        return "hbm", 2 * 4 * 1024
    if name in ("dw", "dw_adam", "dw_adam_perwrite", "dw_perwrite"):
        return "mfma", 2 * B * (D * H1 + H1 * H2 + H2 * (1 + A))
    return "hbm", 0


PMC_KEYS = {   # bench kernel label -> (kernel name in profiles/*_pmc.json, FETCH_SIZE correction)
    # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of wide (16 B/lane) coalesced reads; other widths
    # are uncalibrated and taken as reported. The forward kernel appears once per grid size (actor vs 3-pass launch).
    "actor_steps": ("k_actor", 1.0), "actor_step": ("k_qnet_fwd:min", 2.0), "qnet_fwd_x3": ("k_qnet_fwd:max", 2.0), "sample_fwd_x3": ("k_qnet_fwd:max", 2.0), "sample_fwd_x3_bwd": ("k_qnet_fwd:max", 2.0),
    "td_bwd_rows": ("k_bwd_rows", 2.0), "dw_adam_perwrite": ("k_dw", 2.0), "dw_perwrite": ("k_dw", 2.0), "dw_adam": ("k_dw", 2.0), "dw": ("k_dw", 2.0),
    "per_sample": ("k_per_sample", 1.0), "per_top": ("k_per_top", 1.0),
}


KERNEL_OF = {   # bench label -> kernel (symbol in the rocprofv3 summaries under profiles/)
    "actor_steps": "k_actor (4 vector env steps of 256 envs in one launch + leaf insert + the next batch's PER draw)",
    "td_bwd_rows": "k_bwd_rows", "dw_adam_perwrite": "k_dw (+ Adam + PER write-back)", "dw_perwrite": "k_dw (+ PER write-back)",
    "dw_adam": "k_dw (+ Adam)", "dw": "k_dw", "per_top": "k_per_top", "adam": "k_adam", "allreduce": "RCCL all-reduce",
}


def pmc_traffic(label):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/), or None"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if files and label.startswith("per_sample_B"):     # the stand-alone sampler, one record per batch size
        rec = json.load(open(files[-1])).get("k_per_sample2/" + label[len("per_sample_"):])
        if not rec or "FETCH_SIZE_KB_per_launch_median" not in rec or "WRITE_SIZE_KB_per_launch_median" not in rec:
            return None
        return (rec["FETCH_SIZE_KB_per_launch_median"] * rec.get("fetch_correction", 1.0) + rec["WRITE_SIZE_KB_per_launch_median"]) * 1024.0
    if not files or label not in PMC_KEYS:
        return None
    key, corr = PMC_KEYS[label]
    recs = json.load(open(files[-1]))
    name, _, pick = key.partition(":")
    cands = [k for k in recs if k.split("<")[0].split("/")[0] == name]
    if name == "k_qnet_fwd":                     # the forward launch with / without the fused row backward (template flag)
        want = "true>" if label.endswith("_bwd") else "false>"
        cands = [k for k in cands if want in k or ("true>" not in k and "false>" not in k)]
    if not cands:
        return None
    if pick:                                     # several grid sizes of one kernel: smallest / largest grid
        cands.sort(key=lambda k: int(k.rsplit("grid", 1)[1]) if "grid" in k else 0)
        cands = [cands[0] if pick == "min" else cands[-1]]
    rec = recs[cands[0]]
    if "FETCH_SIZE_KB_per_launch_median" not in rec or "WRITE_SIZE_KB_per_launch_median" not in rec:
        return None
    return (rec["FETCH_SIZE_KB_per_launch_median"] * corr + rec["WRITE_SIZE_KB_per_launch_median"]) * 1024.0


def mfma_counters(kernel):
    """MFMA-pipe counters of `kernel` from the committed rocprofv3 --pmc pass (profiles/r*_pmc_mfma.json), or None"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")))
    if not files:
        return None
    recs = json.load(open(files[-1]))
    hits = {k: v for k, v in recs.items() if k.split("<")[0].split("/")[0] == kernel.split(" ")[0]}
    return hits or None


def prefill(eng, gen):
    """ring full (2^20 transitions) with SURVEY.md 8(d)'s synthetic distribution, priorities U(0,1)^0.6"""
    N = 1 << LOG2N
    chunk = 1 << 16
    dev = eng.device
    for k in range(0, N, chunk):
        s = torch.randn(chunk, D, device=dev, generator=gen)
        s2 = torch.randn(chunk, D, device=dev, generator=gen)
        a = torch.randint(0, A, (chunk,), device=dev, generator=gen, dtype=torch.int32)
        d = torch.rand(chunk, device=dev, generator=gen) < P_DONE
        r = torch.randn(chunk, device=dev, generator=gen)
        sign = torch.where(torch.rand(chunk, device=dev, generator=gen) < 0.5, -100.0, 100.0)
        r = torch.where(d, sign, r)
        eng.replay_add(s, a, r, s2, d)
    for k in range(0, N, chunk):
        idx = torch.arange(k, k + chunk, device=dev, dtype=torch.int32)
        pr = torch.rand(chunk, device=dev, generator=gen).clamp_min(1e-4) ** 0.6
        eng.per_set(idx, pr)


def host_threads():
    """threads the CPU legs may use: CPUs this process may run on, capped by the cgroup CPU quota (a GPU box gives one
    GPU's share of a 256-core host)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def _cpu_port_world():
    """ring 2^20 + tree filled with the bench distribution, a learner on it (the plain-C oracle; checker code only)"""
    sys.path[:0] = [p for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")) if p not in sys.path]
    import _oracle as oc
    from _oracle import onp
    dims = (D, H1, H2, A)
    N = 1 << LOG2N
    rng = np.random.default_rng(0)
    rb, per = oc.CReplay(N, D), oc.CPer(LOG2N)
    chunk = 1 << 18
    for k in range(0, N, chunk):
        s = rng.standard_normal((chunk, D), dtype=np.float32); s2 = rng.standard_normal((chunk, D), dtype=np.float32)
        a = rng.integers(0, A, chunk).astype(np.int32); d = rng.random(chunk) < P_DONE
        r = np.where(d, np.where(rng.random(chunk) < 0.5, -100.0, 100.0), rng.standard_normal(chunk)).astype(np.float32)
        slots = rb.add(s, a, r, s2, d)
        per.set(slots, (np.maximum(rng.random(chunk), 1e-4) ** 0.6).astype(np.float32))
    lrn = oc.CLearner(dims, oc.Opt(2e-4, 0.9, 0.999, 1e-8, 1e-4, 1), 0.99, B, rb, per, onp.init_params(dims, 0), 0)
    obs = rng.standard_normal((N_ENVS, D), dtype=np.float32)
    return oc, lrn, obs


def _time_steps(step, seconds):
    """warm up twice, then whole steps until `seconds` have passed (at least 3): (steps, elapsed)"""
    step(); step()
    n, t0 = 0, time.perf_counter()
    while True:
        step(); n += 1
        dt = time.perf_counter() - t0
        if (dt >= seconds and n >= 3) or n >= 20000:
            return n, dt


def _torch_cpu_leg(seconds, threads):
    """the same step in PyTorch-CPU (autograd + torch.optim.AdamW; SURVEY.md 8(d) "CPU-torch"): proportional PER as
    cumsum + searchsorted over the 2^20 priorities (what a torch user would write; no sum-tree), stratified draws"""
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    N = 1 << LOG2N
    P0 = init_params(D * H1 + H1 + H1 * H2 + H2 + H2 + 1 + H2 * A + A)
    shapes = ((D, H1), (H1,), (H1, H2), (H2,), (H2, 1), (1,), (H2, A), (A,))
    ps, o = [], 0
    for sh in shapes:
        n = int(np.prod(sh)); ps.append(P0[o:o + n].view(sh).clone().requires_grad_(True)); o += n
    tg = [p.detach().clone() for p in ps]
    opt = torch.optim.AdamW(ps, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)

    def fwd(p, x):                           # LunarLander/dddqn.py:24-31
        h = torch.relu(torch.relu(x @ p[0] + p[1]) @ p[2] + p[3])
        v, adv = h @ p[4] + p[5], h @ p[6] + p[7]
        return v + adv - adv.mean(1, keepdim=True)
    S = torch.randn(N, D, generator=g); S2 = torch.randn(N, D, generator=g)
    Aa = torch.randint(0, A, (N,), generator=g); R = torch.randn(N, generator=g); Dn = (torch.rand(N, generator=g) < P_DONE).float()
    prio = torch.rand(N, generator=g).clamp_min(1e-4) ** 0.6
    obs = torch.randn(N_ENVS, D, generator=g)
    state = {"c": 0, "pmax": 1.0}

    def step():
        nonlocal obs
        with torch.no_grad():
            for _ in range(TRAIN_FREQ):      # q_agent.py:176-183
                a = torch.where(torch.rand(N_ENVS, generator=g) > 0.15, fwd(ps, obs).argmax(1), torch.randint(0, A, (N_ENVS,), generator=g))
                nxt = torch.randn(N_ENVS, D, generator=g); dn = (torch.rand(N_ENVS, generator=g) < P_DONE).float()
                sl = (torch.arange(N_ENVS) + state["c"]) % N
                S[sl] = obs; S2[sl] = nxt; Aa[sl] = a; R[sl] = torch.randn(N_ENVS, generator=g); Dn[sl] = dn; prio[sl] = state["pmax"]
                state["c"] += N_ENVS; obs = nxt
            cs = torch.cumsum(prio, 0)       # proportional PER, stratified
            u = (torch.arange(B) + torch.rand(B, generator=g)) * (cs[-1] / B)
            idx = torch.searchsorted(cs, u).clamp_max(N - 1)
            w = (N * prio[idx] / cs[-1]) ** -0.4; w = w / w.max()
            s, s2, a, r, d = S[idx], S2[idx], Aa[idx], R[idx], Dn[idx]
            q, nq, nt = fwd(ps, s), fwd(ps, s2), fwd(tg, s2)       # q_learning_functions.py:52-60
            delta = r + (1 - d) * (0.99 * nt.gather(1, nq.argmax(1, keepdim=True)).squeeze(1) - q.gather(1, a[:, None]).squeeze(1))
            targets = q + delta[:, None] * torch.nn.functional.one_hot(a, A)
        opt.zero_grad(set_to_none=True)
        loss = (w * torch.nn.functional.huber_loss(fwd(ps, s), targets, reduction="none", delta=1.0).sum(1)).mean()   # :35-36
        loss.backward(); opt.step()          # :23-25
        with torch.no_grad():
            prio[idx] = (delta.abs() + 1e-6) ** 0.6
            state["pmax"] = max(state["pmax"], float(prio[idx].max()))
    n, dt = _time_steps(step, seconds)
    return {"value": n / dt, "unit": "grad-updates/sec", "cores": threads, "kind": "port",
            "sample": f"{n} steps in {dt:.1f} s, PyTorch-CPU {torch.__version__} autograd + torch.optim.AdamW, "
                      f"PER by cumsum + searchsorted, {threads} threads"}


def cpu_baseline(seconds=float(os.environ.get("DQN_BENCH_CPU_SECONDS", "8"))):
    """CPU restatements of the same step timed on this host on a BOUNDED sample (about `seconds` each):
    the plain-C oracle on all cores this process may use (OpenMP, oracle/dqn_oracle_omp.c -- the headline `value`),
    the same code on one thread, and a PyTorch-CPU version. kind "port": the reference's JAX path cannot run here."""
    threads = host_threads()
    oc, lrn, obs = _cpu_port_world()
    env = {"c": 0}

    def step_1t():
        for _ in range(TRAIN_FREQ):
            env["c"] = lrn.actor_step(obs, 0.15, P_DONE, env["c"])
        lrn.update(B)

    def step_omp():
        for _ in range(TRAIN_FREQ):
            env["c"] = lrn.actor_step_omp(obs, 0.15, P_DONE, env["c"])
        lrn.update_omp(B)

    n1, dt1 = _time_steps(step_1t, seconds)
    oc.lib().orc_omp_set_threads(threads)
    nt_, dtt = _time_steps(step_omp, seconds)
    what = "steps (4x256 env-steps + 1 update of B=1024 each) of the same workload"
    out = {"value": nt_ / dtt, "unit": "grad-updates/sec", "cores": threads, "kind": "port",
           "sample": f"{nt_} {what} in {dtt:.1f} s, plain-C oracle with OpenMP (oracle/dqn_oracle_*.c, gcc -O3 -mavx2 -mfma "
                     f"-fopenmp; bit-identical to its 1-thread form), {threads} threads = the CPUs this process may use of "
                     f"{os.cpu_count()} host cores",
           "env_steps_per_sec": nt_ * N_ENVS * TRAIN_FREQ / dtt,
           "one_thread": {"value": n1 / dt1, "unit": "grad-updates/sec", "cores": 1, "kind": "port",
                          "sample": f"{n1} {what} in {dt1:.1f} s, the same C code on 1 thread"}}
    try:
        out["torch_cpu"] = _torch_cpu_leg(seconds, threads)
    except Exception as ex:                  # noqa: BLE001 -- a baseline leg must not lose the GPU line
        out["torch_cpu"] = {"error": repr(ex)}
    return out


def quick_rate(dq, precision, rank, world, steps, obs_wrapper=False):
    """secondary measurement: the same step loop on another precision mode (single GPU only); obs_wrapper: the reference's
    own observation shape -- D + 1 = 9 columns, the last one ObsWrapper's step / max_steps (LunarLander/env.py:19-24,
    max_steps 1 500 as Test/lunar_lander.py:29), kept by the actor kernel"""
    L = dq._lib
    Dd = D + 1 if obs_wrapper else D
    eng = dq.Engine(dq.EngineConfig(obs_dim=Dd, hidden1=H1, hidden2=H2, num_actions=A, capacity=1 << LOG2N, use_per=True,
                                    max_batch=B, optimizer="adamw", lr=2e-4, gamma=0.99, seed=1000 + rank,
                                    world_size=world, precision=precision))
    gen = torch.Generator(device=eng.device); gen.manual_seed(1234 + rank)
    eng.set_params(init_params(eng.param_count)); eng.sync_target()
    if obs_wrapper:
        N = 1 << LOG2N
        for k in range(0, N, 1 << 16):
            n = 1 << 16
            s = torch.randn(n, Dd, device=eng.device, generator=gen); s[:, Dd - 1] = torch.rand(n, device=eng.device, generator=gen)
            eng.replay_add(s, torch.randint(0, A, (n,), device=eng.device, generator=gen, dtype=torch.int32), torch.randn(n, device=eng.device, generator=gen),
                           s.roll(1, 0), torch.rand(n, device=eng.device, generator=gen) < P_DONE)
        eng.env_config("synthetic", 1500, 1.0); eng.env_time_feature(True)
        obs0 = torch.randn(N_ENVS, Dd, device=eng.device, generator=gen); obs0[:, Dd - 1] = 0.0
        eng.env_reset(obs0, P_DONE)
    else:
        prefill(eng, gen)
        eng.env_reset(torch.randn(N_ENVS, D, device=eng.device, generator=gen), P_DONE)
    eng.set_epsilon(0.15)
    st = eng.stream
    n = steps // ITERS_PER_GRAPH
    with torch.cuda.stream(st):
        for _ in range(max(n // 4, 2)):
            eng.train_iters(ITERS_PER_GRAPH, TRAIN_FREQ, B, st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            eng.train_iters(ITERS_PER_GRAPH, TRAIN_FREQ, B, st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    loss = float(eng.last_loss().item())
    eng.close()
    if obs_wrapper:
        return {"dtype": precision, "value": n * ITERS_PER_GRAPH / dt, "unit": "grad-updates/sec", "ms_per_step": dt / (n * ITERS_PER_GRAPH) * 1e3,
                "obs_dim": Dd, "final_loss": loss,
                "note": "the reference's own shape: 8 observations + ObsWrapper's step / max_steps column (env.py:19-24), kept inside "
                        "the one-launch actor kernel; same loop as the headline otherwise"}
    return {"dtype": precision, "value": n * ITERS_PER_GRAPH / dt, "unit": "grad-updates/sec",
            "ms_per_step": dt / (n * ITERS_PER_GRAPH) * 1e3, "env_steps_per_sec": n * ITERS_PER_GRAPH * N_ENVS * TRAIN_FREQ / dt,
            "steps": n * ITERS_PER_GRAPH, "final_loss": loss,
            "note": "bf16 MFMA operands, f32 accumulate / master weights; tolerance 2e-2 of scale (tests/test_gpu_bf16.py); "
                    "the headline value is the f32 path that meets the 1e-5 parity bar"}


def init_params(n_params):
    """haiku-style init (TruncatedNormal(1/sqrt(fan_in)) cut at 2 sigma, b = 0), identical on every rank"""
    P0 = torch.empty(n_params)
    g0 = torch.Generator().manual_seed(0)
    o = 0
    for (k, n) in ((D, H1), (H1, H2), (H2, 1), (H2, A)):
        w = torch.empty(k * n); torch.nn.init.trunc_normal_(w, std=1.0 / k ** 0.5, a=-2.0 / k ** 0.5, b=2.0 / k ** 0.5, generator=g0)
        P0[o:o + k * n] = w; o += k * n
        P0[o:o + n] = 0; o += n
    return P0


def visible_gpu_count():
    """GPUs this process would see, WITHOUT touching the HIP / ROCr runtime (the parent of the rank processes must not initialise
    the GPU): the KFD topology in sysfs (nodes with SIMDs are GPUs), narrowed by the *_VISIBLE_DEVICES lists the launcher set."""
    n = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(root):
            try:
                props = dict(ln.split()[:2] for ln in open(os.path.join(root, node, "properties")) if len(ln.split()) >= 2)
                n += int(props.get("simd_count", "0")) > 0
            except OSError:
                pass
    except OSError:
        n = 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([x for x in v.split(",") if x.strip() != ""])
            n = min(n, listed) if n else listed
    return n


def cnn_dp_line(dq, rank, world, dist):
    """BASELINE configs[4] in its multi-GPU form: every rank a full CNN learner on its own minibatch of 512 frame
    stacks, ONE gradient exchange per update on the handle's own RCCL communicator -- the 6.4 MB fc leaf on the communicator's own stream
    beside the rest of the backward, the 0.3 MB of small leaves behind it -- AdamW with grad_scale = 1 / world. The communicator
    is first checked against torch.distributed's all-reduce; any failure on any rank drops the line on every rank."""
    Bc, A_ = 512, 6
    flop = 2 * (400 * 32 * 256 + 81 * 64 * 512 + 49 * 64 * 576 + 3136 * 512 + 512 * 7)
    bwd = 2 * (2 * (81 * 64 * 512 + 49 * 64 * 576 + 3136 * 512 + 512 * 7) + 400 * 32 * 256)
    ok, e = 1, None
    try:
        e = dq.CnnEngine(num_actions=A_, max_batch=Bc, precision="bf16")
        P = torch.randn(e.param_count, generator=torch.Generator().manual_seed(3)) * 0.02          # same on every rank
        e.set_params(P); e.set_params(P, target=True)
        e.comm_init_native()
        g = e.buffer("grad")
        probe = (torch.arange(e.param_count, device=e.device, dtype=torch.float32) % 251) * (1.0 + rank)
        g.copy_(probe); e.allreduce_grads(); torch.cuda.synchronize()
        want = probe.clone(); dist.all_reduce(want)
        ok = int(torch.allclose(g, want, rtol=1e-6, atol=0.0) and e.comm_ranks() == world)
    except Exception as ex:                                   # noqa: BLE001
        print(f"[bench] CNN communicator unavailable on rank {rank}: {ex}", file=sys.stderr)
        ok = 0
    flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if e is not None:
            e.close()
        return {}
    gen = torch.Generator(device="cuda"); gen.manual_seed(70 + rank)
    fr = torch.randint(0, 256, (Bc, 84, 84, 4), dtype=torch.uint8, device="cuda", generator=gen)
    fr2 = torch.randint(0, 256, (Bc, 84, 84, 4), dtype=torch.uint8, device="cuda", generator=gen)
    act = torch.randint(0, A_, (Bc,), dtype=torch.int32, device="cuda", generator=gen)
    r = torch.randn(Bc, device="cuda", generator=gen); d = (torch.rand(Bc, device="cuda", generator=gen) < 0.05).float()
    for _ in range(5):
        e.update(fr, act, r, fr2, d)
    reps = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); dist.barrier(); e0.record()
        for _ in range(10):
            e.update(fr, act, r, fr2, d)
        e1.record(); e1.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) * 1e3 / 10], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        reps.append(float(t.item()))
    same = e.get_buffer("params")
    lo, hi = same.clone(), same.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    identical = bool(torch.equal(lo, hi))
    e.close()
    us = float(np.median(reps))
    fl = (3 * flop + bwd) * Bc * world
    return {f"cnn_update_dp_B{Bc}_bf16": {"bound": "mfma", "avg_us": us, "launches_per_step": 0, "achieved": fl / us / 1e6,
                                          "peak": MFMA_BF16_PEAK_TFLOPS * world, "unit": "TFLOP/s", "frac": fl / us / 1e6 / (MFMA_BF16_PEAK_TFLOPS * world),
                                          "traffic": None, "updates_per_sec_all_ranks": world * 1e6 / us, "replicas_identical": identical,
                                          "allreduce_bytes_per_rank": 4 * 1687719,
                                          "note": "BASELINE configs[4] per-GPU learners: Agent._step on 512 frame stacks per rank, gradient all-reduce on the "
                                                  "handle's RCCL communicator (fc leaf beside the backward), MAX over ranks, median of 5 regions of 10"}}


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher environment: start the N rank processes ourselves, as CHILDREN created
    before this process has made any GPU call (never an exec of a process that touched the GPU), relay rank 0's single
    JSON line and the children's exit status."""
    import socket
    import subprocess
    if os.environ.get("DQN_BENCH_SPAWN_SELFTEST") != "1":
        have = visible_gpu_count()                            # sysfs / environment only: no HIP or ROCr call in the parent
        if have < args.gpus:
            print(f"[bench] --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
            return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:                                     # rank 0 prints exactly one JSON line
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr)
    rc = proc.wait()
    if line is None:
        print(f"[bench] the rank processes printed no result line (exit status {rc})", file=sys.stderr)
        return rc or 4
    print(line, flush=True)
    return rc


def spawn_selftest(args, json_fd):
    """CPU rehearsal of the launcher path (tests/test_host.py): gloo ranks, no GPU work, one JSON line from rank 0"""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(4) * (dist.get_rank() + 1)
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        os.write(json_fd, (json.dumps({"metric": "grad-updates/sec", "value": 0.0, "n_gpus": dist.get_world_size(),
                                       "steps": args.steps, "warmup": args.warmup, "selftest": True,
                                       "allreduce_sum": float(t[0])}) + "\n").encode())
    dist.barrier(); dist.destroy_process_group()


def per_sample_lines(dq, rank):
    """the stand-alone PER-sample kernel (dqn_per_sample: stratified descent + gather) timed live with
    HIP events at the bench shape (B = 1024: its latency floor) and at large B (its bandwidth slope); ring 2^20, D = 8"""
    eng = dq.Engine(dq.EngineConfig(obs_dim=D, hidden1=16, hidden2=16, num_actions=A, capacity=1 << LOG2N, use_per=True,
                                    max_batch=1 << 20, seed=77 + rank))
    gen = torch.Generator(device=eng.device); gen.manual_seed(99)
    prefill(eng, gen)
    out = {}
    st = eng.stream
    with torch.cuda.stream(st):
        for lb in (10, 16, 18, 20):
            Bs = 1 << lb
            bufs = eng._batch_out(Bs) + (eng.empty((Bs,), torch.int32), eng.empty((Bs,), torch.float32))
            ms = []
            for it in range(12):
                eng.profile_begin(st)
                eng.per_sample_into(Bs, 0.4, 1, it, bufs)
                ms += [m for n, m in eng.profile_end(st) if n == "per_sample"]
            us = float(np.median(ms[2:])) * 1e3
            alg = (4 * LOG2N + 16 * D + 26) * Bs
            out[f"per_sample_B{Bs}"] = {"bound": "hbm", "avg_us": us, "launches_per_step": 0, "achieved": alg / us / 1e3,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / us / 1e3 / HBM_PEAK_GBS,
                                        "traffic": pmc_traffic(f"per_sample_B{Bs}"), "algorithmic_bytes": alg,
                                        "note": ("latency floor at the bench batch" if lb == 10 else "sweep point (not the bench batch)")
                                                + "; stand-alone kernel k_per_sample2 -- inside the bench step the batch is drawn by k_actor's sampler workgroups"}
    eng.close()
    return out


def cnn_lines(dq):
    """BASELINE.json configs[4] (PongNoFrameskip-v4 shape, 512 envs): the Nature-CNN dueling Q-net (dqn_cnn.hip) on 512 frame
    stacks -- the forward (five launches) and one whole Agent._step on a given minibatch (dqn_cnn_update: three forwards, TD
    rule, loss gradient through the CNN, AdamW, shadow refresh), HIP events around back-to-back calls (median of 5 regions of 10)"""
    out = {}
    Bc, A_, flop = 512, 6, 2 * (400 * 32 * 256 + 81 * 64 * 512 + 49 * 64 * 576 + 3136 * 512 + 512 * 7)
    bwd = 2 * (2 * (81 * 64 * 512 + 49 * 64 * 576 + 3136 * 512 + 512 * 7) + 400 * 32 * 256)      # dX and dW per layer; conv1 has no dX
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    frames = torch.randint(0, 256, (Bc, 84, 84, 4), dtype=torch.uint8, device="cuda", generator=g)
    frames2 = torch.randint(0, 256, (Bc, 84, 84, 4), dtype=torch.uint8, device="cuda", generator=g)
    act = torch.randint(0, A_, (Bc,), dtype=torch.int32, device="cuda", generator=g)
    r = torch.randn(Bc, device="cuda", generator=g); d = (torch.rand(Bc, device="cuda", generator=g) < 0.05).float()
    for prec, peak in (("bf16", MFMA_BF16_PEAK_TFLOPS), ("f32", MFMA_F32_PEAK_TFLOPS)):
        e = dq.CnnEngine(num_actions=A_, max_batch=Bc, precision=prec)
        P = torch.randn(e.param_count) * 0.02
        e.set_params(P); e.set_params(P, target=True)
        q = torch.empty((Bc, A_), dtype=torch.float32, device="cuda")
        for name, fn, fl, note in (
                (f"cnn_fwd_B{Bc}_{prec}", lambda: e.forward(frames, out=q), flop,
                 "BASELINE configs[4] forward: 5 launches, launch gaps included"),
                (f"cnn_update_B{Bc}_{prec}", lambda: e.update(frames, act, r, frames2, d), 3 * flop + bwd,
                 "BASELINE configs[4] Agent._step on a given minibatch: 3 forwards + TD + backward (3 dX, 4 dW) + reduce + AdamW, ~27 launches on two streams")):
            for _ in range(5):
                fn()
            reps = []
            for _ in range(5):                      # median of 5 regions of 10 calls (a first region was seen 3x slow once: allocation transients)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(10):
                    fn()
                e1.record(); e1.synchronize()
                reps.append(e0.elapsed_time(e1) * 1e3 / 10)
            us = float(np.median(reps))
            out[name] = {"bound": "mfma", "avg_us": us, "launches_per_step": 0, "achieved": fl * Bc / us / 1e6, "peak": peak,
                         "unit": "TFLOP/s", "frac": fl * Bc / us / 1e6 / peak, "traffic": None, "note": note}
        e.close()
    # the loop around it: 512 synthetic frame-stack envs, 4 vector env steps (act, add to the frame ring and the PER index) per
    # update of 512 PER-sampled transitions (General/QLearning/cnn_agent.py); the synthetic env (Philox frames, on the device) is inside
    from deep_q_learning_amd.General.QLearning.cnn_agent import CnnVectorAgent
    ag = CnnVectorAgent(n_envs=Bc, num_actions=A_, capacity=1 << 14, batch_size=Bc, precision="bf16", train_frequency=4, seed=5, n_step=3)
    ag.init_params(torch.randn(ag.cnn.param_count) * 0.02)
    ag.training(12)                    # past the first lap of the ring (32 vector steps): every kernel of the loop has run once
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n_it = 10
    ag.training(n_it)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n_it
    fl = (4 * flop + 3 * flop + bwd) * Bc
    out[f"cnn_loop_{Bc}envs_bf16"] = {"bound": "mfma", "avg_us": us, "launches_per_step": 0, "achieved": fl / us / 1e6, "peak": MFMA_BF16_PEAK_TFLOPS,
                                      "unit": "TFLOP/s", "frac": fl / us / 1e6 / MFMA_BF16_PEAK_TFLOPS, "traffic": None,
                                      "updates_per_sec": 1e6 / us, "env_steps_per_sec": 4 * Bc * 1e6 / us, "device_errors": ag.index.device_errors(),
                                      "note": "BASELINE configs[4] shape on ONE GPU, n-step 3 PER: 4 vector env steps of 512 synthetic frame-stack envs (CNN act + "
                                              "frame-ring add + PER index add) + 1 update from the ring (PER sample, gather, 3 forwards, backward, AdamW, priority write-back); "
                                              "host-driven loop (no graph); r03: the synthetic env lives on the device (dqn_cnn_env_step_synth: forward 3 launches -- the three "
                                              "convolutions are one persistent kernel -- + 1 policy / transition / ring kernel per vector step, + 1 index launch: "
                                              "dqn_per_index_step), 43 launches per iteration"}
    ag.close()
    return out


def per_sample_child():
    """per_sample_lines in a child process of its own (started, not exec'ed). The sampler at B = 2^20 is sensitive to where
    the handle's arena lands: the FIRST device allocation of a process gathers at 37-39 us, a handle created after another
    one (alive or freed, torch cache emptied or not) at 54-56 us (tools/diag/ps_repro.py) -- so it is measured the way a
    long-lived learner process would see it, as that process's first handle; the in-process figure is kept beside it."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--per-sample-only"], capture_output=True, text=True, timeout=600)
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            return json.loads(ln)
    print(f"[bench] per-sample child failed (exit {r.returncode}): {r.stderr[-500:]}", file=sys.stderr)
    return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=7, help="timed regions of exactly --steps steps; value = their median")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=50)
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary bf16 measurement and the PER-sample lines")
    ap.add_argument("--per-sample-only", action="store_true", help=argparse.SUPPRESS)   # child mode: the stand-alone sampler lines only
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                    help="f32 = exact f32 MFMA (meets the 1e-5 parity bar; default); bf16 = bf16 MFMA throughput path")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))                          # nothing above or inside has touched the GPU
    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a version banner on communicator creation)
    # write to file descriptor 1 directly, so fd 1 is pointed at stderr for the whole run and the line goes to the saved fd
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")
    if os.environ.get("DQN_BENCH_SPAWN_SELFTEST") == "1":
        return spawn_selftest(args, json_fd)
    torch.cuda.set_device(local_rank)
    if args.per_sample_only:
        import deep_q_learning_amd as dq
        os.write(json_fd, (json.dumps(per_sample_lines(dq, rank)) + "\n").encode())
        return
    import torch.distributed as dist
    # DQN_BENCH_FORCE_DP=1: take the multi-GPU code path with one rank (plumbing rehearsal on a 1-GPU box; a one-rank
    # all-reduce moves nothing, so this says nothing about the collective itself)
    force_dp = os.environ.get("DQN_BENCH_FORCE_DP") == "1"
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dp = world > 1 or force_dp

    import deep_q_learning_amd as dq
    L = dq._lib
    cfg = dq.EngineConfig(obs_dim=D, hidden1=H1, hidden2=H2, num_actions=A, capacity=1 << LOG2N, use_per=True,
                          max_batch=B, optimizer="adamw", lr=2e-4, gamma=0.99, seed=1000 + rank, world_size=world,
                          precision=args.precision)
    eng = dq.Engine(cfg)
    gen = torch.Generator(device=eng.device); gen.manual_seed(1234 + rank)
    P0 = init_params(eng.param_count)
    eng.set_params(P0); eng.set_params(P0, L.BUF_TARGET)
    prefill(eng, gen)
    eng.env_reset(torch.randn(N_ENVS, D, device=eng.device, generator=gen), P_DONE)
    eng.set_epsilon(0.15)                                    # MIN_EPSILON, Test/lunar_lander.py:33
    grad = eng.buffer(L.BUF_GRAD) if dp else None
    if dp:                                                   # replicas must start bit-identical
        flat = eng.get_params()
        dist.broadcast(flat, src=0)
        eng.set_params(flat); eng.set_params(flat, L.BUF_TARGET)
    st = eng.stream

    # N > 1, preferred path: the handle's own RCCL communicator, its all-reduce captured inside the inner-loop graph
    # (ITERS_PER_GRAPH iterations per launch, as on one GPU). Checked once against torch.distributed's all-reduce;
    # any failure on any rank sends every rank to the eager torch.distributed path.
    native, rccl_ranks = False, None
    if dp and os.environ.get("DQN_BENCH_DP", "native") == "native":
        ok = 1
        try:
            eng.comm_init_native()
            rccl_ranks = eng.comm_ranks()
            probe = (torch.arange(eng.param_count, device=eng.device, dtype=torch.float32) % 251) * (1.0 + rank)
            grad.copy_(probe)
            with torch.cuda.stream(st):
                eng.allreduce_grads_native(st)
            torch.cuda.synchronize()
            want = probe.clone(); dist.all_reduce(want)
            ok = int(torch.allclose(grad, want, rtol=1e-6, atol=0.0) and rccl_ranks == world)
        except Exception as ex:                               # noqa: BLE001 -- fall back, report below
            print(f"[bench] native RCCL path unavailable on rank {rank}: {ex}", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], device=eng.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        native = bool(flag.item())
        if native:                                            # the captured form once, guarded: a capture / instantiate failure
            ok = 1                                            # (same software on every rank => on every rank) also falls back
            try:
                with torch.cuda.stream(st):
                    eng.train_iters(ITERS_PER_GRAPH, TRAIN_FREQ, B, st)
                torch.cuda.synchronize()
            except Exception as ex:                           # noqa: BLE001
                print(f"[bench] captured all-reduce unavailable on rank {rank}: {ex}", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], device=eng.device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native = bool(flag.item())
    if dp and rccl_ranks is None:
        rccl_ranks = dist.get_world_size()

    def run_steps(k):
        """exactly k steps; a step = TRAIN_FREQ vector env steps + one update"""
        if dp and not native:
            for _ in range(k):
                eng.actor_backward(TRAIN_FREQ, B, st)        # 4 vector env steps + sample..grads, one graph launch
                dist.all_reduce(grad)                        # RCCL, sum; /world is inside the optimizer
                eng.update_apply(B, st)                      # optimizer, one graph launch
        else:
            for _ in range(k // ITERS_PER_GRAPH):
                eng.train_iters(ITERS_PER_GRAPH, TRAIN_FREQ, B, st)
            if k % ITERS_PER_GRAPH:
                eng.train_iters(k % ITERS_PER_GRAPH, TRAIN_FREQ, B, st)

    def barrier():
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        """one timed region: barrier + synchronize on both sides; HIP events on the launch stream bracket the work on the
        GPU, perf_counter the host's view; both as the MAX over ranks"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record(st)
        fn()
        e1.record(st)
        barrier()
        wall = time.perf_counter() - t0
        t = torch.tensor([e0.elapsed_time(e1) * 1e-3, wall], device=eng.device, dtype=torch.float64)
        if dp:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].item()), float(t[1].item())

    with torch.cuda.stream(st):
        run_steps(args.warmup)
        run_steps(args.steps)                                # also instantiates every graph shape used below
        reps = [timed(lambda: run_steps(args.steps)) for _ in range(max(args.repeats, 1))]
        ev = sorted(r[0] for r in reps); wl = sorted(r[1] for r in reps)
        dt, dt_wall = ev[len(ev) // 2], wl[len(wl) // 2]

        # update-only and actor-only rates (same run, extra information)
        def upd_only(k):
            if dp and not native:
                for _ in range(k):
                    eng.update_backward(B, st); dist.all_reduce(grad); eng.update_apply(B, st)
            else:
                for _ in range(max(k // ITERS_PER_GRAPH, 1)):
                    eng.train_iters(ITERS_PER_GRAPH, 0, B, st)
        upd_only(ITERS_PER_GRAPH)
        n_upd = args.steps if (dp and not native) else max(args.steps // ITERS_PER_GRAPH, 1) * ITERS_PER_GRAPH
        dt_upd = sorted(timed(lambda: upd_only(args.steps))[0] for _ in range(3))[1]
        eng.actor_step(st)
        dt_act = sorted(timed(lambda: [eng.actor_step(st) for _ in range(args.steps)])[0] for _ in range(3))[1]

        # live per-kernel timing with HIP events on the launch stream (rank 0 of N=1 is enough)
        kern = {}
        if rank == 0:
            for _ in range(args.profile_steps):
                eng.profile_begin(st)
                if dp and not native:
                    eng.actor_backward(TRAIN_FREQ, B, st); eng.update_apply(B, st)
                else:
                    eng.train_iters(1, TRAIN_FREQ, B, st)    # profiling mode: eager, one launch per event pair
                for name, ms in eng.profile_end(st):
                    kern.setdefault(name, []).append(ms)
    loss = float(eng.last_loss().item())
    assert np.isfinite(loss), "non-finite loss"
    dev_err = eng.device_errors()
    assert dev_err == 0, f"in-kernel hand-over timed out {dev_err} time(s)"

    # N > 1: the CNN learners' data-parallel update (every rank takes part; rank 0 reports it under "kernels")
    cnn_dp = {}
    if world > 1 and not args.no_secondary and os.environ.get("DQN_BENCH_CNN_DP", "1") == "1":
        try:
            cnn_dp = cnn_dp_line(dq, rank, world, dist)
        except Exception as ex:                               # noqa: BLE001 -- the headline line must still be printed
            print(f"[bench] cnn_update_dp failed on rank {rank}: {ex}", file=sys.stderr)
            cnn_dp = {}

    if rank == 0:
        per_step = {}
        for name, v in kern.items():
            launches = TRAIN_FREQ if name in ("act_fwd_policy", "env_step_add", "actor_step") and not dp else 1
            if name == "actor_step" and dp:
                launches = TRAIN_FREQ
            bound, units = algorithmic_cost(name, LOG2N, eng.param_count)
            avg_ms = float(np.median(v))
            ach = units / (avg_ms * 1e-3) / (1e9 if bound == "hbm" else 1e12) if avg_ms > 0 else 0.0
            peak = HBM_PEAK_GBS if bound == "hbm" else (MFMA_F32_PEAK_TFLOPS if args.precision == "f32" else MFMA_BF16_PEAK_TFLOPS)
            per_step[name] = {"bound": bound, "avg_us": avg_ms * 1e3, "launches_per_step": launches, "traffic": pmc_traffic(name),
                              "achieved": ach, "peak": peak, "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                              "frac": ach / peak}
        # dominant KERNEL = largest time share per step. k_qnet_fwd is launched in two shapes (4 actor launches of
        # 256 rows + 1 three-pass launch of 3x1024 rows): its roofline entry aggregates all five launches of a step.
        groups = {"k_qnet_fwd": [k for k in ("actor_step", "qnet_fwd_x3", "sample_fwd_x3", "sample_fwd_x3_bwd") if k in per_step]}
        fwd_shapes = " + ".join(f"{per_step[k]['launches_per_step']} x {k}" for k in groups["k_qnet_fwd"])
        for k in per_step:
            if k not in groups["k_qnet_fwd"]:
                groups[k] = [k]
        gtime = {g: sum(per_step[k]["avg_us"] * per_step[k]["launches_per_step"] for k in ks) for g, ks in groups.items() if ks}
        dom = max(gtime, key=gtime.get)
        ks = groups[dom]
        bound = per_step[ks[0]]["bound"]; peak = per_step[ks[0]]["peak"]
        units = sum(algorithmic_cost(k, LOG2N, eng.param_count)[1] * per_step[k]["launches_per_step"] for k in ks)
        nl = sum(per_step[k]["launches_per_step"] for k in ks)
        ach = units / (gtime[dom] * 1e-6) / (1e9 if bound == "hbm" else 1e12)
        tr = [pmc_traffic(k) for k in ks]
        roof = {"bound": bound, "achieved": ach, "peak": peak, "unit": per_step[ks[0]]["unit"], "frac": ach / peak,
                "traffic": (sum(t * per_step[k]["launches_per_step"] for t, k in zip(tr, ks)) / nl) if all(t is not None for t in tr) else None,
                "kernel": KERNEL_OF.get(dom, dom) + (f" (per step: {fwd_shapes}; per-shape lines under \"kernels\")" if dom == "k_qnet_fwd" else ""),
                "avg_us": gtime[dom] / nl, "launches_per_step": nl,
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), profiles/r*_pmc.json, bytes per launch",
                "mfma_counters": mfma_counters(dom),
                "timing": "hipExtLaunchKernelGGL start/stop events of each eager launch on the launch stream, median of "
                          f"{args.profile_steps}; rocprofv3 summary of the same command in profiles/"}
        out = {
            "metric": "grad-updates/sec", "value": world * args.steps / dt, "unit": "grad-updates/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "repeats": len(reps), "min": world * args.steps / ev[-1], "max": world * args.steps / ev[0],
            "timing": f"median of {len(reps)} timed regions of exactly {args.steps} steps each, barrier + synchronize on both "
                      "sides, HIP events on the launch stream, MAX over ranks; host perf_counter view of the same regions "
                      "under wall_clock",
            "wall_clock": {"value": world * args.steps / dt_wall, "ms_per_step": dt_wall / args.steps * 1e3,
                           "min": world * args.steps / wl[-1], "max": world * args.steps / wl[0]},
            "config": {"workload": "LunarLander-v2 shape, 256 vectorised synthetic envs, PER batch=1024, 2x256 dueling MLP "
                                   "(BASELINE.json configs[1]); step = 4 vector env steps (1024 env-steps) + 1 grad update",
                       "obs_dim": D, "num_actions": A, "hidden": [H1, H2], "batch": B, "global_batch": B * world,
                       "n_envs_per_gpu": N_ENVS, "replay_capacity": 1 << LOG2N, "per": True, "optimizer": "adamw",
                       "train_frequency": TRAIN_FREQ, "parallelism": f"dp{world} independent learners + grad all-reduce",
                       "allreduce": ("none (1 GPU)" if not dp else "RCCL, captured in the inner-loop graph" if native
                                     else "torch.distributed (RCCL), eager between two graph launches")},
            "rccl_ranks": rccl_ranks,
            "env_steps_per_sec": world * args.steps * N_ENVS * TRAIN_FREQ / dt,
            "update_only_per_sec": world * n_upd / dt_upd,
            "actor_only_env_steps_per_sec": world * args.steps * N_ENVS / dt_act,
            "final_loss": loss,
            "roofline": roof,
            "kernels": per_step,
        }
        if cnn_dp:
            out["kernels"].update(cnn_dp)
        if world == 1 and not dp and args.precision == "f32" and not args.no_secondary:
            eng.close()
            out["bf16"] = quick_rate(dq, "bf16", rank, world, max(args.steps // 2, 10 * ITERS_PER_GRAPH))
            out["obs_wrapper_d9"] = quick_rate(dq, "f32", rank, world, max(args.steps // 2, 10 * ITERS_PER_GRAPH), obs_wrapper=True)
            out["kernels"].update(per_sample_child())
            out["kernels"].update(cnn_lines(dq))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dp:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
