#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (numpy f64 restatement, cross-checked against the plain-C
one and PyTorch autograd in tests/test_oracle.py). Run from the repo root:  python tests/golden/make_golden.py

PARITY UNPINNED: the reference holds no known-answer vectors, so these vectors pin the build against ITS OWN oracle
(regression), not against the reference's outputs. The one fixture the reference does hold -- its initial parameters and
zero optimizer state, Test/lunar_lander/*.pickle -- is the START of `cfg1_B64_refinit.npz` (r03): ref_init_params.npz, read
out of the pickles token by token by extract_ref_init.py (no unpickling). Fixtures are data only (inputs + expected outputs)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle_np as onp  # noqa: E402
from test_oracle import CFGS, make_batch  # noqa: E402

SUB = {"cfg1": 1, "cfg2": 23, "cfg3": 1}       # cfg2 keeps every 23rd element of P-sized outputs (+ checksums)


def ref_init_flat():
    """the reference's own initial parameters (Test/lunar_lander/params.pickle) in the flat haiku leaf order w b w b ..."""
    z = np.load(os.path.join(HERE, "ref_init_params.npz"), allow_pickle=False)
    return np.concatenate([z[f"model/~/linear{sfx}/{leaf}"].reshape(-1) for sfx in ("", "_1", "_2", "_3") for leaf in ("w", "b")]).astype(np.float32)


def one(name, B, seed, ref_init=False):
    dims = CFGS[name]
    if ref_init:                              # Agent.__init__: target_params = params (q_agent.py:88-91)
        P = ref_init_flat(); Pt = P.copy()
        assert P.size == onp.param_count(*dims)
    else:
        P = onp.init_params(dims, seed)
        P = (P + 0.05 * np.random.default_rng(seed + 100).standard_normal(P.size)).astype(np.float32)
        Pt = onp.init_params(dims, seed + 1)
    s, a, r, s2, d = make_batch(dims, B, seed + 2)
    r = np.clip(r, -3, 3) if seed % 2 else r
    full = onp.q_targets(P, Pt, s, a, r, s2, d, 0.99, dims, np.float64, full=True)
    targets32 = full["targets"].astype(np.float32)
    isw = np.random.default_rng(seed + 3).uniform(0.2, 1.0, B).astype(np.float32)
    g, L, dq = onp.grads(P, s, targets32, dims, None, np.float64)
    gw, Lw, _ = onp.grads(P, s, targets32, dims, isw, np.float64)
    out = dict(dims=np.array(dims), P=P, Pt=Pt, s=s, a=a, r=r, s2=s2, d=d, isw=isw, gamma=np.float64(0.99),
               q=full["q"], next_q=full["next_q"], next_q_tm=full["next_q_tm"], astar=full["astar"],
               delta=full["delta"], targets=full["targets"], loss=np.float64(L), loss_w=np.float64(Lw), dq=dq)
    k = SUB[name]
    out["sub"] = np.int64(k)
    out["grads"] = g[::k]; out["grads_sum"] = np.float64(g.sum()); out["grads_abs"] = np.float64(np.abs(g).sum())
    out["grads_w"] = gw[::k]
    for opt, lr in (("adamw", 2e-4), ("adam", 1e-4)):
        P64, mu, nu, cnt = P.astype(np.float64), np.zeros(P.size), np.zeros(P.size), 0
        for it in range(3):
            tg = onp.q_targets(P64, Pt, s, a, r, s2, d, 0.99, dims, np.float64).astype(np.float32)
            gg, _, _ = onp.grads(P64, s, tg, dims, None, np.float64)
            P64, mu, nu, cnt = onp.adam_step(P64, gg, mu, nu, cnt, lr, adamw=(opt == "adamw"), dtype=np.float64)
            if it in (0, 2):
                out[f"{opt}_params_{it + 1}"] = P64[::k]
                out[f"{opt}_params_{it + 1}_sum"] = np.float64(P64.sum())
        out[f"{opt}_mu_3"] = mu[::k]; out[f"{opt}_nu_3"] = nu[::k]
    return out


def per_case(L, n_add, B, seed):
    rng = np.random.default_rng(seed)
    t = onp.SumTree(L)
    t.add(np.arange(n_add))
    pr = (rng.random(n_add) + 0.01).astype(np.float32)
    t.set(np.arange(n_add), pr)
    out = dict(L=np.int64(L), n_add=np.int64(n_add), B=np.int64(B), prio=pr, seed=np.int64(seed))
    for it in range(3):
        idx, isw = t.sample(n_add, B, 0.4 + 0.2 * it, seed, it)
        td = (np.abs(rng.standard_normal(B)) * 2).astype(np.float32)
        t.update(idx, td)
        out[f"idx_{it}"] = idx; out[f"isw_{it}"] = isw; out[f"td_{it}"] = td
        out[f"total_{it}"] = t.tree[1].copy(); out[f"pmax_{it}"] = np.float32(t.pmax)
        out[f"leafsum_{it}"] = np.float64(t.tree[t.N:].astype(np.float64).sum())
    out["tree_top"] = t.tree[:64].copy()
    return out


def main():
    if "--ref-init-only" in sys.argv or "--all" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "cfg1_B64_refinit.npz"), **one("cfg1", 64, 6, ref_init=True))
        print("wrote cfg1 64 from the reference's initial parameters")
        if "--all" not in sys.argv:
            return
    for name, B, seed in (("cfg1", 64, 0), ("cfg1", 1024, 1), ("cfg2", 64, 2), ("cfg2", 1024, 3), ("cfg3", 64, 4), ("cfg3", 1024, 5)):
        np.savez_compressed(os.path.join(HERE, f"{name}_B{B}_seed{seed}.npz"), **one(name, B, seed))
        print("wrote", name, B, seed)
    np.savez_compressed(os.path.join(HERE, "per_L12.npz"), **per_case(12, 3000, 256, 7))
    np.savez_compressed(os.path.join(HERE, "per_L16.npz"), **per_case(16, 50000, 1024, 8))
    print("wrote per cases")


if __name__ == "__main__":
    main()
