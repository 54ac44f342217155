"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/dqn_hip.h declares, the
host-side mirror behaves like the reference where that is checkable without a device, the oracle reproduces
the committed golden vectors, and the N>1 data-parallel recipe is exercised with gloo at world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import _oracle as oc
from _oracle import onp
from test_oracle import CFGS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


# --------------------------------------------------------------------------- C ABI
def header_functions():
    src = open(os.path.join(ROOT, "include", "dqn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dqn_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import deep_q_learning_amd as dq
    lib = dq._lib.load()                                     # binds all of SIGNATURES / OTHER or raises
    declared = header_functions()
    assert len(declared) >= 35
    bound = set(dq._lib.SIGNATURES) | set(dq._lib.OTHER)
    assert set(declared) == bound, (set(declared) ^ bound)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.dqn_abi_version() == dq._lib.ABI_VERSION == 4
    c = dq._lib.DqnConfig()
    lib.dqn_default_config(C.byref(c))                       # Test/lunar_lander.py:23-48 defaults
    assert (c.obs_dim, c.hidden1, c.hidden2, c.num_actions, c.capacity, c.max_batch) == (9, 32, 64, 4, 100000, 64)
    assert abs(c.lr - 2e-4) < 1e-9 and abs(c.gamma - 0.99) < 1e-7 and c.optimizer == dq._lib.OPT_ADAMW
    out = subprocess.check_output(["nm", "-D", "--defined-only", dq._lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (dqn_[a-z_0-9]+)", out))
    assert set(declared) <= exported


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment (how the driver starts it) must itself start 2 rank
    processes as children, rendezvous on 127.0.0.1, and relay rank 0's single JSON line with exit status 0. CPU rehearsal
    of that path (DQN_BENCH_SPAWN_SELFTEST=1: gloo ranks, no GPU work)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DQN_BENCH_SPAWN_SELFTEST"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["allreduce_sum"] == 3.0


def test_bench_refuses_more_gpus_than_visible():
    if torch.cuda.device_count() >= 2:
        pytest.skip("2+ GPUs visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DQN_BENCH_SPAWN_SELFTEST")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr and not r.stdout.strip()


def test_no_silent_fallback_without_gpu():
    import deep_q_learning_amd as dq
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dq.Engine(dq.EngineConfig())


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "deep-q-learning_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"oracle_np|liboracle|dqn_oracle|import _oracle|from oracle", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


# ------------------------------------------------------------------ host-side mirror
def test_param_tree_roundtrip_and_names():
    from deep_q_learning_amd._tree import NAMES, dims_of, flatten, shapes, unflatten
    dims = CFGS["cfg1"]
    P = torch.tensor(onp.init_params(dims, 0))
    tree = unflatten(P, dims)
    assert list(tree) == list(NAMES) and tree["model/~/linear"]["w"].shape == (9, 32)      # haiku: w is [in,out]
    assert tree["model/~/linear_2"]["w"].shape == (64, 1) and tree["model/~/linear_3"]["b"].shape == (4,)
    assert flatten(tree) is P and dims_of(tree) == dims
    plain = {k: dict(v) for k, v in tree.items()}
    assert torch.equal(flatten(plain), P)
    assert [s for _, _, s in shapes(dims)] == [s for _, _, s in onp.param_shapes(*dims)]


def test_obs_wrapper_matches_reference_semantics():
    """LunarLander/env.py:19-31"""
    from deep_q_learning_amd.LunarLander.env import ObsWrapper

    class Env:
        def __init__(self): self.t = 0
        def reset(self): self.t = 0; return np.arange(8, dtype=np.float64)
        def step(self, a): self.t += 1; return np.full(8, self.t, np.float64), 1.5, self.t == 3, {}
    w = ObsWrapper(Env(), 1500)
    o = w.reset()
    assert o.shape == (1, 9) and o.dtype == np.float32 and o[0, 8] == 0.0
    o, r, d, _ = w.step(0)
    assert o[0, 8] == np.float32(1 / 1500) and r == 1.5 and not d
    assert np.array_equal(o, onp.obs_augment(np.full((1, 8), 1.0), np.array([1]), 1500))
    w.step(0); w.reset()
    assert w._step == 0


def test_optimizer_state_shapes_like_optax():
    from deep_q_learning_amd import optim
    from deep_q_learning_amd._tree import unflatten
    tree = unflatten(torch.zeros(onp.param_count(*CFGS["cfg1"])), CFGS["cfg1"])
    st = optim.adamw(2e-4).init(tree)
    assert len(st) == 3 and st[0].count == 0 and st[1] == optim.EmptyState() and optim.adamw(1e-3).weight_decay == 1e-4
    assert len(optim.adam(1e-4).init(tree)) == 2 and optim.adam(1e-4).weight_decay == 0.0
    assert st[0].mu["model/~/linear_1"]["w"].shape == (32, 64)


def test_param_agent_inject_mirrors_the_reference():
    """General/QLearning/hyperparameter_optimization.py:76-91: inject rebinds the seven searched hyper-parameters; as in the
    reference the q-target closure built in the constructor keeps its gamma unless rebuild_closures is asked for"""
    from deep_q_learning_amd.General.QLearning.hyperparameter_optimization import ParamAgent
    a = ParamAgent.__new__(ParamAgent)                      # no device needed for the attribute semantics
    a._compute_q_targets = sentinel = object()
    a.inject(0.95, 0.8, 0.97, 0.05, 30, 48, 3)
    assert (a._gamma, a._epsilon, a._epsilon_decay_rate, a._min_epsilon, a._replace_frequency, a._batch_size, a._train_frequency) == \
        (0.95, 0.8, 0.97, 0.05, 30, 48, 3)
    assert a._compute_q_targets is sentinel
    a.max_episodes = 7
    assert a.max_episodes == 7 and a._max_episodes == 7


def test_transform_finds_the_model():
    from deep_q_learning_amd.LunarLander.dddqn import Model, transform, without_apply_rng
    m = without_apply_rng(transform(lambda *args: Model(4)(*args)))     # Test/lunar_lander.py:47
    assert m.dims(9) == (9, 32, 64, 4)
    assert transform(lambda *a: Model(2, hidden=(64, 64))(*a)).dims(4) == (4, 64, 64, 2)


# -------------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("fn", sorted(f for f in os.listdir(GOLD) if f.startswith("cfg")))
def test_oracles_reproduce_golden(fn):
    z = np.load(os.path.join(GOLD, fn), allow_pickle=False)
    dims = tuple(int(x) for x in z["dims"])
    k = int(z["sub"])
    full = onp.q_targets(z["P"], z["Pt"], z["s"], z["a"], z["r"], z["s2"], z["d"], 0.99, dims, np.float64, full=True)
    for key in ("q", "next_q", "next_q_tm", "delta", "targets"):
        assert np.array_equal(full[key], z[key]), key
    assert np.array_equal(full["astar"], z["astar"])
    g, L, dq = onp.grads(z["P"], z["s"], z["targets"].astype(np.float32), dims, None, np.float64)
    assert L == z["loss"] and np.array_equal(g[::k], z["grads"]) and g.sum() == z["grads_sum"]
    # the plain-C restatement (f32) against the same vectors, north_star tolerance
    c = oc.q_targets(dims, z["P"], z["Pt"], z["s"], z["a"], z["r"], z["s2"], z["d"], 0.99)
    assert np.array_equal(c["astar"], z["astar"])
    assert np.allclose(c["targets"], z["targets"], rtol=1e-5, atol=1e-4)
    gc, Lc, _ = oc.grads(dims, z["P"], z["s"], z["targets"].astype(np.float32))
    assert abs(Lc - z["loss"]) <= 1e-5 * max(1, abs(z["loss"]))
    assert np.max(np.abs(gc[::k] - z["grads"])) <= 2e-5 * max(np.abs(z["grads"]).max(), 1e-3)


@pytest.mark.parametrize("fn", ["per_L12.npz", "per_L16.npz"])
def test_sumtree_reproduces_golden(fn):
    z = np.load(os.path.join(GOLD, fn), allow_pickle=False)
    L, n, B, seed = int(z["L"]), int(z["n_add"]), int(z["B"]), int(z["seed"])
    ct = oc.CPer(L)
    ct.add(np.arange(n, dtype=np.int32)); ct.set(np.arange(n, dtype=np.int32), z["prio"])
    for it in range(3):
        idx, isw = ct.sample(n, B, 0.4 + 0.2 * it, seed, it)
        assert np.array_equal(idx, z[f"idx_{it}"])
        assert np.array_equal(isw.view(np.uint32), z[f"isw_{it}"].view(np.uint32))
        ct.update(idx, z[f"td_{it}"])
        assert ct.tree[1] == z[f"total_{it}"] and ct.pmax == z[f"pmax_{it}"]
    assert np.array_equal(ct.tree[:64], z["tree_top"])


# --------------------------------------------------------- N > 1 recipe on gloo, 2 ranks
def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    from deep_q_learning_amd import dist as dd
    r, w, _ = dd.init_from_env("gloo")
    dims = CFGS["cfg1"]
    P = torch.tensor(onp.init_params(dims, 0 if r == 0 else 99))
    dd.broadcast_params(P)                                        # replicas start identical
    P = P.numpy().copy()
    from test_oracle import make_batch
    mu, nu, cnt = np.zeros(P.size), np.zeros(P.size), 0
    Pw = P.astype(np.float64)
    for it in range(3):                                           # own minibatch per rank, one all-reduce per update
        s, a, rr, s2, d = make_batch(dims, 64, 1000 + 10 * it + r)
        rr = np.clip(rr, -3, 3)
        tg = onp.q_targets(Pw, P, s, a, rr, s2, d, 0.99, dims, np.float64)
        g, _, _ = onp.grads(Pw, s, tg, dims, None, np.float64)
        gt = torch.tensor(g)
        dd.allreduce_grads(gt)                                    # SUM; the optimizer divides by world
        Pw, mu, nu, cnt = onp.adam_step(Pw, gt.numpy(), mu, nu, cnt, 2e-4, dtype=np.float64, grad_scale=1.0 / w)
    t = dd.max_over_ranks(float(r + 1))
    q.put((r, Pw, t))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_recipe_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1])                   # replicas stay bit-identical
    assert res[0][2] == res[1][2] == 2.0                          # max-over-ranks timing helper
    # equals one process applying the mean of the two per-rank gradients
    dims = CFGS["cfg1"]
    from test_oracle import make_batch
    P0 = onp.init_params(dims, 0)
    Pw, mu, nu, cnt = P0.astype(np.float64), np.zeros(P0.size), np.zeros(P0.size), 0
    for it in range(3):
        gs = []
        for r in range(2):
            s, a, rr, s2, d = make_batch(dims, 64, 1000 + 10 * it + r)
            rr = np.clip(rr, -3, 3)
            tg = onp.q_targets(Pw, P0, s, a, rr, s2, d, 0.99, dims, np.float64)
            gs.append(onp.grads(Pw, s, tg, dims, None, np.float64)[0])
        Pw, mu, nu, cnt = onp.adam_step(Pw, gs[0] + gs[1], mu, nu, cnt, 2e-4, dtype=np.float64, grad_scale=0.5)
    assert np.allclose(res[0][1], Pw, rtol=0, atol=1e-12)


def _cnn_dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    from deep_q_learning_amd import dist as dd
    r, w, _ = dd.init_from_env("gloo")
    A, B = 6, 3
    P = torch.tensor(onp.cnn_init_params(A, 0 if r == 0 else 99))
    dd.broadcast_params(P)
    P = P.numpy().copy()
    rng = np.random.default_rng(500 + r)                          # own minibatch per rank
    frames = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
    qv, _ = oc.cnn_forward(P, frames, A)
    targets = (qv + rng.standard_normal((B, A)) * 0.7).astype(np.float32)
    g, _ = oc.cnn_grads(P, frames, targets, None, A)
    # the message plan of dqn_cnn_update with a communicator (csrc/dqn_cnn.hip): the fc weight leaf first (beside the backward),
    # then the range below it and the range above it -- three in-place SUM all-reduces that together cover the buffer once
    lo = 8 * 8 * 4 * 32 + 32 + 4 * 4 * 32 * 64 + 64 + 3 * 3 * 64 * 64 + 64
    hi = lo + 3136 * 512
    gt = torch.tensor(g)
    for seg in (gt[lo:hi], gt[:lo], gt[hi:]):
        dd.allreduce_grads(seg)
    Pn, _, _, cnt, _, _ = oc.adam_step(oc.Opt(1e-3, 0.9, 0.999, 1e-8, 1e-4, 1), P, gt.numpy(), np.zeros_like(P), np.zeros_like(P), 0, 1.0, 1.0, grad_scale=1.0 / w)
    q.put((r, Pn, gt.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_cnn_data_parallel_recipe_gloo_world2():
    """SURVEY 8(e) for the CNN (BASELINE configs[4] in its multi-GPU form): two learners with their own minibatches, the
    gradient summed in the three messages dqn_cnn_update issues with a communicator (fc leaf; below it; above it), AdamW with
    grad_scale = 1 / world -- the replicas stay bit-identical and equal ONE learner stepping on the sum of the two gradients.
    (CPU ranks on gloo with the restatement's gradients; on hardware the same plan runs on RCCL: unmeasured, no multi-GPU box.)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_cnn_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    A, B = 6, 3
    P = onp.cnn_init_params(A, 0)
    gs = []
    for r in range(2):
        rng = np.random.default_rng(500 + r)
        frames = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
        qv, _ = oc.cnn_forward(P, frames, A)
        targets = (qv + rng.standard_normal((B, A)) * 0.7).astype(np.float32)
        gs.append(oc.cnn_grads(P, frames, targets, None, A)[0])
    assert np.array_equal(res[0][2], gs[0] + gs[1])               # every element summed exactly once
    Pn, _, _, _, _, _ = oc.adam_step(oc.Opt(1e-3, 0.9, 0.999, 1e-8, 1e-4, 1), P, gs[0] + gs[1], np.zeros_like(P), np.zeros_like(P), 0, 1.0, 1.0, grad_scale=0.5)
    assert np.array_equal(res[0][1], Pn)


def test_cnn_oracle_reproduces_golden():
    """the Nature-CNN restatement (forward f32 / f64, loss gradient f32 / f64) against the committed vector
    tests/golden/cnn_B4_seed7.npz (regression pin; inputs re-generated from the recorded seeds and checked by their sums)"""
    sys.path.insert(0, GOLD)
    import make_cnn_golden as mk
    g = np.load(os.path.join(GOLD, "cnn_B4_seed7.npz"))
    A = int(g["A"])
    P, frames, noise, scale, isw = mk.inputs(int(g["seed"]), int(g["B"]))
    assert P.astype(np.float64).sum() == float(g["param_sum"]) and int(frames.astype(np.int64).sum()) == int(g["frame_sum"])
    q32, feat = oc.cnn_forward(P, frames, A)
    assert np.array_equal(q32, g["q32"]) and feat.astype(np.float64).sum() == float(g["feat32_sum"])
    assert np.allclose(onp.cnn_forward(P, frames, A, np.float64), g["q64"], rtol=1e-12, atol=1e-12)
    targets = (q32 + noise * scale).astype(np.float32)
    assert np.array_equal(targets, g["targets"])
    g64, l64 = oc.cnn_grads(P, frames, targets, isw, A, f64=True)
    g32, l32 = oc.cnn_grads(P, frames, targets, isw, A)
    assert abs(l64 - float(g["loss64"])) <= 1e-12 and np.float32(l32) == g["loss32"]
    assert np.allclose(g64[::997], g["grad64_strided"], rtol=1e-10, atol=1e-14) and np.array_equal(g32[::997], g["grad32_strided"])
    o = 0
    for n, s_, a_ in zip(mk.LEAVES, g["leaf_sums"], g["leaf_abs"]):
        assert np.isclose(g64[o:o + n].sum(), s_, rtol=1e-9, atol=1e-13) and np.isclose(np.abs(g64[o:o + n]).sum(), a_, rtol=1e-10)
        o += n


def test_isa_scan_flags_inflight_atomic_results_and_sources_have_none():
    """a returning atomic written as inline asm hands the compiler a result register that is not valid until
    the atomic returns. tools/isa_scan.py must flag a destination VGPR touched before the next `s_waitcnt vmcnt(0)` (the
    two snippets are the shape of the r02 miscompile and of its fix), and the kernels must not contain such an atomic at
    all (the tickets are compiler-tracked now: ticket_take_early, csrc/dqn_device.h). `make -C csrc check` runs the scan
    over the real ISA (2 min of hipcc; not part of this suite)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_scan
    bad = """
	;;#ASMSTART
	global_atomic_add v82, v[86:87], v82, off sc0
	;;#ASMEND
	v_accvgpr_write_b32 a74, v82
	v_bfe_u32 v82, v114, 4, 2
	s_waitcnt vmcnt(0)
	v_mov_b32 v1, v82
""".split("\n")
    got = isa_scan.inflight_atomic_hazards(bad)
    assert len(got) == 2 and "v_accvgpr_write_b32" in got[0][1] and "v_bfe_u32" in got[1][1]
    in_range = "\t;;#ASMSTART\n\tglobal_atomic_add v10, v[2:3], v4, off sc0\n\t;;#ASMEND\n\tglobal_load_dwordx4 v[8:11], v[2:3], off\n".split("\n")
    assert len(isa_scan.inflight_atomic_hazards(in_range)) == 1
    good = """
	;;#ASMSTART
	global_atomic_add v82, v[86:87], v82, off sc0
	;;#ASMEND
	v_mov_b32 v83, v84
	s_waitcnt vmcnt(0)
	v_accvgpr_write_b32 a74, v82
	global_atomic_add v5, v[86:87], v7, off sc0
	s_waitcnt vmcnt(0)
	v_mov_b32 v1, v5
""".split("\n")
    assert isa_scan.inflight_atomic_hazards(good) == []                 # waited for; compiler-issued atomics are tracked
    csrc = os.path.join(ROOT, "deep-q-learning_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            for m in re.finditer(r"asm\s+volatile\s*\(\s*\"[^;]*?_atomic_[^;]*?;", open(os.path.join(csrc, f)).read(), re.S):
                assert not re.search(r"\b(sc0|glc)\b", m.group(0)), (f, m.group(0)[:120])


def test_vector_agent_inject_gamma_semantics_match_param_agent():
    """the two mirrors of ParamAgent.inject (hyperparameter_optimization.py:76-91) agree -- an injected gamma is
    an attribute only (the reference's jitted closure keeps the constructor's), unless rebuild_closures=True is asked for."""
    from deep_q_learning_amd.General.QLearning.vector_agent import VectorAgent

    class Eng:
        class cfg: max_batch = 64
        def __init__(self): self.gammas, self.eps = [], []
        def set_gamma(self, g): self.gammas.append(g)
        def set_epsilon(self, x): self.eps.append(x)
    a = VectorAgent.__new__(VectorAgent)
    a.e = Eng()
    a.inject(0.95, 0.8, 0.97, 0.05, 30, 48, 3)
    assert a.e.gammas == [] and a.gamma == 0.95 and a.e.eps == [0.8] and (a.B, a.train_frequency, a.replace_frequency) == (48, 3, 30)
    a.inject(0.9, 0.8, 0.97, 0.05, 30, 48, 3, rebuild_closures=True)
    assert a.e.gammas == [0.9]
    with pytest.raises(ValueError):
        a.inject(0.9, 0.8, 0.97, 0.05, 30, 65, 3)


# ------------------------------------------------- the reference's pickled fixture, read without unpickling
class ScaleByAdamState(__import__("typing").NamedTuple):       # the shape of optax's state classes, for the file the test writes
    count: object
    mu: object
    nu: object


class EmptyState(__import__("typing").NamedTuple):
    pass


def test_pickle_token_reader_on_own_files(tmp_path):
    """General/Base/pickle_tokens.py on pickles THIS test writes (protocol 4, numpy arrays in a haiku-shaped dict, and an
    adam-shaped state: count, mu leaves, nu leaves): keys, shapes and bytes come back exactly, and generate_loading
    (utils.py:32-40) resumes from such a directory. Nothing is unpickled on the read side."""
    import pickle
    from deep_q_learning_amd.General.Base import pickle_tokens as pt
    from deep_q_learning_amd.General.Base.utils import generate_loading
    from deep_q_learning_amd._tree import NAMES, shapes
    dims = (9, 32, 64, 4)
    rng = np.random.default_rng(0)
    tree = {}
    for mod, leaf, shp in shapes(dims):
        tree.setdefault(mod, {})[leaf] = rng.standard_normal(shp).astype(np.float32)
    mu = {m: {k: rng.standard_normal(tree[m][k].shape).astype(np.float32) for k in ("b", "w")} for m in NAMES}   # jax order: b, w
    nu = {m: {k: np.abs(rng.standard_normal(tree[m][k].shape)).astype(np.float32) for k in ("b", "w")} for m in NAMES}
    with open(tmp_path / "params.pickle", "wb") as f:
        pickle.dump(tree, f, protocol=4)
    with open(tmp_path / "opt_state.pickle", "wb") as f:
        pickle.dump((ScaleByAdamState(np.array(7, np.int32), mu, nu), EmptyState(), EmptyState()), f, protocol=4)
    got = pt.read_haiku_params(str(tmp_path / "params.pickle"))
    assert list(got) == list(NAMES)
    for m in NAMES:
        for k in ("w", "b"):
            assert got[m][k].dtype == np.float32 and np.array_equal(got[m][k], tree[m][k])
    count, gmu, gnu, n_empty = pt.read_adam_state(str(tmp_path / "opt_state.pickle"))
    assert count == 7 and n_empty == 2
    assert all(np.array_equal(gmu[m][k], mu[m][k]) and np.array_equal(gnu[m][k], nu[m][k]) for m in NAMES for k in ("w", "b"))
    params, st = generate_loading(str(tmp_path), device=torch.device("cpu"))()
    assert params["model/~/linear_1"]["w"].shape == (32, 64) and st[0].count == 7 and len(st) == 3
    assert np.array_equal(params["model/~/linear_3"]["w"].numpy(), tree["model/~/linear_3"]["w"])
    assert np.array_equal(st[0].nu["model/~/linear"]["b"].numpy(), nu["model/~/linear"]["b"])
    with open(tmp_path / "bad.pickle", "wb") as f:
        pickle.dump({"model/~/linear": {"w": np.zeros((2, 2), np.float64)}}, f, protocol=4)
    with pytest.raises(ValueError):
        pt.read_haiku_params(str(tmp_path / "bad.pickle"))


def test_reference_init_fixture():
    """tests/golden/ref_init_params.npz = the arrays inside the reference's Test/lunar_lander/{params,opt_state}.pickle
    (tests/golden/extract_ref_init.py). What the reference's own fixture pins about this path: the haiku tree of
    LunarLander/dddqn.py:19-22 at D = 9 (8 obs + the ObsWrapper column, env.py:19-24) -- leaf names, [in, out] weight
    layout, hk.Linear's default init (TruncatedNormal(1/sqrt(fan_in)) cut at 2 sigma, zero biases) that oracle_np.init_params
    restates -- and optax.adamw's initial state (count 0, zero moments). Where the reference is present (build container)
    the committed arrays are re-read from its files."""
    z = np.load(os.path.join(GOLD, "ref_init_params.npz"), allow_pickle=False)
    want = {("", "w"): (9, 32), ("", "b"): (32,), ("_1", "w"): (32, 64), ("_1", "b"): (64,), ("_2", "w"): (64, 1), ("_2", "b"): (1,),
            ("_3", "w"): (64, 4), ("_3", "b"): (4,)}
    for (sfx, leaf), shp in want.items():
        a = z[f"model/~/linear{sfx}/{leaf}"]
        assert a.shape == shp and a.dtype == np.float32
        if leaf == "b":
            assert not a.any()
        else:
            fan_in = shp[0]
            assert np.abs(a).max() <= 2.0 / np.sqrt(fan_in) + 1e-6 and 0.6 / np.sqrt(fan_in) < a.std() < 1.0 / np.sqrt(fan_in)
        assert not z[f"opt/mu/model/~/linear{sfx}/{leaf}"].any() and not z[f"opt/nu/model/~/linear{sfx}/{leaf}"].any()
    assert int(z["opt/count"][0]) == 0
    # the restatement's initialiser draws from the same family (same bounds; std of a 2-sigma truncated normal = 0.88 sigma)
    P = onp.init_params((9, 32, 64, 4), 0)
    w1 = P[:9 * 32]
    assert np.abs(w1).max() <= 2.0 / 3.0 + 1e-6 and not P[9 * 32:9 * 32 + 32].any()
    ref_dir = "/root/reference/Test/lunar_lander"
    if os.path.isdir(ref_dir):
        from deep_q_learning_amd.General.Base import pickle_tokens as pt
        tree = pt.read_haiku_params(os.path.join(ref_dir, "params.pickle"))
        assert list(tree) == ["model/~/linear", "model/~/linear_1", "model/~/linear_2", "model/~/linear_3"]
        for m, leaves in tree.items():
            for k, v in leaves.items():
                assert np.array_equal(v, z[f"{m}/{k}"])
        count, mu, nu, n_empty = pt.read_adam_state(os.path.join(ref_dir, "opt_state.pickle"))
        assert count == 0 and n_empty == 2                      # optax.adamw = chain(scale_by_adam, add_decayed_weights, scale)
