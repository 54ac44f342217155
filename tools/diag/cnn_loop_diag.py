"""Where the CNN loop's time goes (host-synchronised phases):  python tools/diag/cnn_loop_diag.py [n_step]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deep_q_learning_amd as dq
from deep_q_learning_amd.General.QLearning.cnn_agent import CnnVectorAgent
n_step = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ag = CnnVectorAgent(n_envs=512, num_actions=6, capacity=1 << 14, batch_size=512, precision="bf16", train_frequency=4, seed=5, n_step=n_step)
ag.init_params(torch.randn(ag.cnn.param_count) * 0.02)
ag.training(12)                      # past the first wrap (32 steps)
def t(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("n_step", n_step, "env_step us", round(t(ag.env_step, 20), 1), "update us", round(t(ag.update, 10), 1))
import torch.profiler as P
parts = {}
def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); parts[name] = parts.get(name, 0) + (time.perf_counter() - t0) * 1e6; return r
for _ in range(10):
    (_, _, _, _, _), idx, isw = timed("per_sample", lambda: ag.index.per_sample(ag.B, ag.per_beta, ag.seed, ag.updates))
    timed("update_from_replay", lambda: ag.cnn.update_from_replay(idx, isw, ag.gamma, td_abs_out=ag.td_abs, n_step=ag.n_step, n_envs=ag.n_envs))
    timed("per_update_sorted", lambda: ag.index.per_update_sorted(idx, ag.td_abs))
    ag.updates += 1
print({k: round(v / 10, 1) for k, v in parts.items()})
ag.close()
for n in (1, 3):
    ag = CnnVectorAgent(n_envs=512, num_actions=6, capacity=1 << 14, batch_size=512, precision="bf16", train_frequency=4, seed=5, n_step=n)
    ag.init_params(torch.randn(ag.cnn.param_count) * 0.02)
    ag.training(3)
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ag.training(10); torch.cuda.synchronize()
        print("n_step", n, "rep", rep, "steps so far", ag.env_steps, "us per iteration", round((time.perf_counter() - t0) / 10 * 1e6, 1))
    ag.close()
