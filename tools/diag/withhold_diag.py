import sys, os, numpy as np, torch
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import deep_q_learning_amd as dq
L = dq._lib
dims = (8, 256, 256, 4); B, N = 1024, 1 << 12
rng = np.random.default_rng(21)
rows = (rng.standard_normal((2048, 8)), rng.integers(0, 4, 2048), rng.standard_normal(2048), rng.standard_normal((2048, 8)), rng.random(2048) < 0.05)
e = dq.Engine(dq.EngineConfig(obs_dim=8, hidden1=256, hidden2=256, num_actions=4, capacity=N, use_per=False, max_batch=B, seed=9))
P0 = (np.random.default_rng(1).standard_normal(e.param_count) * 0.05).astype(np.float32)
e.set_params(P0); e.sync_target(); e.replay_add(*rows)
def rep(tag):
    torch.cuda.synchronize()
    fin = lambda w: bool(np.isfinite(e.get_params(w, host=True)).all())
    print(tag, "loss", float(e.last_loss().item()), "errs", e.device_errors(), "params", fin(L.BUF_PARAMS), "mu", fin(L.BUF_MU), "nu", fin(L.BUF_NU), "grad", fin(L.BUF_GRAD))
with torch.cuda.stream(e.stream):
    e.update(B); rep("normal")
    e.debug_withhold_handover(True)
    e.update(B); rep("withheld")
    e.debug_withhold_handover(False); e.clear_device_errors()
    rep("cleared")
    e.update(B); rep("after1")
    e.set_params(P0); e.sync_target()
    e.update(B); rep("after2")
