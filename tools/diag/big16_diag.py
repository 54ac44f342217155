import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import deep_q_learning_amd as dq
from _oracle import onp
from test_oracle import CFGS, make_batch
dims = CFGS["cfg2"]; B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
e = dq.Engine(dq.EngineConfig(obs_dim=8, hidden1=256, hidden2=256, num_actions=4, precision="bf16", flags=dq._lib.FLAG_BIG_ROWS, max_batch=B))
P = onp.init_params(dims, 3); e.set_params(P)
s, a, r, s2, d = make_batch(dims, B, 5)
t = onp.q_targets(P, P, s, a, r, s2, d, 0.99, dims, np.float64).astype(np.float32)
print("forward", float(e.forward(s).abs().max()), flush=True)
g, L = e.grads(s, t, None)
torch.cuda.synchronize()
print("grads ok", float(L.item()), float(g.abs().max()), flush=True)
g64, L64, _ = onp.grads(P, s, t, dims, None, np.float64)
gg = g.cpu().numpy()
print("cos", float(gg @ g64 / np.linalg.norm(gg) / np.linalg.norm(g64)), "L64", L64)
