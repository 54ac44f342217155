"""diagnostic: fused update at B=8200 (big path) vs oracle, per-iteration differences"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import deep_q_learning_amd as dq
import _oracle as oc
from _oracle import onp
from test_oracle import CFGS, make_batch
host = lambda t: t.detach().cpu().numpy()
dims = CFGS["cfg2"]; D, A = dims[0], dims[3]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8200
L_ = 12; N = 1 << L_
e = dq.Engine(dq.EngineConfig(obs_dim=D, hidden1=dims[1], hidden2=dims[2], num_actions=A, capacity=N, use_per=True, max_batch=B, seed=77, lr=1e-3))
cr = oc.CReplay(N, D); ct = oc.CPer(L_)
s, a, r, s2, d = make_batch(dims, 3000, 70, terminal_frac=0.1)
r = np.clip(r, -2, 2)
for k in range(0, 3000, 1000):
    sl = slice(k, k + 1000)
    ct.add(cr.add(s[sl], a[sl], r[sl], s2[sl], d[sl] > 0)); e.replay_add(s[sl], a[sl], r[sl], s2[sl], d[sl] > 0)
P = onp.init_params(dims, 71); P0 = (P + 0.05 * np.random.default_rng(171).standard_normal(P.size)).astype(np.float32)
e.set_params(P0); e.set_params(P0, dq._lib.BUF_TARGET)
lrn = oc.CLearner(dims, oc.Opt(1e-3, 0.9, 0.999, 1e-8, 1e-4, 1), 0.99, B, cr, ct, P0, 77, beta=0.4)
with torch.cuda.stream(e.stream):
    for it in range(4):
        Lc = lrn.update(B); e.update(B); e.stream.synchronize()
        Lg = host(e.last_loss())[0]
        idx = host(e.buffer(dq._lib.BUF_BATCH_IDX, torch.int32))[:B]
        ci = np.ctypeslib.as_array(lrn.l.idx, shape=(B,))
        td = host(e.buffer(dq._lib.BUF_BATCH_TD))[:B]
        cd = np.ctypeslib.as_array(lrn.l.delta, shape=(B,))      # |delta| after update
        isw = host(e.buffer(dq._lib.BUF_BATCH_ISW))[:B]; cw = np.ctypeslib.as_array(lrn.l.isw, shape=(B,))
        tree = host(e.buffer(dq._lib.BUF_TREE)); dt = np.abs(tree - ct.tree)
        print(it, "loss", Lg, Lc, "idx mismatches", int((idx != ci).sum()), "max|td|-diff", float(np.max(np.abs(np.abs(td) - cd))),
              "isw diff", float(np.max(np.abs(isw - cw))), "tree max diff", float(dt.max()), "at", int(dt.argmax()), "rel", float((dt / np.maximum(ct.tree, 1e-9)).max()),
              "param diff", float(np.max(np.abs(e.get_params(host=True) - lrn.params))), "errs", e.device_errors(), flush=True)
        if it == 1:
            e.sync_target(); lrn.sync_target()
