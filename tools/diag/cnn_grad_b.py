import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import deep_q_learning_amd as dq
import _oracle as oc
from test_gpu_cnn import make_params, host, A, LEAVES, grad_case, leaf_errors
for B in [int(x) for x in sys.argv[1:]]:
    P, frames, targets, isw = grad_case(B, 300 + B)
    Po, o = P.copy(), 0
    for name, n in LEAVES[:8]:
        Po[o:o + n] = 1.0 if name.endswith(".b") else Po[o:o + n] * 0.1
        o += n
    qo = oc.cnn_forward(Po, frames, A)[0]
    tg = (qo + (targets - oc.cnn_forward(P, frames, A)[0])).astype(np.float32)
    g64, l64 = oc.cnn_grads(Po, frames, tg, isw, A, f64=True)
    for prec in ("f32", "bf16"):
        e = dq.CnnEngine(num_actions=A, max_batch=B, precision=prec)
        e.set_params(Po)
        l = e.grads(frames, tg, isw); g = host(e.get_buffer("grad"))
        print(B, prec, "loss", l, l64, {k: float("%.2g" % v[0]) for k, v in leaf_errors(g, g64).items()}, flush=True)
        e.close()
