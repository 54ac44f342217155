"""per_sample_lines' body with the list of batch sizes as an argument:  python tools/diag/ps_order.py 10,16,18,20 [variant.so]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, deep_q_learning_amd as dq
if len(sys.argv) > 2:          # a variant build of the library (tools/diag/_var/*.so: other PS_* macros)
    dq._lib.LIB_PATH = os.path.abspath(sys.argv[2])
lbs = [int(x) for x in sys.argv[1].split(",")]
eng = dq.Engine(dq.EngineConfig(obs_dim=8, hidden1=16, hidden2=16, num_actions=4, capacity=1 << 20, use_per=True, max_batch=1 << 20, seed=77))
gen = torch.Generator(device=eng.device); gen.manual_seed(99)
bench.prefill(eng, gen)
st = eng.stream
res = {}
with torch.cuda.stream(st):
    for lb in lbs:
        Bs = 1 << lb
        bufs = eng._batch_out(Bs) + (eng.empty((Bs,), torch.int32), eng.empty((Bs,), torch.float32))
        ms = []
        for it in range(12):
            eng.profile_begin(st)
            eng.per_sample_into(Bs, 0.4, 1, it, bufs)
            ms += [m for n, m in eng.profile_end(st) if n == "per_sample"]
        res[lb] = round(float(np.median(ms[2:])) * 1e3, 1)
        if lb == 20: print("  addresses", [hex(b.data_ptr()) for b in bufs])
print("order", lbs, "->", res, flush=True)
eng.close()
