import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, deep_q_learning_amd as dq
def lines(tag):
    r = bench.per_sample_lines(dq, 0)
    print(tag, {k: round(v["avg_us"], 1) for k, v in r.items()}, flush=True)
lines("fresh process")
e = dq.Engine(dq.EngineConfig(obs_dim=8, hidden1=256, hidden2=256, num_actions=4, capacity=1 << 20, use_per=True, max_batch=1024, seed=1))
gen = torch.Generator(device=e.device); gen.manual_seed(1)
bench.prefill(e, gen); e.close()
lines("after another engine was created and closed")
e = dq.Engine(dq.EngineConfig(obs_dim=8, hidden1=256, hidden2=256, num_actions=4, capacity=1 << 20, use_per=True, max_batch=1024, seed=1))
bench.prefill(e, gen)
lines("while another engine is alive")
e.close()
torch.cuda.empty_cache()
lines("after torch.cuda.empty_cache()")
