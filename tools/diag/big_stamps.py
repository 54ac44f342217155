"""diagnostic: phase stamps of block 0 of k_big_rows (needs libdqn_hip_stamps.so: make -C deep-q-learning_amd/csrc stamps)"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DQN_HIP_LIB", os.path.join(ROOT, "deep-q-learning_amd", "libdqn_hip_stamps.so"))
import bench, deep_q_learning_amd as dq
B = 1 << 15
eng = dq.Engine(dq.EngineConfig(obs_dim=bench.D, hidden1=256, hidden2=256, num_actions=4, capacity=1 << 16, use_per=True, max_batch=B, seed=1))
eng.set_params(torch.randn(eng.param_count) * 0.05); eng.sync_target()
x = torch.randn(B, bench.D, device=eng.device)
with torch.cuda.stream(eng.stream):
    for _ in range(20):
        eng.forward(x)
    eng.stream.synchronize()
buf = (C.c_ulonglong * (8 * 64 * 2))()
assert eng.lib.dqn_debug_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64, 2).astype(np.int64)
labels = ["start", "x staged, barrier", "L1 mfma done", "L1 epilogue done", "barrier", "L2 mfma done", "h1-read barrier", "L2 epilogue done", "barrier", "heads done", "barrier", "Q done, barrier"]
t = st[1, :len(labels)]
for i, lab in enumerate(labels):
    print(f"{lab:22s} +{int(t[i,0]-t[0,0]):8d} cyc  (+{int(t[i,0]-t[max(i-1,0),0]):7d})   {(t[i,1]-t[0,1])*10/1e3:8.2f} us")
