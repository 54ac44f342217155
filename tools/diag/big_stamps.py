"""diagnostic: phase stamps of block 0 of k_big_rows (needs libdqn_hip_stamps.so: make -C deep-q-learning_amd/csrc stamps)"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DQN_HIP_LIB", os.path.join(ROOT, "deep-q-learning_amd", "libdqn_hip_stamps.so"))
import bench, deep_q_learning_amd as dq
B = 1 << 15
PREC = sys.argv[1] if len(sys.argv) > 1 else "f32"
eng = dq.Engine(dq.EngineConfig(obs_dim=bench.D, hidden1=256, hidden2=256, num_actions=4, capacity=1 << 16, use_per=True, max_batch=B, seed=1, precision=PREC))
eng.set_params(torch.randn(eng.param_count) * 0.05); eng.sync_target()
x = torch.randn(B, bench.D, device=eng.device)
MODE = sys.argv[2] if len(sys.argv) > 2 else "fwd"
if MODE == "update":
    gen = torch.Generator(device=eng.device); gen.manual_seed(0)
    rng = torch.randn(1 << 16, bench.D, device=eng.device)
    eng.replay_add(rng, torch.zeros(1 << 16, dtype=torch.int32, device=eng.device), torch.randn(1 << 16, device=eng.device), rng, torch.zeros(1 << 16, dtype=torch.bool, device=eng.device))
with torch.cuda.stream(eng.stream):
    for _ in range(20):
        if MODE == "update": eng.update(B)
        else: eng.forward(x)
    eng.stream.synchronize()
buf = (C.c_ulonglong * (8 * 64 * 2))()
assert eng.lib.dqn_debug_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64, 2).astype(np.int64)
t = st[1]
names = {0: "kernel start", 1: "p0 x staged+barrier", 2: "p0 L1 mfma", 3: "p0 L1 epi", 4: "p0 barrier", 5: "p0 L2 mfma", 6: "p0 h1-read barrier", 7: "p0 L2 epi", 8: "p0 barrier", 9: "p0 heads+Q", 10: "p0 barrier",
         21: "pL x staged+barrier", 22: "pL L1 mfma", 23: "pL L1 epi(stash)", 24: "pL barrier", 25: "pL L2 mfma", 26: "pL h1-read barrier", 27: "pL L2 epi(stash)", 28: "pL barrier", 29: "pL heads+Q", 30: "pL barrier",
         12: "bwd start", 13: "l3 zero+barrier", 14: "TD rows (thread 0)", 15: "barrier", 16: "loss/colsum/dz3 out, gates req", 17: "dz3.WH^T + barrier", 18: "dz2 epi", 19: "barrier", 20: "dz2.W2^T", 31: "dz1 epi"}
order = sorted((int(t[k, 0]), k) for k in names if t[k, 0])
prev = order[0][0]
for cyc, k in order:
    print(f"{names[k]:34s} {cyc - order[0][0]:9d} cyc  (+{cyc - prev:6d})")
    prev = cyc
sys.exit(0)
with torch.cuda.stream(eng.stream):
    for _ in range(20):
        eng.forward(x)
    eng.stream.synchronize()
buf = (C.c_ulonglong * (8 * 64 * 2))()
assert eng.lib.dqn_debug_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64, 2).astype(np.int64)
labels = ["start", "x staged, barrier", "L1 mfma done", "L1 epilogue done", "barrier", "L2 mfma done", "h1-read barrier", "L2 epilogue done", "barrier", "heads done", "barrier", "Q done, barrier"]
if PREC == "bf16":
    labels = ["start", "x staged, barrier", "L1 mfma done", "L1 epilogue done", "barrier", "L2 mfma done", "h1-read barrier", "L2 epilogue done", "barrier", "heads + Q done", "barrier"]
t = st[1, :len(labels)]
for i, lab in enumerate(labels):
    print(f"{lab:22s} +{int(t[i,0]-t[0,0]):8d} cyc  (+{int(t[i,0]-t[max(i-1,0),0]):7d})   {(t[i,1]-t[0,1])*10/1e3:8.2f} us")
