"""diagnostic: phase stamps of block 0 of k_cnn_trunk16 (needs libdqn_hip_stamps.so: make -C deep-q-learning_amd/csrc stamps)"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DQN_HIP_LIB", os.path.join(ROOT, "deep-q-learning_amd", "libdqn_hip_stamps.so"))
import deep_q_learning_amd as dq
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
e = dq.CnnEngine(num_actions=6, max_batch=B, precision="bf16")
e.set_params(torch.randn(e.param_count) * 0.02)
frames = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device=e.device)
for _ in range(10): e.forward(frames)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(e.stream if hasattr(e, "stream") else torch.cuda.current_stream()):
    st = e.stream if hasattr(e, "stream") else torch.cuda.current_stream()
    e0.record(st)
    for _ in range(20): e.forward(frames)
    e1.record(st)
e1.synchronize()
print("forward wall us (trunk + fc + heads):", round(e0.elapsed_time(e1) * 1e3 / 20, 1))
buf = (C.c_ulonglong * (8 * 64 * 2))()
assert e.lib.dqn_debug_stamps(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64, 2).astype(np.int64)[7]
names = {0: "kernel start", 1: "loop top (weights requested)", 2: "image 0 landed + barrier", 3: "convert 0 + barrier + dma 1", 4: "conv1 image 0 (wave 0: 4 tiles)", 5: "image 1 landed, convert, barrier, dma next", 6: "conv1 image 1", 7: "barrier", 8: "conv2 (3 tiles)", 9: "barrier", 10: "conv3 (2 tiles)", 11: "  conv2 tile 0 MFMA loop", 12: "  conv2 tile 0 next fill + epilogue", 13: "  conv2 tile 1 MFMA loop", 14: "  conv2 tile 1 epilogue", 15: "  conv2 tile 2 MFMA loop", 16: "  conv2 tile 2 epilogue", 20: "[block 0 start, same launch]", 30: "[block 0 conv3 end]"}
order = sorted((int(t[k, 1]), k) for k in names if t[k, 0])
prev = order[0][0]
for cyc, k in order:
    print(f"{names[k]:44s} {(t[k,0]-t[order[0][1],0]):9d} cyc   {(t[k,1]-t[order[0][1],1])*10} ns")
    prev = cyc
