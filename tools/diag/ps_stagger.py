"""What decides the sampler's speed at B = 2^20 (35 us or 54 us)?  Output placement experiment (r02: 54 us in every variant
   here, also with the arena allocated physically contiguous or as scattered 2 MiB chunks in a temporary build; see
   ps_order.py for the one sequence that gives 35 us and DESIGN.md 4.2).
   python tools/diag/ps_stagger.py <mode> [gap_bytes]
   mode: sep  = every output tensor its own torch allocation (fresh process: its own hipMalloc)
         flat = all outputs carved from one torch allocation, tensor starts aligned to 256 B + gap"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, deep_q_learning_amd as dq
mode = sys.argv[1] if len(sys.argv) > 1 else "sep"
gap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = dq.Engine(dq.EngineConfig(obs_dim=8, hidden1=16, hidden2=16, num_actions=4, capacity=1 << 20, use_per=True, max_batch=1 << 20, seed=77))
gen = torch.Generator(device=eng.device); gen.manual_seed(99)
bench.prefill(eng, gen)
Bs = 1 << 20
sizes = [(Bs * 8 * 4, torch.float32, (Bs, 8)), (Bs * 4, torch.int32, (Bs,)), (Bs * 4, torch.float32, (Bs,)), (Bs * 8 * 4, torch.float32, (Bs, 8)), (Bs, torch.uint8, (Bs,)),
         (Bs * 4, torch.int32, (Bs,)), (Bs * 4, torch.float32, (Bs,))]
if mode == "sep":
    bufs = [torch.empty(shape, dtype=dt, device="cuda") for _, dt, shape in sizes]
else:
    flat = torch.empty(sum(s for s, _, _ in sizes) + 8 * (gap + 256), dtype=torch.uint8, device="cuda")
    bufs, off = [], 0
    for nbytes, dt, shape in sizes:
        bufs.append(flat[off:off + nbytes].view(dt).view(shape)); off += nbytes + gap
        off = (off + 255) // 256 * 256
print("  output addresses", [hex(b.data_ptr()) for b in bufs])
st = eng.stream
with torch.cuda.stream(st):
    ms = []
    for it in range(12):
        eng.profile_begin(st)
        eng.per_sample_into(Bs, 0.4, 1, it, tuple(bufs))
        ms += [m for n, m in eng.profile_end(st) if n == "per_sample"]
print("outputs", mode, "gap", gap, "-> us", round(float(np.median(ms[2:])) * 1e3, 1), flush=True)
eng.close()
