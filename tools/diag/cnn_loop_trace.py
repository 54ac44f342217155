"""diagnostic: the configs[4] loop (bench's cnn_loop_512envs_bf16) for a kernel trace -- run under tools/kt_loop.sh"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deep_q_learning_amd as dq
from deep_q_learning_amd.General.QLearning.cnn_agent import CnnVectorAgent
ag = CnnVectorAgent(n_envs=512, num_actions=6, capacity=1 << 14, batch_size=512, precision="bf16", train_frequency=4, seed=5, n_step=3)
ag.init_params(torch.randn(ag.cnn.param_count) * 0.02)
ag.training(12)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ag.training(10); e1.record(); e1.synchronize()
print("us per iteration", e0.elapsed_time(e1) * 100)
