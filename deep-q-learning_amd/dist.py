"""Data-parallel glue for independent per-GPU learners: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests). The ONLY exchange on the data path is one
in-place sum all-reduce of the flat f32 gradient per update; the division by world size happens inside the
optimizer kernel (dqn_config.world_size). The reference has no distributed code."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the launcher (torch.distributed.run). Returns (rank, world, local_rank)."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    return rank, world, local_rank


def allreduce_grads(grad: torch.Tensor):
    """in-place SUM over ranks of the flat gradient buffer (a torch view of DQN_BUF_GRAD, or any tensor)"""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM)
    return grad


def broadcast_params(flat: torch.Tensor, src=0):
    """make the replicated parameters bit-identical at start-up"""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def max_over_ranks(seconds: float, device=None) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
