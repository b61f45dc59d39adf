#!/usr/bin/env python3
"""One-rank check of the torch.distributed (nccl = RCCL) calls the round-parallel exchange makes (a 1-GPU box cannot host two
RCCL ranks): process group on the GPU, variable-size byte all-gather, barrier."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
from downpore_amd.overlap import allgather_bytes
out = allgather_bytes(b"hello round " * 1000, 1, dev)
assert out == [b"hello round " * 1000]
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("nccl one-rank exchange ok")
