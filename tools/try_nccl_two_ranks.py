#!/usr/bin/env python3
"""Does RCCL accept two ranks on ONE GPU?  (The development box has one GPU; if it does, the nccl collectives of dist.py can
run between two processes there instead of through gloo.)  Launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/try_nccl_two_ranks.py
Prints one line per rank: the outcome of init + all_reduce + in-place all_gather_into_tensor + reduce_scatter_tensor."""
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    t = torch.full((1024,), float(rank + 1), device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    flat = torch.zeros(2048, device="cuda")
    flat[rank * 1024:(rank + 1) * 1024] = rank + 1
    dist.all_gather_into_tensor(flat, flat[rank * 1024:(rank + 1) * 1024])
    src = torch.arange(2048, device="cuda", dtype=torch.float32)
    dist.reduce_scatter_tensor(src[rank * 1024:(rank + 1) * 1024], src)
    torch.cuda.synchronize()
    print(f"rank {rank}: nccl with two ranks on one GPU WORKS: all_reduce -> {t[0].item()}, all_gather -> {flat[0].item()},{flat[-1].item()}, "
          f"reduce_scatter -> {src[rank * 1024].item()}", flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: nccl with two ranks on one GPU FAILS: {type(e).__name__}: {str(e)[:300]}", flush=True)
    sys.exit(0)
