"""torch.distributed (gloo) implementations of the library's host-collective callbacks, shared by
the CPU emulation test and the 2-process GPU test."""
import numpy as np
import torch
import torch.distributed as dist


def allreduce(arr, op):
    t = torch.from_numpy(arr)  # shares memory: the reduction lands in the library's buffer
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)


def allgatherv(arr, offs, rank):
    t = torch.from_numpy(arr)
    for r in range(len(offs) - 1):
        if offs[r + 1] > offs[r]:
            dist.broadcast(t[int(offs[r]):int(offs[r + 1])], src=r)


def init(rank, world, port):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
