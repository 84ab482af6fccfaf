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


# bind-race messages only: "Connection reset" / "Connection refused" are also what the survivors print
# when a peer died AFTER the rendezvous -- a worker fault, which must not be retried
_RENDEZVOUS_MARKERS = ("Address already in use", "EADDRINUSE", "The server socket has failed")


def is_rendezvous_error(exc):
    """True only for a lost race on the TCP rendezvous port (a worker fault must NOT be retried)."""
    msg = str(exc)
    return any(m in msg for m in _RENDEZVOUS_MARKERS)


def spawn_with_port_retry(spawn_once):
    """Runs spawn_once() (an mp.spawn call on a fresh port); retries ONCE, and only when the failure
    is the rendezvous itself -- any other worker exception propagates from the first attempt."""
    try:
        return spawn_once()
    except Exception as e:  # noqa: BLE001 -- filtered just below
        if not is_rendezvous_error(e):
            raise
    return spawn_once()
