"""torch.distributed (gloo) implementations of the library's host-collective callbacks, shared by
the CPU emulation test and the 2-process GPU test; and the same callbacks for ranks that are THREADS
of one process (ThreadGroup) -- a GPU box admits six processes on its card, threads are not counted,
so that is how the 8-rank partition runs the real kernels."""
import threading

import numpy as np
import torch
import torch.distributed as dist


class ThreadGroup:
    """Host collectives among `world` threads of this process: every rank leaves its staging buffer in a
    slot, a barrier, everybody reduces in rank order (the same bits on every rank) or copies the other
    ranks' segments, a barrier.  A rank that fails aborts the barrier, so nobody hangs."""

    def __init__(self, world, timeout=600.0):
        self.world = world
        self.timeout = timeout
        self.bar = threading.Barrier(world)
        self.slots = [None] * world

    def allreduce(self, rank):
        def f(arr, op):
            self.slots[rank] = arr
            self.bar.wait(self.timeout)
            res = self.slots[0].copy()
            for r in range(1, self.world):
                res = np.maximum(res, self.slots[r]) if op == 1 else res + self.slots[r]
            self.bar.wait(self.timeout)  # everybody has read every slot
            arr[:] = res
        return f

    def allgatherv(self, rank):
        def f(arr, offs, rk):
            self.slots[rank] = arr
            self.bar.wait(self.timeout)
            for r in range(self.world):
                if r != rank and offs[r + 1] > offs[r]:
                    arr[int(offs[r]):int(offs[r + 1])] = self.slots[r][int(offs[r]):int(offs[r + 1])]
            self.bar.wait(self.timeout)
        return f

    def alltoallv(self, rank):
        def f(send, soffs, recv, roffs, rk):
            self.slots[rank] = (send, soffs.copy())
            self.bar.wait(self.timeout)
            for p in range(self.world):
                n = int(roffs[p + 1] - roffs[p])
                if n:
                    ps, po = self.slots[p]
                    assert int(po[rank + 1] - po[rank]) == n  # the two plans agree on every count
                    recv[int(roffs[p]):int(roffs[p + 1])] = ps[int(po[rank]):int(po[rank + 1])]
            self.bar.wait(self.timeout)
        return f

    def attach(self, G, rank, neighbour=True):
        G.comm_init_callbacks(rank, self.world, self.allreduce(rank), self.allgatherv(rank),
                              self.alltoallv(rank) if neighbour else None)

    def run(self, target):
        """target(rank) -> result, one thread per rank; re-raises the first failure."""
        out, err = [None] * self.world, []

        def body(rank):
            try:
                out[rank] = target(rank)
            except BaseException as e:  # noqa: BLE001 -- reported below
                err.append(e)
                self.bar.abort()

        ts = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if err:
            raise err[0]
        return out


def allreduce(arr, op):
    t = torch.from_numpy(arr)  # shares memory: the reduction lands in the library's buffer
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)


def allgatherv(arr, offs, rank):
    t = torch.from_numpy(arr)
    for r in range(len(offs) - 1):
        if offs[r + 1] > offs[r]:
            dist.broadcast(t[int(offs[r]):int(offs[r + 1])], src=r)


def alltoallv(send, soffs, recv, roffs, rank):
    """Neighbour exchange over gloo: one isend / irecv pair per non-empty span."""
    world = len(soffs) - 1
    ts, tr = torch.from_numpy(send), torch.from_numpy(recv)
    reqs = []
    for p in range(world):
        if roffs[p + 1] > roffs[p]:
            reqs.append(dist.irecv(tr[int(roffs[p]):int(roffs[p + 1])], src=p))
    for p in range(world):
        if soffs[p + 1] > soffs[p]:
            reqs.append(dist.isend(ts[int(soffs[p]):int(soffs[p + 1])].clone(), dst=p))
    for r in reqs:
        r.wait()


def attach(G, rank, world, neighbour=True):
    """Host-staged collectives over the default gloo group for graph handle G (one place for every
    multi-process test, so that a new callback is added once)."""
    G.comm_init_callbacks(rank, world, allreduce, allgatherv, alltoallv if neighbour else None)


def init(rank, world, port):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)


def finish():
    """Orderly end of a worker: everybody has left the last collective, then the group (and gloo's background
    threads) is torn down before the interpreter exits -- a worker that simply returned could die in gloo's
    thread teardown ("terminate called without an active exception"), which mp.spawn reports as a failure."""
    try:
        dist.barrier()
    finally:
        dist.destroy_process_group()


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
