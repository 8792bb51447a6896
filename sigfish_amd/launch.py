"""Start N ranks of a script on ONE node, one fresh child process per rank, before anything has touched the GPU.

`python bench.py --gpus N` has to produce N ranks by itself when no launcher set WORLD_SIZE.  A process that has
initialised the GPU must never be replaced or forked, so the parent here stays a pure supervisor: it imports nothing
that touches HIP, starts N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set (the same
environment `python -m torch.distributed.run` gives them), waits for all of them and returns the worst exit code.
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched_by_a_launcher():
    """True when torchrun (or this module) already set up the rank environment."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def spawn_ranks(n, argv, timeout=None, extra_env=None):
    """Run `python argv...` as n ranks; returns the worst exit code (0 = all ranks succeeded).

    A rank that dies takes the others with it (they would wait for it in the rendezvous forever otherwise); only the
    children this call started are signalled, by PID.
    """
    port = free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SFA_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes needs dmabuf IPC on this pool
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    deadline = None if timeout is None else time.monotonic() + timeout
    worst = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in alive:  # our own children, by handle
                    q.terminate()
        if deadline is not None and time.monotonic() > deadline:
            for q in alive:
                q.kill()
            return worst or 124
        time.sleep(0.05)
    return worst
