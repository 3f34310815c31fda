"""One process per GPU: the launcher behind ``bench.py --gpus N`` and the multi-rank tests.

The path shards by independent column ranges (reference: the static column split of
``algorithm/parallel/rnea.hpp:74-82``); rank r owns slice r, its own model table, workspace and
stream.  Ranks meet only in ``torch.distributed`` barriers around timed regions.

``spawn_ranks`` starts N child processes of a script with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set, exactly what ``python -m torch.distributed.run`` would give them.
The parent never touches the GPU (it must not: a process that initialised HIP may not be replaced
or forked into GPU work) -- this module imports neither torch nor the HIP library.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def under_launcher() -> bool:
    """True inside a rank started by torchrun or by :func:`spawn_ranks`."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    return env


def column_slice(B: int, rank: int, world: int):
    """Contiguous column range [lo, hi) of rank `rank` out of `world` (SURVEY 8e:
    [k*B/G, (k+1)*B/G))."""
    return (rank * B) // world, ((rank + 1) * B) // world


def spawn_ranks(world: int, argv: Sequence[str], env: Optional[Dict[str, str]] = None,
                timeout: Optional[float] = None) -> int:
    """Run ``python argv...`` as `world` ranks on this node; rank 0 inherits stdout, every rank
    inherits stderr.  Returns the largest exit code (and kills the others if one rank fails)."""
    if world < 1:
        raise ValueError("world size must be >= 1")
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(world):
        procs.append(subprocess.Popen(
            [sys.executable] + list(argv), env=rank_env(r, world, port, env),
            stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                try:
                    code = p.wait(timeout=0.1)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = max(rc, code if code > 0 else 1)
                    for o in pending:  # a failed rank leaves the others stuck in a barrier
                        o.terminate()
            if deadline is not None and pending and time.monotonic() > deadline:
                return 124
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc
