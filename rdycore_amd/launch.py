"""One process per GPU, started from a plain command.

RDycore is launched as one MPI rank per GCD (`srun -n8`, docs/common/installation.md:346-350).
There is no MPI launcher in this image, so `bench.py --gpus N` (and the tests) start their ranks
through this module: N fresh child processes of the same script with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set, exactly the environment `torch.distributed.run` would give them.

The parent never touches the GPU (no HIP call, no `torch.cuda.*`): a process that has initialised the
device must not exec or fork workers, and the parent has no use for a context anyway.  Rank 0's stdout
is relayed line by line (the bench's single JSON line); every rank's stderr passes through.  If a rank
fails, the others are terminated by PID and the worst return code is returned.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_environment(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update({
        "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
    })
    # everything else -- HSA_ENABLE_IPC_MODE_LEGACY included, which RCCL's intra-node transport depends on -- is the
    # caller's environment passed through unchanged: no default is invented here
    return env


def launch_ranks(world: int, argv: Sequence[str], timeout: Optional[float] = None, out=None, keep=None) -> int:
    """Runs `argv` (a full command line) once per rank; returns the worst return code.
    Rank 0's stdout goes to `out` (default: this process's stdout); with `keep` (a predicate on a line) only the lines it
    accepts do, the rest (library chatter such as gloo's connection notes) goes to stderr."""
    if world < 1:
        raise ValueError("world must be >= 1")
    out = sys.stdout if out is None else out
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(world):
        procs.append(subprocess.Popen(list(argv), env=rank_environment(r, world, port),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))

    def relay(p):
        for line in p.stdout:
            dst = out if keep is None or keep(line) else sys.stderr
            dst.write(line)
            dst.flush()

    t = threading.Thread(target=relay, args=(procs[0],), daemon=True)
    t.start()
    deadline = None if timeout is None else time.time() + timeout
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
                for q in alive:          # a rank failed: the others would wait in a collective for ever
                    q.terminate()
        if deadline is not None and time.time() > deadline and alive:
            worst = worst or 124
            for q in alive:
                q.terminate()
            deadline = None
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    t.join(timeout=5)
    return worst
