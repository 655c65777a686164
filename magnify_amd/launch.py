"""Self-starting multi-rank runs: one child process per GPU, started by a parent that never touches
the GPU (a process that has initialised HIP must not be replaced or forked on this pool) -- it does not even count
the GPUs: whether the ranks have a GPU each or share one is decided by every rank for itself
(``magnify_amd.distributed._share_gpu``), the only GPU count this file can give comes from sysfs.

``spawn_ranks(argv, n)`` runs ``python argv...`` n times with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in the environment (what ``torch.distributed.run`` would set; the children
call ``magnify_amd.distributed.init_from_env``), relays rank 0's stdout and returns the worst exit
code.  The reference has no launcher (it is single-process).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus() -> int:
    """GPUs of this node as the kernel driver lists them (KFD topology in sysfs: nodes with SIMDs), narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES; -1 if sysfs has no answer.  No HIP, no torch: nothing is initialised."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(root)
    except OSError:
        return -1
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(root, node, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            n = min(n, len([x for x in os.environ[var].split(",") if x.strip()]))
    return n


def launched_by_torchrun(env=None) -> bool:
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


def rank_env(rank: int, world: int, port: int, share_gpu: bool | None, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MG_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    if share_gpu is not None:  # (None: every rank finds out for itself, distributed._share_gpu)
        env["MG_SHARE_GPU"] = "1" if share_gpu else "0"
        if share_gpu:
            # fewer GPUs than ranks (the one-GPU test box): every rank on cuda:0, gloo between them
            env.setdefault("MG_DIST_BACKEND", "gloo")
    return env


def spawn_ranks(argv, n: int, share_gpu: bool | None = None, timeout: float | None = None, out=None) -> int:
    """Start ``n`` ranks of ``[sys.executable] + argv``; rank 0's stdout is relayed line by line to
    ``out`` (default ``sys.stdout``), the other ranks' stdout goes to stderr.  If a rank fails the
    others are terminated (their own PIDs only).  Returns 0 or the first non-zero exit code."""
    out = sys.stdout if out is None else out
    port = free_port()
    procs = []
    for rank in range(n):
        env = rank_env(rank, n, port, share_gpu)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=subprocess.PIPE,
                                      stderr=None, text=True, bufsize=1))

    def relay(p, dst):
        for line in p.stdout:
            # (gloo announces its connections on stdout: not part of rank 0's report)
            (sys.stderr if line.startswith("[Gloo]") else dst).write(line)
            dst.flush()

    threads = [threading.Thread(target=relay, args=(p, out if r == 0 else sys.stderr), daemon=True)
               for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    rc = 0
    try:
        pending = set(range(n))
        import time

        deadline = None if timeout is None else time.monotonic() + timeout
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (deadline is not None and time.monotonic() > deadline):
                if rc == 0:
                    rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        for t in threads:
            t.join(timeout=5)
    return rc
